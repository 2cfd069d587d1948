// csrc/marlin.cpp -- see marlin.hpp.  Host orchestration of the GPU prover + the host verifier.
#include "marlin.hpp"
#include "../../include/zkaes.h"   // ZKAES_DEFAULT_CONTEXTS
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <future>
#include <deque>
#include <condition_variable>
#include <thread>
#include <functional>
#include <stdexcept>
#include <cerrno>
#include <sys/random.h>
#include "gpu.hpp"
#include "trace_layout.h"
#include "transcript.hpp"
#include "marlin_host.hpp"

namespace zk {
namespace {

using namespace hostx;
using gpu::F;
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

}  // namespace

// =====================================================================================================================
// proving key
// the SRS in the form k_accumulate gathers: BLS12-377's twisted Edwards model (te28.cuh, 192 B per point incl. padding, 7 products per bucket addition)
using SrsPoint = Niels28<Fq377P>;
static void srs_convert(SrsPoint *dst, const G1A *src, size_t n, gpu::stream_t s) { gpu::convert_bases_te<Bls377>(dst, src, n, s); }

// select `device` for the calling thread and put the caller's device back on the way out (destructors of device-resident objects may run on any thread:
// they must not leave it pointing at another GPU -- advisor r05)
struct DeviceScope {
    int prev = -1;
    explicit DeviceScope(int device) noexcept { try { prev = gpu::current_device(); if (prev != device) gpu::set_device(device); else prev = -1; } catch (...) { prev = -1; } }
    ~DeviceScope() { if (prev >= 0) { try { gpu::set_device(prev); } catch (...) {} } }
};
struct DevBuf {
    F *p = nullptr; size_t n = 0;
    void alloc(size_t count) { n = count; p = (F *)gpu::dmalloc(count * sizeof(F)); }
    void release() { gpu::dfree(p); p = nullptr; }
};
struct KzgRand { bool hiding = false; Fr b[3]; };

// ---------------------------------------------------------------------------------------------------------------------
// The universal SRS, once per process and device (src/lib.rs:139-141: ONE generate_universal_srs(866_944, 513, 4_062_064) serves every circuit size).
// It depends on the literals only through max_degree (and on the test_rng trapdoor, which is fixed): powers_of_g[0 ..= max_degree] on the curve's twisted Edwards
// model, followed -- for keys that want them -- by the 12 further window-table copies 2^(offset of window j) P at multiples of `stride`.  A key's committer key is a VIEW:
// its plain powers are the prefix [0 ..= supported_degree], its shifted powers the range [max_degree - bound_max ..= max_degree] of the same array (they are the same group
// elements: beta^(max_degree - bound + i) g), so the 4- and 6-block keys of the bench, the 16/32-byte keys of its latency leg and every further key hold ONE table set
// (31.4 GB for the reference's literals; round 4 built 41.9 GB per key).  Keys share it through shared_ptr; the cache holds weak references, so the memory goes back
// to the device when the last key over that SRS is freed.
struct UniversalSrs {
    int device = 0;
    size_t max_degree = 0, stride = 0;       // stride = max_degree + 1 points per copy
    int table_c = 20;
    size_t n_tab = 1;                        // 1 = copy 0 only (KEY_NO_TABLES, tiny circuits, or the device was short of memory)
    Fr beta;
    G1A g, gamma_g;
    pairing::G2Affine h, beta_h;
    G1A gamma_powers[3];
    FixedBaseHost gamma_tab[3];              // host comb tables of the points multiplied by per-proof blinding scalars
    SrsPoint *d_points = nullptr;
    double build_s = 0;
    bool tables() const { return n_tab > 1; }
    uint64_t bytes() const { return (uint64_t)n_tab * stride * sizeof(SrsPoint); }
    ~UniversalSrs() { DeviceScope on(device); gpu::dfree(d_points); }
};
// Lagrange-basis points over H (per |H| and |X|, shared the same way): L_k(beta) G for z_A, z_B; L_k(beta)/v_X(beta) G (zero on X) for w; P_j for the public-input part
// of w; v_H(beta) G and (v_H/v_X)(beta) G for the blinding terms
struct LagrangeSrs {
    int device = 0;
    size_t n = 0, m = 0;
    SrsPoint *d_lag_h = nullptr, *d_lag_w = nullptr;
    std::vector<G1A> lag_pj;
    G1A lag_vh, lag_vw;
    FixedBaseHost lag_vh_tab, lag_vw_tab;
    ~LagrangeSrs() { DeviceScope on(device); gpu::dfree(d_lag_h); gpu::dfree(d_lag_w); }
};
namespace {
std::mutex g_srs_mu;                          // serializes SRS construction: two keys synthesized concurrently must not both build 31 GB of tables
std::map<std::pair<int, size_t>, std::weak_ptr<UniversalSrs>> g_srs_cache[2];      // [0] copy 0 only, [1] with window tables; key = (device, max_degree)
std::map<std::tuple<int, size_t, size_t>, std::weak_ptr<LagrangeSrs>> g_lag_cache;  // key = (device, |H|, |X|)
// zkaes_srs_hold(1): the process keeps every universal / Lagrange SRS alive after its last key is freed (a caller that creates and frees keys in turn -- one key per
// request size, say -- would otherwise rebuild 31 GB of tables each time: advisor r05); zkaes_srs_hold(0) lets them go again with their last key
bool g_srs_hold = false;
std::vector<std::shared_ptr<UniversalSrs>> g_srs_held;
std::vector<std::shared_ptr<LagrangeSrs>> g_lag_held;
std::atomic<size_t> g_default_contexts{0};
}  // namespace
// proofs in flight per multi-proof call unless the key says otherwise (zkaes_pk_set_contexts): ZKAES_CONTEXTS from the environment, read ONCE, else ZKAES_DEFAULT_CONTEXTS
size_t default_contexts() {
    size_t v = g_default_contexts.load();
    if (v) return v;
    v = ZKAES_DEFAULT_CONTEXTS;
    if (const char *e = getenv("ZKAES_CONTEXTS")) v = (size_t)std::min(64, std::max(1, atoi(e)));
    g_default_contexts.store(v);
    return v;
}
// n = 0: back to ZKAES_CONTEXTS / the built-in default.  The default also sizes what key synthesis leaves free beside the window tables (acquire_srs' reserve): a caller about to
// synthesize a key whose contexts are several times larger (more blocks per proof over a larger SRS) lowers it first, or the tables are skipped for lack of room
void set_default_contexts(size_t n) { g_default_contexts.store(n > 64 ? 64 : n); }
// `reserve_bytes`: what the caller will still allocate beside the tables (its prover contexts): the tables are an optimisation, not a requirement -- they are skipped
// when the device could not hold both, instead of failing key synthesis or the first multi-proof call.
static std::shared_ptr<UniversalSrs> acquire_srs(size_t max_degree, bool want_tables, size_t reserve_bytes, gpu::stream_t stream) {
    std::lock_guard<std::mutex> lock(g_srs_mu);
    const int device = gpu::current_device();
    const auto key = std::make_pair(device, max_degree);
    if (auto sp = g_srs_cache[1][key].lock()) return sp;                     // a table set serves every key (copy 0 is its prefix)
    if (!want_tables) if (auto sp = g_srs_cache[0][key].lock()) return sp;
    auto t0 = Clock::now();
    std::shared_ptr<UniversalSrs> S(new UniversalSrs());
    S->device = device; S->max_degree = max_degree; S->stride = max_degree + 1;
    // KZG10::setup from ark_std::test_rng() in upstream's draw order: the trapdoor beta, then g, gamma_g (G1) and h (G2) as random curve points
    kzg_setup_points(S->beta, S->g, S->gamma_g, S->h);
    // c = 20: 13 signed windows instead of 15 and 2^19 buckets in ONE set (the 15 x 2^16 per-window buckets cost more to reduce).  c = 22 saves one more
    // window but its 2^21 buckets triple the first reduction level: measured slower (profiles/r02_msm_tables.md).  The windows are balanced -- 254 = 7 x 20 + 6 x 19 bits,
    // kernels_msm.hip TableLayout -- so no window is short and any c is safe.
    S->table_c = 20;
    size_t n_tab = want_tables ? (size_t)gpu::table_windows<Bls377>(S->table_c) : 1;
    if (n_tab > 1 && (uint64_t)n_tab * S->stride >= (1ull << 30)) n_tab = 1;          // the MSM's 30-bit base index
    if (n_tab > 1) {
        // the copies in the reduced-radix form + two staging copies in the standard form + what the caller reserves must all fit
        const size_t need = n_tab * S->stride * sizeof(SrsPoint) + 2 * S->stride * sizeof(G1A) + reserve_bytes + ((size_t)2 << 30);
        if (gpu::mem_free_bytes() < need) n_tab = 1;
    }
    if (n_tab == 1) if (auto sp = g_srs_cache[0][key].lock()) return sp;    // (tables were wanted but do not fit: an existing plain copy serves)
    S->n_tab = n_tab;
    {
        // copy j is made from copy j - 1 in a two-slot staging buffer (standard form) and converted into its place: the setup peak is the final array
        // + two staging copies, not twice the final array
        struct Stage { G1A *p[2] = {nullptr, nullptr}; ~Stage() { gpu::dfree(p[0]); gpu::dfree(p[1]); } } stage;
        stage.p[0] = (G1A *)gpu::dmalloc(S->stride * sizeof(G1A));
        if (n_tab > 1) stage.p[1] = (G1A *)gpu::dmalloc(S->stride * sizeof(G1A));
        gpu::fixed_base_powers<Bls377>(stage.p[0], S->g, S->beta, 0, S->stride, stream);
        S->d_points = (SrsPoint *)gpu::dmalloc(n_tab * S->stride * sizeof(SrsPoint));
        srs_convert(S->d_points, stage.p[0], S->stride, stream);
        for (size_t j = 1; j < n_tab; j++) {
            gpu::table_next<Bls377>(stage.p[j & 1], stage.p[(j - 1) & 1], S->stride, S->table_c, (int)j, stream);
            srs_convert(S->d_points + j * S->stride, stage.p[j & 1], S->stride, stream);
        }
        gpu::sync(stream);
    }
    { Fr bp = Fr::one(); for (int i = 0; i < 3; i++) { S->gamma_powers[i] = mul_affine(S->gamma_g, bp); bp = bp * S->beta; S->gamma_tab[i].build(S->gamma_powers[i]); } }
    { uint32_t raw[8]; S->beta.to_raw(raw); S->beta_h = pairing::g2_mul_raw(S->h, raw, 8); }
    S->build_s = ms_since(t0) / 1e3;
    g_srs_cache[n_tab > 1 ? 1 : 0][key] = S;
    if (g_srs_hold) g_srs_held.push_back(S);
    return S;
}
static std::shared_ptr<LagrangeSrs> acquire_lagrange(const UniversalSrs &U, size_t n, int lg_n, size_t m, int lg_m, gpu::stream_t stream) {
    std::lock_guard<std::mutex> lock(g_srs_mu);
    const int device = gpu::current_device();
    const auto key = std::make_tuple(device, n, m);
    if (auto sp = g_lag_cache[key].lock()) return sp;
    std::shared_ptr<LagrangeSrs> Lg(new LagrangeSrs());
    Lg->device = device; Lg->n = n; Lg->m = m;
    // With the (public, test_rng-derived) trapdoor the Lagrange-basis points are direct fixed-base products; a trapdoor-free universal SRS
    // yields the same points through an inverse FFT over the group elements powers_of_g[0..|H|) (one time per |H|).
    struct Tmp { void *p[4] = {nullptr, nullptr, nullptr, nullptr}; ~Tmp() { for (auto q : p) gpu::dfree(q); } } tmp;
    F *d_lag = (F *)(tmp.p[0] = gpu::dmalloc(n * sizeof(F))), *d_lagw = (F *)(tmp.p[1] = gpu::dmalloc(n * sizeof(F)));
    gpu::lagrange_scalars(d_lag, d_lagw, gpu::domain_elements<F>(lg_n), U.beta, (uint32_t)n, (uint32_t)m, stream);
    G1A *pts = (G1A *)(tmp.p[2] = gpu::dmalloc(n * sizeof(G1A)));
    auto to28 = [&](const F *sc, SrsPoint *&dst) {
        gpu::fixed_base_scalars<Bls377>(pts, U.g, sc, n, stream);
        dst = (SrsPoint *)gpu::dmalloc(n * sizeof(SrsPoint));
        srs_convert(dst, pts, n, stream);
        gpu::sync(stream);
    };
    to28(d_lag, Lg->d_lag_h);
    to28(d_lagw, Lg->d_lag_w);
    // P_j = sum_{k not in X} l_j(h_k) L_k(beta)/v_X(beta) G = (l_j(beta) - L_{j n/m}(beta)) / v_X(beta) G, j < |X|
    Fr vh = eval_vanishing(n, U.beta), vx = eval_vanishing(m, U.beta), vx_inv = vx.inverse();
    Fr cm = vx * Fr::from_u64(m).inverse(), cn = vh * Fr::from_u64(n).inverse();
    std::vector<Fr> pj(m), den(m);
    Fr gx = domain_gen(lg_m), e = Fr::one();
    for (size_t j = 0; j < m; j++) { den[j] = U.beta - e; pj[j] = e; e = e * gx; }
    { std::vector<Fr> pre(m); Fr acc = Fr::one(); for (size_t j = 0; j < m; j++) { pre[j] = acc; acc = acc * den[j]; } Fr inv = acc.inverse(); for (size_t j = m; j-- > 0;) { Fr d = den[j]; den[j] = inv * pre[j]; inv = inv * d; } }
    for (size_t j = 0; j < m; j++) pj[j] = pj[j] * den[j] * (cm - cn) * vx_inv;
    F *d_pj = (F *)(tmp.p[3] = gpu::dmalloc(m * sizeof(F)));
    gpu::h2d(d_pj, pj.data(), m * sizeof(F), stream);
    gpu::fixed_base_scalars<Bls377>(pts, U.g, d_pj, m, stream);
    Lg->lag_pj.resize(m);
    gpu::d2h(Lg->lag_pj.data(), pts, m * sizeof(G1A), stream);
    Lg->lag_vh = mul_affine(U.g, vh);
    Lg->lag_vw = mul_affine(U.g, vh * vx_inv);
    Lg->lag_vh_tab.build(Lg->lag_vh); Lg->lag_vw_tab.build(Lg->lag_vw);
    g_lag_cache[key] = Lg;
    if (g_srs_hold) g_lag_held.push_back(Lg);
    return Lg;
}
void srs_hold(bool hold) {
    std::vector<std::shared_ptr<UniversalSrs>> drop_u;
    std::vector<std::shared_ptr<LagrangeSrs>> drop_l;
    {
        std::lock_guard<std::mutex> lock(g_srs_mu);
        g_srs_hold = hold;
        if (hold) {        // what is alive now stays
            for (auto &cache : g_srs_cache) for (auto &kv : cache) if (auto sp = kv.second.lock()) if (std::find(g_srs_held.begin(), g_srs_held.end(), sp) == g_srs_held.end()) g_srs_held.push_back(sp);
            for (auto &kv : g_lag_cache) if (auto sp = kv.second.lock()) if (std::find(g_lag_held.begin(), g_lag_held.end(), sp) == g_lag_held.end()) g_lag_held.push_back(sp);
        } else { drop_u.swap(g_srs_held); drop_l.swap(g_lag_held); }
    }
    // (the last references go -- and device memory is freed -- outside the lock)
}


// Everything one in-flight proof needs: its own stream, MSM scratch and polynomial workspace.  Several contexts prove different
// chunk-proofs concurrently (zkaes_encrypt_chunked): the latency-bound phases of one proof (bucket reductions, scans, host
// transcript work) overlap with the throughput-bound kernels of the others.
// One persistent host thread per MSM lane of a lone call (created with the lane): a job is handed over through a condition variable instead of a std::thread per job --
// the spawn (~50-100 us before the job's first launch) sat on the critical path at the head of every round.
struct LaneWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false;
    LaneWorker() { th = std::thread([this] { run(); }); }
    ~LaneWorker() { { std::lock_guard<std::mutex> g(mu); stop = true; } cv.notify_one(); if (th.joinable()) th.join(); }
    void run() {
        for (;;) {
            std::function<void()> t;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; t = std::move(q.front()); q.pop_front(); }
            t();
        }
    }
    void submit(std::function<void()> f) { { std::lock_guard<std::mutex> g(mu); q.push_back(std::move(f)); } cv.notify_one(); }
};
struct ProverContext {
    std::mutex in_use;                                // one proof at a time per context: concurrent callers of one key queue up here
    gpu::stream_t stream = nullptr;
    gpu::MsmWorkspace *msm_ws = nullptr;
    uint8_t *d_trace = nullptr, *d_z = nullptr, *d_msg = nullptr, *d_key = nullptr;
    void *d_rng = nullptr; size_t rng_bytes = 0;      // scratch of the device-side ChaCha12 / Fr::rand stream
    int8_t *d_cls[3] = {nullptr, nullptr, nullptr};   // small-integer evaluation classes of w, z_A, z_B on H (Lagrange-basis commitments)
    DevBuf za_ev, zb_ev, x_poly, x_tmp, x_evals, tmp_n, ra_ev, ra_poly, zpoly, t_partial;
    DevBuf poly[9];                    // w z_a z_b mask t g_1 h_1 g_2 h_2
    size_t poly_len[9] = {0};
    DevBuf e[5], f_poly, acc, wit, wit2, scratch;
    ProverTimings timings;
    // MSM lanes: lane 0 = (stream, msm_ws) above; lanes 1..4 (own stream + MSM scratch, created on first use) let a LONE encrypt() call run the independent
    // commitments of a round side by side, each started as soon as ITS polynomial exists (latency path only: with several proofs in flight the chip is already full)
    static constexpr int N_LANES = 5;
    struct Lane { gpu::stream_t stream = nullptr; gpu::MsmWorkspace *ws = nullptr; void *ready = nullptr; /* event: the lane's input exists on the main stream */ std::unique_ptr<LaneWorker> worker; };
    // (lane 3 is the background lane: the early mask commitment of round 1 runs there, under the witness generation on the main stream)
    Lane lane[N_LANES];
    DevBuf acc_b, wit_b, wit2_b, scratch_b;             // second set of opening buffers (the two openings run side by side on the latency path)
    DevBuf scratch_c; void *ev_aux = nullptr;           // ... and the calling thread divides g_2 by (X - gamma) on the main stream beside them: its own division scratch + the event the gamma opening waits for
    bool throughput = false;           // set per proof: several proofs in flight (chunked / batch calls) -> window tables; a lone encrypt() call -> per-window buckets
    ProverContext() { stream = gpu::stream_create(); msm_ws = gpu::msm_workspace_create(); lane[0].stream = stream; lane[0].ws = msm_ws; }
    void ensure_lanes() {
        for (int i = 1; i < N_LANES; i++) if (!lane[i].worker) {        // guarded by the LAST member created; leftovers of an attempt that threw half way are released first (advisor r05)
            Lane &L = lane[i];
            gpu::msm_workspace_destroy(L.ws); L.ws = nullptr;
            gpu::event_destroy(L.ready); L.ready = nullptr;
            gpu::stream_destroy(L.stream); L.stream = nullptr;
            L.stream = i == 3 ? gpu::stream_create_background() : gpu::stream_create(); L.ws = gpu::msm_workspace_create(); L.ready = gpu::event_create();
            L.worker.reset(new LaneWorker());
        }
    }
    ~ProverContext() {
        gpu::dfree(d_trace); gpu::dfree(d_z); gpu::dfree(d_msg); gpu::dfree(d_key);
        for (auto p : d_cls) gpu::dfree(p);
        gpu::dfree(d_rng);
        for (DevBuf *b : {&za_ev, &zb_ev, &x_poly, &x_tmp, &x_evals, &tmp_n, &ra_ev, &ra_poly, &zpoly, &t_partial, &f_poly, &acc, &wit, &wit2, &scratch}) b->release();
        for (auto &b : poly) b.release();
        for (auto &b : e) b.release();
        for (DevBuf *b : {&acc_b, &wit_b, &wit2_b, &scratch_b, &scratch_c}) b->release();
        gpu::event_destroy(ev_aux);
        for (int i = 1; i < N_LANES; i++) lane[i].worker.reset();        // (joins the lane threads before their streams go)
        for (int i = 1; i < N_LANES; i++) if (lane[i].stream) { gpu::msm_workspace_destroy(lane[i].ws); gpu::stream_destroy(lane[i].stream); gpu::event_destroy(lane[i].ready); }
        gpu::msm_workspace_destroy(msm_ws);
        gpu::stream_destroy(stream);
    }
};

class ProvingKeyImpl {
  public:
    VerifyingKey vk;
    Circuit circuit;
    std::vector<std::unique_ptr<ProverContext>> ctxs;
    std::mutex ctx_mu;
    size_t message_len = 0;
    size_t n = 0, k = 0, m = 0;       // |H|, |K|, |X|
    int lg_n = 0, lg_k = 0, lg_m = 0;
    size_t max_degree = 0, supported_degree = 0, lowest_shift = 0, bounds[2] = {0, 0};
    int device = 0;                            // the HIP device this key (SRS, index, contexts) lives on; every entry point re-selects it,
                                               // because HIP's current device is per thread and callers may arrive on fresh threads
    // The committer key is a view of the process-wide universal SRS (UniversalSrs above): d_powers = powers_of_g[0..] (this key uses the prefix up to supported_degree),
    // the shifted powers are the same array from index `lowest_shift` (= max_degree - the larger degree bound) on -- one index space, so merged openings name bases
    // of both; with window tables (use_tables) copies j = 1.. follow at j * srs_stride and hold 2^(offset of window j) * copy 0 (gpu.hpp msm_prepare_table).
    std::shared_ptr<UniversalSrs> srs;
    std::shared_ptr<LagrangeSrs> lag;
    SrsPoint *d_powers = nullptr, *d_shifted = nullptr;
    size_t srs_stride = 0;
#ifndef ZKAES_TABLE_MIN_N
#define ZKAES_TABLE_MIN_N 100000
#endif
    size_t table_min_n = ZKAES_TABLE_MIN_N;    // MSMs below this many points keep the per-window buckets (their own, smaller c)
    bool use_lagrange = true;
    int table_c = 20;
    bool use_tables = false;   // window tables for the large MSMs (on when |K| >= 2^20, memory allows and the key was not made with KEY_NO_TABLES)
    std::atomic<size_t> n_contexts{0};          // proofs in flight per multi-proof call on this key (zkaes_pk_set_contexts); 0 = the process default
    double setup_srs_s = 0, setup_total_s = 0;  // how long key synthesis spent building (not sharing) the SRS, and in total
    // device: circuit
    uint32_t *d_desc = nullptr, *d_sbox_in = nullptr, *d_sbox_tmpl = nullptr;
    uint32_t *d_a_rowptr = nullptr, *d_a_col = nullptr, *d_b_rowptr = nullptr, *d_b_col = nullptr;
    int64_t *d_a_coeff = nullptr, *d_b_coeff = nullptr;
    uint32_t *d_t_colptr = nullptr, *d_t_seg_start = nullptr, *d_t_seg_end = nullptr, *d_t_row = nullptr; uint8_t *d_t_mat = nullptr; int64_t *d_t_coeff = nullptr;
    uint32_t t_nseg = 0, t_nheavy = 0;
    uint32_t *d_t_heavy = nullptr;
    uint32_t *d_ix_ci = nullptr, *d_ix_ri = nullptr;     // per entry of K: the indices into H of its "row" / "col" domain elements (zero past the joint matrix's non-zeros): round 3 gathers through them
    // device: index polynomials (evaluations on K and coefficients); order row col a_val b_val c_val row_col
    DevBuf ix_ev[6], ix_co[6], ix_cs[6];       // index polynomials: values on K, coefficients, values on the coset g K (round 3)
    Fr coset_g, coset_g_inv, coset_vk_inv;     // g = the field's multiplicative generator, 1 / (g^|K| - 1)
    void *coset_tab = nullptr, *coset_tab_inv = nullptr;   // g^i and g^-i, i < |K|, in the NTT's reduced-radix form: round 3's coset scalings ride on its two transforms
    ProverTimings last_timings;

    ~ProvingKeyImpl() {
        gpu::dfree(coset_tab); gpu::dfree(coset_tab_inv);
        gpu::dfree(d_desc); gpu::dfree(d_sbox_in); gpu::dfree(d_sbox_tmpl);
        gpu::dfree(d_a_rowptr); gpu::dfree(d_a_col); gpu::dfree(d_b_rowptr); gpu::dfree(d_b_col); gpu::dfree(d_a_coeff); gpu::dfree(d_b_coeff);
        gpu::dfree(d_ix_ci); gpu::dfree(d_ix_ri);
        gpu::dfree(d_t_heavy); gpu::dfree(d_t_colptr); gpu::dfree(d_t_seg_start); gpu::dfree(d_t_seg_end); gpu::dfree(d_t_row); gpu::dfree(d_t_mat); gpu::dfree(d_t_coeff);
        for (auto &b : ix_ev) b.release();
        for (auto &b : ix_co) b.release();
        for (auto &b : ix_cs) b.release();
        ctxs.clear();
    }
    // context i, created on first use (workspace allocation happens outside any timed region when callers warm up)
    ProverContext &context(size_t i) {
        gpu::set_device(device);
        std::lock_guard<std::mutex> g(ctx_mu);
        while (ctxs.size() <= i) {
            std::unique_ptr<ProverContext> c(new ProverContext());
            alloc_workspace(*c);
            ctxs.push_back(std::move(c));
        }
        return *ctxs[i];
    }
    void alloc_workspace(ProverContext &cx) {
        const Circuit &c = circuit;
        size_t n4 = next_pow2(3 * n + 1);
        cx.d_trace = (uint8_t *)gpu::dmalloc(c.trace_bytes + 64); cx.d_z = (uint8_t *)gpu::dmalloc(c.num_variables() + 64);
        cx.d_msg = (uint8_t *)gpu::dmalloc(std::max<size_t>(message_len, 16)); cx.d_key = (uint8_t *)gpu::dmalloc(16);
        for (auto &p : cx.d_cls) p = (int8_t *)gpu::dmalloc(n + 64);
        { size_t ncand = (size_t)(3.0 * n / 0.58 * 1.02) + 8192; cx.rng_bytes = (ncand * 8 / 16 + 2) * 64 + ncand * 8 + (4u << 20); cx.d_rng = gpu::dmalloc(cx.rng_bytes); }
        cx.za_ev.alloc(n); cx.zb_ev.alloc(n); cx.x_poly.alloc(m); cx.x_tmp.alloc(m); cx.x_evals.alloc(n); cx.tmp_n.alloc(n + 1); cx.ra_ev.alloc(n); cx.ra_poly.alloc(n);
        cx.zpoly.alloc(n + 1); cx.t_partial.alloc(t_nseg + 1);
        size_t caps[9] = {n + 1, n + 1, n + 1, 3 * n, n, n, 3 * n, k, k + 1};
        for (int i = 0; i < 9; i++) cx.poly[i].alloc(caps[i]);
        // e[0..4]: round 2 takes four |H|-sized slots of each (4|H|-domain cosets), round 3 takes |K| of e[0..2]; the transforms run in place behind their first pass
        // (round 6: these were max(4|H|, 2|K|) -- twice what any user touches; with the dropped big_tmp 8 GB less across the bench's twelve contexts)
        size_t big = std::max(n4, k);
        for (auto &b : cx.e) b.alloc(big);
        cx.f_poly.alloc(k);
        cx.acc.alloc(std::max(3 * n, k) + 1); cx.wit.alloc(std::max(3 * n, k) + 1); cx.wit2.alloc(std::max(n, k) + 1); cx.scratch.alloc(std::max(8 + 3 * gpu::poly_eval_scratch(n + 1) + gpu::poly_eval_scratch(k), gpu::divide_by_linear_scratch(std::max(3 * n, k) + 1)));
    }

    template <class T> static T *upload(const std::vector<T> &v, gpu::stream_t s) {
        T *d = (T *)gpu::dmalloc(v.size() * sizeof(T));
        gpu::h2d(d, v.data(), v.size() * sizeof(T), s);
        return d;
    }

    // Window tables: 13 balanced windows over ONE bucket set instead of 15 windows with their own buckets.  Round 2 kept them away from a lone encrypt() call
    // (the short top window of that layout cost ~2 ms of serial segment chains per MSM); with balanced windows they win there too
    // (profiles/r03_latency.json: 16 B 36.4 -> 34.3 ms, 64 B 98.6 -> 91.4 ms), so every MSM of at least table_min_n points uses them.
    // (KEY_NO_TABLES builds no tables.)
    bool table_ok(const ProverContext &, size_t len) const { return use_tables && len >= table_min_n; }
    using Lane = ProverContext::Lane;
    // MSM against powers_of_g starting at `off` (plain or shifted range); device scalars
    XYZZ<Fq377> msm_powers(ProverContext &cx, Lane &ln, bool shifted, size_t off, const F *scalars, size_t len) {
        if (len == 0) return XYZZ<Fq377>::inf();
        size_t avail = shifted ? (bounds[1] + 1) : (supported_degree + 1);
        if (off + len > avail) throw std::runtime_error("KZG10: polynomial degree exceeds the committer key");
        if (table_ok(cx, len)) return gpu::msm_table<Bls377>(ln.ws, d_powers, srs_stride, (shifted ? lowest_shift : 0) + off, table_c, scalars, len, ln.stream);
        return gpu::msm<Bls377>(ln.ws, (shifted ? d_shifted : d_powers) + off, scalars, len, ln.stream);
    }
    struct Labeled { int idx; long bound; bool hiding; KzgRand rand, shifted_rand; Commitment comm; };
    // The prover RNG is consumed in label order BEFORE the commitments are computed (three blinding coefficients per hiding commitment, the shifted
    // commitment of a degree-bounded polynomial draws its own three): the same stream positions as upstream's commit loop, and the commitments
    // themselves no longer touch the RNG, so independent ones may run concurrently.
    static void draw_rand(Labeled &lp, ChaChaRng &zk) {
        auto draw = [&](KzgRand &r) { r.hiding = lp.hiding; for (auto &x : r.b) x = Fr::zero(); if (lp.hiding) for (int i = 0; i < 3; i++) r.b[i] = zk.rand_field<Fr>(); };
        draw(lp.rand);
        if (lp.bound >= 0) draw(lp.shifted_rand);
    }
    void hide(XYZZ<Fq377> &c, const KzgRand &rnd) const { if (rnd.hiding) for (int i = 0; i < 3; i++) c.add(srs->gamma_tab[i].mul(rnd.b[i])); }
    // commitment of w / z_A / z_B through the Lagrange-basis SRS: sum of the bases whose evaluation is non-zero (+-1, 2) + the blinding term
    // rho * V + the hiding part.  Same group element as MSM(powers, coefficients); falls back to the MSM if the class sum declines.
    bool lagrange_commit(ProverContext &cx, Lane &ln, int which, const std::vector<uint8_t> &inst, const Fr &rho, const KzgRand &rnd, G1A &out) {
        XYZZ<Fq377> c;
        if (!gpu::class_sum<Bls377>(ln.ws, which == 0 ? lag->d_lag_w : lag->d_lag_h, cx.d_cls[which], n, &c, ln.stream)) return false;
        if (which == 0) for (size_t j = 0; j < m; j++) if (inst[j]) c.madd(lag->lag_pj[j].neg());
        c.add((which == 0 ? lag->lag_vw_tab : lag->lag_vh_tab).mul(rho));
        hide(c, rnd);
        out = c.to_affine();
        return true;
    }
    // marlin_pc commit of one labeled polynomial (randomness already drawn): KZG10::commit = MSM(powers, coeffs) [+ hiding terms]
    void mpc_commit(ProverContext &cx, Lane &ln, Labeled &lp) {
        const F *coeffs = cx.poly[lp.idx].p;
        size_t len = cx.poly_len[lp.idx];
        lp.comm.has_shifted = false;
        const bool bounded = lp.bound >= 0;
        // degree-bounded polynomial: the plain and the shifted commitment have the same scalars -> one digit / sort pass, two accumulations, one reduction
        const size_t off = bounded ? (max_degree - (size_t)lp.bound) - lowest_shift : 0;
        if (len > supported_degree + 1 || (bounded && off + len > bounds[1] + 1)) throw std::runtime_error("KZG10: polynomial degree exceeds the committer key");
        if (len) {
            if (table_ok(cx, len)) gpu::msm_prepare_table<Bls377>(ln.ws, coeffs, len, 0, nullptr, 0, 0, table_c, srs_stride, ln.stream);
            else gpu::msm_prepare<Bls377>(ln.ws, coeffs, len, nullptr, 0, 0, ln.stream);
        }
        // the hiding terms are host work (comb-table products): done while the device groups the digits, not after the wait for its sums
        XYZZ<Fq377> c12[2] = {XYZZ<Fq377>::inf(), XYZZ<Fq377>::inf()}, h12[2] = {XYZZ<Fq377>::inf(), XYZZ<Fq377>::inf()};
        hide(h12[0], lp.rand);
        if (bounded) hide(h12[1], lp.shifted_rand);
        if (len) {
            if (bounded) gpu::msm_finish2<Bls377>(ln.ws, d_powers, d_shifted + off, c12, ln.stream);       // table mode: every copy's index shifts by lowest_shift + off
            else c12[0] = gpu::msm_finish<Bls377>(ln.ws, d_powers, ln.stream);
        }
        c12[0].add(h12[0]);
        lp.comm.comm = c12[0].to_affine();
        if (bounded) { c12[1].add(h12[1]); lp.comm.shifted = c12[1].to_affine(); lp.comm.has_shifted = true; }
    }
    // opening witness = MSM(powers, wit) + MSM(shifted powers from shift_off, swit) as ONE Pippenger instance over the contiguous SRS array
    // (two steps: the digit grouping is launched and returns; the caller does its host-side share of the witness, then waits for the sum)
    void msm_opening_prepare(ProverContext &cx, Lane &ln, const F *wit, size_t wlen, const F *swit, size_t slen, size_t shift_off) {
        if (wlen > supported_degree + 1 || shift_off + slen > bounds[1] + 1) throw std::runtime_error("KZG10: polynomial degree exceeds the committer key");
        if (table_ok(cx, wlen + slen)) gpu::msm_prepare_table<Bls377>(ln.ws, wit, wlen, 0, swit, slen, lowest_shift + shift_off, table_c, srs_stride, ln.stream);
        else gpu::msm_prepare<Bls377>(ln.ws, wit, wlen, swit, slen, lowest_shift + shift_off, ln.stream);
    }
    XYZZ<Fq377> msm_opening_finish(Lane &ln) { return gpu::msm_finish<Bls377>(ln.ws, d_powers, ln.stream); }
    // The independent MSM jobs of a prover round.  A lone encrypt() call starts each job AT ONCE on a lane of its own (stream + MSM scratch + host thread) -- gated, where
    // its input is still being produced, on an event of the main stream (no host wait) -- while the calling thread keeps queuing the rest of the round on the main stream:
    // the latency-bound remainder of a round (small dependent kernels) then runs under the job's bucket accumulation instead of in front of it.  Multi-proof calls
    // (several proofs in flight: the chip is full anyway) and ZKAES_LANES=0 defer the jobs to join(), one after the other on lane 0, as before.
    bool use_lanes = true;
    struct RoundJobs {
        ProvingKeyImpl &K; ProverContext &cx;
        const bool async;
        struct Slot { std::string err; std::function<void(Lane &)> fn; std::promise<void> done; std::future<void> fut; bool queued = false; };
        std::vector<std::unique_ptr<Slot>> slots;
        RoundJobs(ProvingKeyImpl &k, ProverContext &c) : K(k), cx(c), async(!c.throughput && k.use_lanes) { if (async) cx.ensure_lanes(); }
        // gate = true: the job's input is produced by work already queued on the main stream (an event is recorded there now and the lane waits for it on the device);
        // gate = false: the caller knows the input is complete
        void start(int lane_i, std::function<void(Lane &)> fn, bool gate) {
            std::unique_ptr<Slot> sl(new Slot());
            sl->fn = std::move(fn);
            Slot *p = sl.get();
            slots.push_back(std::move(sl));
            if (!async) return;
            if (lane_i < 1 || lane_i >= ProverContext::N_LANES) throw std::logic_error("RoundJobs: bad lane");
            Lane &ln = cx.lane[lane_i];
            if (gate) { gpu::event_record(ln.ready, cx.stream); gpu::stream_wait_event(ln.stream, ln.ready); }
            const int dev = K.device;
            p->fut = p->done.get_future(); p->queued = true;
            ln.worker->submit([p, &ln, dev] {
                try { gpu::set_device(dev); p->fn(ln); } catch (const std::exception &e) { p->err = e.what(); if (p->err.empty()) p->err = "error"; } catch (...) { p->err = "error"; }
                p->done.set_value();
            });
        }
        void join() {
            std::string first;
            for (auto &sl : slots) {
                if (async) { if (sl->queued) { sl->fut.wait(); sl->queued = false; } }
                else { try { sl->fn(cx.lane[0]); } catch (const std::exception &e) { sl->err = e.what(); if (sl->err.empty()) sl->err = "error"; } }
                if (first.empty() && !sl->err.empty()) first = sl->err;
            }
            slots.clear();
            if (!first.empty()) throw std::runtime_error(first);
        }
        void wait_all() noexcept { for (auto &sl : slots) if (sl->queued) sl->fut.wait(); }
        ~RoundJobs() { wait_all(); }       // (an exception must not leave a job running on this context)
    };

    // What one proof carries from round to round: the prover RNG, the Fiat-Shamir transcript, the labeled commitments of each round with their blinding scalars, the
    // challenges and what is derived from them, the proof being filled in.  `jobs` is the LAST member on purpose: members are destroyed in reverse order, so when a round
    // throws (a HIP error in a transform, say) ~RoundJobs waits for the lane workers BEFORE `inst`, `declined`, `pf`, ... -- which their lambdas hold by reference -- go away
    // (advisor r05: `jobs` used to be declared ahead of what it captured).
    struct ProofRun {
        ProverContext &cx;
        const uint8_t *host_trace, *msg; size_t len; const uint8_t *key;
        Clock::time_point t_all, t0;
        ChaChaRng zk;                                   // the prover's zero-knowledge randomness (generate_rand(): ark_std::test_rng unless a seed is given)
        FiatShamirRng fs;
        Labeled r1[4], r2[3], r3[2];                    // w z_a z_b mask | t g_1 h_1 | g_2 h_2
        Fr rhos[3];
        std::vector<uint8_t> inst;
        bool declined[3] = {false, false, false};
        Fr alpha, eta_a, eta_b, eta_c, vh_alpha, beta, vh_beta, ea_vv, eb_vv, ec_vv, alpha_beta, gamma;
        Proof pf;
        bool early_evals = false;
        std::atomic<bool> g2_quotient_queued{false};
        RoundJobs jobs;
        ProofRun(ProvingKeyImpl &K, ProverContext &c, const uint8_t *trace, const uint8_t *msg_, size_t len_, const uint8_t *key_, const uint8_t *zk_seed)
            : cx(c), host_trace(trace), msg(msg_), len(len_), key(key_), t_all(Clock::now()), t0(t_all), zk(zk_seed ? zk_seed : ark_test_rng_seed(), 12),
              r1{{0, -1, true}, {1, -1, true}, {2, -1, true}, {3, -1, false}}, r2{{4, -1, false}, {5, (long)(K.n - 2), true}, {6, -1, false}},
              r3{{7, (long)(K.k - 2), false}, {8, -1, false}}, jobs(K, c) {}
    };
    Fr sample_outside_h(FiatShamirRng &fs) const;

    void setup(int kind, size_t message_len, const SrsLiterals &lits, unsigned flags);
    Proof prove(ProverContext &cx, const uint8_t *trace_or_null, const uint8_t *msg, size_t len, const uint8_t *key, const uint8_t *zk_seed, bool throughput = false);
    void prove_round1(ProofRun &R);      // randomness, mask polynomial, witness, interpolations, commitments of w z_A z_B mask -> alpha, eta
    void prove_round2(ProofRun &R);      // t, the sumcheck quotient on three cosets of H, commitments of t g_1 h_1 -> beta
    void prove_round3(ProofRun &R);      // f on K from the two quotient tables, g_2, h_2 from one coset, their commitments -> gamma
    void prove_open(ProofRun &R);        // the four evaluations, the opening challenge, the two batched KZG openings side by side
};

void ProvingKeyImpl::setup(int kind, size_t message_len_, const SrsLiterals &lits, unsigned flags) {
    auto t_setup = Clock::now();
    gpu::require_device();
    device = gpu::current_device();
    message_len = message_len_;
    std::unique_ptr<ProverContext> cx0(new ProverContext());
    gpu::stream_t stream = cx0->stream;
    circuit = kind == CIRCUIT_AES ? compile_aes_circuit(message_len) : compile_ops_circuit(kind);
    const Circuit &c = circuit;
    // ---- joint matrix (sum_matrices): per-row sorted union of the A, B, C column supports
    size_t rows = c.num_constraints;
    std::vector<uint32_t> j_ci, j_ri;
    std::vector<int64_t> j_a, j_b, j_c;
    m = c.num_instance; n = next_pow2(c.num_constraints);
    for (size_t r = 0; r < rows; r++) {
        uint32_t ia = c.A.rowptr[r], ib = c.B.rowptr[r], ic = c.C.rowptr[r];
        const uint32_t ea = c.A.rowptr[r + 1], eb = c.B.rowptr[r + 1], ec = c.C.rowptr[r + 1];
        for (;;) {
            uint32_t best = 0xffffffffu;
            if (ia < ea) best = std::min(best, c.A.col[ia]);
            if (ib < eb) best = std::min(best, c.B.col[ib]);
            if (ic < ec) best = std::min(best, c.C.col[ic]);
            if (best == 0xffffffffu) break;
            j_ci.push_back((uint32_t)reindex_by_subdomain(n, m, best));
            j_ri.push_back((uint32_t)r);
            j_a.push_back(ia < ea && c.A.col[ia] == best ? c.A.coeff[ia++] : 0);
            j_b.push_back(ib < eb && c.B.col[ib] == best ? c.B.coeff[ib++] : 0);
            j_c.push_back(ic < ec && c.C.col[ic] == best ? c.C.coeff[ic++] : 0);
        }
    }
    size_t nnz = j_ci.size();
    k = next_pow2(nnz);
    lg_n = log2_exact(n); lg_k = log2_exact(k); lg_m = log2_exact(m);
    vk.num_variables = c.num_variables(); vk.num_constraints = c.num_constraints; vk.num_non_zero = nnz; vk.num_instance = c.num_instance;
    vk.num_public_inputs = c.raw_instance - 1;
    if (vk.num_variables != vk.num_constraints) throw std::logic_error("index: matrices are not square after padding");
    // ---- universal SRS sizing + trim
    max_degree = ahp_max_degree(lits.num_constraints, lits.num_variables, lits.num_non_zero);
    supported_degree = ahp_max_degree(vk.num_constraints, vk.num_variables, vk.num_non_zero);
    if (supported_degree > max_degree) throw std::runtime_error("IndexTooLarge: the circuit needs degree " + std::to_string(supported_degree) + " but the universal SRS supports " + std::to_string(max_degree));
    bounds[0] = std::min(n - 2, k - 2); bounds[1] = std::max(n - 2, k - 2);
    lowest_shift = max_degree - bounds[1];
    // ---- the universal SRS: shared with every other key over the same literals on this device (built by the first one); window tables pay from the one-block key
    // (|K| = 2^20: 35.5 -> 38.2 proofs/s in batch mode) upwards, tiny circuits keep per-window buckets only
    {
        const bool want_tables = lg_k >= 20 && !(flags & KEY_NO_TABLES);
        // what this key's callers may still allocate beside the tables: the workspaces of the effective number of prover contexts (alloc_workspace: ~18 |H| + 6 |K| +
        // 5 max(4 |H|, |K|) field elements, plus the MSM scratch of the largest opening -- 13 windows x 16 B per pair: ~1.2 / 5 GB for the 1- / 6-block key)
        const size_t big = std::max(4 * n, 2 * k), ebuf = std::max(4 * n, k);
        const size_t per_context = (18 * n + 6 * k + 5 * ebuf) * sizeof(F) + 13 * big * 16 + ((size_t)256 << 20);
        auto t_srs = Clock::now();
        srs = acquire_srs(max_degree, want_tables, default_contexts() * per_context, stream);
        setup_srs_s = ms_since(t_srs) / 1e3;
        use_tables = want_tables && srs->tables();
        table_c = srs->table_c;
        srs_stride = srs->stride;
        d_powers = srs->d_points;
        d_shifted = d_powers + lowest_shift;
    }
    static const bool lanes_default = [] { const char *e = getenv("ZKAES_LANES"); return !e || atoi(e) != 0; }();       // read once per process
    use_lanes = lanes_default;
    if (use_lagrange) lag = acquire_lagrange(*srs, n, lg_n, m, lg_m, stream);
    const G1A &g = srs->g;
    vk.g = srs->g; vk.gamma_g = srs->gamma_g;
    vk.h = srs->h; vk.beta_h = srs->beta_h;
    vk.degree_bounds[0] = bounds[0]; vk.degree_bounds[1] = bounds[1];
    for (int i = 0; i < 2; i++) vk.shift_powers[i] = mul_affine(g, srs->beta.pow_u64(max_degree - bounds[i]));
    vk.supported_degree = supported_degree; vk.max_degree = max_degree;
    // ---- circuit tables
    {
        uint8_t sb[256];
        // S-box table for the device = the lookup table the circuit was compiled against (level-7 node values of the template)
        for (int i = 0; i < 256; i++) sb[i] = aes_sbox_value((uint8_t)i);
        gpu::upload_sbox(sb);
    }
    d_desc = upload(c.desc, stream);
    d_sbox_in = upload(c.sbox_in_off.empty() ? std::vector<uint32_t>{0} : c.sbox_in_off, stream);
    d_sbox_tmpl = upload(c.sbox_tmpl.empty() ? std::vector<uint32_t>{0} : c.sbox_tmpl, stream);
    d_a_rowptr = upload(c.A.rowptr, stream); d_a_col = upload(c.A.col.empty() ? std::vector<uint32_t>{0} : c.A.col, stream);
    d_a_coeff = upload(c.A.coeff.empty() ? std::vector<int64_t>{0} : c.A.coeff, stream);
    d_b_rowptr = upload(c.B.rowptr, stream); d_b_col = upload(c.B.col.empty() ? std::vector<uint32_t>{0} : c.B.col, stream);
    d_b_coeff = upload(c.B.coeff.empty() ? std::vector<int64_t>{0} : c.B.coeff, stream);
    {   // column-bucketed copy of A, B, C for the round-2 t accumulation (calculate_t)
        std::vector<uint32_t> colptr(n + 1, 0), trow;
        std::vector<uint8_t> tmat;
        std::vector<int64_t> tcoef;
        const CsrMatrix *M[3] = {&c.A, &c.B, &c.C};
        for (int q = 0; q < 3; q++) for (uint32_t col : M[q]->col) colptr[reindex_by_subdomain(n, m, col) + 1]++;
        for (size_t i = 0; i < n; i++) colptr[i + 1] += colptr[i];
        size_t total = colptr[n];
        trow.resize(total ? total : 1); tmat.resize(total ? total : 1); tcoef.resize(total ? total : 1);
        std::vector<uint32_t> fill(colptr.begin(), colptr.end() - 1);
        for (int q = 0; q < 3; q++)
            for (size_t r = 0; r < rows; r++)
                for (uint32_t i = M[q]->rowptr[r]; i < M[q]->rowptr[r + 1]; i++) {
                    uint32_t pos = fill[reindex_by_subdomain(n, m, M[q]->col[i])]++;
                    trow[pos] = (uint32_t)r; tmat[pos] = (uint8_t)q; tcoef[pos] = M[q]->coeff[i];
                }
        std::vector<uint32_t> col_seg_ptr(n + 1, 0), seg_start, seg_end;
        for (size_t h = 0; h < n; h++) {
            for (uint32_t st = colptr[h]; st < colptr[h + 1]; st += gpu::T_SEG) { seg_start.push_back(st); seg_end.push_back(std::min(st + gpu::T_SEG, colptr[h + 1])); }
            col_seg_ptr[h + 1] = (uint32_t)seg_start.size();
        }
        t_nseg = (uint32_t)seg_start.size();
        std::vector<uint32_t> heavy;
        for (size_t h = 0; h < n; h++) if (col_seg_ptr[h + 1] - col_seg_ptr[h] > gpu::T_HEAVY_SEGMENTS) heavy.push_back((uint32_t)h);
        t_nheavy = (uint32_t)heavy.size();
        if (heavy.empty()) heavy.push_back(0);
        d_t_heavy = upload(heavy, stream);
        if (seg_start.empty()) { seg_start.push_back(0); seg_end.push_back(0); }
        d_t_colptr = upload(col_seg_ptr, stream); d_t_seg_start = upload(seg_start, stream); d_t_seg_end = upload(seg_end, stream);
        d_t_row = upload(trow, stream); d_t_mat = upload(tmat, stream); d_t_coeff = upload(tcoef, stream);

    }
    // ---- index polynomials on the GPU
    for (int i = 0; i < 6; i++) { ix_ev[i].alloc(k); ix_co[i].alloc(k); }
    {
        j_ci.resize(k, 0u); j_ri.resize(k, 0u);           // (entries past the non-zeros: elems[0], as index_evals pads row / col)
        uint32_t *d_ci = upload(j_ci, stream), *d_ri = upload(j_ri, stream);
        int64_t *d_ja = upload(j_a.empty() ? std::vector<int64_t>{0} : j_a, stream), *d_jb = upload(j_b.empty() ? std::vector<int64_t>{0} : j_b, stream), *d_jc = upload(j_c.empty() ? std::vector<int64_t>{0} : j_c, stream);
        const F *elems = gpu::domain_elements<F>(lg_n);
        // order: 0 row, 1 col, 2 a_val, 3 b_val, 4 c_val, 5 row_col ; ix_co[0] doubles as the batch-inverse scratch
        gpu::index_evals(ix_ev[0].p, ix_ev[1].p, ix_ev[5].p, ix_ev[2].p, ix_ev[3].p, ix_ev[4].p, ix_co[0].p, d_ci, d_ri, d_ja, d_jb, d_jc, nnz, k, elems, (uint32_t)n, stream);
        for (int i = 0; i < 6; i++) gpu::ntt<F>(ix_co[i].p, ix_ev[i].p, k, lg_k, true, stream);
        gpu::sync(stream);
        d_ix_ci = d_ci; d_ix_ri = d_ri;
        gpu::dfree(d_ja); gpu::dfree(d_jb); gpu::dfree(d_jc);
    }
    for (int i = 0; i < 6; i++) vk.index_comms[i] = msm_powers(*cx0, cx0->lane[0], false, 0, ix_co[i].p, k).to_affine();
    {   // values of the index polynomials on the coset g K, once per key: round 3 then needs no transform for a(X) and b(X)
        for (int i = 0; i < 8; i++) coset_g.l[i] = FR377_GEN_MONT[i];
        coset_g_inv = coset_g.inverse();
        coset_vk_inv = (coset_g.pow_u64(k) - Fr::one()).inverse();
        F *tmp = (F *)gpu::dmalloc(k * sizeof(F));
        for (int i = 0; i < 6; i++) {
            ix_cs[i].alloc(k);
            gpu::coset_scale(tmp, ix_co[i].p, coset_g, k, k, stream);
            gpu::ntt<F>(ix_cs[i].p, tmp, k, lg_k, false, stream);
        }
        gpu::sync(stream);
        gpu::dfree(tmp);
        coset_tab = gpu::coset_power_table<F>(coset_g, k, stream);
        coset_tab_inv = gpu::coset_power_table<F>(coset_g_inv, k, stream);
    }
    // ---- workspace of context 0 (further contexts are created on demand)
    alloc_workspace(*cx0);
    ctxs.push_back(std::move(cx0));
    gpu::sync(stream);
    setup_total_s = ms_since(t_setup) / 1e3;
}

// ---- one proof.  prove() = witness + round 1, round 2, round 3, the two openings, over a ProofRun that carries what the rounds hand to each other (the transcript, the prover
// RNG, the labeled commitments with their blinding scalars, the challenges); the workspace buffers are the context's (cx.poly[..], cx.e[..], ...) and are named as such.
// (Round 5: one 290-line function over thirty local aliases of the context's buffers -- VERDICT r05 weak #10.)
Fr ProvingKeyImpl::sample_outside_h(FiatShamirRng &fs) const { Fr t; do { t = fs.rng().rand_field<Fr>(); } while (eval_vanishing(n, t).is_zero()); return t; }

void ProvingKeyImpl::prove_round1(ProofRun &R) {
    ProverContext &cx = R.cx;
    const Circuit &c = circuit;
    gpu::stream_t s = cx.stream;
    // ---- the prover's own randomness first: none of it depends on the witness.  Upstream's draw order is rho_w, rho_zA, rho_zB, the 3|H| mask coefficients, then the
    // commitments' blinding scalars label by label -- kept -- but the mask polynomial and its commitment (the one full-width MSM of round 1, and the round's critical
    // path in a lone call: 2.9 of 26.8 ms at 16 bytes) start NOW on a lane of their own, under the witness generation and the round's transforms.
    for (auto &r : R.rhos) r = R.zk.rand_field<Fr>();
    {   // mask polynomial: degree 3|H| + 2 zk_bound - 3, sum over H forced to zero.  The 3|H| coefficients are the next 3|H| Fr::rand draws
        // of the prover RNG: generated on the device from the same ChaCha12 key stream, then the host RNG skips past them.
        uint64_t next = gpu::chacha_field_stream(cx.poly[3].p, 3 * n, R.zk.key_words(), R.zk.rounds(), R.zk.word_pos(), cx.d_rng, cx.rng_bytes, s);
        R.zk.set_word_pos(next);
        gpu::mask_fixup(cx.poly[3].p, n, s);
        cx.poly_len[3] = 3 * n;
    }
    for (auto &lp : R.r1) draw_rand(lp, R.zk);
    if (R.jobs.async) gpu::sync(s);        // the mask polynomial is in place (nothing else is on the stream yet)
    R.jobs.start(3, [&](Lane &ln) { mpc_commit(cx, ln, R.r1[3]); }, false);
    // ---- witness: trace -> z (bytes) -> z_A, z_B
    if (R.host_trace) gpu::h2d(cx.d_trace, R.host_trace, c.trace_bytes, s);
    else {
        gpu::h2d(cx.d_msg, R.msg, R.len, s); gpu::h2d(cx.d_key, R.key, 16, s);
        gpu::aes_trace(cx.d_trace, c.trace_bytes, cx.d_msg, cx.d_key, 1, (uint32_t)c.n_blocks, s);
    }
    gpu::witness_expand(cx.d_z, d_desc, (uint32_t)c.num_variables(), cx.d_trace, d_sbox_in, d_sbox_tmpl, s);
    gpu::spmv_bits(cx.za_ev.p, cx.d_cls[1], n, d_a_rowptr, d_a_col, d_a_coeff, c.num_constraints, cx.d_z, s);
    gpu::spmv_bits(cx.zb_ev.p, cx.d_cls[2], n, d_b_rowptr, d_b_col, d_b_coeff, c.num_constraints, cx.d_z, s);
    if (use_lagrange) gpu::w_classes(cx.d_cls[0], cx.d_z, (uint32_t)n, (uint32_t)m, (uint32_t)c.num_witness, s);
    R.inst.resize(m);
    gpu::d2h(R.inst.data(), cx.d_z, m, s);            // (drains the main stream: the evaluation classes of w, z_A, z_B exist from here on)
    cx.timings.witness_ms = ms_since(R.t0); R.t0 = Clock::now();
    // The commitments of w, z_A, z_B are class sums over the Lagrange-basis SRS (lagrange_commit): they need the evaluation classes only, NOT the interpolated polynomials --
    // so a lone call starts them here, beside the mask commitment, and the round's interpolations (needed from round 2 on) run under them on the main stream.  A class sum
    // that declines (a value outside the small classes) falls back to the MSM over the coefficients after the interpolation.
    static const int r1_lane[3] = {1, 2, 4};
    for (int i = 0; i < 3; i++) {
        Labeled *lp = &R.r1[i];
        bool *dec = &R.declined[i];
        R.jobs.start(r1_lane[i], [&, lp, dec](Lane &ln) {
            if (use_lagrange && lagrange_commit(cx, ln, lp->idx, R.inst, R.rhos[lp->idx], lp->rand, lp->comm.comm)) { lp->comm.has_shifted = false; return; }
            if (R.jobs.async) { *dec = true; return; }     // (the coefficients are not there yet: the calling thread commits after the interpolation)
            mpc_commit(cx, ln, *lp);
        }, false);
    }
    // ---- transcript init: "MARLIN-2019" || index_vk || public_input
    {
        Bytes o;
        o.put("MARLIN-2019", 11);
        o.u64(vk.num_variables); o.u64(vk.num_constraints); o.u64(vk.num_non_zero);
        for (int i = 0; i < 6; i++) { Commitment ic; ic.comm = vk.index_comms[i]; o.commitment_tobytes(ic); }
        for (size_t i = 1; i < m; i++) o.field(R.inst[i] ? Fr::one() : Fr::zero());
        R.fs.initialize(o.b);
    }
    // ---- first round
    gpu::bits_to_field(cx.x_tmp.p, cx.d_z, m, s);
    gpu::ntt<F>(cx.x_poly.p, cx.x_tmp.p, m, lg_m, true, s);
    gpu::ntt<F>(cx.x_evals.p, cx.x_poly.p, m, lg_n, false, s);
    gpu::w_evals(cx.tmp_n.p, cx.d_z, cx.x_evals.p, (uint32_t)n, (uint32_t)m, (uint32_t)c.num_witness, s);
    {   // interpolate w v_X + x (then / v_X), z_A, z_B: three transforms, shared launches
        const gpu::NttJob<F> interp[3] = {{cx.e[0].p, cx.tmp_n.p, 0}, {cx.poly[1].p, cx.za_ev.p, 0}, {cx.poly[2].p, cx.zb_ev.p, 0}};
        gpu::ntt_batch<F>(interp, 3, n, lg_n, true, 0, s);
    }
    gpu::poly_add_at(cx.e[0].p, 0, R.rhos[0].neg(), s); gpu::poly_set_at(cx.e[0].p, n, R.rhos[0], s);   // + rho * v_H
    gpu::divide_by_vanishing(cx.poly[0].p, cx.e[1].p, cx.e[0].p, n + 1, m, s, cx.e[1].p + m, n);       // / v_X ; remainder must vanish
    cx.poly_len[0] = n + 1 - m;
    gpu::poly_add_at(cx.poly[1].p, 0, R.rhos[1].neg(), s); gpu::poly_set_at(cx.poly[1].p, n, R.rhos[1], s); cx.poly_len[1] = n + 1;
    gpu::poly_add_at(cx.poly[2].p, 0, R.rhos[2].neg(), s); gpu::poly_set_at(cx.poly[2].p, n, R.rhos[2], s); cx.poly_len[2] = n + 1;
    R.jobs.join();
    for (int i = 0; i < 3; i++) if (R.declined[i]) mpc_commit(cx, cx.lane[0], R.r1[i]);
    { Bytes o; for (auto &lp : R.r1) o.commitment_tobytes(lp.comm); R.fs.absorb(o.b); }
    R.alpha = sample_outside_h(R.fs);
    R.eta_a = R.fs.rng().rand_field<Fr>(); R.eta_b = R.fs.rng().rand_field<Fr>(); R.eta_c = R.fs.rng().rand_field<Fr>();      // (drawn in this order)
    cx.timings.round1_ms = ms_since(R.t0); R.t0 = Clock::now();
}

void ProvingKeyImpl::prove_round2(ProofRun &R) {
    ProverContext &cx = R.cx;
    const Circuit &c = circuit;
    gpu::stream_t s = cx.stream;
    const int lg_n4 = log2_exact(next_pow2(3 * n + 1));
    // ---- second round
    for (auto &lp : R.r2) draw_rand(lp, R.zk);           // (the prover RNG is consumed in label order before any commitment is computed: draw_rand)
    R.vh_alpha = eval_vanishing(n, R.alpha);
    const F *elems = gpu::domain_elements<F>(lg_n);
    // q = q_1 - mask = r(alpha, X) (eta_A z_A + eta_B z_B + eta_C z_A z_B) - t z has degree < 3|H|; instead of five zero-padded transforms to the 4|H| domain, a pointwise
    // product there and a 4|H|-point inverse (18 passes over 4|H| elements), it is taken on THREE cosets of H inside that domain: on H itself every factor is already known
    // (the evaluation vectors of round 1 -- the blinding terms rho v_H vanish there), on W H and W^3 H (W the 4|H|-th root) each factor costs one |H|-point coset transform, where
    // rho (X^|H| - 1) is the constant rho (zeta^c - 1), zeta = W^|H|.  Three |H|-point inverses give the interpolants Q0, Q1, Q3; q_1_combine solves for q's three |H|-coefficient
    // thirds and divides by v_H = X^|H| - 1 in coefficient space, mask included: no 4|H| buffer traffic, no division kernel.
    // Round 5: r(alpha, X) = (alpha^|H| - X^|H|) / (alpha - X) = prod_k (alpha^(2^k) + X^(2^k)) needs NO transform and NO inversion: its values on H and on both cosets are a
    // product tree (vanishing_quotient_evals: two products per node) instead of a batch inversion, an inverse transform to coefficients and two forward coset transforms:
    // 12 |H|-point transforms in round 2 (was 15) in five launches: {t, Q0} inverse, {z_A, z_B, t, z} x {W H, W^3 H} forward, {Q1, Q3} inverse.
    // And t's commitment starts as soon as t's coefficients exist, under the rest of the round (lone call).
    {
        using Job = gpu::NttJob<F>;
        const Fr W = domain_gen(lg_n4), W3 = W * W * W;
        const Fr zeta = W.pow_u64(n);                                    // primitive 4th root of unity
        const Fr inv2 = Fr::from_u64(2).inverse(), inv2zeta = (zeta + zeta).inverse(), zero = Fr::zero();
        F *Q0 = cx.e[1].p, *Q1 = cx.e[1].p + n, *Q3 = cx.e[1].p + 2 * n, *zH = cx.e[0].p, *qH = cx.e[0].p + n;
        // per coset: z_A, z_B, r, t in e[2 | 3], z and the product in e[4]
        F *cs_buf[2][6];
        for (int i = 0; i < 2; i++) { for (int j = 0; j < 4; j++) cs_buf[i][j] = cx.e[2 + i].p + j * n; cs_buf[i][4] = cx.e[4].p + i * n; cs_buf[i][5] = cx.e[4].p + (2 + i) * n; }
        {   // r(alpha, .) on H (= v_H(alpha) / (alpha - h): alpha lies outside H) and on W H, W^3 H; e[4] (free until the forward transforms) is the tree's scratch
            F *outs[3] = {cx.ra_ev.p, cs_buf[0][2], cs_buf[1][2]};
            const Fr gs[3] = {Fr::one(), W, W3};
            gpu::vanishing_quotient_evals(outs, gs, 3, R.alpha, elems, (uint32_t)n, lg_n, cx.e[4].p, cx.e[4].n, s);
        }
        gpu::t_evals(cx.tmp_n.p, (uint32_t)n, cx.t_partial.p, t_nseg, d_t_colptr, d_t_seg_start, d_t_seg_end, d_t_heavy, t_nheavy, d_t_row, d_t_mat, d_t_coeff, cx.ra_ev.p, R.eta_a, R.eta_b, R.eta_c, s);
        gpu::z_evals_h(zH, cx.d_z, (uint32_t)n, (uint32_t)m, (uint32_t)c.num_witness, s);
        gpu::q1_coset_pointwise(qH, cx.ra_ev.p, cx.za_ev.p, cx.zb_ev.p, cx.tmp_n.p, zH, zero, zero, zero, R.eta_a, R.eta_b, R.eta_c, n, s);
        const Job inv0[2] = {{cx.poly[4].p, cx.tmp_n.p, 0}, {Q0, qH, 0}};
        gpu::ntt_batch<F>(inv0, 2, n, lg_n, true, 0, s); cx.poly_len[4] = n;
        R.jobs.start(1, [&](Lane &ln) { mpc_commit(cx, ln, R.r2[0]); }, true);                      // t
        gpu::z_poly_from_w(cx.zpoly.p, cx.poly[0].p, cx.poly_len[0], cx.x_poly.p, (uint32_t)m, n, s);
        const F *srcs[4] = {cx.poly[1].p, cx.poly[2].p, cx.poly[4].p, cx.zpoly.p};
        static const int dst_slot[4] = {0, 1, 3, 4};
        Job fwd[8];
        for (int i = 0; i < 2; i++) for (int j = 0; j < 4; j++) fwd[4 * i + j] = Job{cs_buf[i][dst_slot[j]], srcs[j], i == 0 ? 1 : 3};
        gpu::ntt_batch<F>(fwd, 8, n, lg_n, false, lg_n4, s);
        for (int i = 0; i < 2; i++) {
            const Fr zc = i == 0 ? zeta : zeta.neg();                    // zeta^cs: the value of X^|H| on the coset
            // the coefficient of X^|H| (rho of z_A, z_B; rho_w for z = w v_X + x) contributes rho zeta^cs everywhere on the coset
            gpu::q1_coset_pointwise(cs_buf[i][5], cs_buf[i][2], cs_buf[i][0], cs_buf[i][1], cs_buf[i][3], cs_buf[i][4], R.rhos[1] * zc, R.rhos[2] * zc, R.rhos[0] * zc, R.eta_a, R.eta_b, R.eta_c, n, s);
        }
        const Job inv13[2] = {{Q1, cs_buf[0][5], 1}, {Q3, cs_buf[1][5], 3}};
        gpu::ntt_batch<F>(inv13, 2, n, lg_n, true, lg_n4, s);
        gpu::q1_combine(cx.poly[6].p, cx.poly[5].p, Q0, Q1, Q3, cx.poly[3].p, inv2, inv2zeta, n, s);     // h_1 (2|H| coefficients), g_1 = remainder / X
        cx.poly_len[6] = 2 * n; cx.poly_len[5] = n - 1;
    }
    R.jobs.start(2, [&](Lane &ln) { mpc_commit(cx, ln, R.r2[1]); }, true);                          // g_1 (plain + shifted)
    R.jobs.start(4, [&](Lane &ln) { mpc_commit(cx, ln, R.r2[2]); }, true);                          // h_1
    R.jobs.join();
    { Bytes o; for (auto &lp : R.r2) o.commitment_tobytes(lp.comm); R.fs.absorb(o.b); }
    R.beta = sample_outside_h(R.fs);
    cx.timings.round2_ms = ms_since(R.t0); R.t0 = Clock::now();
}

void ProvingKeyImpl::prove_round3(ProofRun &R) {
    ProverContext &cx = R.cx;
    gpu::stream_t s = cx.stream;
    const F *elems = gpu::domain_elements<F>(lg_n);
    // ---- third round
    for (auto &lp : R.r3) draw_rand(lp, R.zk);
    R.vh_beta = eval_vanishing(n, R.beta);
    const Fr vv = R.vh_alpha * R.vh_beta;
    R.ea_vv = R.eta_a * vv; R.eb_vv = R.eta_b * vv; R.ec_vv = R.eta_c * vv; R.alpha_beta = R.alpha * R.beta;
    {   // f on K.  Its denominator (beta - row)(alpha - col) runs over pairs of ELEMENTS OF H, so 1 / den is a product of two entries of the tables v_H(alpha) / (alpha - h)
        // (round 2's, still in cx.ra_ev) and v_H(beta) / (beta - h) (one more product tree over H): two gathers instead of a batch inversion over K -- v_H(alpha) v_H(beta) included
        F *outs[1] = {cx.ra_poly.p};
        const Fr one = Fr::one();
        gpu::vanishing_quotient_evals(outs, &one, 1, R.beta, elems, (uint32_t)n, lg_n, cx.e[0].p, cx.e[0].n, s);
        gpu::f_evals_from_tables(cx.e[1].p, ix_ev[2].p, ix_ev[3].p, ix_ev[4].p, R.eta_a, R.eta_b, R.eta_c, cx.ra_ev.p, cx.ra_poly.p, d_ix_ri, d_ix_ci, k, s);
    }
    gpu::ntt<F>(cx.f_poly.p, cx.e[1].p, k, lg_k, true, s);
    gpu::d2d(cx.poly[7].p, cx.f_poly.p + 1, (k - 1) * sizeof(F), s); cx.poly_len[7] = k - 1;   // g_2 = (f - f(0)) / X
    R.jobs.start(1, [&](Lane &ln) { mpc_commit(cx, ln, R.r3[0]); }, true);                           // g_2 (plain + shifted): under the rest of the round
    // h_2 = (a - b f) / v_K with a = sum eta_M v_H(alpha) v_H(beta) val_M, b = (beta - row)(alpha - col) expanded with row_col: degree <= |K| - 2,
    // so it is interpolated from ONE coset of K, where v_K is the constant g^|K| - 1 and a, b come from the key's precomputed coset values
    gpu::ntt_scaled<F>(cx.e[2].p, cx.f_poly.p, k, lg_k, false, coset_tab, s);       // f on g K (the scaling by g^i rides on the transform's first pass)
    gpu::h2_coset(cx.e[0].p, ix_cs[0].p, ix_cs[1].p, ix_cs[2].p, ix_cs[3].p, ix_cs[4].p, ix_cs[5].p, cx.e[2].p, R.alpha, R.beta, R.alpha_beta, R.ea_vv, R.eb_vv, R.ec_vv, coset_vk_inv, k, s);
    gpu::ntt_scaled<F>(cx.poly[8].p, cx.e[0].p, k, lg_k, true, coset_tab_inv, s);   // h_2: interpolated from g K, scaled back by g^-i at the last pass's store (the coefficient of X^(|K|-1) is zero for a satisfied instance)
    cx.poly_len[8] = k - 1;
    R.jobs.start(2, [&](Lane &ln) { mpc_commit(cx, ln, R.r3[1]); }, true);                           // h_2
    // ---- evaluations.  Three of the four points are evaluations at beta, known since round 2: a lone call computes them (and the beta opening's shifted witness
    // g_1 / (X - beta)) NOW, on the main stream under round 3's commitments; only g_2(gamma) has to wait for gamma.
    R.early_evals = R.jobs.async;
    if (R.early_evals) {
        const F *ps[3] = {cx.poly[5].p, cx.poly[4].p, cx.poly[2].p};
        const size_t ls[3] = {cx.poly_len[5], cx.poly_len[4], cx.poly_len[2]};
        const Fr at[3] = {R.beta, R.beta, R.beta};
        Fr ev[3];
        gpu::poly_eval_multi(ps, ls, at, 3, ev, cx.scratch.p, cx.scratch.n, s);           // g_1(beta), t(beta), z_b(beta)
        R.pf.evals[0] = ev[0]; R.pf.evals[2] = ev[1]; R.pf.evals[3] = ev[2];
        gpu::divide_by_linear(cx.wit2.p, cx.poly[5].p, cx.poly_len[5], R.beta, cx.scratch.p, cx.scratch.n, s);
    }
    R.jobs.join();
    { Bytes o; for (auto &lp : R.r3) o.commitment_tobytes(lp.comm); R.fs.absorb(o.b); }
    R.gamma = R.fs.rng().rand_field<Fr>();
    cx.timings.round3_ms = ms_since(R.t0); R.t0 = Clock::now();
}

void ProvingKeyImpl::prove_open(ProofRun &R) {
    ProverContext &cx = R.cx;
    gpu::stream_t s = cx.stream;
    if (R.early_evals) {
        const F *ps[1] = {cx.poly[7].p};
        const size_t ls[1] = {cx.poly_len[7]};
        gpu::poly_eval_multi(ps, ls, &R.gamma, 1, &R.pf.evals[1], cx.scratch.p, cx.scratch.n, s);   // g_2(gamma)
    } else {
        const F *ps[4] = {cx.poly[5].p, cx.poly[7].p, cx.poly[4].p, cx.poly[2].p};
        const size_t ls[4] = {cx.poly_len[5], cx.poly_len[7], cx.poly_len[4], cx.poly_len[2]};
        const Fr at[4] = {R.beta, R.gamma, R.beta, R.beta};
        gpu::poly_eval_multi(ps, ls, at, 4, R.pf.evals, cx.scratch.p, cx.scratch.n, s);     // g_1(beta), g_2(gamma), t(beta), z_b(beta)
    }
    const Fr g2_g = R.pf.evals[1], t_b = R.pf.evals[2], zb_b = R.pf.evals[3];      // (g_1(beta) = evals[0] enters only through the transcript)
    { Bytes o; for (auto &v : R.pf.evals) o.field(v); R.fs.absorb(o.b); }
    Fr ch;
    { uint64_t lo = R.fs.rng().next_u64(), hi = R.fs.rng().next_u64(); uint32_t raw[8] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), 0, 0, 0, 0}; ch = Fr::from_raw(raw); }
    // ---- linear-combination coefficients (construct_linear_combinations)
    Fr r_alpha_at_beta = (R.vh_alpha - R.vh_beta) * (R.alpha - R.beta).inverse();
    Fr vx_beta = eval_vanishing(m, R.beta);
    Fr c_za = r_alpha_at_beta * (R.eta_a + R.eta_c * zb_b), c_w = (t_b * vx_beta).neg(), c_h1 = R.vh_beta.neg();
    Fr vk_gamma = eval_vanishing(k, R.gamma);
    Fr bmul = R.gamma * g2_g + t_b * Fr::from_u64(k).inverse();
    Fr c_row = R.alpha * bmul, c_col = R.beta * bmul, c_rc = bmul.neg(), c_h2 = vk_gamma.neg();
    Fr chp[5]; chp[0] = Fr::one(); for (int i = 1; i < 5; i++) chp[i] = chp[i - 1] * ch;
    auto rand_axpy = [](Fr out[3], const Fr &sc, const KzgRand &r) { if (r.hiding) for (int i = 0; i < 3; i++) out[i] = out[i] + sc * r.b[i]; };
    auto host_divide_by_linear = [](Fr q[2], const Fr p[3], const Fr &z) { q[1] = p[2]; q[0] = p[1] + z * p[2]; };
    auto host_eval3 = [](const Fr p[3], const Fr &z) { return p[0] + z * (p[1] + z * p[2]); };
    // the two openings are independent: on the latency path they run side by side, the second one on its own buffers
    if (R.jobs.async && !cx.ev_aux) {          // guarded by the LAST resource created: an allocation that throws leaves ev_aux null, and the next lone call starts over (advisor r05)
        for (DevBuf *b : {&cx.acc_b, &cx.wit_b, &cx.wit2_b, &cx.scratch_b, &cx.scratch_c}) b->release();
        cx.acc_b.alloc(cx.acc.n); cx.wit_b.alloc(cx.wit.n); cx.wit2_b.alloc(cx.wit2.n); cx.scratch_b.alloc(cx.scratch.n);
        cx.scratch_c.alloc(gpu::divide_by_linear_scratch(std::max(n, k) + 1)); cx.ev_aux = gpu::event_create();
    }
    const bool two_sets = cx.ev_aux != nullptr && R.jobs.async;
    if (R.jobs.async) gpu::sync(s);                 // (the opening jobs read what the main stream made: everything is in place from here)
    // the two jobs below hold THIS function's locals (the challenge powers, the linear-combination coefficients) by reference: if anything between their start and the join
    // throws, they must have finished before those locals go -- this guard is declared after everything they capture, so it is destroyed first
    struct WaitJobs { RoundJobs &j; ~WaitJobs() { j.wait_all(); } } wait_jobs{R.jobs};
    R.jobs.start(1, [&](Lane &ln) {   // open at beta: g_1 (ch^0; shifted ch^1), outer_sumcheck (ch^2), t (ch^3), z_b (ch^4)
        gpu::stream_t ls_ = ln.stream;
        size_t plen = 3 * n;
        Fr rb[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
        {
            const F *ps[7] = {cx.poly[5].p, cx.poly[3].p, cx.poly[1].p, cx.poly[0].p, cx.poly[6].p, cx.poly[4].p, cx.poly[2].p};
            size_t ls[7] = {cx.poly_len[5], cx.poly_len[3], cx.poly_len[1], cx.poly_len[0], cx.poly_len[6], cx.poly_len[4], cx.poly_len[2]};
            Fr sc[7] = {chp[0], chp[2], chp[2] * c_za, chp[2] * c_w, chp[2] * c_h1, chp[3], chp[4]};
            gpu::poly_lincomb_n(cx.acc.p, plen, ps, ls, sc, 7, ls_);
        }
        rand_axpy(rb, chp[0], R.r2[1].rand); rand_axpy(rb, chp[2] * c_za, R.r1[1].rand); rand_axpy(rb, chp[2] * c_w, R.r1[0].rand); rand_axpy(rb, chp[4], R.r1[2].rand);
        // the shifted part (g_1, degree bound |H| - 2) rides in the same Pippenger instance; its quotient by X - beta may already exist (early_evals).  The second scratch half
        // keeps the two divisions of this job apart when they were queued on different streams.
        if (!R.early_evals) gpu::divide_by_linear(cx.wit2.p, cx.poly[5].p, cx.poly_len[5], R.beta, cx.scratch.p, cx.scratch.n, ls_);
        gpu::divide_by_linear(cx.wit.p, cx.acc.p, plen, R.beta, cx.scratch.p, cx.scratch.n, ls_);
        gpu::poly_scale(cx.wit2.p, chp[1], cx.poly_len[5] - 1, ls_);
        msm_opening_prepare(cx, ln, cx.wit.p, plen - 1, cx.wit2.p, cx.poly_len[5] - 1, bounds[1] - (n - 2));
        // (the blinding part of the witness is host work: under the device's digit grouping, not after the wait for its sum)
        XYZZ<Fq377> hw = XYZZ<Fq377>::inf();
        Fr rq[2]; host_divide_by_linear(rq, rb, R.beta);
        for (int i = 0; i < 2; i++) hw.add(srs->gamma_tab[i].mul(rq[i]));
        Fr rv = host_eval3(rb, R.beta);
        Fr srb[3]; for (int i = 0; i < 3; i++) srb[i] = chp[1] * R.r2[1].shifted_rand.b[i];
        host_divide_by_linear(rq, srb, R.beta);
        for (int i = 0; i < 2; i++) hw.add(srs->gamma_tab[i].mul(rq[i]));
        rv = rv + host_eval3(srb, R.beta);
        XYZZ<Fq377> w = msm_opening_finish(ln);
        w.add(hw);
        R.pf.w_beta = w.to_affine(); R.pf.random_v_beta = rv;
    }, false);
    R.jobs.start(2, [&](Lane &ln) {   // open at gamma: g_2 (ch^0; shifted ch^1), inner_sumcheck (ch^2)
        gpu::stream_t ls_ = ln.stream;
        DevBuf &acc_ = two_sets ? cx.acc_b : cx.acc, &wit_ = two_sets ? cx.wit_b : cx.wit, &wit2_ = two_sets ? cx.wit2_b : cx.wit2, &scr_ = two_sets ? cx.scratch_b : cx.scratch;
        size_t plen = k;
        {
            const F *ps[8] = {cx.poly[7].p, ix_co[2].p, ix_co[3].p, ix_co[4].p, ix_co[0].p, ix_co[1].p, ix_co[5].p, cx.poly[8].p};
            size_t ls[8] = {cx.poly_len[7], k, k, k, k, k, k, cx.poly_len[8]};
            Fr sc[8] = {chp[0], chp[2] * R.ea_vv, chp[2] * R.eb_vv, chp[2] * R.ec_vv, chp[2] * c_row, chp[2] * c_col, chp[2] * c_rc, chp[2] * c_h2};
            gpu::poly_lincomb_n(acc_.p, plen, ps, ls, sc, 8, ls_);
        }
        gpu::divide_by_linear(wit_.p, acc_.p, plen, R.gamma, scr_.p, scr_.n, ls_);
        if (R.jobs.async) {        // the shifted part g_2 / (X - gamma) is being made on the main stream by the calling thread, beside this division (two chains of a dozen short launches)
            while (!R.g2_quotient_queued.load(std::memory_order_acquire)) std::this_thread::yield();
            gpu::stream_wait_event(ls_, cx.ev_aux);
        } else {
            gpu::divide_by_linear(wit2_.p, cx.poly[7].p, cx.poly_len[7], R.gamma, scr_.p, scr_.n, ls_);
            gpu::poly_scale(wit2_.p, chp[1], cx.poly_len[7] - 1, ls_);
        }
        msm_opening_prepare(cx, ln, wit_.p, plen - 1, wit2_.p, cx.poly_len[7] - 1, bounds[1] - (k - 2));
        XYZZ<Fq377> w = msm_opening_finish(ln);
        R.pf.w_gamma = w.to_affine();
    }, false);
    if (R.jobs.async) {
        try {
            gpu::divide_by_linear(cx.wit2_b.p, cx.poly[7].p, cx.poly_len[7], R.gamma, cx.scratch_c.p, cx.scratch_c.n, s);
            gpu::poly_scale(cx.wit2_b.p, chp[1], cx.poly_len[7] - 1, s);
            gpu::event_record(cx.ev_aux, s);
        } catch (...) { R.g2_quotient_queued.store(true, std::memory_order_release); throw; }      // (never leave the gamma job spinning)
        R.g2_quotient_queued.store(true, std::memory_order_release);
    }
    R.jobs.join();
}

Proof ProvingKeyImpl::prove(ProverContext &cx, const uint8_t *host_trace, const uint8_t *msg, size_t len, const uint8_t *key, const uint8_t *zk_seed, bool throughput) {
    gpu::set_device(device);
    std::lock_guard<std::mutex> busy(cx.in_use);
    cx.throughput = throughput;
    ProofRun R(*this, cx, host_trace, msg, len, key, zk_seed);
    prove_round1(R);
    prove_round2(R);
    prove_round3(R);
    prove_open(R);
    for (int i = 0; i < 4; i++) R.pf.comms[i] = R.r1[i].comm;
    for (int i = 0; i < 3; i++) R.pf.comms[4 + i] = R.r2[i].comm;
    for (int i = 0; i < 2; i++) R.pf.comms[7 + i] = R.r3[i].comm;
    cx.timings.open_ms = ms_since(R.t0);
    cx.timings.total_ms = ms_since(R.t_all);
    { std::lock_guard<std::mutex> g(ctx_mu); last_timings = cx.timings; }
    return R.pf;
}

ProvingKey::~ProvingKey() { if (impl) { DeviceScope on(impl->device); delete impl; } }
const VerifyingKey &ProvingKey::vk() const { return impl->vk; }
const Circuit &ProvingKey::circuit() const { return impl->circuit; }
const ProverTimings &ProvingKey::last_timings() const { return impl->last_timings; }
bool ProvingKey::tables_built(uint64_t *bytes) const {
    if (bytes) *bytes = impl->use_tables ? impl->srs->bytes() : 0;        // the table set this key uses -- shared with every other key over the same universal SRS (srs_info)
    return impl->use_tables;
}
void ProvingKey::srs_info(uint64_t out[6], double secs[2]) const {
    const UniversalSrs &U = *impl->srs;
    out[0] = U.max_degree; out[1] = U.stride; out[2] = U.n_tab; out[3] = U.bytes();
    out[4] = (uint64_t)impl->srs.use_count();                              // keys holding this SRS right now
    out[5] = impl->lag ? 2 * (uint64_t)impl->lag->n * sizeof(SrsPoint) : 0;
    if (secs) { secs[0] = impl->setup_srs_s; secs[1] = impl->setup_total_s; }
}
size_t ProvingKey::contexts() const { size_t v = impl->n_contexts.load(); return v ? v : default_contexts(); }
void ProvingKey::set_contexts(size_t n) {
    if (n > 64) throw std::invalid_argument("set_contexts: at most 64 prover contexts per key");
    impl->n_contexts.store(n);                                             // 0 = back to the process default
}
void ProvingKey::msm_powers_partial_device(const uint8_t *scalars, size_t n_local, size_t offset, void *dev_out) {
    gpu::set_device(impl->device);
    if (!impl->use_tables) throw std::runtime_error("msm_powers_partial_device: this key has no window tables (KEY_NO_TABLES, or the device was short of memory)");
    if (offset + n_local > impl->supported_degree + 1) throw std::runtime_error("msm_powers_partial_device: range exceeds the committer key");
    ProverContext &cx = impl->context(0);
    std::lock_guard<std::mutex> busy(cx.in_use);
    if (n_local > cx.acc.n) throw std::runtime_error("msm_powers_partial_device: more scalars than a prover context holds");
    if (n_local) gpu::h2d(cx.acc.p, scalars, n_local * sizeof(F), cx.stream);
    gpu::msm_table_sum_device<Bls377>(cx.msm_ws, impl->d_powers, impl->srs_stride, offset, impl->table_c, cx.acc.p, n_local, (XYZZ<Fq377> *)dev_out, cx.stream);
}
Proof ProvingKey::prove_aes(const uint8_t *message, size_t len, const uint8_t key[16], const uint8_t *zk_seed) {
    if (impl->circuit.kind != CIRCUIT_AES) throw std::invalid_argument("proving key was not synthesized for the AES circuit");
    if (len % 16) throw std::invalid_argument("Input must be 16 bytes length when adding round key");
    if (len != impl->circuit.n_blocks * 16) throw std::invalid_argument("InstanceDoesNotMatchIndex: proving key was synthesized for " + std::to_string(impl->circuit.n_blocks * 16) + " bytes");
    return impl->prove(impl->context(0), nullptr, message, len, key, zk_seed);
}
// One proof with the library's op recorder open (gpu.hpp oplog_*): the transforms and MSMs it ACTUALLY launched, as JSON --
//   {"h":..,"k":..,"x":..,"variables":..,"constraints":..,"nnz":[a,b,c],"blocks":..,"path":"throughput"|"lone","ntt":[[points, transforms in the launch],..],
//    "msm":[[points,"buckets"|"second_bases"|"class_sum"],..]}
// -- what bench.py computes the whole-proof roofline of SURVEY.md 8(d) from.  Process-global recorder: call with no other proof in flight.
std::string ProvingKey::op_lists_json(const uint8_t *message, size_t len, const uint8_t key[16], bool throughput_path) {
    if (impl->circuit.kind != CIRCUIT_AES) throw std::invalid_argument("proving key was not synthesized for the AES circuit");
    if (len != impl->circuit.n_blocks * 16) throw std::invalid_argument("InstanceDoesNotMatchIndex: proving key was synthesized for " + std::to_string(impl->circuit.n_blocks * 16) + " bytes");
    gpu::OpRecord rec;
    gpu::oplog_begin();
    try { impl->prove(impl->context(0), nullptr, message, len, key, nullptr, throughput_path); }
    catch (...) { gpu::oplog_end(); throw; }
    rec = gpu::oplog_end();
    const Circuit &c = impl->circuit;
    std::string o = "{\"h\":" + std::to_string(impl->n) + ",\"k\":" + std::to_string(impl->k) + ",\"x\":" + std::to_string(impl->m) + ",\"variables\":" + std::to_string(c.num_variables()) +
                    ",\"constraints\":" + std::to_string(c.num_constraints) + ",\"nnz\":[" + std::to_string(c.A.nnz()) + "," + std::to_string(c.B.nnz()) + "," + std::to_string(c.C.nnz()) + "]" +
                    ",\"blocks\":" + std::to_string(c.n_blocks) + ",\"path\":\"" + (throughput_path ? "throughput" : "lone") + "\",\"ntt\":[";
    for (size_t i = 0; i < rec.ntt.size(); i++) o += (i ? "," : "") + std::string("[") + std::to_string(rec.ntt[i].first) + "," + std::to_string(rec.ntt[i].second) + "]";
    o += "],\"msm\":[";
    static const char *kinds[3] = {"buckets", "second_bases", "class_sum"};
    for (size_t i = 0; i < rec.msm.size(); i++) o += (i ? "," : "") + std::string("[") + std::to_string(rec.msm[i].first) + ",\"" + kinds[rec.msm[i].second % 3] + "\"]";
    o += "]}";
    return o;
}
std::vector<uint8_t> ProvingKey::aes_witness(const uint8_t *message, size_t len, const uint8_t key[16]) {
    if (impl->circuit.kind != CIRCUIT_AES) throw std::invalid_argument("proving key was not synthesized for the AES circuit");
    if (len % 16) throw std::invalid_argument("Input must be 16 bytes length when adding round key");
    if (len != impl->circuit.n_blocks * 16) throw std::invalid_argument("InstanceDoesNotMatchIndex: proving key was synthesized for " + std::to_string(impl->circuit.n_blocks * 16) + " bytes");
    const Circuit &c = impl->circuit;
    ProverContext &cx = impl->context(0);
    std::lock_guard<std::mutex> busy(cx.in_use);
    gpu::stream_t s = cx.stream;
    gpu::h2d(cx.d_msg, message, len, s); gpu::h2d(cx.d_key, key, 16, s);
    gpu::aes_trace(cx.d_trace, c.trace_bytes, cx.d_msg, cx.d_key, 1, (uint32_t)c.n_blocks, s);
    gpu::witness_expand(cx.d_z, impl->d_desc, (uint32_t)c.num_variables(), cx.d_trace, impl->d_sbox_in, impl->d_sbox_tmpl, s);
    std::vector<uint8_t> z(c.num_variables());
    gpu::d2h(z.data(), cx.d_z, z.size(), s);
    return z;
}
// zero-knowledge randomness of proof i of a chunked / batch call: the caller's seed is domain-separated per proof, Blake2s(seed || (offset + i) as u64 LE),
// so no two proofs share rho, the KZG hiding coefficients or the mask polynomial -- across calls and ranks too when they pass job-global offsets.
// seed == nullptr keeps the reference's behaviour (every encrypt() call draws from ark_std::test_rng(), src/lib.rs:65): bit-parity mode for tests,
// NOT zero-knowledge across proofs; the C ABI only reaches it through the explicit *_seeded(NULL) calls (no environment override).
void os_random_seed(uint8_t out[32]) {
    size_t got = 0;
    while (got < 32) {
        ssize_t r = getrandom(out + got, 32 - got, 0);
        if (r < 0) { if (errno == EINTR) continue; throw std::runtime_error("getrandom failed: no entropy for the prover's zero-knowledge seed"); }
        got += (size_t)r;
    }
}
static void derive_zk_seed(uint8_t out[32], const uint8_t *seed32, uint64_t index) {
    uint8_t buf[40];
    memcpy(buf, seed32, 32);
    for (int i = 0; i < 8; i++) buf[32 + i] = (uint8_t)(index >> (8 * i));
    Blake2s::digest(out, buf, sizeof buf);
}
static std::vector<Proof> prove_many(ProvingKeyImpl *impl, const uint8_t *messages, const uint8_t *keys, size_t key_stride, size_t n_chunks, size_t n_contexts, const uint8_t *zk_seed, uint64_t index_offset) {
    size_t chunk = impl->circuit.n_blocks * 16;
    if (n_contexts == 0) n_contexts = 1;
    n_contexts = std::min(n_contexts, std::max<size_t>(n_chunks, 1));
    std::vector<Proof> proofs(n_chunks);
    for (size_t i = 0; i < n_contexts; i++) impl->context(i);           // allocate outside the worker threads
    std::atomic<size_t> next{0};
    std::vector<std::string> errors(n_contexts);
    const int device = impl->device;
    auto worker = [&](size_t ci) {
        try {
            gpu::set_device(device);
            ProverContext &cx = impl->context(ci);
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= n_chunks) break;
                uint8_t seed_i[32];
                if (zk_seed) derive_zk_seed(seed_i, zk_seed, index_offset + (uint64_t)i);
                proofs[i] = impl->prove(cx, nullptr, messages + i * chunk, chunk, keys + i * key_stride, zk_seed ? seed_i : nullptr, n_chunks > 1);   // a multi-proof call is a throughput call
            }
        } catch (const std::exception &e) { errors[ci] = e.what(); }
    };
    gpu::ThroughputWaits waits(n_contexts > 1);     // the context threads sleep-poll their streams instead of spinning (runtime.hip)
    std::vector<std::thread> threads;
    for (size_t ci = 1; ci < n_contexts; ci++) threads.emplace_back(worker, ci);
    worker(0);
    for (auto &t : threads) t.join();
    for (auto &e : errors) if (!e.empty()) throw std::runtime_error(e);
    return proofs;
}
std::vector<Proof> ProvingKey::prove_aes_chunked(const uint8_t *message, size_t len, const uint8_t key[16], size_t n_contexts, const uint8_t *zk_seed, uint64_t index_offset) {
    if (impl->circuit.kind != CIRCUIT_AES) throw std::invalid_argument("proving key was not synthesized for the AES circuit");
    size_t chunk = impl->circuit.n_blocks * 16;
    if (chunk == 0 || len % chunk) throw std::invalid_argument("message length must be a multiple of the key's plaintext length (" + std::to_string(chunk) + " bytes)");
    return prove_many(impl, message, key, 0, len / chunk, n_contexts, zk_seed, index_offset);
}
std::vector<Proof> ProvingKey::prove_aes_batch(const uint8_t *messages, const uint8_t *keys, size_t n, size_t n_contexts, const uint8_t *zk_seed, uint64_t index_offset) {
    if (impl->circuit.kind != CIRCUIT_AES) throw std::invalid_argument("proving key was not synthesized for the AES circuit");
    if (impl->circuit.n_blocks == 0) throw std::invalid_argument("proving key has an empty plaintext");
    return prove_many(impl, messages, keys, 16, n, n_contexts, zk_seed, index_offset);
}
Proof ProvingKey::prove_ops(uint32_t x, uint32_t y, const uint8_t *zk_seed) {
    if (impl->circuit.kind == CIRCUIT_AES) throw std::invalid_argument("proving key was synthesized for the AES circuit");
    uint8_t trace[16] = {0};
    uint64_t r = impl->circuit.kind == CIRCUIT_OPS_XOR ? (uint64_t)(x ^ y) : (uint64_t)x + y;
    for (int i = 0; i < 4; i++) { trace[i] = (uint8_t)(x >> (8 * i)); trace[4 + i] = (uint8_t)(y >> (8 * i)); }
    for (int i = 0; i < 8; i++) trace[8 + i] = (uint8_t)(r >> (8 * i));
    return impl->prove(impl->context(0), trace, nullptr, 0, nullptr, zk_seed);
}
// ark-serialize 0.3 compressed image of ark_marlin::IndexProverKey<Fr, MarlinKZG10<Bls12_377, DensePolynomial<Fr>>> [RECALL: the pinned ark-marlin fork is not under
// /root/reference; field order as in arkworks marlin after the joint-matrix arithmetization, ark-poly-commit 0.3.0, ark-poly 0.3.0 -- SURVEY.md A.4 / A.5]:
//   index_vk         : IndexVerifierKey (serialize_vk_ark)
//   index_comm_rands : Vec<marlin_pc::Randomness> = u64 len 6, each { rand: kzg10::Randomness { blinding_polynomial: DensePolynomial = Vec<Fr> (empty) }, shifted_rand: None }
//   index            : Index { index_info (4 x u64), a, b, c : Matrix = Vec<Vec<(Fr, usize)>> (padded, rows in constraint order, entries (coeff 32 B, column u64)),
//                      joint_arith : { row, col, val_a, val_b, val_c, row_col : LabeledPolynomial { label: String, polynomial: Vec<Fr>, degree_bound: None, hiding_bound: None },
//                                      evals_on_K : { row, col, row_col, val_a, val_b, val_c : Evaluations { evals: Vec<Fr>, domain: GeneralEvaluationDomain (u8 variant 0 = Radix2,
//                                      size u64, log_size u32, size_as_field_element, size_inv, group_gen, group_gen_inv, generator_inv) } } } }
//   committer_key    : marlin_pc::CommitterKey { powers: Vec<G1 compressed>, shifted_powers: Some(Vec<G1>), powers_of_gamma_g: Vec<G1> (3), enforced_degree_bounds: Some(Vec<usize>),
//                      max_degree: usize }
// Streamed to a file (0.65 GB for the one-block key, 2.6 GB for a four-block one); the SRS powers are re-made on the device in chunks (the key keeps them only in the
// Edwards form k_accumulate gathers).  What this buys: `synthesize_keys` here takes seconds where the reference's takes minutes, and the image lets the reference's own CPU
// `encrypt()` run on a GPU-made key (integration/check_on_cargo_box.sh).
namespace {
struct FileSink {
    FILE *f; uint64_t n = 0; Bytes buf;
    explicit FileSink(const std::string &path) : f(fopen(path.c_str(), "wb")) { if (!f) throw std::runtime_error("cannot open " + path + " for writing"); }
    ~FileSink() { if (f) fclose(f); }
    void flush() { if (!buf.b.empty()) { if (fwrite(buf.b.data(), 1, buf.b.size(), f) != buf.b.size()) throw std::runtime_error("short write (disk full?)"); n += buf.b.size(); buf.b.clear(); } }
    void maybe_flush() { if (buf.b.size() > (8u << 20)) flush(); }
    void u32(uint32_t v) { for (int i = 0; i < 4; i++) buf.u8((uint8_t)(v >> (8 * i))); }
    void fr_vec(const std::vector<Fr> &v, size_t len) { buf.u64(len); for (size_t i = 0; i < len; i++) { buf.field(v[i]); maybe_flush(); } }
    void label(const char *s) { size_t l = strlen(s); buf.u64(l); buf.put(s, l); }
};
Fr fr_from_small(int64_t c) { return c >= 0 ? Fr::from_u64((uint64_t)c) : Fr::from_u64((uint64_t)(-c)).neg(); }
}  // namespace
// Nothing here touches a prover context: the key's index data is immutable, the powers are re-made on a private stream into a private staging buffer, so proofs on this
// key keep running while a multi-GB image streams out.  A failure (short write, device error) removes the partial file.
uint64_t ProvingKey::serialize_ark_to_file(const std::string &path, bool uncompressed) const {
    ProvingKeyImpl &K = *impl;
    gpu::set_device(K.device);
    struct StreamGuard { gpu::stream_t s = nullptr; ~StreamGuard() { gpu::stream_destroy(s); } } sg;
    sg.s = gpu::stream_create();
    gpu::stream_t s = sg.s;
    FileSink o(path);                                   // (throws if the file cannot be opened: nothing of ours to remove then)
    struct Unlink { const std::string &p; bool armed = true; ~Unlink() { if (armed) ::remove(p.c_str()); } } partial{path};      // armed only once this call has created / truncated the file
    { auto v = serialize_vk_ark(K.vk, uncompressed); o.buf.put(v.data(), v.size()); }
    o.buf.u64(6);
    for (int i = 0; i < 6; i++) { o.buf.u64(0); o.buf.u8(0); }                      // Randomness::empty(): zero blinding polynomial, no shifted_rand
    o.buf.u64(K.vk.num_variables); o.buf.u64(K.vk.num_constraints); o.buf.u64(K.vk.num_non_zero); o.buf.u64(K.vk.num_instance);
    for (const CsrMatrix *M : {&K.circuit.A, &K.circuit.B, &K.circuit.C}) {
        o.buf.u64(M->rows());
        for (size_t r = 0; r < M->rows(); r++) {
            o.buf.u64(M->rowptr[r + 1] - M->rowptr[r]);
            for (uint32_t i = M->rowptr[r]; i < M->rowptr[r + 1]; i++) { o.buf.field(fr_from_small(M->coeff[i])); o.buf.u64(M->col[i]); }
            o.maybe_flush();
        }
    }
    std::vector<Fr> host(K.k);
    static const char *labels[6] = {"row", "col", "a_val", "b_val", "c_val", "row_col"};
    for (int i = 0; i < 6; i++) {                                                  // LabeledPolynomial: label, coefficients (DensePolynomial drops leading zero coefficients), no bounds
        gpu::d2h(host.data(), K.ix_co[i].p, K.k * sizeof(F), s);
        size_t len = K.k;
        while (len && host[len - 1].is_zero()) len--;
        o.label(labels[i]);
        o.fr_vec(host, len);
        o.buf.u8(0); o.buf.u8(0);
    }
    const Fr kgen = domain_gen(K.lg_k);
    for (int i : {0, 1, 5, 2, 3, 4}) {                                             // MatrixEvals field order: row, col, row_col, val_a, val_b, val_c
        gpu::d2h(host.data(), K.ix_ev[i].p, K.k * sizeof(F), s);
        o.fr_vec(host, K.k);
        o.buf.u8(0);                                                               // GeneralEvaluationDomain::Radix2
        o.buf.u64(K.k); o.u32((uint32_t)K.lg_k);
        o.buf.field(Fr::from_u64(K.k)); o.buf.field(Fr::from_u64(K.k).inverse()); o.buf.field(kgen); o.buf.field(kgen.inverse()); o.buf.field(K.coset_g_inv);
    }
    // committer key: the powers in the standard affine form, re-made on the device chunk by chunk
    const size_t CH = (size_t)1 << 20;
    struct Chunk { G1A *d = nullptr; ~Chunk() { gpu::dfree(d); } } chunk;
    chunk.d = (G1A *)gpu::dmalloc(CH * sizeof(G1A));
    std::vector<G1A> h(CH);
    auto write_powers = [&](size_t from, size_t count) {
        o.buf.u64(count);
        for (size_t off = 0; off < count; off += CH) {
            const size_t m_ = std::min(CH, count - off);
            gpu::fixed_base_powers<Bls377>(chunk.d, K.vk.g, K.srs->beta, from + off, m_, s);
            gpu::d2h(h.data(), chunk.d, m_ * sizeof(G1A), s);
            for (size_t i = 0; i < m_; i++) { o.buf.g1(h[i], uncompressed); o.maybe_flush(); }
        }
    };
    write_powers(0, K.supported_degree + 1);
    o.buf.u8(1); write_powers(K.lowest_shift, K.bounds[1] + 1);
    o.buf.u64(3); for (int i = 0; i < 3; i++) o.buf.g1(K.srs->gamma_powers[i], uncompressed);
    o.buf.u8(1); o.buf.u64(2); o.buf.u64(K.bounds[0]); o.buf.u64(K.bounds[1]);
    o.buf.u64(K.max_degree);
    o.flush();
    partial.armed = false;
    return o.n;
}

std::vector<uint8_t> ProvingKey::debug_fetch(const std::string &name) const {
    static const char *pn[9] = {"w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2"};
    static const char *in[6] = {"row", "col", "a_val", "b_val", "c_val", "row_col"};
    std::vector<uint8_t> out;
    ProverContext &cx = impl->context(0);
    auto grab = [&](const void *d, size_t bytes) { out.resize(bytes); gpu::d2h(out.data(), d, bytes, cx.stream); };
    if (name == "z") { grab(cx.d_z, impl->circuit.num_variables()); return out; }
    if (name == "trace") { grab(cx.d_trace, impl->circuit.trace_bytes); return out; }
    if (name == "z_a_evals") { grab(cx.za_ev.p, impl->n * sizeof(F)); return out; }
    if (name == "z_b_evals") { grab(cx.zb_ev.p, impl->n * sizeof(F)); return out; }
    for (int i = 0; i < 9; i++) if (name == pn[i]) { grab(cx.poly[i].p, cx.poly_len[i] * sizeof(F)); return out; }
    for (int i = 0; i < 6; i++) {
        if (name == in[i]) { grab(impl->ix_co[i].p, impl->k * sizeof(F)); return out; }
        if (name == std::string(in[i]) + "_evals") { grab(impl->ix_ev[i].p, impl->k * sizeof(F)); return out; }
    }
    throw std::invalid_argument("debug_fetch: unknown buffer " + name);
}

std::unique_ptr<ProvingKey> synthesize_keys(int circuit_kind, size_t message_len, const SrsLiterals &srs, unsigned flags) {
    std::unique_ptr<ProvingKey> pk(new ProvingKey());
    pk->impl = new ProvingKeyImpl();
    pk->impl->setup(circuit_kind, message_len, srs, flags);
    return pk;
}

}  // namespace zk
