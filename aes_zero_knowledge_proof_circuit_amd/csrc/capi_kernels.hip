// csrc/capi_kernels.hip -- kernel-level C entry points (parity tests + roofline measurement) and device selection
#include "../../include/zkaes.h"
#include <cstring>
#include <string>
#include <vector>
#include "hip_util.hpp"
#include "marlin.hpp"
#include "te28.cuh"
#include <algorithm>
#include <chrono>

// stream-copy probe for the measured HBM peak bench.py prints next to the nominal 8 TB/s: 4 x 16 B per lane, all four loads
// issued before the first store, non-temporal both ways
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stream_copy(uint4 *__restrict__ dst_, const uint4 *__restrict__ src_, size_t n16) {
    v4u *dst = (v4u *)dst_; const v4u *src = (const v4u *)src_;
    size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (base + 768 < n16) {
        v4u v0 = __builtin_nontemporal_load(src + base), v1 = __builtin_nontemporal_load(src + base + 256);
        v4u v2 = __builtin_nontemporal_load(src + base + 512), v3 = __builtin_nontemporal_load(src + base + 768);
        __builtin_nontemporal_store(v0, dst + base); __builtin_nontemporal_store(v1, dst + base + 256);
        __builtin_nontemporal_store(v2, dst + base + 512); __builtin_nontemporal_store(v3, dst + base + 768);
    } else {
        for (size_t i = base; i < n16; i += 256) dst[i] = src[i];
    }
}


// ---- per-box calibration of the integer roof (zkaes_int_rate_bench; bench.py prints it as roofline.int_multiplier.calibration).  Two probes, both bracketed per wave by
// s_memtime (shader-clock counter) and s_memrealtime (100 MHz): the isolated Fq377 reduced-radix product stream at four waves per SIMD -- what round 3's tools/ubench/rates.hip
// measured once on one box and every later bench line quoted -- and k_accumulate<EdwardsLaw>'s own loop (te_madd_hot) at the production launch shape over a table that stays
// in L2: the rate the hot kernel would run at on THIS box if its gathers were free.  Stamps: {shader cycles, real-time ticks} of lane 0 of every wave.
struct CalStamp { uint64_t cyc, rt; };
__global__ void __launch_bounds__(64) k_cal_fqmul(CalStamp *st, zk::FpMsm<zk::Fq377P> *sink, zk::FpMsm<zk::Fq377P> a, int iters) {
    using G = zk::FpMsm<zk::Fq377P>;
    const uint64_t bias = G::hot_loop_bias();
    G x = a, y = a;
    x.l[0] += threadIdx.x & 0xff;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { x = G::mul_biased(x, y, bias); y = G::mul_biased(y, x, bias); }
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
    sink[blockIdx.x * 64 + threadIdx.x] = x + y;
}
__global__ void __launch_bounds__(64, 2) k_cal_hot_loop(CalStamp *st, zk::AccTE<zk::Fq377P> *sink, const zk::Niels28<zk::Fq377P> *__restrict__ tab, uint32_t mask, int iters) {
    using P = zk::Fq377P;
    const uint64_t bias = zk::FpMsm<P>::hot_loop_bias();
    uint32_t t = blockIdx.x * 64 + threadIdx.x, x = t * 2654435761u + 12345u;
    zk::AccTE<P> acc = zk::te_identity<P>();
    zk::Niels28<P> pt = zk::niels_load_signed<P>(tab + (x & mask), false);
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        const uint32_t cur = x;
        x = x * 1664525u + 1013904223u;
        zk::te_madd_hot<P>(acc, pt, cur >> 31, tab + ((x >> 8) & mask), x >> 31, bias);
    }
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
    sink[t] = acc;
}
__global__ void k_cal_fill(zk::Niels28<zk::Fq377P> *tab, uint32_t n) {       // arbitrary limbs < 2^28 stand in for curve points: the arithmetic is data-independent
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = i * 2654435761u + 99u;
    zk::Niels28<zk::Fq377P> r;
    for (int k = 0; k < 14; k++) { x = x * 1664525u + 1013904223u; r.ymx.l[k] = x >> 4; x = x * 1664525u + 1013904223u; r.ypx.l[k] = x >> 4; x = x * 1664525u + 1013904223u; r.td.l[k] = x >> 4; }
    r.ymx.l[13] &= 0xffff; r.ypx.l[13] &= 0xffff; r.td.l[13] &= 0xffff;
    for (int k = 0; k < 6; k++) r.pad[k] = 0;
    tab[i] = r;
}

#include "capi_common.hpp"
extern "C" const char *zkaes_last_error(void);
namespace zk { void capi_set_error(const std::string &); }

namespace {
template <class Fn> int guardk(Fn &&fn) {
    try { zk::capi_set_error(""); fn(); return 0; }
    catch (const std::exception &e) { zk::capi_set_error(e.what()); return 1; }
    catch (...) { zk::capi_set_error("unknown error"); return 1; }
}
using zk::gpu::DevPtr; using zk::gpu::StreamGuard; using zk::gpu::WorkspaceGuard; using zk::gpu::EventGuard;
// Every helper below validates its arguments BEFORE it allocates and owns its temporaries through RAII guards (gpu.hpp): a GpuError in the middle of a call -- a bad coset
// index, an MSM whose n x windows exceeds the sort's 2^31 pairs, a device fault -- releases the stream, the buffers and the MSM workspace on the way out
// (tests/test_gpu_kernels.py::test_error_paths_release_device_memory).
int exact_log2(size_t n, const char *who) {
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    if (n == 0 || ((size_t)1 << lg) != n) throw std::invalid_argument(std::string(who) + ": n must be a power of two");
    return lg;
}
template <class Fr> void run_ntt(uint8_t *data, size_t n, int inverse, int coset_c, int lg_big) {
    const int lg = exact_log2(n, "zkaes_ntt");
    if (!data) throw std::invalid_argument("zkaes_ntt: null data");
    zk::gpu::require_device();
    StreamGuard s;
    DevPtr<Fr> a(n), b(n);
    zk::gpu::h2d(a, data, n * sizeof(Fr), s);
    if (coset_c) zk::gpu::ntt_coset<Fr>(b, a, n, lg, inverse != 0, coset_c, lg_big, s);
    else zk::gpu::ntt<Fr>(b, a, n, lg, inverse != 0, s);
    zk::gpu::d2h(data, b, n * sizeof(Fr), s);
}
template <class Fr> void run_ntt_batch(uint8_t *data, size_t n, int count, int inverse, const int *coset_c, int lg_big) {
    const int lg = exact_log2(n, "zkaes_ntt_batch");
    if (count < 1 || count > 12) throw std::invalid_argument("zkaes_ntt_batch: 1..12 transforms");
    if (!data) throw std::invalid_argument("zkaes_ntt_batch: null data");
    zk::gpu::require_device();
    StreamGuard s;
    DevPtr<Fr> a((size_t)count * n), b((size_t)count * n);
    zk::gpu::h2d(a, data, (size_t)count * n * sizeof(Fr), s);
    zk::gpu::NttJob<Fr> jobs[12];
    for (int i = 0; i < count; i++) jobs[i] = zk::gpu::NttJob<Fr>{b + (size_t)i * n, a + (size_t)i * n, coset_c ? coset_c[i] : 0};
    zk::gpu::ntt_batch<Fr>(jobs, count, n, lg, inverse != 0, lg_big, s);
    zk::gpu::d2h(data, b, (size_t)count * n * sizeof(Fr), s);
}
template <class Fq> void give_affine(const zk::Affine<Fq> &a, uint8_t *out_xy, int *out_inf) {
    if (out_inf) *out_inf = a.is_inf() ? 1 : 0;
    if (out_xy) { memcpy(out_xy, a.x.l, 48); memcpy(out_xy + 48, a.y.l, 48); }
}
template <class Curve> void run_msm(const uint8_t *bases, const uint8_t *scalars, size_t n, uint8_t *out_xy, int *out_inf, int reps, double *ms_total, double *ms_acc) {
    using Fq = typename Curve::Fq; using Fr = typename Curve::Fr;
    if (n && (!bases || !scalars)) throw std::invalid_argument("zkaes_msm: null argument");
    if (n >= ((size_t)1 << 30)) throw std::invalid_argument("zkaes_msm: too many points");
    zk::gpu::require_device();
    StreamGuard s;
    DevPtr<zk::Affine<Fq>> db(n);
    DevPtr<Fr> ds(n);
    zk::gpu::h2d(db, bases, n * 96, s); zk::gpu::h2d(ds, scalars, n * 32, s);
    DevPtr<zk::Affine28<typename Curve::FqP>> db28(n);
    zk::gpu::convert_bases<Curve>(db28, db, n, s);
    zk::gpu::sync(s);
    WorkspaceGuard ws;
    zk::XYZZ<Fq> r = zk::gpu::msm<Curve>(ws, db28, ds, n, s);   // warm-up / result
    if (reps > 0) {
        zk::gpu::MsmStats before = zk::gpu::msm_stats(false);
        EventGuard e0, e1;
        zk::gpu::event_record(e0, s);
        for (int i = 0; i < reps; i++) r = zk::gpu::msm<Curve>(ws, db28, ds, n, s);
        zk::gpu::event_record(e1, s);
        float ms = zk::gpu::event_elapsed_ms(e0, e1);
        zk::gpu::MsmStats after = zk::gpu::msm_stats(false);
        if (ms_total) *ms_total = ms / reps;
        if (ms_acc) *ms_acc = (after.accumulate_ms - before.accumulate_ms) / reps;
    }
    give_affine(r.to_affine(), out_xy, out_inf);
}
// EDWARDS (377 only): the prover's SRS path -- tables on the curve's twisted Edwards model; the bases must lie in the prime-order subgroup.  Otherwise the
// Weierstrass law (any curve point).
template <class Curve, bool EDWARDS> void run_msm_table(const uint8_t *bases, const uint8_t *scalars, size_t n, int c, uint8_t *out_xy, int *out_inf) {
    using Fq = typename Curve::Fq; using Fr = typename Curve::Fr;
    if (c < 2 || c > 22) throw std::invalid_argument("window bits must be in [2, 22]");
    if (n && (!bases || !scalars)) throw std::invalid_argument("zkaes_msm_table: null argument");
    zk::gpu::require_device();
    const size_t nt = (size_t)zk::gpu::table_windows<Curve>(c);
    if ((uint64_t)nt * n >= (1ull << 30)) throw std::invalid_argument("zkaes_msm_table: n x table copies exceeds the 2^30 base indices of one MSM");
    StreamGuard s;
    DevPtr<zk::Affine<Fq>> tab(nt * n);
    DevPtr<Fr> ds(n);
    zk::gpu::h2d(tab, bases, n * 96, s); zk::gpu::h2d(ds, scalars, n * 32, s);
    zk::gpu::build_window_tables<Curve>(tab, n, c, s);
    WorkspaceGuard ws;
    zk::Affine<Fq> a;
    if constexpr (EDWARDS) {
        DevPtr<zk::Niels28<typename Curve::FqP>> tabte(nt * n);
        zk::gpu::convert_bases_te<Curve>(tabte, tab, nt * n, s);
        a = zk::gpu::msm_table<Curve>(ws, tabte, n, 0, c, ds, n, s).to_affine();
    } else {
        DevPtr<zk::Affine28<typename Curve::FqP>> tab28(nt * n);
        zk::gpu::convert_bases<Curve>(tab28, tab, nt * n, s);
        a = zk::gpu::msm_table<Curve>(ws, tab28, n, 0, c, ds, n, s).to_affine();
    }
    give_affine(a, out_xy, out_inf);
}
template <class Curve> void run_msm_window_sums_dev(const uint8_t *bases, const uint8_t *scalars, size_t n_local, size_t n_total, void *dev_out, size_t dev_out_bytes) {
    using Fq = typename Curve::Fq; using Fr = typename Curve::Fr;
    zk::gpu::require_device();
    int c = 0, nwin = 0;
    zk::gpu::msm_sharded_plan<Curve>(n_total, &c, &nwin);
    if (!dev_out || dev_out_bytes < (size_t)nwin * sizeof(zk::XYZZ<Fq>)) throw std::invalid_argument("zkaes_msm_window_sums_dev: device buffer too small for the window sums");
    if (n_local > n_total) throw std::invalid_argument("zkaes_msm_window_sums_dev: n_local > n_total");
    if (n_local && (!bases || !scalars)) throw std::invalid_argument("zkaes_msm_window_sums_dev: null argument");
    StreamGuard s;
    DevPtr<zk::Affine<Fq>> db(n_local);
    DevPtr<Fr> ds(n_local);
    zk::gpu::h2d(db, bases, n_local * 96, s); zk::gpu::h2d(ds, scalars, n_local * 32, s);
    DevPtr<zk::Affine28<typename Curve::FqP>> db28(n_local);
    zk::gpu::convert_bases<Curve>(db28, db, n_local, s);
    WorkspaceGuard ws;
    zk::gpu::msm_window_sums_device<Curve>(ws, db28, ds, n_local, n_total, (zk::XYZZ<Fq> *)dev_out, s);    // synchronizes the stream
}
constexpr int MAX_FOLD_WORLD = 4096;        // ranks whose partial sums one fold call adds (an 8-GPU node needs 8)
template <class Curve> void run_msm_fold_dev(const void *dev_in, int world, size_t n_total, uint8_t *out_xy, int *out_inf) {
    using Fq = typename Curve::Fq;
    if (!dev_in || world < 1 || world > MAX_FOLD_WORLD) throw std::invalid_argument("zkaes_msm_fold_window_sums_dev: bad arguments (world must be 1..4096)");
    zk::gpu::require_device();
    StreamGuard s;
    give_affine(zk::gpu::msm_fold_window_sums_device<Curve>((const zk::XYZZ<Fq> *)dev_in, world, n_total, s).to_affine(), out_xy, out_inf);
}
// host-side sum of a handful of affine partial results: the "local EC add" after the all-gather of a point-range-sharded MSM
template <class Curve> void run_g1_sum(const uint8_t *points_xy, const int *inf, size_t n, uint8_t *out_xy, int *out_inf) {
    using Fq = typename Curve::Fq;
    zk::XYZZ<Fq> acc = zk::XYZZ<Fq>::inf();
    for (size_t i = 0; i < n; i++) {
        if (inf && inf[i]) continue;
        zk::Affine<Fq> a;
        memcpy(a.x.l, points_xy + 96 * i, 48); memcpy(a.y.l, points_xy + 96 * i + 48, 48);
        acc.add(zk::XYZZ<Fq>::from_affine(a));
    }
    zk::Affine<Fq> r = acc.to_affine();
    *out_inf = r.is_inf() ? 1 : 0;
    memcpy(out_xy, r.x.l, 48); memcpy(out_xy + 48, r.y.l, 48);
}
}  // namespace

extern "C" {
int zkaes_msm_sharded_plan(int curve_id, size_t n_total, int *window_bits, int *n_windows, size_t *bytes_per_rank) {
    return guardk([&] {
        int c = 0, w = 0;
        if (curve_id == 381) zk::gpu::msm_sharded_plan<zk::Bls381>(n_total, &c, &w); else if (curve_id == 377) zk::gpu::msm_sharded_plan<zk::Bls377>(n_total, &c, &w); else throw std::invalid_argument("curve_id must be 377 or 381");
        if (window_bits) *window_bits = c;
        if (n_windows) *n_windows = w;
        if (bytes_per_rank) *bytes_per_rank = (size_t)w * 192;
    });
}
int zkaes_msm_window_sums_dev(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n_local, size_t n_total, void *dev_out, size_t dev_out_bytes) {
    return guardk([&] { if (curve_id == 381) run_msm_window_sums_dev<zk::Bls381>(bases, scalars, n_local, n_total, dev_out, dev_out_bytes); else if (curve_id == 377) run_msm_window_sums_dev<zk::Bls377>(bases, scalars, n_local, n_total, dev_out, dev_out_bytes); else throw std::invalid_argument("curve_id must be 377 or 381"); });
}
int zkaes_msm_fold_window_sums_dev(int curve_id, const void *dev_in, int world, size_t n_total, uint8_t *out_xy, int *out_inf) {
    return guardk([&] { if (curve_id == 381) run_msm_fold_dev<zk::Bls381>(dev_in, world, n_total, out_xy, out_inf); else if (curve_id == 377) run_msm_fold_dev<zk::Bls377>(dev_in, world, n_total, out_xy, out_inf); else throw std::invalid_argument("curve_id must be 377 or 381"); });
}
int zkaes_g1_sum(int curve_id, const uint8_t *points_xy, const int *inf, size_t n, uint8_t *out_xy, int *out_inf) {
    return guardk([&] { if (curve_id == 381) run_g1_sum<zk::Bls381>(points_xy, inf, n, out_xy, out_inf); else if (curve_id == 377) run_g1_sum<zk::Bls377>(points_xy, inf, n, out_xy, out_inf); else throw std::invalid_argument("curve_id must be 377 or 381"); });
}
int zkaes_msm_fold_partials_dev(int curve_id, const void *dev_in, int world, uint8_t *out_xy, int *out_inf) {
    return guardk([&] {
        if (!dev_in || world < 1 || world > MAX_FOLD_WORLD) throw std::invalid_argument("zkaes_msm_fold_partials_dev: bad arguments (world must be 1..4096)");
        if (curve_id != 377 && curve_id != 381) throw std::invalid_argument("curve_id must be 377 or 381");
        zk::gpu::require_device();
        StreamGuard s;
        if (curve_id == 377) give_affine(zk::gpu::msm_fold_points_device<zk::Bls377>((const zk::XYZZ<zk::Fq377> *)dev_in, world, s).to_affine(), out_xy, out_inf);
        else give_affine(zk::gpu::msm_fold_points_device<zk::Bls381>((const zk::XYZZ<zk::Fq381> *)dev_in, world, s).to_affine(), out_xy, out_inf);
    });
}
int zkaes_msm_table(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n, int window_bits, uint8_t *out_xy, int *out_inf) {
    return guardk([&] { if (curve_id == 381) run_msm_table<zk::Bls381, false>(bases, scalars, n, window_bits, out_xy, out_inf); else if (curve_id == 377) run_msm_table<zk::Bls377, false>(bases, scalars, n, window_bits, out_xy, out_inf); else throw std::invalid_argument("curve_id must be 377 or 381"); });
}
int zkaes_msm_table_srs(const uint8_t *bases, const uint8_t *scalars, size_t n, int window_bits, uint8_t *out_xy, int *out_inf) {
    return guardk([&] { run_msm_table<zk::Bls377, true>(bases, scalars, n, window_bits, out_xy, out_inf); });
}
int zkaes_set_device(int ordinal) { return guardk([&] { HIP_CHECK(hipSetDevice(ordinal)); }); }
int zkaes_ntt(int field_id, uint8_t *data, size_t n, int inverse) {
    return guardk([&] { if (field_id == 381) run_ntt<zk::Fr381>(data, n, inverse, 0, 0); else if (field_id == 377) run_ntt<zk::Fr377>(data, n, inverse, 0, 0); else throw std::invalid_argument("field_id must be 377 or 381"); });
}
int zkaes_ntt_coset(int field_id, uint8_t *data, size_t n, int inverse, int coset_c, int lg_big) {
    return guardk([&] {
        if (coset_c <= 0) throw std::invalid_argument("zkaes_ntt_coset: coset index must be positive");
        if (field_id == 381) run_ntt<zk::Fr381>(data, n, inverse, coset_c, lg_big); else if (field_id == 377) run_ntt<zk::Fr377>(data, n, inverse, coset_c, lg_big); else throw std::invalid_argument("field_id must be 377 or 381");
    });
}
int zkaes_ntt_batch(int field_id, uint8_t *data, size_t n, int count, int inverse, const int *coset_c, int lg_big) {
    return guardk([&] {
        if (field_id == 381) run_ntt_batch<zk::Fr381>(data, n, count, inverse, coset_c, lg_big); else if (field_id == 377) run_ntt_batch<zk::Fr377>(data, n, count, inverse, coset_c, lg_big);
        else throw std::invalid_argument("field_id must be 377 or 381");
    });
}
int zkaes_msm(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n, uint8_t *out_xy, int *out_inf) {
    return guardk([&] { if (curve_id == 381) run_msm<zk::Bls381>(bases, scalars, n, out_xy, out_inf, 0, nullptr, nullptr); else if (curve_id == 377) run_msm<zk::Bls377>(bases, scalars, n, out_xy, out_inf, 0, nullptr, nullptr); else throw std::invalid_argument("curve_id must be 377 or 381"); });
}
int zkaes_msm_bench(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n, int reps, double *ms_total, double *ms_accumulate) {
    return guardk([&] { if (curve_id == 381) run_msm<zk::Bls381>(bases, scalars, n, nullptr, nullptr, reps, ms_total, ms_accumulate); else run_msm<zk::Bls377>(bases, scalars, n, nullptr, nullptr, reps, ms_total, ms_accumulate); });
}
// BLS12-377 only: n_points synthetic bases (powers of a fixed scalar times the generator, made on the device) and pseudo-random scalars;
// window_bits = 0 -> per-window buckets on the Weierstrass model (the generic path), < 0 -> per-window buckets on the twisted Edwards model (the prover's lone-call
// path), > 0 -> the precomputed-table path (Edwards).  Returns ms per MSM (whole pipeline / accumulate kernel).
int zkaes_msm_bench_synth(size_t n, int window_bits, int reps, double *ms_total, double *ms_accumulate, uint8_t *out_xy) {
    return guardk([&] {
        using Fq = zk::Fq377; using Fr = zk::Fr377;
        if (!ms_total || !ms_accumulate || reps < 1 || n == 0 || window_bits > 22) throw std::invalid_argument("zkaes_msm_bench_synth: bad arguments");
        zk::gpu::require_device();
        const size_t nt = window_bits > 0 ? (size_t)zk::gpu::table_windows<zk::Bls377>(window_bits) : 1;
        if ((uint64_t)nt * n >= (1ull << 30)) throw std::invalid_argument("zkaes_msm_bench_synth: n x table copies exceeds the 2^30 base indices of one MSM");
        StreamGuard s;
        DevPtr<zk::Affine<Fq>> tab(nt * n);
        zk::Affine<Fq> g; for (int i = 0; i < 12; i++) { g.x.l[i] = G1_377_X_MONT[i]; g.y.l[i] = G1_377_Y_MONT[i]; }
        Fr beta = Fr::from_u64(0x9e3779b97f4a7c15ull) * Fr::from_u64(0xc2b2ae3d27d4eb4full);
        zk::gpu::fixed_base_powers<zk::Bls377>(tab, g, beta, 1, n, s);
        if (window_bits > 0) zk::gpu::build_window_tables<zk::Bls377>(tab, n, window_bits, s);
        // window_bits == 0: the generic path (Weierstrass model, XYZZ buckets, 112-byte bases); < 0 or > 0: the prover's SRS path on the twisted Edwards model
        const bool edwards = window_bits != 0;
        DevPtr<zk::Affine28<zk::Fq377P>> tab28;
        DevPtr<zk::Niels28<zk::Fq377P>> tabte;
        if (edwards) { tabte.alloc(nt * n); zk::gpu::convert_bases_te<zk::Bls377>(tabte, tab, nt * n, s); }
        else { tab28.alloc(nt * n); zk::gpu::convert_bases<zk::Bls377>(tab28, tab, nt * n, s); }
        std::vector<Fr> sc(n);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < n; i++) { for (int k = 0; k < 8; k += 2) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; sc[i].l[k] = (uint32_t)x; sc[i].l[k + 1] = (uint32_t)(x >> 32); } sc[i].l[7] &= 0x0fffffffu; }
        DevPtr<Fr> ds(n);
        zk::gpu::h2d(ds, sc.data(), n * 32, s);
        zk::gpu::sync(s);
        WorkspaceGuard ws;
        auto run = [&]() -> zk::XYZZ<Fq> {
            if (edwards) return window_bits > 0 ? zk::gpu::msm_table<zk::Bls377>(ws, tabte, n, 0, window_bits, ds, n, s) : zk::gpu::msm<zk::Bls377>(ws, tabte.get(), ds.get(), n, s);
            return window_bits > 0 ? zk::gpu::msm_table<zk::Bls377>(ws, tab28, n, 0, window_bits, ds, n, s) : zk::gpu::msm<zk::Bls377>(ws, tab28.get(), ds.get(), n, s);
        };
        give_affine(run().to_affine(), out_xy, nullptr);
        zk::gpu::MsmStats before = zk::gpu::msm_stats(false);
        EventGuard e0, e1;
        zk::gpu::event_record(e0, s);
        for (int i = 0; i < reps; i++) run();
        zk::gpu::event_record(e1, s);
        float ms = zk::gpu::event_elapsed_ms(e0, e1);
        zk::gpu::MsmStats after = zk::gpu::msm_stats(false);
        *ms_total = ms / reps; *ms_accumulate = (after.accumulate_ms - before.accumulate_ms) / reps;
    });
}

int zkaes_stream_copy_bench(size_t bytes, int reps, double *gb_per_s) {
    return guardk([&] {
        if (bytes < 4096 || reps < 1 || !gb_per_s) throw std::invalid_argument("zkaes_stream_copy_bench: bytes >= 4096, reps >= 1");
        zk::gpu::require_device();
        StreamGuard s;
        const size_t n16 = bytes / 16;
        DevPtr<uint4> a(n16), b(n16);
        zk::gpu::dzero(a, n16 * 16, s);
        hipStream_t hs = (hipStream_t)s.s;
        k_stream_copy<<<(unsigned)((n16 + 1023) / 1024), 256, 0, hs>>>(b, a, n16);
        EventGuard e0, e1;
        zk::gpu::event_record(e0, s);
        for (int i = 0; i < reps; i++) k_stream_copy<<<(unsigned)((n16 + 1023) / 1024), 256, 0, hs>>>(i & 1 ? a.get() : b.get(), i & 1 ? b.get() : a.get(), n16);
        zk::gpu::event_record(e1, s);
        float ms = zk::gpu::event_elapsed_ms(e0, e1);
        *gb_per_s = 2.0 * (double)(n16 * 16) * reps / 1e9 / (ms / 1e3);   // read + write
    });
}
int zkaes_int_rate_bench(double seconds, double out[8]) {
    return guardk([&] {
        if (!out || !(seconds > 0.0) || seconds > 30.0) throw std::invalid_argument("zkaes_int_rate_bench: 0 < seconds <= 30, out != NULL");
        zk::gpu::require_device();
        using P = zk::Fq377P; using G = zk::FpMsm<P>;
        StreamGuard s;
        hipStream_t hs = (hipStream_t)s.s;
        const int GRID_A = 4096, ITERS_A = 300, GRID_B = 8192, ITERS_B = 83;      // A: 4 waves per SIMD, 600 products per lane; B: k_accumulate's launch shape for 2^19 buckets of ~83 points
        const uint32_t NREC = 4096;                                                // 768 KB of records: L2-resident
        DevPtr<CalStamp> st(GRID_B);
        DevPtr<G> sink_a((size_t)GRID_A * 64);
        DevPtr<zk::AccTE<P>> sink_b((size_t)GRID_B * 64);
        DevPtr<zk::Niels28<P>> tab(NREC);
        k_cal_fill<<<NREC / 256, 256, 0, hs>>>(tab, NREC);
        G ga; for (int k = 0; k < 14; k++) ga.l[k] = 0x0123457u + 977u * k;
        std::vector<double> rate_a, mhz_a, rate_b, mhz_b, cyc_b;
        std::vector<CalStamp> h(GRID_B);
        auto median = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
        auto stamps = [&](int grid, double per_wave_units, double *mhz, double *cyc_per_unit) {
            zk::gpu::d2h(h.data(), st, (size_t)grid * sizeof(CalStamp), s);
            std::vector<double> m(grid), c(grid);
            for (int i = 0; i < grid; i++) { m[i] = (double)h[i].cyc / ((double)h[i].rt / 1e8) / 1e6; c[i] = (double)h[i].cyc / per_wave_units; }
            *mhz = median(m); *cyc_per_unit = median(c);
        };
        EventGuard e0, e1;
        const auto t_begin = std::chrono::steady_clock::now();
        int rounds = 0;
        // warm-up launches are part of the loop: the first rounds run while the clocks settle and the median discards them
        while (rounds < 3 || std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < seconds) {
            double mhz, cyc;
            zk::gpu::event_record(e0, s);
            k_cal_fqmul<<<GRID_A, 64, 0, hs>>>(st, sink_a, ga, ITERS_A);
            zk::gpu::event_record(e1, s);
            stamps(GRID_A, 2.0 * ITERS_A, &mhz, &cyc);
            rate_a.push_back(2.0 * ITERS_A * 64.0 * GRID_A / (zk::gpu::event_elapsed_ms(e0, e1) * 1e-3)); mhz_a.push_back(mhz);
            zk::gpu::event_record(e0, s);
            k_cal_hot_loop<<<GRID_B, 64, 0, hs>>>(st, sink_b, tab, NREC - 1, ITERS_B);
            zk::gpu::event_record(e1, s);
            stamps(GRID_B, (double)ITERS_B, &mhz, &cyc);
            rate_b.push_back((double)ITERS_B * 64.0 * GRID_B / (zk::gpu::event_elapsed_ms(e0, e1) * 1e-3)); mhz_b.push_back(mhz); cyc_b.push_back(cyc);
            rounds++;
        }
        out[0] = median(rate_a); out[1] = median(mhz_a); out[2] = median(rate_b); out[3] = median(mhz_b); out[4] = median(cyc_b); out[5] = (double)rounds; out[6] = 0; out[7] = 0;
    });
}
int zkaes_mem_info(uint64_t *free_bytes, uint64_t *total_bytes) {
    return guardk([&] {
        zk::gpu::require_device();
        size_t f = 0, t = 0;
        HIP_CHECK(hipMemGetInfo(&f, &t));
        if (free_bytes) *free_bytes = f;
        if (total_bytes) *total_bytes = t;
    });
}
}
