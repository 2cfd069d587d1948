// csrc/gpu.hpp -- host-callable launch wrappers around the gfx950 kernels of libzkaes.
// Everything here takes/returns DEVICE pointers unless a name says host; all launches go to the given stream.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "ec.cuh"
#include "te28.cuh"

namespace zk {
namespace gpu {

struct GpuError : std::runtime_error { using std::runtime_error::runtime_error; };
typedef void *stream_t;   // hipStream_t

// ---- runtime
int device_count();                       // 0 when no GPU / driver
void require_device();
int current_device();
void set_device(int ordinal);                    // throws GpuError("no HIP device ...") -- the product never falls back to the CPU
void *dmalloc(size_t bytes);
#ifdef ZKAES_MEASURE
int knockin();                                   // ZKAES_KNOCKIN measurement mask (runtime.hip; measurement builds only)
#endif
size_t mem_free_bytes();                         // free device memory right now (hipMemGetInfo)
void dfree(void *p);
void h2d(void *dst, const void *src, size_t bytes, stream_t s);
void d2h(void *dst, const void *src, size_t bytes, stream_t s);
void d2d(void *dst, const void *src, size_t bytes, stream_t s);
void dzero(void *dst, size_t bytes, stream_t s);
void sync(stream_t s);                    // wait for the stream: spins (lone calls) or polls with sleeps while a ThroughputWaits scope is alive
// RAII marker of a multi-proof call: host threads of its prover contexts wait by polling + nanosleep instead of spinning (runtime.hip)
bool throughput_mode();                   // true while a ThroughputWaits scope is alive (a multi-proof call is in flight): kernels may pick the throughput variant of a step
struct ThroughputWaits { explicit ThroughputWaits(bool on); ~ThroughputWaits(); ThroughputWaits(const ThroughputWaits &) = delete; ThroughputWaits &operator=(const ThroughputWaits &) = delete; private: bool on_; };
stream_t stream_create();
stream_t stream_create_background();      // lowest device priority: its kernels yield workgroup slots to every other stream's
void stream_destroy(stream_t s);
// event timing on a stream (ms)
void *event_create();
void event_record(void *ev, stream_t s);
void stream_wait_event(stream_t s, void *ev);   // work queued on s after this call starts only when the event (recorded on another stream) has completed: no host wait
float event_elapsed_ms(void *start, void *stop);
void event_destroy(void *ev);

// RAII owners of the raw handles above: an exception between an allocation and its release must not leak device memory, a stream or an MSM workspace
// (every kernel-level C entry point and the setup helpers hold their temporaries through these)
template <class T> struct DevPtr {
    T *p = nullptr;
    DevPtr() = default;
    explicit DevPtr(size_t count) : p((T *)dmalloc(count * sizeof(T))) {}
    DevPtr(const DevPtr &) = delete; DevPtr &operator=(const DevPtr &) = delete;
    DevPtr(DevPtr &&o) noexcept : p(o.p) { o.p = nullptr; }
    DevPtr &operator=(DevPtr &&o) noexcept { if (this != &o) { dfree(p); p = o.p; o.p = nullptr; } return *this; }
    ~DevPtr() { dfree(p); }
    void alloc(size_t count) { dfree(p); p = nullptr; p = (T *)dmalloc(count * sizeof(T)); }
    T *get() const { return p; }
    operator T *() const { return p; }
};
struct StreamGuard {
    stream_t s = nullptr;
    StreamGuard() : s(stream_create()) {}
    StreamGuard(const StreamGuard &) = delete; StreamGuard &operator=(const StreamGuard &) = delete;
    ~StreamGuard() { stream_destroy(s); }
    operator stream_t() const { return s; }
};
struct EventGuard {
    void *e = nullptr;
    EventGuard() : e(event_create()) {}
    EventGuard(const EventGuard &) = delete; EventGuard &operator=(const EventGuard &) = delete;
    ~EventGuard() { event_destroy(e); }
    operator void *() const { return e; }
};

// ---- op lists (debug; zkaes_pk_op_lists): while a recording is open, every transform and every MSM the library launches is appended -- the ACTUAL lists the whole-proof
// roofline of SURVEY.md 8(d) (bytes = W + S + T + M, T = sum 64 n_i, M = sum 128 m_j) is computed from, instead of a literal in a document.  Process-global: record with no
// other proof in flight.  The hooks cost one relaxed atomic load when no recording is open.
enum : int { OP_MSM_BUCKETS = 0, OP_MSM_SECOND_BASES = 1, OP_MSM_CLASS_SUM = 2 };      // Pippenger over buckets | the same prepared scalars against a second base array | class sum over a Lagrange-basis SRS
struct OpRecord { std::vector<std::pair<uint64_t, int>> ntt /* (points, transforms sharing the launch) */, msm /* (points, kind) */; };
void oplog_begin();
OpRecord oplog_end();
void oplog_ntt(uint64_t n, int count);
void oplog_msm(uint64_t points, int kind);

// ---- NTT (kernels_ntt.hip).  Tables are created lazily per (field, log n) and cached for the life of the process.
// dst <- NTT(src zero-padded from in_len to 2^lg); dst may equal src only if in_len == 2^lg is NOT required (out of place first pass
// reads src completely before any tile of dst is written only when dst != src; pass distinct buffers).
template <class Fr> void ntt(Fr *dst, const Fr *src, size_t in_len, int lg, bool inverse, stream_t s);
// The same transform on the coset g D of the size-2^lg domain D, g = W^coset_c with W the primitive root of the domain of size 2^lg_big (lg_big > lg, 0 < coset_c < 2^(lg_big - lg)):
// forward: dst[i] = p(g w^i) for the coefficients src[0 .. in_len); inverse: the coefficients of the polynomial of degree < 2^lg with those values.  No extra pass: the scaling
// by g^(+-k) rides on the first pass's gather / the last pass's store.  (Together, cosets 0 .. 2^(lg_big - lg) - 1 are the larger domain.)
template <class Fr> void ntt_coset(Fr *dst, const Fr *src, size_t in_len, int lg, bool inverse, int coset_c, int lg_big, stream_t s);
// up to 12 independent transforms of ONE shape (size, direction, input length) in shared launches; coset_c = 0: plain transform, > 0: the coset W^c H as in ntt_coset
template <class Fr> struct NttJob { Fr *dst; const Fr *src; int coset_c; };
template <class Fr> void ntt_batch(const NttJob<Fr> *jobs, int count, size_t in_len, int lg, bool inverse, int lg_big, stream_t s);
// A coset whose generator g is not a root of unity (round 3 uses the field's multiplicative generator): coset_power_table builds g^i (i < n) once per key in the kernel's
// reduced-radix form (free with dfree); ntt_scaled(.., table of g^i) evaluates on g D, ntt_scaled(.., inverse = true, table of g^-i) interpolates from there.
template <class Fr> void *coset_power_table(const Fr &g, size_t n, stream_t s);
template <class Fr> void ntt_scaled(Fr *dst, const Fr *src, size_t in_len, int lg, bool inverse, const void *table, stream_t s);
template <class Fr> const Fr *domain_elements(int lg);   // device table g^i, i < 2^lg  (built lazily)

// ---- MSM (kernels_msm.hip): sum_i scalars[i] * bases[i]; scalars in Montgomery form; result returned to the host (syncs the stream)
// `ws` holds the sort / bucket scratch of one in-flight MSM: use one workspace per stream (per prover context)
struct MsmWorkspace;
MsmWorkspace *msm_workspace_create();
void msm_workspace_destroy(MsmWorkspace *ws);
struct WorkspaceGuard {
    MsmWorkspace *ws = nullptr;
    WorkspaceGuard() : ws(msm_workspace_create()) {}
    WorkspaceGuard(const WorkspaceGuard &) = delete; WorkspaceGuard &operator=(const WorkspaceGuard &) = delete;
    ~WorkspaceGuard() { msm_workspace_destroy(ws); }
    operator MsmWorkspace *() const { return ws; }
};
// Bases are consumed in the reduced-radix form (Affine28, ff28.cuh): convert once with convert_bases (the SRS at key synthesis).
template <class Curve> void convert_bases(Affine28<typename Curve::FqP> *dst, const Affine<typename Curve::Fq> *src, size_t n, stream_t s);
template <class Curve>
XYZZ<typename Curve::Fq> msm(MsmWorkspace *ws, const Affine28<typename Curve::FqP> *bases, const typename Curve::Fr *scalars, size_t n, stream_t s);
// The same MSM in two steps.  msm_prepare: digits + sort + bucket ranges of scal1 (bases 0..n1) and optionally scal2 (bases val_off2..val_off2+n2
// of the same array) as ONE Pippenger instance; msm_finish: accumulate from `bases` + reduce.  msm_finish may be called several times on one
// prepared state with different base arrays (the plain and the shifted commitment of a degree-bounded polynomial share their scalars).
template <class Curve>
void msm_prepare(MsmWorkspace *ws, const typename Curve::Fr *scal1, size_t n1, const typename Curve::Fr *scal2, size_t n2, size_t val_off2, stream_t s, int force_c = 0);
template <class Curve>
XYZZ<typename Curve::Fq> msm_finish(MsmWorkspace *ws, const Affine28<typename Curve::FqP> *bases, stream_t s);
// BLS12-377 only -- the prover's path over its fixed SRS: bases precomputed on the curve's twisted Edwards model (te28.cuh: Niels28 = (y - x, y + x, 2 d x y),
// 168 B padded to a 64-byte aligned 192-byte record), bucket additions of 7 field products instead of 10 and no special cases.  convert_bases_te maps Weierstrass affine points (which MUST lie in the
// prime-order subgroup; a point of order 2 or 4 is refused) to that form; msm / msm_finish / msm_table / class_sum are overloaded on the base type and
// return the same Weierstrass XYZZ results as their Affine28 versions.
template <class Curve> void convert_bases_te(Niels28<typename Curve::FqP> *dst, const Affine<typename Curve::Fq> *src, size_t n, stream_t s);
template <class Curve>
XYZZ<typename Curve::Fq> msm(MsmWorkspace *ws, const Niels28<typename Curve::FqP> *bases, const typename Curve::Fr *scalars, size_t n, stream_t s);
template <class Curve>
XYZZ<typename Curve::Fq> msm_finish(MsmWorkspace *ws, const Niels28<typename Curve::FqP> *bases, stream_t s);
// the prepared scalars against TWO base arrays (plain + shifted powers of a degree-bounded commitment): both accumulations back to back, one reduction, one host wait
template <class Curve>
void msm_finish2(MsmWorkspace *ws, const Niels28<typename Curve::FqP> *bases, const Niels28<typename Curve::FqP> *bases2, XYZZ<typename Curve::Fq> *out, stream_t s);
template <class Curve>
XYZZ<typename Curve::Fq> msm_table(MsmWorkspace *ws, const Niels28<typename Curve::FqP> *tables, size_t stride, size_t off, int c, const typename Curve::Fr *scalars, size_t n, stream_t s);
template <class Curve>
bool class_sum(MsmWorkspace *ws, const Niels28<typename Curve::FqP> *bases, const int8_t *vals, size_t n, XYZZ<typename Curve::Fq> *out, stream_t s);
// ONE MSM sharded by point range over ranks: msm_sharded_plan gives the window plan of the whole MSM (all ranks agree on it);
// msm_window_sums_device runs this rank's slice and leaves n_windows XYZZ window sums (192 B each, standard Montgomery form) at dev_out in HBM
// -- the payload of the one all-gather; msm_fold_window_sums_device adds `world` such blocks (rank-major) per window on the device and finishes
// with the Horner pass over the windows.
template <class Curve> void msm_sharded_plan(size_t n_total, int *window_bits, int *n_windows);
template <class Curve>
void msm_window_sums_device(MsmWorkspace *ws, const Affine28<typename Curve::FqP> *bases, const typename Curve::Fr *scalars, size_t n_local, size_t n_total,
                            XYZZ<typename Curve::Fq> *dev_out, stream_t s);
template <class Curve>
XYZZ<typename Curve::Fq> msm_fold_window_sums_device(const XYZZ<typename Curve::Fq> *dev_in, int world, size_t n_total, stream_t s);
// The same sharding on the prover's own path (BLS12-377 SRS on the twisted Edwards model, window tables, ONE bucket set): a rank's share of the MSM is ONE XYZZ point,
// left in device memory at dev_out; msm_fold_points_device adds `world` such points (rank-major rows of the all-gather) on the device.
template <class Curve>
void msm_table_sum_device(MsmWorkspace *ws, const Niels28<typename Curve::FqP> *tables, size_t stride, size_t off, int c, const typename Curve::Fr *scalars, size_t n,
                          XYZZ<typename Curve::Fq> *dev_out, stream_t s);
template <class Curve>
XYZZ<typename Curve::Fq> msm_fold_points_device(const XYZZ<typename Curve::Fq> *dev_in, int world, stream_t s);
// Precomputed-window variant for FIXED bases (the SRS): tables[j * stride + i] = 2^(c j) * P_i for j < table_windows(c).  With one table copy
// per window ALL windows share ONE set of 2^(c-1) signed-digit buckets, so the bucket reduction is paid once and c can grow to 20-22
// (12-13 windows instead of 15): fewer (point, window) pairs = fewer mixed adds in k_accumulate.
// build_window_tables fills copies 1.. from copy 0 (already in tables[0..stride)).
// msm_prepare_table + msm_finish(ws, tables [+ constant index shift]) mirror msm_prepare / msm_finish: element i of scalar vector v names table
// entry w * stride + off_v + i in window w, so two vectors over two index ranges (plain + shifted powers) share one Pippenger instance, and one
// prepared state can be finished against two index shifts (the plain and the shifted commitment of a degree-bounded polynomial).
template <class Curve> int table_windows(int c);
template <class Curve> void build_window_tables(Affine<typename Curve::Fq> *tables, size_t stride, int c, stream_t s);
// one step of the same: next = 2^(width of window j-1) * prev = table copy j from copy j-1 (`count` points; asynchronous on s)
template <class Curve> void table_next(Affine<typename Curve::Fq> *next, const Affine<typename Curve::Fq> *prev, size_t count, int c, int j, stream_t s);
template <class Curve>
void msm_prepare_table(MsmWorkspace *ws, const typename Curve::Fr *scal1, size_t n1, size_t off1, const typename Curve::Fr *scal2, size_t n2, size_t off2, int c, size_t stride, stream_t s);
template <class Curve>
XYZZ<typename Curve::Fq> msm_table(MsmWorkspace *ws, const Affine28<typename Curve::FqP> *tables, size_t stride, size_t off, int c, const typename Curve::Fr *scalars, size_t n, stream_t s);
// out[i] = (beta^(from+i)) * base for i < count   (KZG powers; fixed-base windows)   -- device output
template <class Curve>
void fixed_base_powers(Affine<typename Curve::Fq> *out, const Affine<typename Curve::Fq> &base, const typename Curve::Fr &beta, size_t from, size_t count, stream_t s);
// out[i] = scalars[i] * base for device-resident scalars
template <class Curve>
void fixed_base_scalars(Affine<typename Curve::Fq> *out, const Affine<typename Curve::Fq> &base, const typename Curve::Fr *scalars, size_t count, stream_t s);
// sum_i vals[i] * bases[i] for small integers |vals[i]| <= 2 (commitments of 0/1-valued evaluation vectors in a Lagrange-basis SRS): about one
// mixed add per non-zero entry instead of one per window.  Returns false (result unusable) if a value is out of range or an addition
// degenerates -- callers then use the generic MSM.  Synchronizes the stream.
template <class Curve>
bool class_sum(MsmWorkspace *ws, const Affine28<typename Curve::FqP> *bases, const int8_t *vals, size_t n, XYZZ<typename Curve::Fq> *out, stream_t s);
// MSM-stage timing hooks for bench.py (accumulated kernel time of the bucket-accumulation kernel, measured with events)
struct MsmStats { double accumulate_ms = 0; double total_ms = 0; uint64_t points = 0; uint64_t launches = 0; uint64_t pairs = 0; };   // pairs = sum of points x windows
MsmStats msm_stats(bool reset);   // process-wide totals (thread-safe)

}  // namespace gpu
}  // namespace zk

// =====================================================================================================================
// Marlin-side kernels over the BLS12-377 scalar field (kernels_poly.hip, kernels_witness.hip)
namespace zk {
namespace gpu {
using F = Fr377;

void poly_set_at(F *p, size_t idx, const F &val, stream_t s);                 // p[idx] = val
void poly_add_at(F *p, size_t idx, const F &val, stream_t s);                 // p[idx] += val
void poly_scale(F *p, const F &sc, size_t n, stream_t s);                     // p[i] *= sc
// out[i] = sum_j scalars[j] * polys[j][i] for i < n, each polynomial contributing only below its own length (1..8 terms)
void poly_lincomb_n(F *out, size_t n, const F *const *polys, const size_t *lens, const F *scalars, int count, stream_t s);
// q = p / (X^m - 1) (len - m coefficients), rem = remainder (m coefficients); requires len > m
void divide_by_vanishing(F *q, F *rem, const F *p, size_t len, size_t m, stream_t s, F *scratch = nullptr, size_t scratch_len = 0);
// q = p / (X - z) (len - 1 coefficients, remainder dropped); scratch >= divide_by_linear_scratch(len) elements (block values and carries of the blocked recurrence)
void divide_by_linear(F *q, const F *p, size_t len, const F &z, F *scratch, size_t scratch_elems, stream_t s);
size_t divide_by_linear_scratch(size_t len);   // field elements of scratch a division of `len` coefficients needs
// p(x) returned to the host (synchronizes); scratch >= ceil(len/64) + 1 elements
F poly_eval(const F *p, size_t len, const F &x, F *scratch, stream_t s);       // scratch: 8 + poly_eval_scratch(len) elements
size_t poly_eval_scratch(size_t len);
// out[i] = p[i](x[i]), 1..8 polynomials: all launches, then one wait and one copy.  scratch: 8 + sum poly_eval_scratch(len[i]) elements
void poly_eval_multi(const F *const *p, const size_t *len, const F *x, int count, F *out, F *scratch, size_t scratch_elems, stream_t s);
// in-place batch inversion, zeros stay zero; every output optionally multiplied by `post`
void batch_inverse(F *v, size_t n, const F *post_or_null, stream_t s);
// r(a, X) = (a^n - X^n) / (a - X) (Marlin's u_H(a, X)) on `ncosets` <= 3 cosets g[c] H of the size-n domain H (elems = its n elements, n = 2^lg_n):
// out[c][i] = r(a, g[c] h_i) = prod_{k < lg_n} (a^(2^k) + (g[c] h_i)^(2^k)) -- a product tree, two products per node, NO inversion and no transform.  With g = 1 this is
// v_H(a) / (a - h_i).  scratch: >= vanishing_quotient_scratch(lg_n, ncosets) field elements.
size_t vanishing_quotient_scratch(int lg_n, int ncosets);
void vanishing_quotient_evals(F *const *out, const F *g, int ncosets, const F &a, const F *elems, uint32_t n, int lg_n, F *scratch, size_t scratch_elems, stream_t s);
// round 3: out[kappa] = (ea va + eb vb + ec vc)[kappa] rb[ci[kappa]] ra[ri[kappa]] with ra[i] = v_H(alpha) / (alpha - h_i), rb[i] = v_H(beta) / (beta - h_i): the values of f on K
void f_evals_from_tables(F *out, const F *va, const F *vb, const F *vc, const F &ea, const F &eb, const F &ec, const F *ra, const F *rb, const uint32_t *ri, const uint32_t *ci, size_t k,
                         stream_t s);

// ---- witness generation + sparse products
void upload_sbox(const uint8_t table[256]);
// one thread per ECB block (+ the key schedule): fills the trace of `nproofs` chunk-proofs of `nblocks` blocks each
void aes_trace(uint8_t *trace, size_t trace_stride, const uint8_t *msgs, const uint8_t *keys, uint32_t nproofs, uint32_t nblocks, stream_t s);
// z[col] (0/1 bytes) for every column, by descriptor
void witness_expand(uint8_t *z, const uint32_t *desc, uint32_t ncols, const uint8_t *trace, const uint32_t *sbox_in_off, const uint32_t *sbox_tmpl, stream_t s);
// out[r] = sum_i coeff[i] * z[col[i]] as a field element, rows with no entries give 0 (out has `rows_out` >= rows entries, tail zeroed)
// small_out (may be null): the same values as int8 (saturated to +-127) for the Lagrange-basis commitment path
void spmv_bits(F *out, int8_t *small_out, size_t rows_out, const uint32_t *rowptr, const uint32_t *col, const int64_t *coeff, size_t rows, const uint8_t *z, stream_t s);
// w evaluation classes on H: 0 at the positions of X, else the witness bit (ark-marlin prover_first_round indexing)
void w_classes(int8_t *out, const uint8_t *z, uint32_t n, uint32_t m, uint32_t num_witness, stream_t s);
// Lagrange-basis scalars at the KZG trapdoor beta: lag[k] = L_k(beta) = (beta^n - 1) g^k / (n (beta - g^k));  lag_w[k] = 0 on X, else L_k(beta) / v_X(beta)
void lagrange_scalars(F *lag, F *lag_w, const F *elems, const F &beta, uint32_t n, uint32_t m, stream_t s);
// w evaluations on H (ark-marlin prover_first_round): out[k] = 0 if k % ratio == 0 else w_ext[k - k/ratio - 1] - x_evals[k]
void w_evals(F *out, const uint8_t *z, const F *x_evals, uint32_t n, uint32_t m, uint32_t num_witness, stream_t s);
void bits_to_field(F *out, const uint8_t *z, size_t n, stream_t s);
// t evaluations on H: out[h] = sum over entries (row r, weight w = eta_M * coeff) in column bucket h of  w * r_alpha[r].
// The indexer buckets A, B, C by (re-indexed) column and cuts every bucket into segments of <= T_SEG entries:
// seg_start/seg_end[nseg] (entry ranges), col_seg_ptr[n+1] (segments of column h), row[], mat[] (0/1/2), coeff[]; partial = nseg scratch elements
// heavy[n_heavy] = columns with more than T_HEAVY_SEGMENTS segments (summed by a whole workgroup each)
constexpr uint32_t T_SEG = 32, T_HEAVY_SEGMENTS = 256;
void t_evals(F *out, uint32_t n, F *partial, uint32_t nseg, const uint32_t *col_seg_ptr, const uint32_t *seg_start, const uint32_t *seg_end, const uint32_t *heavy, uint32_t n_heavy,
             const uint32_t *row, const uint8_t *mat, const int64_t *coeff, const F *r_alpha, const F &eta_a, const F &eta_b, const F &eta_c, stream_t s);
// `count` field elements drawn exactly as ark-ff's Fp::rand would draw them from a ChaCha block RNG (rand_chacha layout: 64-bit block
// counter, word stream) positioned at word `word_pos`: 8 words per candidate, top bits shaved, candidates >= p rejected, limbs used AS the
// Montgomery form.  Candidates are generated and filtered in parallel (flag + exclusive scan + compaction).  Returns the stream position
// (in words) after the last accepted candidate so the host RNG can continue from there.  scratch: >= 2.2 * count * 9 words.
uint64_t chacha_field_stream(F *out, size_t count, const uint32_t key[8], int rounds, uint64_t word_pos, void *scratch, size_t scratch_bytes, stream_t s);
// mask-polynomial fix-up of ark-marlin: p[0] -= p[0] + p[n] + p[2n]
void mask_fixup(F *p, size_t n, stream_t s);
// Round 2 on cosets of H (marlin.cpp second round): out[i] = r[i] (eta_a A + eta_b B + eta_c A B) - t[i] Z with A = za[i] + ca, B = zb[i] + cb, Z = z[i] + cz (the constants are
// the blinding terms rho (X^|H| - 1), constant on a coset); z_evals_h = the full assignment on H as field elements; q1_combine turns the interpolants Q0 (on H), Q1, Q3 (on the
// cosets W H, W^3 H) of q_1 - mask into h_1 (2n coefficients) and g_1 (n - 1), adding the mask's own quotient and remainder by X^n - 1.
void q1_coset_pointwise(F *out, const F *r, const F *za, const F *zb, const F *t, const F *z, const F &ca, const F &cb, const F &cz, const F &eta_a, const F &eta_b, const F &eta_c,
                        size_t n, stream_t s);
void z_evals_h(F *out, const uint8_t *z, uint32_t n, uint32_t m, uint32_t num_witness, stream_t s);
void q1_combine(F *h1, F *g1, const F *q0, const F *q1, const F *q3, const F *mask, const F &inv2, const F &inv2zeta, size_t n, stream_t s);
// out[j] = in[j] * g^j for j < n (in zero-padded beyond in_len): coefficients of p(g X)
void coset_scale(F *out, const F *in, const F &g, size_t in_len, size_t n, stream_t s);
// round 3 on one coset of K: out = ((ea va + eb vb + ec vc) - (alpha beta - alpha row - beta col + row_col) f) * vinv, all arrays = values on the coset
void h2_coset(F *out, const F *row, const F *col, const F *va, const F *vb, const F *vc, const F *rc, const F *f, const F &alpha, const F &beta, const F &alpha_beta,
              const F &ea, const F &eb, const F &ec, const F &vinv, size_t k, stream_t s);                                             // acc -= b * f
// indexer (one-time): row/col/row_col/val_* evaluations on K from the joint matrix entries; tmp = k scratch elements
void index_evals(F *row, F *col, F *rowcol, F *va, F *vb, F *vc, F *tmp, const uint32_t *ci, const uint32_t *ri, const int64_t *ca, const int64_t *cb, const int64_t *cc, size_t cnt,
                 size_t k, const F *elems, uint32_t n, stream_t s);
void z_poly_from_w(F *zp, const F *w, size_t wlen, const F *x_poly, uint32_t m, size_t n, stream_t s);          // w * v_X + x  (n + 1 coefficients)

}  // namespace gpu
}  // namespace zk
