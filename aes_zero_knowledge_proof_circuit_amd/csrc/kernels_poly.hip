// csrc/kernels_poly.hip -- elementwise / scan / reduction kernels over Fr(BLS12-377) used by the Marlin prover rounds.
//
// They replace the dense-polynomial plumbing of ark-poly 0.3.0 (DensePolynomial add/mul_by_vanishing/divide_by_vanishing_poly,
// `p / (X - z)`, evaluate, batch_inversion of ark-ff) that ark-marlin's prover_{first,second,third}_round and
// KZG10::open call.  All are HBM-streaming: one 32-byte element per lane per access.
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include "ff29.cuh"
#include "hip_util.hpp"

namespace zk {
namespace gpu {

#define GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

__global__ void k_set_at(F *p, size_t idx, F val, bool add) { if (threadIdx.x == 0 && blockIdx.x == 0) p[idx] = add ? p[idx] + val : val; }
void poly_set_at(F *p, size_t idx, const F &val, stream_t s) { hipLaunchKernelGGL(k_set_at, dim3(1), dim3(64), 0, (hipStream_t)s, p, idx, val, false); HIP_LAUNCH_CHECK(); }
void poly_add_at(F *p, size_t idx, const F &val, stream_t s) { hipLaunchKernelGGL(k_set_at, dim3(1), dim3(64), 0, (hipStream_t)s, p, idx, val, true); HIP_LAUNCH_CHECK(); }

__global__ void k_scale(F *p, F sc, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * sc; }
void poly_scale(F *p, const F &sc, size_t n, stream_t s) { if (!n) return; hipLaunchKernelGGL(k_scale, GRID(n), 0, (hipStream_t)s, p, sc, n); HIP_LAUNCH_CHECK(); }
// ---- linear combinations and the pointwise kernels of rounds 2 / 3 on the NTT's reduced-radix field (ff29.cuh): data is only re-limbed (x R stays x R), scalars travel as
// s R' ("twiddle form", converted on the host), and a SUM of up to four products shares ONE Montgomery reduction (Fp29::dot) -- an 8-term combination is 8 x 81 + 2 x 81
// multiply-accumulates instead of 8 full 8 x 32-bit CIOS products with their carry fix-ups (~3x fewer instructions; exact arithmetic, identical results).
using G29 = Fp29<Fr377P>;
__device__ __forceinline__ G29 ld29(const F &x) { return G29::split(x.l); }
__device__ __forceinline__ F st29(const G29 &g) { F r; g.template canonical<1>().pack(r.l); return r; }        // values < 4 p
// out[i] = sum_j sc[j] * p[j][i] over the (up to 8) polynomials long enough to have a coefficient i: the opening combinations in ONE pass, two dot products of four terms
struct LincombArgs { const F *p[8]; size_t len[8]; G29 sc[8]; int count; };
__global__ void k_lincomb_n(F *__restrict__ out, LincombArgs a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G29 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (j < a.count && i < a.len[j]) ? ld29(a.p[j][i]) : G29::zero();
    G29 acc = G29::dot<4>(v, a.sc);
    if (a.count > 4) acc = acc + G29::dot<4>(v + 4, a.sc + 4);
    out[i] = st29(acc);
}
void poly_lincomb_n(F *out, size_t n, const F *const *polys, const size_t *lens, const F *scalars, int count, stream_t s) {
    if (!n) return;
    if (count < 1 || count > 8) throw GpuError("poly_lincomb_n: 1..8 terms");
    LincombArgs a;
    a.count = count;
    for (int j = 0; j < 8; j++) { a.p[j] = j < count ? polys[j] : nullptr; a.len[j] = j < count ? lens[j] : 0; a.sc[j] = j < count ? G29::twiddle_from_std(scalars[j]) : G29::zero(); }
    hipLaunchKernelGGL(k_lincomb_n, GRID(n), 0, (hipStream_t)s, out, a, n); HIP_LAUNCH_CHECK();
}

// ---- p / (X^m - 1): residue class j: q_i = p_{i+m} + q_{i+m}; rem_j = p_j + q_j
__global__ void k_div_vanishing(F *__restrict__ q, F *__restrict__ rem, const F *__restrict__ p, size_t len, size_t m) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    size_t qlen = len - m;
    F carry = F::zero();
    // highest index i = j (mod m) with i < qlen
    if (qlen > j) {
        size_t top = j + ((qlen - 1 - j) / m) * m;
        for (size_t i = top;; i -= m) {
            carry = p[i + m] + carry;
            q[i] = carry;
            if (i < m) break;
        }
    }
    if (rem) rem[j] = (j < len ? p[j] : F::zero()) + carry;
}
// A SMALL divisor (round 1 divides by the input domain's v_X, m = 64 against |H| + 1 coefficients) makes the m chains long and few: cut every chain into S segments of
// C = 16 steps -- k_divvan_sums leaves the segment sums, k_divvan_apply starts every segment from the sum of the segments above it.  (One lane per residue class took
// 524 us of a lone 16-byte encrypt(): 64 lanes x 4,096 dependent steps.)  Additions only: the same values in the same order class by class, bit-identical.
__global__ void k_divvan_sums(F *__restrict__ part, const F *__restrict__ p, size_t qlen, size_t m, size_t C, size_t S) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * m) return;
    const size_t seg = idx / m, j = idx % m;
    F sum = F::zero();
    for (size_t t = seg * C; t < (seg + 1) * C; t++) { size_t i = j + t * m; if (i < qlen) sum = sum + p[i + m]; }
    part[idx] = sum;
}
__global__ void k_divvan_apply(F *__restrict__ q, F *__restrict__ rem, const F *__restrict__ p, const F *__restrict__ part, size_t len, size_t m, size_t C, size_t S) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * m) return;
    const size_t seg = idx / m, j = idx % m, qlen = len - m;
    F carry = F::zero();
    for (size_t u = S; u-- > seg + 1;) carry = carry + part[u * m + j];
    for (size_t t = (seg + 1) * C; t-- > seg * C;) { size_t i = j + t * m; if (i < qlen) { carry = carry + p[i + m]; q[i] = carry; } }
    if (seg == 0 && rem) rem[j] = (j < len ? p[j] : F::zero()) + carry;
}
// `scratch` (optional, `scratch_len` elements, must not overlap q / rem / p): the segment sums of the small-divisor path
void divide_by_vanishing(F *q, F *rem, const F *p, size_t len, size_t m, stream_t s, F *scratch, size_t scratch_len) {
    if (len <= m) throw GpuError("divide_by_vanishing: dividend shorter than divisor");
    const size_t chain = (len - m + m - 1) / m;            // steps of the longest residue class
    const size_t C = 16, S = (chain + C - 1) / C;          // 16 dependent steps per lane; the carry of a segment is a sum of up to S segment sums (independent loads)
    if (chain >= 64 && scratch && S * m <= scratch_len) {
        hipLaunchKernelGGL(k_divvan_sums, GRID(S * m), 0, (hipStream_t)s, scratch, p, len - m, m, C, S); HIP_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_divvan_apply, GRID(S * m), 0, (hipStream_t)s, q, rem, p, (const F *)scratch, len, m, C, S); HIP_LAUNCH_CHECK();
        return;
    }
    hipLaunchKernelGGL(k_div_vanishing, GRID(m), 0, (hipStream_t)s, q, rem, p, len, m); HIP_LAUNCH_CHECK();
}

// ---- p / (X - z):  q_i = p_{i+1} + z q_{i+1}  =  sum_{j > i} p_j z^(j - i - 1)   (synthetic division; the remainder p(z) is dropped).
// Blocked Horner, recursively: cut p into blocks of DL_B coefficients.  One lane per block runs the recurrence INSIDE its block (k_divlin_local: one product per
// coefficient) and leaves the block's value V_t = sum_{j in block} p_j z^(j - start); what a block still misses is the carry from the blocks above it,
// C_t = sum_{t' > t} V_t' (z^B)^(t' - t - 1) -- the same division applied to V at the point z^B, DL_B times shorter; k_divlin_apply then adds z^(end - 1 - i) C_t to every
// coefficient of the block (one more product).  Two products per coefficient and a few tiny launches for the upper levels, against four products per coefficient plus the
// operator applications of a generic device scan over scaled terms (rounds 1-3: ~600 M wave-instructions per proof, 2 % of everything the saturated prover issues).
// Field arithmetic is exact, so the result is bit-identical to the recurrence.  Short blocks: a lane's chain of DL_B dependent products is what a lone proof waits for
// at every level (64-coefficient blocks: 125 us per level, 0.83 ms for the two divisions of an opening; 16: four cheap levels more, a third of the wait).
constexpr int DL_B = 16;
__global__ void k_divlin_local(F *__restrict__ q, const F *__restrict__ p, size_t len, F z, F *__restrict__ v, F *__restrict__ zp) {
    if (blockIdx.x == 0 && threadIdx.x < DL_B) zp[threadIdx.x] = z.pow_u64(threadIdx.x);        // the level's power table z^0 .. z^(B-1) for k_divlin_apply (next launch but one)
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s0 = t * DL_B;
    if (s0 >= len) return;
    const size_t e = s0 + DL_B < len ? s0 + DL_B : len;
    F l = F::zero();
    for (size_t i = e; i-- > s0;) {
        if (i + 1 < len) q[i] = l;                      // (q has len - 1 entries)
        l = p[i] + z * l;
    }
    v[t] = l;
}
// one workgroup: the WHOLE division of a short sequence (len <= DL_TOP = 256 blocks of DL_B), the top of the recursion.  Lane t runs the recurrence inside block t, the 256
// block values are combined by a suffix scan in LDS -- S_t = sum_{t' >= t} V_t' Y^(t' - t), Y = z^DL_B: eight doubling steps S_t += Y^(2^k) S_(t + 2^k) -- and every lane
// adds its carry C_t = S_(t+1) times z^(block end - 1 - i) to its own coefficients.  Round 5 recursed down to 16 coefficients: three more levels of k_divlin_local /
// k_divlin_apply launches and a single-lane base case per division, ~40 tiny launches in front of the openings of a lone proof.  Exact field
// arithmetic: bit-identical to the recurrence.
constexpr int DL_TOP_THREADS = 256;
constexpr size_t DL_TOP = (size_t)DL_TOP_THREADS * DL_B;
__global__ void __launch_bounds__(DL_TOP_THREADS) k_divlin_top(F *__restrict__ q, const F *__restrict__ p, size_t len, F z) {
    __shared__ F sh[DL_TOP_THREADS];
    const uint32_t t = threadIdx.x;
    const size_t s0 = (size_t)t * DL_B, e = s0 + DL_B < len ? s0 + DL_B : len;
    F l = F::zero();
    if (s0 < len) {
        for (size_t i = e; i-- > s0;) {
            if (i + 1 < len) q[i] = l;                  // (q has len - 1 entries)
            l = p[i] + z * l;
        }
    }
    sh[t] = l;                                          // V_t (zero for blocks beyond the sequence)
    __syncthreads();
    F y = z.pow_u64(DL_B);
    for (int k = 1; k < DL_TOP_THREADS; k <<= 1) {
        const F other = t + k < DL_TOP_THREADS ? sh[t + k] : F::zero();
        __syncthreads();
        sh[t] = sh[t] + y * other;
        __syncthreads();
        y = y.sqr();
    }
    if (s0 >= len || e >= len) return;                  // the top block misses nothing
    const F c = sh[t + 1];                              // (t + 1 < DL_TOP_THREADS: a block below the top one)
    F w = F::one();
    for (size_t i = e; i-- > s0;) { q[i] = q[i] + w * c; w = w * z; }
}
// q_i += z^(end of i's block - 1 - i) * carry[block of i]     (carry has nblocks - 1 entries: the top block misses nothing)
__global__ void __launch_bounds__(256) k_divlin_apply(F *__restrict__ q, size_t qlen, const F *__restrict__ zp_table, const F *__restrict__ carry, size_t nblocks) {
    __shared__ F zp[DL_B];
    if (threadIdx.x < DL_B) zp[threadIdx.x] = zp_table[threadIdx.x];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qlen) return;
    const size_t t = i / DL_B;
    if (t + 1 >= nblocks) return;
    const size_t e = (t + 1) * DL_B;                    // (every block but the top one is full)
    q[i] = q[i] + zp[e - 1 - i] * carry[t];
}
size_t divide_by_linear_scratch(size_t len) {       // in field elements: block values and carries of every level
    size_t total = 8;
    for (size_t n = len; n > DL_B; ) { n = (n + DL_B - 1) / DL_B; total += 2 * n + 2 + DL_B; }
    return total;
}
static void divlin_rec(F *q, const F *p, size_t len, const F &z, F *scratch, hipStream_t s) {
    if (len < 2) return;
    if (len <= DL_TOP) { hipLaunchKernelGGL(k_divlin_top, dim3(1), dim3(DL_TOP_THREADS), 0, s, q, p, len, z); HIP_LAUNCH_CHECK(); return; }
    const size_t nblocks = (len + DL_B - 1) / DL_B;
    F *v = scratch, *carry = scratch + nblocks + 1, *zp = carry + nblocks + 1;
    hipLaunchKernelGGL(k_divlin_local, GRID(nblocks), 0, s, q, p, len, z, v, zp); HIP_LAUNCH_CHECK();
    divlin_rec(carry, v, nblocks, z.pow_u64(DL_B), zp + DL_B, s);
    hipLaunchKernelGGL(k_divlin_apply, GRID(len - 1), 0, s, q, len - 1, (const F *)zp, (const F *)carry, nblocks); HIP_LAUNCH_CHECK();
}
void divide_by_linear(F *q, const F *p, size_t len, const F &z, F *scratch, size_t scratch_elems, stream_t s_) {
    hipStream_t s = (hipStream_t)s_;
    if (len < 2) return;
    if (len >= ((size_t)1 << 31)) throw GpuError("divide_by_linear: polynomial too long");
    if (z.is_zero()) { HIP_CHECK(hipMemcpyAsync(q, p + 1, (len - 1) * sizeof(F), hipMemcpyDeviceToDevice, s)); return; }
    if (divide_by_linear_scratch(len) > scratch_elems) throw GpuError("divide_by_linear: scratch too small");
    divlin_rec(q, p, len, z, scratch, s);
}

// ---- evaluation: chunks of EV_CHUNK coefficients by Horner, level by level (the partials of one level are the coefficients of the next, in x^EV_CHUNK) until one
// workgroup can combine what is left: sum_t partial_t * y^t.  Short chunks: a lane's chain of dependent products is what a lone proof waits for (64-coefficient chunks:
// 100 us per level at |H| = 2^18 on 16 workgroups); the extra partial traffic is 1/16 of reading the polynomial.
constexpr size_t EV_CHUNK = 16, EV_COMBINE_MAX = 1024;
__global__ void k_eval_chunks(const F *__restrict__ p, size_t len, F x, F *__restrict__ partial) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t nch = (len + EV_CHUNK - 1) / EV_CHUNK;
    if (t >= nch) return;
    size_t s0 = t * EV_CHUNK, e = s0 + EV_CHUNK < len ? s0 + EV_CHUNK : len;
    F acc = F::zero();
    for (size_t i = e; i-- > s0;) acc = acc * x + p[i];
    partial[t] = acc;
}
__global__ void __launch_bounds__(256) k_eval_combine(const F *__restrict__ partial, size_t nch, F y, F *__restrict__ out) {
    __shared__ F sh[256];
    uint32_t t = threadIdx.x;
    F y256 = y.pow_u64(256), pw = y.pow_u64(t), acc = F::zero();
    for (size_t i = t; i < nch; i += 256) { acc = acc + partial[i] * pw; pw = pw * y256; }
    sh[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)t < s) sh[t] = sh[t] + sh[t + s]; __syncthreads(); }
    if (t == 0) out[0] = sh[0];
}
size_t poly_eval_scratch(size_t len) {
    size_t tot = 0, n = len;
    do { n = (n + EV_CHUNK - 1) / EV_CHUNK; tot += n; } while (n > EV_COMBINE_MAX);
    return tot;
}
// launches only: the value lands in *out (device)
static void eval_launch(const F *p, size_t len, F x, F *scratch, F *out, hipStream_t s) {
    const F *src = p;
    size_t n = len;
    F y = x;
    do {
        size_t nch = (n + EV_CHUNK - 1) / EV_CHUNK;
        hipLaunchKernelGGL(k_eval_chunks, GRID(nch), 0, s, src, n, y, scratch); HIP_LAUNCH_CHECK();
        src = scratch; scratch += nch; n = nch; y = y.pow_u64(EV_CHUNK);
    } while (n > EV_COMBINE_MAX);
    hipLaunchKernelGGL(k_eval_combine, dim3(1), dim3(256), 0, s, src, n, y, out); HIP_LAUNCH_CHECK();
}
// out[i] = p[i](x[i]) for up to 8 polynomials: every launch first, then ONE wait and one copy (the four evaluations of a proof go into the transcript together)
void poly_eval_multi(const F *const *p, const size_t *len, const F *x, int count, F *out, F *scratch, size_t scratch_elems, stream_t s_) {
    hipStream_t s = (hipStream_t)s_;
    if (count < 1 || count > 8) throw GpuError("poly_eval_multi: 1..8 polynomials");
    size_t need = 8;
    for (int i = 0; i < count; i++) need += poly_eval_scratch(len[i]);
    if (need > scratch_elems) throw GpuError("poly_eval_multi: scratch too small");
    F *res = scratch, *at = scratch + 8;
    bool any = false;
    for (int i = 0; i < count; i++) {
        if (!len[i]) continue;
        eval_launch(p[i], len[i], x[i], at, res + i, s);
        at += poly_eval_scratch(len[i]); any = true;
    }
    F host[8];
    if (any) {
        sync((stream_t)s);        // drain first (sleeps in throughput mode): the pageable device-to-host copy below waits actively inside HIP
        HIP_CHECK(hipMemcpyAsync(host, res, count * sizeof(F), hipMemcpyDeviceToHost, s));
        sync((stream_t)s);
    }
    for (int i = 0; i < count; i++) out[i] = len[i] ? host[i] : F::zero();
}
// (`scratch`: 8 + poly_eval_scratch(len) elements)
F poly_eval(const F *p, size_t len, const F &x, F *scratch, stream_t s_) {
    F out;
    poly_eval_multi(&p, &len, &x, 1, &out, scratch, 8 + poly_eval_scratch(len), s_);
    return out;
}

// ---- batch inversion (Montgomery's trick) with ONE inversion per 512-lane workgroup: BI_CHUNK elements per lane, the 512 lane products are
// combined by prefix / suffix product scans in LDS, lane 0 inverts the block product, every lane recovers the inverse of its own product as
// inv_total * prefix(lane-1) * suffix(lane+1).  ~4 field products per element instead of 27 (one 380-product Fermat chain per 16 elements);
// the chain left over is latency, not ALU throughput, so it hides beside other proofs' kernels.  Exact arithmetic: results are identical.
// BI_CHUNK: 16 elements per lane when proofs overlap (the scans cost 18 / BI_CHUNK products per element); 4 for a lone call on a small vector, where the lane's chain
// of dependent products (3 BI_CHUNK + 18 + the inversion) is the kernel's duration and 16-element chunks leave most of the chip without a workgroup.
constexpr int BI_BLOCK = 512;
template <int BI_CHUNK>
__global__ void __launch_bounds__(BI_BLOCK) k_batch_inverse(F *__restrict__ v, size_t n, bool has_post, F post) {
    __shared__ F pre_s[BI_BLOCK], suf_s[BI_BLOCK];
    __shared__ F inv_total;
    const int lane = threadIdx.x;
    size_t t = (size_t)blockIdx.x * BI_BLOCK + lane;
    size_t s0 = t * BI_CHUNK;
    if (s0 > n) s0 = n;
    size_t e = s0 + BI_CHUNK < n ? s0 + BI_CHUNK : n;
    F pre[BI_CHUNK];
    F acc = F::one();
    for (size_t i = s0; i < e; i++) { pre[i - s0] = acc; F x = v[i]; if (!x.is_zero()) acc = acc * x; }
    pre_s[lane] = acc; suf_s[lane] = acc;
    __syncthreads();
    for (int d = 1; d < BI_BLOCK; d <<= 1) {          // inclusive scans: pre_s[l] = prod_{u <= l}, suf_s[l] = prod_{u >= l}
        F a, b;
        const bool ha = lane >= d, hb = lane + d < BI_BLOCK;
        if (ha) a = pre_s[lane - d];
        if (hb) b = suf_s[lane + d];
        __syncthreads();
        if (ha) pre_s[lane] = pre_s[lane] * a;
        if (hb) suf_s[lane] = suf_s[lane] * b;
        __syncthreads();
    }
    // the lone-call variant takes the Euclidean inverse (a third of the Fermat chain's latency); between overlapping proofs the cost of the step is its instruction
    // count, and there the branchy shift-and-subtract loop issues ~1.5 x what the 380 products do (SQ_INSTS_VALU, profiles/r04_valu_by_kernel.md): Fermat
    if (lane == 0) { F iv = BI_CHUNK == 4 ? suf_s[0].inverse() : suf_s[0].inverse_fermat(); inv_total = has_post ? iv * post : iv; }
    __syncthreads();
    acc = inv_total;
    if (lane > 0) acc = acc * pre_s[lane - 1];
    if (lane + 1 < BI_BLOCK) acc = acc * suf_s[lane + 1];
    for (size_t i = e; i-- > s0;) {
        F x = v[i];
        if (x.is_zero()) continue;
        v[i] = acc * pre[i - s0];
        acc = acc * x;
    }
}
// r(a, X) = (a^n - X^n) / (a - X) -- Marlin's u_H(a, X) -- on cosets g H of the size-n domain H, with NO inversion and no transform: over a field
//      (a^n - y^n) / (a - y)  =  prod_{k < lg n} (a^(2^k) + y^(2^k)),
// and the factors are shared along a binary tree: T_k[j] = prod_{k' >= k} (a^(2^k') + Y_k'[j]) with Y_k[j] = (g h_j)^(2^k) = g^(2^k) elems[j 2^k] only depends on
// j mod n / 2^k, so T_k[j] = (a^(2^k) + Y_k[j]) T_(k+1)[j mod n / 2^(k+1)], T_(lg n) = 1, and T_0[i] = r(a, g h_i): two products per tree node, 2n nodes, depth lg n, every
// node of a level independent.  Replaces (round 5) the batch inversions of the prover's rounds 2 and 3 -- v_H(alpha) / (alpha - h) on H and on round 2's two cosets, and
// 1 / ((beta - row)(alpha - col)) on K, which is a product of two such tables' entries -- and with them a ~300 us single-lane inversion chain per call of a lone proof.
// k_vq_top runs levels lg n .. lg n - 10 of every coset in one workgroup each (<= 1024 nodes, LDS); k_vq_expand takes one node of level k_hi per workgroup down to level
// k_lo >= k_hi - 10 (its descendants are the indices j + m (n >> k_hi)).  tables (device, uploaded per call): a^(2^k) then, per coset, g^(2^k), k < lg n.
constexpr int VQ_MAX_COSETS = 3, VQ_STEP = 10;
struct VqArgs { F *out[VQ_MAX_COSETS]; const F *tab; const F *elems; uint32_t n; int lg_n; bool unit_g[VQ_MAX_COSETS]; };
__device__ __forceinline__ F vq_factor(const VqArgs &A, int c, int k, uint32_t j) {       // a^(2^k) + (g_c h_j)^(2^k)
    const F e = A.elems[(size_t)j << k];
    return A.tab[k] + (A.unit_g[c] ? e : e * A.tab[(size_t)(c + 1) * A.lg_n + k]);
}
__global__ void __launch_bounds__(1024) k_vq_top(VqArgs A, F *__restrict__ mid) {
    __shared__ F buf[2][1 << VQ_STEP];
    const int c = blockIdx.x, L = A.lg_n, k_stop = L > VQ_STEP ? L - VQ_STEP : 0;
    const uint32_t t = threadIdx.x;
    if (t == 0) buf[0][0] = F::one();
    __syncthreads();
    int cur = 0;
    for (int k = L - 1; k >= k_stop; k--) {
        const uint32_t s = A.n >> k;                       // nodes of level k
        if (t < s) buf[cur ^ 1][t] = vq_factor(A, c, k, t) * buf[cur][t & ((s >> 1) - 1)];
        __syncthreads();
        cur ^= 1;
    }
    const uint32_t s = A.n >> k_stop;
    F *dst = k_stop == 0 ? A.out[c] : mid + (size_t)c * (1u << VQ_STEP);
    if (t < s) dst[t] = buf[cur][t];
}
// src: level k_hi of every coset (n >> k_hi nodes each, coset-major); dst: level k_lo likewise, or the cosets' out arrays when k_lo == 0
__global__ void __launch_bounds__(256) k_vq_expand(VqArgs A, const F *__restrict__ src, F *__restrict__ dst, int k_hi, int k_lo) {
    __shared__ F buf[2][1 << VQ_STEP];
    const uint32_t s_hi = A.n >> k_hi;
    const int c = blockIdx.x / s_hi;
    const uint32_t j = blockIdx.x % s_hi, t = threadIdx.x;
    if (t == 0) buf[0][0] = src[(size_t)c * s_hi + j];
    __syncthreads();
    int cur = 0;
    for (int k = k_hi - 1; k >= k_lo; k--) {
        const uint32_t cnt = 1u << (k_hi - k);             // this node's descendants at level k: local m <-> global j + m s_hi; parent: local m mod cnt / 2
        for (uint32_t m = t; m < cnt; m += 256) buf[cur ^ 1][m] = vq_factor(A, c, k, j + m * s_hi) * buf[cur][m & ((cnt >> 1) - 1)];
        __syncthreads();
        cur ^= 1;
    }
    const uint32_t cnt = 1u << (k_hi - k_lo);
    F *o = k_lo == 0 ? A.out[c] : dst + (size_t)c * (A.n >> k_lo);
    for (uint32_t m = t; m < cnt; m += 256) o[j + (size_t)m * s_hi] = buf[cur][m];
}
// the level an expansion from level k_hi stops at: the SHORT step comes first, so that the last launch -- the one that writes n values per coset -- always runs full
// VQ_STEP-level workgroups (lg n = 22 used to end on a two-level step: 3 * 2^20 workgroups of 256 lanes for four values each, 11.6 ms per call)
static inline int vq_next_level(int k_hi) { const int r = k_hi % VQ_STEP; return k_hi - (r ? r : VQ_STEP); }
size_t vanishing_quotient_scratch(int lg_n, int ncosets) {          // field elements of `scratch` vanishing_quotient_evals needs
    size_t tab = (size_t)(ncosets + 1) * (lg_n > 0 ? lg_n : 1), mids = 0;
    for (int k = lg_n - VQ_STEP; k > 0; k = vq_next_level(k)) mids += (size_t)ncosets << (lg_n - k);
    return tab + mids + 8;
}
void vanishing_quotient_evals(F *const *out, const F *g, int ncosets, const F &a, const F *elems, uint32_t n, int lg_n, F *scratch, size_t scratch_elems, stream_t s_) {
    hipStream_t s = (hipStream_t)s_;
    if (ncosets < 1 || ncosets > VQ_MAX_COSETS) throw GpuError("vanishing_quotient_evals: 1..3 cosets");
    if (n == 0 || n != (1u << lg_n)) throw GpuError("vanishing_quotient_evals: n must be 2^lg_n");
    if (vanishing_quotient_scratch(lg_n, ncosets) > scratch_elems) throw GpuError("vanishing_quotient_evals: scratch too small");
    if (lg_n == 0) { for (int c = 0; c < ncosets; c++) { F one = F::one(); h2d(out[c], &one, sizeof(F), s_); } return; }
    // host: the power chains a^(2^k), g_c^(2^k)
    std::vector<F> tab((size_t)(ncosets + 1) * lg_n);
    F v = a;
    for (int k = 0; k < lg_n; k++) { tab[k] = v; v = v.sqr(); }
    VqArgs A;
    for (int c = 0; c < VQ_MAX_COSETS; c++) { A.out[c] = c < ncosets ? out[c] : nullptr; A.unit_g[c] = true; }
    for (int c = 0; c < ncosets; c++) {
        A.unit_g[c] = g[c] == F::one();
        v = g[c];
        for (int k = 0; k < lg_n; k++) { tab[(size_t)(c + 1) * lg_n + k] = v; v = v.sqr(); }
    }
    h2d(scratch, tab.data(), tab.size() * sizeof(F), s_);
    A.tab = scratch; A.elems = elems; A.n = n; A.lg_n = lg_n;
    F *mid = scratch + tab.size();
    hipLaunchKernelGGL(k_vq_top, dim3(ncosets), dim3(1024), 0, s, A, mid); HIP_LAUNCH_CHECK();
    for (int k_hi = lg_n - VQ_STEP; k_hi > 0; ) {
        const int k_lo = vq_next_level(k_hi);
        F *dst = mid + ((size_t)ncosets << (lg_n - k_hi));
        hipLaunchKernelGGL(k_vq_expand, dim3((unsigned)((size_t)ncosets * (n >> k_hi))), dim3(256), 0, s, A, (const F *)mid, dst, k_hi, k_lo); HIP_LAUNCH_CHECK();
        mid = dst; k_hi = k_lo;
    }
}
// round 3 of Marlin: f(kappa) = v_H(alpha) v_H(beta) (eta_a val_a + eta_b val_b + eta_c val_c)(kappa) / ((beta - row(kappa)) (alpha - col(kappa))) on K, where row(kappa) and col(kappa)
// are ELEMENTS OF H (the indexer's "row" holds elems[ci], "col" elems[ri]): with ra[i] = v_H(alpha) / (alpha - h_i) and rb[i] = v_H(beta) / (beta - h_i) (vanishing_quotient_evals)
// the quotient is rb[ci[kappa]] ra[ri[kappa]] -- two gathers and two products instead of a batch inversion over K.  Entries past the index's non-zeros carry (0, 0) and zero values.
__global__ void k_f_from_tables(F *__restrict__ out, const F *__restrict__ va, const F *__restrict__ vb, const F *__restrict__ vc, G29 ea, G29 eb, G29 ec, const F *__restrict__ ra,
                                const F *__restrict__ rb, const uint32_t *__restrict__ ri, const uint32_t *__restrict__ ci, size_t k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const G29 v[3] = {ld29(va[i]), ld29(vb[i]), ld29(vc[i])}, w[3] = {ea, eb, ec};       // (scalars in twiddle form: one Montgomery reduction for the three terms)
    const F num = st29(G29::dot<3>(v, w));
    out[i] = num.is_zero() ? num : num * (rb[ci[i]] * ra[ri[i]]);
}
void f_evals_from_tables(F *out, const F *va, const F *vb, const F *vc, const F &ea, const F &eb, const F &ec, const F *ra, const F *rb, const uint32_t *ri, const uint32_t *ci, size_t k,
                         stream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(k_f_from_tables, GRID(k), 0, (hipStream_t)s, out, va, vb, vc, G29::twiddle_from_std(ea), G29::twiddle_from_std(eb), G29::twiddle_from_std(ec), ra, rb, ri, ci, k);
    HIP_LAUNCH_CHECK();
}
void batch_inverse(F *v, size_t n, const F *post, stream_t s) {
    if (!n) return;
    const F post_v = post ? *post : F::one();
    if (throughput_mode() || n > ((size_t)1 << 21)) {
        size_t threads = (n + 15) / 16;
        hipLaunchKernelGGL(k_batch_inverse<16>, dim3((unsigned)((threads + BI_BLOCK - 1) / BI_BLOCK)), dim3(BI_BLOCK), 0, (hipStream_t)s, v, n, post != nullptr, post_v); HIP_LAUNCH_CHECK();
    } else {
        size_t threads = (n + 3) / 4;
        hipLaunchKernelGGL(k_batch_inverse<4>, dim3((unsigned)((threads + BI_BLOCK - 1) / BI_BLOCK)), dim3(BI_BLOCK), 0, (hipStream_t)s, v, n, post != nullptr, post_v); HIP_LAUNCH_CHECK();
    }
}

// ---- round 2 / 3 pointwise kernels
__global__ void k_q1_coset(F *__restrict__ out, const F *__restrict__ r, const F *__restrict__ za, const F *__restrict__ zb, const F *__restrict__ t, const F *__restrict__ z,
                           F ca, F cb, F cz, G29 ea, G29 eb, G29 ec, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // (all in the standard form x R; a product of two such values goes through the five-bit shift: mul(shl5(x R), y R) = x y R)
    const G29 a = ld29(za[i]) + ld29(ca), b = ld29(zb[i]) + ld29(cb), zz = ld29(z[i]) + ld29(cz);        // < 2 p each
    const G29 v[3] = {a.shl5() * b, a, b}, w[3] = {ec, ea, eb};                                             // a b < 1.5 p
    const G29 sum = G29::dot<3>(v, w);                                                                      // eta_c a b + eta_a a + eta_b b     < 1.1 p
    const G29 l[2] = {ld29(r[i]).shl5(), ld29(t[i]).shl5()}, m[2] = {sum, G29::zero().template sub<2>(zz)};  // r sum + t (2 p - z)
    out[i] = st29(G29::dot<2>(l, m));
}
void q1_coset_pointwise(F *out, const F *r, const F *za, const F *zb, const F *t, const F *z, const F &ca, const F &cb, const F &cz, const F &ea, const F &eb, const F &ec, size_t n, stream_t s) {
    hipLaunchKernelGGL(k_q1_coset, GRID(n), 0, (hipStream_t)s, out, r, za, zb, t, z, ca, cb, cz, G29::twiddle_from_std(ea), G29::twiddle_from_std(eb), G29::twiddle_from_std(ec), n); HIP_LAUNCH_CHECK();
}
// the assignment on H: instance values at the multiples of |H| / |X|, witness values in between (the index map of k_w_evals)
__global__ void k_z_evals_h(F *__restrict__ out, const uint8_t *__restrict__ z, uint32_t n, uint32_t m, uint32_t num_witness) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t ratio = n / m;
    uint8_t bit;
    if (k % ratio == 0) bit = z[k / ratio];
    else { const uint32_t wi = k - k / ratio - 1; bit = wi < num_witness ? z[m + wi] : 0; }
    out[k] = bit ? F::one() : F::zero();
}
void z_evals_h(F *out, const uint8_t *z, uint32_t n, uint32_t m, uint32_t num_witness, stream_t s) {
    hipLaunchKernelGGL(k_z_evals_h, GRID(n), 0, (hipStream_t)s, out, z, n, m, num_witness); HIP_LAUNCH_CHECK();
}
// q = q_1 - mask = q_lo + X^n q_mid + X^2n q_hi (each of degree < n) has the interpolants Q0 = q_lo + q_mid + q_hi on H, Q1 = q_lo + zeta q_mid - q_hi on W H and
// Q3 = q_lo - zeta q_mid - q_hi on W^3 H (zeta = W^n, zeta^2 = -1).  Dividing by v_H = X^n - 1 in coefficient space: quotient (q_mid + q_hi) + X^n q_hi, remainder Q0;
// the mask m_lo + X^n m_mid + X^2n m_hi divides the same way.  h_1 = the quotient (2n coefficients), x g_1 = the remainder (its constant term vanishes: sumcheck).
__global__ void k_q1_combine(F *__restrict__ h1, F *__restrict__ g1, const F *__restrict__ q0, const F *__restrict__ q1, const F *__restrict__ q3, const F *__restrict__ mask,
                             F inv2, F inv2zeta, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F a = q1[i], b = q3[i], c = q0[i];
    const F q_mid = (a - b) * inv2zeta;
    const F q_hi = ((c - q_mid) - (a + b) * inv2) * inv2;
    const F m_lo = mask[i], m_mid = mask[n + i], m_hi = mask[2 * n + i];
    h1[i] = q_mid + q_hi + m_mid + m_hi;
    h1[n + i] = q_hi + m_hi;
    if (i >= 1) g1[i - 1] = c + m_lo + m_mid + m_hi;
}
void q1_combine(F *h1, F *g1, const F *q0, const F *q1, const F *q3, const F *mask, const F &inv2, const F &inv2zeta, size_t n, stream_t s) {
    hipLaunchKernelGGL(k_q1_combine, GRID(n), 0, (hipStream_t)s, h1, g1, q0, q1, q3, mask, inv2, inv2zeta, n); HIP_LAUNCH_CHECK();
}
// ---- coset tools for round 3: h_2 = (a - b f) / v_K has degree <= |K| - 2, so its |K| values on ONE coset g K determine it and there
// v_K(g w^i) = g^|K| - 1 is a non-zero constant: no 2|K|-point transforms, no division by the vanishing polynomial.
// out[j] = in[j] * g^j  (coefficients of p(g X)); 16 consecutive coefficients per lane, one pow per lane
__global__ void k_coset_scale(F *__restrict__ out, const F *__restrict__ in, F g, size_t in_len, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t j0 = t * 16;
    if (j0 >= n) return;
    size_t e = j0 + 16 < n ? j0 + 16 : n;
    F pw = g.pow_u64(j0);
    for (size_t j = j0; j < e; j++) { out[j] = j < in_len ? in[j] * pw : F::zero(); pw = pw * g; }
}
void coset_scale(F *out, const F *in, const F &g, size_t in_len, size_t n, stream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_coset_scale, GRID((n + 15) / 16), 0, (hipStream_t)s, out, in, g, in_len, n); HIP_LAUNCH_CHECK();
}
// values of h_2 on the coset from the coset values of the six index polynomials and of f:
//   a = ea va + eb vb + ec vc,  b = alpha beta - alpha row - beta col + row_col,  out = (a - b f) * vinv
__global__ void k_h2_coset(F *__restrict__ out, const F *__restrict__ row, const F *__restrict__ col, const F *__restrict__ va, const F *__restrict__ vb,
                           const F *__restrict__ vc, const F *__restrict__ rc, const F *__restrict__ f, G29 alpha, G29 beta, F alpha_beta, G29 ea, G29 eb, G29 ec, G29 vinv, size_t k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const G29 v3[3] = {ld29(va[i]), ld29(vb[i]), ld29(vc[i])}, w3[3] = {ea, eb, ec};
    const G29 a = G29::dot<3>(v3, w3);                                                     // < 1.02 p
    const G29 v2[2] = {ld29(row[i]), ld29(col[i])}, w2[2] = {alpha, beta};
    const G29 b = (ld29(alpha_beta) + ld29(rc[i])).template sub<2>(G29::dot<2>(v2, w2));    // alpha beta + row_col - (alpha row + beta col) + 2 p   < 4 p
    const G29 bf = b.shl5() * ld29(f[i]);                                                   // < 1.5 p
    out[i] = st29(a.template sub<2>(bf) * vinv);
}
void h2_coset(F *out, const F *row, const F *col, const F *va, const F *vb, const F *vc, const F *rc, const F *f, const F &alpha, const F &beta, const F &alpha_beta,
              const F &ea, const F &eb, const F &ec, const F &vinv, size_t k, stream_t s) {
    hipLaunchKernelGGL(k_h2_coset, GRID(k), 0, (hipStream_t)s, out, row, col, va, vb, vc, rc, f, G29::twiddle_from_std(alpha), G29::twiddle_from_std(beta), alpha_beta,
                       G29::twiddle_from_std(ea), G29::twiddle_from_std(eb), G29::twiddle_from_std(ec), G29::twiddle_from_std(vinv), k); HIP_LAUNCH_CHECK();
}
// z_poly = w * (X^m - 1) + x_poly : zp[i] = (i >= m ? w[i-m] : 0) - (i < wlen ? w[i] : 0) + (i < m ? x[i] : 0), i <= n
__global__ void k_z_poly(F *__restrict__ zp, const F *__restrict__ w, size_t wlen, const F *__restrict__ x, uint32_t m, size_t n1) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    F v = F::zero();
    if (i >= m && i - m < wlen) v = w[i - m];
    if (i < wlen) v = v - w[i];
    if (i < m) v = v + x[i];
    zp[i] = v;
}
void z_poly_from_w(F *zp, const F *w, size_t wlen, const F *x_poly, uint32_t m, size_t n, stream_t s) { hipLaunchKernelGGL(k_z_poly, GRID(n + 1), 0, (hipStream_t)s, zp, w, wlen, x_poly, m, n + 1); HIP_LAUNCH_CHECK(); }

// ---- prover randomness on the device: the ChaCha key stream -> rejection-sampled field elements (ark-ff UniformRand), in parallel
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
__global__ void k_chacha_blocks(uint32_t *__restrict__ words, uint64_t first_block, uint32_t nblocks, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t k4, uint32_t k5,
                                uint32_t k6, uint32_t k7, int rounds) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    uint64_t ctr = first_block + b;
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, k0, k1, k2, k3, k4, k5, k6, k7, (uint32_t)ctr, (uint32_t)(ctr >> 32), 0, 0};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define ZK_QR(a, b, c, d) \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    for (int r = 0; r < rounds; r += 2) {
        ZK_QR(0, 4, 8, 12) ZK_QR(1, 5, 9, 13) ZK_QR(2, 6, 10, 14) ZK_QR(3, 7, 11, 15)
        ZK_QR(0, 5, 10, 15) ZK_QR(1, 6, 11, 12) ZK_QR(2, 7, 8, 13) ZK_QR(3, 4, 9, 14)
    }
#undef ZK_QR
#pragma unroll
    for (int i = 0; i < 16; i++) words[(size_t)b * 16 + i] = x[i] + s[i];
}
__global__ void k_rand_flags(const uint32_t *__restrict__ words, uint32_t word_off, uint32_t ncand, uint32_t *__restrict__ flags) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncand) return;
    uint32_t l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) l[i] = words[(size_t)word_off + 8 * (size_t)j + i];
    l[7] &= 0xffffffffu >> (256 - F::BITS);
    flags[j] = F::geq_mod(l) ? 0u : 1u;
}
__global__ void k_rand_compact(const uint32_t *__restrict__ words, uint32_t word_off, uint32_t ncand, const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos,
                               F *__restrict__ out, uint32_t count, uint32_t *__restrict__ last_cand) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncand || !flags[j]) return;
    uint32_t q = pos[j];
    if (q >= count) return;
    F v;
#pragma unroll
    for (int i = 0; i < 8; i++) v.l[i] = words[(size_t)word_off + 8 * (size_t)j + i];
    v.l[7] &= 0xffffffffu >> (256 - F::BITS);
    out[q] = v;
    if (q + 1 == count) *last_cand = j;
}
uint64_t chacha_field_stream(F *out, size_t count, const uint32_t key[8], int rounds, uint64_t word_pos, void *scratch, size_t scratch_bytes, stream_t s_) {
    hipStream_t s = (hipStream_t)s_;
    if (count == 0) return word_pos;
    // acceptance probability p / 2^BITS; oversample, and loop in the (practically impossible) case the batch falls short
    const double accept = 0.58;   // BLS12-377 Fr: 0x12ab.. / 0x2000..
    uint64_t pos_words = word_pos;
    size_t done = 0;
    while (done < count) {
        size_t want = count - done;
        uint32_t ncand = (uint32_t)((double)want / accept * 1.02) + 4096;
        uint64_t first_block = pos_words / 16;
        uint32_t word_off = (uint32_t)(pos_words % 16);
        uint32_t nblocks = (uint32_t)(((uint64_t)word_off + 8ull * ncand + 15) / 16);
        size_t need = (size_t)nblocks * 64 + (size_t)ncand * 8 + 64 + (1 << 20);
        if (need > scratch_bytes) throw GpuError("chacha_field_stream: scratch too small");
        uint32_t *words = (uint32_t *)scratch, *flags = words + (size_t)nblocks * 16, *pos = flags + ncand, *last = pos + ncand;
        void *tmp = (void *)(last + 16);
        size_t tmp_bytes = scratch_bytes - ((char *)tmp - (char *)scratch);
        hipLaunchKernelGGL(k_chacha_blocks, GRID(nblocks), 0, s, words, first_block, nblocks, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7], rounds); HIP_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_rand_flags, GRID(ncand), 0, s, (const uint32_t *)words, word_off, ncand, flags); HIP_LAUNCH_CHECK();
        size_t tb = 0;
        HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, flags, pos, 0u, (size_t)ncand, rocprim::plus<uint32_t>(), s));
        if (tb > tmp_bytes) throw GpuError("chacha_field_stream: scan scratch too small");
        HIP_CHECK(rocprim::exclusive_scan(tmp, tb, flags, pos, 0u, (size_t)ncand, rocprim::plus<uint32_t>(), s));
        HIP_CHECK(hipMemsetAsync(last, 0xff, 4, s));
        hipLaunchKernelGGL(k_rand_compact, GRID(ncand), 0, s, (const uint32_t *)words, word_off, ncand, (const uint32_t *)flags, (const uint32_t *)pos, out + done, (uint32_t)want, last); HIP_LAUNCH_CHECK();
        uint32_t h_last = 0, h_tail[2] = {0, 0};
        sync((stream_t)s);        // drain first (sleeps in throughput mode): the pageable device-to-host copies below wait actively inside HIP
        HIP_CHECK(hipMemcpyAsync(&h_last, last, 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(&h_tail[0], pos + ncand - 1, 4, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipMemcpyAsync(&h_tail[1], flags + ncand - 1, 4, hipMemcpyDeviceToHost, s));
        sync((stream_t)s);
        if (h_last != 0xffffffffu) { pos_words += 8ull * ((uint64_t)h_last + 1); done = count; }
        else { done += (size_t)h_tail[0] + h_tail[1]; pos_words += 8ull * ncand; }   // every accepted candidate of the batch was used
    }
    return pos_words;
}
__global__ void k_mask_fixup(F *p, size_t n) { if (blockIdx.x == 0 && threadIdx.x == 0) p[0] = p[0] - (p[0] + p[n] + p[2 * n]); }
void mask_fixup(F *p, size_t n, stream_t s) { hipLaunchKernelGGL(k_mask_fixup, dim3(1), dim3(64), 0, (hipStream_t)s, p, n); HIP_LAUNCH_CHECK(); }

// ---- indexer: evaluations of the joint-matrix arithmetization on K (ark-marlin arithmetize_matrix)
__global__ void k_index_rowcol(F *__restrict__ row, F *__restrict__ col, F *__restrict__ rowcol, F *__restrict__ u, const uint32_t *__restrict__ ci, const uint32_t *__restrict__ ri,
                               size_t cnt, size_t k, const F *__restrict__ elems, uint32_t n, F n_fe) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    if (i < cnt) {
        uint32_t c = ci[i], r = ri[i];
        F rv = elems[c], cv = elems[r];          // transposed: "row" holds the column's domain element
        row[i] = rv; col[i] = cv; rowcol[i] = rv * cv;
        u[i] = elems[(n - c) & (n - 1)] * n_fe;   // u_H(x, x) = |H| x^(|H|-1)
    } else {
        row[i] = elems[0]; col[i] = elems[0]; rowcol[i] = elems[0]; u[i] = F::zero();
    }
}
__global__ void k_index_vals(F *__restrict__ va, F *__restrict__ vb, F *__restrict__ vc, const F *__restrict__ uinv, const int64_t *__restrict__ ca, const int64_t *__restrict__ cb,
                             const int64_t *__restrict__ cc, size_t cnt, size_t k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    if (i < cnt) {
        F w = uinv[i];
        va[i] = ca[i] ? F::from_i64(ca[i]) * w : F::zero();
        vb[i] = cb[i] ? F::from_i64(cb[i]) * w : F::zero();
        vc[i] = cc[i] ? F::from_i64(cc[i]) * w : F::zero();
    } else { va[i] = F::zero(); vb[i] = F::zero(); vc[i] = F::zero(); }
}
void index_evals(F *row, F *col, F *rowcol, F *va, F *vb, F *vc, F *tmp, const uint32_t *ci, const uint32_t *ri, const int64_t *ca, const int64_t *cb, const int64_t *cc, size_t cnt,
                 size_t k, const F *elems, uint32_t n, stream_t s) {
    hipLaunchKernelGGL(k_index_rowcol, GRID(k), 0, (hipStream_t)s, row, col, rowcol, tmp, ci, ri, cnt, k, elems, n, F::from_u64(n)); HIP_LAUNCH_CHECK();
    batch_inverse(tmp, k, nullptr, s);
    hipLaunchKernelGGL(k_index_vals, GRID(k), 0, (hipStream_t)s, va, vb, vc, tmp, ca, cb, cc, cnt, k); HIP_LAUNCH_CHECK();
}

}  // namespace gpu
}  // namespace zk
