// csrc/kernels_ntt.hip -- radix-2 NTT over the BLS12-377 / BLS12-381 scalar fields for gfx950.
//
// Replaces ark-poly 0.3.0 Radix2EvaluationDomain::{fft, ifft} (Cargo.lock:234) -- SURVEY.md §8 a16.
// Decimation-in-time with the bit-reversal folded into the first pass's gather, so one transform of 2^lg points is
// ceil-ish(lg / 10) kernel passes instead of lg: every pass stages a tile of up to 1024 field elements in LDS
// (limb-major, 8 x u32 planes -> conflict-free unit-stride lanes), runs up to 10 butterfly stages there, and writes
// the tile back.  Pass 1 reads `src` through bit-reversed addresses (zero-padding short inputs for free) and covers
// stages 0..S1-1 on contiguous 2^S1 tiles; later passes cover S more stages on tiles made of 2^S strided runs of
// 2^L contiguous elements (L = 2 -> 128-byte runs) so HBM sees full-sector accesses.  The inverse transform uses the
// inverse twiddle table and folds the 1/n scaling into the last pass's store.
#include <map>
#include <mutex>
#include <vector>
#include "hip_util.hpp"
#include "ff29.cuh"

namespace zk {
namespace gpu {

template <class Fr>
__global__ void k_fill_powers(Fr w, uint32_t n, Fr *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = w.pow_u64(i);
}

// standard twiddle table (w^i R) -> reduced-radix table (w^i R' as 9 x 29-bit limbs, ff29.cuh)
template <class Fr>
__global__ void k_twiddles29(const Fr *__restrict__ tw, uint32_t n, Fp29<typename Fr::Params> *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Fp29<typename Fr::Params>::twiddle_from_std(tw[i]);
}

// tile element e = (a << L) | lo ; global index = (hi << (s0+S)) | (a << s0) | (mid << L) | lo
//
// Butterflies run on the reduced-radix form (ff29.cuh): data is only re-limbed (8 x 32 -> 9 x 29 bits, still x R), the twiddles come from a
// table of w R' so one 162-multiply product gives t = y w (< 1.5 p); x + t and x - t + 2 p are carry-light limb additions that let the values
// grow -- bounded statically: the first pass starts from canonical inputs (< p); in its first three stages the butterflies with twiddle 1 skip the
// product (y is still small: bounds 1, 3, 7 p -> K = 1, 4, 8 in x - y + K p), from then on every y is multiplied, so the bound grows by 2 p per
// stage: < 29 p after ten stages, far below R' = 2^261 >= 64 p.  Later passes start from values < 2 p (what the previous pass's store leaves) and multiply
// every y: < 22 p after ten stages.  The store takes v - q p with q estimated from the top limb ([0, 2 p), < 2^256: round 5 -- canonical<4>'s five
// compare-and-subtract rounds were ~335 of the ~2,000 instructions an element costs per pass) and only a transform's last pass finishes to [0, p).
// Coset transforms (ntt_coset): the points are g w^i with g = W^c, W a primitive root of a LARGER power-of-two domain of size cs_mask + 1 -- the cosets of H inside the
// 4|H| domain the prover's round 2 evaluates on.  Forward: coefficient k is scaled by g^k while pass 1 gathers it (one more product per element, from the larger
// domain's twiddle table: W^e for e < half, -W^(e - half) above); inverse: coefficient k is scaled by g^-k (and 1/n) at the last pass's store.
// Several independent transforms of one shape (same size, direction and input length) share a launch: blockIdx.y names the job (destination, source, coset index).
// A lone proof's |H|-point passes are latency-bound (30 us for 8 MB): round 2's fifteen transforms are six launches instead of thirty.
constexpr int NTT_MAX_BATCH = 12;
template <class Fr> struct NttBatchArgs { Fr *dst[NTT_MAX_BATCH]; const Fr *src[NTT_MAX_BATCH]; uint32_t cs_c[NTT_MAX_BATCH]; };
template <class Fr, int TILE_LG>
__global__ void __launch_bounds__(256) k_ntt_pass(NttBatchArgs<Fr> batch, bool from_dst, uint32_t in_len, int lg, int s0, int S, int L,
                                                   const Fp29<typename Fr::Params> *__restrict__ tw, bool bitrev_load, bool scale, Fp29<typename Fr::Params> scale_by,
                                                   const Fp29<typename Fr::Params> *__restrict__ cs_tw_all, uint32_t cs_mask, bool final_pass) {
    using G = Fp29<typename Fr::Params>;
    Fr *dst = batch.dst[blockIdx.y];
    const Fr *src = from_dst ? (const Fr *)dst : batch.src[blockIdx.y];
    const uint32_t cs_c = batch.cs_c[blockIdx.y];
    const G *__restrict__ cs_tw = cs_c ? cs_tw_all : nullptr;          // (a plain transform inside a batch of coset transforms)
    constexpr int N = G::N;
    constexpr uint32_t TILE = 1u << TILE_LG;
    __shared__ uint32_t lds[N][TILE];
    const uint32_t tile_elems = 1u << (S + L);
    const uint32_t tile_id = blockIdx.x;
    const uint32_t mid_bits = s0 - L;
    const uint32_t mid = tile_id & ((1u << mid_bits) - 1), hi = tile_id >> mid_bits;
    const uint32_t base = (hi << (s0 + S)) | (mid << L);
    for (uint32_t e = threadIdx.x; e < tile_elems; e += 256) {
        uint32_t a = e >> L, lo = e & ((1u << L) - 1);
        uint32_t gi = base | (a << s0) | lo;
        Fr v;
        if (bitrev_load) {
            uint32_t sidx = __brev(gi) >> (32 - lg);
            if (sidx < in_len) v = src[sidx]; else v = Fr::zero();
        } else {
            v = src[gi];
        }
        G g = G::split(v.l);
        if (bitrev_load && cs_tw && !scale) {         // forward coset transform (scale marks the inverse's last pass): x_k g^k, back to the canonical range the stage bounds start from
            const uint32_t sidx = __brev(gi) >> (32 - lg), ex = (cs_c * sidx) & cs_mask, half = (cs_mask >> 1) + 1;
            G t = g * cs_tw[ex & (half - 1)];
            if (ex >= half) t = G::zero().template sub<2>(t);
            g = t.template canonical<1>();
        }
#pragma unroll
        for (int k = 0; k < N; k++) lds[k][e] = g.l[k];
    }
    __syncthreads();
    const uint32_t half_tile = tile_elems >> 1;
    for (int st = 0; st < S; st++) {
        const int s = s0 + st;                       // butterflies of span 2^s
        // Lazy stages (round 5): from stage 3 of a pass on (the first three keep the normalized forms their twiddle-1 shortcut needs) two stages in three skip the carry
        // chains of x + t and x - t + 2p -- 18 instructions instead of 81 per butterfly.  Limb bound L (units of 2^29) of what sits in LDS: 1 after a normalized stage;
        // a lazy stage adds 1 (p) or 2 (q): 1 -> 3 -> 5, the next stage is normalized again (its y operand, limbs < 5 x 2^29, is within the product's bound of 6; its
        // x operand within operator+ / sub<2>'s bound of 5).  VALUE bounds are untouched (+ 2p per stage as before).  A pass that ends on a lazy stage normalizes at its store.
        // Later passes (s0 > 0) start from canonical values too and their twiddle-1 butterflies are one in 2^s0: they take the product like everybody else and the
        // schedule starts at their first stage (lazy, lazy, normalized, ...).
        const bool lazy = (s0 > 0 || st >= 3) && (st % 3) != 2;
        for (uint32_t b = threadIdx.x; b < half_tile; b += 256) {
            uint32_t lo = b & ((1u << L) - 1), rest = b >> L;
            uint32_t a_low = rest & ((1u << st) - 1), a_high = rest >> st;
            uint32_t a0 = (a_high << (st + 1)) | a_low, a1 = a0 | (1u << st);
            uint32_t e0 = (a0 << L) | lo, e1 = (a1 << L) | lo;
            uint32_t j = (a_low << s0) | (mid << L) | lo;               // position inside the half-span
            G x, y;
#pragma unroll
            for (int k = 0; k < N; k++) { x.l[k] = lds[k][e0]; y.l[k] = lds[k][e1]; }
            G p, q;
            if (s0 == 0 && j == 0 && st < 3) {       // first pass, twiddle 1 while the values are still small: no product
                p = x + y;
                q = st == 0 ? x.template sub<1>(y) : (st == 1 ? x.template sub<4>(y) : x.template sub<8>(y));
            } else {
                G t = y * tw[(size_t)j << (lg - s - 1)];
                if (lazy) { p = x.add_lazy(t); q = x.template sub_lazy<2>(t); }
                else { p = x + t; q = x.template sub<2>(t); }
            }
#pragma unroll
            for (int k = 0; k < N; k++) { lds[k][e0] = p.l[k]; lds[k][e1] = q.l[k]; }
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < tile_elems; e += 256) {
        uint32_t a = e >> L, lo = e & ((1u << L) - 1);
        uint32_t gi = base | (a << s0) | lo;
        G g;
#pragma unroll
        for (int k = 0; k < N; k++) g.l[k] = lds[k][e];
        Fr v;
        const bool last_lazy = (s0 > 0 || S >= 4) && ((S - 1) % 3) != 2;          // (limbs < 3 or 5 x 2^29: fine as the lazy operand of the scaling products, normalized before a bare canonicalisation)
        if (scale && cs_tw) {                         // inverse coset transform: coefficient gi times g^-gi / n
            const uint32_t ex = (cs_c * gi) & cs_mask, half = (cs_mask >> 1) + 1;
            G t = (g * scale_by) * cs_tw[ex & (half - 1)];
            if (ex >= half) t = G::zero().template sub<2>(t);
            t.template canonical<1>().pack(v.l);
        } else if (scale) (g * scale_by).template canonical<1>().pack(v.l);
        else {                                        // < 32 p -> [0, 2 p) by a quotient estimate from the top limb (ff29.cuh): that fits the 8 x 32-bit words a later pass
            const G r = (last_lazy ? g.normalized() : g).reduce_by_top_limb();   // re-limbs (its stage bounds start from 2 p); only the transform's LAST pass pays the round to [0, p)
            (final_pass ? r.template canonical<0>() : r).pack(v.l);
        }
        dst[gi] = v;
    }
}

namespace {
template <class Fr> struct RootOf;
template <> struct RootOf<Fr377> { static constexpr int TWO_ADICITY = FR377_TWO_ADICITY; static Fr377 root() { Fr377 r; for (int i = 0; i < 8; i++) r.l[i] = FR377_ROOT_MONT[i]; return r; } };
template <> struct RootOf<Fr381> { static constexpr int TWO_ADICITY = FR381_TWO_ADICITY; static Fr381 root() { Fr381 r; for (int i = 0; i < 8; i++) r.l[i] = FR381_ROOT_MONT[i]; return r; } };

template <class Fr>
struct Tables {
    std::mutex mu;
    std::map<int, Fr *> fwd, inv, elems;
    std::map<int, Fp29<typename Fr::Params> *> fwd29, inv29;
    static Fr gen(int lg) {
        Fr r = RootOf<Fr>::root();
        for (int i = lg; i < RootOf<Fr>::TWO_ADICITY; i++) r = r.sqr();
        return r;
    }
    // the caches are keyed by (HIP device, log n): keys synthesized on different GPUs of one process get their own tables
    static int cache_key(int lg) { int d = 0; HIP_CHECK(hipGetDevice(&d)); return d * 64 + lg; }
    Fr *powers(std::map<int, Fr *> &m, int lg_, const Fr &w, uint32_t count) {
        std::lock_guard<std::mutex> g(mu);
        const int lg = cache_key(lg_);
        auto it = m.find(lg);
        if (it != m.end()) return it->second;
        Fr *d = (Fr *)dmalloc((size_t)(count ? count : 1) * sizeof(Fr));
        if (count) {
            hipLaunchKernelGGL((k_fill_powers<Fr>), dim3((count + 255) / 256), dim3(256), 0, 0, w, count, d);
            HIP_LAUNCH_CHECK();
            HIP_CHECK(hipDeviceSynchronize());
        }
        m[lg] = d;
        return d;
    }
};
template <class Fr> Tables<Fr> &tables() { static Tables<Fr> t; return t; }
// reduced-radix copy of a twiddle table (built once per domain size and direction)
template <class Fr>
const Fp29<typename Fr::Params> *twiddles29(std::map<int, Fp29<typename Fr::Params> *> &m, int lg_, const Fr *std_table, uint32_t count) {
    using G = Fp29<typename Fr::Params>;
    Tables<Fr> &T = tables<Fr>();
    std::lock_guard<std::mutex> g(T.mu);
    const int lg = Tables<Fr>::cache_key(lg_);
    auto it = m.find(lg);
    if (it != m.end()) return it->second;
    G *d = (G *)dmalloc((size_t)(count ? count : 1) * sizeof(G));
    if (count) {
        hipLaunchKernelGGL((k_twiddles29<Fr>), dim3((count + 255) / 256), dim3(256), 0, 0, std_table, count, d);
        HIP_LAUNCH_CHECK();
        HIP_CHECK(hipDeviceSynchronize());
    }
    m[lg] = d;
    return d;
}
}  // namespace

template <class Fr>
const Fr *domain_elements(int lg) {
    if (lg > RootOf<Fr>::TWO_ADICITY) throw GpuError("domain too large for the field's 2-adicity");
    return tables<Fr>().powers(tables<Fr>().elems, lg, Tables<Fr>::gen(lg), 1u << lg);
}

template <class Fr>
static void ntt_impl(const NttJob<Fr> *jobs, int count, size_t in_len, int lg, bool inverse, int lg_big, stream_t s_, const void *scale_table = nullptr) {
    using G = Fp29<typename Fr::Params>;
    hipStream_t s = (hipStream_t)s_;
    if (count < 1 || count > NTT_MAX_BATCH) throw GpuError("ntt: 1..12 transforms per batch");
    if (lg > RootOf<Fr>::TWO_ADICITY || lg > 30) throw GpuError("ntt: domain too large");
    const uint32_t n = 1u << lg;
    if (in_len > n) in_len = n;
    NttBatchArgs<Fr> batch;
    bool any_coset = scale_table != nullptr;
    for (int i = 0; i < NTT_MAX_BATCH; i++) {
        const NttJob<Fr> &j = jobs[i < count ? i : 0];
        if (j.dst == j.src) throw GpuError("ntt: dst must differ from src (first pass gathers bit-reversed)");
        if (j.coset_c > 0 && (lg_big <= lg || lg_big > RootOf<Fr>::TWO_ADICITY || lg_big > 30)) throw GpuError("ntt_coset: the coset generator must come from a larger domain (lg < lg_big <= 30)");
        if (j.coset_c < 0 || (j.coset_c > 0 && (uint32_t)j.coset_c >= (1u << (lg_big - lg)))) throw GpuError("ntt_coset: coset index out of range");
        batch.dst[i] = j.dst; batch.src[i] = j.src; batch.cs_c[i] = scale_table ? 1u : (uint32_t)j.coset_c;
        any_coset = any_coset || j.coset_c > 0;
    }
    if (lg == 0 && !any_coset) {
        for (int i = 0; i < count; i++) { HIP_CHECK(hipMemcpyAsync(jobs[i].dst, jobs[i].src, sizeof(Fr) * (in_len ? 1 : 0), hipMemcpyDeviceToDevice, s)); if (!in_len) HIP_CHECK(hipMemsetAsync(jobs[i].dst, 0, sizeof(Fr), s)); }
        return;
    }
    if (lg == 0) throw GpuError("ntt_coset: domain of size one");
    oplog_ntt(n, count);
    Fr w = Tables<Fr>::gen(lg);
    const Fr *tw_std = inverse ? tables<Fr>().powers(tables<Fr>().inv, lg, w.inverse(), n / 2) : tables<Fr>().powers(tables<Fr>().fwd, lg, w, n / 2);
    const G *tw = twiddles29<Fr>(inverse ? tables<Fr>().inv29 : tables<Fr>().fwd29, lg, tw_std, n / 2);
    G n_inv = G::twiddle_from_std(Fr::from_u64(n).inverse());
    const G *cs_tw = nullptr;
    uint32_t cs_mask = 0;
    if (scale_table) { cs_tw = (const G *)scale_table; cs_mask = 0xffffffffu; }      // an explicit table g^(+-k), k < 2^lg (coset_power_table): no wrap, no sign
    else if (any_coset) {   // powers of the larger domain's root (its forward or inverse twiddle table: W^(+-e), e < 2^(lg_big - 1))
        if (lg_big <= lg || lg_big > RootOf<Fr>::TWO_ADICITY || lg_big > 30) throw GpuError("ntt_coset: the coset generator must come from a larger domain");
        const uint32_t nb = 1u << lg_big;
        Fr wb = Tables<Fr>::gen(lg_big);
        const Fr *big_std = inverse ? tables<Fr>().powers(tables<Fr>().inv, lg_big, wb.inverse(), nb / 2) : tables<Fr>().powers(tables<Fr>().fwd, lg_big, wb, nb / 2);
        cs_tw = twiddles29<Fr>(inverse ? tables<Fr>().inv29 : tables<Fr>().fwd29, lg_big, big_std, nb / 2);
        cs_mask = nb - 1;
    }
    // pass plan: pass 1 covers min(lg, 10) stages on contiguous 1024-element tiles; the remaining R stages are split EVENLY over
    // ceil(R / 8) passes, each on full 1024-element tiles made of 2^S strided runs of 2^L = 2^(10 - S) contiguous elements (>= 128 B runs),
    // so that every pass keeps all 256 lanes busy (a lopsided 10 + 8 + 4 plan left the last pass with 64-element tiles and made it the slowest).
#ifdef ZKAES_MEASURE
    for (int rep = (knockin() & 32) ? 0 : 1; rep < 2; rep++)         // (measurement builds, ZKAES_KNOCKIN=32: every transform twice -- out of place from src, hence idempotent)
#endif
    {
    int s0 = 0;
    int S1 = lg < 10 ? lg : 10;
    int remaining = lg - S1;
    {
        bool last = remaining == 0;
        hipLaunchKernelGGL((k_ntt_pass<Fr, 10>), dim3(n >> S1, count), dim3(256), 0, s, batch, false, (uint32_t)in_len, lg, 0, S1, 0, tw, true, inverse && last, n_inv,
                           (!inverse || last) ? cs_tw : (const G *)nullptr, cs_mask, last);
        HIP_LAUNCH_CHECK();
        s0 = S1;
    }
    int passes = (remaining + 7) / 8;
    while (remaining > 0) {
        int S = (remaining + passes - 1) / passes;        // even split
        int L = 10 - S;
        if (L > s0) L = s0;
        if (L < 2) L = 2;
        bool last = remaining == S;
        hipLaunchKernelGGL((k_ntt_pass<Fr, 10>), dim3(n >> (S + L), count), dim3(256), 0, s, batch, true, n, lg, s0, S, L, tw, false, inverse && last, n_inv,
                           (inverse && last) ? cs_tw : (const G *)nullptr, cs_mask, last);
        HIP_LAUNCH_CHECK();
        s0 += S;
        remaining -= S;
        passes--;
    }
    }
}
template <class Fr>
static void ntt_impl(Fr *dst, const Fr *src, size_t in_len, int lg, bool inverse, int coset_c, int lg_big, stream_t s, const void *scale_table = nullptr) {
    NttJob<Fr> j{dst, src, coset_c};
    ntt_impl<Fr>(&j, 1, in_len, lg, inverse, lg_big, s, scale_table);
}
template <class Fr>
void ntt_batch(const NttJob<Fr> *jobs, int count, size_t in_len, int lg, bool inverse, int lg_big, stream_t s) { ntt_impl<Fr>(jobs, count, in_len, lg, inverse, lg_big, s); }
template <class Fr>
void ntt(Fr *dst, const Fr *src, size_t in_len, int lg, bool inverse, stream_t s) { ntt_impl<Fr>(dst, src, in_len, lg, inverse, 0, 0, s); }
template <class Fr>
void ntt_coset(Fr *dst, const Fr *src, size_t in_len, int lg, bool inverse, int coset_c, int lg_big, stream_t s) {
    if (lg_big <= lg || lg_big > RootOf<Fr>::TWO_ADICITY || lg_big > 30) throw GpuError("ntt_coset: the coset generator must come from a larger domain (lg < lg_big <= 30)");
    if (coset_c <= 0 || (uint32_t)coset_c >= (1u << (lg_big - lg))) throw GpuError("ntt_coset: coset index out of range");
    ntt_impl<Fr>(dst, src, in_len, lg, inverse, coset_c, lg_big, s);
}

// g^i R' for i < n as reduced-radix limbs: the scaling table of a coset whose generator is NOT a root of unity (the field's multiplicative generator in round 3)
template <class Fr>
void *coset_power_table(const Fr &g, size_t n, stream_t s_) {
    using G = Fp29<typename Fr::Params>;
    hipStream_t s = (hipStream_t)s_;
    DevPtr<Fr> tmp(n);
    DevPtr<G> out(n);
    hipLaunchKernelGGL((k_fill_powers<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, (uint32_t)n, tmp.get());
    HIP_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_twiddles29<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const Fr *)tmp.get(), (uint32_t)n, out.get());
    HIP_LAUNCH_CHECK();
    sync(s_);
    G *r = out.p;
    out.p = nullptr;              // ownership passes to the caller (free with dfree)
    return r;
}
template <class Fr>
void ntt_scaled(Fr *dst, const Fr *src, size_t in_len, int lg, bool inverse, const void *table, stream_t s) {
    if (!table) throw GpuError("ntt_scaled: null table");
    ntt_impl<Fr>(dst, src, in_len, lg, inverse, 0, 0, s, table);
}
template void *coset_power_table<Fr377>(const Fr377 &, size_t, stream_t);
template void ntt_scaled<Fr377>(Fr377 *, const Fr377 *, size_t, int, bool, const void *, stream_t);
template void ntt<Fr377>(Fr377 *, const Fr377 *, size_t, int, bool, stream_t);
template void ntt_batch<Fr377>(const NttJob<Fr377> *, int, size_t, int, bool, int, stream_t);
template void ntt_batch<Fr381>(const NttJob<Fr381> *, int, size_t, int, bool, int, stream_t);
template void ntt<Fr381>(Fr381 *, const Fr381 *, size_t, int, bool, stream_t);
template void ntt_coset<Fr377>(Fr377 *, const Fr377 *, size_t, int, bool, int, int, stream_t);
template void ntt_coset<Fr381>(Fr381 *, const Fr381 *, size_t, int, bool, int, int, stream_t);
template const Fr377 *domain_elements<Fr377>(int);
template const Fr381 *domain_elements<Fr381>(int);

}  // namespace gpu
}  // namespace zk
