// csrc/kernels_msm.hip -- Pippenger multi-scalar multiplication over BLS12-377 / BLS12-381 G1 for gfx950.
//
// Replaces ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul (one rayon task per window, Cargo.lock:118) behind
// KZG10::commit / open (ark-poly-commit 0.3.0) -- SURVEY.md §8 a17.  GPU shape:
//   1. digits + grouping : table mode (the prover's SRS path: 13 balanced signed windows over ONE bucket set, the window weights live in 12 table copies): a two-level
//                      bucket partition written for this layout and for instruction count (k_part_hist / k_part_scatter / k_part_fine) leaves the (point, window) pairs
//                      bucket-contiguous and the bucket ranges; per-window mode and small instances: k_digits + one stable rocPRIM radix sort + k_bounds
//   2. k_order_*     : visiting order of the buckets by descending size (counting sort, deterministic) + the list of oversized buckets
//   3. k_accumulate  : ONE LANE PER BUCKET, the dominant kernel (integer-ALU bound).  EdwardsLaw (BLS12-377 SRS paths): extended twisted Edwards accumulator, 7-product
//                      unified additions of gathered 192-byte Niels28 records (te28.cuh); WeierLaw (arbitrary points, BLS12-381): XYZZ accumulator, 10-product mixed
//                      additions with the P = +-Q case deferred.  k_accumulate_tail: overflow segments of oversized buckets, folded into their buckets by its last workgroup (Edwards) or by k_fold_overflow (Weierstrass)
//   4. reduction     : Edwards: k_reduce_l1_pair (8-bucket lane-interleaved segments, two lanes per segment) -> k_reduce_rc (row / column sums on DPP quads) ->
//                      k_reduce_terms (six unweighted terms per set) -> the weighted tail on the host (or k_reduce_combine); Weierstrass: k_reduce_l1 / l2 -> k_sum_tree -> k_reduce_window
//   5. host          : table mode: nothing (one bucket set); per-window mode: Horner over the <= 37 window sums (c doublings each)
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
#include "hip_util.hpp"
#include "te28.cuh"

namespace zk {
namespace gpu {

// The bucket kernels are written once over a "law": the Weierstrass model with XYZZ accumulators and 112-byte affine bases (both curves, any points), or
// BLS12-377's twisted Edwards model with extended accumulators and 168-byte precomputed (y - x, y + x, 2 d x y) bases (te28.cuh: 7 products per
// bucket addition instead of 10, no special cases) -- what the prover runs over its fixed SRS.
template <class P> struct WeierLaw { using Params = P; using Base = Affine28<P>; using Acc = Acc28<P>; static constexpr bool edwards = false; };
template <class P> struct EdwardsLaw { using Params = P; using Base = Niels28<P>; using Acc = AccTE<P>; static constexpr bool edwards = true; };

static MsmStats g_stats;
static std::mutex g_stats_mu;
MsmStats msm_stats(bool reset) {
    std::lock_guard<std::mutex> g(g_stats_mu);
    MsmStats s = g_stats;
    if (reset) g_stats = MsmStats();
    return s;
}

// Signed window digits d in [-2^(c-1), 2^(c-1)] (carry recoding): bucket index = |d| - 1 in a window of 2^(c-1) buckets, the sign rides
// in bit 31 of the value (the accumulate kernel negates y).  Zero digits (probability 2^-c) stay in bucket 0 of their window with the SKIP bit
// (bit 30) set in the value, so every key is a valid (window, bucket) pair: the pairs are written window-major and the radix sort is stable,
// hence sorting on the c-1 BUCKET bits alone (two 8-bit passes instead of three) already leaves every (window, bucket) group contiguous --
// ordered by (bucket, window), which k_bounds does not care about.
// A launch covers one PART of the scalar list: part element i sits at position i_off + i of every window segment (segment length ntot) and
// names base index val_off + i, so that two scalar vectors over two base ranges can share one Pippenger instance.
constexpr uint32_t VAL_SKIP = 1u << 30, VAL_INDEX = VAL_SKIP - 1;     // value word: bit 31 = negate, bit 30 = skip (zero digit), low 30 bits = base index
template <class Fr>
__global__ void k_digits(const Fr *__restrict__ scalars, uint32_t n, uint32_t i_off, uint32_t ntot, uint32_t val_off, int c, int nwin, uint32_t nb,
                         uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t raw[Fr::N + 1];
    scalars[i].to_raw(raw);
    raw[Fr::N] = 0;
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < nwin; w++) {
        int bit = w * c, limb = bit >> 5, sh = bit & 31;
        uint64_t two = limb <= Fr::N - 1 ? ((uint64_t)raw[limb] | ((uint64_t)raw[limb + 1] << 32)) : 0;
        uint32_t v = ((uint32_t)(two >> sh) & mask) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (v > half) { v = (1u << c) - v; neg = 1u << 31; carry = 1; }
        keys[(size_t)w * ntot + i_off + i] = ((uint32_t)w << (c - 1)) | (v ? v - 1 : 0u);
        vals[(size_t)w * ntot + i_off + i] = v ? ((val_off + i) | neg) : VAL_SKIP;
    }
}

// table mode: every window j reads its own precomputed copy T_j[i] = 2^(off_j) * P_i, so ALL windows share ONE bucket set of 2^(c-1) signed-digit
// buckets: key = |digit| - 1 (zero digits: bucket 0 + SKIP), value = absolute index j * stride + val_off + i into the table array (+ sign / skip bits).
// With one set the running-sum reduction is paid once instead of once per window, so c can grow to 20-22 and the windows shrink to 12-13.
// The windows are BALANCED (TableLayout): the BITS + 1 bits (one for the recoding carry) are spread over the nwin windows as evenly as possible --
// 254 = 7 x 20 + 6 x 19 for c = 20 -- instead of 12 x 20 + a 14-bit top window whose digits all land on 2^13 buckets (n / 2^13 extra points each:
// serial overflow-segment chains, 2.3 ms of k_accumulate_tail per launch in round 2).  The top window's spare bit is always zero, so it never recodes
// and the carry dies there.
struct TableLayout {
    int nwin, c_hi, n_hi;                         // windows j < n_hi are c_hi bits wide, the others c_hi - 1
    __host__ __device__ int width(int j) const { return j < n_hi ? c_hi : c_hi - 1; }
    __host__ __device__ int offset(int j) const { return j < n_hi ? j * c_hi : n_hi * c_hi + (j - n_hi) * (c_hi - 1); }
};
static TableLayout table_layout(int total_bits, int c) {
    TableLayout L;
    L.nwin = (total_bits + c - 1) / c;
    int base = total_bits / L.nwin, rem = total_bits % L.nwin;
    L.c_hi = base + (rem ? 1 : 0);
    L.n_hi = rem ? rem : L.nwin;
    return L;
}
template <class Fr>
__global__ void k_digits_table(const Fr *__restrict__ scalars, uint32_t n, uint32_t i_off, uint32_t ntot, uint32_t val_off, TableLayout L, uint32_t stride,
                               uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t raw[Fr::N + 1];
    scalars[i].to_raw(raw);
    raw[Fr::N] = 0;
    uint32_t carry = 0;
    for (int w = 0; w < L.nwin; w++) {
        const int c = L.width(w), bit = L.offset(w), limb = bit >> 5, sh = bit & 31;
        const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
        uint64_t two = limb <= Fr::N - 1 ? ((uint64_t)raw[limb] | ((uint64_t)raw[limb + 1] << 32)) : 0;
        uint32_t v = ((uint32_t)(two >> sh) & mask) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (v > half) { v = (1u << c) - v; neg = 1u << 31; carry = 1; }
        keys[(size_t)w * ntot + i_off + i] = v ? v - 1 : 0u;
        vals[(size_t)w * ntot + i_off + i] = v ? (((uint32_t)w * stride + val_off + i) | neg) : VAL_SKIP;
    }
}
// T_j = 2^c * T_{j-1}: 8 points per lane, c doublings each, ONE field inversion per lane (Montgomery's trick) back to affine
template <class Fq>
__global__ void __launch_bounds__(64) k_table_next(const Affine<Fq> *__restrict__ prev, Affine<Fq> *__restrict__ next, uint32_t count, int c) {
    constexpr int B = 8;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s0 = t * B;
    if (s0 >= count) return;
    uint32_t e = s0 + B < count ? s0 + B : count;
    Fq xs[B], ys[B], zz[B], zzz[B], pre[B];
    Fq acc = Fq::one();
    for (uint32_t i = s0; i < e; i++) {
        XYZZ<Fq> p = XYZZ<Fq>::from_affine(prev[i]);
        for (int k = 0; k < c; k++) p = p.dbl();
        xs[i - s0] = p.x; ys[i - s0] = p.y; zz[i - s0] = p.zz; zzz[i - s0] = p.zzz;
        pre[i - s0] = acc;
        if (!p.zzz.is_zero()) acc = acc * p.zzz;
    }
    acc = acc.inverse();
    for (uint32_t i = e; i-- > s0;) {
        uint32_t q = i - s0;
        if (zzz[q].is_zero()) { next[i] = Affine<Fq>::inf(); continue; }
        Fq zi3 = acc * pre[q];          // 1 / ZZZ_i
        acc = acc * zzz[q];
        Fq zi2 = (zi3 * zz[q]).sqr();   // 1 / ZZ_i
        Affine<Fq> a; a.x = xs[q] * zi2; a.y = ys[q] * zi3;
        next[i] = a;
    }
}

// four consecutive sorted keys per lane (one 16-byte load + the two neighbours): a group starts where the key changes.  Nothing is memset beforehand: the lane that sees
// the key change also writes the (empty) ranges of the buckets the sorted list skips, the first / last lane those before the first / after the last key.
// The list is ordered by `rank`: table mode (nwin = 1) by the key itself; per-window mode -- keys (window << cbits) | bucket, sorted stably on the bucket bits of a
// window-major list -- by (bucket, window).
__device__ __forceinline__ uint32_t bounds_rank(uint32_t key, int cbits, uint32_t nwin) { return nwin == 1 ? key : (key & ((1u << cbits) - 1)) * nwin + (key >> cbits); }
__device__ __forceinline__ uint32_t bounds_key(uint32_t rank, int cbits, uint32_t nwin) { return nwin == 1 ? rank : ((rank % nwin) << cbits) | (rank / nwin); }
__global__ void k_bounds(const uint32_t *__restrict__ keys, size_t total, int cbits, uint32_t nwin, uint32_t *__restrict__ start, uint32_t *__restrict__ end) {
    size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= total) return;
    const uint32_t nb = nwin << cbits;
    uint32_t k[6];                                        // k[0] = key before the quad, k[5] = key after it
    if (i0 + 4 <= total) { uint4 v = *reinterpret_cast<const uint4 *>(keys + i0); k[1] = v.x; k[2] = v.y; k[3] = v.z; k[4] = v.w; }
    else { for (int j = 0; j < 4; j++) k[1 + j] = i0 + j < total ? keys[i0 + j] : 0u; }
    k[0] = i0 ? keys[i0 - 1] : 0u;
    k[5] = i0 + 4 < total ? keys[i0 + 4] : 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        size_t i = i0 + j;
        if (i >= total) break;
        const uint32_t key = k[1 + j];
        if (i == 0 || k[j] != key) {
            const uint32_t r1 = bounds_rank(key, cbits, nwin);
            uint32_t r0 = 0;
            if (i != 0) { end[k[j]] = (uint32_t)i; r0 = bounds_rank(k[j], cbits, nwin) + 1; }
            for (uint32_t r = r0; r < r1; r++) { const uint32_t g = bounds_key(r, cbits, nwin); start[g] = (uint32_t)i; end[g] = (uint32_t)i; }
            start[key] = (uint32_t)i;
        }
        if (i + 1 == total) {
            end[key] = (uint32_t)total;
            for (uint32_t r = bounds_rank(key, cbits, nwin) + 1; r < nb; r++) { const uint32_t g = bounds_key(r, cbits, nwin); start[g] = (uint32_t)total; end[g] = (uint32_t)total; }
        }
    }
}

// standard (12 x 32, R = 2^384) bases -> reduced-radix copies; done once per SRS at key synthesis, or per call for ad-hoc bases
template <class P>
__global__ void k_convert_bases(const Affine<Fp<P>> *__restrict__ src, Affine28<P> *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = Affine28<P>::from_std(src[i]);
}

// A bucket lane adds at most `cap` points; the rest of an oversized bucket (skewed scalars: many equal digits) is cut into cap-point overflow
// segments that k_accumulate_tail sums in parallel and k_reduce_l1 folds back -- so no input can serialise the whole MSM on one lane.
// cap = 2048 for the per-window buckets (average bucket ~50 points), 512 in table mode (balanced windows: the 2^18 buckets every window reaches hold
// ~20 n / 2^19 points, ~330 for the largest MSM of a 6-block proof; uniform digits therefore never overflow).
constexpr uint32_t BUCKET_CAP = 2048, BUCKET_CAP_TABLE = 512;
constexpr uint32_t NO_SLOT = 0xffffffffu;

// Visiting order of the buckets: descending size, so that the 64 lanes of a wave run the same trip count.  A counting sort written for it (the key is the bucket's
// size clamped to 10 bits; a generic sort of 2^19 (key, id) pairs is ~16 tiny launches): per-workgroup LDS histograms over contiguous bucket ranges, ONE scan by one
// workgroup (bin-major, so buckets of one size keep the order of their ranges: deterministic positions, no global atomics), one scatter.
// The histogram pass also lists the oversized buckets (size > cap) -- slot, bucket and number of overflow segments -- through one atomic counter; the scan workgroup turns
// the segment counts into offsets and re-arms the counter.  For uniformly distributed digits the list is empty.
// ctrl words: [0] list counter (armed at zero between MSMs), [1] list length, [2] overflow segments in total, [3] class-sum flags
constexpr int ORD_BINS = 1024, ORD_THREADS = 256, ORD_MAX_BLOCKS = 128;
__device__ __forceinline__ uint32_t order_key(uint32_t sz) { return (ORD_BINS - 1) - (sz > ORD_BINS - 1 ? ORD_BINS - 1 : sz); }
__global__ void __launch_bounds__(ORD_THREADS) k_order_hist(const uint32_t *__restrict__ start, const uint32_t *__restrict__ end, uint32_t nb, uint32_t per_block, uint32_t cap,
                                                             uint32_t *__restrict__ ovf_slot, uint32_t *__restrict__ ovf_bucket, uint32_t *__restrict__ ovf_nseg, uint32_t ovf_cap,
                                                             uint32_t *__restrict__ ctrl, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[ORD_BINS];
    for (int b = threadIdx.x; b < ORD_BINS; b += ORD_THREADS) h[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * per_block, hi = lo + per_block < nb ? lo + per_block : nb;
    for (uint32_t k = lo + threadIdx.x; k < hi; k += ORD_THREADS) {
        const uint32_t sz = end[k] - start[k];
        uint32_t slot = NO_SLOT;
        if (sz > cap) {
            slot = atomicAdd(&ctrl[0], 1u);
            if (slot < ovf_cap) { ovf_bucket[slot] = k; ovf_nseg[slot] = (sz - 1) / cap; } else slot = NO_SLOT;      // (cannot happen: at most pairs / cap buckets exceed cap)
        }
        ovf_slot[k] = slot;
        atomicAdd(&h[order_key(sz)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < ORD_BINS; b += ORD_THREADS) hist[(size_t)blockIdx.x * ORD_BINS + b] = h[b];
}
// one workgroup, one bin per lane: exclusive scan of the (bin, workgroup) counts in bin-major order (the arrays are workgroup-major, so that the lanes' loads coalesce);
// then the overflow list's segment offsets
__global__ void __launch_bounds__(ORD_BINS) k_order_scan(const uint32_t *__restrict__ hist, uint32_t *__restrict__ offs, uint32_t nblk, uint32_t *__restrict__ ctrl,
                                                          const uint32_t *__restrict__ ovf_nseg, uint32_t *__restrict__ ovf_off, uint32_t ovf_cap) {
    __shared__ uint32_t part[ORD_BINS];
    const uint32_t t = threadIdx.x;
    auto block_exclusive = [&](uint32_t v) {            // exclusive prefix of v over the workgroup; part[ORD_BINS - 1] holds the total afterwards
        part[t] = v;
        __syncthreads();
        for (int d = 1; d < ORD_BINS; d <<= 1) {
            uint32_t add = t >= (unsigned)d ? part[t - d] : 0;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        return part[t] - v;
    };
    // (a lone call waits for this one workgroup: the loads of a lane are independent, so they go out eight at a time instead of one latency each)
    uint32_t sum = 0;
    uint32_t i0 = 0;
    for (; i0 + 8 <= nblk; i0 += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = hist[(size_t)(i0 + u) * ORD_BINS + t];
#pragma unroll
        for (int u = 0; u < 8; u++) sum += c[u];
    }
    for (; i0 < nblk; i0++) sum += hist[(size_t)i0 * ORD_BINS + t];
    uint32_t run = block_exclusive(sum);
    for (i0 = 0; i0 + 8 <= nblk; i0 += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = hist[(size_t)(i0 + u) * ORD_BINS + t];
#pragma unroll
        for (int u = 0; u < 8; u++) { offs[(size_t)(i0 + u) * ORD_BINS + t] = run; run += c[u]; }
    }
    for (; i0 < nblk; i0++) { const uint32_t c = hist[(size_t)i0 * ORD_BINS + t]; offs[(size_t)i0 * ORD_BINS + t] = run; run += c; }
    uint32_t n = ctrl[0];
    if (n > ovf_cap) n = ovf_cap;
    uint32_t base = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += ORD_BINS) {     // (empty for uniformly distributed digits)
        __syncthreads();
        const uint32_t v = c0 + t < n ? ovf_nseg[c0 + t] : 0;
        const uint32_t ex = block_exclusive(v);
        if (c0 + t < n) ovf_off[c0 + t] = base + ex;
        base += part[ORD_BINS - 1];
    }
    __syncthreads();
    if (t == 0) { ovf_off[n] = base; ctrl[1] = n; ctrl[2] = base; ctrl[0] = 0; }
}
__global__ void __launch_bounds__(ORD_THREADS) k_order_scatter(const uint32_t *__restrict__ start, const uint32_t *__restrict__ end, uint32_t nb, uint32_t per_block,
                                                                const uint32_t *__restrict__ offs, uint32_t *__restrict__ order) {
    __shared__ uint32_t h[ORD_BINS];
    for (int b = threadIdx.x; b < ORD_BINS; b += ORD_THREADS) h[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * per_block, hi = lo + per_block < nb ? lo + per_block : nb;
    for (uint32_t k = lo + threadIdx.x; k < hi; k += ORD_THREADS) {
        const uint32_t key = order_key(end[k] - start[k]);
        order[offs[(size_t)blockIdx.x * ORD_BINS + key] + atomicAdd(&h[key], 1u)] = k;
    }
}

// ONE LANE PER BUCKET, buckets visited in descending-size order so the 64 lanes of a wave run the same trip count.
// The loop body has no function call: the (cryptographically negligible, but reachable with repeated bases) P == +-Q case is
// appended to a deferred list and replayed by the last workgroup of k_accumulate_tail with the complete addition law.  Buckets stay in the
// reduced-radix form through the reduction kernels; only the per-window sums are converted back for the host.
// At least two waves per SIMD: the kernel issues at its limit there (three or four waves with spills are slower: profiles/r03_accumulate_occupancy.txt; round 4's build
// fits three without spills -- 162 registers -- and measures exactly like the same code capped at two, profiles/r04_accumulate_waves.txt).
template <class Law>
__global__ void __launch_bounds__(64, 2) k_accumulate(const typename Law::Base *__restrict__ bases, const uint32_t *__restrict__ vals,
                                                       const uint32_t *__restrict__ start, const uint32_t *__restrict__ end, const uint32_t *__restrict__ order,
                                                       uint32_t nbuckets_total, uint32_t cap, typename Law::Acc *__restrict__ buckets,
                                                       uint32_t *__restrict__ deferred, uint32_t deferred_cap, uint32_t *__restrict__ deferred_count) {
    using P = typename Law::Params;
    using G = FpMsm<P>;
    [[maybe_unused]] uint64_t bias = 0;
    if constexpr (Law::edwards) bias = FpMsm<P>::hot_loop_bias();           // before anything else: see ff28.cuh mul_biased
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbuckets_total) return;
    uint32_t k = order[t];
    uint32_t s = start[k], e = end[k];
    if (e - s > cap) e = s + cap;                                          // the remainder goes through the overflow segments of k_accumulate_tail
    if constexpr (Law::edwards) {
        // unified law: the accumulator starts at the identity, P = +-Q and identity bases need no branch, nothing is deferred.  The next point's gather is issued in the
        // middle of the current addition, into the registers the current point no longer needs; a negative digit's record arrives with (y - x, y + x) swapped (te_madd_hot).
        AccTE<P> acc = te_identity<P>();
        if (s < e) {
            uint32_t idx = vals[s];
            Niels28<P> p = niels_load_signed<P>(bases + (idx & VAL_INDEX), idx >> 31);
            for (uint32_t i = s; i < e; i++) {
                const uint32_t cur = idx;
                if (i + 1 < e) idx = vals[i + 1];
                const Niels28<P> *next = bases + (idx & VAL_INDEX);        // the last iteration re-reads its own point: no branch around the load
                if (cur & VAL_SKIP) { p = niels_load_signed<P>(next, idx >> 31); continue; }
                te_madd_hot<P>(acc, p, cur >> 31, next, idx >> 31, bias);
            }
        }
        buckets[k] = acc;
    } else {
        Acc28<P> acc;
        bool acc_inf = true;
        if (s < e) {
            uint32_t idx = vals[s];
            Affine28<P> nxt = bases[idx & VAL_INDEX];
            for (uint32_t i = s; i < e; i++) {
                Affine28<P> p = nxt;
                uint32_t cur = idx;
                if (i + 1 < e) { idx = vals[i + 1]; nxt = bases[idx & VAL_INDEX]; }   // prefetch the next gather under this add's ALU work
                if ((cur & VAL_SKIP) || p.is_inf()) continue;
                if (cur >> 31) p.y = G::zero().template sub<2>(p.y);                 // negative digit: add -P  (y < 1.2 p as a product, so 2p - y > 0)
                if (acc_inf) { acc.x = p.x; acc.y = p.y; acc.zz = G::k_one(); acc.zzz = acc.zz; acc_inf = false; continue; }
                if (!madd28(acc, p)) {
                    uint32_t slot = atomicAdd(deferred_count, 1u);
                    if (slot < deferred_cap) { deferred[2 * slot] = k; deferred[2 * slot + 1] = cur; }
                }
            }
        }
        if (acc_inf) acc = inf28<P>();
        buckets[k] = acc;
    }
}
// replay of the deferred (bucket, point) pairs with the complete formulas; one lane, sequential (the list is empty in practice)
template <class P>
__device__ void accumulate_fixup(const Affine28<P> *__restrict__ bases, Acc28<P> *__restrict__ buckets, const uint32_t *__restrict__ deferred, uint32_t deferred_cap,
                                 const volatile uint32_t *deferred_count) {
    uint32_t n = *deferred_count;
    if (n > deferred_cap) n = deferred_cap;
    for (uint32_t i = 0; i < n; i++) {
        XYZZ<Fp<P>> b = to_std_point<P>(buckets[deferred[2 * i]]);
        uint32_t v = deferred[2 * i + 1];
        Affine<Fp<P>> q = bases[v & VAL_INDEX].to_std();
        b.madd((v >> 31) ? q.neg() : q);
        buckets[deferred[2 * i]] = from_std_point<P>(b);
    }
}

// overflow segment t of list entry q (bucket k = ovf_bucket[q]) covers sorted positions [start[k] + (j+1) CAP, min(start[k] + (j+2) CAP, end[k])), j = t - ovf_off[q]
// ONE tail launch per MSM: the overflow segments, then -- Weierstrass law only, in whichever workgroup finishes last (ticket counter deferred_count[1]) -- the replay of
// the deferred degenerate additions of this kernel and of k_accumulate.  The overflow partials are folded into their buckets by k_fold_overflow, launched behind this kernel (Weierstrass), or by
// this kernel's last workgroup itself (Edwards, round 6: the same ticket).
// For uniformly distributed digits there are no segments and no deferred pairs: every lane exits after one load.
template <class Law>
__global__ void __launch_bounds__(64) k_accumulate_tail(const typename Law::Base *__restrict__ bases, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ start,
                                                            const uint32_t *__restrict__ end, const uint32_t *__restrict__ ctrl, const uint32_t *__restrict__ ovf_bucket,
                                                            const uint32_t *__restrict__ ovf_off, uint32_t max_segments, uint32_t cap,
                                                            typename Law::Acc *__restrict__ partial, typename Law::Acc *__restrict__ buckets, uint32_t *__restrict__ deferred, uint32_t deferred_cap,
                                                            uint32_t *__restrict__ deferred_count) {
    using P = typename Law::Params;
    using A = typename Law::Acc;
    using G = FpMsm<P>;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nlist = ctrl[1];
    uint32_t total = ctrl[2];
    if (total > max_segments) total = max_segments;
    // Edwards law (nothing is ever deferred): with no oversized bucket -- always, for uniformly distributed digits -- there is nothing to accumulate and nothing to fold
    if constexpr (Law::edwards) { if (nlist == 0) return; }
    // workgroups whose 64 segment slots are all beyond `total` (every workgroup, for uniform digits) skip the segment work and its LDS fold
    if (blockIdx.x * blockDim.x < total) {
        __shared__ A sh[64];
        __shared__ uint32_t key[64];
        uint32_t q = NO_SLOT;
        A acc;
        if (t < total) {
            uint32_t lo = 0, hi = nlist;                      // largest q with ovf_off[q] <= t
            while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (ovf_off[mid] <= t) lo = mid; else hi = mid; }
            q = lo;
            const uint32_t k = ovf_bucket[q], j = t - ovf_off[q];
            uint32_t s = start[k] + (j + 1) * cap, e = s + cap < end[k] ? s + cap : end[k];
            if constexpr (Law::edwards) {
                acc = te_identity<P>();
                for (uint32_t i = s; i < e; i++) {
                    uint32_t cur = vals[i];
                    if (cur & VAL_SKIP) continue;
                    Niels28<P> p = bases[cur & VAL_INDEX];
                    if (cur >> 31) p = niels_neg<P>(p);
                    te_madd<P>(acc, p);
                }
            } else {
                bool acc_inf = true;
                for (uint32_t i = s; i < e; i++) {
                    uint32_t cur = vals[i];
                    Affine28<P> p = bases[cur & VAL_INDEX];
                    if ((cur & VAL_SKIP) || p.is_inf()) continue;
                    if (cur >> 31) p.y = G::zero().template sub<2>(p.y);
                    if (acc_inf) { acc.x = p.x; acc.y = p.y; acc.zz = G::k_one(); acc.zzz = acc.zz; acc_inf = false; continue; }
                    if (!madd28(acc, p)) {
                        uint32_t slot = atomicAdd(deferred_count, 1u);
                        if (slot < deferred_cap) { deferred[2 * slot] = k; deferred[2 * slot + 1] = cur; }
                    }
                }
                if (acc_inf) acc = inf28<P>();
            }
            sh[threadIdx.x] = acc;
        }
        key[threadIdx.x] = q;
        __syncthreads();
        // the segments of one bucket are consecutive t: the first lane of every (workgroup, bucket) run folds its run (<= 63 additions) and stores ONE
        // partial at its own slot; k_fold_overflow then visits one slot per workgroup the bucket's segments span, not one per segment
        if (t < total && (threadIdx.x == 0 || key[threadIdx.x - 1] != q)) {
            for (uint32_t r = threadIdx.x + 1; r < 64 && key[r] == q; r++) {
                if constexpr (Law::edwards) PtOps<A>::add_inline(acc, sh[r]); else PtOps<A>::add(acc, sh[r]);        // (inlined on the Edwards law: no call, no stack)
            }
            partial[t] = acc;
        }
    }
    // whichever workgroup finishes last (ticket counter deferred_count[1], re-armed by its taker) closes the pass: on the Weierstrass law it replays the deferred P = +-Q
    // additions with the complete formulas (k_fold_overflow follows as its own launch); on the Edwards law it folds the overflow partials into their buckets itself --
    // round 6: one dependent launch less behind every accumulation, and no stack in either (round 5's two kernels carried 240 B of scratch each for a call to the point
    // addition that uniform digits never reach, and took 45-77 us to do nothing behind a lone call's accumulations, profiles/r06_lone_timeline_16.md)
    __shared__ uint32_t ticket;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(deferred_count + 1, 1u);
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    __threadfence();
    if (threadIdx.x == 0) deferred_count[1] = 0;
    if constexpr (Law::edwards) {
        // a bucket's partials sit one per 64-segment workgroup its segments [a, b) span, at the run's first slot
        for (uint32_t q = threadIdx.x; q < nlist; q += blockDim.x) {
            uint32_t a = ovf_off[q], b = ovf_off[q + 1];
            if (b > max_segments) b = max_segments;
            if (a >= b) continue;
            const uint32_t k = ovf_bucket[q];
            A acc = buckets[k];
            for (uint32_t w = a / 64; w * 64 < b; w++) { const uint32_t i = w * 64 > a ? w * 64 : a; if (i < b) PtOps<A>::add_inline(acc, partial[i]); }
            buckets[k] = acc;
        }
    } else {
        if (threadIdx.x != 0) return;
        accumulate_fixup<P>(bases, buckets, deferred, deferred_cap, deferred_count);
    }
}
// Overflow partials -> their buckets, in place, BEFORE the reduction (round 5; the reduction's loads used to fold them in, which put a call to the point addition --
// and its register save area: 464-704 B of scratch -- into every first-level reduction kernel for a case uniform digits never reach).  One lane per entry of the overflow
// list (ctrl[1] entries: zero for uniformly distributed digits, where every lane leaves after one load); a bucket's partials sit one per 64-segment workgroup of
// k_accumulate_tail that its segments [a, b) span, at the run's first slot.
template <class A>
__global__ void __launch_bounds__(64) k_fold_overflow(A *__restrict__ buckets, const uint32_t *__restrict__ ctrl, const uint32_t *__restrict__ ovf_bucket, const uint32_t *__restrict__ ovf_off,
                                                       uint32_t max_segments, const A *__restrict__ partial) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ctrl[1]) return;
    uint32_t a = ovf_off[q], b = ovf_off[q + 1];
    if (b > max_segments) b = max_segments;
    if (a >= b) return;
    const uint32_t k = ovf_bucket[q];
    A acc = buckets[k];
    for (uint32_t w = a / 64; w * 64 < b; w++) { uint32_t i = w * 64 > a ? w * 64 : a; if (i < b) PtOps<A>::add(acc, partial[i]); }
    buckets[k] = acc;
}

// Bucket reduction  sum_j (j + 1) * B_j  per window (bucket j holds digit magnitude j + 1), in three fully parallel levels:
//   k_reduce_l1: lane (w, g) over the 8 buckets j0 = 8 g ..: S_g = sum B_j, W_g = sum (j - j0 + 1) B_j             (running sums only)
//   k_reduce_l2: lane (w, h) over 8 segments: sum_g [W_g + 8 g S_g] via a second running sum + ONE small scalar product
//   k_reduce_window: LDS tree over the group partials of a window
// A = the accumulator type (Acc28: XYZZ on the Weierstrass model, AccTE: extended twisted Edwards); PtOps<A> is the group law.
constexpr int RED_L1 = 8, RED_L2 = 8;
template <class A>
__global__ void __launch_bounds__(64) k_reduce_l1(const A *__restrict__ buckets, int c, int nwin, A *__restrict__ seg_s, A *__restrict__ seg_w) {
    uint32_t segs = (1u << c) / RED_L1;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= segs * (uint32_t)nwin) return;
    uint32_t w = t / segs, g = t % segs;
    const size_t k0 = ((size_t)w << c) + (size_t)g * RED_L1;
    // run_d = B_7 + ... + B_d, W = sum_d run_d.  The running sum starts from the top bucket itself and the sum of the running sums lags one step behind it: 14 additions instead
    // of 16 (nothing is added to an identity), and the two additions of a step are independent of each other -- tot takes the PREVIOUS running sum while the next one forms
    A run = buckets[k0 + RED_L1 - 1], tot = run;
    for (int d = RED_L1 - 2; d >= 0; d--) {
        A b = buckets[k0 + d];
        if (d != RED_L1 - 2) PtOps<A>::add_inline(tot, run);   // + run_{d+1}  (tot already holds run_7 when d = 6)
        PtOps<A>::add_inline(run, b);                           // run_d
    }
    PtOps<A>::add(tot, run);                                    // + run_0
    seg_s[t] = run;
    seg_w[t] = tot;
}
// The same first level for the Edwards law with LANE-INTERLEAVED segments: segment g of a set takes buckets d G + g, d < 8 (G = 2^c / 8 segments per set), so for every d the
// lanes of a wave read consecutive 224-byte buckets -- one contiguous stretch per wave and load step instead of 16-byte pieces of 64 lines 1,792 B apart (the
// consecutive-bucket layout above spent ~130 of its 290 us on those loads, profiles/r04_reduce_scan.txt).  With j = d G + g:
//     sum_j (j + 1) B_j  =  G sum_g W'_g  +  sum_g g S_g  +  sum_g S_g,      S_g = sum_d B_{d,g},   W'_g = sum_d d B_{d,g}
// -- the same row / column decomposition of sum_g g S_g behind it (k_reduce_rc), one more plain sum and lg G doublings at the end.
// Round 5: TWO LANES PER SEGMENT.  Round 4's kernel kept both running sums of a segment in one lane (run, tot and the loaded bucket: 3 x 56 registers beside the
// product's 56 + 56 -> 229 VGPRs and 688 B of scratch, one wave per SIMD, 13 dependent additions).  Here the even lane of a pair carries run_d = B_7 + ... + B_d and the odd
// lane tot = run_7 + ... + run_1, one step behind: at every step the odd lane adds what the even lane held BEFORE the step (56 DPP quad_perm moves, no LDS), the even lane
// adds the next bucket.  Both lanes execute the same seven additions -- the odd lane's first one adds run_7 to the identity, which the unified law takes in its stride --
// so there is no divergence; a lane holds ONE accumulator and ONE operand (no scratch), the chain is 7 additions instead of 13, and twice the waves share a SIMD.
template <class A>
__global__ void __launch_bounds__(64) k_reduce_l1_pair(const A *__restrict__ buckets, int c, int nwin, A *__restrict__ seg_s, A *__restrict__ seg_w) {
    using P = typename PtOps<A>::Params;
    using G = FpMsm<P>;
    const uint32_t segs = (1u << c) / RED_L1;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, seg = t >> 1;
    const bool odd = t & 1;
    if (seg >= segs * (uint32_t)nwin) return;              // (pairs never straddle the bound: both lanes of a pair leave together)
    const uint32_t w = seg / segs, g = seg % segs;
    const size_t k0 = ((size_t)w << c) + g;
    A acc = odd ? PtOps<A>::identity() : buckets[k0 + (size_t)(RED_L1 - 1) * segs];
    for (int d = RED_L1 - 2; d >= 0; d--) {
        A opnd;
        // what the partner holds now: for the odd lane that is run_{d+1}
        opnd.x = quad_move<QP_SWAP_PAIRS, G>(acc.x); opnd.y = quad_move<QP_SWAP_PAIRS, G>(acc.y); opnd.z = quad_move<QP_SWAP_PAIRS, G>(acc.z); opnd.t = quad_move<QP_SWAP_PAIRS, G>(acc.t);
        if (!odd) opnd = buckets[k0 + (size_t)d * segs];
        PtOps<A>::add_inline(acc, opnd);
    }
    (odd ? seg_w : seg_s)[seg] = acc;                       // even: S_g = run_0;  odd: W'_g = run_7 + ... + run_1
}
// Two roles per group (whole 64-lane workgroups take one role, so nobody diverges): a lane's time is its NUMBER of point operations, and one lane doing both the
// running sums (29 operations) and the scalar product (8 + up to 28) was the longest chain of the whole reduction (630 us of the 1.36 ms a bucket reduction costs
// a lone encrypt() call).  Role 0 writes sw + L1 (tot2 - run) to partial[.. 2h], role 1 sums its S_g again and writes (L1 g0) run to partial[.. 2h + 1]; the trees behind
// add everything up anyway.
template <class A>
__global__ void __launch_bounds__(64) k_reduce_l2(const A *__restrict__ seg_s, const A *__restrict__ seg_w, int c, int nwin, A *__restrict__ partial) {
    uint32_t segs = (1u << c) / RED_L1, groups = (segs + RED_L2 - 1) / RED_L2;
    const uint32_t role = blockIdx.x & 1;
    uint32_t t = (blockIdx.x >> 1) * blockDim.x + threadIdx.x;
    if (t >= groups * (uint32_t)nwin) return;
    uint32_t w = t / groups, h = t % groups;
    uint32_t g0 = h * RED_L2, g1 = g0 + RED_L2 < segs ? g0 + RED_L2 : segs;
    const A *S = seg_s + (size_t)w * segs, *W = seg_w + (size_t)w * segs;
    A *out = partial + (size_t)w * 2 * groups + 2 * h + role;
    if (role == 0) {
        A run = PtOps<A>::identity(), tot2 = PtOps<A>::identity(), sw = PtOps<A>::identity();
        for (int g = (int)g1 - 1; g >= (int)g0; g--) {
            PtOps<A>::add(run, S[g]);
            PtOps<A>::add(tot2, run);           // tot2 = sum (g - g0 + 1) S_g
            PtOps<A>::add(sw, W[g]);
        }
        // bucket j carries weight j + 1:  sum_g [W_g + L1 g S_g] = sw + L1 (tot2 - run) + (L1 g0) run     (run = sum S_g, L1 = 8); the last term is role 1's
        A a = tot2;
        PtOps<A>::add(a, PtOps<A>::neg(run));
        for (int i = 0; i < 3; i++) PtOps<A>::dbl(a);      // * RED_L1
        PtOps<A>::add(sw, a);
        *out = sw;
    } else {
        A acc = PtOps<A>::identity();
        if (g0 != 0) {
            A run = PtOps<A>::identity();
            for (int g = (int)g1 - 1; g >= (int)g0; g--) PtOps<A>::add(run, S[g]);
            uint32_t m = RED_L1 * g0;
            int top = 31 - __clz(m);
            for (int bit = top; bit >= 0; bit--) {
                PtOps<A>::dbl(acc);
                if ((m >> bit) & 1) PtOps<A>::add(acc, run);
            }
        }
        *out = acc;
    }
}

template <class A>
__global__ void __launch_bounds__(256) k_reduce_window(const A *__restrict__ partial, uint32_t per_window, XYZZ<Fp<typename PtOps<A>::Params>> *__restrict__ out,
                                                        XYZZ<Fp<typename PtOps<A>::Params>> *__restrict__ out2) {
    __shared__ A sh[256];
    uint32_t w = blockIdx.x, t = threadIdx.x;
    A acc = PtOps<A>::identity();
    for (uint32_t i = t; i < per_window; i += 256) PtOps<A>::add(acc, partial[(size_t)w * per_window + i]);
    sh[t] = acc;
    __syncthreads();
    int s0 = 128;
    while (s0 >= 1 && (uint32_t)s0 >= per_window) s0 >>= 1;        // slots beyond per_window hold the identity: skip the tree levels that would only add those
    for (int s = s0; s > 0; s >>= 1) {
        if ((int)t < s) { A a = sh[t]; PtOps<A>::add(a, sh[t + s]); sh[t] = a; }
        __syncthreads();
    }
    if (t == 0) {                                     // always the Weierstrass XYZZ form in the library-wide Montgomery representation
        const auto r = PtOps<A>::to_std(sh[0]);
        out[w] = r;                                   // (host-mapped pinned memory in the prover: no copy launch)
        if (out2) out2[w] = r;
    }
}

// ---- The reduction behind k_reduce_l1 on the Edwards law: a 2-D decomposition with four lanes per point operation (te28.cuh te_add_quad / te_dbl_quad).
// k_reduce_l1_interleaved leaves, per segment g of a bucket set (8 buckets G apart), S_g = sum B and W'_g = sum d B; the set's sum is  G sum_g W'_g + sum_g g S_g + sum_g S_g.
// With g = r C + c (G = R C segments, both powers of two):  sum_g g S_g = sum_c c CS_c + C sum_r r RS_r, where CS_c (column sums) and RS_r (row sums) are PLAIN sums -- trees, no running
// sums, no scalar products on thousands of lanes (k_reduce_l2's second role did a 16-bit double-and-add per 64 buckets).
//   k_reduce_rc    : one 64-quad workgroup per plain sum: the R row sums of W (their total is sum_g W_g), the R row sums RS_r and the C column sums CS_c of S
//   k_reduce_terms : six small workgroups per set, one per unweighted term (the total of the W' sums; the total of the row sums; V1, V2 of sum_c c CS_c; V1, V2 of
//                    sum_r r RS_r -- an index-weighted sum of <= 256 points is split 16 x 16 once more, then taken bit plane by bit plane); the weights are applied by the
//                    host (reduce_host_tail) or by k_reduce_combine
// Depth of the whole reduction: 16 sequential additions in k_reduce_l1, then ~40 quad operations of 2-3 product-times each, instead of ~70 whole additions.
constexpr int RQ_THREADS = 256, RQ_QUADS = RQ_THREADS / 4;          // k_reduce_rc
constexpr int PT_WORDS = 4 * FpMsm<Fq377P>::N;                      // one extended point: 4 coordinates x 14 limbs (224 B: sizeof(AccTE))
template <class P> __device__ __forceinline__ FpMsm<P> quad_load(const uint32_t *pt, int q) {
    FpMsm<P> r;
#pragma unroll
    for (int i = 0; i < FpMsm<P>::N; i++) r.l[i] = pt[q * FpMsm<P>::N + i];
    return r;
}
template <class P> __device__ __forceinline__ void quad_store(uint32_t *pt, int q, const FpMsm<P> &v) {
#pragma unroll
    for (int i = 0; i < FpMsm<P>::N; i++) pt[q * FpMsm<P>::N + i] = v.l[i];
}
// sum of pt[0 .. count) (count a power of two <= the number of quads) into pt[0]; every lane of the workgroup must call it.  Ends with a barrier.
template <class P>
__device__ __forceinline__ void quad_tree_sum(uint32_t *pt, uint32_t count, uint32_t quad, int q) {
    for (uint32_t half = count >> 1; half >= 1; half >>= 1) {
        if (quad < half) quad_store<P>(pt + quad * PT_WORDS, q, te_add_quad<P>(quad_load<P>(pt + quad * PT_WORDS, q), quad_load<P>(pt + (quad + half) * PT_WORDS, q), q));
        __syncthreads();
    }
}
// jobs per set: [0, R) row sums of W, [R, 2R) row sums of S, [2R, 2R + C) column sums of S; out[set][job]
template <class P>
__global__ void __launch_bounds__(RQ_THREADS) k_reduce_rc(const AccTE<P> *__restrict__ seg_s, const AccTE<P> *__restrict__ seg_w, int lgR, int lgC, AccTE<P> *__restrict__ out) {
    __shared__ uint32_t pt[RQ_QUADS * PT_WORDS];
    const uint32_t R = 1u << lgR, C = 1u << lgC, jobs = 2 * R + C;
    const uint32_t set = blockIdx.x / jobs, job = blockIdx.x % jobs, quad = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    const AccTE<P> *src = (job < R ? seg_w : seg_s) + ((size_t)set << (lgR + lgC));
    uint32_t first, stride, len;
    if (job < 2 * R) { first = (job < R ? job : job - R) << lgC; stride = 1; len = C; }
    else { first = job - 2 * R; stride = C; len = R; }
    FpMsm<P> acc = te_identity_quad<P>(q);
    for (uint32_t i = quad; i < len; i += RQ_QUADS)
        acc = te_add_quad<P>(acc, quad_load<P>(reinterpret_cast<const uint32_t *>(src + first + (size_t)i * stride), q), q);
    quad_store<P>(pt + quad * PT_WORDS, q, acc);
    __syncthreads();
    quad_tree_sum<P>(pt, RQ_QUADS, quad, q);
    if (quad == 0) quad_store<P>(reinterpret_cast<uint32_t *>(out + (size_t)set * jobs + job), q, quad_load<P>(pt, q));
}
// ---- k_reduce_terms: the last level of the Edwards reduction, SIX small workgroups per bucket set -- one per term of
//     set sum = 2^(lgR+lgC) [0] + [1] + ([2] + 2^lc(C) [3]) + 2^lgC ([4] + 2^lc(R) [5])
//   [0] = sum of the R row sums of W'      [1] = sum of the R row sums RS_r of S
//   [2], [3] = V1, V2 of the column term sum_c c CS_c      [4], [5] = V1, V2 of the row term sum_r r RS_r
// where an index-weighted sum over M <= 256 points P_i is split once more, i = r c1 + c (c1 = 2^lc columns, r1 = M / c1 rows):
//     sum_i i P_i = V1 + c1 V2,   V1 = sum_c c (sum_r P[r c1 + c]),   V2 = sum_r r (sum_c P[r c1 + c])
// and a weighted sum over <= 16 group sums is taken bit plane by bit plane (plane b = the sum of the groups whose index has bit b set; Horner over the planes).
// Round 5's k_reduce_final did the same arithmetic in FOUR 1024-lane workgroups per set with every quad operation inlined where it was used: 90,456 instructions (0.7 MB of
// straight-line code against a 64 KB instruction cache), 128 registers + 392 B of scratch, and a workgroup that needs a whole empty CU (16 waves x 128 registers) -- 174 us alone,
// 620 us behind another lane's accumulation grid.  Here every point operation of a workgroup goes through ONE of two te_add_quad call sites (a doubling is
// the unified addition of a point to itself): a gather loop from global memory and a step loop over LDS slots whose (destination, operand, operand) triples are plain index
// arithmetic.  256 lanes (64 quads), ~6 k instructions, no scratch, 21 KB of LDS: it fits beside two accumulation waves per SIMD.
// The weights are applied by the host (reduce_host_tail: ~24 XYZZ doublings of 0.9 us) when the terms go to host-mapped memory, or by k_reduce_combine below when the sum has to
// stay on the device (sharded MSM entry points) or there are dozens of sets (per-window buckets).
ZK_HD int quad_weighted_lc(uint32_t M) { int m = 0; while ((1u << m) < M) m++; return M <= 16 ? m : (m + 1) / 2; }
constexpr int RF_OUT = 6;
constexpr int RT_THREADS = 256, RT_QUADS = RT_THREADS / 4;
constexpr int RT_PLANES = RT_QUADS, RT_IDENT = RT_QUADS + 32, RT_SLOTS = RT_IDENT + 1;      // LDS slots: 64 partials | 4 planes x 8 | the identity
__device__ __forceinline__ uint32_t rt_plane_member(uint32_t b, uint32_t i) {                 // i-th index of [0, 16) with bit b set
    const uint32_t lo = i & ((1u << b) - 1), hi = i >> b;
    return (hi << (b + 1)) | (1u << b) | lo;
}
template <class P>
__global__ void __launch_bounds__(RT_THREADS) k_reduce_terms(const AccTE<P> *__restrict__ rc, int lgR, int lgC, AccTE<P> *__restrict__ part, XYZZ<Fp<P>> *__restrict__ out, bool to_host) {
    __shared__ uint32_t pt[RT_SLOTS * PT_WORDS];
    const uint32_t R = 1u << lgR, C = 1u << lgC, jobs = 2 * R + C;
    const uint32_t set = blockIdx.x / RF_OUT, term = blockIdx.x % RF_OUT, quad = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    const bool plain = term < 2, second = term & 1;                                           // weighted terms: V1 (even) or V2 (odd) of their point list
    const AccTE<P> *src = rc + (size_t)set * jobs + (term == 0 ? 0 : (term == 2 || term == 3) ? 2 * R : R);
    const uint32_t M = (term == 2 || term == 3) ? C : R;
    // ---- gather: every quad sums its share of the source points (<= 4 of them) into its slot
    uint32_t first = 0, stride = 0, count = 0;
    if (plain) { first = quad; stride = RT_QUADS; count = quad < M ? (M - quad + RT_QUADS - 1) / RT_QUADS : 0; }
    else {
        const int lc = quad_weighted_lc(M);
        const uint32_t c1 = 1u << lc, r1 = M >> lc, G = second ? r1 : c1, S = second ? c1 : r1, g = quad >> 2, p = quad & 3;      // group g, member share p of 4
        if (g < G && p < S) { count = (S - p + 3) / 4; first = second ? g * c1 + p : p * c1 + g; stride = second ? 4 : 4 * c1; }
    }
    FpMsm<P> acc = te_identity_quad<P>(q);
    for (uint32_t i = 0; i < count; i++) acc = te_add_quad<P>(acc, quad_load<P>(reinterpret_cast<const uint32_t *>(src + first + (size_t)i * stride), q), q);
    quad_store<P>(pt + quad * PT_WORDS, q, acc);
    if (quad == 0) quad_store<P>(pt + RT_IDENT * PT_WORDS, q, te_identity_quad<P>(q));
    __syncthreads();
    // ---- steps over the LDS slots
    const int nsteps = plain ? 6 : 11;
    for (int step = 0; step < nsteps; step++) {
        bool active = false;
        uint32_t dst = 0, a = 0, b = 0;
        if (plain) { const uint32_t half = (RT_QUADS / 2) >> step; active = quad < half; dst = a = quad; b = quad + half; }
        else if (step == 0) { active = (quad & 3) < 2; dst = a = quad; b = quad + 2; }                         // group sums: shares 0 + 2, 1 + 3
        else if (step == 1) { active = (quad & 3) == 0; dst = a = quad; b = quad + 1; }                        // ... group g's sum is slot 4 g
        else if (step == 2) { active = quad < 16; const uint32_t pl = quad >> 2, j = quad & 3; dst = RT_PLANES + 8 * pl + j; a = 4 * rt_plane_member(pl, 2 * j); b = 4 * rt_plane_member(pl, 2 * j + 1); }
        else if (step == 3) { active = quad < 8; const uint32_t pl = quad >> 1, j = quad & 1; dst = a = RT_PLANES + 8 * pl + j; b = a + 2; }
        else if (step == 4) { active = quad < 4; dst = a = RT_PLANES + 8 * quad; b = a + 1; }
        else { active = quad == 0; dst = a = RT_PLANES + 24; const int h = step - 5; b = (h & 1) ? RT_PLANES + 8 * (2 - h / 2) : a; }      // Horner from plane 3 down: double, add plane 2, double, add plane 1, ...
        if (active) quad_store<P>(pt + dst * PT_WORDS, q, te_add_quad<P>(quad_load<P>(pt + a * PT_WORDS, q), quad_load<P>(pt + b * PT_WORDS, q), q));
        __syncthreads();
    }
    const uint32_t *res = pt + (plain ? 0 : RT_PLANES + 24) * PT_WORDS;
    if (to_host) { if (threadIdx.x == 0) out[(size_t)set * RF_OUT + term] = te_to_std_point<P>(*reinterpret_cast<const AccTE<P> *>(res)); }
    else if (quad == 0) quad_store<P>(reinterpret_cast<uint32_t *>(part + (size_t)set * RF_OUT + term), q, quad_load<P>(res, q));
}
// the weights of the six terms applied on the device (what reduce_host_tail does on the host): one quad per set runs a short program of unified additions over six LDS slots
// -- 2^k X as k self-additions -- and converts the sum to the Weierstrass XYZZ form.  Rare paths only: ~30 dependent quad operations (~0.1 ms).
template <class P>
__global__ void __launch_bounds__(64) k_reduce_combine(const AccTE<P> *__restrict__ part, int lgR, int lgC, XYZZ<Fp<P>> *__restrict__ out, XYZZ<Fp<P>> *__restrict__ dev_total) {
    __shared__ uint32_t pt[RF_OUT * PT_WORDS];
    __shared__ uint8_t prog[64][3];
    __shared__ int nprog;
    const uint32_t set = blockIdx.x, quad = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    if (quad < RF_OUT) quad_store<P>(pt + quad * PT_WORDS, q, quad_load<P>(reinterpret_cast<const uint32_t *>(part + (size_t)set * RF_OUT + quad), q));
    if (threadIdx.x == 0) {
        int n = 0;
        auto op = [&](int d, int a, int b) { prog[n][0] = (uint8_t)d; prog[n][1] = (uint8_t)a; prog[n][2] = (uint8_t)b; n++; };
        const int lcC = quad_weighted_lc(1u << lgC), lcR = quad_weighted_lc(1u << lgR);
        for (int i = 0; i < lcR; i++) op(5, 5, 5);
        op(5, 5, 4);                                     // row = 2^lcR [5] + [4]
        for (int i = 0; i < lcC; i++) op(3, 3, 3);
        op(3, 3, 2);                                     // col = 2^lcC [3] + [2]
        for (int i = 0; i < lgR; i++) op(0, 0, 0);
        op(0, 0, 5);                                     // 2^lgR [0] + row
        for (int i = 0; i < lgC; i++) op(0, 0, 0);
        op(0, 0, 3); op(0, 0, 1);                        // ... 2^lgC (.) + col + [1]
        nprog = n;
    }
    __syncthreads();
    const int n = nprog;
    for (int i = 0; i < n; i++) {
        // (every quad of the wave runs the same program on the same slots; quad 0 stores)
        const FpMsm<P> v = te_add_quad<P>(quad_load<P>(pt + prog[i][1] * PT_WORDS, q), quad_load<P>(pt + prog[i][2] * PT_WORDS, q), q);
        __syncthreads();
        if (quad == 0) quad_store<P>(pt + prog[i][0] * PT_WORDS, q, v);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const auto r = te_to_std_point<P>(*reinterpret_cast<const AccTE<P> *>(pt));
        out[set] = r;
        if (dev_total) dev_total[set] = r;
    }
}
// the host's share of the Edwards reduction (k_reduce_terms with to_host): the weighted sum of a set's six terms, Horner over the weights' shifts
template <class Fq>
static XYZZ<Fq> reduce_host_tail(const XYZZ<Fq> *o, int lgR, int lgC) {
    auto shl = [](XYZZ<Fq> v, int k) { for (int i = 0; i < k; i++) v = v.dbl(); return v; };
    const int lcC = quad_weighted_lc(1u << lgC), lcR = quad_weighted_lc(1u << lgR);
    XYZZ<Fq> row = shl(o[5], lcR);                  // the row term: (V1 + 2^lc V2) 2^lgC
    row.add(o[4]);
    XYZZ<Fq> col = shl(o[3], lcC);
    col.add(o[2]);
    // 2^(lgR + lgC) [0] + 2^lgC row + col + [1]  =  2^lgC (2^lgR [0] + row) + col + [1]
    XYZZ<Fq> t = shl(o[0], lgR);
    t.add(row);
    t = shl(t, lgC);
    t.add(col);
    t.add(o[1]);
    return t;
}

template <class A> __global__ void k_sum_tree(const A *__restrict__ in, uint32_t total, uint32_t per, A *__restrict__ out, uint32_t in_stride, uint32_t out_stride);

constexpr uint32_t DEFERRED_CAP = 1u << 20;
constexpr size_t ACC_BYTES = sizeof(Acc28<Fq377P>);          // one bucket in the reduced-radix form: 224 B, XYZZ and extended Edwards alike, same for both curves
static_assert(sizeof(Acc28<Fq381P>) == ACC_BYTES, "bucket size differs between the curves");
static_assert(sizeof(AccTE<Fq377P>) == ACC_BYTES, "the Edwards accumulator must fit the XYZZ bucket slots");
constexpr int MAX_WSUMS = 64;                                // window sums per MSM (per-window mode: <= 37 windows)
struct MsmWorkspace {
    size_t cap_pairs = 0, cap_buckets = 0, cap_tmp = 0, cap_result = 0;
    uint32_t *keys_a = nullptr, *keys_b = nullptr, *vals_a = nullptr, *vals_b = nullptr, *start = nullptr, *end = nullptr;
    uint32_t *sorted_keys = nullptr, *sorted_vals = nullptr;      // whichever half of the double buffers the radix sort finished in
    uint32_t *order = nullptr, *ovf_slot = nullptr;               // per bucket: visiting order, slot in the overflow list (NO_SLOT for all but oversized buckets)
    uint32_t *ovf_bucket = nullptr, *ovf_nseg = nullptr, *ovf_off = nullptr; void *ovf_partial = nullptr; size_t cap_ovf = 0;    // overflow list + the segments' partial sums
    uint32_t *part_hist = nullptr, *part_offs = nullptr; size_t cap_part = 0;      // two-level partition: (coarse bin, workgroup) counts and their scan
    uint32_t *canon = nullptr; size_t cap_canon = 0;                               // two-level partition: the scalars as canonical integers (8 words each), between its two passes
    uint32_t *ord_hist = nullptr, *ord_offs = nullptr;            // ORD_BINS x ORD_MAX_BLOCKS counts and their scan
    uint32_t *ctrl = nullptr;                                     // 8 control words (see k_order_hist); armed at zero between MSMs
    bool ctrl_dirty = false;                                      // an exception left the order pass half done: re-arm ctrl before the next one
    size_t plan_n = 0, plan_pairs = 0; int plan_c = 0, plan_nwin = 0;     // state between msm_prepare and msm_finish
    bool plan_table = false; uint32_t plan_cap = BUCKET_CAP;
    uint32_t *deferred = nullptr, *deferred_count = nullptr;
    void *buckets = nullptr, *seg_s = nullptr, *seg_w = nullptr, *partial = nullptr, *tmp = nullptr;
    void *h_res = nullptr, *d_res = nullptr;                      // pinned host memory the last reduction kernel writes the window sums (+ flags) into, and its device address
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};      // [2 rep], [2 rep + 1]: around the k_accumulate launch of base array `rep`
};
constexpr size_t RES_BYTES = 192 * MAX_WSUMS * 6 + 64;        // six terms per set when the Edwards reduction ends on the host (k_reduce_terms RF_OUT)
// the accumulators and the reduction's levels, sized by the number of buckets ACCUMULATED (twice the prepared ones when one prepared state serves two base arrays at once)
static void ensure_result(MsmWorkspace &S, size_t buckets) {
    if (buckets <= S.cap_result) return;
    dfree(S.buckets); dfree(S.partial); dfree(S.seg_s); dfree(S.seg_w);
    S.cap_result = buckets;
    S.buckets = dmalloc(buckets * ACC_BYTES);
    S.seg_s = dmalloc((buckets / RED_L1 + 64) * ACC_BYTES); S.seg_w = dmalloc((buckets / RED_L1 + 64) * ACC_BYTES);
    S.partial = dmalloc((2 * (buckets / (RED_L1 * RED_L2)) + 2 * 64 * 64) * ACC_BYTES);
}
static void ensure_scratch(MsmWorkspace &S, size_t pairs, size_t buckets, size_t cap) {
    if (cap == 0) cap = BUCKET_CAP;
    if (!S.ev[0]) {
        for (auto &e : S.ev) HIP_CHECK(hipEventCreate(&e));
        S.deferred = (uint32_t *)dmalloc(2 * DEFERRED_CAP * 4); S.deferred_count = (uint32_t *)dmalloc(8);
        HIP_CHECK(hipMemset(S.deferred_count, 0, 8));                 // [1] is the tail kernel's ticket: armed at zero, re-armed by the workgroup that takes the last one
        S.ord_hist = (uint32_t *)dmalloc((size_t)ORD_BINS * ORD_MAX_BLOCKS * 4); S.ord_offs = (uint32_t *)dmalloc((size_t)ORD_BINS * ORD_MAX_BLOCKS * 4);
        S.ctrl = (uint32_t *)dmalloc(32);
        HIP_CHECK(hipMemset(S.ctrl, 0, 32));
        HIP_CHECK(hipHostMalloc(&S.h_res, RES_BYTES, hipHostMallocMapped));
        HIP_CHECK(hipHostGetDevicePointer(&S.d_res, S.h_res, 0));
    }
    if (pairs / cap + 64 > S.cap_ovf) {
        dfree(S.ovf_partial); dfree(S.ovf_bucket); dfree(S.ovf_nseg); dfree(S.ovf_off);
        S.cap_ovf = pairs / cap + 64;
        S.ovf_partial = dmalloc(2 * S.cap_ovf * ACC_BYTES);          // (x 2: run_buckets with a second base array)
        S.ovf_bucket = (uint32_t *)dmalloc(S.cap_ovf * 4); S.ovf_nseg = (uint32_t *)dmalloc(S.cap_ovf * 4); S.ovf_off = (uint32_t *)dmalloc((S.cap_ovf + 1) * 4);
    }
    if (pairs > S.cap_pairs) {
        dfree(S.keys_a); dfree(S.keys_b); dfree(S.vals_a); dfree(S.vals_b);
        S.cap_pairs = pairs;
        S.keys_a = (uint32_t *)dmalloc(pairs * 4); S.keys_b = (uint32_t *)dmalloc(pairs * 4);
        S.vals_a = (uint32_t *)dmalloc(pairs * 4); S.vals_b = (uint32_t *)dmalloc(pairs * 4);
    }
    if (buckets > S.cap_buckets) {
        dfree(S.start); dfree(S.end); dfree(S.order); dfree(S.ovf_slot);
        S.cap_buckets = buckets;
        S.start = (uint32_t *)dmalloc(buckets * 4); S.end = (uint32_t *)dmalloc(buckets * 4);
        S.order = (uint32_t *)dmalloc(buckets * 4); S.ovf_slot = (uint32_t *)dmalloc(buckets * 4);
    }
    ensure_result(S, buckets);
}
MsmWorkspace *msm_workspace_create() { return new MsmWorkspace(); }
void msm_workspace_destroy(MsmWorkspace *w) {
    if (!w) return;
    for (void *p : {(void *)w->keys_a, (void *)w->keys_b, (void *)w->vals_a, (void *)w->vals_b, (void *)w->start, (void *)w->end, (void *)w->order, (void *)w->ovf_slot,
                    (void *)w->ovf_bucket, (void *)w->ovf_nseg, (void *)w->ovf_off, w->ovf_partial, (void *)w->part_hist, (void *)w->part_offs, (void *)w->canon, (void *)w->ord_hist, (void *)w->ord_offs,
                    (void *)w->ctrl, (void *)w->deferred, (void *)w->deferred_count, w->buckets, w->seg_s, w->seg_w, w->partial, w->tmp}) dfree(p);
    if (w->h_res) (void)hipHostFree(w->h_res);
    for (auto e : w->ev) if (e) (void)hipEventDestroy(e);
    delete w;
}

// size-balanced visiting order of the buckets + the overflow list: three small launches, no memset, no generic sort (k_order_* above).  The same path serves lone calls
// and multi-proof calls: positions are deterministic (buckets of one size keep the order of their index ranges), which the generic stable sort used to provide at ~16 launches.
static void order_buckets(MsmWorkspace &S, size_t nb, uint32_t cap, hipStream_t s) {
    if (S.ctrl_dirty) HIP_CHECK(hipMemsetAsync(S.ctrl, 0, 8, s));
    S.ctrl_dirty = true;
    unsigned nblk = (unsigned)((nb + 4095) / 4096);
    if (nblk > (unsigned)ORD_MAX_BLOCKS) nblk = ORD_MAX_BLOCKS;
    if (nblk < 1) nblk = 1;
    const uint32_t per_block = (uint32_t)((nb + nblk - 1) / nblk);
    hipLaunchKernelGGL(k_order_hist, dim3(nblk), dim3(ORD_THREADS), 0, s, S.start, S.end, (uint32_t)nb, per_block, cap, S.ovf_slot, S.ovf_bucket, S.ovf_nseg, (uint32_t)S.cap_ovf - 1, S.ctrl, S.ord_hist);
    HIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORD_BINS), 0, s, S.ord_hist, S.ord_offs, nblk, S.ctrl, S.ovf_nseg, S.ovf_off, (uint32_t)S.cap_ovf - 1);
    HIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_order_scatter, dim3(nblk), dim3(ORD_THREADS), 0, s, S.start, S.end, (uint32_t)nb, per_block, S.ord_offs, S.order);
    HIP_LAUNCH_CHECK();
    S.ctrl_dirty = false;
}

// shared middle: sort the (key, value) pairs, find bucket ranges, order buckets by descending size, list oversized buckets
template <class P>
static void prepare_buckets(MsmWorkspace &S, size_t pairs, int c, int nsets, int sort_bits, uint32_t cap, hipStream_t s, int begin_bit = 0) {
    // c here = log2(buckets per set); every key is a valid bucket index in [0, nsets << c)
    size_t nb = (size_t)nsets << c;
    int key_bits = 1;
    while (((size_t)1 << key_bits) < nb) key_bits++;
    if (sort_bits > 0 && sort_bits < key_bits) key_bits = sort_bits;       // stable sort on the bucket bits only (window-major input)
    size_t tmp_bytes = 0;
    // ping-pong sort: the result stays in whichever buffer the last radix pass wrote (no copy back)
    rocprim::double_buffer<uint32_t> dk(S.keys_a, S.keys_b), dv(S.vals_a, S.vals_b);
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, pairs, (unsigned)begin_bit, (unsigned)key_bits, s));
    if (tmp_bytes > S.cap_tmp) { dfree(S.tmp); S.tmp = dmalloc(tmp_bytes); S.cap_tmp = tmp_bytes; }
    HIP_CHECK(rocprim::radix_sort_pairs(S.tmp, tmp_bytes, dk, dv, pairs, (unsigned)begin_bit, (unsigned)key_bits, s));
#ifdef ZKAES_MEASURE
    if (knockin() & 1) HIP_CHECK(rocprim::radix_sort_pairs(S.tmp, tmp_bytes, dk, dv, pairs, (unsigned)begin_bit, (unsigned)key_bits, s));
#endif
    S.sorted_keys = dk.current(); S.sorted_vals = dv.current();
    hipLaunchKernelGGL(k_bounds, dim3((unsigned)(((pairs + 3) / 4 + 255) / 256)), dim3(256), 0, s, S.sorted_keys, pairs, c, (uint32_t)nsets, S.start, S.end);
    HIP_LAUNCH_CHECK();
    order_buckets(S, nb, cap, s);
}
// shared tail over prepared buckets: accumulate from `bases`, fold overflow segments and deferred degenerate additions, reduce; returns the
// nsets window sums.  May be called several times on one prepared state with different base arrays (same scalars, e.g. plain + shifted powers).
template <class Law>
static std::vector<XYZZ<Fp<typename Law::Params>>> run_buckets(MsmWorkspace &S, const typename Law::Base *bases, size_t pairs, int c, int nsets_one, size_t n_points, uint32_t cap, hipStream_t s, float *acc_ms,
                                            XYZZ<Fp<typename Law::Params>> *dev_wsum_out = nullptr, const typename Law::Base *bases2 = nullptr) {
    using P = typename Law::Params;
    using A = typename Law::Acc;
    using Fq = Fp<P>;
    // `bases2` (Edwards law): the same prepared scalars against a second base array (plain + shifted powers of a degree-bounded commitment).  Both accumulations are
    // launched back to back into consecutive bucket sets and ONE reduction (and one host wait) serves both: the returned sums are [sets of bases..., sets of bases2...]
    const int nrep = bases2 ? 2 : 1, nsets = nsets_one * nrep;
    if (bases2 && !Law::edwards) throw GpuError("msm: two base arrays per prepared state need the Edwards law");
    if (nsets > MAX_WSUMS) throw GpuError("msm: too many windows");
    const size_t nb = (size_t)nsets_one << c;
    ensure_result(S, nb * nrep);
    if constexpr (!Law::edwards) HIP_CHECK(hipMemsetAsync(S.deferred_count, 0, 8, s));        // [0] deferred pairs, [1] the tail kernel's workgroup ticket (the Edwards law defers nothing)
#ifdef ZKAES_MEASURE
    if (knockin() & 8) {        // (measurement builds only) one extra, untimed launch
        hipLaunchKernelGGL((k_accumulate<Law>), dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, s, bases, S.sorted_vals, S.start, S.end, S.order, (uint32_t)nb, cap,
                           (A *)S.buckets, S.deferred, DEFERRED_CAP, S.deferred_count);
        if constexpr (!Law::edwards) HIP_CHECK(hipMemsetAsync(S.deferred_count, 0, 8, s));
    }
#endif
    uint32_t max_seg = (uint32_t)(pairs / cap + 1);
    // (A side stream of its own for k_accumulate -- at lower priority than, or beside higher-priority streams for, the short kernels of a lone proof's other lanes -- was
    // measured in round 5 and dropped: the 64-byte lone call went from 74 to 105 ms, profiles/r05_lone_latency.md.)
    for (int rep = 0; rep < nrep; rep++) {
        const typename Law::Base *src = rep ? bases2 : bases;
        HIP_CHECK(hipEventRecord(S.ev[2 * rep], s));          // every k_accumulate launch is bracketed and booked by itself (msm_stats: launches, points, pairs, ms)
        hipLaunchKernelGGL((k_accumulate<Law>), dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, s, src, S.sorted_vals, S.start, S.end, S.order, (uint32_t)nb, cap,
                           (A *)S.buckets + rep * nb, S.deferred, DEFERRED_CAP, S.deferred_count);
        HIP_LAUNCH_CHECK();
        HIP_CHECK(hipEventRecord(S.ev[2 * rep + 1], s));
        // oversized buckets + deferred degenerate additions (none for uniformly distributed digits: every lane exits at once)
        hipLaunchKernelGGL((k_accumulate_tail<Law>), dim3((max_seg + 63) / 64), dim3(64), 0, s, src, S.sorted_vals, S.start, S.end, S.ctrl, S.ovf_bucket, S.ovf_off, max_seg, cap,
                           (A *)S.ovf_partial + rep * S.cap_ovf, (A *)S.buckets + rep * nb, S.deferred, DEFERRED_CAP, S.deferred_count);
        HIP_LAUNCH_CHECK();
    }
    if constexpr (!Law::edwards) for (int rep = 0; rep < nrep; rep++) {
        // overflow partials -> their buckets (overflow list entries <= buckets with more than `cap` pairs <= pairs / cap); the Edwards law's tail kernel does it itself
        hipLaunchKernelGGL((k_fold_overflow<A>), dim3((max_seg + 63) / 64), dim3(64), 0, s, (A *)S.buckets + rep * nb, S.ctrl, S.ovf_bucket, S.ovf_off, max_seg, (const A *)S.ovf_partial + rep * S.cap_ovf);
        HIP_LAUNCH_CHECK();
    }
    XYZZ<Fq> *res = (XYZZ<Fq> *)S.d_res;
    uint32_t segs = (1u << c) / RED_L1, groups = (segs + RED_L2 - 1) / RED_L2;
    [[maybe_unused]] int te_lgR = 0, te_lgC = 0;
    [[maybe_unused]] bool te_host_tail = false;
#ifdef ZKAES_MEASURE
    for (int rep = (knockin() & 2) ? 0 : 1; rep < 2; rep++)
#endif
    {
    if constexpr (Law::edwards)
        hipLaunchKernelGGL((k_reduce_l1_pair<A>), dim3((unsigned)((2 * segs * nsets + 63) / 64)), dim3(64), 0, s, (const A *)S.buckets, c, nsets, (A *)S.seg_s, (A *)S.seg_w);
    else
        hipLaunchKernelGGL((k_reduce_l1<A>), dim3((unsigned)((segs * nsets + 63) / 64)), dim3(64), 0, s, (const A *)S.buckets, c, nsets, (A *)S.seg_s, (A *)S.seg_w);
    HIP_LAUNCH_CHECK();
    if constexpr (Law::edwards) {
        // Edwards law: plain row / column sums of the segment sums, then six small quad-cooperative workgroups per set (k_reduce_rc / k_reduce_terms above)
        const int lgG = c - 3 > 0 ? c - 3 : 0, lgC = (lgG + 1) / 2, lgR = lgG - lgC;
        if (lgC > 8) throw GpuError("msm: more than 2^19 buckets per set");
        const unsigned jobs = 2u * (1u << lgR) + (1u << lgC);
        A *rc = (A *)S.partial, *part = rc + (size_t)nsets * jobs;
        hipLaunchKernelGGL((k_reduce_rc<P>), dim3((unsigned)nsets * jobs), dim3(RQ_THREADS), 0, s, (const A *)S.seg_s, (const A *)S.seg_w, lgR, lgC, rc);
        HIP_LAUNCH_CHECK();
        te_host_tail = !dev_wsum_out && nsets <= 4;
        hipLaunchKernelGGL((k_reduce_terms<P>), dim3((unsigned)(RF_OUT * nsets)), dim3(RT_THREADS), 0, s, (const A *)rc, lgR, lgC, part, res, te_host_tail);
        HIP_LAUNCH_CHECK();
        if (!te_host_tail) {
            hipLaunchKernelGGL((k_reduce_combine<P>), dim3((unsigned)nsets), dim3(64), 0, s, (const A *)part, lgR, lgC, res, dev_wsum_out);
            HIP_LAUNCH_CHECK();
        }
        te_lgR = lgR; te_lgC = lgC;
    } else {
    hipLaunchKernelGGL((k_reduce_l2<A>), dim3(2u * (unsigned)((groups * nsets + 63) / 64)), dim3(64), 0, s, (const A *)S.seg_s, (const A *)S.seg_w, c, nsets, (A *)S.partial);
    HIP_LAUNCH_CHECK();
    const uint32_t parts = 2 * groups;                 // two partials per group (k_reduce_l2's two roles)
    if (nsets == 1 && parts > 2048) {
        // one big bucket set: 256-partial blocks first, so the final LDS tree does not walk tens of thousands of partials serially
        uint32_t mid = (parts + 255) / 256;
        hipLaunchKernelGGL((k_sum_tree<A>), dim3(mid), dim3(256), 0, s, (const A *)S.partial, parts, 256u, (A *)S.seg_s, 0u, 0u);
        HIP_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_reduce_window<A>), dim3(1), dim3(256), 0, s, (const A *)S.seg_s, mid, res, dev_wsum_out);
    } else {
        hipLaunchKernelGGL((k_reduce_window<A>), dim3((unsigned)nsets), dim3(256), 0, s, (const A *)S.partial, parts, res, dev_wsum_out);
    }
    HIP_LAUNCH_CHECK();
    }
    }
    std::vector<XYZZ<Fq>> ws(nsets);
    uint32_t n_deferred = 0;
    sync((stream_t)s);        // (sleeps in throughput mode) the window sums are in pinned host memory once the stream has drained: no copy launch
    if (Law::edwards && te_host_tail) {       // host tail of the Edwards reduction: six unweighted terms per set
        std::vector<XYZZ<Fq>> terms((size_t)nsets * RF_OUT);
        memcpy(terms.data(), S.h_res, sizeof(XYZZ<Fq>) * terms.size());
        for (int i = 0; i < nsets; i++) ws[i] = reduce_host_tail<Fq>(terms.data() + (size_t)i * RF_OUT, te_lgR, te_lgC);
    } else memcpy(ws.data(), S.h_res, sizeof(XYZZ<Fq>) * nsets);
    if constexpr (!Law::edwards) {
        HIP_CHECK(hipMemcpyAsync(&n_deferred, S.deferred_count, 4, hipMemcpyDeviceToHost, s));
        sync((stream_t)s);
        if (n_deferred > DEFERRED_CAP) throw GpuError("msm: more than 2^20 degenerate additions (repeated base points); refusing to return a wrong sum");
    }
    for (int rep = 0; rep < nrep; rep++) HIP_CHECK(hipEventElapsedTime(acc_ms + rep, S.ev[2 * rep], S.ev[2 * rep + 1]));
    (void)n_points;
    return ws;
}
// one entry per k_accumulate LAUNCH: a prepared state finished against two base arrays (plain + shifted powers) is two launches over n points each
static void add_stats(const float *acc_ms, int launches, size_t n, size_t pairs, std::chrono::steady_clock::time_point t_begin) {
    std::lock_guard<std::mutex> g(g_stats_mu);
    for (int i = 0; i < launches; i++) g_stats.accumulate_ms += acc_ms[i];
    g_stats.points += n * (size_t)launches;
    g_stats.pairs += pairs * (size_t)launches;
    g_stats.launches += (uint64_t)launches;
    g_stats.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
}

// Weierstrass affine (library-wide form) -> precomputed Edwards form (y - x, y + x, 2 d x y), 8 points per lane sharing ONE field inversion
// (Montgomery's trick) for the two divisions of the map.  A point of order 2 or 4 (den = 0; never in the prime-order subgroup) raises *bad.
__global__ void __launch_bounds__(64) k_convert_bases_te(const Affine<Fq377> *__restrict__ src, Niels28<Fq377P> *__restrict__ dst, size_t n, uint32_t *__restrict__ bad) {
    constexpr int B = 8;
    size_t s0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * B;
    if (s0 >= n) return;
    size_t e = s0 + B < n ? s0 + B : n;
    TeMapParts m[B];
    Fq377 pre[B];
    bool skip[B];
    Fq377 acc = Fq377::one();
    for (size_t i = s0; i < e; i++) {
        Affine<Fq377> p = src[i];
        int q = (int)(i - s0);
        skip[q] = p.is_inf();
        if (!skip[q]) {
            m[q] = te_map_parts(p);
            if (m[q].den.is_zero()) { skip[q] = true; atomicOr(bad, 1u); }
        }
        pre[q] = acc;
        if (!skip[q]) acc = acc * m[q].den;
    }
    acc = acc.inverse();
    for (size_t i = e; i-- > s0;) {
        int q = (int)(i - s0);
        if (skip[q]) { dst[i] = niels_identity<Fq377P>(); continue; }
        Fq377 inv = acc * pre[q];
        acc = acc * m[q].den;
        dst[i] = te_niels_finish(m[q], inv);
    }
}
template <class Curve>
void convert_bases_te(Niels28<typename Curve::FqP> *dst, const Affine<typename Curve::Fq> *src, size_t n, stream_t s_) {
    static_assert(Curve::ID == 377, "only BLS12-377's G1 has a twisted Edwards model");
    if (!n) return;
    hipStream_t s = (hipStream_t)s_;
    DevPtr<uint32_t> d_bad(1);
    uint32_t h_bad = 0;
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, s));
    size_t lanes = (n + 7) / 8;
    hipLaunchKernelGGL(k_convert_bases_te, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, s, src, dst, n, d_bad.get());
    HIP_LAUNCH_CHECK();
    sync((stream_t)s);        // drain first (sleeps in throughput mode): the pageable device-to-host copies below wait actively inside HIP
    HIP_CHECK(hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, s));
    sync((stream_t)s);
    if (h_bad) throw GpuError("convert_bases_te: a base point has order 2 or 4 -- the Edwards path needs points of the prime-order subgroup");
}
template void convert_bases_te<Bls377>(Niels28<Fq377P> *, const Affine<Fq377> *, size_t, stream_t);

template <class Curve>
void convert_bases(Affine28<typename Curve::FqP> *dst, const Affine<typename Curve::Fq> *src, size_t n, stream_t s_) {
    if (!n) return;
    hipLaunchKernelGGL((k_convert_bases<typename Curve::FqP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s_, src, dst, n);
    HIP_LAUNCH_CHECK();
}

// ---- signed-digit Pippenger as two steps, so that one digit/sort pass can serve several base arrays:
//   msm_prepare : window digits of up to two scalar vectors (the second one naming bases `val_off2` further on), sort, bucket ranges
//   msm_finish  : accumulate from `bases` + reduce + Horner on the host
template <class Curve>
void msm_prepare(MsmWorkspace *ws_, const typename Curve::Fr *scal1, size_t n1, const typename Curve::Fr *scal2, size_t n2, size_t val_off2, stream_t s_, int force_c) {
    using Fr = typename Curve::Fr;
    hipStream_t s = (hipStream_t)s_;
    if (!ws_) throw GpuError("msm: null workspace");
    MsmWorkspace &S = *ws_;
    size_t n = n1 + n2;
    S.plan_n = n;
    if (n == 0) return;
    if (n >= (1u << 30) || val_off2 + n2 >= (1u << 30)) throw GpuError("msm: too many points");
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    int c = lg - 2;                               // signed digits: 2^(c-1) buckets per window
    if (c < 7) c = 7;
    if (c > 17) c = 17;
    if (force_c >= 4 && force_c <= 22) c = force_c;     // ranks sharing one MSM by point range must cut the scalars into the same windows
    const int nwin = (Fr::BITS + 1 + c - 1) / c;  // one extra bit for the recoding carry
    size_t pairs = n * (size_t)nwin;
    if (pairs >= ((size_t)1 << 31)) throw GpuError("msm: n x windows exceeds the 2^31 pairs the sort indexes with int");
    const size_t nb = (size_t)nwin << (c - 1);
    S.plan_c = c; S.plan_nwin = nwin; S.plan_pairs = pairs; S.plan_table = false; S.plan_cap = BUCKET_CAP;
    ensure_scratch(S, pairs, nb, BUCKET_CAP);
    if (n1) { hipLaunchKernelGGL((k_digits<Fr>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, scal1, (uint32_t)n1, 0u, (uint32_t)n, 0u, c, nwin, (uint32_t)nb, S.keys_a, S.vals_a); HIP_LAUNCH_CHECK(); }
    if (n2) { hipLaunchKernelGGL((k_digits<Fr>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, scal2, (uint32_t)n2, (uint32_t)n1, (uint32_t)n, (uint32_t)val_off2, c, nwin, (uint32_t)nb, S.keys_a, S.vals_a); HIP_LAUNCH_CHECK(); }
    prepare_buckets<typename Curve::FqP>(S, pairs, c - 1, nwin, c - 1, BUCKET_CAP, s);
}
template <class Curve, class Law>
static XYZZ<typename Curve::Fq> msm_finish_impl(MsmWorkspace *ws_, const typename Law::Base *bases, stream_t s_, const typename Law::Base *bases2 = nullptr,
                                                XYZZ<typename Curve::Fq> *out2 = nullptr) {
    using Fq = typename Curve::Fq;
    static_assert(sizeof(XYZZ<Fq>) == 192, "XYZZ layout");
    hipStream_t s = (hipStream_t)s_;
    if (!ws_) throw GpuError("msm: null workspace");
    MsmWorkspace &S = *ws_;
    if (out2) *out2 = XYZZ<Fq>::inf();
    if (S.plan_n == 0) return XYZZ<Fq>::inf();
    auto t_begin = std::chrono::steady_clock::now();
    const int c = S.plan_c, nwin = S.plan_nwin;
    oplog_msm(S.plan_n, OP_MSM_BUCKETS);
    if (bases2) oplog_msm(S.plan_n, OP_MSM_SECOND_BASES);
    float ms[2] = {0, 0};
    if (S.plan_table) {      // `bases` = table copy 0 (+ a constant index shift): the window weights live in the copies, ONE bucket set, no Horner
        std::vector<XYZZ<Fq>> one = run_buckets<Law>(S, bases, S.plan_pairs, c - 1, 1, S.plan_n, S.plan_cap, s, ms, nullptr, bases2);
        add_stats(ms, bases2 ? 2 : 1, S.plan_n, S.plan_pairs, t_begin);
        if (bases2) *out2 = one[1];
        return one[0];
    }
    if (bases2 && 2 * nwin > MAX_WSUMS) {      // (more window sums than one reduction returns: one after the other)
        *out2 = msm_finish_impl<Curve, Law>(ws_, bases2, s_);
        bases2 = nullptr;
    }
    std::vector<XYZZ<Fq>> ws = run_buckets<Law>(S, bases, S.plan_pairs, c - 1, nwin, S.plan_n, S.plan_cap, s, ms, nullptr, bases2);
    static const bool msm_debug = getenv("ZKAES_MSM_DEBUG") != nullptr;       // read once per process
    if (msm_debug) {
        for (int w = 0; w < nwin; w++) {
            Affine<Fq> a = ws[w].to_affine();
            fprintf(stderr, "window %d inf=%d x0=%08x y0=%08x\n", w, (int)a.is_inf(), a.x.l[0], a.y.l[0]);
        }
    }
    XYZZ<Fq> total[2];
    for (int rep = 0; rep < (bases2 ? 2 : 1); rep++) {
        XYZZ<Fq> t = XYZZ<Fq>::inf();
        for (int w = nwin - 1; w >= 1; w--) {
            t.add(ws[rep * nwin + w]);
            for (int k = 0; k < c; k++) t = t.dbl();
        }
        t.add(ws[rep * nwin]);
        total[rep] = t;
    }
    if (bases2) *out2 = total[1];
    add_stats(ms, bases2 ? 2 : 1, S.plan_n, S.plan_pairs, t_begin);
    return total[0];
}
// plain + shifted powers of one degree-bounded commitment: out[0] = sum over `bases`, out[1] = sum over `bases2`, one reduction and one host wait for both
template <class Curve>
void msm_finish2(MsmWorkspace *ws_, const Niels28<typename Curve::FqP> *bases, const Niels28<typename Curve::FqP> *bases2, XYZZ<typename Curve::Fq> *out, stream_t s_) {
    out[0] = msm_finish_impl<Curve, EdwardsLaw<typename Curve::FqP>>(ws_, bases, s_, bases2, out + 1);
}
template <class Curve>
XYZZ<typename Curve::Fq> msm_finish(MsmWorkspace *ws_, const Affine28<typename Curve::FqP> *bases, stream_t s_) { return msm_finish_impl<Curve, WeierLaw<typename Curve::FqP>>(ws_, bases, s_); }
template <class Curve>
XYZZ<typename Curve::Fq> msm(MsmWorkspace *ws_, const Affine28<typename Curve::FqP> *bases, const typename Curve::Fr *scalars, size_t n, stream_t s_) {
    if (n == 0) return XYZZ<typename Curve::Fq>::inf();
    msm_prepare<Curve>(ws_, scalars, n, nullptr, 0, 0, s_);
    return msm_finish<Curve>(ws_, bases, s_);
}
template <class Curve>
XYZZ<typename Curve::Fq> msm_finish(MsmWorkspace *ws_, const Niels28<typename Curve::FqP> *bases, stream_t s_) { return msm_finish_impl<Curve, EdwardsLaw<typename Curve::FqP>>(ws_, bases, s_); }
template <class Curve>
XYZZ<typename Curve::Fq> msm(MsmWorkspace *ws_, const Niels28<typename Curve::FqP> *bases, const typename Curve::Fr *scalars, size_t n, stream_t s_) {
    if (n == 0) return XYZZ<typename Curve::Fq>::inf();
    msm_prepare<Curve>(ws_, scalars, n, nullptr, 0, 0, s_);
    return msm_finish<Curve>(ws_, bases, s_);
}

// ---- ONE MSM sharded by point range over ranks (SURVEY.md 8e second row): every rank runs the buckets of its slice with the window plan of the WHOLE
// MSM and leaves its window sums in device memory; after the all-gather (RCCL, HBM to HBM) k_fold_ranks adds the ranks' sums per window.
template <class Curve>
void msm_sharded_plan(size_t n_total, int *c_out, int *nwin_out) {
    int lg = 0;
    while (((size_t)1 << lg) < n_total) lg++;
    int c = lg - 2;
    if (c < 7) c = 7;
    if (c > 17) c = 17;
    *c_out = c;
    *nwin_out = (Curve::Fr::BITS + 1 + c - 1) / c;
}
template <class Curve>
void msm_window_sums_device(MsmWorkspace *ws_, const Affine28<typename Curve::FqP> *bases, const typename Curve::Fr *scalars, size_t n_local, size_t n_total,
                            XYZZ<typename Curve::Fq> *dev_out, stream_t s_) {
    using Fq = typename Curve::Fq;
    hipStream_t s = (hipStream_t)s_;
    int c, nwin;
    msm_sharded_plan<Curve>(n_total, &c, &nwin);
    if (n_local == 0) {       // an empty share contributes the point at infinity in every window (zz = 0)
        HIP_CHECK(hipMemsetAsync(dev_out, 0, sizeof(XYZZ<Fq>) * nwin, s));
        sync((stream_t)s);
        return;
    }
    msm_prepare<Curve>(ws_, scalars, n_local, nullptr, 0, 0, s_, c);
    MsmWorkspace &S = *ws_;
    if (S.plan_nwin != nwin) throw GpuError("msm_window_sums_device: window plan mismatch");
    float ms = 0;
    auto t_begin = std::chrono::steady_clock::now();
    oplog_msm(n_local, OP_MSM_BUCKETS);
    run_buckets<WeierLaw<typename Curve::FqP>>(S, bases, S.plan_pairs, c - 1, nwin, n_local, S.plan_cap, s, &ms, dev_out);
    add_stats(&ms, 1, n_local, S.plan_pairs, t_begin);
}
template <class Fq>
__global__ void k_fold_ranks(const XYZZ<Fq> *__restrict__ in, int world, int nwin, XYZZ<Fq> *__restrict__ out) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwin) return;
    XYZZ<Fq> acc = in[w];
    for (int r = 1; r < world; r++) acc.add(in[(size_t)r * nwin + w]);
    out[w] = acc;
}
template <class Curve>
XYZZ<typename Curve::Fq> msm_fold_window_sums_device(const XYZZ<typename Curve::Fq> *dev_in, int world, size_t n_total, stream_t s_) {
    using Fq = typename Curve::Fq;
    hipStream_t s = (hipStream_t)s_;
    int c, nwin;
    msm_sharded_plan<Curve>(n_total, &c, &nwin);
    DevPtr<XYZZ<Fq>> d_out((size_t)nwin);
    hipLaunchKernelGGL((k_fold_ranks<Fq>), dim3(1), dim3(64), 0, s, dev_in, world, nwin, d_out.get());
    HIP_LAUNCH_CHECK();
    std::vector<XYZZ<Fq>> ws(nwin);
    sync((stream_t)s);        // drain first (sleeps in throughput mode): the pageable device-to-host copies below wait actively inside HIP
    HIP_CHECK(hipMemcpyAsync(ws.data(), d_out, sizeof(XYZZ<Fq>) * nwin, hipMemcpyDeviceToHost, s));
    sync((stream_t)s);
    XYZZ<Fq> total = XYZZ<Fq>::inf();
    for (int w = nwin - 1; w >= 1; w--) {
        total.add(ws[w]);
        for (int k = 0; k < c; k++) total = total.dbl();
    }
    total.add(ws[0]);
    return total;
}

// ---- the same sharding on the prover's own path: tables on the twisted Edwards model, ONE bucket set, hence ONE partial sum per rank (no Horner)
template <class Curve>
void msm_table_sum_device(MsmWorkspace *ws_, const Niels28<typename Curve::FqP> *tables, size_t stride, size_t off, int c, const typename Curve::Fr *scalars, size_t n,
                          XYZZ<typename Curve::Fq> *dev_out, stream_t s_) {
    using Fq = typename Curve::Fq;
    hipStream_t s = (hipStream_t)s_;
    if (n == 0) {             // an empty share contributes the point at infinity (zz = 0)
        HIP_CHECK(hipMemsetAsync(dev_out, 0, sizeof(XYZZ<Fq>), s));
        sync((stream_t)s);
        return;
    }
    msm_prepare_table<Curve>(ws_, scalars, n, off, nullptr, 0, 0, c, stride, s_);
    MsmWorkspace &S = *ws_;
    float ms = 0;
    auto t_begin = std::chrono::steady_clock::now();
    oplog_msm(n, OP_MSM_BUCKETS);
    run_buckets<EdwardsLaw<typename Curve::FqP>>(S, tables, S.plan_pairs, S.plan_c - 1, 1, n, S.plan_cap, s, &ms, dev_out);
    add_stats(&ms, 1, n, S.plan_pairs, t_begin);
}
template <class Curve>
XYZZ<typename Curve::Fq> msm_fold_points_device(const XYZZ<typename Curve::Fq> *dev_in, int world, stream_t s_) {
    using Fq = typename Curve::Fq;
    hipStream_t s = (hipStream_t)s_;
    DevPtr<XYZZ<Fq>> d_out(1);
    hipLaunchKernelGGL((k_fold_ranks<Fq>), dim3(1), dim3(64), 0, s, dev_in, world, 1, d_out.get());
    HIP_LAUNCH_CHECK();
    XYZZ<Fq> r;
    sync((stream_t)s);
    HIP_CHECK(hipMemcpyAsync(&r, d_out, sizeof r, hipMemcpyDeviceToHost, s));
    sync((stream_t)s);
    return r;
}

template <class Curve>
int table_windows(int c) { return table_layout(Curve::Fr::BITS + 1, c).nwin; }      // signed digits: one extra bit for the recoding carry

template <class Curve>
void build_window_tables(Affine<typename Curve::Fq> *tables, size_t stride, int c, stream_t s_) {
    using Fq = typename Curve::Fq;
    hipStream_t s = (hipStream_t)s_;
    const TableLayout L = table_layout(Curve::Fr::BITS + 1, c);
    for (int j = 1; j < L.nwin; j++) {          // copy j = 2^(width of window j-1) * copy j-1 = 2^(offset of window j) * copy 0
        uint32_t lanes = (uint32_t)((stride + 7) / 8);
        hipLaunchKernelGGL((k_table_next<Fq>), dim3((lanes + 63) / 64), dim3(64), 0, s, tables + (size_t)(j - 1) * stride, tables + (size_t)j * stride, (uint32_t)stride, L.width(j - 1));
        HIP_LAUNCH_CHECK();
    }
    sync((stream_t)s);
}

template <class Curve>
void table_next(Affine<typename Curve::Fq> *next, const Affine<typename Curve::Fq> *prev, size_t count, int c, int j, stream_t s_) {
    const TableLayout L = table_layout(Curve::Fr::BITS + 1, c);
    if (j < 1 || j >= L.nwin) throw GpuError("table_next: window index out of range");
    uint32_t lanes = (uint32_t)((count + 7) / 8);
    hipLaunchKernelGGL((k_table_next<typename Curve::Fq>), dim3((lanes + 63) / 64), dim3(64), 0, (hipStream_t)s_, prev, next, (uint32_t)count, L.width(j - 1));
    HIP_LAUNCH_CHECK();
}

// ---- Two-level bucket partition: the table path's grouping (digits -> bucket-contiguous value list + bucket ranges), written for INSTRUCTION count.
// Per-kernel VALU accounting of a proof (profiles/r04_valu_by_kernel_before_partition.md) shows the saturated prover is VALU-issue bound as a whole -- k_accumulate already issues at
// the pipe's limit, so every other kernel costs its instruction count -- and that the generic route (digit kernels that pre-split on 3 bits with wave ballots + two rocPRIM
// onesweep passes, whose stable ranking is a chain of match-any ballots per item) spent ~470 lane-instructions per (point, window) pair, 13 % of what the bucket
// additions themselves take.  Nothing here needs a stable sort -- only "bucket-contiguous, and inside a bucket by window" (the lanes of a wave then gather from the same
// table copy, which the round-3 partition got wrong: 5.7 % slower accumulation) -- so ranks come from LDS atomics, ~70 lane-instructions per pair:
//   k_part_hist    : one pass over the scalars: Montgomery -> canonical (stored: 8 words per scalar), window digits (carry recoding), LDS histogram of the pairs over the
//                    COARSE bins (the high bucket bits, <= 1024 bins); fixed tiling over a fixed grid, counts written per (bin, workgroup)
//   one scan       : every (bin, workgroup) gets its range -- no global atomics, deterministic
//   k_part_scatter : same tiling, digits re-derived from the canonical scalars (round 6; rounds 4-5 stored the 16 digit words per scalar): a tile's pairs are ranked by LDS atomics into bin-contiguous runs staged in LDS and written out as
//                    (value, 13-bit key = fine bucket bits x 16 + window)
//   k_part_fine    : one workgroup per coarse bin: LDS counts of its 8,192 (fine bucket, window) keys, in-place scan, bucket [start, end) ranges (what k_bounds used to
//                    find), values scattered to their final place through LDS cursors
// Zero digits are not emitted.  Skewed scalars (few distinct digits) only make some workgroups of the last pass long; oversized buckets still go through the overflow list.
constexpr int PART_FINE_BITS = 9, PART_FINE = 1 << PART_FINE_BITS, PART_WBITS = 4, PART_KEYS = PART_FINE << PART_WBITS;
constexpr int PART_THREADS = 512, PART_SPT = 2, PART_TILE = PART_THREADS * PART_SPT;      // scalars per tile
constexpr int PART_MAXW = 1 << PART_WBITS;                                                // windows per scalar (c_hi >= 16)
constexpr int PART_STAGE = PART_TILE * PART_MAXW;                                         // staged pairs per tile (LDS: 8 B each = 128 KB)
constexpr uint32_t PART_NBIN_MAX = 1024;                                                  // coarse bins (bucket bits <= 19, i.e. c_hi <= 20)
constexpr uint32_t PART_GRID = 512, PART_NONE = 0xffffffffu;
constexpr int PART_FINE_THREADS = 1024, PART_FINE_UNROLL = 8;     // the last pass is latency-bound (load -> LDS atomic -> LDS store): many lanes per bin, several loads in flight per lane

// signed window digits of one scalar from its canonical integer (raw[Fr::N] = 0 on entry): d[w] = (|digit| - 1) | neg << 31, PART_NONE for a zero digit or w >= nwin
template <int NW>
__device__ __forceinline__ void part_digits(const uint32_t (&raw)[NW + 1], const TableLayout &L, uint32_t d[PART_MAXW]) {
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < PART_MAXW; w++) {
        d[w] = PART_NONE;
        if (w < L.nwin) {
            const int c = L.width(w), bit = L.offset(w), limb = bit >> 5, sh = bit & 31;
            const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
            uint64_t two = limb <= NW - 1 ? ((uint64_t)raw[limb] | ((uint64_t)raw[limb + 1] << 32)) : 0;
            uint32_t v = ((uint32_t)(two >> sh) & mask) + carry;
            uint32_t neg = 0;
            carry = 0;
            if (v > half) { v = (1u << c) - v; neg = 1u << 31; carry = 1; }
            if (v) d[w] = (v - 1) | neg;
        }
    }
}
// block-wide exclusive scan of `len` LDS counters (len a multiple of the block size) IN PLACE; the total goes to *total_out (LDS).  Ends with a barrier.
template <int THREADS>
__device__ __forceinline__ void part_block_scan(uint32_t *cnt, uint32_t len, uint32_t *wave_sums, uint32_t *total_out) {
    const uint32_t per = len / THREADS, t = threadIdx.x, base = t * per;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; i++) sum += cnt[base + i];
    uint32_t incl = sum;                                     // inclusive scan across the 64 lanes of the wave
    for (int d = 1; d < 64; d <<= 1) { uint32_t v = __shfl_up(incl, d, 64); if ((int)(t & 63) >= d) incl += v; }
    if ((t & 63) == 63) wave_sums[t >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (uint32_t w = 0; w < (t >> 6); w++) run += wave_sums[w];
    for (uint32_t i = 0; i < per; i++) { const uint32_t c = cnt[base + i]; cnt[base + i] = run; run += c; }
    if (t == THREADS - 1) *total_out = run;
    __syncthreads();
}
template <class Fr>
__global__ void __launch_bounds__(PART_THREADS) k_part_hist(const Fr *__restrict__ s1, uint32_t n1, const Fr *__restrict__ s2, uint32_t n2, TableLayout L, uint32_t nbin,
                                                            uint32_t *__restrict__ canon, uint32_t *__restrict__ hist) {
    __shared__ uint32_t sh[PART_NBIN_MAX];
    for (uint32_t b = threadIdx.x; b < nbin; b += PART_THREADS) sh[b] = 0;
    __syncthreads();
    // every workgroup takes a CONTIGUOUS range of tiles: a bin's list then runs in (roughly) scalar-index order, which the last pass keeps inside each (bucket, window) group
    const uint32_t n = n1 + n2, ntiles = (n + PART_TILE - 1) / PART_TILE, per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint32_t tile_lo = blockIdx.x * per_wg, tile_hi = tile_lo + per_wg < ntiles ? tile_lo + per_wg : ntiles;
    for (uint32_t tile = tile_lo; tile < tile_hi; tile++)
        for (int q = 0; q < PART_SPT; q++) {
            const uint32_t g = tile * PART_TILE + q * PART_THREADS + threadIdx.x;
            if (g >= n) continue;
            uint32_t d[PART_MAXW], raw[Fr::N + 1];
            (g < n1 ? s1[g] : s2[g - n1]).to_raw(raw);              // Montgomery -> canonical: the expensive half of the recoding (~0.5 k instructions), done ONCE per scalar
            raw[Fr::N] = 0;
            part_digits<Fr::N>(raw, L, d);
#pragma unroll
            for (int w = 0; w < PART_MAXW; w++) if (d[w] != PART_NONE) atomicAdd(&sh[(d[w] & 0x7fffffffu) >> PART_FINE_BITS], 1u);
            static_assert(Fr::N == 8, "the canonical scalar is stored as two 16-byte words");
            uint4 *dst = reinterpret_cast<uint4 *>(canon + (size_t)g * Fr::N);
            dst[0] = make_uint4(raw[0], raw[1], raw[2], raw[3]); dst[1] = make_uint4(raw[4], raw[5], raw[6], raw[7]);
        }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbin; b += PART_THREADS) hist[(size_t)b * gridDim.x + blockIdx.x] = sh[b];      // bin-major: the scan gives every (bin, workgroup) its range
    if (blockIdx.x == 0 && threadIdx.x == 0) hist[(size_t)nbin * gridDim.x] = 0;                                       // the scan's last entry = the number of pairs
}
template <int NW>
__global__ void __launch_bounds__(PART_THREADS) k_part_scatter(const uint32_t *__restrict__ canon, uint32_t n, uint32_t n1, TableLayout L, uint32_t off1, uint32_t off2, uint32_t stride,
                                                               uint32_t nbin, const uint32_t *__restrict__ offs, uint32_t *__restrict__ out_val, uint16_t *__restrict__ out_key) {
    __shared__ uint32_t cursor[PART_NBIN_MAX], cnt[PART_NBIN_MAX], fill[PART_NBIN_MAX], st_val[PART_STAGE], st_key[PART_STAGE];
    __shared__ uint32_t wave_sums[PART_THREADS / 64], total;
    for (uint32_t b = threadIdx.x; b < nbin; b += PART_THREADS) cursor[b] = offs[(size_t)b * gridDim.x + blockIdx.x];
    const uint32_t ntiles = (n + PART_TILE - 1) / PART_TILE, per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint32_t tile_lo = blockIdx.x * per_wg, tile_hi = tile_lo + per_wg < ntiles ? tile_lo + per_wg : ntiles;
    for (uint32_t tile = tile_lo; tile < tile_hi; tile++) {
        for (uint32_t b = threadIdx.x; b < PART_NBIN_MAX; b += PART_THREADS) { cnt[b] = 0; fill[b] = 0; }
        __syncthreads();
        uint32_t d[PART_SPT][PART_MAXW];
#pragma unroll
        for (int q = 0; q < PART_SPT; q++) {
            const uint32_t g = tile * PART_TILE + q * PART_THREADS + threadIdx.x;
            // the digits are re-derived from the CANONICAL scalar k_part_hist left (8 words per scalar; rounds 4-5 stored the 16 digit words: 2.0 GB written + 2.0 GB read per
            // proof; recomputing them from the Montgomery form here cost +1.1 % of the prover's instructions, profiles/r06_valu_by_kernel.md: the shifts and masks alone are ~100)
            if (g < n) {
                uint32_t raw[NW + 1];
                const uint4 *src = reinterpret_cast<const uint4 *>(canon + (size_t)g * NW);
                const uint4 a = src[0], b = src[1];
                raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = a.w; raw[4] = b.x; raw[5] = b.y; raw[6] = b.z; raw[7] = b.w; raw[NW] = 0;
                part_digits<NW>(raw, L, d[q]);
            } else {
#pragma unroll
                for (int w = 0; w < PART_MAXW; w++) d[q][w] = PART_NONE;
            }
#pragma unroll
            for (int w = 0; w < PART_MAXW; w++) if (d[q][w] != PART_NONE) atomicAdd(&cnt[(d[q][w] & 0x7fffffffu) >> PART_FINE_BITS], 1u);
        }
        __syncthreads();
        part_block_scan<PART_THREADS>(cnt, PART_NBIN_MAX, wave_sums, &total);           // cnt[b] = the tile-local start of bin b
#pragma unroll
        for (int q = 0; q < PART_SPT; q++) {
            const uint32_t g = tile * PART_TILE + q * PART_THREADS + threadIdx.x;
            const uint32_t base = g < n1 ? off1 + g : off2 + (g - n1);
#pragma unroll
            for (int w = 0; w < PART_MAXW; w++) {
                if (d[q][w] == PART_NONE) continue;
                const uint32_t bucket = d[q][w] & 0x7fffffffu, bin = bucket >> PART_FINE_BITS;
                const uint32_t e = cnt[bin] + atomicAdd(&fill[bin], 1u);
                st_val[e] = ((uint32_t)w * stride + base) | (d[q][w] & 0x80000000u);
                st_key[e] = bucket | ((uint32_t)w << 20);
            }
        }
        __syncthreads();
        const uint32_t tot = total;
        for (uint32_t e = threadIdx.x; e < tot; e += PART_THREADS) {                      // bin-contiguous runs: consecutive lanes write consecutive addresses
            const uint32_t key = st_key[e], bucket = key & 0xfffffu, bin = bucket >> PART_FINE_BITS, dst = cursor[bin] + (e - cnt[bin]);
            out_val[dst] = st_val[e];
            out_key[dst] = (uint16_t)(((bucket & (PART_FINE - 1)) << PART_WBITS) | (key >> 20));
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nbin; b += PART_THREADS) cursor[b] += fill[b];
        __syncthreads();
    }
}
// one workgroup per coarse bin.  Round 5: the values reach their final place THROUGH LDS.  The direct form (out_val[lo + rank] = val from every lane) wrote 4 scattered
// bytes per pair into a region far larger than the L2s hold together; the lines left the chip partly filled: WRITE_SIZE 1.83 GB for 218 MB of values at 2^22 points,
// 1.08 ms per launch and the longest stretch a lone call's MSM waits for (profiles/r05_pmc_partition.md).  Now a sweep ranks the pairs of a KEY RANGE whose runs fit
// the staging array (30,720 values), drops them at their final relative position in LDS and streams the range out in order (full lines); a bin of 2^22 / 2^23 points
// takes two / four sweeps, each re-reading the bin's keys and values (6 bytes per pair, coalesced).  A single key with more pairs than the stage (skewed scalars) keeps
// the direct form for that key alone.
constexpr uint32_t PART_FINE_STAGE = 30720;
__global__ void __launch_bounds__(PART_FINE_THREADS) k_part_fine(const uint32_t *__restrict__ offs, uint32_t grid_a, const uint32_t *__restrict__ in_val, const uint16_t *__restrict__ in_key,
                                                                 uint32_t *__restrict__ out_val, uint32_t *__restrict__ start, uint32_t *__restrict__ end) {
    __shared__ uint32_t cur[PART_KEYS], stage[PART_FINE_STAGE], wave_sums[PART_FINE_THREADS / 64], total, split_key;
    const uint32_t bin = blockIdx.x, t = threadIdx.x;
    const uint32_t lo = offs[(size_t)bin * grid_a], hi = offs[(size_t)(bin + 1) * grid_a];
    for (uint32_t f = t; f < PART_KEYS; f += PART_FINE_THREADS) cur[f] = 0;
    __syncthreads();
    for (uint32_t e0 = lo + t; e0 < hi; e0 += PART_FINE_THREADS * PART_FINE_UNROLL) {
        uint32_t key[PART_FINE_UNROLL];
#pragma unroll
        for (int u = 0; u < PART_FINE_UNROLL; u++) { const uint32_t e = e0 + u * PART_FINE_THREADS; key[u] = e < hi ? in_key[e] : PART_NONE; }
#pragma unroll
        for (int u = 0; u < PART_FINE_UNROLL; u++) if (key[u] != PART_NONE) atomicAdd(&cur[key[u]], 1u);
    }
    __syncthreads();
    part_block_scan<PART_FINE_THREADS>(cur, PART_KEYS, wave_sums, &total);
    for (uint32_t f = t; f < PART_FINE; f += PART_FINE_THREADS) {
        const uint32_t k = bin * PART_FINE + f;
        start[k] = lo + cur[f << PART_WBITS];
        end[k] = lo + (f + 1 < PART_FINE ? cur[(f + 1) << PART_WBITS] : total);
    }
    __syncthreads();
    const uint32_t tot = total;
    uint32_t k_lo = 0, base = 0;                   // keys below k_lo are done (their cursors have run to the next key's start); `base` = where key k_lo's run begins
    while (base < tot) {                           // (uniform over the workgroup)
        // keys [k_lo, k_hi): the longest range whose runs end within base + STAGE.  The run that holds position base + STAGE belongs to exactly one key >= k_lo.
        const uint32_t lim = base + PART_FINE_STAGE;
        if (tot <= lim) { if (t == 0) split_key = PART_KEYS; }
        else for (uint32_t f = k_lo + t; f < PART_KEYS; f += PART_FINE_THREADS) {
            const uint32_t e_f = f + 1 < PART_KEYS ? cur[f + 1] : tot;
            if (cur[f] <= lim && e_f > lim) split_key = f;
        }
        __syncthreads();
        uint32_t k_hi = split_key;
        const bool direct = k_hi == k_lo;          // one key alone overflows the stage
        if (direct) k_hi = k_lo + 1;
        const uint32_t stop = k_hi < PART_KEYS ? cur[k_hi] : tot;      // (key k_hi's cursor is not touched by this sweep)
        __syncthreads();
        for (uint32_t e0 = lo + t; e0 < hi; e0 += PART_FINE_THREADS * PART_FINE_UNROLL) {
            uint32_t key[PART_FINE_UNROLL], val[PART_FINE_UNROLL];
#pragma unroll
            for (int u = 0; u < PART_FINE_UNROLL; u++) {          // (keys and values in one round of loads: a sweep's latency is what a lone call waits for, the re-read is coalesced)
                const uint32_t e = e0 + u * PART_FINE_THREADS;
                key[u] = PART_NONE;
                if (e < hi) { key[u] = in_key[e]; val[u] = in_val[e]; }
            }
#pragma unroll
            for (int u = 0; u < PART_FINE_UNROLL; u++) if (key[u] >= k_lo && key[u] < k_hi) {
                const uint32_t pos = atomicAdd(&cur[key[u]], 1u);
                if (direct) out_val[lo + pos] = val[u]; else stage[pos - base] = val[u];
            }
        }
        __syncthreads();
        if (!direct) for (uint32_t i = t; i < stop - base; i += PART_FINE_THREADS) out_val[lo + base + i] = stage[i];
        __syncthreads();
        base = stop; k_lo = k_hi;
    }
}
// digits + grouping of the table path; leaves S.sorted_vals / S.start / S.end as prepare_buckets would (bucket-contiguous, by window inside a bucket, zero digits absent)
template <class Fr>
static void partition_buckets(MsmWorkspace &S, const Fr *scal1, size_t n1, size_t off1, const Fr *scal2, size_t n2, size_t off2, const TableLayout &L, size_t stride, uint32_t cap, hipStream_t s) {
    const int B = L.c_hi - 1;                                         // bucket bits
    const uint32_t nbin = 1u << (B - PART_FINE_BITS), G = PART_GRID;
    const size_t n = n1 + n2, nb = (size_t)1 << B, nh = (size_t)nbin * G + 1;
    if (n > S.cap_canon) { dfree(S.canon); S.canon = (uint32_t *)dmalloc(n * Fr::N * 4); S.cap_canon = n; }
    if (nh > S.cap_part) { dfree(S.part_hist); dfree(S.part_offs); S.part_hist = (uint32_t *)dmalloc(nh * 4); S.part_offs = (uint32_t *)dmalloc(nh * 4); S.cap_part = nh; }
#ifdef ZKAES_MEASURE
    for (int rep = (knockin() & 1) ? 0 : 1; rep < 2; rep++)
#endif
    {
    hipLaunchKernelGGL((k_part_hist<Fr>), dim3(G), dim3(PART_THREADS), 0, s, scal1, (uint32_t)n1, scal2, (uint32_t)n2, L, nbin, S.canon, S.part_hist);
    HIP_LAUNCH_CHECK();
    size_t tb = 0;
    HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, S.part_hist, S.part_offs, 0u, nh, rocprim::plus<uint32_t>(), s));
    if (tb > S.cap_tmp) { dfree(S.tmp); S.tmp = dmalloc(tb); S.cap_tmp = tb; }
    HIP_CHECK(rocprim::exclusive_scan(S.tmp, tb, S.part_hist, S.part_offs, 0u, nh, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL((k_part_scatter<Fr::N>), dim3(G), dim3(PART_THREADS), 0, s, (const uint32_t *)S.canon, (uint32_t)n, (uint32_t)n1, L, (uint32_t)off1, (uint32_t)off2, (uint32_t)stride, nbin,
                       (const uint32_t *)S.part_offs, S.vals_a, (uint16_t *)S.keys_a);
    HIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_part_fine, dim3(nbin), dim3(PART_FINE_THREADS), 0, s, (const uint32_t *)S.part_offs, G, (const uint32_t *)S.vals_a, (const uint16_t *)S.keys_a, S.vals_b, S.start, S.end);
    HIP_LAUNCH_CHECK();
    }
    S.sorted_vals = S.vals_b; S.sorted_keys = nullptr;
    order_buckets(S, nb, cap, s);
}

// Table-mode Pippenger in the same two steps as the per-window variant.  msm_prepare_table: signed c-bit digits of up to two scalar vectors
// (element i of vector v names table entry w * stride + off_v + i in window w), sort on the c - 1 bucket bits, bucket ranges;
// msm_finish (above) with `tables` = copy 0 (optionally shifted by a constant index, e.g. to the shifted-powers part of every copy).
template <class Curve>
void msm_prepare_table(MsmWorkspace *ws_, const typename Curve::Fr *scal1, size_t n1, size_t off1, const typename Curve::Fr *scal2, size_t n2, size_t off2, int c, size_t stride,
                       stream_t s_) {
    using Fr = typename Curve::Fr;
    hipStream_t s = (hipStream_t)s_;
    if (!ws_) throw GpuError("msm: null workspace");
    MsmWorkspace &S = *ws_;
    size_t n = n1 + n2;
    S.plan_n = n;
    if (n == 0) return;
    if (c < 4 || c > 24) throw GpuError("msm_table: window bits out of range");
    const TableLayout L = table_layout(Fr::BITS + 1, c);
    const int nwin = L.nwin;
    if (off1 + n1 > stride || off2 + n2 > stride) throw GpuError("msm_table: range exceeds the table");
    if ((uint64_t)nwin * stride >= (1ull << 30) || (uint64_t)n * nwin >= (1ull << 31)) throw GpuError("msm_table: index range too large");
    size_t pairs = n * (size_t)nwin;
    const size_t nb = (size_t)1 << (L.c_hi - 1);
    // the per-lane cap only exists to bound the longest lane under skewed scalars; it must stay well above the MEAN bucket size or every bucket overflows into the tail kernel
    // (uniform digits: pairs / 2^19 per bucket -- ~210 for the largest MSM of a 6-block proof, ~830 for a 28-block one over a larger SRS: round 6)
    uint32_t cap = BUCKET_CAP_TABLE;
    while ((uint64_t)cap < 2 * (pairs / nb) + 64) cap <<= 1;
    S.plan_c = L.c_hi; S.plan_nwin = nwin; S.plan_pairs = pairs; S.plan_table = true; S.plan_cap = cap;
    ensure_scratch(S, pairs, nb, cap);
    const int B = L.c_hi - 1;
    if (B > PART_FINE_BITS && (1u << (B - PART_FINE_BITS)) <= PART_NBIN_MAX && nwin <= PART_MAXW && pairs >= ((size_t)1 << 16)) {
        partition_buckets<Fr>(S, scal1, n1, off1, scal2, n2, off2, L, stride, cap, s);
        return;
    }
    // small instances and window plans outside the partition's limits: digits + one stable radix sort over the bucket bits
    if (n1) { hipLaunchKernelGGL((k_digits_table<Fr>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, scal1, (uint32_t)n1, 0u, (uint32_t)n, (uint32_t)off1, L, (uint32_t)stride, S.keys_a, S.vals_a); HIP_LAUNCH_CHECK(); }
    if (n2) { hipLaunchKernelGGL((k_digits_table<Fr>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, scal2, (uint32_t)n2, (uint32_t)n1, (uint32_t)n, (uint32_t)off2, L, (uint32_t)stride, S.keys_a, S.vals_a); HIP_LAUNCH_CHECK(); }
    prepare_buckets<typename Curve::FqP>(S, pairs, L.c_hi - 1, 1, L.c_hi - 1, cap, s);
}
template <class Curve>
XYZZ<typename Curve::Fq> msm_table(MsmWorkspace *ws_, const Affine28<typename Curve::FqP> *tables, size_t stride, size_t off, int c, const typename Curve::Fr *scalars, size_t n, stream_t s_) {
    if (n == 0) return XYZZ<typename Curve::Fq>::inf();
    msm_prepare_table<Curve>(ws_, scalars, n, off, nullptr, 0, 0, c, stride, s_);
    return msm_finish<Curve>(ws_, tables, s_);
}
template <class Curve>
XYZZ<typename Curve::Fq> msm_table(MsmWorkspace *ws_, const Niels28<typename Curve::FqP> *tables, size_t stride, size_t off, int c, const typename Curve::Fr *scalars, size_t n, stream_t s_) {
    if (n == 0) return XYZZ<typename Curve::Fq>::inf();
    msm_prepare_table<Curve>(ws_, scalars, n, off, nullptr, 0, 0, c, stride, s_);
    return msm_finish<Curve>(ws_, tables, s_);
}

// ---- fixed-base powers: out[i] = beta^(from + i) * base
template <class Fr>
__global__ void k_power_scalars(Fr beta, uint64_t from, uint32_t count, Fr *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = beta.pow_u64(from + i);
}
template <class Fq, class Fr>
__global__ void __launch_bounds__(64) k_fixed_base(const Affine<Fq> *__restrict__ table, const Fr *__restrict__ scalars, uint32_t count, Affine<Fq> *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t raw[Fr::N];
    scalars[i].to_raw(raw);
    XYZZ<Fq> acc = XYZZ<Fq>::inf();
    for (int w = 0; w < Fr::N * 4; w++) {
        uint32_t d = (raw[w >> 2] >> ((w & 3) * 8)) & 0xff;
        if (d) acc.madd(table[w * 255 + d - 1]);
    }
    out[i] = acc.to_affine();
}

namespace {
template <class Curve>
DevPtr<Affine<typename Curve::Fq>> upload_fixed_base_table(const Affine<typename Curve::Fq> &base, hipStream_t s) {
    using Fq = typename Curve::Fq;
    const int NW = Curve::Fr::N * 4;
    std::vector<Affine<Fq>> table((size_t)NW * 255);
    XYZZ<Fq> wb = XYZZ<Fq>::from_affine(base);
    for (int w = 0; w < NW; w++) {
        XYZZ<Fq> acc = wb;
        for (int d = 1; d <= 255; d++) { table[(size_t)w * 255 + d - 1] = acc.to_affine(); acc.add(wb); }
        for (int k = 0; k < 8; k++) wb = wb.dbl();
    }
    DevPtr<Affine<Fq>> d_table(table.size());
    HIP_CHECK(hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(Affine<Fq>), hipMemcpyHostToDevice, s));
    sync((stream_t)s);
    return d_table;
}
}  // namespace

template <class Curve>
void fixed_base_powers(Affine<typename Curve::Fq> *out, const Affine<typename Curve::Fq> &base, const typename Curve::Fr &beta, size_t from, size_t count, stream_t s_) {
    using Fq = typename Curve::Fq;
    using Fr = typename Curve::Fr;
    hipStream_t s = (hipStream_t)s_;
    if (!count) return;
    DevPtr<Affine<Fq>> d_table = upload_fixed_base_table<Curve>(base, s);
    const size_t CH = 1 << 20;
    DevPtr<Fr> d_sc(CH);
    for (size_t off = 0; off < count; off += CH) {
        uint32_t m = (uint32_t)((count - off) < CH ? (count - off) : CH);
        hipLaunchKernelGGL((k_power_scalars<Fr>), dim3((m + 255) / 256), dim3(256), 0, s, beta, (uint64_t)(from + off), m, d_sc.get());
        HIP_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_fixed_base<Fq, Fr>), dim3((m + 63) / 64), dim3(64), 0, s, (const Affine<Fq> *)d_table.get(), (const Fr *)d_sc.get(), m, out + off);
        HIP_LAUNCH_CHECK();
    }
    sync((stream_t)s);
}
// out[i] = scalars[i] * base for device-resident scalars (Lagrange-basis SRS points)
template <class Curve>
void fixed_base_scalars(Affine<typename Curve::Fq> *out, const Affine<typename Curve::Fq> &base, const typename Curve::Fr *scalars, size_t count, stream_t s_) {
    using Fq = typename Curve::Fq;
    using Fr = typename Curve::Fr;
    hipStream_t s = (hipStream_t)s_;
    if (!count) return;
    DevPtr<Affine<Fq>> d_table = upload_fixed_base_table<Curve>(base, s);
    const size_t CH = 1 << 20;
    for (size_t off = 0; off < count; off += CH) {
        uint32_t m = (uint32_t)((count - off) < CH ? (count - off) : CH);
        hipLaunchKernelGGL((k_fixed_base<Fq, Fr>), dim3((m + 63) / 64), dim3(64), 0, s, (const Affine<Fq> *)d_table.get(), scalars + off, m, out + off);
        HIP_LAUNCH_CHECK();
    }
    sync((stream_t)s);
}

// ---- sum of bases weighted by SMALL integers (|v| <= 2): the Lagrange-basis commitments of 0/1-valued evaluation vectors.
// One lane per 16 consecutive bases (short chains: the lanes of a wave diverge between the two classes) keeps two accumulators (|v| = 1, |v| = 2; the sign negates y); a two-level tree sums the partials.
constexpr int CLS_CHUNK = 16;
template <class Law>
__global__ void __launch_bounds__(64, 2) k_class_partials(const typename Law::Base *__restrict__ bases, const int8_t *__restrict__ vals, uint32_t n, typename Law::Acc *__restrict__ part1,
                                                           typename Law::Acc *__restrict__ part2, uint32_t *__restrict__ flags) {
    using P = typename Law::Params;
    using G = FpMsm<P>;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s0 = t * CLS_CHUNK;
    if (s0 >= n) return;
    uint32_t e = s0 + CLS_CHUNK < n ? s0 + CLS_CHUNK : n;
    if constexpr (Law::edwards) {
        // ONE accumulator and ONE te_madd call site: the chunk is walked once per class (round 5 kept both classes' accumulators live around two inlined additions:
        // 256 registers + 164 B of scratch).  The second walk costs 16 byte loads when -- as almost always -- no value of the chunk is +-2.
#pragma unroll 1
        for (int cls = 1; cls <= 2; cls++) {
            AccTE<P> a = te_identity<P>();
            for (uint32_t i = s0; i < e; i++) {
                const int v = vals[i], m = v < 0 ? -v : v;
                if (m != cls) { if (m > 2 && cls == 1) atomicOr(flags, 2u); continue; }
                Niels28<P> p = bases[i];
                if (v < 0) p = niels_neg<P>(p);
                te_madd<P>(a, p);
            }
            (cls == 1 ? part1 : part2)[t] = a;
        }
        return;
    } else
    {
    Acc28<P> a1, a2;
    bool inf1 = true, inf2 = true;
    for (uint32_t i = s0; i < e; i++) {
        int v = vals[i];
        if (v == 0) continue;
        Affine28<P> p = bases[i];
        if (p.is_inf()) continue;
        if (v < 0) { p.y = G::zero().template sub<2>(p.y); v = -v; }
        if (v == 1) {
            if (inf1) { a1.x = p.x; a1.y = p.y; a1.zz = G::k_one(); a1.zzz = a1.zz; inf1 = false; }
            else if (!madd28(a1, p)) atomicOr(flags, 1u);
        } else if (v == 2) {
            if (inf2) { a2.x = p.x; a2.y = p.y; a2.zz = G::k_one(); a2.zzz = a2.zz; inf2 = false; }
            else if (!madd28(a2, p)) atomicOr(flags, 1u);
        } else atomicOr(flags, 2u);
    }
    part1[t] = inf1 ? inf28<P>() : a1;
    part2[t] = inf2 ? inf28<P>() : a2;
    }
}
// out[b] = sum of in[b * per .. (b+1) * per)   (one block of 256 lanes per output); blockIdx.y selects one of several independent arrays laid out at fixed strides
// (the two classes of class_sum share a launch).  The LDS tree starts at the width that holds data: 64 partials take six levels, not eight.
template <class A>
__global__ void __launch_bounds__(256) k_sum_tree(const A *__restrict__ in, uint32_t total, uint32_t per, A *__restrict__ out, uint32_t in_stride, uint32_t out_stride) {
    __shared__ A sh[256];
    in += (size_t)blockIdx.y * in_stride; out += (size_t)blockIdx.y * out_stride;
    uint32_t b = blockIdx.x, t = threadIdx.x;
    uint32_t lo = b * per, hi = lo + per < total ? lo + per : total;
    A acc = PtOps<A>::identity();
    for (uint32_t i = lo + t; i < hi; i += 256) PtOps<A>::add(acc, in[i]);
    sh[t] = acc;
    __syncthreads();
    int width = 256;
    while (width > 2 && (uint32_t)(width >> 1) >= hi - lo) width >>= 1;          // lanes >= hi - lo hold the identity
    for (int s = width >> 1; s > 0; s >>= 1) {
        if ((int)t < s) { A a = sh[t]; PtOps<A>::add(a, sh[t + s]); sh[t] = a; }
        __syncthreads();
    }
    if (t == 0) out[b] = sh[0];
}
// the same on the Edwards law with four lanes per point operation (te28.cuh te_add_quad: 91 registers, no scratch, no call): 64 quads gather their shares, then a tree in LDS.
// k_sum_tree's full-lane additions go through a function call -- 229 registers + 464 B of stack -- and took 451 us per launch in a lone 16-byte call, twice per class sum.
template <class P>
__global__ void __launch_bounds__(RQ_THREADS) k_quad_sum(const AccTE<P> *__restrict__ in, uint32_t total, uint32_t per, AccTE<P> *__restrict__ out, uint32_t in_stride, uint32_t out_stride) {
    __shared__ uint32_t pt[RQ_QUADS * PT_WORDS];
    in += (size_t)blockIdx.y * in_stride; out += (size_t)blockIdx.y * out_stride;
    const uint32_t b = blockIdx.x, quad = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    const uint32_t lo = b * per, hi = lo + per < total ? lo + per : total;
    FpMsm<P> acc = te_identity_quad<P>(q);
    for (uint32_t i = lo + quad; i < hi; i += RQ_QUADS) acc = te_add_quad<P>(acc, quad_load<P>(reinterpret_cast<const uint32_t *>(in + i), q), q);
    quad_store<P>(pt + quad * PT_WORDS, q, acc);
    __syncthreads();
    quad_tree_sum<P>(pt, RQ_QUADS, quad, q);
    if (quad == 0) quad_store<P>(reinterpret_cast<uint32_t *>(out + b), q, quad_load<P>(pt, q));
}
// the two class sums in the library-wide form + the flag word, straight into the workspace's pinned host result; re-arms the flag word
template <class A>
__global__ void __launch_bounds__(64) k_class_result(const A *__restrict__ in, uint32_t *__restrict__ flags, XYZZ<Fp<typename PtOps<A>::Params>> *__restrict__ out) {
    uint32_t i = threadIdx.x;
    if (i < 2) out[i] = PtOps<A>::to_std(in[i]);
    if (i == 2) { *reinterpret_cast<uint32_t *>(out + 2) = *flags; *flags = 0; }
}

template <class Curve, class Law>
static bool class_sum_impl(MsmWorkspace *ws_, const typename Law::Base *bases, const int8_t *vals, size_t n, XYZZ<typename Curve::Fq> *out, stream_t s_) {
    using A = typename Law::Acc;
    using Fq = typename Curve::Fq;
    hipStream_t s = (hipStream_t)s_;
    *out = XYZZ<Fq>::inf();
    if (n == 0) return true;
    if (!ws_) throw GpuError("class_sum: null workspace");
    MsmWorkspace &S = *ws_;
    uint32_t chunks = (uint32_t)((n + CLS_CHUNK - 1) / CLS_CHUNK), mid = (chunks + 255) / 256;
    // scratch carved from the bucket array (sized for >= 2 * chunks + 2 * mid + 2 points by any prior msm of this context, else grown here)
    size_t need_buckets = 2 * (size_t)chunks + 2 * mid + 8;
    ensure_scratch(S, 1, need_buckets, 0);
    A *p1 = (A *)S.buckets, *p2 = p1 + chunks, *m1 = p2 + chunks, *m2 = m1 + mid, *fin = m2 + mid;
    hipLaunchKernelGGL((k_class_partials<Law>), dim3((chunks + 63) / 64), dim3(64), 0, s, bases, vals, (uint32_t)n, p1, p2, S.ctrl + 3);
    HIP_LAUNCH_CHECK();
    (void)p2; (void)m2;          // (class 2's arrays follow class 1's at the strides below)
    if constexpr (Law::edwards) {
        hipLaunchKernelGGL((k_quad_sum<typename Law::Params>), dim3(mid, 2), dim3(RQ_THREADS), 0, s, (const A *)p1, chunks, 256u, m1, chunks, mid); HIP_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_quad_sum<typename Law::Params>), dim3(1, 2), dim3(RQ_THREADS), 0, s, (const A *)m1, mid, mid, fin, mid, 1u); HIP_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL((k_sum_tree<A>), dim3(mid, 2), dim3(256), 0, s, (const A *)p1, chunks, 256u, m1, chunks, mid); HIP_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_sum_tree<A>), dim3(1, 2), dim3(256), 0, s, (const A *)m1, mid, mid, fin, mid, 1u); HIP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((k_class_result<A>), dim3(1), dim3(64), 0, s, (const A *)fin, S.ctrl + 3, (XYZZ<Fq> *)S.d_res); HIP_LAUNCH_CHECK();
    XYZZ<Fq> r[2];
    uint32_t flags = 0;
    sync((stream_t)s);        // (sleeps in throughput mode) results are in pinned host memory once the stream has drained
    memcpy(r, S.h_res, sizeof r);
    memcpy(&flags, (const char *)S.h_res + sizeof r, 4);
    if (flags) return false;          // a value outside [-2, 2] or a degenerate addition: the caller falls back to the generic MSM
    oplog_msm(n, OP_MSM_CLASS_SUM);
    XYZZ<Fq> t = r[1].dbl();
    t.add(r[0]);
    *out = t;
    return true;
}
template <class Curve>
bool class_sum(MsmWorkspace *ws_, const Affine28<typename Curve::FqP> *bases, const int8_t *vals, size_t n, XYZZ<typename Curve::Fq> *out, stream_t s_) {
    return class_sum_impl<Curve, WeierLaw<typename Curve::FqP>>(ws_, bases, vals, n, out, s_);
}
template <class Curve>
bool class_sum(MsmWorkspace *ws_, const Niels28<typename Curve::FqP> *bases, const int8_t *vals, size_t n, XYZZ<typename Curve::Fq> *out, stream_t s_) {
    return class_sum_impl<Curve, EdwardsLaw<typename Curve::FqP>>(ws_, bases, vals, n, out, s_);
}
template XYZZ<Fq377> msm_finish<Bls377>(MsmWorkspace *, const Niels28<Fq377P> *, stream_t);
template void msm_finish2<Bls377>(MsmWorkspace *, const Niels28<Fq377P> *, const Niels28<Fq377P> *, XYZZ<Fq377> *, stream_t);
template XYZZ<Fq377> msm<Bls377>(MsmWorkspace *, const Niels28<Fq377P> *, const Fr377 *, size_t, stream_t);
template XYZZ<Fq377> msm_table<Bls377>(MsmWorkspace *, const Niels28<Fq377P> *, size_t, size_t, int, const Fr377 *, size_t, stream_t);
template bool class_sum<Bls377>(MsmWorkspace *, const Niels28<Fq377P> *, const int8_t *, size_t, XYZZ<Fq377> *, stream_t);

template void msm_table_sum_device<Bls377>(MsmWorkspace *, const Niels28<Fq377P> *, size_t, size_t, int, const Fr377 *, size_t, XYZZ<Fq377> *, stream_t);
template XYZZ<Fq377> msm_fold_points_device<Bls377>(const XYZZ<Fq377> *, int, stream_t);
template XYZZ<Fq381> msm_fold_points_device<Bls381>(const XYZZ<Fq381> *, int, stream_t);
template void msm_sharded_plan<Bls377>(size_t, int *, int *);
template void msm_sharded_plan<Bls381>(size_t, int *, int *);
template void msm_window_sums_device<Bls377>(MsmWorkspace *, const Affine28<Fq377P> *, const Fr377 *, size_t, size_t, XYZZ<Fq377> *, stream_t);
template void msm_window_sums_device<Bls381>(MsmWorkspace *, const Affine28<Fq381P> *, const Fr381 *, size_t, size_t, XYZZ<Fq381> *, stream_t);
template XYZZ<Fq377> msm_fold_window_sums_device<Bls377>(const XYZZ<Fq377> *, int, size_t, stream_t);
template XYZZ<Fq381> msm_fold_window_sums_device<Bls381>(const XYZZ<Fq381> *, int, size_t, stream_t);
template void msm_prepare_table<Bls377>(MsmWorkspace *, const Fr377 *, size_t, size_t, const Fr377 *, size_t, size_t, int, size_t, stream_t);
template void msm_prepare_table<Bls381>(MsmWorkspace *, const Fr381 *, size_t, size_t, const Fr381 *, size_t, size_t, int, size_t, stream_t);
template XYZZ<Fq377> msm_table<Bls377>(MsmWorkspace *, const Affine28<Fq377P> *, size_t, size_t, int, const Fr377 *, size_t, stream_t);
template XYZZ<Fq381> msm_table<Bls381>(MsmWorkspace *, const Affine28<Fq381P> *, size_t, size_t, int, const Fr381 *, size_t, stream_t);
template void convert_bases<Bls377>(Affine28<Fq377P> *, const Affine<Fq377> *, size_t, stream_t);
template void convert_bases<Bls381>(Affine28<Fq381P> *, const Affine<Fq381> *, size_t, stream_t);
template void build_window_tables<Bls377>(Affine<Fq377> *, size_t, int, stream_t);
template void table_next<Bls377>(Affine<Fq377> *, const Affine<Fq377> *, size_t, int, int, stream_t);
template void table_next<Bls381>(Affine<Fq381> *, const Affine<Fq381> *, size_t, int, int, stream_t);
template void build_window_tables<Bls381>(Affine<Fq381> *, size_t, int, stream_t);
template int table_windows<Bls377>(int);
template int table_windows<Bls381>(int);
template void msm_prepare<Bls377>(MsmWorkspace *, const Fr377 *, size_t, const Fr377 *, size_t, size_t, stream_t, int);
template void msm_prepare<Bls381>(MsmWorkspace *, const Fr381 *, size_t, const Fr381 *, size_t, size_t, stream_t, int);
template XYZZ<Fq377> msm_finish<Bls377>(MsmWorkspace *, const Affine28<Fq377P> *, stream_t);
template XYZZ<Fq381> msm_finish<Bls381>(MsmWorkspace *, const Affine28<Fq381P> *, stream_t);
template XYZZ<Fq377> msm<Bls377>(MsmWorkspace *, const Affine28<Fq377P> *, const Fr377 *, size_t, stream_t);
template XYZZ<Fq381> msm<Bls381>(MsmWorkspace *, const Affine28<Fq381P> *, const Fr381 *, size_t, stream_t);
template void fixed_base_scalars<Bls377>(Affine<Fq377> *, const Affine<Fq377> &, const Fr377 *, size_t, stream_t);
template bool class_sum<Bls377>(MsmWorkspace *, const Affine28<Fq377P> *, const int8_t *, size_t, XYZZ<Fq377> *, stream_t);
template void fixed_base_powers<Bls377>(Affine<Fq377> *, const Affine<Fq377> &, const Fr377 &, size_t, size_t, stream_t);
template void fixed_base_powers<Bls381>(Affine<Fq381> *, const Affine<Fq381> &, const Fr381 &, size_t, size_t, stream_t);

}  // namespace gpu
}  // namespace zk
