// csrc/ff29.cuh -- reduced-radix ("lazy carry") Montgomery arithmetic for the 253/255-bit SCALAR fields inside the NTT butterflies (gfx950).
//
// Same idea as ff28.cuh for the base field: 9 limbs of 29 bits, a limb product is < 2^58 and a 64-bit accumulator absorbs the 18 products of a
// column, so a field product is 2 * 81 = 162 v_mad_u64_u32 with the accumulator as the addend plus ~70 shifts/masks -- no carry chain.  The
// 8x32-bit CIOS product of ff.cuh is 128 multiplies plus several hundred carry fix-ups, which cost more than the multiplies on this ISA.
//
// Representation: value = sum l[i] 2^(29 i), limbs normalized to < 2^29 after every operation (the top limb keeps the excess),
// Montgomery radix R' = 2^261 >= 64 p.  "Almost Montgomery": for a < B p and a canonical b < p the product is < (1 + B/64) p, so no conditional
// subtraction is needed while B stays small; additions let values grow and callers bound them statically (see k_ntt_pass).
// The NTT keeps its data in the library-wide form x R (R = 2^256) and only re-limbs it (split / pack, no arithmetic): the twiddles are stored
// as w R' so that  mul(x R, w R') = x w R.
#pragma once
#include "ff.cuh"

namespace zk {

template <class P>   // P = Fr377P / Fr381P (8 x 32-bit parameter pack)
struct Fp29 {
    static constexpr int N = 9;
    static constexpr int B = 29;
    static constexpr uint32_t MASK = (1u << B) - 1;
    uint32_t l[N];

    ZK_HD static constexpr uint32_t mod29(int i) {           // limb i of p in radix 2^29
        int bit = B * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = (w < P::N ? (uint64_t)P::mod(w) : 0) | ((w + 1 < P::N ? (uint64_t)P::mod(w + 1) : 0) << 32);
        return (uint32_t)(two >> sh) & MASK;
    }
    static constexpr uint32_t PINV = P::INV & MASK;          // -p^-1 mod 2^29
    // limb i of K p (normalized), and the same with the borrows pre-distributed: sum c_i 2^(29 i) = K p with c_i >= 2^29 - 1 for i < N - 1,
    // so that a_i - b_i + c_i never goes negative for normalized b (no signed arithmetic, no borrow chain)
    template <int K> ZK_HD static constexpr uint32_t kp_limb(int i) {
        uint64_t c = 0, v = 0;
        for (int j = 0; j <= i; j++) { v = (uint64_t)K * mod29(j) + c; c = v >> B; }
        return i == N - 1 ? (uint32_t)v : (uint32_t)v & MASK;
    }
    template <int K> ZK_HD static constexpr uint32_t kp_spread(int i) {
        return i == 0 ? kp_limb<K>(0) + (1u << B) : (i == N - 1 ? kp_limb<K>(i) - 1u : kp_limb<K>(i) + MASK);
    }

    ZK_HD static Fp29 zero() { Fp29 r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
    // re-limb an 8x32 little-endian integer (no arithmetic)
    ZK_HD static Fp29 split(const uint32_t *w) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int bit = B * i, k = bit >> 5, sh = bit & 31;
            uint64_t two = (k < P::N ? (uint64_t)w[k] : 0) | ((k + 1 < P::N ? (uint64_t)w[k + 1] : 0) << 32);
            r.l[i] = (uint32_t)(two >> sh) & MASK;
        }
        return r;
    }
    // pack normalized limbs of a value < 2^256 back into 8x32
    ZK_HD void pack(uint32_t *w) const {
#pragma unroll
        for (int k = 0; k < P::N; k++) {
            int bit = 32 * k, i = bit / B, sh = bit % B;       // word k = bits [32k, 32k + 32)
            uint64_t v = (uint64_t)l[i] >> sh;
            int have = B - sh;
            if (i + 1 < N) v |= (uint64_t)l[i + 1] << have;
            if (have + B < 32 && i + 2 < N) v |= (uint64_t)l[i + 2] << (have + B);
            w[k] = (uint32_t)v;
        }
    }
    ZK_HD Fp29 operator+(const Fp29 &b) const {               // no modular reduction; the value grows
        Fp29 r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint32_t v = l[i] + b.l[i] + c; r.l[i] = v & MASK; c = v >> B; }
        r.l[N - 1] = l[N - 1] + b.l[N - 1] + c;
        return r;
    }
    // a - b + K p, K p >= b required (keeps the value non-negative); b normalized
    template <int K>
    ZK_HD Fp29 sub(const Fp29 &b) const {
        Fp29 r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint32_t v = l[i] + kp_spread<K>(i) - b.l[i] + c; r.l[i] = v & MASK; c = v >> B; }
        r.l[N - 1] = l[N - 1] + kp_spread<K>(N - 1) - b.l[N - 1] + c;
        return r;
    }
    // ---- lazy limb arithmetic (round 5): NO carry propagation, limbs may exceed 29 bits.  A product takes ONE lazy operand with limbs < 6 x 2^29 beside a normalized one
    // (a column then holds 9 limb products of < 6 x 2^58 plus the reduction's 9 x 2^58: 63 x 2^58 < 2^64); operator+ and sub<K> take a lazy left operand with limbs
    // < 5 x 2^29 and return normalized limbs (their own carry chain absorbs it: 5 + 2 + carry stays below 2^32).  k_ntt_pass keeps the schedule.
    ZK_HD Fp29 add_lazy(const Fp29 &b) const { Fp29 r; for (int i = 0; i < N; i++) r.l[i] = l[i] + b.l[i]; return r; }      // limbs: + 1 x 2^29
    // a - b + K p for a NORMALIZED b (a product): limbs grow by less than 2 x 2^29.  On the device limbs 0..7 are ONE instruction each: kp_spread >= 2^29 - 1 >= b_i there, so
    // l + (kp - b) = |kp - b| + l = v_sad_u32 (the top limb keeps the two-instruction form: b's excess sits there)
    template <int K> ZK_HD Fp29 sub_lazy(const Fp29 &b) const {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < N; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (i < N - 1) { asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r.l[i]) : "s"(kp_spread<K>(i)), "v"(b.l[i]), "v"(l[i])); continue; }
#endif
            r.l[i] = l[i] + kp_spread<K>(i) - b.l[i];
        }
        return r;
    }
    // carry-propagate lazy limbs (each < 2^32) into normalized ones; the value is unchanged (the top limb keeps the excess)
    ZK_HD Fp29 normalized() const {
        Fp29 r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { const uint64_t v = (uint64_t)l[i] + c; r.l[i] = (uint32_t)v & MASK; c = (uint32_t)(v >> B); }
        r.l[N - 1] = l[N - 1] + c;
        return r;
    }
    // almost-Montgomery product (radix R' = 2^261): row-wise operand scanning, 64-bit column accumulators
    ZK_HD Fp29 operator*(const Fp29 &b) const {
        uint64_t t[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (uint64_t)l[j] * b.l[i];
            if constexpr (mod29(0) == 1u && PINV == MASK) {       // p = 1 (mod 2^29) (both BLS scalar fields): m = -t_i, no multiply for limb 0
                const uint64_t u = t[i] + MASK;                   // high part: the carry (t_i + m) >> 29; low 29 bits: those of t_i - 1, so m = -t_i = ~u (mod 2^29), one v_bitop3_b32
                const uint32_t m = ~(uint32_t)u & MASK;
#pragma unroll
                for (int j = 1; j < N; j++) t[i + j] += (uint64_t)m * mod29(j);
                t[i + 1] += u >> B;
            } else {
                uint32_t m = (((uint32_t)t[i] & MASK) * PINV) & MASK;
#pragma unroll
                for (int j = 0; j < N; j++) t[i + j] += (uint64_t)m * mod29(j);
                t[i + 1] += t[i] >> B;
            }
        }
        Fp29 r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint64_t v = t[N + i] + c; r.l[i] = (uint32_t)v & MASK; c = v >> B; }
        r.l[N - 1] = (uint32_t)(t[2 * N - 1] + c);
        return r;
    }
    // sum_{i < N} a[i] b[i] / R' with ONE Montgomery reduction: the N x 81 limb products accumulate in the same 64-bit columns.  N <= 4 (36 limb products of < 2^58 plus
    // the reduction's 9 per column stay below 2^64); limbs normalized.  For a[i] < A_i p, b[i] < B_i p the result is < (sum A_i B_i / 256 + 1) p  (p / R' < 2^-8).
    template <int N>
    ZK_HD static Fp29 dot(const Fp29 *a, const Fp29 *b) {
        static_assert(N >= 1 && N <= 4, "column accumulators hold at most four products per limb pair");
        static_assert(mod29(0) == 1u && PINV == MASK, "written for p = 1 (mod 2^29)");
        uint64_t t[2 * Fp29::N];
#pragma unroll
        for (int i = 0; i < 2 * Fp29::N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < Fp29::N; i++) {
#pragma unroll
            for (int n = 0; n < N; n++)
#pragma unroll
                for (int j = 0; j < Fp29::N; j++) t[i + j] += (uint64_t)a[n].l[j] * b[n].l[i];
            const uint64_t u = t[i] + MASK;
            const uint32_t m = ~(uint32_t)u & MASK;
#pragma unroll
            for (int j = 1; j < Fp29::N; j++) t[i + j] += (uint64_t)m * mod29(j);
            t[i + 1] += u >> B;
        }
        Fp29 r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < Fp29::N - 1; i++) { uint64_t v = t[Fp29::N + i] + c; r.l[i] = (uint32_t)v & MASK; c = v >> B; }
        r.l[Fp29::N - 1] = (uint32_t)(t[2 * Fp29::N - 1] + c);
        return r;
    }
    // 32 x: the limbs shifted up by five bits (x < 2^256: the result fits the 261 bits of nine limbs).  For the product of two values that are BOTH in the standard
    // form x R, y R (R = 2^256 = R' / 32):  mul(shl5(x R), y R) = x y R.
    ZK_HD Fp29 shl5() const {
        Fp29 r;
        r.l[0] = (l[0] << 5) & MASK;
#pragma unroll
        for (int i = 1; i < N; i++) r.l[i] = ((l[i] << 5) | (l[i - 1] >> (B - 5))) & (i == N - 1 ? 0xffffffffu : MASK);
        return r;
    }
    // fully reduce a value < 2^(LOG + 1) p to [0, p): peel 2^LOG p, ..., 2 p, p
    template <int LOG>
    ZK_HD Fp29 canonical() const {
        Fp29 v = *this;
#pragma unroll
        for (int k = LOG; k >= 0; k--) {
            int64_t c = 0;
            uint32_t tl[N];
#pragma unroll
            for (int i = 0; i < N - 1; i++) { int64_t x = (int64_t)v.l[i] - ((int64_t)mod29(i) << k) + c; tl[i] = (uint32_t)x & MASK; c = x >> B; }
            int64_t top = (int64_t)v.l[N - 1] - ((int64_t)mod29(N - 1) << k) + c;
            if (top >= 0) {
#pragma unroll
                for (int i = 0; i < N - 1; i++) v.l[i] = tl[i];
                v.l[N - 1] = (uint32_t)top;
            }
        }
        return v;
    }
    // v < 32 p with normalized limbs -> v - q p in [0, 1.07 p) for an UNDER-estimate q of floor(v / p) taken from the top limb alone (round 5: replaces the 16 p, 8 p, 4 p
    // rounds of canonical<4> at the NTT's stores -- ~55 instructions instead of ~190).  l_8 = floor(v / 2^232) < 2^28 and p_8 = floor(p / 2^232) >= 2^20; with
    // RECIP = floor(2^32 / (p_8 + 1)) and q = floor(l_8 RECIP / 2^32):
    //   q <= l_8 / (p_8 + 1) <= v / p                                  (never an over-estimate: the remainder is >= 0)
    //   q >  l_8 / (p_8 + 1) - l_8 / 2^32 - 1 >= l_8 / (p_8 + 1) - 17/16
    //   v / p < (l_8 + 1) / p_8   =>   v / p - q < l_8 / (p_8 (p_8 + 1)) + 1 / p_8 + 17/16 < 2^-15 + 2^-20 + 1.0625        (l_8 <= 32 p_8 + 31)
    // so the remainder is below 1.07 p < 2 p, which k_ntt_pass relies on when it packs the value into 8 x 32-bit words between passes (2 p < 2^256 is asserted below:
    // a bound of 3 p -- what this comment claimed in round 5 -- would NOT fit for BLS12-381's scalar field).  The signed carry chain re-normalizes the limbs (the top limb
    // keeps what is left).
    ZK_HD Fp29 reduce_by_top_limb() const {
        constexpr uint32_t RECIP = (uint32_t)((1ull << 32) / ((uint64_t)mod29(N - 1) + 1));
        static_assert(mod29(N - 1) >= (1u << 20), "the estimate's error bound assumes a modulus of at least 253 bits");
        static_assert(P::mod(P::N - 1) < (1u << 31), "the callers pack a remainder < 2 p into 8 x 32-bit words: needs 2 p < 2^256");
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t q = __umulhi(l[N - 1], RECIP);
#else
        const uint32_t q = (uint32_t)(((uint64_t)l[N - 1] * RECIP) >> 32);
#endif
        Fp29 r;
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { const int64_t x = (int64_t)l[i] - (int64_t)((uint64_t)q * mod29(i)) + c; r.l[i] = (uint32_t)x & MASK; c = x >> B; }
        r.l[N - 1] = (uint32_t)((int64_t)l[N - 1] - (int64_t)((uint64_t)q * mod29(N - 1)) + c);
        return r;
    }
    // w (standard form w R, canonical) -> w R' as reduced-radix limbs: R' / R = 2^5, five modular doublings of the raw integer
    ZK_HD static Fp29 twiddle_from_std(const Fp<P> &w) {
        Fp<P> v = w;
        for (int i = 0; i < 261 - 32 * P::N; i++) v = v.dbl();
        return split(v.l);
    }
    ZK_HD static Fp29 from_std_relimb(const Fp<P> &a) { return split(a.l); }
    template <int LOG> ZK_HD Fp<P> to_std_relimb() const { Fp<P> r; canonical<LOG>().pack(r.l); return r; }
};

}  // namespace zk
