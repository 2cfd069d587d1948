// csrc/ff28.cuh -- reduced-radix ("lazy carry") Montgomery arithmetic for the 377/381-bit base fields on gfx950.
//
// Why: a 12x32-bit CIOS product compiles to 276 v_mad_u64_u32 + ~1000 full-rate fix-up instructions (64-bit adds and
// register-pair moves that only propagate carries); on MI355X v_mad_u64_u32 issues at ~0.4x the simple-VALU rate, so the
// fix-ups cost more than the multiplies (tools/ubench/rates.hip).  With 14 limbs of 28 bits a limb product is < 2^56 and a 64-bit
// accumulator absorbs all 28 products of a column, so the whole product is v_mad_u64_u32 with the accumulator as the addend:
// 2 * 14^2 = 392 multiply-accumulates and ~100 shifts/masks, no carry chain.
//
// Representation: value = sum l[i] 2^(28 i), limbs normalized to < 2^28 after every operation, Montgomery radix R' = 2^392.
// "Almost Montgomery": R' > 2^13 p, so for inputs < 64 p the product is < 1.2 p and NO conditional subtraction is needed; additions and
// subtractions let values grow (sub adds a multiple of p) and callers bound them statically (see madd28 in kernels_msm.hip).
// Conversion from the library-wide 12x32 form (R = 2^384): split limbs, multiply by 2^8 R' mod p; back: multiply by R mod p... (to_std).
#pragma once
#include "ff.cuh"

namespace zk {

template <class P>   // P = Fq377P / Fq381P (12 x 32-bit parameter pack)
struct Fp28 {
    static constexpr int N = 14;
    static constexpr uint32_t MASK = (1u << 28) - 1;
    uint32_t l[N];

    // ---- constants derived at compile time from the 32-bit parameter pack
    ZK_HD static constexpr uint32_t mod28(int i) {           // limb i of p in radix 2^28
        int bit = 28 * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = (w < P::N ? (uint64_t)P::mod(w) : 0) | ((w + 1 < P::N ? (uint64_t)P::mod(w + 1) : 0) << 32);
        return (uint32_t)(two >> sh) & MASK;
    }
    static constexpr uint32_t PINV = P::INV & MASK;   // -p^-1 mod 2^28 = the low 28 bits of -p^-1 mod 2^32

    ZK_HD static Fp28 zero() { Fp28 r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
    ZK_HD bool limbs_zero() const { uint32_t o = 0; for (int i = 0; i < N; i++) o |= l[i]; return o == 0; }
    // v is a product (< 1.2 p): v == 0 (mod p)  <=>  v in {0, p}
    ZK_HD static bool product_is_zero(const Fp28 &v) {
        // one-limb pre-test first (hot path of k_accumulate: every VALU instruction counts): v in {0, p} needs limb 0 in {0, p_0}
        if (v.l[0] != 0u && v.l[0] != mod28(0)) return false;
        uint32_t z0 = 0, zp = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { z0 |= v.l[i]; zp |= v.l[i] ^ mod28(i); }
        return z0 == 0 || zp == 0;
    }

    // split a 12x32 little-endian integer into 28-bit limbs (no arithmetic)
    ZK_HD static Fp28 split(const uint32_t *w) {
        Fp28 r;
#pragma unroll
        for (int i = 0; i < N; i++) {
            int bit = 28 * i, k = bit >> 5, sh = bit & 31;
            uint64_t two = (k < P::N ? (uint64_t)w[k] : 0) | ((k + 1 < P::N ? (uint64_t)w[k + 1] : 0) << 32);
            r.l[i] = (uint32_t)(two >> sh) & MASK;
        }
        return r;
    }
    // pack normalized limbs (value < 2^384) back into 12x32
    ZK_HD void pack(uint32_t *w) const {
#pragma unroll
        for (int k = 0; k < P::N; k++) {
            int bit = 32 * k, i = bit / 28, sh = bit % 28;     // word k = bits [32k, 32k+32)
            uint64_t v = (uint64_t)l[i] >> sh;
            int have = 28 - sh;
            if (i + 1 < N) v |= (uint64_t)l[i + 1] << have;
            if (have + 28 < 32 && i + 2 < N) v |= (uint64_t)l[i + 2] << (have + 28);
            w[k] = (uint32_t)v;
        }
    }
    // carry-propagate signed 64-bit limb values into normalized limbs (top limb keeps the excess)
    ZK_HD static Fp28 normalize(const int64_t *t) {
        Fp28 r;
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { int64_t v = t[i] + c; r.l[i] = (uint32_t)v & MASK; c = v >> 28; }
        r.l[N - 1] = (uint32_t)(t[N - 1] + c);
        return r;
    }
    // a + b (no modular reduction; value grows)
    ZK_HD Fp28 operator+(const Fp28 &b) const {
        Fp28 r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint32_t v = l[i] + b.l[i] + c; r.l[i] = v & MASK; c = v >> 28; }
        r.l[N - 1] = l[N - 1] + b.l[N - 1] + c;
        return r;
    }
    ZK_HD Fp28 dbl() const { return *this + *this; }
    // ---- lazy limb arithmetic: NO carry propagation, limbs may exceed 28 bits.  Valid as ONE operand of a product whose other operand is normalized:
    // a column then holds at most 14 limb products of (2^30.4 x 2^28) plus the reduction's 14 x 2^56, below 2^63; two lazy operands only where the caller bounds
    // 14 x (limb bound a) x (limb bound b) + 14 x 2^56 below 2^64 (te28.cuh lists the operands that use it, with their bounds).
    ZK_HD Fp28 add_lazy(const Fp28 &b) const { Fp28 r; for (int i = 0; i < N; i++) r.l[i] = l[i] + b.l[i]; return r; }
    ZK_HD Fp28 dbl_lazy() const { Fp28 r; for (int i = 0; i < N; i++) r.l[i] = l[i] << 1; return r; }
    // b normalized; limbs <= (K + 2) 2^28.  On the device limbs 0..12 are ONE instruction each: kp_spread >= 2^28 - 1 >= b_i there, so l + (kp - b) = |kp - b| + l = v_sad_u32
    // (the top limb keeps the two-instruction form: b's excess sits there)
    template <int K> ZK_HD Fp28 sub_lazy(const Fp28 &b) const {
        Fp28 r;
#pragma unroll
        for (int i = 0; i < N; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (i < N - 1) { asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r.l[i]) : "s"(kp_spread<K>(i)), "v"(b.l[i]), "v"(l[i])); continue; }
#endif
            r.l[i] = l[i] + kp_spread<K>(i) - b.l[i];
        }
        return r;
    }
    // limb i of K p (normalized; the top limb keeps the excess) and the same with the borrows pre-distributed:
    // sum c_i 2^(28 i) = K p with c_i >= 2^28 - 1 below the top, so a_i - b_i + c_i never goes negative for a normalized b
    template <int K> ZK_HD static constexpr uint32_t kp_limb(int i) {
        uint64_t c = 0, v = 0;
        for (int j = 0; j <= i; j++) { v = (uint64_t)K * mod28(j) + c; c = v >> 28; }
        return i == N - 1 ? (uint32_t)v : (uint32_t)v & MASK;
    }
    template <int K> ZK_HD static constexpr uint32_t kp_spread(int i) {
        return i == 0 ? kp_limb<K>(0) + (1u << 28) : (i == N - 1 ? kp_limb<K>(i) - 1u : kp_limb<K>(i) + MASK);
    }
    // a - b + K p, K chosen by the caller so that K p >= b (keeps the result non-negative); 32-bit unsigned arithmetic only, no borrow chain
    template <int K>
    ZK_HD Fp28 sub(const Fp28 &b) const {
        Fp28 r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint32_t v = l[i] + kp_spread<K>(i) - b.l[i] + c; r.l[i] = v & MASK; c = v >> 28; }
        r.l[N - 1] = l[N - 1] + kp_spread<K>(N - 1) - b.l[N - 1] + c;
        return r;
    }

    // one row of the Montgomery reduction: make limb i of t a multiple of 2^28 by adding m p, then carry it into limb i + 1.
    // For p = 1 (mod 2^28) (BLS12-377's q: -p^-1 = -1, limb 0 of p = 1) m = -t_i mod 2^28 and the low limb needs no multiply and no
    // 64-bit add of m: t_i + m clears the low 28 bits and carries exactly one unit when they were non-zero, i.e. (t_i + 2^28 - 1) >> 28.
    ZK_HD static void reduce_row(uint64_t *t, int i) {
        if constexpr (mod28(0) == 1u && PINV == MASK) {
            // u = t_i + 2^28 - 1: its high part is the carry (t_i + m) >> 28 (one unit exactly when the low 28 bits of t_i are non-zero) and its low 28 bits are
            // those of t_i - 1, so m = -t_i mod 2^28 = ~u mod 2^28
            const uint64_t u = t[i] + MASK;
            const uint32_t m = ~(uint32_t)u & MASK;
#pragma unroll
            for (int j = 1; j < N; j++) t[i + j] += (uint64_t)m * mod28(j);
            t[i + 1] += u >> 28;
        } else {
            uint32_t m = (((uint32_t)t[i] & MASK) * PINV) & MASK;
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (uint64_t)m * mod28(j);
            t[i + 1] += t[i] >> 28;
        }
    }

    // almost-Montgomery product: row-wise operand scanning, 64-bit column accumulators, no carry chain
    ZK_HD Fp28 operator*(const Fp28 &b) const {
        uint64_t t[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (uint64_t)l[j] * b.l[i];
            reduce_row(t, i);
        }
        Fp28 r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint64_t v = t[N + i] + c; r.l[i] = (uint32_t)v & MASK; c = v >> 28; }
        r.l[N - 1] = (uint32_t)(t[2 * N - 1] + c);
        return r;
    }
    // The same product with the rows' "+ 2^28 - 1" riding on the multiply-accumulate chains: columns 0..13 START at `bias`, which must hold 2^28 - 1 in a register pair
    // the compiler cannot see through and that was defined before the operands (hot_loop_bias() at kernel entry).  A known constant is re-associated to the end
    // of every column's sum and costs a 64-bit add per row -- 14 per product; an unknown early value stays the chain's first addend.
    ZK_HD static Fp28 mul_biased(const Fp28 &a, const Fp28 &b, uint64_t bias) {
        static_assert(mod28(0) == 1u && PINV == MASK, "written for p = 1 (mod 2^28)");
        uint64_t t[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; i++) t[i] = i < N ? bias : 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (uint64_t)a.l[j] * b.l[i];
            const uint64_t u = t[i];                            // = column + 2^28 - 1 (see reduce_row)
            const uint32_t m = ~(uint32_t)u & MASK;
#pragma unroll
            for (int j = 1; j < N; j++) t[i + j] += (uint64_t)m * mod28(j);
            t[i + 1] += u >> 28;
        }
        Fp28 r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint64_t v = t[N + i] + c; r.l[i] = (uint32_t)v & MASK; c = v >> 28; }
        r.l[N - 1] = (uint32_t)(t[2 * N - 1] + c);
        return r;
    }
    ZK_HD static uint64_t hot_loop_bias() {
        uint64_t v = MASK;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(v));
#endif
        return v;
    }
    // (a b + c d) / R' with ONE Montgomery reduction: both limb products accumulate into the same 64-bit columns (28 products of < 2^56
    // plus the 14 reduction products per column stay below 2^62), which saves the 196 multiplies of a second reduction.
    // Inputs normalized; the result is < ((a b + c d) / R') + p, e.g. < 1.01 p for a b + c d < 64 p^2.
    ZK_HD static Fp28 fma2(const Fp28 &a, const Fp28 &b, const Fp28 &c, const Fp28 &d) {
        uint64_t t[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (uint64_t)a.l[j] * b.l[i];
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (uint64_t)c.l[j] * d.l[i];
            reduce_row(t, i);
        }
        Fp28 r;
        uint64_t cy = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint64_t v = t[N + i] + cy; r.l[i] = (uint32_t)v & MASK; cy = v >> 28; }
        r.l[N - 1] = (uint32_t)(t[2 * N - 1] + cy);
        return r;
    }
    // squaring: the 105 distinct limb products (cross terms doubled by pre-doubling one operand) instead of 196; same reduction
    ZK_HD Fp28 sqr() const {
        uint64_t t[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; i++) t[i] = 0;
        uint32_t d[N];
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = l[i] << 1;                      // < 2^29 (top limb of a bounded value stays far below 2^31)
#pragma unroll
        for (int i = 0; i < N; i++) {
            t[2 * i] += (uint64_t)l[i] * l[i];
#pragma unroll
            for (int j = i + 1; j < N; j++) t[i + j] += (uint64_t)d[i] * l[j];
        }
#pragma unroll
        for (int i = 0; i < N; i++) {
            reduce_row(t, i);
        }
        Fp28 r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { uint64_t v = t[N + i] + c; r.l[i] = (uint32_t)v & MASK; c = v >> 28; }
        r.l[N - 1] = (uint32_t)(t[2 * N - 1] + c);
        return r;
    }

    // fully reduce a value < 64 p to the canonical range [0, p) (used only at conversions and for zero tests)
    ZK_HD Fp28 canonical() const {
        Fp28 v = *this;
        // subtract p while v >= p: at most 64 rounds would be wasteful; peel powers of two: 32p, 16p, ... p
#pragma unroll
        for (int k = 5; k >= 0; k--) {
            // t = v - (2^k) p
            int64_t c = 0;
            uint32_t tl[N];
#pragma unroll
            for (int i = 0; i < N - 1; i++) { int64_t x = (int64_t)v.l[i] - ((int64_t)mod28(i) << k) + c; tl[i] = (uint32_t)x & MASK; c = x >> 28; }
            int64_t top = (int64_t)v.l[N - 1] - ((int64_t)mod28(N - 1) << k) + c;
            if (top >= 0) {
#pragma unroll
                for (int i = 0; i < N - 1; i++) v.l[i] = tl[i];
                v.l[N - 1] = (uint32_t)top;
            }
        }
        return v;
    }
    ZK_HD bool is_zero_mod_p() const { return canonical().limbs_zero(); }

    // constants: 2^8 R' mod p ... obtained at run time from the 32-bit field (host + device) to avoid a second generated table
    ZK_HD static Fp28 from_std(const Fp<P> &a) {
        // a = x R (R = 2^384).  want x R' (R' = 2^392) = a 2^8.  mul28(split(a), c) = a c / R'  with c = 2^8 R'^1 ... = 2^400 mod p
        Fp28 s = split(a.l);
        return s * k_2_400();
    }
    ZK_HD Fp<P> to_std() const {
        // w = x R'.  mul28(w, 2^384 mod p) = x 2^384 = standard Montgomery form; then canonicalize and repack
        Fp28 y = (*this * k_2_384()).canonical();
        Fp<P> r;
        y.pack(r.l);
        return r;
    }
    // 2^400 mod p and 2^384 mod p as 28-bit limb integers, computed from Fp<P> (R = 2^384: one() = 2^384 mod p)
    ZK_HD static Fp28 k_2_384() { Fp<P> o = Fp<P>::one(); return split(o.l); }
    ZK_HD static Fp28 k_one() { return k_2_392(); }
    ZK_HD static Fp28 k_2_392() {               // the Montgomery one of this representation (R' mod p)
        Fp<P> v = Fp<P>::one();
        for (int i = 0; i < 8; i++) v = v.dbl();
        return split(v.l);
    }
    ZK_HD static Fp28 k_2_400() {
        // 2^400 mod p = (2^384 mod p) * 2^16 mod p : in Montgomery form of Fp<P>: from_u64(2^16) * one ... compute raw integer:
        Fp<P> v = Fp<P>::one();                 // raw limbs = 2^384 mod p  (as an integer)
        for (int i = 0; i < 16; i++) v = v.dbl();   // integer doubling mod p of the raw limbs (Fp add is plain modular add on the limbs)
        return split(v.l);
    }
};

using Fq377x28 = Fp28<Fq377P>;
using Fq381x28 = Fp28<Fq381P>;

}  // namespace zk
