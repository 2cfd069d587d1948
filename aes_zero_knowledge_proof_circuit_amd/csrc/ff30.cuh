// csrc/ff30.cuh -- signed reduced-radix Montgomery arithmetic for the 377/381-bit base fields on gfx950: 13 limbs of 30 bits.
//
// Successor of ff28.cuh (14 x 28-bit unsigned limbs, 378 v_mad_u64_u32 per product) for the MSM kernels: one limb less is 13^2 + 12 * 13 = 325
// multiply-accumulates per product (-14 %).  What makes 30-bit limbs fit a 64-bit column accumulator is the SIGN: limbs are centred,
// l_i in [-2^29, 2^29), so a limb product is < 2^58 in magnitude and the 13 products of a column plus the 13 products of the Montgomery
// reduction stay below 2^62.7 (v_mad_i64_i32, the accumulator is the addend: no carry chain inside a product).  Unsigned 30-bit limbs
// would overflow (26 x 2^60).
//
// Representation: value = sum l[i] 2^(30 i) (an integer that may be NEGATIVE), limbs 0..11 centred after every operation, limb 12 keeps the excess.
// Montgomery radix R' = 2^390 > 2^13 p.  The reduction multipliers m_i are centred as well, so |M| <= R'/2 and a product of inputs below 8 p in magnitude
// lies in (-0.51 p, 0.51 p): no conditional subtraction anywhere, and -- unlike ff28 -- subtraction needs no multiple of p added (limbs are signed), and a
// product that is 0 mod p IS the integer 0 (all limbs zero).  sub<K> keeps ff28's signature; K is ignored.
#pragma once
#include "ff.cuh"

namespace zk {

template <class P>   // P = Fq377P / Fq381P (12 x 32-bit parameter pack)
struct Fp30 {
    static constexpr int N = 13;
    static constexpr int BITS = 30;
    static constexpr uint32_t MASK = (1u << 30) - 1;
    static constexpr int32_t HALF = 1 << 29;
    int32_t l[N];

    // ---- constants derived at compile time from the 32-bit parameter pack
    ZK_HD static constexpr uint32_t modu(int i) {            // unsigned limb i of p in radix 2^30
        int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = (w < P::N ? (uint64_t)P::mod(w) : 0) | ((w + 1 < P::N ? (uint64_t)P::mod(w + 1) : 0) << 32);
        return (uint32_t)(two >> sh) & MASK;
    }
    ZK_HD static constexpr int32_t modl(int i) {             // centred limb i of p
        int64_t carry = 0, v = 0;
        for (int j = 0; j <= i; j++) {
            int64_t x = (int64_t)modu(j) + carry;
            if (j < N - 1 && x >= HALF) { v = x - ((int64_t)1 << 30); carry = 1; } else { v = x; carry = 0; }
        }
        return (int32_t)v;
    }
    static constexpr uint32_t PINV = P::INV & MASK;          // -p^-1 mod 2^30

    ZK_HD static int32_t center(uint32_t x) { return (int32_t)(x << 2) >> 2; }      // low 30 bits as a value in [-2^29, 2^29)

    ZK_HD static Fp30 zero() { Fp30 r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
    ZK_HD bool limbs_zero() const { int32_t o = 0; for (int i = 0; i < N; i++) o |= l[i]; return o == 0; }
    // v is a product (|v| < 0.51 p): v == 0 (mod p)  <=>  v == 0  <=>  all limbs zero (centred digits are unique)
    ZK_HD static bool product_is_zero(const Fp30 &v) { return v.limbs_zero(); }

    // carry-propagate signed 64-bit limb values into centred limbs (top limb keeps the excess)
    ZK_HD static Fp30 normalize(const int64_t *t) {
        Fp30 r;
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { int64_t v = t[i] + c; r.l[i] = center((uint32_t)v); c = (v + HALF) >> 30; }     // v - centre(v) = floor((v + 2^29) / 2^30) 2^30
        r.l[N - 1] = (int32_t)(t[N - 1] + c);
        return r;
    }
    // a + b, a - b: 32-bit limb arithmetic (|limb| <= 2^29 each, the sum fits), one centring carry chain; values simply grow
    ZK_HD Fp30 operator+(const Fp30 &b) const {
        Fp30 r;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { int32_t v = l[i] + b.l[i] + c; r.l[i] = center((uint32_t)v); c = (v + HALF) >> 30; }
        r.l[N - 1] = l[N - 1] + b.l[N - 1] + c;
        return r;
    }
    ZK_HD Fp30 dbl() const { return *this + *this; }
    template <int K>
    ZK_HD Fp30 sub(const Fp30 &b) const {                    // K (ff28: the multiple of p that kept the difference non-negative) is not needed here
        Fp30 r;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { int32_t v = l[i] - b.l[i] + c; r.l[i] = center((uint32_t)v); c = (v + HALF) >> 30; }
        r.l[N - 1] = l[N - 1] - b.l[N - 1] + c;
        return r;
    }

    // one row of the Montgomery reduction: make column i a multiple of 2^30 by adding m p (m centred), then carry it into column i + 1.
    // For p = 1 (mod 2^30) (BLS12-377's q) -p^-1 = -1: m = -t_i mod 2^30 and limb 0 of p needs no multiply.
    ZK_HD static void reduce_row(int64_t *t, int i) {
        if constexpr (modu(0) == 1u && PINV == MASK) {
            int32_t m = center(0u - (uint32_t)t[i]);
#pragma unroll
            for (int j = 1; j < N; j++) t[i + j] += (int64_t)m * modl(j);
            t[i + 1] += (t[i] + (HALF - 1)) >> 30;           // = (t_i + m) >> 30 (exact): the centred m rounds column i to the NEAREST multiple of 2^30
        } else {
            int32_t m = center((uint32_t)t[i] * PINV);
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (int64_t)m * modl(j);
            t[i + 1] += t[i] >> 30;                          // exact
        }
    }

    // almost-Montgomery product: row-wise operand scanning, signed 64-bit column accumulators, no carry chain
    ZK_HD Fp30 operator*(const Fp30 &b) const {
        int64_t t[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (int64_t)l[j] * b.l[i];
            reduce_row(t, i);
        }
        return normalize(t + N);
    }
    // (a b + c d) / R' with ONE Montgomery reduction.  Two products per column (26 x 2^58) leave no room for the 13 reduction products, so the
    // double-width sum is centred once (one carry chain over the 26 columns) before the reduction rows: ~100 plain instructions instead of 156 multiplies.
    ZK_HD static Fp30 fma2(const Fp30 &a, const Fp30 &b, const Fp30 &c, const Fp30 &d) {
        int64_t t[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (int64_t)a.l[j] * b.l[i];
#pragma unroll
            for (int j = 0; j < N; j++) t[i + j] += (int64_t)c.l[j] * d.l[i];
        }
        {
            int64_t cy = 0;
#pragma unroll
            for (int i = 0; i < 2 * N - 1; i++) { int64_t v = t[i] + cy; t[i] = center((uint32_t)v); cy = (v + HALF) >> 30; }
            t[2 * N - 1] += cy;
        }
#pragma unroll
        for (int i = 0; i < N; i++) reduce_row(t, i);
        return normalize(t + N);
    }
    // squaring: the 91 distinct limb products (cross terms through a pre-doubled operand: |2 l_i| <= 2^30, product < 2^59, <= 6 per column)
    ZK_HD Fp30 sqr() const {
        int64_t t[2 * N];
#pragma unroll
        for (int i = 0; i < 2 * N; i++) t[i] = 0;
        int32_t d[N];
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = l[i] * 2;
#pragma unroll
        for (int i = 0; i < N; i++) {
            t[2 * i] += (int64_t)l[i] * l[i];
#pragma unroll
            for (int j = i + 1; j < N; j++) t[i + j] += (int64_t)d[i] * l[j];
        }
#pragma unroll
        for (int i = 0; i < N; i++) reduce_row(t, i);
        return normalize(t + N);
    }

    // the canonical representative in [0, p) as UNSIGNED 30-bit limbs (conversions and zero tests only); any |value| < 64 p
    ZK_HD void canonical_unsigned(uint32_t *u) const {
        int64_t t[N];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = (int64_t)l[i] + 64 * (int64_t)modl(i);       // + 64 p: positive
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) { int64_t v = t[i] + c; t[i] = v & MASK; c = v >> 30; }
        t[N - 1] += c;
        // peel 64 p, 32 p, ..., p while the value stays non-negative
#pragma unroll
        for (int k = 6; k >= 0; k--) {
            int64_t s[N], b = 0;
#pragma unroll
            for (int i = 0; i < N - 1; i++) { int64_t x = t[i] - ((int64_t)modu(i) << k) + b; s[i] = x & MASK; b = x >> 30; }
            s[N - 1] = t[N - 1] - ((int64_t)modu(N - 1) << k) + b;
            if (s[N - 1] >= 0) {
#pragma unroll
                for (int i = 0; i < N; i++) t[i] = s[i];
            }
        }
#pragma unroll
        for (int i = 0; i < N; i++) u[i] = (uint32_t)t[i];
    }
    ZK_HD bool is_zero_mod_p() const { uint32_t u[N]; canonical_unsigned(u); uint32_t o = 0; for (int i = 0; i < N; i++) o |= u[i]; return o == 0; }

    // split a 12x32 little-endian non-negative integer into centred 30-bit limbs (no modular arithmetic)
    ZK_HD static Fp30 split(const uint32_t *w) {
        int64_t t[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            int bit = 30 * i, k = bit >> 5, sh = bit & 31;
            uint64_t two = (k < P::N ? (uint64_t)w[k] : 0) | ((k + 1 < P::N ? (uint64_t)w[k + 1] : 0) << 32);
            t[i] = (int64_t)((two >> sh) & MASK);
        }
        return normalize(t);
    }
    // pack unsigned 30-bit limbs (value < 2^384) into 12x32
    ZK_HD static void pack(const uint32_t *u, uint32_t *w) {
#pragma unroll
        for (int k = 0; k < P::N; k++) {
            int bit = 32 * k, i = bit / 30, sh = bit % 30;      // word k = bits [32k, 32k+32)
            uint64_t v = (uint64_t)u[i] >> sh;
            int have = 30 - sh;
            if (i + 1 < N) v |= (uint64_t)u[i + 1] << have;
            if (have + 30 < 32 && i + 2 < N) v |= (uint64_t)u[i + 2] << (have + 30);
            w[k] = (uint32_t)v;
        }
    }

    // conversions from / to the library-wide 12x32 Montgomery form (R = 2^384): x R -> x R' needs the factor 2^396 / R', back the factor 2^384 / R'
    ZK_HD static Fp30 from_std(const Fp<P> &a) { return split(a.l) * k_pow2(12); }
    ZK_HD Fp<P> to_std() const {
        Fp30 y = *this * k_pow2(0);
        uint32_t u[N];
        y.canonical_unsigned(u);
        Fp<P> r;
        pack(u, r.l);
        return r;
    }
    // 2^(384 + e) mod p as centred limbs, from Fp<P> (R = 2^384: the raw limbs of one() are 2^384 mod p; Fp::dbl is a modular doubling of the raw limbs)
    ZK_HD static Fp30 k_pow2(int e) {
        Fp<P> v = Fp<P>::one();
        for (int i = 0; i < e; i++) v = v.dbl();
        return split(v.l);
    }
    ZK_HD static Fp30 k_one() { return k_pow2(6); }          // R' mod p = 2^390 mod p: the Montgomery one of this representation
};

}  // namespace zk
