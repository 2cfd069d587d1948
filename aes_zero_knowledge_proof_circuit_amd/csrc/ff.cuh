// csrc/ff.cuh -- Montgomery prime fields on 32-bit limbs for gfx950 (and the host side of libzkaes).
//
// CDNA4 has no 64x64 integer multiplier in the VALU; the natural primitive is v_mad_u64_u32
// (32x32 + 64 -> 64), so every field element is N 32-bit limbs and the CIOS inner step is written as
// one 64-bit multiply-accumulate.  Elements live in Montgomery form everywhere (device buffers, SRS,
// index polynomials); conversion to canonical integers happens only at serialization and when MSM
// digits are extracted.  Replaces ark-ff 0.3.0 Fp256/Fp384 (Cargo.lock:159) for BLS12-377/381.
#pragma once
#include <stdint.h>
#include <string.h>
#include "consts32.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

// ---- parameter packs: constexpr accessors so that fully unrolled loops fold the limbs to immediates
#define ZK_FIELD_PARAMS(NAME, PFX, NL, NBITS)                                                        \
    struct NAME {                                                                                    \
        static constexpr int N = NL;                                                                 \
        static constexpr int BITS = NBITS;                                                           \
        ZK_HD static constexpr uint32_t mod(int i) { constexpr uint32_t v[NL] = PFX##_P_INIT; return v[i]; }   \
        ZK_HD static constexpr uint32_t one(int i) { constexpr uint32_t v[NL] = PFX##_ONE_INIT; return v[i]; } \
        ZK_HD static constexpr uint32_t r2(int i) { constexpr uint32_t v[NL] = PFX##_R2_INIT; return v[i]; }   \
        static constexpr uint32_t INV = PFX##_INV;                                                   \
    };

// initializer lists (*_INIT) come from the generated consts32.h

ZK_FIELD_PARAMS(Fr377P, FR377, 8, 253)
ZK_FIELD_PARAMS(Fr381P, FR381, 8, 255)
ZK_FIELD_PARAMS(Fq377P, FQ377, 12, 377)
ZK_FIELD_PARAMS(Fq381P, FQ381, 12, 381)

template <class P>
struct Fp {
    using Params = P;
    static constexpr int N = P::N;
    static constexpr int BITS = P::BITS;
    uint32_t l[N];

    ZK_HD static Fp zero() { Fp r; for (int i = 0; i < N; i++) r.l[i] = 0; return r; }
    ZK_HD static Fp one() { Fp r; for (int i = 0; i < N; i++) r.l[i] = P::one(i); return r; }
    ZK_HD static Fp r2() { Fp r; for (int i = 0; i < N; i++) r.l[i] = P::r2(i); return r; }
    ZK_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < N; i++) o |= l[i]; return o == 0; }
    ZK_HD bool operator==(const Fp &b) const { uint32_t o = 0; for (int i = 0; i < N; i++) o |= l[i] ^ b.l[i]; return o == 0; }
    ZK_HD bool operator!=(const Fp &b) const { return !(*this == b); }

    // a >= p ?
    ZK_HD static bool geq_mod(const uint32_t *a) {
        for (int i = N - 1; i >= 0; i--) {
            if (a[i] > P::mod(i)) return true;
            if (a[i] < P::mod(i)) return false;
        }
        return true;
    }
    ZK_HD static void sub_mod_inplace(uint32_t *a) {
        uint64_t br = 0;
        for (int i = 0; i < N; i++) {
            uint64_t d = (uint64_t)a[i] - P::mod(i) - br;
            a[i] = (uint32_t)d;
            br = (d >> 32) & 1;
        }
    }
    ZK_HD Fp operator+(const Fp &b) const {
        Fp r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { c += (uint64_t)l[i] + b.l[i]; r.l[i] = (uint32_t)c; c >>= 32; }
        if (c || geq_mod(r.l)) sub_mod_inplace(r.l);   // moduli here leave >=1 spare bit, c stays 0
        return r;
    }
    ZK_HD Fp operator-(const Fp &b) const {
        Fp r;
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { uint64_t d = (uint64_t)l[i] - b.l[i] - br; r.l[i] = (uint32_t)d; br = (d >> 32) & 1; }
        if (br) {
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < N; i++) { c += (uint64_t)r.l[i] + P::mod(i); r.l[i] = (uint32_t)c; c >>= 32; }
        }
        return r;
    }
    ZK_HD Fp neg() const { return is_zero() ? *this : (zero() - *this); }
    ZK_HD Fp dbl() const { return *this + *this; }

    // CIOS Montgomery product; inner step = one 32x32+64 multiply-accumulate (v_mad_u64_u32)
    ZK_HD Fp operator*(const Fp &b) const {
#if !defined(__HIP_DEVICE_COMPILE__)
        // host code path: same algorithm on 64-bit limbs (the limb pairs have the same little-endian layout)
        constexpr int M = N / 2;
        uint64_t a64[M], b64[M], p64[M], t[M + 2];
        for (int i = 0; i < M; i++) {
            a64[i] = (uint64_t)l[2 * i] | (uint64_t)l[2 * i + 1] << 32;
            b64[i] = (uint64_t)b.l[2 * i] | (uint64_t)b.l[2 * i + 1] << 32;
            p64[i] = (uint64_t)P::mod(2 * i) | (uint64_t)P::mod(2 * i + 1) << 32;
        }
        for (int i = 0; i < M + 2; i++) t[i] = 0;
        uint64_t inv64 = (uint64_t)P::INV;   // -p^-1 mod 2^64 from the 32-bit one (one Newton step)
        inv64 = inv64 * (2 + p64[0] * inv64);
        for (int i = 0; i < M; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; j < M; j++) { c += (unsigned __int128)a64[j] * b64[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[M]; t[M] = (uint64_t)c; t[M + 1] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * inv64;
            c = (unsigned __int128)m * p64[0] + t[0];
            c >>= 64;
            for (int j = 1; j < M; j++) { c += (unsigned __int128)m * p64[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[M]; t[M - 1] = (uint64_t)c; t[M] = t[M + 1] + (uint64_t)(c >> 64);
        }
        Fp r;
        for (int i = 0; i < M; i++) { r.l[2 * i] = (uint32_t)t[i]; r.l[2 * i + 1] = (uint32_t)(t[i] >> 32); }
        if (t[M] || geq_mod(r.l)) sub_mod_inplace(r.l);
        return r;
#else
        uint32_t t[N + 2];
#pragma unroll
        for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < N; j++) { c += (uint64_t)l[j] * b.l[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
            c += t[N]; t[N] = (uint32_t)c; t[N + 1] = (uint32_t)(c >> 32);
            uint32_t m = t[0] * P::INV;
            c = (uint64_t)m * P::mod(0) + t[0];
            c >>= 32;
#pragma unroll
            for (int j = 1; j < N; j++) { c += (uint64_t)m * P::mod(j) + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
            c += t[N]; t[N - 1] = (uint32_t)c; t[N] = t[N + 1] + (uint32_t)(c >> 32);
        }
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = t[i];
        if (t[N] || geq_mod(r.l)) sub_mod_inplace(r.l);
        return r;
#endif
    }
    ZK_HD Fp sqr() const { return *this * *this; }
    ZK_HD Fp &operator+=(const Fp &b) { *this = *this + b; return *this; }
    ZK_HD Fp &operator-=(const Fp &b) { *this = *this - b; return *this; }
    ZK_HD Fp &operator*=(const Fp &b) { *this = *this * b; return *this; }

    // canonical little-endian limbs <-> Montgomery
    ZK_HD static Fp from_raw(const uint32_t *raw) { Fp a; for (int i = 0; i < N; i++) a.l[i] = raw[i]; return a * r2(); }
    ZK_HD void to_raw(uint32_t *raw) const {
        Fp o = zero(); o.l[0] = 1;
        Fp c = *this * o;
        for (int i = 0; i < N; i++) raw[i] = c.l[i];
    }
    ZK_HD static Fp from_u64(uint64_t v) { uint32_t raw[N]; for (int i = 0; i < N; i++) raw[i] = 0; raw[0] = (uint32_t)v; raw[1] = (uint32_t)(v >> 32); return from_raw(raw); }
    ZK_HD static Fp from_i64(int64_t v) { return v >= 0 ? from_u64((uint64_t)v) : from_u64((uint64_t)(-v)).neg(); }
    // e: little-endian 32-bit limbs
    ZK_HD Fp pow(const uint32_t *e, int elimbs) const {
        Fp acc = one();
        int top = elimbs * 32 - 1;
        while (top >= 0 && !((e[top / 32] >> (top % 32)) & 1)) top--;
        for (int i = top; i >= 0; i--) {
            acc = acc.sqr();
            if ((e[i / 32] >> (i % 32)) & 1) acc = acc * *this;
        }
        return acc;
    }
    ZK_HD Fp pow_u64(uint64_t e) const { uint32_t w[2] = {(uint32_t)e, (uint32_t)(e >> 32)}; return pow(w, 2); }
    // Fermat inverse; inverse(0) = 0.  ~1.5 BITS field products in ONE dependent chain (~380 for the scalar fields): kept as the cross-check of inverse()
    ZK_HD Fp inverse_fermat() const {
        uint32_t e[N];
        uint64_t br = 2;
        for (int i = 0; i < N; i++) { uint64_t d = (uint64_t)P::mod(i) - br; e[i] = (uint32_t)d; br = (d >> 32) & 1; }
        return pow(e, N);
    }
    // inverse(0) = 0.  Kaliski's almost-Montgomery inverse (binary extended Euclid: shifts, additions and subtractions only, at most 2 BITS rounds of ~6 N
    // instructions) followed by modular doublings: for the stored value a = x R it ends with r = -a^-1 2^k (mod p), BITS <= k <= 2 BITS, and
    // (p - r) 2^(64 N - k) = x^-1 R^-1 R^2 = x^-1 R.  ~6 x fewer instructions than the Fermat chain -- it is the serial tail of every batch inversion
    // (kernels_poly.hip k_batch_inverse: one lane per workgroup) and of every host-side to_affine.  Invariant u s + v r = p keeps s <= p, r <= 2 p.
    ZK_HD Fp inverse() const {
        if (is_zero()) return zero();
        uint32_t u[N], v[N], r[N], s[N];
        for (int i = 0; i < N; i++) { u[i] = P::mod(i); v[i] = l[i]; r[i] = 0; s[i] = 0; }
        s[0] = 1;
        int k = 0;
        for (;;) {
            uint32_t nz = 0;
            for (int i = 0; i < N; i++) nz |= v[i];
            if (!nz) break;
            if (!(u[0] & 1)) { raw_shr1(u); raw_shl1(s); }
            else if (!(v[0] & 1)) { raw_shr1(v); raw_shl1(r); }
            else if (raw_gt(u, v)) { raw_sub(u, v); raw_shr1(u); raw_add(r, s); raw_shl1(s); }
            else { raw_sub(v, u); raw_shr1(v); raw_add(s, r); raw_shl1(r); }
            k++;
        }
        if (geq_mod(r)) sub_mod_inplace(r);
        Fp x;
        uint64_t br = 0;
        for (int i = 0; i < N; i++) { uint64_t d = (uint64_t)P::mod(i) - r[i] - br; x.l[i] = (uint32_t)d; br = (d >> 32) & 1; }
        for (int i = k; i < 64 * N; i++) x = x + x;
        return x;
    }
    ZK_HD static void raw_shr1(uint32_t *x) { for (int i = 0; i < N - 1; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31); x[N - 1] >>= 1; }
    ZK_HD static void raw_shl1(uint32_t *x) { for (int i = N - 1; i > 0; i--) x[i] = (x[i] << 1) | (x[i - 1] >> 31); x[0] <<= 1; }
    ZK_HD static void raw_add(uint32_t *x, const uint32_t *y) { uint64_t c = 0; for (int i = 0; i < N; i++) { c += (uint64_t)x[i] + y[i]; x[i] = (uint32_t)c; c >>= 32; } }
    ZK_HD static void raw_sub(uint32_t *x, const uint32_t *y) { uint64_t br = 0; for (int i = 0; i < N; i++) { uint64_t d = (uint64_t)x[i] - y[i] - br; x[i] = (uint32_t)d; br = (d >> 32) & 1; } }
    ZK_HD static bool raw_gt(const uint32_t *x, const uint32_t *y) { for (int i = N - 1; i >= 0; i--) { if (x[i] > y[i]) return true; if (x[i] < y[i]) return false; } return false; }
};

using Fr377 = Fp<Fr377P>;
using Fr381 = Fp<Fr381P>;
using Fq377 = Fp<Fq377P>;
using Fq381 = Fp<Fq381P>;

}  // namespace zk
