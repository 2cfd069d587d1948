// csrc/marlin_host.hpp -- small host-side helpers shared by the GPU prover (marlin.cpp) and the host-only codec + verifier (marlin_codec.cpp).
// Nothing here touches a device: marlin_codec.cpp, circuit.cpp and capi_host.cpp build without HIP (the sanitizer / fuzz target of tests/fuzz_host.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>
#include "marlin.hpp"
#include "consts32.h"

namespace zk {
namespace hostx {

// (n beyond 2^62 would shift p to zero and never terminate: sizes that large only arrive in hostile verifying-key bytes, which deserialize_vk_ark rejects long before)
inline size_t next_pow2(size_t n) { if (n > ((size_t)1 << 62)) throw std::length_error("next_pow2: size out of range"); size_t p = 1; while (p < n) p <<= 1; return p; }
inline int log2_exact(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return l; }

// ------------------------------------------------------------------ host field / domain helpers
inline Fr fr_from_limbs(const uint32_t *l) { Fr r; for (int i = 0; i < 8; i++) r.l[i] = l[i]; return r; }
inline Fr domain_gen(int lg) { Fr r = fr_from_limbs(FR377_ROOT_MONT); for (int i = lg; i < FR377_TWO_ADICITY; i++) r = r.sqr(); return r; }
inline Fr eval_vanishing(size_t size, const Fr &x) { return x.pow_u64(size) - Fr::one(); }
inline size_t reindex_by_subdomain(size_t self_size, size_t other_size, size_t index) {   // ark-poly EvaluationDomain::reindex_by_subdomain
    size_t period = self_size / other_size;
    if (index < other_size) return index * period;
    size_t i = index - other_size, x = period - 1;
    return i + (i / x) + 1;
}
inline size_t ahp_max_degree(size_t nc, size_t nv, size_t nnz) {                        // AHPForR1CS::max_degree, zk_bound = 1
    size_t h = next_pow2(std::max(nc, nv)), k = next_pow2(nnz);
    return std::max({2 * h - 1, 3 * h - 1, h, 3 * k - 3});
}
inline G1A mul_affine(const G1A &p, const Fr &k) { return mul_fr(XYZZ<Fq377>::from_affine(p), k).to_affine(); }

// Host-side fixed-base scalar multiplication for the handful of points every proof multiplies by fresh blinding scalars (gamma_g powers for the
// hiding terms, the two Lagrange blinding points): 8-bit windows, 32 x 255 affine multiples built once per key (batch-normalised with one
// inversion), so a product is <= 32 mixed additions instead of a 253-step double-and-add -- about 0.5 ms less host time per product, 22 products per
// proof: what a lone encrypt() call (the reference's criterion metric) waits for between its MSMs.
struct FixedBaseHost {
    std::vector<G1A> t;                     // t[w * 255 + d - 1] = d * 2^(8 w) * base
    void build(const G1A &base) {
        const int NW = 32;
        std::vector<XYZZ<Fq377>> j((size_t)NW * 255);
        XYZZ<Fq377> wb = XYZZ<Fq377>::from_affine(base);
        for (int w = 0; w < NW; w++) {
            XYZZ<Fq377> acc = wb;
            for (int d = 1; d <= 255; d++) { j[(size_t)w * 255 + d - 1] = acc; acc.add(wb); }
            for (int k = 0; k < 8; k++) wb = wb.dbl();
        }
        // batch to affine: x / zz, y / zzz with one inversion of the product of all zzz (zz^3 = zzz^2 => 1/zz = zzz * (1/zzz)^2 * zz ... use 1/zzz and zz)
        t.assign(j.size(), G1A::inf());
        std::vector<Fq377> pre(j.size());
        Fq377 acc = Fq377::one();
        for (size_t i = 0; i < j.size(); i++) { pre[i] = acc; if (!j[i].is_inf()) acc = acc * j[i].zzz; }
        Fq377 inv = acc.inverse();
        for (size_t i = j.size(); i-- > 0;) {
            if (j[i].is_inf()) continue;
            Fq377 zi3 = inv * pre[i];           // 1 / zzz_i
            inv = inv * j[i].zzz;
            Fq377 zi2 = (zi3 * j[i].zz).sqr();  // (zz / zzz)^2 = 1 / zz   (zz^3 = zzz^2)
            t[i].x = j[i].x * zi2; t[i].y = j[i].y * zi3;
        }
    }
    XYZZ<Fq377> mul(const Fr &k) const {
        uint32_t raw[8];
        k.to_raw(raw);
        XYZZ<Fq377> acc = XYZZ<Fq377>::inf();
        for (int w = 0; w < 32; w++) {
            uint32_t d = (raw[w >> 2] >> ((w & 3) * 8)) & 0xff;
            if (d) acc.madd(t[(size_t)w * 255 + d - 1]);
        }
        return acc;
    }
};

// ------------------------------------------------------------------ byte encodings (ark-ff ToBytes / ark-serialize)
struct Bytes {
    std::vector<uint8_t> b;
    void put(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
    void u8(uint8_t v) { b.push_back(v); }
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    template <class Fld> void field(const Fld &a) { uint32_t raw[Fld::N]; a.to_raw(raw); for (int i = 0; i < Fld::N; i++) for (int k = 0; k < 4; k++) b.push_back((uint8_t)(raw[i] >> (8 * k))); }
    void g1_tobytes(const G1A &p) {        // GroupAffine ToBytes: x, y, infinity (zero() = (0, 1, true))
        if (p.is_inf()) { field(Fq377::zero()); field(Fq377::one()); u8(1); }
        else { field(p.x); field(p.y); u8(0); }
    }
    void commitment_tobytes(const Commitment &c) {   // marlin_pc::Commitment ToBytes
        g1_tobytes(c.comm); u8(c.has_shifted ? 1 : 0);
        g1_tobytes(c.has_shifted ? c.shifted : G1A::inf());
    }
    // ark-ec 0.3 GroupAffine::serialize_uncompressed: x, then y with the flags byte (only the infinity bit is ever set; zero() = (0, 1, infinity)) -- 96 bytes,
    // what deserialize_uncompressed / deserialize_unchecked read
    void g1_uncompressed(const G1A &p) {
        const size_t at = b.size();
        if (p.is_inf()) { field(Fq377::zero()); field(Fq377::one()); b[at + 95] |= 1 << 6; }
        else { field(p.x); field(p.y); }
    }
    void g1(const G1A &p, bool uncompressed) { if (uncompressed) g1_uncompressed(p); else g1_compressed(p); }
    void g1_compressed(const G1A &p) {
        uint8_t buf[48] = {0};
        if (p.is_inf()) { buf[47] |= 1 << 6; put(buf, 48); return; }
        uint32_t x[12], y[12], ny[12];
        p.x.to_raw(x); p.y.to_raw(y); p.y.neg().to_raw(ny);
        bool y_gt = false;
        for (int i = 11; i >= 0; i--) if (y[i] != ny[i]) { y_gt = y[i] > ny[i]; break; }
        for (int i = 0; i < 48; i++) buf[i] = (uint8_t)(x[i / 4] >> (8 * (i % 4)));
        if (y_gt) buf[47] |= 1 << 7;
        put(buf, 48);
    }
};

}  // namespace hostx
}  // namespace zk
