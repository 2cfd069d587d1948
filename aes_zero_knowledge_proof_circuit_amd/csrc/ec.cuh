// csrc/ec.cuh -- short-Weierstrass (a = 0) G1 arithmetic for BLS12-377 / BLS12-381, host + gfx950 device.
//
// Bucket accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a mixed add is 8M + 2S with no inversion and no special doubling formula for the accumulator
// (EFD madd-2008-s / add-2008-s / dbl-2008-s-1).  Affine points use (0, 0) as the point at infinity
// (never on y^2 = x^3 + b since b != 0).  Replaces ark-ec 0.3.0 short_weierstrass_jacobian
// (Cargo.lock:118) for this path.
#pragma once
#include "ff.cuh"
#include "ff28.cuh"

// the reduced-radix base field the MSM kernels compute in: 14 unsigned 28-bit limbs (ff28.cuh, 378 multiply-accumulates per product).  A 13 x 30-bit signed-limb
// variant measured equal on MI355X (profiles/r02_msm_radix30.md: k_accumulate is bound by its TOTAL VALU instruction count) and was removed in round 4.
namespace zk {
template <class P> using FpMsm = Fp28<P>;
}

// The group operations are deliberately NOT inlined on the device: each is 9-14 Fq products (~300 VALU instructions
// apiece), so a call costs <1% while keeping kernels (and hipcc's compile time) bounded.
#if defined(__HIPCC__)
#define ZK_EC_FN __host__ __device__ __noinline__
#else
#define ZK_EC_FN
#endif

namespace zk {

template <class Fq>
struct Affine {
    Fq x, y;
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    ZK_HD static Affine inf() { Affine a; a.x = Fq::zero(); a.y = Fq::zero(); return a; }
    ZK_HD Affine neg() const { Affine a; a.x = x; a.y = y.neg(); return a; }
};

template <class Fq>
struct XYZZ {
    Fq x, y, zz, zzz;
    ZK_HD bool is_inf() const { return zz.is_zero(); }
    ZK_HD static XYZZ inf() { XYZZ p; p.x = Fq::zero(); p.y = Fq::zero(); p.zz = Fq::zero(); p.zzz = Fq::zero(); return p; }
    ZK_HD static XYZZ from_affine(const Affine<Fq> &a) {
        if (a.is_inf()) return inf();
        XYZZ p; p.x = a.x; p.y = a.y; p.zz = Fq::one(); p.zzz = Fq::one(); return p;
    }
    ZK_HD XYZZ neg() const { XYZZ p = *this; p.y = y.neg(); return p; }

    // dbl-2008-s-1
    ZK_EC_FN XYZZ dbl() const {
        if (is_inf()) return *this;
        Fq u = y.dbl(), v = u.sqr(), w = u * v, s = x * v;
        Fq xx = x.sqr(), m = xx.dbl() + xx;
        XYZZ r;
        r.x = m.sqr() - s.dbl();
        r.y = m * (s - r.x) - w * y;
        r.zz = v * zz;
        r.zzz = w * zzz;
        return r;
    }
    // madd-2008-s (this += affine)
    ZK_EC_FN void madd(const Affine<Fq> &q) {
        if (q.is_inf()) return;
        if (is_inf()) { *this = from_affine(q); return; }
        Fq u2 = q.x * zz, s2 = q.y * zzz;
        Fq p = u2 - x, r = s2 - y;
        if (p.is_zero()) {
            if (r.is_zero()) { *this = from_affine(q).dbl(); } else { *this = inf(); }
            return;
        }
        Fq pp = p.sqr(), ppp = p * pp, qq = x * pp;
        Fq x3 = r.sqr() - ppp - qq.dbl();
        y = r * (qq - x3) - y * ppp;
        x = x3;
        zz = zz * pp;
        zzz = zzz * ppp;
    }
    // add-2008-s (this += other)
    ZK_EC_FN void add(const XYZZ &o) {
        if (o.is_inf()) return;
        if (is_inf()) { *this = o; return; }
        Fq u1 = x * o.zz, u2 = o.x * zz, s1 = y * o.zzz, s2 = o.y * zzz;
        Fq p = u2 - u1, r = s2 - s1;
        if (p.is_zero()) {
            if (r.is_zero()) { *this = dbl(); } else { *this = inf(); }
            return;
        }
        Fq pp = p.sqr(), ppp = p * pp, qq = u1 * pp;
        Fq x3 = r.sqr() - ppp - qq.dbl();
        y = r * (qq - x3) - s1 * ppp;
        x = x3;
        zz = zz * o.zz * pp;
        zzz = zzz * o.zzz * ppp;
    }
    // host-side / rare: to affine (one inversion)
    ZK_EC_FN Affine<Fq> to_affine() const {
        if (is_inf()) return Affine<Fq>::inf();
        Fq zi3 = zzz.inverse();          // 1/ZZZ
        Fq zi2 = (zi3 * zz).sqr();       // (ZZ/ZZZ)^2 = 1/ZZ   (since ZZ^3 = ZZZ^2)
        Affine<Fq> a; a.x = x * zi2; a.y = y * zi3; return a;
    }
    // k: canonical little-endian 32-bit limbs
    ZK_HD XYZZ mul_raw(const uint32_t *k, int klimbs) const {
        XYZZ acc = inf();
        int top = klimbs * 32 - 1;
        while (top >= 0 && !((k[top / 32] >> (top % 32)) & 1)) top--;
        for (int i = top; i >= 0; i--) {
            acc = acc.dbl();
            if ((k[i / 32] >> (i % 32)) & 1) acc.add(*this);
        }
        return acc;
    }
};

template <class Fq, class Fr>
ZK_HD XYZZ<Fq> mul_fr(const XYZZ<Fq> &p, const Fr &k) {
    uint32_t raw[Fr::N];
    k.to_raw(raw);
    return p.mul_raw(raw, Fr::N);
}

template <class Fq>
ZK_HD bool on_curve(const Affine<Fq> &a, const Fq &b) {
    if (a.is_inf()) return true;
    return a.y.sqr() == a.x.sqr() * a.x + b;
}

// affine point in the reduced-radix form the Weierstrass-law accumulate kernel consumes (FpMsm): 2 x 14 x 28-bit limbs = 112 B (ff28.cuh), (0,0) = infinity
template <class P>
struct Affine28 {
    FpMsm<P> x, y;
    ZK_HD bool is_inf() const { return (x.l[0] | y.l[0]) == 0 && x.limbs_zero() && y.limbs_zero(); }      // one-word pre-test: the full OR only when both low limbs are zero
    ZK_HD static Affine28 from_std(const Affine<Fp<P>> &a) { Affine28 r; r.x = FpMsm<P>::from_std(a.x); r.y = FpMsm<P>::from_std(a.y); return r; }
    ZK_HD Affine<Fp<P>> to_std() const { Affine<Fp<P>> r; r.x = x.to_std(); r.y = y.to_std(); return r; }
};

// per-curve bundles
struct Bls377 {
    using Fr = Fr377; using Fq = Fq377; using FqP = Fq377P;
    static constexpr int ID = 377;
    ZK_HD static Fq b() { Fq v; constexpr uint32_t t[12] = FQ377_ONE_INIT; for (int i = 0; i < 12; i++) v.l[i] = t[i]; return v; }   // b = 1
};
struct Bls381 {
    using Fr = Fr381; using Fq = Fq381; using FqP = Fq381P;
    static constexpr int ID = 381;
    ZK_HD static Fq b() { Fq v = Fq::one(); v = v.dbl().dbl(); return v; }   // b = 4
};

}  // namespace zk
