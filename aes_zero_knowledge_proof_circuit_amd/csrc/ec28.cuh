// csrc/ec28.cuh -- G1 group law on reduced-radix coordinates (ff28.cuh) for the MSM kernels; host-callable so that the
// formulas and their value bounds are unit-tested on the CPU (tests/test_ff28_host.py).
#pragma once
#include "ec.cuh"

namespace zk {

// Bucket accumulator in the reduced-radix form (FpMsm = ff28.cuh).  Values are NOT kept below p: the bounds are tracked statically (unsigned limbs: every
// subtraction adds the multiple of p named in sub<K> that keeps it non-negative: x < 6.2 p, y < 3.2 p, zz, zzz < 1.2 p).
template <class P>
struct Acc28 { FpMsm<P> x, y, zz, zzz; };

// mixed add, every product inlined, no carry chains (hot path of k_accumulate).  Returns false for P == +-Q (left to the fix-up pass).
template <class P>
ZK_HD bool madd28(Acc28<P> &a, const Affine28<P> &q) {
    using G = FpMsm<P>;
    G u2 = q.x * a.zz, s2 = q.y * a.zzz;                        // < 1.2 p
    G pd = u2.template sub<7>(a.x), r = s2.template sub<4>(a.y); // < 8.2 p, < 5.2 p
    G pp = pd.sqr();                                            // pd == 0 (mod p)  <=>  pp == 0 (mod p)
    if (G::product_is_zero(pp)) return false;
    G ppp = pd * pp, qq = a.x * pp;
    G x3 = (r.sqr().template sub<2>(ppp)).template sub<3>(qq.dbl());   // r^2 - ppp - 2 qq + 5p  < 6.2 p
    G t = qq.template sub<7>(x3);                                      // < 8.2 p
    G y3 = G::fma2(r, t, G::zero().template sub<4>(a.y), ppp);         // r t + (4p - y1) ppp, one reduction: < 1.01 p
    a.x = x3; a.y = y3;
    a.zz = a.zz * pp;
    a.zzz = a.zzz * ppp;
    return true;
}

// ---- complete group law on reduced-radix XYZZ points (bucket reduction kernels).  Infinity <=> zz limbs all zero.
// Coordinate bounds maintained by every routine: x < 6.2 p, y < 4 p, zz, zzz < 1.2 p.
template <class P>
ZK_HD bool is_zero_product(const FpMsm<P> &v) { return FpMsm<P>::product_is_zero(v); }
template <class P>
ZK_EC_FN void dbl28(Acc28<P> &a) {                        // dbl-2008-s-1
    using G = FpMsm<P>;
    if (a.zz.limbs_zero()) return;
    G u = a.y.dbl(), v = u.sqr(), w = u * v, s = a.x * v;                // u < 8 p
    G xx = a.x.sqr(), m = xx.dbl() + xx;                                 // m < 3.6 p
    G x3 = m.sqr().template sub<3>(s.dbl());                             // < 4.2 p
    G y3 = (m * s.template sub<5>(x3)).template sub<2>(w * a.y);         // < 3.2 p
    a.x = x3; a.y = y3; a.zz = v * a.zz; a.zzz = w * a.zzz;
}
template <class P>
ZK_EC_FN void add28(Acc28<P> &a, const Acc28<P> &b) {     // add-2008-s, complete
    using G = FpMsm<P>;
    if (b.zz.limbs_zero()) return;
    if (a.zz.limbs_zero()) { a = b; return; }
    G u1 = a.x * b.zz, u2 = b.x * a.zz, s1 = a.y * b.zzz, s2 = b.y * a.zzz;
    G pd = u2.template sub<2>(u1), r = s2.template sub<2>(s1);           // < 3.2 p
    G pp = pd.sqr();
    if (is_zero_product<P>(pp)) {
        if (is_zero_product<P>(r.sqr())) dbl28<P>(a);
        else { a.x = G::zero(); a.y = G::zero(); a.zz = G::zero(); a.zzz = G::zero(); }
        return;
    }
    G ppp = pd * pp, qq = u1 * pp;
    G x3 = (r.sqr().template sub<2>(ppp)).template sub<3>(qq.dbl());     // < 6.2 p
    G y3 = G::fma2(r, qq.template sub<7>(x3), G::zero().template sub<2>(s1), ppp);   // r (qq - x3) + (2p - s1) ppp, one reduction: < 1.01 p
    a.x = x3; a.y = y3;
    a.zz = a.zz * b.zz * pp;
    a.zzz = a.zzz * b.zzz * ppp;
}
template <class P>
ZK_HD Acc28<P> neg28(const Acc28<P> &a) { Acc28<P> r = a; r.y = FpMsm<P>::zero().template sub<4>(a.y); return r; }
template <class P>
ZK_HD Acc28<P> inf28() { Acc28<P> r; r.x = FpMsm<P>::zero(); r.y = r.x; r.zz = r.x; r.zzz = r.x; return r; }
template <class P>
ZK_HD XYZZ<Fp<P>> to_std_point(const Acc28<P> &a) {
    XYZZ<Fp<P>> o;
    if (a.zz.limbs_zero()) return XYZZ<Fp<P>>::inf();
    o.x = a.x.to_std(); o.y = a.y.to_std(); o.zz = a.zz.to_std(); o.zzz = a.zzz.to_std();
    return o;
}
template <class P>
ZK_HD Acc28<P> from_std_point(const XYZZ<Fp<P>> &a) {
    if (a.is_inf()) return inf28<P>();
    Acc28<P> o; o.x = FpMsm<P>::from_std(a.x); o.y = FpMsm<P>::from_std(a.y); o.zz = FpMsm<P>::from_std(a.zz); o.zzz = FpMsm<P>::from_std(a.zzz);
    return o;
}


}  // namespace zk
