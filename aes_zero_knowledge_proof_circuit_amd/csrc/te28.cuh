// csrc/te28.cuh -- BLS12-377 G1 on its twisted Edwards model, for the MSM kernels (host-callable: unit-tested on the CPU by tests/te28_host_check.cpp).
//
// y^2 = x^3 + 1 over Fq (BLS12-377) has a point of order 2 and sqrt(3), hence a Montgomery and a twisted Edwards model; scaled to a = -1 it is
//      -x^2 + y^2 = 1 + d x^2 y^2            (constants and the maps: tools/curve_math.py edwards_377, generated into consts32.h)
// In extended coordinates (X : Y : Z : T), T = X Y / Z, adding a PRECOMPUTED affine point stored as (y - x, y + x, 2 d x y) costs SEVEN field
// products (Hisil-Wong-Carter-Dawson 2008, "add-2008-hwcd-3" with Z2 = 1), no squarings, no special cases -- against 8 products + 2 squarings (and a
// P = +-Q branch) for the XYZZ mixed addition of ec28.cuh.  The SRS is fixed, so its points (and their 2^k multiples in the window tables) are converted
// once at key synthesis; a bucket accumulation is then 7 x 378 = 2,646 v_mad_u64_u32 instead of 3,416.  The bucket reduction uses the full unified
// addition (9 products against 14) and the dedicated doubling (4 squarings + 4 products).
// The law is unified (P + P and P + identity go through the same formulas) but, d being a square, not complete on the whole curve: an addition fails only
// when the sum or difference of its operands has even order, which cannot happen for points of the prime-order subgroup -- where KZG SRS points live.
// Callers must not feed points outside that subgroup to the Edwards path (the generic zkaes_msm entry point stays on the Weierstrass model).
// BLS12-381's G1 has odd cofactor and therefore no such model: it keeps ec28.cuh.
//
// Values follow ff28.cuh's lazy discipline: every coordinate an operation returns is a product (< 1.2 p); te_neg returns 2p - v (< 2 p); all inputs
// of products stay far below the 64 p the almost-Montgomery product accepts.
#pragma once
#include "ec28.cuh"

#if ZK_MSM_RADIX == 28
#define ZK_MSM_EDWARDS 1
#else
#define ZK_MSM_EDWARDS 0
#endif

namespace zk {

// precomputed affine point (y - x, y + x, 2 d x y) in the reduced-radix Montgomery form: 3 x 56 B (+ 24 B of padding to a 64-byte aligned 192-byte record).  The identity is (1, 1, 0): the SRS's "infinity"
// entries (none in practice) need no test in the hot loop.
#ifndef ZK_NIELS_PAD
#define ZK_NIELS_PAD 1          // 1 (default): records padded to 192 B and 64-byte aligned -- exactly three 64-byte sectors per gather instead of 3.5 on average for packed 168-byte
                                // records: k_accumulate 7.31 -> 7.13 ms at 2^22 points, bench +1.1 % (profiles/r03_niels_padding.txt), for 14 % more table memory.  0 = packed.
#endif
#if ZK_NIELS_PAD
template <class P>
struct alignas(64) Niels28 { FpMsm<P> ymx, ypx, td; uint32_t pad[6]; };
#else
template <class P>
struct Niels28 { FpMsm<P> ymx, ypx, td; };
#endif
// extended projective point: same 224 B as the XYZZ accumulator, so the MSM scratch buffers serve both
template <class P>
struct AccTE { FpMsm<P> x, y, z, t; };

#if ZK_MSM_EDWARDS
struct Te377 {
    using P = Fq377P;
    using G = Fp28<P>;
    using F = Fp<P>;
    ZK_HD static G lit28(const uint32_t (&t)[14]) { G r; for (int i = 0; i < 14; i++) r.l[i] = t[i]; return r; }
    ZK_HD static F lit(const uint32_t (&t)[12]) { F r; for (int i = 0; i < 12; i++) r.l[i] = t[i]; return r; }
    ZK_HD static F s_std() { constexpr uint32_t t[12] = FQ377_TE_S_INIT; return lit(t); }          // 1 / sqrt(3)
    ZK_HD static F f_std() { constexpr uint32_t t[12] = FQ377_TE_F_INIT; return lit(t); }          // sqrt(-a')
    ZK_HD static F k2d_std() { constexpr uint32_t t[12] = FQ377_TE_K2D_INIT; return lit(t); }
    ZK_HD static G k2d() { constexpr uint32_t t[14] = FQ377_TE_K2D_28_INIT; return lit28(t); }     // 2 d
    ZK_HD static G si() { constexpr uint32_t t[14] = FQ377_TE_SI_28_INIT; return lit28(t); }       // sqrt(3) = 1 / s
    ZK_HD static G sif() { constexpr uint32_t t[14] = FQ377_TE_SIF_28_INIT; return lit28(t); }     // sqrt(3) f
};

template <class P>
ZK_HD AccTE<P> te_identity() { AccTE<P> r; r.x = FpMsm<P>::zero(); r.t = r.x; r.y = FpMsm<P>::k_one(); r.z = r.y; return r; }
template <class P>
ZK_HD Niels28<P> niels_identity() { Niels28<P> r; r.ymx = FpMsm<P>::k_one(); r.ypx = r.ymx; r.td = FpMsm<P>::zero(); return r; }
// -(x, y) = (-x, y): swap y - x and y + x, negate 2 d x y
template <class P>
ZK_HD Niels28<P> niels_neg(const Niels28<P> &n) { Niels28<P> r; r.ymx = n.ypx; r.ypx = n.ymx; r.td = FpMsm<P>::zero().template sub<2>(n.td); return r; }
// the same for the hot loop: 2p - td without the carry chain (td only ever multiplies the normalized T1)
template <class P>
ZK_HD Niels28<P> niels_neg_lazy(const Niels28<P> &n) { Niels28<P> r; r.ymx = n.ypx; r.ypx = n.ymx; r.td = FpMsm<P>::zero().template sub_lazy<2>(n.td); return r; }

// acc += n, seven products, everything inlined (hot loop of k_accumulate).  acc coordinates are products (normalized limbs, < 1.2 p) in and out; n's
// coordinates normalized (the negated 2dxy may be lazy).
// ZK_TE_LAZY >= 1 propagates carries only where a product needs them (2: D -+ C lazy too).  In te_madd (negated copy of the point, the next gather in a second register set) it was
// measurably SLOWER (3,846 vs 4,013 VALU instructions but 214 vs 198 VGPRs: k_accumulate 7.17 vs 7.08 ms at 2^22 points, profiles/r03_te_lazy.txt); in the hot
// loop's te_madd_signed below (sign as selects, gather mid-addition: 175 VGPRs either way) it pays: 3,550 vs 3,820 instructions
// (profiles/r03_accumulate_instruction_diet.txt).  On; tests/te28_host_check.cpp builds all three.
#ifndef ZK_TE_LAZY
#define ZK_TE_LAZY 2
#endif
template <class P>
ZK_HD void te_madd(AccTE<P> &a, const Niels28<P> &n) {
    using G = FpMsm<P>;
#if ZK_TE_LAZY
    G A = a.y.template sub_lazy<3>(a.x) * n.ymx;     // (Y1 - X1)(y2 - x2)
    G B = a.y.add_lazy(a.x) * n.ypx;                 // (Y1 + X1)(y2 + x2)
    G C = a.t * n.td;                                // T1 2 d x2 y2          (T1 normalized, td possibly lazy)
    G D = a.z.dbl_lazy();                            // 2 Z1                  (Z2 = 1)
    G E = B.template sub<2>(A), H = B.add_lazy(A);   // E normalized, H lazy
    G F = D.template sub_lazy<2>(C), Gg = D + C;     // F lazy (D lazy + 2p - C), G normalized
#else
    G A = a.y.template sub<3>(a.x) * n.ymx;
    G B = (a.y + a.x) * n.ypx;
    G C = a.t * n.td;
    G D = a.z.dbl();
    G E = B.template sub<2>(A), H = B + A;           // < 3.2 p, < 2.4 p
    G F = D.template sub<2>(C), Gg = D + C;          // < 6.2 p, < 5.4 p
#endif
    a.x = E * F; a.y = Gg * H; a.t = E * H; a.z = F * Gg;
}
// acc += (neg ? -n : n) without touching n: -(x, y) = (-x, y) swaps n's first two coordinates and negates the third, i.e. A = (Y1 - X1)(y2 + x2), B = (Y1 + X1)(y2 - x2) and
// C changes sign, which swaps F = D - C and G = D + C.  56 per-limb selects instead of a divergent branch with a 14-limb negation and 42 register moves (ZK_TE_SIGN_SELECT).
#ifndef ZK_TE_PRESWAP
#define ZK_TE_PRESWAP 1
#endif
#ifndef ZK_TE_BIASED
#define ZK_TE_BIASED 1
#endif
#ifndef ZK_TE_SIGN_SELECT
#define ZK_TE_SIGN_SELECT 1
#endif
// `next` (may be null): the record of the lane's NEXT addition, loaded into n as soon as the three products that read n are done -- the gather flies under the remaining
// four products and lands in the registers the current point just vacated (no second register set, no copies at the loop's back edge).
template <bool BIASED, class G>
ZK_HD G mul_maybe_biased(const G &x, const G &y, uint64_t bias) {
    if constexpr (BIASED) return G::mul_biased(x, y, bias);
    else return x * y;
}
// a table record with its first two coordinates swapped when the digit that uses it is negative: the swap costs nothing at the gather (two field addresses), 28 selects after it
template <class P>
ZK_HD Niels28<P> niels_load_signed(const Niels28<P> *rec, bool neg) {
    const FpMsm<P> *f = &rec->ymx;                   // ymx and ypx are adjacent
    Niels28<P> r;
    r.ymx = f[neg ? 1 : 0]; r.ypx = f[neg ? 0 : 1]; r.td = rec->td;
    return r;
}
// PRESWAPPED: n was loaded with niels_load_signed(.., neg) and `next` is loaded with the sign of ITS digit, next_neg
template <class P, bool BIASED = false, bool PRESWAPPED = false>
ZK_HD void te_madd_signed(AccTE<P> &a, Niels28<P> &n, bool neg, const Niels28<P> *next = nullptr, uint64_t bias = 0, bool next_neg = false) {
    using G = FpMsm<P>;
#if ZK_TE_BIASED
    // BIASED: bias = FpMsm<P>::hot_loop_bias() taken at kernel entry (ff28.cuh mul_biased: 14 fewer 64-bit adds per product)
#define ZK_TE_MUL(x, y) mul_maybe_biased<BIASED>(x, y, bias)
#else
#define ZK_TE_MUL(x, y) ((x) * (y))
#endif
    G m1, m2;
    if constexpr (PRESWAPPED) { m1 = n.ymx; m2 = n.ypx; }
    else {
#pragma unroll
        for (int i = 0; i < G::N; i++) { m1.l[i] = neg ? n.ypx.l[i] : n.ymx.l[i]; m2.l[i] = neg ? n.ymx.l[i] : n.ypx.l[i]; }
    }
#if ZK_TE_LAZY
    // carries only where a product needs a normalized operand: Y1 -+ X1 skip the chain (limbs < 2^30.4, their partners are table entries)
    G A = ZK_TE_MUL(a.y.template sub_lazy<3>(a.x), m1);
    G B = ZK_TE_MUL(a.y.add_lazy(a.x), m2);
#else
    G A = ZK_TE_MUL(a.y.template sub<3>(a.x), m1);
    G B = ZK_TE_MUL(a.y + a.x, m2);
#endif
    G C = ZK_TE_MUL(a.t, n.td);
    if (next) {                                      // callers in hot loops pass a non-null pointer on every iteration (straight-line code)
#if defined(__HIP_DEVICE_COMPILE__)
        // pin the gather behind the three products: without the fence the compiler hoists the loads to the top of the iteration, into a second register set
        asm volatile("" : "+v"(A.l[G::N - 1]), "+v"(B.l[G::N - 1]), "+v"(C.l[G::N - 1]) : : "memory");
#endif
        if constexpr (PRESWAPPED) n = niels_load_signed<P>(next, next_neg);
        else n = *next;
    }
#if ZK_TE_LAZY
    // all four factors of the second level lazy.  Limb bounds in units of 2^28: E = B + 2p' - A < 3, H = B + A < 2, D = 2 Z1 < 2, U = D + 2p' - C < 4, V = D + C < 3
    // (2p' = kp_spread<2> < 2 per limb).  The widest products, E U and U V, put 14 x 12 x 2^56 into a column, the reduction 13 x 2^56 more: 181 x 2^56 < 2^64.
    G E = B.template sub_lazy<2>(A), H = B.add_lazy(A);
#if ZK_TE_LAZY >= 2
    G D = a.z.dbl_lazy();
    G U = D.template sub_lazy<2>(C), V = D.add_lazy(C), F, Gg;
#else
    G D = a.z.dbl_lazy();
    G U = D.template sub<2>(C), V = D + C, F, Gg;          // A/B: D -+ C with the carry chain (3,603 instead of 3,550 instructions)
#endif
#else
    G E = B.template sub<2>(A), H = B + A;
    G D = a.z.dbl();
    G U = D.template sub<2>(C), V = D + C, F, Gg;
#endif
#pragma unroll
    for (int i = 0; i < G::N; i++) { F.l[i] = neg ? V.l[i] : U.l[i]; Gg.l[i] = neg ? U.l[i] : V.l[i]; }
    a.x = ZK_TE_MUL(E, F); a.y = ZK_TE_MUL(Gg, H); a.t = ZK_TE_MUL(E, H); a.z = ZK_TE_MUL(F, Gg);
#undef ZK_TE_MUL
}
// a += b, unified (add-2008-hwcd-3): nine products
template <class P>
ZK_EC_FN void te_add(AccTE<P> &a, const AccTE<P> &b) {
    using G = FpMsm<P>;
    G A = a.y.template sub<3>(a.x) * b.y.template sub<3>(b.x);
    G B = (a.y + a.x) * (b.y + b.x);
    G C = (a.t * b.t) * Te377::k2d();
    G D = (a.z * b.z).dbl();
    G E = B.template sub<2>(A), H = B + A;
    G F = D.template sub<2>(C), Gg = D + C;
    a.x = E * F; a.y = Gg * H; a.t = E * H; a.z = F * Gg;
}
// a = 2 a (dbl-2008-hwcd with a = -1): four squarings + four products
template <class P>
ZK_EC_FN void te_dbl(AccTE<P> &a) {
    using G = FpMsm<P>;
    G A = a.x.sqr(), B = a.y.sqr(), C = a.z.sqr().dbl();             // < 1.2 p, < 1.2 p, < 2.4 p
    G S = (a.x + a.y).sqr(), AB = A + B;                              // < 1.2 p, < 2.4 p
    G E = S.template sub<3>(AB);                                      // 2 X Y                     < 4.2 p
    G Gg = B.template sub<2>(A);                                      // D + B = B - A             < 3.2 p
    G F = Gg.template sub<3>(C);                                      // G - C                     < 6.2 p
    G H = G::zero().template sub<3>(AB);                              // D - B = -(A + B)          < 3 p
    a.x = E * F; a.y = Gg * H; a.t = E * H; a.z = F * Gg;
}
template <class P>
ZK_HD AccTE<P> te_neg(const AccTE<P> &a) { AccTE<P> r = a; r.x = FpMsm<P>::zero().template sub<2>(a.x); r.t = FpMsm<P>::zero().template sub<2>(a.t); return r; }

// back to the Weierstrass model WITHOUT an inversion: the XYZZ point (x = X' / ZZ, y = Y' / ZZZ) with Zc = X (Z - Y), ZZ = Zc^2, ZZZ = Zc^3,
//      X' = [sqrt3 (Z + Y) - (Z - Y)] X^2 (Z - Y),   Y' = sqrt3 f Z (Z + Y) X^2 (Z - Y)^2
// (from u = (Z + Y) / (Z - Y), v = f u Z / X, x_w = sqrt3 u - 1, y_w = sqrt3 v).  The identity has X = 0, hence ZZ = 0: XYZZ's infinity.
template <class P>
ZK_HD XYZZ<Fp<P>> te_to_std_point(const AccTE<P> &a) {
    using G = FpMsm<P>;
    G zmy = a.z.template sub<3>(a.y), zpy = a.z + a.y;
    G xz = a.x * zmy;
    G zz = xz.sqr();
    if (zz.is_zero_mod_p()) return XYZZ<Fp<P>>::inf();
    G numx = (Te377::si() * zpy).template sub<6>(zmy);
    XYZZ<Fp<P>> o;
    o.x = ((numx * a.x) * xz).to_std();
    o.y = (((Te377::sif() * a.z) * zpy) * zz).to_std();
    o.zz = zz.to_std();
    o.zzz = (zz * xz).to_std();
    return o;
}

// Weierstrass affine (library-wide 12 x 32 Montgomery form) -> the pieces of the Edwards map that need ONE shared inversion:
//   u = s (x + 1), v = s y, den = v (u + 1);   x_e = f u (u + 1) / den,   y_e = (u - 1) v / den
// den = 0 only for points of order 2 or 4 (never in the prime-order subgroup): reported through *bad.
struct TeMapParts { Fp<Fq377P> u, v, den; };
ZK_HD TeMapParts te_map_parts(const Affine<Fp<Fq377P>> &p) {
    using F = Fp<Fq377P>;
    TeMapParts m;
    F s = Te377::s_std();
    m.u = s * (p.x + F::one());
    m.v = s * p.y;
    m.den = m.v * (m.u + F::one());
    return m;
}
ZK_HD Niels28<Fq377P> te_niels_finish(const TeMapParts &m, const Fp<Fq377P> &den_inv) {
    using F = Fp<Fq377P>;
    using G = FpMsm<Fq377P>;
    F xe = Te377::f_std() * m.u * (m.u + F::one()) * den_inv;
    F ye = (m.u - F::one()) * m.v * den_inv;
    Niels28<Fq377P> n;
    n.ymx = G::from_std(ye - xe);
    n.ypx = G::from_std(ye + xe);
    n.td = G::from_std(Te377::k2d_std() * xe * ye);
    return n;
}
// one point, own inversion (host side and tests; the device converts in batches: kernels_msm.hip k_convert_bases_te)
ZK_HD Niels28<Fq377P> niels_from_weierstrass(const Affine<Fp<Fq377P>> &p, bool *bad) {
    if (p.is_inf()) return niels_identity<Fq377P>();
    TeMapParts m = te_map_parts(p);
    if (m.den.is_zero()) { if (bad) *bad = true; return niels_identity<Fq377P>(); }
    return te_niels_finish(m, m.den.inverse());
}
#endif  // ZK_MSM_EDWARDS

// ---- one vocabulary for the bucket-reduction kernels over either accumulator type
template <class A> struct PtOps;
template <class P> struct PtOps<Acc28<P>> {
    using Params = P;
    ZK_HD static Acc28<P> identity() { return inf28<P>(); }
    ZK_HD static void add(Acc28<P> &a, const Acc28<P> &b) { add28<P>(a, b); }
    ZK_HD static void dbl(Acc28<P> &a) { dbl28<P>(a); }
    ZK_HD static Acc28<P> neg(const Acc28<P> &a) { return neg28<P>(a); }
    ZK_HD static XYZZ<Fp<P>> to_std(const Acc28<P> &a) { return to_std_point<P>(a); }
};
#if ZK_MSM_EDWARDS
template <class P> struct PtOps<AccTE<P>> {
    using Params = P;
    ZK_HD static AccTE<P> identity() { return te_identity<P>(); }
    ZK_HD static void add(AccTE<P> &a, const AccTE<P> &b) { te_add<P>(a, b); }
    ZK_HD static void dbl(AccTE<P> &a) { te_dbl<P>(a); }
    ZK_HD static AccTE<P> neg(const AccTE<P> &a) { return te_neg<P>(a); }
    ZK_HD static XYZZ<Fp<P>> to_std(const AccTE<P> &a) { return te_to_std_point<P>(a); }
};
#endif

}  // namespace zk
