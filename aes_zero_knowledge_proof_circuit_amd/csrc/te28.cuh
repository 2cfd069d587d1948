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

namespace zk {

// precomputed affine point (y - x, y + x, 2 d x y) in the reduced-radix Montgomery form: 3 x 56 B (+ 24 B of padding to a 64-byte aligned 192-byte record).  The identity is (1, 1, 0): the SRS's "infinity"
// entries (none in practice) need no test in the hot loop.
// (padded to 192 B and 64-byte aligned: exactly three 64-byte sectors per gather instead of 3.5 on average for packed 168-byte records -- k_accumulate 7.31 -> 7.13 ms
// at 2^22 points, bench +1.1 %, profiles/r03_niels_padding.txt -- for 14 % more table memory)
template <class P>
struct alignas(64) Niels28 { FpMsm<P> ymx, ypx, td; uint32_t pad[6]; };
// extended projective point: same 224 B as the XYZZ accumulator, so the MSM scratch buffers serve both
template <class P>
struct AccTE { FpMsm<P> x, y, z, t; };

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

// acc += n, seven products, everything inlined.  acc coordinates are products (normalized limbs, < 1.2 p) in and out; n's coordinates normalized.
// Carries are propagated only where a product needs them (profiles/r03_te_lazy.txt, r03_accumulate_instruction_diet.txt).  The cold callers (overflow segments,
// class sums) use this one; the hot loop of k_accumulate uses te_madd_hot below.
template <class P>
ZK_HD void te_madd(AccTE<P> &a, const Niels28<P> &n) {
    using G = FpMsm<P>;
    G A = a.y.template sub_lazy<3>(a.x) * n.ymx;     // (Y1 - X1)(y2 - x2)
    G B = a.y.add_lazy(a.x) * n.ypx;                 // (Y1 + X1)(y2 + x2)
    G C = a.t * n.td;                                // T1 2 d x2 y2          (T1 normalized)
    G D = a.z.dbl_lazy();                            // 2 Z1                  (Z2 = 1)
    G E = B.template sub<2>(A), H = B.add_lazy(A);   // E normalized, H lazy
    G F = D.template sub_lazy<2>(C), Gg = D + C;     // F lazy (D lazy + 2p - C), G normalized
    a.x = E * F; a.y = Gg * H; a.t = E * H; a.z = F * Gg;
}
// a table record with its first two coordinates swapped when the digit that uses it is negative: -(x, y) = (-x, y) swaps y - x and y + x (and negates 2dxy, which te_madd_hot
// folds into F <-> G).  The swap costs nothing at the gather (two field addresses).
template <class P>
ZK_HD Niels28<P> niels_load_signed(const Niels28<P> *rec, bool neg) {
    const FpMsm<P> *f = &rec->ymx;                   // ymx and ypx are adjacent
    Niels28<P> r;
    r.ymx = f[neg ? 1 : 0]; r.ypx = f[neg ? 0 : 1]; r.td = rec->td;
    return r;
}
// The hot loop's addition: acc += (neg ? -n : n), where n was loaded with niels_load_signed(.., neg).  A negative digit changes the sign of C, which swaps F = D - C and G = D + C
// (28 per-limb selects, no negated copy of the point).  `next` is the record of the lane's NEXT addition, loaded (with the sign of ITS digit, next_neg) into n as soon as the three
// products that read n are done: the gather flies under the remaining four products and lands in the registers the current point just vacated -- no second register set, no copies
// at the loop's back edge.  `bias` = FpMsm<P>::hot_loop_bias() taken at kernel entry (ff28.cuh mul_biased: 14 fewer 64-bit adds per product).
// All four factors of the second level are lazy.  Limb bounds in units of 2^28: E = B + 2p' - A < 3, H = B + A < 2, D = 2 Z1 < 2, U = D + 2p' - C < 4, V = D + C < 3
// (2p' = kp_spread<2> < 2 per limb).  The widest products, E U and U V, put 14 x 12 x 2^56 into a column, the reduction 13 x 2^56 more: 181 x 2^56 < 2^64
// (tests/test_ff28_host.py multiplies at exactly those bounds).
template <class P>
ZK_HD void te_madd_hot(AccTE<P> &a, Niels28<P> &n, bool neg, const Niels28<P> *next, bool next_neg, uint64_t bias) {
    using G = FpMsm<P>;
    G A = G::mul_biased(a.y.template sub_lazy<3>(a.x), n.ymx, bias);
    G B = G::mul_biased(a.y.add_lazy(a.x), n.ypx, bias);
    G C = G::mul_biased(a.t, n.td, bias);
#if defined(__HIP_DEVICE_COMPILE__)
    // pin the gather between the three products and the other four: without the first fence the compiler hoists the loads to the top of the iteration (a second register
    // set), without the second one the scheduler sinks them to its end (the gather's latency is then exposed at the top of the next addition)
    asm volatile("" : "+v"(A.l[G::N - 1]), "+v"(B.l[G::N - 1]), "+v"(C.l[G::N - 1]) : : "memory");
#endif
    n = niels_load_signed<P>(next, next_neg);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(A.l[0]), "+v"(C.l[0]) : : "memory");
#endif
    G E = B.template sub_lazy<2>(A), H = B.add_lazy(A);
    G D = a.z.dbl_lazy();
    G U = D.template sub_lazy<2>(C), V = D.add_lazy(C), F, Gg;
#pragma unroll
    for (int i = 0; i < G::N; i++) { F.l[i] = neg ? V.l[i] : U.l[i]; Gg.l[i] = neg ? U.l[i] : V.l[i]; }
    a.x = G::mul_biased(E, F, bias); a.y = G::mul_biased(Gg, H, bias); a.t = G::mul_biased(E, H, bias); a.z = G::mul_biased(F, Gg, bias);
}
// a += b, unified (add-2008-hwcd-3): nine products.  te_add_inline is the body for the one kernel that wants two additions of a step in ONE instruction stream
// (k_reduce_l1: a lone wave per SIMD issues dependent multiply-adds at half rate; the two independent additions of its running sums interleave); te_add is the call
// everything else makes (code size: ~4,000 instructions per copy).
template <class P>
ZK_HD void te_add_inline(AccTE<P> &a, const AccTE<P> &b) {
    using G = FpMsm<P>;
    G A = a.y.template sub<3>(a.x) * b.y.template sub<3>(b.x);
    G B = (a.y + a.x) * (b.y + b.x);
    G C = (a.t * b.t) * Te377::k2d();
    G D = (a.z * b.z).dbl();
    G E = B.template sub<2>(A), H = B + A;
    G F = D.template sub<2>(C), Gg = D + C;
    a.x = E * F; a.y = Gg * H; a.t = E * H; a.z = F * Gg;
}
template <class P>
ZK_EC_FN void te_add(AccTE<P> &a, const AccTE<P> &b) { te_add_inline<P>(a, b); }
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

#if defined(__HIPCC__)
// ---- FOUR LANES PER POINT OPERATION (the narrow stages of the bucket reduction).  A lane running a whole te_add alone issues its ~4,700 instructions one after the other
// (~12 us); where few points are left and most lanes idle anyway, a quad of lanes shares one operation instead: lane q of the quad holds coordinate q (x, y, z, t: the
// AccTE order) and multiplies one of the operation's independent products, so an addition is THREE product-times (9 products over 4 lanes; the t lane also multiplies by 2d)
// and a doubling TWO, with the operands exchanged inside the quad by DPP quad_perm moves (no LDS, no barrier).  Same formulas and value bounds as te_add / te_dbl, same results.
template <int CTRL> __device__ __forceinline__ uint32_t quad_dpp(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
#else
    return v;            // (host pass of hipcc: never executed)
#endif
}
template <int CTRL, class G> __device__ __forceinline__ G quad_move(const G &v) {
    G r;
#pragma unroll
    for (int i = 0; i < G::N; i++) r.l[i] = quad_dpp<CTRL>(v.l[i]);
    return r;
}
constexpr int QP_LANE0 = 0x00, QP_LANE1 = 0x55, QP_LANE2 = 0xAA, QP_LANE3 = 0xFF, QP_SWAP_PAIRS = 0xB1;      // quad_perm: broadcast lane k; [1, 0, 3, 2]
template <class G> __device__ __forceinline__ G quad_pick(int q, const G &v0, const G &v1, const G &v2, const G &v3) {
    G r;
#pragma unroll
    for (int i = 0; i < G::N; i++) r.l[i] = q == 0 ? v0.l[i] : (q == 1 ? v1.l[i] : (q == 2 ? v2.l[i] : v3.l[i]));
    return r;
}
// second level shared by addition and doubling: from A, B, C, D (one per lane) to X3 = E F, Y3 = G H, Z3 = F G, T3 = E H
template <class P>
__device__ __forceinline__ FpMsm<P> te_quad_finish(int q, const FpMsm<P> &E, const FpMsm<P> &F, const FpMsm<P> &Gg, const FpMsm<P> &H) {
    return quad_pick(q, E, Gg, F, E) * quad_pick(q, F, H, Gg, H);
}
// my coordinate of a + b, given my coordinate of a and of b
template <class P>
__device__ __forceinline__ FpMsm<P> te_add_quad(const FpMsm<P> &a, const FpMsm<P> &b, int q) {
    using G = FpMsm<P>;
    const G oa = quad_move<QP_SWAP_PAIRS>(a), ob = quad_move<QP_SWAP_PAIRS>(b);             // the x and y lanes see each other's coordinate
    // x lane: (Y1 - X1)(Y2 - X2); y lane: (Y1 + X1)(Y2 + X2); z lane: Z1 Z2; t lane: T1 T2 (then times 2d)
    const G u = quad_pick(q, oa.template sub<3>(a), a + oa, a, a), v = quad_pick(q, ob.template sub<3>(b), b + ob, b, b);
    G m = u * v;
    const G mk = m * Te377::k2d();
    m = quad_pick(q, m, m, m, mk);
    const G A = quad_move<QP_LANE0>(m), B = quad_move<QP_LANE1>(m), D = quad_move<QP_LANE2>(m).dbl(), C = quad_move<QP_LANE3>(m);
    return te_quad_finish<P>(q, B.template sub<2>(A), D.template sub<2>(C), D + C, B + A);
}
template <class P>
__device__ __forceinline__ FpMsm<P> te_dbl_quad(const FpMsm<P> &a, int q) {
    using G = FpMsm<P>;
    const G X = quad_move<QP_LANE0>(a), Y = quad_move<QP_LANE1>(a);
    const G m = quad_pick(q, a, a, a, X + Y).sqr();                                          // X^2, Y^2, Z^2, (X + Y)^2
    const G A = quad_move<QP_LANE0>(m), B = quad_move<QP_LANE1>(m), C = quad_move<QP_LANE2>(m).dbl(), S = quad_move<QP_LANE3>(m);
    const G AB = A + B, Gg = B.template sub<2>(A);
    return te_quad_finish<P>(q, S.template sub<3>(AB), Gg.template sub<3>(C), Gg, G::zero().template sub<3>(AB));
}
// coordinate q of the identity (0 : 1 : 1 : 0)
template <class P>
__device__ __forceinline__ FpMsm<P> te_identity_quad(int q) { return (q == 1 || q == 2) ? FpMsm<P>::k_one() : FpMsm<P>::zero(); }
#endif

// ---- one vocabulary for the bucket-reduction kernels over either accumulator type
template <class A> struct PtOps;
template <class P> struct PtOps<Acc28<P>> {
    using Params = P;
    ZK_HD static Acc28<P> identity() { return inf28<P>(); }
    ZK_HD static void add(Acc28<P> &a, const Acc28<P> &b) { add28<P>(a, b); }
    ZK_HD static void add_inline(Acc28<P> &a, const Acc28<P> &b) { add28<P>(a, b); }
    ZK_HD static void dbl(Acc28<P> &a) { dbl28<P>(a); }
    ZK_HD static Acc28<P> neg(const Acc28<P> &a) { return neg28<P>(a); }
    ZK_HD static XYZZ<Fp<P>> to_std(const Acc28<P> &a) { return to_std_point<P>(a); }
};
template <class P> struct PtOps<AccTE<P>> {
    using Params = P;
    ZK_HD static AccTE<P> identity() { return te_identity<P>(); }
    ZK_HD static void add(AccTE<P> &a, const AccTE<P> &b) { te_add<P>(a, b); }
    ZK_HD static void add_inline(AccTE<P> &a, const AccTE<P> &b) { te_add_inline<P>(a, b); }
    ZK_HD static void dbl(AccTE<P> &a) { te_dbl<P>(a); }
    ZK_HD static AccTE<P> neg(const AccTE<P> &a) { return te_neg<P>(a); }
    ZK_HD static XYZZ<Fp<P>> to_std(const AccTE<P> &a) { return te_to_std_point<P>(a); }
};

}  // namespace zk
