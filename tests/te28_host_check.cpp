// tests/te28_host_check.cpp -- host-compiled check of csrc/te28.cuh (driven by tests/test_ff28_host.py): the twisted Edwards path of the MSM kernels
// against the Weierstrass XYZZ reference on random points of BLS12-377's prime-order subgroup.
#include "te28.cuh"
#include <cstdio>
#include <cstdlib>
using namespace zk;
int main() {
    using P = Fq377P; using Fq = Fp<P>; using G = FpMsm<P>;
    Affine<Fq> g; for (int k = 0; k < 12; k++) { g.x.l[k] = G1_377_X_MONT[k]; g.y.l[k] = G1_377_Y_MONT[k]; }
    XYZZ<Fq> Gs = XYZZ<Fq>::from_affine(g);
    srand(5);
    int bad = 0;
    auto same = [&](const AccTE<P> &a, const XYZZ<Fq> &ref) {
        XYZZ<Fq> s = te_to_std_point<P>(a);
        if (s.is_inf() || ref.is_inf()) return s.is_inf() == ref.is_inf();
        Affine<Fq> x = s.to_affine(), y = ref.to_affine();
        return x.x == y.x && x.y == y.y;
    };
    // the map itself: identity <-> infinity, round trip through one madd onto the identity
    {
        bool flag = false;
        Niels28<P> n = niels_from_weierstrass(Affine<Fq>::inf(), &flag);
        AccTE<P> a = te_identity<P>(); te_madd<P>(a, n);
        if (flag || !te_to_std_point<P>(a).is_inf()) { bad++; printf("identity mismatch\n"); }
        Affine<Fq> two_torsion; two_torsion.x = Fq::one().neg(); two_torsion.y = Fq::zero();      // (-1, 0): outside the subgroup, must be flagged
        niels_from_weierstrass(two_torsion, &flag);
        if (!flag) { bad++; printf("2-torsion point not flagged\n"); }
    }
    for (int it = 0; it < 200; it++) {
        uint32_t k1[2] = {(uint32_t)rand(), (uint32_t)rand() & 0xffff}, k2[2] = {(uint32_t)rand(), (uint32_t)rand() & 0xffff};
        Affine<Fq> p1 = Gs.mul_raw(k1, 2).to_affine(), p2 = Gs.mul_raw(k2, 2).to_affine();
        bool flag = false;
        Niels28<P> q1 = niels_from_weierstrass(p1, &flag), q2 = niels_from_weierstrass(p2, &flag);
        if (flag) { bad++; printf("subgroup point flagged\n"); }
        for (int variant = 0; variant < 4; variant++) {
            Niels28<P> a1 = q1, a2 = q2; Affine<Fq> s1 = p1, s2 = p2;
            if (variant & 1) { a1 = niels_neg<P>(a1); s1 = s1.neg(); }
            if (variant & 2) { a2 = niels_neg<P>(a2); s2 = s2.neg(); }
            // accumulate from the identity: a1, a2, a1, a2, ... and a1 twice in a row (P + P through the unified law)
            AccTE<P> acc = te_identity<P>();
            XYZZ<Fq> ref = XYZZ<Fq>::inf();
            for (int r = 0; r < 7; r++) {
                const bool first = (r & 1) == 0 || r == 5;
                te_madd<P>(acc, first ? a1 : a2); ref.madd(first ? s1 : s2);
                if (!same(acc, ref)) { bad++; if (bad < 5) printf("madd mismatch it=%d variant=%d r=%d\n", it, variant, r); }
            }
            // the hot loop's form (te_madd_hot): the record arrives with its first two coordinates swapped for a negative digit, the sign swaps F and G, the products start their
            // columns at the rows' bias, and the next record is loaded into the current one's storage mid-addition
            {
                const uint64_t bias = FpMsm<P>::hot_loop_bias();
                AccTE<P> sg = te_identity<P>(); XYZZ<Fq> rs = XYZZ<Fq>::inf();
                bool negs[7];
                for (int r = 0; r < 7; r++) negs[r] = (r * 5 + it + variant) % 3 == 0;
                Niels28<P> cur = niels_load_signed<P>(&q1, negs[0]);
                for (int r = 0; r < 6; r++) {
                    const bool neg = negs[r], use1 = (r & 1) == 0, next1 = ((r + 1) & 1) == 0;
                    te_madd_hot<P>(sg, cur, neg, next1 ? &q1 : &q2, negs[r + 1], bias);
                    Affine<Fq> sp = use1 ? p1 : p2; if (neg) sp = sp.neg();
                    rs.madd(sp);
                    if (!same(sg, rs)) { bad++; if (bad < 5) printf("hot madd mismatch it=%d variant=%d r=%d\n", it, variant, r); }
                    const Niels28<P> want = niels_load_signed<P>(next1 ? &q1 : &q2, negs[r + 1]);
                    for (int i = 0; i < 14; i++) if (cur.ymx.l[i] != want.ymx.l[i] || cur.ypx.l[i] != want.ypx.l[i] || cur.td.l[i] != want.td.l[i]) { bad++; break; }
                }
                AccTE<P> z = te_identity<P>(); Niels28<P> c2 = niels_load_signed<P>(&q1, false);
                te_madd_hot<P>(z, c2, false, &q1, true, bias); te_madd_hot<P>(z, c2, true, &q1, false, bias);
                if (!te_to_std_point<P>(z).is_inf()) { bad++; if (bad < 5) printf("P-P (hot madd) not identity\n"); }
            }
            // P + (-P) = identity through madd
            { AccTE<P> z = te_identity<P>(); te_madd<P>(z, a1); te_madd<P>(z, niels_neg<P>(a1)); if (!te_to_std_point<P>(z).is_inf()) { bad++; if (bad < 5) printf("P-P (madd) not identity\n"); } }
            // reduction pattern: run = B, tot += run repeatedly (doubling on the first repeat), neg, dbl, scalar double-and-add
            AccTE<P> B = acc, run = te_identity<P>(), tot = te_identity<P>();
            XYZZ<Fq> rB = ref, rrun = XYZZ<Fq>::inf(), rtot = XYZZ<Fq>::inf();
            for (int d = 0; d < 8; d++) {
                if (d == (it & 7)) { te_add<P>(run, B); rrun.add(rB); }
                te_add<P>(tot, run); rtot.add(rrun);
                if (!same(tot, rtot)) { bad++; if (bad < 5) printf("running-sum mismatch it=%d variant=%d d=%d\n", it, variant, d); }
            }
            AccTE<P> t2 = tot; te_add<P>(t2, te_neg<P>(run)); XYZZ<Fq> rt2 = rtot; rt2.add(rrun.neg());
            for (int i = 0; i < 3; i++) { te_dbl<P>(t2); rt2 = rt2.dbl(); }
            if (!same(t2, rt2)) { bad++; if (bad < 5) printf("neg/dbl mismatch it=%d variant=%d\n", it, variant); }
            { AccTE<P> m = B; uint32_t k = 0x9d3 + it; AccTE<P> a = te_identity<P>(); XYZZ<Fq> ra = XYZZ<Fq>::inf();
              for (int bit = 11; bit >= 0; bit--) { te_dbl<P>(a); ra = ra.dbl(); if ((k >> bit) & 1) { te_add<P>(a, m); ra.add(rB); } }
              if (!same(a, ra)) { bad++; if (bad < 5) printf("double-and-add mismatch it=%d\n", it); } }
            AccTE<P> z = B; te_add<P>(z, te_neg<P>(B)); if (!te_to_std_point<P>(z).is_inf()) { bad++; if (bad < 5) printf("P-P not identity\n"); }
            AccTE<P> w = B; te_add<P>(w, B); if (!same(w, rB.dbl())) { bad++; if (bad < 5) printf("P+P mismatch\n"); }
            AccTE<P> w2 = B; te_dbl<P>(w2); if (!same(w2, rB.dbl())) { bad++; if (bad < 5) printf("dbl mismatch\n"); }
        }
    }
    printf("te377 %d\n", bad);
    return bad;
}
