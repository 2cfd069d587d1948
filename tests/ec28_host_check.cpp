// tests/ec28_host_check.cpp -- host-compiled check of csrc/ec28.cuh (driven by tests/test_ff28_host.py)
#include "ec28.cuh"
#include <cstdio>
#include <cstdlib>
using namespace zk;
// p itself as a field value (a non-canonical representative of 0): built from the parameter pack through the 12x32 split
template <class P> FpMsm<P> mod_as_field() { uint32_t w[12]; for (int i = 0; i < 12; i++) w[i] = P::mod(i); return FpMsm<P>::split(w); }
template <class Curve, class P> int run(const char *name, const uint32_t *gx, const uint32_t *gy) {
    using Fq = Fp<P>; using G = FpMsm<P>;
    Affine<Fq> g; for (int k = 0; k < 12; k++) { g.x.l[k] = gx[k]; g.y.l[k] = gy[k]; }
    XYZZ<Fq> Gs = XYZZ<Fq>::from_affine(g);
    srand(3);
    int bad = 0;
    auto same = [&](const Acc28<P> &a, const XYZZ<Fq> &ref) {
        Affine<Fq> x = to_std_point<P>(a).to_affine(), y = ref.to_affine();
        return x.x == y.x && x.y == y.y;
    };
    for (int it = 0; it < 300; it++) {
        uint32_t k1[2] = {(uint32_t)rand(), (uint32_t)rand() & 0xffff}, k2[2] = {(uint32_t)rand(), (uint32_t)rand() & 0xffff};
        Affine<Fq> p1 = Gs.mul_raw(k1, 2).to_affine(), p2 = Gs.mul_raw(k2, 2).to_affine();
        Affine28<P> q1 = Affine28<P>::from_std(p1), q2 = Affine28<P>::from_std(p2);
        for (int variant = 0; variant < 4; variant++) {
            Affine28<P> a1 = q1, a2 = q2; Affine<Fq> s1 = p1, s2 = p2;
            if (variant & 1) { a1.y = G::zero().template sub<2>(a1.y); s1 = s1.neg(); }
            if (variant & 2) { a2.y = G::zero().template sub<2>(a2.y); s2 = s2.neg(); }
            // non-canonical representatives: add p to the coordinates (still < 2.2 p)
            if (it % 3 == 0) a1.y = a1.y + mod_as_field<P>();
            // accumulate: first = a1, then madd a2, then madd a1 again, a2 again ...
            Acc28<P> acc; acc.x = a1.x; acc.y = a1.y; acc.zz = G::k_one(); acc.zzz = acc.zz;
            XYZZ<Fq> ref = XYZZ<Fq>::from_affine(s1);
            for (int r = 0; r < 5; r++) {
                const Affine28<P> &q = (r & 1) ? a1 : a2; const Affine<Fq> &sq = (r & 1) ? s1 : s2;
                if (!madd28<P>(acc, q)) { printf("%s unexpected degenerate\n", name); bad++; }
                ref.madd(sq);
                if (!same(acc, ref)) { bad++; if (bad < 5) printf("%s madd mismatch it=%d variant=%d r=%d\n", name, it, variant, r); }
            }
            // reduction pattern: run = B, tot += run repeatedly (doubling on the first repeat), plus neg28 and scalar double-and-add
            Acc28<P> B = acc, run = inf28<P>(), tot = inf28<P>();
            XYZZ<Fq> rB = ref, rrun = XYZZ<Fq>::inf(), rtot = XYZZ<Fq>::inf();
            for (int d = 0; d < 8; d++) {
                if (d == (it & 7)) { add28<P>(run, B); rrun.add(rB); }
                add28<P>(tot, run); rtot.add(rrun);
                if (!same(tot, rtot)) { bad++; if (bad < 5) printf("%s running-sum mismatch it=%d variant=%d d=%d\n", name, it, variant, d); }
            }
            Acc28<P> t2 = tot; add28<P>(t2, neg28<P>(run)); XYZZ<Fq> rt2 = rtot; rt2.add(rrun.neg());
            for (int i = 0; i < 3; i++) { dbl28<P>(t2); rt2 = rt2.dbl(); }
            if (!same(t2, rt2)) { bad++; if (bad < 5) printf("%s neg/dbl mismatch it=%d variant=%d\n", name, it, variant); }
            // P + (-P) = infinity, P + P = 2P through the complete law
            Acc28<P> z = B; add28<P>(z, neg28<P>(B)); if (!z.zz.limbs_zero()) { bad++; if (bad < 5) printf("%s P-P not infinity\n", name); }
            Acc28<P> w = B; add28<P>(w, B); if (!same(w, rB.dbl())) { bad++; if (bad < 5) printf("%s P+P mismatch\n", name); }
        }
    }
    printf("%s %d\n", name, bad);
    return bad;
}
int main() { return run<Bls377, Fq377P>("bls377", G1_377_X_MONT, G1_377_Y_MONT) + run<Bls381, Fq381P>("bls381", G1_381_X_MONT, G1_381_Y_MONT); }
