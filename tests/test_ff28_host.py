"""CPU test of the reduced-radix field arithmetic (csrc/ff28.cuh) used by the MSM kernels: compiled for the host with g++ and checked
against the 12x32-bit Montgomery implementation over random inputs, including value growth through lazy additions / subtractions."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aes_zero_knowledge_proof_circuit_amd", "csrc")

SRC = r'''
#include "ff28.cuh"
#include <cstdio>
#include <cstdlib>
using namespace zk;
template <class P, class G> int run(const char *name) {
    using F = Fp<P>;
    srand(7);
    int bad = 0;
    for (int it = 0; it < 3000; it++) {
        F a, b;
        for (int i = 0; i < 12; i++) { a.l[i] = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand(); b.l[i] = (uint32_t)rand() * 40503u ^ ((uint32_t)rand() << 3); }
        a.l[11] &= 0x00ffffff; b.l[11] &= 0x00ffffff;
        if (it == 0) a = F::zero();
        if (it == 1) { a = F::one(); b = F::one().neg(); }
        G a8 = G::from_std(a), b8 = G::from_std(b);
        bad += !(a8.to_std() == a);
        bad += !((a8 * b8).to_std() == a * b);
        bad += !((a8 + b8).to_std() == a + b);
        bad += !((a8.template sub<2>(b8)).to_std() == a - b);
        G u = (a8 + b8).dbl(), v = a8.template sub<4>(b8);
        bad += !((u * v).to_std() == ((a + b).dbl()) * (a - b));
        G w = ((u * v).template sub<16>(u.dbl().dbl())).sqr();           // inputs up to ~17p
        F ws = (((a + b).dbl()) * (a - b) - (a + b).dbl().dbl().dbl()).sqr();
        bad += !(w.to_std() == ws);
        bad += (a8.template sub<2>(a8)).is_zero_mod_p() != true;
        bad += a8.is_zero_mod_p() != a.is_zero();
        // a two-product sum with one reduction, on grown operands; a product that is 0 mod p must test as zero
        bad += !(G::fma2(u, v, a8.template sub<2>(b8), b8.dbl()).to_std() == ((a + b).dbl()) * (a - b) + (a - b) * b.dbl());
        bad += G::product_is_zero(a8 * b8) != (a * b).is_zero();
        bad += G::product_is_zero((a8.template sub<2>(a8)) * b8) != true;
        bad += G::product_is_zero((u.template sub<8>(u)).sqr()) != true;
        // long lazy chain: values grow to many multiples of p before the next product
        G acc = a8; F racc = a;
        for (int k = 0; k < 6; k++) { acc = (acc * b8).template sub<7>(a8.dbl()) + v; racc = racc * b - a.dbl() + (a - b); }
        bad += !(acc.to_std() == racc);
    }
    printf("%s %d\n", name, bad);
    return bad;
}
// products of two LAZY operands at the limb bounds the hot loop allows (te28.cuh te_madd_hot: 4 x 2^28 and 3 x 2^28 per limb), plain and with the
// reduction rows' bias as the column start: the 64-bit columns must not wrap
int lazy_bounds() {
    using G = Fp28<Fq377P>;
    int bad = 0;
    srand(11);
    for (int it = 0; it < 2000; it++) {
        G x, y;
        int64_t tx[14], ty[14];
        for (int i = 0; i < 14; i++) {
            const bool top = i == 13;
            uint32_t bx = top ? (1u << 18) : (4u << 28) - 1, by = top ? (1u << 17) : (3u << 28) - 1;
            x.l[i] = it == 0 ? bx : bx - ((uint32_t)rand() & (it & 1 ? 0xffu : 0x7ffffffu) % bx);
            y.l[i] = it == 0 ? by : by - ((uint32_t)rand() & (it & 2 ? 0xfu : 0x3ffffffu) % by);
            tx[i] = x.l[i]; ty[i] = y.l[i];
        }
        Fp<Fq377P> want = G::normalize(tx).to_std() * G::normalize(ty).to_std();
        bad += !((x * y).to_std() == want);
        bad += !(G::mul_biased(x, y, G::hot_loop_bias()).to_std() == want);
        G pl = x * y, pb = G::mul_biased(x, y, G::hot_loop_bias());
        for (int i = 0; i < 14; i++) bad += pl.l[i] != pb.l[i];
    }
    printf("lazy_bounds %d\n", bad);
    return bad;
}
int main() {
    return lazy_bounds() + run<Fq377P, Fp28<Fq377P>>("fq377x28") + run<Fq381P, Fp28<Fq381P>>("fq381x28");
}
'''


def test_reduced_radix_field_matches_montgomery_reference():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(SRC)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", CSRC, src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.split() == ["lazy_bounds", "0", "fq377x28", "0", "fq381x28", "0"]



def test_reduced_radix_group_law_matches_xyzz_reference():
    """madd28 / add28 / dbl28 / neg28 (csrc/ec28.cuh) against XYZZ<Fq> on random points of both curves: accumulation chains with negated and
    non-canonical (value >= p) coordinates, the running-sum pattern of the bucket reduction (P + P through the complete law), P - P = infinity."""
    src_path = os.path.join(ROOT, "tests", "ec28_host_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", CSRC, src_path, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.split() == ["bls377", "0", "bls381", "0"], out.stdout


def test_twisted_edwards_group_law_matches_xyzz_reference():
    """csrc/te28.cuh (BLS12-377 G1 on its twisted Edwards model: the 7-product precomputed-point addition of k_accumulate, the unified full addition and
    doubling of the bucket reduction, negation, both maps) against XYZZ<Fq> on random points of the prime-order subgroup: accumulation chains from the
    identity incl. P + P and P - P through the unified law, running sums, double-and-add, infinity <-> identity, a 2-torsion point is refused."""
    src_path = os.path.join(ROOT, "tests", "te28_host_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", CSRC, src_path, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.split() == ["te377", "0"], out.stdout


SRC29 = r'''
#include "ff29.cuh"
#include <cstdio>
#include <cstdlib>
using namespace zk;
template <class P> int run(const char *name) {
    using F = Fp<P>; using G = Fp29<P>;
    srand(11);
    int bad = 0;
    for (int it = 0; it < 4000; it++) {
        uint32_t ra[8], rb[8];
        for (int i = 0; i < 8; i++) { ra[i] = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand(); rb[i] = (uint32_t)rand() * 40503u ^ ((uint32_t)rand() << 3); }
        ra[7] &= 0x0fffffff; rb[7] &= 0x0fffffff;            // < 2^252 < p for both fields
        F a = F::from_raw(ra), b = F::from_raw(rb);
        if (it == 0) a = F::zero();
        if (it == 1) { a = F::one(); b = F::one().neg(); }
        if (it == 2) { a = F::one().neg(); b = F::one().neg(); }
        G A = G::from_std_relimb(a), B = G::from_std_relimb(b), W = G::twiddle_from_std(b);
        bad += !(A.template to_std_relimb<0>() == a);
        bad += !((A * W).template to_std_relimb<1>() == a * b);
        bad += !((A + B).template to_std_relimb<1>() == a + b);
        bad += !((A.template sub<1>(B)).template to_std_relimb<1>() == a - b);
        // the NTT's growth pattern: three product-free stages (K = 1, 4, 8), then seven multiplied stages, values up to ~29 p
        G x = A, y = B; F xs = a, ys = b;
        { G p = x + y, q = x.template sub<1>(y); x = p; y = q; F ps = xs + ys, qs = xs - ys; xs = ps; ys = qs; }
        { G p = x + y, q = x.template sub<4>(y); x = p + p; y = q; F ps = xs + ys, qs = xs - ys; xs = ps + ps; ys = qs; }     // x < 8 p
        { G p = x + y, q = x.template sub<8>(y); x = p; y = q; F ps = xs + ys, qs = xs - ys; xs = ps; ys = qs; }             // x < 14 p, y < 16 p
        for (int st = 0; st < 7; st++) {
            G t = y * W; F ts = ys * b;
            G p = x + t, q = x.template sub<2>(t); F ps = xs + ts, qs = xs - ts;
            x = p; y = q; xs = ps; ys = qs;
            if (st & 1) { G tmp = x; x = y; y = tmp; F tt = xs; xs = ys; ys = tt; }
        }
        bad += !(x.template to_std_relimb<4>() == xs);
        bad += !(y.template to_std_relimb<4>() == ys);
        bad += !((x * W).template to_std_relimb<1>() == xs * b);
        // the NTT's stores (round 5): quotient estimate from the top limb -> [0, 2 p), fits the 8 x 32-bit words between passes; one more round gives the canonical value
        {
            G rx = x.reduce_by_top_limb(), ry = y.reduce_by_top_limb();
            bad += (rx.l[8] >> 24) != 0 || (ry.l[8] >> 24) != 0;                       // < 2^256
            for (int i = 0; i < 8; i++) bad += rx.l[i] > G::MASK || ry.l[i] > G::MASK;
            bad += !(rx.template to_std_relimb<0>() == xs);
            bad += !(ry.template to_std_relimb<0>() == ys);
            uint32_t w8[8]; rx.pack(w8);
            bad += !(G::split(w8).template to_std_relimb<0>() == xs);                  // survives the trip through memory
        }
        // dot products with one reduction (the polynomial kernels' linear combinations): data x scalar-in-twiddle-form terms, lazy operands (< 2 p, < 4 p), and
        // data x data through the five-bit shift: mul(shl5(a R), b R) = a b R
        {
            F c = a * a + b, d2 = b * b - a;
            G C = G::from_std_relimb(c), D2 = G::from_std_relimb(d2), WC = G::twiddle_from_std(c), WA = G::twiddle_from_std(a);
            G av[4] = {A, B, C, D2}, bv[4] = {W, WC, WA, W};
            bad += !(G::template dot<4>(av, bv).template to_std_relimb<1>() == a * b + b * c + c * a + d2 * b);
            bad += !(G::template dot<3>(av, bv).template to_std_relimb<1>() == a * b + b * c + c * a);
            bad += !(G::template dot<1>(av, bv).template to_std_relimb<1>() == a * b);
            G lazy[2] = {A + B, (C + D2).template sub<2>(B)}, wv[2] = {WC, WA};
            bad += !(G::template dot<2>(lazy, wv).template to_std_relimb<1>() == (a + b) * c + (c + d2 - b) * a);
            bad += !((A.shl5() * B).template to_std_relimb<1>() == a * b);
            G s5[2] = {A.shl5(), (C + C).shl5()}, dd[2] = {B, G::zero().template sub<2>(D2 + D2)};
            bad += !(G::template dot<2>(s5, dd).template to_std_relimb<1>() == a * b - (c + c) * (d2 + d2));
        }
    }
    // edges of the estimate: K p - 1, K p, K p + 1 for every K <= 32 (a ten-stage pass holds values below 32 p); the expected residues as raw integers
    {
        G pp, one = G::zero();
        for (int i = 0; i < 9; i++) pp.l[i] = G::mod29(i);
        one.l[0] = 1;
        F r0 = F::zero(), r1 = F::zero(), rm = F::zero();
        r1.l[0] = 1;
        for (int i = 0; i < 8; i++) rm.l[i] = P::mod(i);
        rm.l[0] -= 1;                                                  // p is odd
        G v = G::zero();
        for (int K = 0; K <= 32; K++) {
            for (int d = -1; d <= 1; d++) {
                if ((K == 0 && d < 0) || (K == 32 && d >= 0)) continue;
                G u = d < 0 ? v.template sub<0>(one) : (d > 0 ? v + one : v);
                G r = u.reduce_by_top_limb();
                bad += (r.l[8] >> 24) != 0;
                F got; r.template canonical<0>().pack(got.l);
                bad += !(got == (d < 0 ? rm : (d > 0 ? r1 : r0)));
            }
            v = v + pp;
        }
    }
    printf("%s %d\n", name, bad);
    return bad;
}
int main() { return run<Fr377P>("fr377") + run<Fr381P>("fr381"); }
'''


def test_reduced_radix_scalar_field_matches_montgomery_reference():
    """csrc/ff29.cuh (9 x 29-bit limbs, the NTT butterflies' arithmetic) against the 8 x 32-bit Montgomery field, including the lazy value growth
    of a ten-stage pass (K = 1, 4, 8 product-free stages, then multiplied stages) and the final canonicalisation; dot products of up to four terms with one reduction
    and the five-bit shift that multiplies two standard-form values (the polynomial kernels of rounds 2-3 and the openings)."""
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(SRC29)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", CSRC, src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.split() == ["fr377", "0", "fr381", "0"]


SRC_INV = r'''
#include "ff.cuh"
#include <cstdio>
#include <cstdlib>
using namespace zk;
template <class F> int run(const char *name) {
    srand(23);
    int bad = 0;
    for (int it = 0; it < 600; it++) {
        uint32_t raw[F::N];
        for (int i = 0; i < F::N; i++) raw[i] = (uint32_t)rand() * 2654435761u ^ (uint32_t)rand();
        raw[F::N - 1] &= (1u << ((F::BITS - 2) % 32)) - 1;        // < p
        F a = F::from_raw(raw);
        if (it == 0) a = F::zero();
        if (it == 1) a = F::one();
        if (it == 2) a = F::one().neg();
        if (it == 3) a = F::one() + F::one();
        if (it == 4) a = (F::one() + F::one()).inverse_fermat();                 // (p + 1) / 2
        if (it >= 5 && it < 40) a = F::from_u64(1ull << (it - 5)) ;              // powers of two: the longest runs of halvings
        if (it >= 40 && it < 80) a = F::from_u64((uint64_t)it * 0x9e3779b97f4a7c15ull).neg();
        F i1 = a.inverse(), i2 = a.inverse_fermat();
        bad += !(i1 == i2);
        if (!a.is_zero()) bad += !(i1 * a == F::one());
        for (int i = 0; i < F::N; i++) raw[i] = i1.l[i];
        bad += !a.is_zero() && F::geq_mod(raw);                                   // canonical
    }
    printf("%s %d\n", name, bad);
    return bad;
}
int main() { return run<Fr377>("fr377") + run<Fr381>("fr381") + run<Fq377>("fq377") + run<Fq381>("fq381"); }
'''


def test_euclid_inverse_matches_fermat():
    """Fp::inverse (csrc/ff.cuh: Kaliski's almost-Montgomery inverse + doublings) against the Fermat chain on all four fields: zero, +-1, 2, 1/2, powers of two,
    negated small values and random elements; the result is canonical and a * a^-1 = 1."""
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(SRC_INV)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", CSRC, src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.split() == ["fr377", "0", "fr381", "0", "fq377", "0", "fq381", "0"], out.stdout
