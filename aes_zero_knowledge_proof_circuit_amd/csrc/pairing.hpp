// csrc/pairing.hpp -- host-only BLS12-377 pairing for the verifier (KZG10::batch_check's product_of_pairings).
//
// The reference verifies on the CPU through ark-ec 0.3.0's Bls12 pairing (simpleworks::marlin::verify_proof, call site
// /root/reference/src/lib.rs:130-135); verification is milliseconds and stays on the host here too.
// Tower: Fq2 = Fq[u]/(u^2 + 5), Fq6 = Fq2[v]/(v^3 - u), Fq12 = Fq6[w]/(w^2 - v); D-type twist E': y^2 = x^3 + 1/u with
// untwist (x', y') -> (x' w^2, y' w^3).  Optimal-ate Miller loop over x = 0x8508c00000000001 with affine twist points and
// sparse lines  yP - (lambda xP) w + (lambda xT - yT) w^3,  then f^((q^12-1)/r) = (conj(f)/f)^((q^6+1)/r).
// Only bilinearity + non-degeneracy matter for the check e(-W, beta H) e(C, H) == 1; GT values never leave this file.
#pragma once
#include "ec.cuh"

namespace zk {
namespace pairing {

using Fq = Fq377;
inline Fq fq_from(const uint32_t *limbs) { Fq r; for (int i = 0; i < 12; i++) r.l[i] = limbs[i]; return r; }
inline Fq times5(const Fq &a) { Fq d = a.dbl().dbl(); return d + a; }

struct Fq2 {
    Fq c0, c1;
    static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2 &o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq2 operator+(const Fq2 &o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fq2 operator-(const Fq2 &o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fq2 neg() const { return {c0.neg(), c1.neg()}; }
    Fq2 operator*(const Fq2 &o) const { return {c0 * o.c0 - times5(c1 * o.c1), c0 * o.c1 + c1 * o.c0}; }   // u^2 = -5
    Fq2 sqr() const { return *this * *this; }
    Fq2 scale(const Fq &s) const { return {c0 * s, c1 * s}; }
    Fq2 mul_xi() const { return {times5(c1).neg(), c0}; }                                                   // * u
    Fq2 inverse() const { Fq n = (c0.sqr() + times5(c1.sqr())).inverse(); return {c0 * n, (c1 * n).neg()}; }
    Fq2 dbl() const { return *this + *this; }
};
struct Fq6 {
    Fq2 c0, c1, c2;
    static Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    Fq6 operator+(const Fq6 &o) const { return {c0 + o.c0, c1 + o.c1, c2 + o.c2}; }
    Fq6 operator-(const Fq6 &o) const { return {c0 - o.c0, c1 - o.c1, c2 - o.c2}; }
    Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
    Fq6 operator*(const Fq6 &o) const {
        return {c0 * o.c0 + (c1 * o.c2 + c2 * o.c1).mul_xi(), c0 * o.c1 + c1 * o.c0 + (c2 * o.c2).mul_xi(), c0 * o.c2 + c1 * o.c1 + c2 * o.c0};
    }
    Fq6 mul_v() const { return {c2.mul_xi(), c0, c1}; }
    Fq6 inverse() const {
        Fq2 t0 = c0.sqr() - (c1 * c2).mul_xi(), t1 = c2.sqr().mul_xi() - c0 * c1, t2 = c1.sqr() - c0 * c2;
        Fq2 d = (c0 * t0 + (c2 * t1 + c1 * t2).mul_xi()).inverse();
        return {t0 * d, t1 * d, t2 * d};
    }
    bool operator==(const Fq6 &o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
};
struct Fq12 {
    Fq6 c0, c1;
    static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
    Fq12 operator*(const Fq12 &o) const { return {c0 * o.c0 + (c1 * o.c1).mul_v(), c0 * o.c1 + c1 * o.c0}; }
    Fq12 sqr() const { return *this * *this; }
    Fq12 conj() const { return {c0, c1.neg()}; }
    Fq12 inverse() const { Fq6 d = (c0 * c0 - (c1 * c1).mul_v()).inverse(); return {c0 * d, (c1 * d).neg()}; }
    bool is_one() const { return c0 == Fq6::one() && c1 == Fq6::zero(); }
};

struct G2Affine {
    Fq2 x, y;
    bool inf;
    static G2Affine infinity() { return {Fq2::zero(), Fq2::zero(), true}; }
};
inline Fq2 g2_b() { return {fq_from(G2_377_B_C0_MONT), fq_from(G2_377_B_C1_MONT)}; }
inline G2Affine g2_generator() { return {{fq_from(G2_377_X_C0_MONT), fq_from(G2_377_X_C1_MONT)}, {fq_from(G2_377_Y_C0_MONT), fq_from(G2_377_Y_C1_MONT)}, false}; }
inline bool g2_on_curve(const G2Affine &p) { return p.inf || p.y.sqr() == p.x.sqr() * p.x + g2_b(); }
inline G2Affine g2_add(const G2Affine &a, const G2Affine &b) {
    if (a.inf) return b;
    if (b.inf) return a;
    Fq2 lam;
    if (a.x == b.x) {
        if ((a.y + b.y).is_zero()) return G2Affine::infinity();
        Fq2 xx = a.x.sqr();
        lam = (xx.dbl() + xx) * a.y.dbl().inverse();
    } else {
        lam = (b.y - a.y) * (b.x - a.x).inverse();
    }
    Fq2 x3 = lam.sqr() - a.x - b.x;
    return {x3, lam * (a.x - x3) - a.y, false};
}
inline G2Affine g2_mul_raw(const G2Affine &p, const uint32_t *k, int klimbs) {
    G2Affine acc = G2Affine::infinity();
    for (int i = klimbs * 32 - 1; i >= 0; i--) {
        acc = g2_add(acc, acc);
        if ((k[i / 32] >> (i % 32)) & 1) acc = g2_add(acc, p);
    }
    return acc;
}

// f_{x,Q}(P) without the final exponentiation; P affine on E(Fq), Q affine on the twist
inline Fq12 miller_loop(const Affine<Fq> &P, const G2Affine &Q) {
    if (P.is_inf() || Q.inf) return Fq12::one();
    Fq12 f = Fq12::one();
    G2Affine T = Q;
    auto line = [&](const Fq2 &lam, const G2Affine &at) {
        Fq12 l;
        l.c0 = {Fq2{P.y, Fq::zero()}, Fq2::zero(), Fq2::zero()};
        l.c1 = {lam.scale(P.x).neg(), lam * at.x - at.y, Fq2::zero()};
        return l;
    };
    const uint64_t x = BLS12_377_X;
    int top = 63;
    while (!((x >> top) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        Fq2 xx = T.x.sqr();
        Fq2 lam = (xx.dbl() + xx) * T.y.dbl().inverse();
        f = f.sqr() * line(lam, T);
        T = g2_add(T, T);
        if ((x >> i) & 1) {
            Fq2 lam2 = (Q.y - T.y) * (Q.x - T.x).inverse();
            f = f * line(lam2, T);
            T = g2_add(T, Q);
        }
    }
    return f;
}
inline Fq12 final_exponentiation(const Fq12 &f) {
    Fq12 g = f.conj() * f.inverse();                 // f^(q^6 - 1)
    Fq12 acc = Fq12::one();
    for (int i = BLS12_377_FINAL_EXP_LIMBS * 32 - 1; i >= 0; i--) {
        acc = acc.sqr();
        if ((BLS12_377_FINAL_EXP[i / 32] >> (i % 32)) & 1) acc = acc * g;
    }
    return acc;
}
// prod_i e(P_i, Q_i) == 1 ?
inline bool pairing_product_is_one(const Affine<Fq> *P, const G2Affine *Q, int n) {
    Fq12 f = Fq12::one();
    for (int i = 0; i < n; i++) f = f * miller_loop(P[i], Q[i]);
    return final_exponentiation(f).is_one();
}

}  // namespace pairing
}  // namespace zk
