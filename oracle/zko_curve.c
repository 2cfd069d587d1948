/* oracle/zko_curve.c -- TEST INFRASTRUCTURE (CPU oracle).
 * BLS12-377 / BLS12-381 G1 in Jacobian coordinates and Pippenger MSM, restating
 * ark-ec 0.3.0 short_weierstrass_jacobian + msm/variable_base.rs (Cargo.lock:118; not under
 * /root/reference) -- SURVEY.md §A.4 "VariableBaseMSM::multi_scalar_mul".
 */
#include "zko.h"
#include <omp.h>
#include "zko_consts.h"
#include <math.h>
#include <stdlib.h>

fr_params ZKO_FR377, ZKO_FR381;
fq_params ZKO_FQ377, ZKO_FQ381;
zko_curve ZKO_BLS377, ZKO_BLS381;

static void fr_fill(fr_params *P, const uint64_t *p, const uint64_t *one, const uint64_t *r2, uint64_t inv, int bits) {
    memcpy(P->p, p, 32); memcpy(P->one, one, 32); memcpy(P->r2, r2, 32); P->inv = inv; P->bits = bits;
}
static void fq_fill(fq_params *P, const uint64_t *p, const uint64_t *one, const uint64_t *r2, uint64_t inv, int bits) {
    memcpy(P->p, p, 48); memcpy(P->one, one, 48); memcpy(P->r2, r2, 48); P->inv = inv; P->bits = bits;
}
__attribute__((constructor)) static void zko_init_params(void) {
    fr_fill(&ZKO_FR377, FR377_P, FR377_ONE, FR377_R2, FR377_INV, FR377_BITS);
    fr_fill(&ZKO_FR381, FR381_P, FR381_ONE, FR381_R2, FR381_INV, FR381_BITS);
    fq_fill(&ZKO_FQ377, FQ377_P, FQ377_ONE, FQ377_R2, FQ377_INV, FQ377_BITS);
    fq_fill(&ZKO_FQ381, FQ381_P, FQ381_ONE, FQ381_R2, FQ381_INV, FQ381_BITS);
    ZKO_BLS377.fr = &ZKO_FR377; ZKO_BLS377.fq = &ZKO_FQ377; ZKO_BLS377.id = 377;
    memcpy(ZKO_BLS377.b.l, G1_377_B_MONT, 48);
    memcpy(ZKO_BLS377.gen.x.l, G1_377_X_MONT, 48); memcpy(ZKO_BLS377.gen.y.l, G1_377_Y_MONT, 48); ZKO_BLS377.gen.inf = 0;
    ZKO_BLS381.fr = &ZKO_FR381; ZKO_BLS381.fq = &ZKO_FQ381; ZKO_BLS381.id = 381;
    memcpy(ZKO_BLS381.b.l, G1_381_B_MONT, 48);
    memcpy(ZKO_BLS381.gen.x.l, G1_381_X_MONT, 48); memcpy(ZKO_BLS381.gen.y.l, G1_381_Y_MONT, 48); ZKO_BLS381.gen.inf = 0;
}
const zko_curve *zko_curve_by_id(int id) { return id == 381 ? &ZKO_BLS381 : &ZKO_BLS377; }
const fr_params *zko_fr_by_id(int id) { return id == 381 ? &ZKO_FR381 : &ZKO_FR377; }

void g1j_set_inf(g1j_t *r) { memset(r, 0, sizeof *r); }
int g1j_is_inf(const g1j_t *a) { return fq_is_zero(&a->z); }
void g1j_from_affine(g1j_t *r, const g1a_t *a, const zko_curve *C) {
    if (a->inf) { g1j_set_inf(r); return; }
    r->x = a->x; r->y = a->y; fq_set_one(&r->z, C->fq);
}
/* dbl-2009-l (a = 0) */
void g1j_dbl(g1j_t *r, const g1j_t *p, const zko_curve *C) {
    const fq_params *F = C->fq;
    if (g1j_is_inf(p)) { *r = *p; return; }
    fq_t a, b, c, d, e, f, t, z3;
    fq_sqr(&a, &p->x, F);
    fq_sqr(&b, &p->y, F);
    fq_sqr(&c, &b, F);
    fq_add(&t, &p->x, &b, F); fq_sqr(&t, &t, F); fq_sub(&t, &t, &a, F); fq_sub(&t, &t, &c, F); fq_dbl(&d, &t, F);
    fq_dbl(&e, &a, F); fq_add(&e, &e, &a, F);
    fq_sqr(&f, &e, F);
    fq_mul(&z3, &p->y, &p->z, F); fq_dbl(&z3, &z3, F);
    fq_sub(&t, &f, &d, F); fq_sub(&r->x, &t, &d, F);
    fq_dbl(&c, &c, F); fq_dbl(&c, &c, F); fq_dbl(&c, &c, F);
    fq_sub(&t, &d, &r->x, F); fq_mul(&t, &e, &t, F); fq_sub(&r->y, &t, &c, F);
    r->z = z3;
}
/* madd-2007-bl */
void g1j_madd(g1j_t *r, const g1j_t *p, const g1a_t *q, const zko_curve *C) {
    const fq_params *F = C->fq;
    if (q->inf) { *r = *p; return; }
    if (g1j_is_inf(p)) { g1j_from_affine(r, q, C); return; }
    fq_t z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
    fq_sqr(&z1z1, &p->z, F);
    fq_mul(&u2, &q->x, &z1z1, F);
    fq_mul(&s2, &q->y, &p->z, F); fq_mul(&s2, &s2, &z1z1, F);
    if (fq_eq(&u2, &p->x)) {
        if (fq_eq(&s2, &p->y)) { g1j_dbl(r, p, C); } else { g1j_set_inf(r); }
        return;
    }
    fq_sub(&h, &u2, &p->x, F);
    fq_sqr(&hh, &h, F);
    fq_dbl(&i, &hh, F); fq_dbl(&i, &i, F);
    fq_mul(&j, &h, &i, F);
    fq_sub(&rr, &s2, &p->y, F); fq_dbl(&rr, &rr, F);
    fq_mul(&v, &p->x, &i, F);
    fq_sqr(&x3, &rr, F); fq_sub(&x3, &x3, &j, F); fq_sub(&x3, &x3, &v, F); fq_sub(&x3, &x3, &v, F);
    fq_mul(&t, &p->y, &j, F); fq_dbl(&t, &t, F);
    fq_sub(&y3, &v, &x3, F); fq_mul(&y3, &rr, &y3, F); fq_sub(&y3, &y3, &t, F);
    fq_add(&z3, &p->z, &h, F); fq_sqr(&z3, &z3, F); fq_sub(&z3, &z3, &z1z1, F); fq_sub(&z3, &z3, &hh, F);
    r->x = x3; r->y = y3; r->z = z3;
}
/* add-2007-bl */
void g1j_add(g1j_t *r, const g1j_t *p, const g1j_t *q, const zko_curve *C) {
    const fq_params *F = C->fq;
    if (g1j_is_inf(p)) { *r = *q; return; }
    if (g1j_is_inf(q)) { *r = *p; return; }
    fq_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    fq_sqr(&z1z1, &p->z, F); fq_sqr(&z2z2, &q->z, F);
    fq_mul(&u1, &p->x, &z2z2, F); fq_mul(&u2, &q->x, &z1z1, F);
    fq_mul(&s1, &p->y, &q->z, F); fq_mul(&s1, &s1, &z2z2, F);
    fq_mul(&s2, &q->y, &p->z, F); fq_mul(&s2, &s2, &z1z1, F);
    if (fq_eq(&u1, &u2)) {
        if (fq_eq(&s1, &s2)) { g1j_dbl(r, p, C); } else { g1j_set_inf(r); }
        return;
    }
    fq_sub(&h, &u2, &u1, F);
    fq_dbl(&i, &h, F); fq_sqr(&i, &i, F);
    fq_mul(&j, &h, &i, F);
    fq_sub(&rr, &s2, &s1, F); fq_dbl(&rr, &rr, F);
    fq_mul(&v, &u1, &i, F);
    fq_sqr(&x3, &rr, F); fq_sub(&x3, &x3, &j, F); fq_sub(&x3, &x3, &v, F); fq_sub(&x3, &x3, &v, F);
    fq_mul(&t, &s1, &j, F); fq_dbl(&t, &t, F);
    fq_sub(&y3, &v, &x3, F); fq_mul(&y3, &rr, &y3, F); fq_sub(&y3, &y3, &t, F);
    fq_add(&z3, &p->z, &q->z, F); fq_sqr(&z3, &z3, F); fq_sub(&z3, &z3, &z1z1, F); fq_sub(&z3, &z3, &z2z2, F);
    fq_mul(&z3, &z3, &h, F);
    r->x = x3; r->y = y3; r->z = z3;
}
void g1j_neg(g1j_t *r, const g1j_t *a, const zko_curve *C) { *r = *a; fq_neg(&r->y, &a->y, C->fq); }
void g1j_to_affine(g1a_t *r, const g1j_t *a, const zko_curve *C) {
    const fq_params *F = C->fq;
    if (g1j_is_inf(a)) { memset(r, 0, sizeof *r); fq_set_one(&r->y, F); r->inf = 1; return; }
    fq_t zi, zi2, zi3;
    fq_inv(&zi, &a->z, F); fq_sqr(&zi2, &zi, F); fq_mul(&zi3, &zi2, &zi, F);
    fq_mul(&r->x, &a->x, &zi2, F); fq_mul(&r->y, &a->y, &zi3, F); r->inf = 0;
}
void g1j_batch_to_affine(g1a_t *r, const g1j_t *a, size_t n, const zko_curve *C) {
    const fq_params *F = C->fq;
    if (n == 0) return;
    fq_t *z = malloc(n * sizeof(fq_t)), *scr = malloc(n * sizeof(fq_t));
    for (size_t i = 0; i < n; i++) z[i] = a[i].z;
    fq_batch_inv(z, n, scr, F);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        if (fq_is_zero(&a[i].z)) { memset(&r[i], 0, sizeof r[i]); fq_set_one(&r[i].y, F); r[i].inf = 1; continue; }
        fq_t zi2, zi3;
        fq_sqr(&zi2, &z[i], F); fq_mul(&zi3, &zi2, &z[i], F);
        fq_mul(&r[i].x, &a[i].x, &zi2, F); fq_mul(&r[i].y, &a[i].y, &zi3, F); r[i].inf = 0;
    }
    free(z); free(scr);
}
void g1j_mul_raw(g1j_t *r, const g1j_t *a, const uint64_t *k, int klimbs, const zko_curve *C) {
    g1j_t acc; g1j_set_inf(&acc);
    int top = klimbs * 64 - 1;
    while (top >= 0 && !((k[top / 64] >> (top % 64)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        g1j_dbl(&acc, &acc, C);
        if ((k[i / 64] >> (i % 64)) & 1) g1j_add(&acc, &acc, a, C);
    }
    *r = acc;
}
void g1j_mul_fr(g1j_t *r, const g1j_t *a, const fr_t *k, const zko_curve *C) {
    uint64_t raw[4];
    fr_to_raw(raw, k, C->fr);
    g1j_mul_raw(r, a, raw, 4, C);
}
int g1a_on_curve(const g1a_t *a, const zko_curve *C) {
    if (a->inf) return 1;
    fq_t l, r;
    fq_sqr(&l, &a->y, C->fq);
    fq_sqr(&r, &a->x, C->fq); fq_mul(&r, &r, &a->x, C->fq); fq_add(&r, &r, &C->b, C->fq);
    return fq_eq(&l, &r);
}
int g1a_eq(const g1a_t *a, const g1a_t *b) {
    if (a->inf || b->inf) return a->inf == b->inf;
    return fq_eq(&a->x, &b->x) && fq_eq(&a->y, &b->y);
}

/* ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul: c = 3 if n < 32 else ln(n)+2; one task per window;
 * scalar == 1 handled in window 0; zero scalars skipped; running-sum bucket reduction; Horner over windows. */
/* ---- random group elements as ark-ec 0.3.0 draws them (short_weierstrass_jacobian.rs `impl Distribution<GroupProjective<P>> for Standard`,
 * not under /root/reference) [RECALL]: loop { x = BaseField::rand(rng); greatest = rng.gen::<bool>(); get_point_from_x(x, greatest) } then
 * scale_by_cofactor.  BaseField::rand = ark-ff UniformRand (limbs from next_u64, shave, reject, limbs ARE the Montgomery form);
 * gen::<bool>() = top bit of one next_u32 (rand 0.8); get_point_from_x picks y by (y < -y) ^ greatest on the canonical integers. */
void zko_fq_rand(fq_t *out, zko_chacha *rng, const fq_params *Q) {
    int shave = 384 - Q->bits;
    for (;;) {
        for (int i = 0; i < 6; i++) out->l[i] = zko_chacha_u64(rng);
        out->l[5] &= (~(uint64_t)0) >> shave;
        if (!fq_geq_raw(out->l, Q->p)) return;
    }
}
/* Tonelli-Shanks square root in Fq; returns 0 for a non-residue */
int zko_fq_sqrt(fq_t *out, const fq_t *a, const fq_params *Q) {
    if (fq_is_zero(a)) { *out = *a; return 1; }
    uint64_t t[6], half[6], tp1h[6];
    int S = 0;
    memcpy(t, Q->p, 48); t[0] -= 1;                                   /* p - 1 (p odd: no borrow) */
    for (int i = 0; i < 6; i++) half[i] = (t[i] >> 1) | (i < 5 ? t[i + 1] << 63 : 0);
    while (!(t[0] & 1)) { for (int i = 0; i < 6; i++) t[i] = (t[i] >> 1) | (i < 5 ? t[i + 1] << 63 : 0); S++; }
    memcpy(tp1h, t, 48); tp1h[0] += 1;                                /* t odd and != 2^64 - 1 in limb 0 for both curves */
    for (int i = 0; i < 6; i++) tp1h[i] = (tp1h[i] >> 1) | (i < 5 ? tp1h[i + 1] << 63 : 0);
    fq_t one, minus_one, chk;
    fq_set_one(&one, Q); fq_neg(&minus_one, &one, Q);
    fq_pow(&chk, a, half, 6, Q);
    if (!fq_eq(&chk, &one)) return 0;
    fq_t z, c, x, b;
    for (uint64_t cand = 2;; cand++) { fq_from_u64(&z, cand, Q); fq_pow(&chk, &z, half, 6, Q); if (fq_eq(&chk, &minus_one)) break; }
    fq_pow(&c, &z, t, 6, Q); fq_pow(&x, a, tp1h, 6, Q); fq_pow(&b, a, t, 6, Q);
    int m = S;
    while (!fq_eq(&b, &one)) {
        int i = 0;
        fq_t bb = b;
        while (!fq_eq(&bb, &one)) { fq_sqr(&bb, &bb, Q); i++; }
        fq_t g = c;
        for (int k = 0; k < m - i - 1; k++) fq_sqr(&g, &g, Q);
        fq_mul(&x, &x, &g, Q); fq_sqr(&c, &g, Q); fq_mul(&b, &b, &c, Q); m = i;
    }
    *out = x;
    return 1;
}
void zko_g1_rand(g1a_t *out, zko_chacha *rng, const zko_curve *C) {
    const fq_params *Q = C->fq;
    for (;;) {
        fq_t x, y, ny, rhs;
        zko_fq_rand(&x, rng, Q);
        int greatest = (int)(zko_chacha_u32(rng) >> 31);
        fq_sqr(&rhs, &x, Q); fq_mul(&rhs, &rhs, &x, Q); fq_add(&rhs, &rhs, &C->b, Q);
        if (!zko_fq_sqrt(&y, &rhs, Q)) continue;
        fq_neg(&ny, &y, Q);
        uint64_t ry[6], rny[6];
        fq_to_raw(ry, &y, Q); fq_to_raw(rny, &ny, Q);
        int y_lt = 0;
        for (int i = 5; i >= 0; i--) if (ry[i] != rny[i]) { y_lt = ry[i] < rny[i]; break; }
        g1a_t p; p.x = x; p.y = (y_lt ^ greatest) ? y : ny; p.inf = 0;
        g1j_t pj, r;
        g1j_from_affine(&pj, &p, C);
        if (C->id != 377) abort();                                    /* cofactor constant generated for BLS12-377 only */
        g1j_mul_raw(&r, &pj, G1_377_COFACTOR, G1_377_COFACTOR_LIMBS, C);
        g1j_to_affine(out, &r, C);
        return;
    }
}

/* ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul restated: window c = ln(n) + 2 (3 below 32 points), windows at bit offsets 0, c, 2c, ...,
 * 2^c - 1 buckets per window, scalar == 1 added directly in window 0, zero scalars skipped, running-sum bucket reduction, Horner over the
 * window sums.  Upstream runs ONE rayon task per window (<= 17 useful threads); so that the CPU baseline uses all the cores of the box, each
 * window's points are additionally cut into `slices` contiguous ranges with their own bucket arrays (merged bucket-wise afterwards) and the
 * running sum is done in 4096-bucket segments (W_seg + lo * S_seg).  Group addition is associative and commutative: the resulting point -- and
 * every serialized byte downstream -- is the same as the one-task-per-window order gives (tests/test_oracle_primitives.py compares against
 * plain double-and-add). */
void zko_msm(g1j_t *out, const g1a_t *bases, const fr_t *scalars, size_t n, const zko_curve *C) {
    const fr_params *F = C->fr;
    if (n == 0) { g1j_set_inf(out); return; }
    int c = 3;
    {   /* ln_without_floats(a) = (log2(a) * 69 / 100) with log2 = ceil-ish bit length; restate exactly */
        size_t a = n; int lg = 0; while ((1ull << lg) < a) lg++;   /* ark_std::log2(x) = ceil(log2 x) */
        if (n >= 32) c = (lg * 69 / 100) + 2;
    }
    int num_bits = F->bits;
    int nwin = (num_bits + c - 1) / c;
    uint64_t (*raw)[4] = malloc(n * sizeof *raw);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) fr_to_raw(raw[i], &scalars[i], F);
    g1j_t *wsum = malloc(nwin * sizeof(g1j_t));
    size_t nb = ((size_t)1 << c) - 1;
    int threads = omp_get_max_threads();
    int slices = (threads + nwin - 1) / nwin;
    while (slices > 1 && (n / slices < 8 * nb || (size_t)nwin * slices * nb * sizeof(g1j_t) > ((size_t)6 << 30))) slices--;   /* merging costs nb adds per extra slice */
    g1j_t *bk = malloc((size_t)nwin * slices * nb * sizeof(g1j_t));
    g1j_t *ones = malloc((size_t)slices * sizeof(g1j_t));           /* scalar == 1 terms (window 0 only) */
    if (!bk || !ones) abort();
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int w = 0; w < nwin; w++) {
        for (int sl = 0; sl < slices; sl++) {
            int w_start = w * c;
            g1j_t res; g1j_set_inf(&res);
            g1j_t *buckets = bk + ((size_t)w * slices + sl) * nb;
            for (size_t b = 0; b < nb; b++) g1j_set_inf(&buckets[b]);
            size_t i0 = n * (size_t)sl / slices, i1 = n * (size_t)(sl + 1) / slices;
            for (size_t i = i0; i < i1; i++) {
                const uint64_t *s = raw[i];
                if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
                if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {
                    if (w_start == 0) g1j_madd(&res, &res, &bases[i], C);
                    continue;
                }
                /* (scalar >> w_start) % 2^c */
                int limb = w_start / 64, sh = w_start % 64;
                uint64_t v = s[limb] >> sh;
                if (sh && limb + 1 < 4) v |= s[limb + 1] << (64 - sh);
                v &= ((uint64_t)1 << c) - 1;
                if (v != 0) g1j_madd(&buckets[v - 1], &buckets[v - 1], &bases[i], C);
            }
            if (w == 0) ones[sl] = res;
        }
    }
    /* running sum in segments: bucket b carries weight b + 1 = (b - lo + 1) + lo */
    const size_t SEG = 4096;
    size_t nseg = (nb + SEG - 1) / SEG;
    g1j_t *segres = malloc((size_t)nwin * nseg * sizeof(g1j_t));
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int w = 0; w < nwin; w++) {
        for (size_t sg = 0; sg < nseg; sg++) {
            size_t lo = sg * SEG, hi = lo + SEG < nb ? lo + SEG : nb;
            g1j_t running, acc; g1j_set_inf(&running); g1j_set_inf(&acc);
            for (size_t b = hi; b-- > lo;) {
                g1j_t tot = bk[((size_t)w * slices) * nb + b];
                for (int sl = 1; sl < slices; sl++) g1j_add(&tot, &tot, &bk[((size_t)w * slices + sl) * nb + b], C);
                g1j_add(&running, &running, &tot, C);
                g1j_add(&acc, &acc, &running, C);
            }
            if (lo) {   /* + lo * S_seg (running == S_seg now) */
                g1j_t m; g1j_set_inf(&m);
                int top = 63; while (top > 0 && !((lo >> top) & 1)) top--;
                for (int bit = top; bit >= 0; bit--) { g1j_dbl(&m, &m, C); if ((lo >> bit) & 1) g1j_add(&m, &m, &running, C); }
                g1j_add(&acc, &acc, &m, C);
            }
            segres[(size_t)w * nseg + sg] = acc;
        }
    }
    for (int w = 0; w < nwin; w++) {
        g1j_t res; g1j_set_inf(&res);
        if (w == 0) for (int sl = 0; sl < slices; sl++) g1j_add(&res, &res, &ones[sl], C);
        for (size_t sg = 0; sg < nseg; sg++) g1j_add(&res, &res, &segres[(size_t)w * nseg + sg], C);
        wsum[w] = res;
    }
    free(segres); free(ones); free(bk);
    /* lowest + sum_{w>=1 from high} (total + wsum[w]) doubled c times */
    g1j_t total; g1j_set_inf(&total);
    for (int w = nwin - 1; w >= 1; w--) {
        g1j_add(&total, &total, &wsum[w], C);
        for (int k = 0; k < c; k++) g1j_dbl(&total, &total, C);
    }
    g1j_add(out, &total, &wsum[0], C);
    free(wsum); free(raw);
}

/* out[i] = scalars[i] * base, via an 8-bit fixed-base window table (role of ark-ec FixedBaseMSM in
 * KZG10::setup; any table shape gives the same group elements). */
void zko_fixed_base_batch(g1a_t *out, const g1a_t *base, const fr_t *scalars, size_t n, const zko_curve *C) {
    const fr_params *F = C->fr;
    const int W = 8, NW = 32;
    g1a_t *table = malloc((size_t)NW * 255 * sizeof(g1a_t));
    g1j_t *tj = malloc((size_t)NW * 255 * sizeof(g1j_t));
    g1j_t wb; g1j_from_affine(&wb, base, C);
    for (int w = 0; w < NW; w++) {
        g1j_t acc = wb;
        for (int d = 1; d <= 255; d++) {
            tj[w * 255 + d - 1] = acc;
            g1j_add(&acc, &acc, &wb, C);
        }
        for (int k = 0; k < W; k++) g1j_dbl(&wb, &wb, C);
    }
    g1j_batch_to_affine(table, tj, (size_t)NW * 255, C);
    free(tj);
    g1j_t *res = malloc(n * sizeof(g1j_t));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint64_t raw[4];
        fr_to_raw(raw, &scalars[i], F);
        g1j_t acc; g1j_set_inf(&acc);
        for (int w = 0; w < NW; w++) {
            unsigned d = (raw[w / 8] >> ((w % 8) * 8)) & 0xff;
            if (d) g1j_madd(&acc, &acc, &table[w * 255 + d - 1], C);
        }
        res[i] = acc;
    }
    g1j_batch_to_affine(out, res, n, C);
    free(res); free(table);
}
