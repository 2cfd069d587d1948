/* oracle/zko_marlin.c -- TEST INFRASTRUCTURE (CPU oracle).
 *
 * CPU restatement of what `simpleworks::marlin::{generate_universal_srs, generate_proving_and_verifying_keys,
 * generate_proof}` do for the reference (call sites: /root/reference/src/lib.rs:111,141,173):
 *   ark-marlin 0.3.0 (Entropy1729 fork bde002de, Cargo.lock:199-213): AHP indexer / prover rounds 1-3,
 *     Fiat-Shamir transcript (SimpleHashFiatShamirRng<Blake2s, ChaChaRng>),
 *   ark-poly-commit 0.3.0 (Cargo.lock:248): MarlinKZG10 setup / trim / commit / open_combinations,
 * none of which is under /root/reference -> restated from the published algorithms;
 * "parity unpinned" at this boundary.  The SRS follows KZG10::setup's draw order from ark_std::test_rng(): beta, then g and gamma_g as
 * random curve points (ark-ec GroupProjective::rand: x, sign bit, cofactor clearing) -- restated from memory like the rest, SURVEY §8f item 4.
 */
#include "zko_marlin.h"
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include "zko_consts.h"

static fr_t *fr_alloc(size_t n) { fr_t *p = calloc(n ? n : 1, sizeof(fr_t)); if (!p) abort(); return p; }
static zko_poly poly_new(size_t len) { zko_poly p = {fr_alloc(len), len}; return p; }
static void poly_free(zko_poly *p) { free(p->c); p->c = NULL; p->len = 0; }
static size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

size_t zko_ahp_max_degree(size_t nc, size_t nv, size_t nnz, const fr_params *F) {
    (void)F;
    size_t dim = nc > nv ? nc : nv, zk = 1;
    size_t h = next_pow2(dim), k = next_pow2(nnz);
    size_t c[5] = {2 * h + zk - 2, 3 * h + 2 * zk - 3, h, h, 3 * k - 3}, m = 0;
    for (int i = 0; i < 5; i++) if (c[i] > m) m = c[i];
    return m;
}

/* ---------------------------------------------------------------- KZG10 commit (no hiding part) */
/* MSM of coeffs against powers_of_g starting at `off` in the plain or shifted table, skipping leading zeros */
static void commit_range(g1j_t *out, const zko_curve *C, const g1a_t *powers, const fr_t *coeffs, size_t len) {
    size_t lead = 0;
    while (lead < len && fr_is_zero(&coeffs[lead])) lead++;
    zko_msm(out, powers + lead, coeffs + lead, len - lead, C);
}
void zko_commit_plain(g1a_t *out, const zko_ck *ck, const fr_t *coeffs, size_t len, size_t off, int use_shifted) {
    g1j_t j;
    commit_range(&j, ck->C, (use_shifted ? ck->shifted_powers : ck->powers) + off, coeffs, len);
    g1j_to_affine(out, &j, ck->C);
}
typedef struct { int hiding; fr_t b[3]; } kzg_rand;   /* blinding polynomial of degree hiding_bound+1 = 2 */
/* KZG10::commit with optional hiding (hiding_bound = 1) */
static void kzg_commit(g1a_t *out, kzg_rand *rnd, const zko_ck *ck, const g1a_t *powers, const fr_t *coeffs, size_t len, int hiding, zko_chacha *rng) {
    const zko_curve *C = ck->C;
    g1j_t acc;
    commit_range(&acc, C, powers, coeffs, len);
    memset(rnd, 0, sizeof *rnd);
    if (hiding) {
        rnd->hiding = 1;
        for (int i = 0; i < 3; i++) zko_fr_rand(&rnd->b[i], rng, C->fr);
        g1j_t r;
        zko_msm(&r, ck->gamma_powers, rnd->b, 3, C);
        g1a_t ra; g1j_to_affine(&ra, &r, C);
        g1j_madd(&acc, &acc, &ra, C);
    }
    g1j_to_affine(out, &acc, C);
}
typedef struct { zko_poly p; long bound; int hiding; kzg_rand rand, shifted_rand; zko_commitment comm; } lpoly;
/* MarlinKZG10::commit for one labeled polynomial */
static void mpc_commit(lpoly *lp, const zko_ck *ck, zko_chacha *rng) {
    kzg_commit(&lp->comm.comm, &lp->rand, ck, ck->powers, lp->p.c, lp->p.len, lp->hiding, rng);
    lp->comm.has_shifted = 0;
    memset(&lp->shifted_rand, 0, sizeof lp->shifted_rand);
    if (lp->bound >= 0) {
        size_t off = (ck->max_degree - (size_t)lp->bound) - ck->lowest_shift;
        kzg_commit(&lp->comm.shifted, &lp->shifted_rand, ck, ck->shifted_powers + off, lp->p.c, lp->p.len, lp->hiding, rng);
        lp->comm.has_shifted = 1;
    }
}

/* ---------------------------------------------------------------- byte encodings */
typedef struct { uint8_t *b; size_t n, cap; } bytes;
static void by_put(bytes *o, const void *p, size_t n) {
    if (o->n + n > o->cap) { o->cap = (o->n + n) * 2 + 64; o->b = realloc(o->b, o->cap); }
    memcpy(o->b + o->n, p, n); o->n += n;
}
static void by_u64(bytes *o, uint64_t v) { uint8_t t[8]; for (int i = 0; i < 8; i++) t[i] = v >> (8 * i); by_put(o, t, 8); }
static void by_fr(bytes *o, const fr_t *a, const fr_params *F) { uint64_t raw[4]; fr_to_raw(raw, a, F); for (int i = 0; i < 4; i++) by_u64(o, raw[i]); }
static void by_fq(bytes *o, const fq_t *a, const fq_params *F) { uint64_t raw[6]; fq_to_raw(raw, a, F); for (int i = 0; i < 6; i++) by_u64(o, raw[i]); }
/* ark-ff ToBytes for GroupAffine: x, y (canonical LE), infinity flag byte */
static void by_g1_tobytes(bytes *o, const g1a_t *p, const zko_curve *C) {
    by_fq(o, &p->x, C->fq); by_fq(o, &p->y, C->fq);
    uint8_t f = (uint8_t)p->inf; by_put(o, &f, 1);
}
/* marlin_pc::Commitment ToBytes: comm, shifted_exists, shifted_comm or Commitment::empty() */
static void by_commitment_tobytes(bytes *o, const zko_commitment *c, const zko_curve *C) {
    by_g1_tobytes(o, &c->comm, C);
    uint8_t e = (uint8_t)c->has_shifted; by_put(o, &e, 1);
    if (c->has_shifted) by_g1_tobytes(o, &c->shifted, C);
    else { g1a_t z; memset(&z, 0, sizeof z); fq_set_one(&z.y, C->fq); z.inf = 1; by_g1_tobytes(o, &z, C); }
}
/* ark-serialize 0.3 compressed short-Weierstrass point: x LE with flags in the top bits of the last byte */
static void by_g1_compressed(bytes *o, const g1a_t *p, const zko_curve *C) {
    uint8_t buf[48];
    memset(buf, 0, 48);
    if (p->inf) { buf[47] |= 1 << 6; by_put(o, buf, 48); return; }
    uint64_t x[6], y[6], ny[6];
    fq_to_raw(x, &p->x, C->fq); fq_to_raw(y, &p->y, C->fq);
    fq_t n; fq_neg(&n, &p->y, C->fq); fq_to_raw(ny, &n, C->fq);
    int y_gt = 0;
    for (int i = 5; i >= 0; i--) { if (y[i] != ny[i]) { y_gt = y[i] > ny[i]; break; } }
    for (int i = 0; i < 48; i++) buf[i] = x[i / 8] >> (8 * (i % 8));
    if (y_gt) buf[47] |= 1 << 7;
    by_put(o, buf, 48);
}
/* ark-ec 0.3 GroupAffine::serialize_uncompressed: x, then y with the flags byte (infinity only; zero() = (0, 1, infinity)) -- 96 bytes, what deserialize_unchecked reads */
static void by_g1_uncompressed(bytes *o, const g1a_t *p, const zko_curve *C) {
    uint8_t buf[96];
    uint64_t x[6], y[6];
    if (p->inf) { fq_t one; fq_set_one(&one, C->fq); memset(x, 0, sizeof x); fq_to_raw(y, &one, C->fq); }
    else { fq_to_raw(x, &p->x, C->fq); fq_to_raw(y, &p->y, C->fq); }
    for (int i = 0; i < 48; i++) { buf[i] = x[i / 8] >> (8 * (i % 8)); buf[48 + i] = y[i / 8] >> (8 * (i % 8)); }
    if (p->inf) buf[95] |= 1 << 6;
    by_put(o, buf, 96);
}
static void by_g1_mode(bytes *o, const g1a_t *p, const zko_curve *C, int uncompressed) { if (uncompressed) by_g1_uncompressed(o, p, C); else by_g1_compressed(o, p, C); }
/* ark-serialize image of ark_marlin::IndexProverKey WITHOUT its first field (index_vk: the G2 side lives in the product), streamed to a file -- the checker for the product's
 * zkaes_pk_serialize_ark_to_file.  Layout [RECALL]: index_comm_rands (6 x empty Randomness), index (info, A, B, C as Vec<Vec<(Fr, usize)>>, six LabeledPolynomials, six
 * Evaluations in the order row, col, row_col, val_a, val_b, val_c with their Radix2 domain), committer key (powers, Some(shifted_powers), powers_of_gamma_g, Some(bounds), max_degree). */
static void fl_flush(bytes *o, FILE *f, uint64_t *total, int force) {
    if (o->n && (force || o->n > (8u << 20))) { if (fwrite(o->b, 1, o->n, f) != o->n) abort(); *total += o->n; o->n = 0; }
}
static void by_u32(bytes *o, uint32_t v) { uint8_t t[4]; for (int i = 0; i < 4; i++) t[i] = v >> (8 * i); by_put(o, t, 4); }
uint64_t zko_pk_serialize_ark_to_file(const zko_index *ix, const char *path) { return zko_pk_serialize_ark_to_file_mode(ix, path, 0); }
/* uncompressed != 0: serialize_uncompressed's image (only the G1 points differ: 96 bytes each) */
uint64_t zko_pk_serialize_ark_to_file_mode(const zko_index *ix, const char *path, int uncompressed) {
    const zko_curve *C = ix->ck.C;
    const fr_params *F = C->fr;
    FILE *f = fopen(path, "wb");
    if (!f) return 0;
    bytes o = {0};
    uint64_t total = 0;
    by_u64(&o, 6);
    for (int i = 0; i < 6; i++) { by_u64(&o, 0); uint8_t z = 0; by_put(&o, &z, 1); }
    by_u64(&o, ix->num_variables); by_u64(&o, ix->num_constraints); by_u64(&o, ix->num_non_zero); by_u64(&o, ix->num_instance);
    const zko_mat *M[3] = {&ix->cs->A, &ix->cs->B, &ix->cs->C};
    for (int q = 0; q < 3; q++) {
        by_u64(&o, M[q]->n);
        for (size_t r = 0; r < M[q]->n; r++) {
            by_u64(&o, M[q]->rowptr[r + 1] - M[q]->rowptr[r]);
            for (size_t i = M[q]->rowptr[r]; i < M[q]->rowptr[r + 1]; i++) {
                fr_t c; fr_from_i64(&c, M[q]->coeff[i], F);
                by_fr(&o, &c, F);
                uint32_t v = M[q]->var[i];
                by_u64(&o, v < ZKO_WIT_BASE ? v : (uint64_t)ix->cs->num_instance + (v - ZKO_WIT_BASE));
            }
            fl_flush(&o, f, &total, 0);
        }
    }
    const zko_poly *P[6] = {&ix->row, &ix->col, &ix->val_a, &ix->val_b, &ix->val_c, &ix->row_col};
    static const char *labels[6] = {"row", "col", "a_val", "b_val", "c_val", "row_col"};
    for (int i = 0; i < 6; i++) {
        size_t len = P[i]->len;
        while (len && fr_is_zero(&P[i]->c[len - 1])) len--;
        by_u64(&o, strlen(labels[i])); by_put(&o, labels[i], strlen(labels[i]));
        by_u64(&o, len);
        for (size_t j = 0; j < len; j++) { by_fr(&o, &P[i]->c[j], F); fl_flush(&o, f, &total, 0); }
        uint8_t none[2] = {0, 0}; by_put(&o, none, 2);
    }
    const fr_t *E[6] = {ix->row_evals, ix->col_evals, ix->row_col_evals, ix->val_a_evals, ix->val_b_evals, ix->val_c_evals};
    fr_t cg_inv; fr_inv(&cg_inv, &ix->K.coset_gen, F);
    for (int i = 0; i < 6; i++) {
        by_u64(&o, ix->K.size);
        for (size_t j = 0; j < ix->K.size; j++) { by_fr(&o, &E[i][j], F); fl_flush(&o, f, &total, 0); }
        uint8_t radix2 = 0; by_put(&o, &radix2, 1);
        by_u64(&o, ix->K.size); by_u32(&o, (uint32_t)ix->K.log_size);
        by_fr(&o, &ix->K.size_as_fe, F); by_fr(&o, &ix->K.size_inv, F); by_fr(&o, &ix->K.gen, F); by_fr(&o, &ix->K.gen_inv, F); by_fr(&o, &cg_inv, F);
    }
    const zko_ck *ck = &ix->ck;
    by_u64(&o, ck->supported_degree + 1);
    for (size_t i = 0; i <= ck->supported_degree; i++) { by_g1_mode(&o, &ck->powers[i], C, uncompressed); fl_flush(&o, f, &total, 0); }
    { uint8_t some = 1; by_put(&o, &some, 1); }
    const size_t nshift = ck->max_degree - ck->lowest_shift + 1;
    by_u64(&o, nshift);
    for (size_t i = 0; i < nshift; i++) { by_g1_mode(&o, &ck->shifted_powers[i], C, uncompressed); fl_flush(&o, f, &total, 0); }
    by_u64(&o, 3);
    for (int i = 0; i < 3; i++) by_g1_mode(&o, &ck->gamma_powers[i], C, uncompressed);
    { uint8_t some = 1; by_put(&o, &some, 1); }
    by_u64(&o, 2); by_u64(&o, ck->bounds[0]); by_u64(&o, ck->bounds[1]);
    by_u64(&o, ck->max_degree);
    fl_flush(&o, f, &total, 1);
    free(o.b);
    fclose(f);
    return total;
}
size_t zko_proof_serialize(const zko_proof *p, const zko_curve *C, uint8_t *out, size_t cap) {
    bytes o = {0};
    static const int round_len[3] = {4, 3, 2};
    by_u64(&o, 3);                                          /* commitments: Vec<Vec<Commitment>> */
    int ci = 0;
    for (int r = 0; r < 3; r++) {
        by_u64(&o, round_len[r]);
        for (int i = 0; i < round_len[r]; i++, ci++) {
            by_g1_compressed(&o, &p->comms[ci].comm, C);
            uint8_t tag = (uint8_t)p->comms[ci].has_shifted; by_put(&o, &tag, 1);
            if (tag) by_g1_compressed(&o, &p->comms[ci].shifted, C);
        }
    }
    by_u64(&o, 4);                                          /* evaluations: Vec<Fr> */
    for (int i = 0; i < 4; i++) by_fr(&o, &p->evals[i], C->fr);
    by_u64(&o, 3);                                          /* prover_messages: 3 x EmptyMessage = Option::None */
    for (int i = 0; i < 3; i++) { uint8_t z = 0; by_put(&o, &z, 1); }
    by_u64(&o, 2);                                          /* pc_proof.proof: Vec<kzg10::Proof> */
    by_g1_compressed(&o, &p->w_beta, C); { uint8_t t = 1; by_put(&o, &t, 1); } by_fr(&o, &p->random_v_beta, C->fr);
    by_g1_compressed(&o, &p->w_gamma, C); { uint8_t t = 0; by_put(&o, &t, 1); }
    { uint8_t t = 0; by_put(&o, &t, 1); }                    /* pc_proof.evals: None */
    size_t n = o.n;
    if (out && n <= cap) memcpy(out, o.b, n);
    free(o.b);
    return n;
}

/* ---------------------------------------------------------------- setup + index */
static void fr_from_small(fr_t *r, int64_t v, const fr_params *F) { fr_from_i64(r, v, F); }
static void srs_powers(g1a_t *out, const zko_curve *C, const g1a_t *base, const fr_t *beta, size_t from, size_t count) {
    fr_t *sc = fr_alloc(count);
    fr_t cur; fr_pow_u64(&cur, beta, from, C->fr);
    for (size_t i = 0; i < count; i++) { sc[i] = cur; fr_mul(&cur, &cur, beta, C->fr); }
    zko_fixed_base_batch(out, base, sc, count, C);
    free(sc);
}
static void poly_from_evals(zko_poly *out, const fr_t *evals, const zko_domain *D) {
    *out = poly_new(D->size);
    memcpy(out->c, evals, D->size * sizeof(fr_t));
    zko_ifft(D, out->c);
}
zko_index *zko_marlin_index(zko_cs *cs, size_t srs_nc, size_t srs_nv, size_t srs_nnz) {
    const zko_curve *C = zko_curve_by_id(cs->field_id);
    const fr_params *F = C->fr;
    zko_index *ix = calloc(1, sizeof *ix);
    zko_cs_pad_for_marlin(cs);
    ix->cs = cs;
    size_t nrows = cs->A.n;
    /* joint matrix = per-row sorted union of the column supports of A, B, C (sum_matrices) */
    size_t *jptr = malloc((nrows + 1) * sizeof(size_t));
    size_t jcap = cs->A.nnz + cs->B.nnz + cs->C.nnz + 1;
    uint32_t *jcol = malloc(jcap * 4);
    size_t jn = 0;
    jptr[0] = 0;
    for (size_t r = 0; r < nrows; r++) {
        size_t ia = cs->A.rowptr[r], ib = cs->B.rowptr[r], ic = cs->C.rowptr[r];
        size_t ea = cs->A.rowptr[r + 1], eb = cs->B.rowptr[r + 1], ec = cs->C.rowptr[r + 1];
        for (;;) {
            uint32_t best = 0xffffffffu;
            if (ia < ea && zko_var_col(cs, cs->A.var[ia]) < best) best = zko_var_col(cs, cs->A.var[ia]);
            if (ib < eb && zko_var_col(cs, cs->B.var[ib]) < best) best = zko_var_col(cs, cs->B.var[ib]);
            if (ic < ec && zko_var_col(cs, cs->C.var[ic]) < best) best = zko_var_col(cs, cs->C.var[ic]);
            if (best == 0xffffffffu) break;
            jcol[jn++] = best;
            if (ia < ea && zko_var_col(cs, cs->A.var[ia]) == best) ia++;
            if (ib < eb && zko_var_col(cs, cs->B.var[ib]) == best) ib++;
            if (ic < ec && zko_var_col(cs, cs->C.var[ic]) == best) ic++;
        }
        jptr[r + 1] = jn;
    }
    ix->num_instance = cs->num_instance;
    ix->num_variables = (size_t)cs->num_instance + cs->num_witness;
    ix->num_constraints = nrows;
    ix->num_non_zero = jn;
    if (ix->num_constraints != ix->num_variables) { fprintf(stderr, "zko: non-square after padding\n"); abort(); }
    zko_domain_init(&ix->H, ix->num_constraints, F);
    zko_domain_init(&ix->K, ix->num_non_zero, F);
    zko_domain_init(&ix->X, ix->num_instance, F);
    size_t n = ix->H.size, k = ix->K.size;
    /* elements of H and u_H(x,x) = |H| x^(|H|-1) */
    fr_t *elems = fr_alloc(n);
    fr_set_one(&elems[0], F);
    for (size_t i = 1; i < n; i++) fr_mul(&elems[i], &elems[i - 1], &ix->H.gen, F);
    ix->row_evals = fr_alloc(k); ix->col_evals = fr_alloc(k); ix->row_col_evals = fr_alloc(k);
    ix->val_a_evals = fr_alloc(k); ix->val_b_evals = fr_alloc(k); ix->val_c_evals = fr_alloc(k);
    fr_t *inv = fr_alloc(k), *scr = fr_alloc(k);
    size_t cnt = 0;
    for (size_t r = 0; r < nrows; r++) {
        size_t ia = cs->A.rowptr[r], ib = cs->B.rowptr[r], ic = cs->C.rowptr[r];
        for (size_t q = jptr[r]; q < jptr[r + 1]; q++, cnt++) {
            uint32_t col = jcol[q];
            size_t ci = zko_reindex_by_subdomain(&ix->H, &ix->X, col);
            ix->row_evals[cnt] = elems[ci];           /* transposed: row <- col_val */
            ix->col_evals[cnt] = elems[r];
            if (ia < cs->A.rowptr[r + 1] && zko_var_col(cs, cs->A.var[ia]) == col) { fr_from_small(&ix->val_a_evals[cnt], cs->A.coeff[ia], F); ia++; }
            if (ib < cs->B.rowptr[r + 1] && zko_var_col(cs, cs->B.var[ib]) == col) { fr_from_small(&ix->val_b_evals[cnt], cs->B.coeff[ib], F); ib++; }
            if (ic < cs->C.rowptr[r + 1] && zko_var_col(cs, cs->C.var[ic]) == col) { fr_from_small(&ix->val_c_evals[cnt], cs->C.coeff[ic], F); ic++; }
            fr_mul(&inv[cnt], &elems[(n - ci) % n], &ix->H.size_as_fe, F);   /* u_H(col_val, col_val) */
        }
    }
    fr_batch_inv(inv, cnt, scr, F);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < cnt; i++) {
        fr_mul(&ix->val_a_evals[i], &ix->val_a_evals[i], &inv[i], F);
        fr_mul(&ix->val_b_evals[i], &ix->val_b_evals[i], &inv[i], F);
        fr_mul(&ix->val_c_evals[i], &ix->val_c_evals[i], &inv[i], F);
    }
    for (size_t i = cnt; i < k; i++) { ix->row_evals[i] = elems[0]; ix->col_evals[i] = elems[0]; }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < k; i++) fr_mul(&ix->row_col_evals[i], &ix->row_evals[i], &ix->col_evals[i], F);
    free(inv); free(scr); free(elems); free(jptr); free(jcol);
    poly_from_evals(&ix->row, ix->row_evals, &ix->K); poly_from_evals(&ix->col, ix->col_evals, &ix->K);
    poly_from_evals(&ix->row_col, ix->row_col_evals, &ix->K);
    poly_from_evals(&ix->val_a, ix->val_a_evals, &ix->K); poly_from_evals(&ix->val_b, ix->val_b_evals, &ix->K);
    poly_from_evals(&ix->val_c, ix->val_c_evals, &ix->K);

    /* universal_setup(literals) -> KZG10::setup(max_degree); trim(index.max_degree(), hiding 1, bounds) */
    zko_ck *ck = &ix->ck;
    ck->C = C;
    ck->max_degree = zko_ahp_max_degree(srs_nc, srs_nv, srs_nnz, F);
    size_t index_max = zko_ahp_max_degree(ix->num_constraints, ix->num_variables, ix->num_non_zero, F);
    if (index_max > ck->max_degree) { fprintf(stderr, "zko: IndexTooLarge %zu > %zu\n", index_max, ck->max_degree); abort(); }
    zko_chacha rng; zko_chacha_init(&rng, ARK_TEST_RNG_SEED, 12);
    zko_fr_rand(&ck->beta, &rng, F);                     /* KZG10::setup draw order [RECALL ark-poly-commit 0.3.0]: beta, g, gamma_g, h */
    zko_g1_rand(&ck->g, &rng, C);
    zko_g1_rand(&ck->gamma_g, &rng, C);                  /* (h is a G2 point: only the verifying key needs it) */
    ck->supported_degree = index_max;
    ck->powers = malloc((index_max + 1) * sizeof(g1a_t));
    srs_powers(ck->powers, C, &ck->g, &ck->beta, 0, index_max + 1);
    srs_powers(ck->gamma_powers, C, &ck->gamma_g, &ck->beta, 0, 3);
    ck->bounds[0] = n - 2; ck->bounds[1] = k - 2;
    if (ck->bounds[0] > ck->bounds[1]) { size_t t = ck->bounds[0]; ck->bounds[0] = ck->bounds[1]; ck->bounds[1] = t; }
    ck->lowest_shift = ck->max_degree - ck->bounds[1];
    size_t ns = ck->bounds[1] + 1;
    ck->shifted_powers = malloc(ns * sizeof(g1a_t));
    srs_powers(ck->shifted_powers, C, &ck->g, &ck->beta, ck->lowest_shift, ns);
    /* index commitments (no hiding, no bounds) */
    const zko_poly *ip[6] = {&ix->row, &ix->col, &ix->val_a, &ix->val_b, &ix->val_c, &ix->row_col};
    for (int i = 0; i < 6; i++) zko_commit_plain(&ix->index_comms[i], ck, ip[i]->c, ip[i]->len, 0, 0);
    return ix;
}
void zko_index_free(zko_index *ix) {
    if (!ix) return;
    zko_cs_free(ix->cs);
    free(ix->row_evals); free(ix->col_evals); free(ix->row_col_evals); free(ix->val_a_evals); free(ix->val_b_evals); free(ix->val_c_evals);
    poly_free(&ix->row); poly_free(&ix->col); poly_free(&ix->row_col); poly_free(&ix->val_a); poly_free(&ix->val_b); poly_free(&ix->val_c);
    free(ix->ck.powers); free(ix->ck.shifted_powers);
    free(ix);
}

/* ---------------------------------------------------------------- prover helpers */
/* q = p / (X^m - 1), rem (len m) */
static void divide_by_vanishing(zko_poly *q, fr_t *rem, const fr_t *p, size_t len, size_t m, const fr_params *F) {
    if (len < m + 1) { *q = poly_new(0); if (rem) { memset(rem, 0, m * sizeof(fr_t)); memcpy(rem, p, len * sizeof(fr_t)); } return; }
    *q = poly_new(len - m);
    for (size_t i = len - m; i-- > 0;) {
        q->c[i] = p[i + m];
        if (i + m < len - m) fr_add(&q->c[i], &q->c[i], &q->c[i + m], F);
    }
    if (rem) for (size_t i = 0; i < m; i++) { rem[i] = p[i]; if (i < len - m) fr_add(&rem[i], &rem[i], &q->c[i], F); }
}
static void fft_padded(fr_t *out, const zko_domain *D, const fr_t *c, size_t len) {
    memset(out, 0, D->size * sizeof(fr_t));
    memcpy(out, c, (len < D->size ? len : D->size) * sizeof(fr_t));
    zko_fft(D, out);
}
/* quotient of p by (X - z), remainder dropped (ark `p / &divisor`) */
static void divide_by_linear(zko_poly *q, const fr_t *p, size_t len, const fr_t *z, const fr_params *F) {
    if (len < 2) { *q = poly_new(0); return; }
    *q = poly_new(len - 1);
    fr_t carry = p[len - 1];
    for (size_t i = len - 1; i-- > 0;) {
        q->c[i] = carry;
        fr_t t; fr_mul(&t, &carry, z, F); fr_add(&carry, &p[i], &t, F);
    }
}
static void poly_axpy(zko_poly *acc, const fr_t *s, const fr_t *p, size_t len, const fr_params *F) {
    if (len > acc->len) { acc->c = realloc(acc->c, len * sizeof(fr_t)); memset(acc->c + acc->len, 0, (len - acc->len) * sizeof(fr_t)); acc->len = len; }
#pragma omp parallel for schedule(static) if (len >= 4096)
    for (size_t i = 0; i < len; i++) { fr_t t; fr_mul(&t, s, &p[i], F); fr_add(&acc->c[i], &acc->c[i], &t, F); }
}
static void fs_absorb_commitments(zko_fsrng *fs, const lpoly *lp, int n, const zko_curve *C) {
    bytes o = {0};
    for (int i = 0; i < n; i++) by_commitment_tobytes(&o, &lp[i].comm, C);
    zko_fs_absorb(fs, o.b, o.n);
    free(o.b);
}
static void sample_outside(fr_t *out, zko_fsrng *fs, const zko_domain *D) {
    fr_t v;
    do { zko_fr_rand(out, &fs->r, D->F); zko_domain_eval_vanishing(&v, D, out); } while (fr_is_zero(&v));
}

/* ZKO_TIMING=1: per-phase wall times of the prover on stderr (finding what does not scale with the thread count) */
static double tm_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#define TM(label) do { if (tm_on) { double t_ = tm_now(); fprintf(stderr, "zko timing %-28s %8.3f s\n", label, t_ - tm_last); tm_last = t_; } } while (0)
zko_proof *zko_marlin_prove(const zko_index *ix, zko_cs *cs, const uint8_t *zk_seed) {
    const int tm_on = getenv("ZKO_TIMING") != NULL; double tm_last = tm_now();
    const zko_curve *C = ix->ck.C;
    const fr_params *F = C->fr;
    const zko_ck *ck = &ix->ck;
    const zko_cs *ics = ix->cs;
    zko_cs_pad_for_marlin(cs);
    if ((size_t)cs->num_instance + cs->num_witness != ix->num_variables || cs->A.n != ix->num_constraints) { fprintf(stderr, "zko: InstanceDoesNotMatchIndex\n"); return NULL; }
    zko_proof *pf = calloc(1, sizeof *pf);
    zko_chacha zk; zko_chacha_init(&zk, zk_seed ? zk_seed : ARK_TEST_RNG_SEED, 12);
    const zko_domain *H = &ix->H, *K = &ix->K, *X = &ix->X;
    size_t n = H->size, k = K->size, m = X->size, nrows = ix->num_constraints;
    fr_t one, zero; fr_set_one(&one, F); fr_set_zero(&zero);

    /* ---- prover_init: z, z_A = A z, z_B = B z (index matrices, prover assignment) */
    fr_t *small = fr_alloc(8);   /* Montgomery forms of -2..5 for the tiny coefficients */
    (void)small;
    fr_t *za = fr_alloc(n), *zb = fr_alloc(n);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < nrows; r++) {
        const zko_mat *M[2] = {&ics->A, &ics->B};
        fr_t *dst[2] = {&za[r], &zb[r]};
        for (int q = 0; q < 2; q++) {
            __int128 s = 0;       /* assignments are 0/1 and coefficients are small integers: exact in Z, then mapped to Fr */
            for (size_t i = M[q]->rowptr[r]; i < M[q]->rowptr[r + 1]; i++) {
                uint32_t v = M[q]->var[i];
                int val = v < ZKO_WIT_BASE ? cs->instance_val[v] : cs->witness_val[v - ZKO_WIT_BASE];
                s += (__int128)M[q]->coeff[i] * val;
            }
            fr_from_i64(dst[q], (int64_t)s, F);
        }
    }
    TM("pad + z_A z_B");
    /* ---- FS init: "MARLIN-2019" || index_vk || public_input (padded instance without the leading One) */
    zko_fsrng fs;
    {
        bytes o = {0};
        by_put(&o, "MARLIN-2019", 11);
        by_u64(&o, ix->num_variables); by_u64(&o, ix->num_constraints); by_u64(&o, ix->num_non_zero);
        for (int i = 0; i < 6; i++) { zko_commitment c; memset(&c, 0, sizeof c); c.comm = ix->index_comms[i]; by_commitment_tobytes(&o, &c, C); }
        for (size_t i = 1; i < m; i++) { fr_t v = cs->instance_val[i] ? one : zero; by_fr(&o, &v, F); }
        zko_fs_init(&fs, o.b, o.n);
        free(o.b);
    }
    /* ---- first round */
    lpoly r1[4]; memset(r1, 0, sizeof r1);
    zko_poly x_poly = poly_new(m);
    for (size_t i = 0; i < m; i++) x_poly.c[i] = cs->instance_val[i] ? one : zero;
    zko_ifft(X, x_poly.c);
    fr_t *x_evals = fr_alloc(n);
    fft_padded(x_evals, H, x_poly.c, m);
    size_t ratio = n / m;
    fr_t *wev = fr_alloc(n + 1);
    for (size_t kk = 0; kk < n; kk++) {
        if (kk % ratio == 0) { fr_set_zero(&wev[kk]); continue; }
        size_t wi = kk - kk / ratio - 1;
        fr_t w = (wi < cs->num_witness && cs->witness_val[wi]) ? one : zero;
        fr_sub(&wev[kk], &w, &x_evals[kk], F);
    }
    zko_ifft(H, wev);
    fr_t rho; zko_fr_rand(&rho, &zk, F);                        /* + rho * v_H */
    fr_sub(&wev[0], &wev[0], &rho, F); wev[n] = rho;
    fr_t *rem = fr_alloc(m);
    divide_by_vanishing(&r1[0].p, rem, wev, n + 1, m, F);
    for (size_t i = 0; i < m; i++) if (!fr_is_zero(&rem[i])) { fprintf(stderr, "zko: w not divisible by v_X\n"); abort(); }
    free(rem); free(wev);
    r1[1].p = poly_new(n + 1); memcpy(r1[1].p.c, za, n * sizeof(fr_t)); zko_ifft(H, r1[1].p.c);
    zko_fr_rand(&rho, &zk, F); fr_sub(&r1[1].p.c[0], &r1[1].p.c[0], &rho, F); r1[1].p.c[n] = rho;
    r1[2].p = poly_new(n + 1); memcpy(r1[2].p.c, zb, n * sizeof(fr_t)); zko_ifft(H, r1[2].p.c);
    zko_fr_rand(&rho, &zk, F); fr_sub(&r1[2].p.c[0], &r1[2].p.c[0], &rho, F); r1[2].p.c[n] = rho;
    TM("round 1 polynomials");
    r1[3].p = poly_new(3 * n);                                  /* mask: degree 3|H| + 2 zk - 3 */
    for (size_t i = 0; i < 3 * n; i++) zko_fr_rand(&r1[3].p.c[i], &zk, F);
    { fr_t s = r1[3].p.c[0]; fr_add(&s, &s, &r1[3].p.c[n], F); fr_add(&s, &s, &r1[3].p.c[2 * n], F); fr_sub(&r1[3].p.c[0], &r1[3].p.c[0], &s, F); }
    TM("mask draws");
    for (int i = 0; i < 4; i++) { r1[i].bound = -1; r1[i].hiding = i < 3; }
    for (int i = 0; i < 4; i++) mpc_commit(&r1[i], ck, &zk);
    TM("round 1 commitments");
    fs_absorb_commitments(&fs, r1, 4, C);
    fr_t alpha, eta_a, eta_b, eta_c;
    sample_outside(&alpha, &fs, H);
    zko_fr_rand(&eta_a, &fs.r, F); zko_fr_rand(&eta_b, &fs.r, F); zko_fr_rand(&eta_c, &fs.r, F);

    /* ---- second round */
    lpoly r2[3]; memset(r2, 0, sizeof r2);
    fr_t vh_alpha; zko_domain_eval_vanishing(&vh_alpha, H, &alpha);
    fr_t *r_alpha_evals = fr_alloc(n), *scr = fr_alloc(n);
    { fr_t e; fr_set_one(&e, F); for (size_t i = 0; i < n; i++) { fr_sub(&r_alpha_evals[i], &alpha, &e, F); fr_mul(&e, &e, &H->gen, F); } }
    fr_batch_inv(r_alpha_evals, n, scr, F);
    for (size_t i = 0; i < n; i++) fr_mul(&r_alpha_evals[i], &r_alpha_evals[i], &vh_alpha, F);
    free(scr);
    TM("r(alpha, .) on H");
    /* calculate_t (row-major accumulate, as the reference's upstream loop) */
    fr_t *t_evals = fr_alloc(n);
    {
        const zko_mat *M[3] = {&ics->A, &ics->B, &ics->C};
        const fr_t *eta[3] = {&eta_a, &eta_b, &eta_c};
        for (int q = 0; q < 3; q++)
            for (size_t r = 0; r < nrows; r++) {
                fr_t er; fr_mul(&er, eta[q], &r_alpha_evals[r], F);
                for (size_t i = M[q]->rowptr[r]; i < M[q]->rowptr[r + 1]; i++) {
                    size_t idx = zko_reindex_by_subdomain(H, X, zko_var_col(ics, M[q]->var[i]));
                    fr_t c, t; fr_from_i64(&c, M[q]->coeff[i], F); fr_mul(&t, &er, &c, F);
                    fr_add(&t_evals[idx], &t_evals[idx], &t, F);
                }
            }
    }
    TM("calculate_t");
    r2[0].p = poly_new(n); memcpy(r2[0].p.c, t_evals, n * sizeof(fr_t)); zko_ifft(H, r2[0].p.c);
    free(t_evals);
    zko_poly r_alpha_poly = poly_new(n); memcpy(r_alpha_poly.c, r_alpha_evals, n * sizeof(fr_t)); zko_ifft(H, r_alpha_poly.c);
    /* z_poly = w * v_X + x */
    zko_poly z_poly = poly_new(n + 1);
    for (size_t i = 0; i < r1[0].p.len; i++) { fr_add(&z_poly.c[i + m], &z_poly.c[i + m], &r1[0].p.c[i], F); fr_sub(&z_poly.c[i], &z_poly.c[i], &r1[0].p.c[i], F); }
    for (size_t i = 0; i < m; i++) fr_add(&z_poly.c[i], &z_poly.c[i], &x_poly.c[i], F);
    zko_domain D4; zko_domain_init(&D4, 3 * n + 1, F);
    size_t n4 = D4.size;
    fr_t *e_za = fr_alloc(n4), *e_zb = fr_alloc(n4), *e_ra = fr_alloc(n4), *e_t = fr_alloc(n4), *e_z = fr_alloc(n4);
    fft_padded(e_za, &D4, r1[1].p.c, n + 1); fft_padded(e_zb, &D4, r1[2].p.c, n + 1);
    fft_padded(e_ra, &D4, r_alpha_poly.c, n); fft_padded(e_t, &D4, r2[0].p.c, n); fft_padded(e_z, &D4, z_poly.c, n + 1);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n4; i++) {
        fr_t s, t, u;
        fr_mul(&s, &e_za[i], &e_zb[i], F); fr_mul(&s, &s, &eta_c, F);
        fr_mul(&t, &eta_a, &e_za[i], F); fr_add(&s, &s, &t, F);
        fr_mul(&t, &eta_b, &e_zb[i], F); fr_add(&s, &s, &t, F);
        fr_mul(&s, &s, &e_ra[i], F);
        fr_mul(&u, &e_t[i], &e_z[i], F);
        fr_sub(&e_ra[i], &s, &u, F);
    }
    zko_ifft(&D4, e_ra);
    for (size_t i = 0; i < 3 * n; i++) fr_add(&e_ra[i], &e_ra[i], &r1[3].p.c[i], F);      /* q_1 = mask + rhs */
    {
        size_t qlen = n4; while (qlen > 0 && fr_is_zero(&e_ra[qlen - 1])) qlen--;
        fr_t *xg = fr_alloc(n);
        divide_by_vanishing(&r2[2].p, xg, e_ra, qlen, n, F);                                 /* h_1, x*g_1 */
        if (!fr_is_zero(&xg[0])) { fprintf(stderr, "zko: outer sumcheck remainder has a constant term\n"); abort(); }
        r2[1].p = poly_new(n - 1); memcpy(r2[1].p.c, xg + 1, (n - 1) * sizeof(fr_t));
        free(xg);
    }
    free(e_za); free(e_zb); free(e_ra); free(e_t); free(e_z);
    r2[0].bound = -1; r2[0].hiding = 0; r2[1].bound = (long)(n - 2); r2[1].hiding = 1; r2[2].bound = -1; r2[2].hiding = 0;
    TM("round 2 polynomials");
    for (int i = 0; i < 3; i++) mpc_commit(&r2[i], ck, &zk);
    TM("round 2 commitments");
    fs_absorb_commitments(&fs, r2, 3, C);
    fr_t beta; sample_outside(&beta, &fs, H);

    /* ---- third round */
    lpoly r3[2]; memset(r3, 0, sizeof r3);
    fr_t vh_beta; zko_domain_eval_vanishing(&vh_beta, H, &beta);
    fr_t vv, ea_vv, eb_vv, ec_vv, alpha_beta;
    fr_mul(&vv, &vh_alpha, &vh_beta, F); fr_mul(&ea_vv, &eta_a, &vv, F); fr_mul(&eb_vv, &eta_b, &vv, F); fr_mul(&ec_vv, &eta_c, &vv, F);
    fr_mul(&alpha_beta, &alpha, &beta, F);
    fr_t *f_evals = fr_alloc(k), *den = fr_alloc(k), *scr2 = fr_alloc(k);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < k; i++) {
        fr_t a, b; fr_sub(&a, &beta, &ix->row_evals[i], F); fr_sub(&b, &alpha, &ix->col_evals[i], F); fr_mul(&den[i], &a, &b, F);
    }
    fr_batch_inv(den, k, scr2, F);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < k; i++) {
        fr_t s, t;
        fr_mul(&s, &ea_vv, &ix->val_a_evals[i], F);
        fr_mul(&t, &eb_vv, &ix->val_b_evals[i], F); fr_add(&s, &s, &t, F);
        fr_mul(&t, &ec_vv, &ix->val_c_evals[i], F); fr_add(&s, &s, &t, F);
        fr_mul(&f_evals[i], &s, &den[i], F);
    }
    free(den); free(scr2);
    zko_poly f_poly = poly_new(k); memcpy(f_poly.c, f_evals, k * sizeof(fr_t)); zko_ifft(K, f_poly.c);
    free(f_evals);
    r3[0].p = poly_new(k - 1); memcpy(r3[0].p.c, f_poly.c + 1, (k - 1) * sizeof(fr_t));          /* g_2 */
    {
        zko_domain D2; zko_domain_init(&D2, 2 * k, F);
        size_t n2 = D2.size;
        fr_t *ea = fr_alloc(n2), *eb = fr_alloc(n2), *ef = fr_alloc(n2);
        zko_poly a_poly = poly_new(k), b_poly = poly_new(k);
        for (size_t i = 0; i < k; i++) {
            fr_t s, t;
            fr_mul(&s, &ea_vv, &ix->val_a.c[i], F); fr_mul(&t, &eb_vv, &ix->val_b.c[i], F); fr_add(&s, &s, &t, F);
            fr_mul(&t, &ec_vv, &ix->val_c.c[i], F); fr_add(&a_poly.c[i], &s, &t, F);
            fr_mul(&s, &alpha, &ix->row.c[i], F); fr_mul(&t, &beta, &ix->col.c[i], F); fr_add(&s, &s, &t, F);
            fr_sub(&b_poly.c[i], &ix->row_col.c[i], &s, F);
        }
        fr_add(&b_poly.c[0], &b_poly.c[0], &alpha_beta, F);
        fft_padded(ea, &D2, a_poly.c, k); fft_padded(eb, &D2, b_poly.c, k); fft_padded(ef, &D2, f_poly.c, k);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n2; i++) { fr_t t; fr_mul(&t, &eb[i], &ef[i], F); fr_sub(&ea[i], &ea[i], &t, F); }
        zko_ifft(&D2, ea);
        size_t qlen = n2; while (qlen > 0 && fr_is_zero(&ea[qlen - 1])) qlen--;
        fr_t *rm = fr_alloc(k);
        divide_by_vanishing(&r3[1].p, rm, ea, qlen, k, F);                                      /* h_2 */
        for (size_t i = 0; i < k; i++) if (!fr_is_zero(&rm[i])) { fprintf(stderr, "zko: inner sumcheck not divisible\n"); abort(); }
        free(rm); free(ea); free(eb); free(ef); poly_free(&a_poly); poly_free(&b_poly);
    }
    r3[0].bound = (long)(k - 2); r3[0].hiding = 0; r3[1].bound = -1; r3[1].hiding = 0;
    TM("round 3 polynomials");
    for (int i = 0; i < 2; i++) mpc_commit(&r3[i], ck, &zk);
    TM("round 3 commitments");
    fs_absorb_commitments(&fs, r3, 2, C);
    fr_t gamma; zko_fr_rand(&gamma, &fs.r, F);

    /* ---- evaluations (sorted by label: g_1, g_2, t, z_b) */
    fr_t g1_b, g2_g, t_b, zb_b;
    zko_poly_eval(&g1_b, r2[1].p.c, r2[1].p.len, &beta, F);
    zko_poly_eval(&g2_g, r3[0].p.c, r3[0].p.len, &gamma, F);
    zko_poly_eval(&t_b, r2[0].p.c, r2[0].p.len, &beta, F);
    zko_poly_eval(&zb_b, r1[2].p.c, r1[2].p.len, &beta, F);
    pf->evals[0] = g1_b; pf->evals[1] = g2_g; pf->evals[2] = t_b; pf->evals[3] = zb_b;
    { bytes o = {0}; for (int i = 0; i < 4; i++) by_fr(&o, &pf->evals[i], F); zko_fs_absorb(&fs, o.b, o.n); free(o.b); }
    fr_t ch;
    { uint64_t lo = zko_chacha_u64(&fs.r), hi = zko_chacha_u64(&fs.r); uint64_t raw[4] = {lo, hi, 0, 0}; fr_from_raw(&ch, raw, F); }   /* u128::rand -> F::from */

    /* ---- linear combinations (construct_linear_combinations) */
    fr_t r_alpha_at_beta, vx_beta, x_at_beta;
    { fr_t d, nmr; fr_sub(&nmr, &vh_alpha, &vh_beta, F); fr_sub(&d, &alpha, &beta, F); fr_inv(&d, &d, F); fr_mul(&r_alpha_at_beta, &nmr, &d, F); }
    zko_domain_eval_vanishing(&vx_beta, X, &beta);
    zko_poly_eval(&x_at_beta, x_poly.c, m, &beta, F);                 /* = sum_i L_i(beta) x_i */
    /* outer_sumcheck = mask + c_za z_a + c_w w + c_h1 h_1 (+ constants) */
    fr_t c_za, c_w, c_h1, tmp;
    fr_mul(&tmp, &eta_c, &zb_b, F); fr_add(&tmp, &tmp, &eta_a, F); fr_mul(&c_za, &r_alpha_at_beta, &tmp, F);
    fr_mul(&c_w, &t_b, &vx_beta, F); fr_neg(&c_w, &c_w, F);
    fr_neg(&c_h1, &vh_beta, F);
    /* inner_sumcheck */
    fr_t vk_gamma, k_inv, bmul, c_row, c_col, c_h2;
    zko_domain_eval_vanishing(&vk_gamma, K, &gamma);
    fr_inv(&k_inv, &K->size_as_fe, F);
    fr_mul(&bmul, &gamma, &g2_g, F); fr_mul(&tmp, &t_b, &k_inv, F); fr_add(&bmul, &bmul, &tmp, F);     /* gamma g_2(gamma) + t(beta)/|K| */
    fr_mul(&c_row, &alpha, &bmul, F);            /* -(-alpha * bmul) */
    fr_mul(&c_col, &beta, &bmul, F);
    fr_neg(&c_h2, &vk_gamma, F);
    fr_t neg_bmul; fr_neg(&neg_bmul, &bmul, F);

    /* ---- open at beta: labels g_1 (ch^0, shifted ch^1), outer_sumcheck (ch^2), t (ch^3), z_b (ch^4) */
    fr_t chp[5]; chp[0] = one; for (int i = 1; i < 5; i++) fr_mul(&chp[i], &chp[i - 1], &ch, F);
    {
        zko_poly p = poly_new(3 * n);
        fr_t rb[3]; memset(rb, 0, sizeof rb);
        fr_t s;
        poly_axpy(&p, &chp[0], r2[1].p.c, r2[1].p.len, F);
        for (int i = 0; i < 3; i++) { fr_mul(&s, &chp[0], &r2[1].rand.b[i], F); fr_add(&rb[i], &rb[i], &s, F); }
        /* outer */
        poly_axpy(&p, &chp[2], r1[3].p.c, r1[3].p.len, F);
        fr_mul(&s, &chp[2], &c_za, F); poly_axpy(&p, &s, r1[1].p.c, r1[1].p.len, F);
        for (int i = 0; i < 3; i++) { fr_t u; fr_mul(&u, &s, &r1[1].rand.b[i], F); fr_add(&rb[i], &rb[i], &u, F); }
        fr_mul(&s, &chp[2], &c_w, F); poly_axpy(&p, &s, r1[0].p.c, r1[0].p.len, F);
        for (int i = 0; i < 3; i++) { fr_t u; fr_mul(&u, &s, &r1[0].rand.b[i], F); fr_add(&rb[i], &rb[i], &u, F); }
        fr_mul(&s, &chp[2], &c_h1, F); poly_axpy(&p, &s, r2[2].p.c, r2[2].p.len, F);
        poly_axpy(&p, &chp[3], r2[0].p.c, r2[0].p.len, F);
        poly_axpy(&p, &chp[4], r1[2].p.c, r1[2].p.len, F);
        for (int i = 0; i < 3; i++) { fr_t u; fr_mul(&u, &chp[4], &r1[2].rand.b[i], F); fr_add(&rb[i], &rb[i], &u, F); }
        size_t plen = p.len; while (plen > 0 && fr_is_zero(&p.c[plen - 1])) plen--;
        zko_poly wit, rwit, swit, srwit;
        divide_by_linear(&wit, p.c, plen, &beta, F);
        divide_by_linear(&rwit, rb, 3, &beta, F);
        g1j_t w, t2;
        commit_range(&w, C, ck->powers, wit.c, wit.len);
        zko_msm(&t2, ck->gamma_powers, rwit.c, rwit.len, C); g1j_add(&w, &w, &t2, C);
        fr_t rv; zko_poly_eval(&rv, rb, 3, &beta, F);
        /* shifted part: g_1 */
        divide_by_linear(&swit, r2[1].p.c, r2[1].p.len, &beta, F);
        fr_t srb[3]; for (int i = 0; i < 3; i++) fr_mul(&srb[i], &chp[1], &r2[1].shifted_rand.b[i], F);
        divide_by_linear(&srwit, srb, 3, &beta, F);
        for (size_t i = 0; i < swit.len; i++) fr_mul(&swit.c[i], &swit.c[i], &chp[1], F);
        size_t off = (ck->bounds[1] - (n - 2));              /* shift_polynomial: largest bound - degree bound leading zeros */
        commit_range(&t2, C, ck->shifted_powers + off, swit.c, swit.len); g1j_add(&w, &w, &t2, C);
        zko_msm(&t2, ck->gamma_powers, srwit.c, srwit.len, C); g1j_add(&w, &w, &t2, C);
        fr_t srv; zko_poly_eval(&srv, srb, 3, &beta, F); fr_add(&rv, &rv, &srv, F);
        g1j_to_affine(&pf->w_beta, &w, C); pf->random_v_beta = rv;
        poly_free(&p); poly_free(&wit); poly_free(&rwit); poly_free(&swit); poly_free(&srwit);
    }
    /* ---- open at gamma: g_2 (ch^0, shifted ch^1), inner_sumcheck (ch^2) */
    {
        zko_poly p = poly_new(k);
        fr_t s;
        poly_axpy(&p, &chp[0], r3[0].p.c, r3[0].p.len, F);
        fr_mul(&s, &chp[2], &ea_vv, F); poly_axpy(&p, &s, ix->val_a.c, k, F);
        fr_mul(&s, &chp[2], &eb_vv, F); poly_axpy(&p, &s, ix->val_b.c, k, F);
        fr_mul(&s, &chp[2], &ec_vv, F); poly_axpy(&p, &s, ix->val_c.c, k, F);
        fr_mul(&s, &chp[2], &c_row, F); poly_axpy(&p, &s, ix->row.c, k, F);
        fr_mul(&s, &chp[2], &c_col, F); poly_axpy(&p, &s, ix->col.c, k, F);
        fr_mul(&s, &chp[2], &neg_bmul, F); poly_axpy(&p, &s, ix->row_col.c, k, F);
        fr_mul(&s, &chp[2], &c_h2, F); poly_axpy(&p, &s, r3[1].p.c, r3[1].p.len, F);
        size_t plen = p.len; while (plen > 0 && fr_is_zero(&p.c[plen - 1])) plen--;
        zko_poly wit, swit;
        divide_by_linear(&wit, p.c, plen, &gamma, F);
        g1j_t w, t2;
        commit_range(&w, C, ck->powers, wit.c, wit.len);
        divide_by_linear(&swit, r3[0].p.c, r3[0].p.len, &gamma, F);
        for (size_t i = 0; i < swit.len; i++) fr_mul(&swit.c[i], &swit.c[i], &chp[1], F);
        size_t off = (ck->bounds[1] - (k - 2));
        commit_range(&t2, C, ck->shifted_powers + off, swit.c, swit.len); g1j_add(&w, &w, &t2, C);
        g1j_to_affine(&pf->w_gamma, &w, C);
        poly_free(&p); poly_free(&wit); poly_free(&swit);
    }
    /* ---- assemble */
    for (int i = 0; i < 4; i++) { pf->comms[i] = r1[i].comm; pf->polys[i] = r1[i].p; }
    for (int i = 0; i < 3; i++) { pf->comms[4 + i] = r2[i].comm; pf->polys[4 + i] = r2[i].p; }
    for (int i = 0; i < 2; i++) { pf->comms[7 + i] = r3[i].comm; pf->polys[7 + i] = r3[i].p; }
    pf->alpha = alpha; pf->eta_a = eta_a; pf->eta_b = eta_b; pf->eta_c = eta_c; pf->beta = beta; pf->gamma = gamma; pf->opening_challenge = ch;
    free(za); free(zb); free(x_evals); free(r_alpha_evals); free(small);
    poly_free(&x_poly); poly_free(&r_alpha_poly); poly_free(&z_poly); poly_free(&f_poly);
    TM("evaluations + openings");
    return pf;
}
void zko_proof_free(zko_proof *p) { if (!p) return; for (int i = 0; i < 9; i++) free(p->polys[i].c); free(p); }
