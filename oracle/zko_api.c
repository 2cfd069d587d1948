/* oracle/zko_api.c -- TEST INFRASTRUCTURE (CPU oracle): flat C entry points for ctypes (tests/, smoke(), bench cpu_baseline).
 * Field elements cross this API as raw little-endian MONTGOMERY limbs (32 B Fr, 48 B Fq), the same byte layout the
 * product's device buffers use; G1 affine points as x||y (96 B) + separate infinity flags where needed. */
#include "zko_marlin.h"
#include <stdlib.h>
#include <time.h>
#include <omp.h>

int zko_api_num_threads(void) { return omp_get_max_threads(); }
void zko_api_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* ---- fields / curves ---- */
void zko_api_fr_mul(int id, const fr_t *a, const fr_t *b, fr_t *out) { fr_mul(out, a, b, zko_fr_by_id(id)); }
void zko_api_fr_add(int id, const fr_t *a, const fr_t *b, fr_t *out) { fr_add(out, a, b, zko_fr_by_id(id)); }
void zko_api_fr_sub(int id, const fr_t *a, const fr_t *b, fr_t *out) { fr_sub(out, a, b, zko_fr_by_id(id)); }
void zko_api_fr_inv(int id, const fr_t *a, fr_t *out) { fr_inv(out, a, zko_fr_by_id(id)); }
void zko_api_fq_mul(int id, const fq_t *a, const fq_t *b, fq_t *out) { fq_mul(out, a, b, zko_curve_by_id(id)->fq); }
void zko_api_fq_inv(int id, const fq_t *a, fq_t *out) { fq_inv(out, a, zko_curve_by_id(id)->fq); }
static void load_pt(g1a_t *p, const uint8_t *xy) { memcpy(p->x.l, xy, 48); memcpy(p->y.l, xy + 48, 48); p->inf = 0; }
static int store_pt(uint8_t *xy, const g1a_t *p) { memcpy(xy, p->x.l, 48); memcpy(xy + 48, p->y.l, 48); return p->inf; }
void zko_api_g1_generator(int id, uint8_t *xy) { store_pt(xy, &zko_curve_by_id(id)->gen); }
int zko_api_g1_on_curve(int id, const uint8_t *xy) { g1a_t p; load_pt(&p, xy); return g1a_on_curve(&p, zko_curve_by_id(id)); }
/* out = k*P (k Montgomery Fr); returns infinity flag */
int zko_api_g1_mul(int id, const uint8_t *xy, const fr_t *k, uint8_t *out) {
    const zko_curve *C = zko_curve_by_id(id);
    g1a_t p, r; load_pt(&p, xy);
    g1j_t j, q; g1j_from_affine(&j, &p, C); g1j_mul_fr(&q, &j, k, C); g1j_to_affine(&r, &q, C);
    return store_pt(out, &r);
}
int zko_api_g1_add(int id, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    const zko_curve *C = zko_curve_by_id(id);
    g1a_t p, q, r; load_pt(&p, a); load_pt(&q, b);
    g1j_t j; g1j_from_affine(&j, &p, C); g1j_madd(&j, &j, &q, C); g1j_to_affine(&r, &j, C);
    return store_pt(out, &r);
}
/* bases: n x 96 B affine Montgomery; scalars: n x 32 B Montgomery */
int zko_api_msm(int id, const uint8_t *bases, const fr_t *scalars, size_t n, uint8_t *out) {
    const zko_curve *C = zko_curve_by_id(id);
    g1a_t *b = malloc((n ? n : 1) * sizeof(g1a_t));
    for (size_t i = 0; i < n; i++) load_pt(&b[i], bases + 96 * i);
    g1j_t r; zko_msm(&r, b, scalars, n, C);
    g1a_t a; g1j_to_affine(&a, &r, C);
    free(b);
    return store_pt(out, &a);
}
/* out[i] = scalars[i] * G  (n x 96 B) */
void zko_api_fixed_base(int id, const fr_t *scalars, size_t n, uint8_t *out) {
    const zko_curve *C = zko_curve_by_id(id);
    g1a_t *o = malloc((n ? n : 1) * sizeof(g1a_t));
    zko_fixed_base_batch(o, &C->gen, scalars, n, C);
    for (size_t i = 0; i < n; i++) store_pt(out + 96 * i, &o[i]);
    free(o);
}
/* ---- NTT: data = n Montgomery Fr, transformed in place. kind: 0 fft, 1 ifft, 2 coset fft, 3 coset ifft */
int zko_api_ntt(int id, fr_t *data, size_t n, int kind) {
    zko_domain D;
    if (zko_domain_init(&D, n, zko_fr_by_id(id)) || D.size != n) return -1;
    if (kind == 0) zko_fft(&D, data); else if (kind == 1) zko_ifft(&D, data); else if (kind == 2) zko_coset_fft(&D, data); else zko_coset_ifft(&D, data);
    return 0;
}
void zko_api_domain_gen(int id, size_t n, fr_t *out) { zko_domain D; zko_domain_init(&D, n, zko_fr_by_id(id)); *out = D.gen; }
/* ---- rng / hash ---- */
void zko_api_chacha_words(const uint8_t seed[32], int rounds, size_t n, uint32_t *out) { zko_chacha r; zko_chacha_init(&r, seed, rounds); for (size_t i = 0; i < n; i++) out[i] = zko_chacha_u32(&r); }
void zko_api_fr_rand_stream(int id, const uint8_t seed[32], int rounds, size_t n, fr_t *out) { zko_chacha r; zko_chacha_init(&r, seed, rounds); for (size_t i = 0; i < n; i++) zko_fr_rand(&out[i], &r, zko_fr_by_id(id)); }

/* ---- circuits ---- */
zko_cs *zko_api_synth_aes(int field_id, const uint8_t *msg, size_t len, const uint8_t *key, uint8_t *ct_out) {
    zko_cs *cs = zko_cs_new(field_id);
    if (zko_synth_aes(cs, msg, len, key, ct_out)) { zko_cs_free(cs); return NULL; }
    return cs;
}
zko_cs *zko_api_synth_ops(int field_id, int which, uint32_t x, uint32_t y, uint32_t *out) {
    zko_cs *cs = zko_cs_new(field_id);
    *out = which == 0 ? zko_synth_ops_xor(cs, x, y) : zko_synth_ops_add(cs, x, y);
    return cs;
}
void zko_api_cs_counts(const zko_cs *cs, uint64_t out[6]) {
    out[0] = cs->A.n; out[1] = cs->num_instance; out[2] = cs->num_witness; out[3] = cs->A.nnz; out[4] = cs->B.nnz; out[5] = cs->C.nnz;
}
void zko_api_cs_assignment(const zko_cs *cs, uint8_t *instance, uint8_t *witness) {
    if (instance) memcpy(instance, cs->instance_val, cs->num_instance);
    if (witness) memcpy(witness, cs->witness_val, cs->num_witness);
}
/* CSR of matrix `which` (0 A, 1 B, 2 C): rowptr[n+1] (u64), col[nnz] (final column index), coeff[nnz] (i64) */
/* fault injection for tests: overwrite witness variable `idx` (corrupted-witness test) */
int zko_api_cs_set_witness(zko_cs *cs, size_t idx, int val) { if (idx >= cs->num_witness) return -1; cs->witness_val[idx] = (uint8_t)(val != 0); return 0; }
void zko_api_cs_matrix(const zko_cs *cs, int which, uint64_t *rowptr, uint32_t *col, int64_t *coeff) {
    const zko_mat *M = which == 0 ? &cs->A : which == 1 ? &cs->B : &cs->C;
    for (size_t i = 0; i <= M->n; i++) rowptr[i] = M->rowptr[i];
    for (size_t i = 0; i < M->nnz; i++) { col[i] = zko_var_col(cs, M->var[i]); coeff[i] = M->coeff[i]; }
}

/* ---- marlin ---- */
zko_index *zko_api_index(zko_cs *cs, size_t nc, size_t nv, size_t nnz) { return zko_marlin_index(cs, nc, nv, nnz); }
void zko_api_index_info(const zko_index *ix, uint64_t out[8]) {
    out[0] = ix->num_variables; out[1] = ix->num_constraints; out[2] = ix->num_non_zero; out[3] = ix->num_instance;
    out[4] = ix->H.size; out[5] = ix->K.size; out[6] = ix->ck.max_degree; out[7] = ix->ck.supported_degree;
}
/* which: 0 row,1 col,2 val_a,3 val_b,4 val_c,5 row_col ; form: 0 evals on K, 1 coefficients */
void zko_api_index_poly(const zko_index *ix, int which, int form, fr_t *out) {
    const fr_t *e[6] = {ix->row_evals, ix->col_evals, ix->val_a_evals, ix->val_b_evals, ix->val_c_evals, ix->row_col_evals};
    const zko_poly *p[6] = {&ix->row, &ix->col, &ix->val_a, &ix->val_b, &ix->val_c, &ix->row_col};
    memcpy(out, form ? p[which]->c : e[which], ix->K.size * sizeof(fr_t));
}
void zko_api_index_comms(const zko_index *ix, uint8_t *out /* 6 x 96 */) { for (int i = 0; i < 6; i++) store_pt(out + 96 * i, &ix->index_comms[i]); }
/* trapdoor beta and the two G1 base points of the SRS (96 B affine each) */
void zko_api_srs_info(const zko_index *ix, fr_t *beta, uint8_t *g_xy, uint8_t *gamma_g_xy) { *beta = ix->ck.beta; store_pt(g_xy, &ix->ck.g); store_pt(gamma_g_xy, &ix->ck.gamma_g); }
/* powers_of_g[from .. from+count) as affine; only ranges the committer key holds */
int zko_api_srs_powers(const zko_index *ix, size_t from, size_t count, uint8_t *out) {
    const zko_ck *ck = &ix->ck;
    for (size_t i = 0; i < count; i++) {
        size_t d = from + i;
        const g1a_t *p;
        if (d <= ck->supported_degree) p = &ck->powers[d];
        else if (d >= ck->lowest_shift && d <= ck->max_degree) p = &ck->shifted_powers[d - ck->lowest_shift];
        else return -1;
        store_pt(out + 96 * i, p);
    }
    return 0;
}
uint64_t zko_api_pk_serialize(const zko_index *ix, const char *path) { return zko_pk_serialize_ark_to_file(ix, path); }
uint64_t zko_api_pk_serialize_mode(const zko_index *ix, const char *path, int uncompressed) { return zko_pk_serialize_ark_to_file_mode(ix, path, uncompressed); }
zko_proof *zko_api_prove(const zko_index *ix, zko_cs *cs, const uint8_t *zk_seed) { return zko_marlin_prove(ix, cs, zk_seed); }
size_t zko_api_proof_bytes(const zko_proof *p, int curve_id, uint8_t *out, size_t cap) { return zko_proof_serialize(p, zko_curve_by_id(curve_id), out, cap); }
size_t zko_api_proof_poly_len(const zko_proof *p, int i) { return p->polys[i].len; }
void zko_api_proof_poly(const zko_proof *p, int i, fr_t *out) { memcpy(out, p->polys[i].c, p->polys[i].len * sizeof(fr_t)); }
/* challenges: alpha eta_a eta_b eta_c beta gamma opening_challenge, then the 4 evaluations */
void zko_api_proof_scalars(const zko_proof *p, fr_t out[11]) {
    out[0] = p->alpha; out[1] = p->eta_a; out[2] = p->eta_b; out[3] = p->eta_c; out[4] = p->beta; out[5] = p->gamma; out[6] = p->opening_challenge;
    for (int i = 0; i < 4; i++) out[7 + i] = p->evals[i];
}
double zko_api_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
