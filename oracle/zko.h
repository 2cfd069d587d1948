/* oracle/zko.h -- TEST INFRASTRUCTURE: CPU oracle ("zko") for the zk-aes hot path.
 *
 * A plain-C restatement of the reference's AES-128 Marlin proving path, used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / baseline.  The product
 * (aes_zero_knowledge_proof_circuit_amd/csrc) never includes, links or calls anything in here.
 *
 * PARITY STATUS: byte-level AES values are pinned by the reference's FIPS-197 vectors
 * (tests/integration_tests.rs:52-64,67-276; src/aes_circuit.rs:722-845).  Everything at and below
 * the gadget / Marlin layer restates crates that are NOT under /root/reference (ark-r1cs-std 0.3.1,
 * ark-relations 0.3.0, ark-marlin 0.3.0 fork bde002de, ark-poly-commit 0.3.0, ark-ec/ark-ff 0.3.0,
 * blake2 0.9.2, rand_chacha 0.3.1 -- Cargo.lock:76-356,414,1303,1496) from their published
 * algorithms: "parity unpinned" there.
 */
#ifndef ZKO_H
#define ZKO_H
#include <stddef.h>
#include <stdint.h>

#define FP_N 4
#define FP_(x) fr_##x
#include "zko_fp_tmpl.h"
#undef FP_N
#undef FP_
#define FP_N 6
#define FP_(x) fq_##x
#include "zko_fp_tmpl.h"
#undef FP_N
#undef FP_

/* ---- curves ---- */
typedef struct { fq_t x, y; int inf; } g1a_t;          /* affine, Montgomery coords */
typedef struct { fq_t x, y, z; } g1j_t;                /* Jacobian; z==0 <=> infinity */
typedef struct {
    const fr_params *fr;
    const fq_params *fq;
    fq_t b;
    g1a_t gen;
    int id; /* 377 / 381 */
} zko_curve;

extern fr_params ZKO_FR377, ZKO_FR381;
extern fq_params ZKO_FQ377, ZKO_FQ381;
extern zko_curve ZKO_BLS377, ZKO_BLS381;
const zko_curve *zko_curve_by_id(int id);
const fr_params *zko_fr_by_id(int id);

void g1j_set_inf(g1j_t *r);
int g1j_is_inf(const g1j_t *a);
void g1j_from_affine(g1j_t *r, const g1a_t *a, const zko_curve *C);
void g1j_dbl(g1j_t *r, const g1j_t *a, const zko_curve *C);
void g1j_add(g1j_t *r, const g1j_t *a, const g1j_t *b, const zko_curve *C);
void g1j_madd(g1j_t *r, const g1j_t *a, const g1a_t *b, const zko_curve *C);
void g1j_neg(g1j_t *r, const g1j_t *a, const zko_curve *C);
void g1j_to_affine(g1a_t *r, const g1j_t *a, const zko_curve *C);
void g1j_batch_to_affine(g1a_t *r, const g1j_t *a, size_t n, const zko_curve *C);
/* scalar given as canonical little-endian limbs */
void g1j_mul_raw(g1j_t *r, const g1j_t *a, const uint64_t *k, int klimbs, const zko_curve *C);
void g1j_mul_fr(g1j_t *r, const g1j_t *a, const fr_t *k, const zko_curve *C);
int g1a_on_curve(const g1a_t *a, const zko_curve *C);
int g1a_eq(const g1a_t *a, const g1a_t *b);

/* Pippenger, restating ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul.
 * scalars: Montgomery-form Fr (converted to canonical inside). */
void zko_msm(g1j_t *out, const g1a_t *bases, const fr_t *scalars, size_t n, const zko_curve *C);
/* fixed-base batch: out[i] = scalars[i] * base (affine outputs) */
void zko_fixed_base_batch(g1a_t *out, const g1a_t *base, const fr_t *scalars, size_t n, const zko_curve *C);

/* ---- radix-2 domains (ark-poly 0.3.0 Radix2EvaluationDomain) ---- */
typedef struct {
    const fr_params *F;
    uint64_t size;
    int log_size;
    fr_t size_as_fe, size_inv, gen, gen_inv, coset_gen; /* coset_gen = F multiplicative generator */
} zko_domain;
int zko_domain_init(zko_domain *D, uint64_t min_size, const fr_params *F);
void zko_fft(const zko_domain *D, fr_t *a);                 /* in place, a has D->size entries */
void zko_ifft(const zko_domain *D, fr_t *a);
void zko_coset_fft(const zko_domain *D, fr_t *a);
void zko_coset_ifft(const zko_domain *D, fr_t *a);
void zko_domain_element(fr_t *r, const zko_domain *D, uint64_t i);
void zko_domain_eval_vanishing(fr_t *r, const zko_domain *D, const fr_t *x);
size_t zko_reindex_by_subdomain(const zko_domain *self, const zko_domain *other, size_t index);
void zko_poly_eval(fr_t *r, const fr_t *coeffs, size_t n, const fr_t *x, const fr_params *F);

/* ---- Fiat-Shamir / rngs ---- */
void zko_blake2s(uint8_t out[32], const uint8_t *in, size_t len);
typedef struct {
    uint32_t key[8];
    uint64_t counter;   /* 64-byte block counter of the NEXT refill's first block */
    uint32_t buf[64];   /* 4 blocks, as rand_chacha's BlockRng buffer */
    int idx;            /* next unread word */
    int rounds;         /* 20 (ChaChaRng) or 12 (StdRng) */
} zko_chacha;
void zko_chacha_init(zko_chacha *r, const uint8_t seed[32], int rounds);
uint32_t zko_chacha_u32(zko_chacha *r);
uint64_t zko_chacha_u64(zko_chacha *r);
void zko_fr_rand(fr_t *out, zko_chacha *rng, const fr_params *F);   /* ark-ff UniformRand */
void zko_fq_rand(fq_t *out, zko_chacha *rng, const fq_params *Q);
int zko_fq_sqrt(fq_t *out, const fq_t *a, const fq_params *Q);
void zko_g1_rand(g1a_t *out, zko_chacha *rng, const zko_curve *C);   /* ark-ec GroupProjective::rand (BLS12-377 G1) */

typedef struct { zko_chacha r; uint8_t seed[32]; } zko_fsrng;       /* SimpleHashFiatShamirRng<Blake2s,ChaChaRng> */
void zko_fs_init(zko_fsrng *fs, const uint8_t *bytes, size_t len);
void zko_fs_absorb(zko_fsrng *fs, const uint8_t *bytes, size_t len);

/* ---- byte-level AES (src/aes.rs) ---- */
void zko_aes_add_round_key(uint8_t out[16], const uint8_t in[16], const uint8_t key[16]);
uint8_t zko_aes_substitute_byte(uint8_t b);
void zko_aes_substitute_bytes(uint8_t out[16], const uint8_t in[16]);
void zko_aes_shift_rows(uint8_t out[16], const uint8_t in[16]);
void zko_aes_mix_columns(uint8_t out[16], const uint8_t in[16]);
void zko_aes_derive_keys(uint8_t out[11][16], const uint8_t key[16]);
void zko_aes_encrypt_ecb(uint8_t *out, const uint8_t *msg, size_t len, const uint8_t key[16]);

/* ---- R1CS (ark-relations 0.3.0 ConstraintSystem + ark-r1cs-std 0.3.1 gadgets) ---- */
#define ZKO_WIT_BASE (1u << 30)   /* var ids: 0 = One, 1.. = Instance(i), ZKO_WIT_BASE+j = Witness(j) */
typedef struct { int32_t n; int32_t cap; uint32_t *var; int64_t *coeff; } zko_lc;
typedef struct {
    size_t n, cap; size_t *rowptr;  /* n rows; rowptr has n+1 entries */
    size_t nnz, nzcap; uint32_t *var; int64_t *coeff;
} zko_mat;
typedef struct {
    int field_id;                /* 377 or 381 */
    uint32_t num_instance;       /* includes the constant One at index 0 */
    uint32_t num_witness;
    size_t icap, wcap;
    uint8_t *instance_val;       /* all assignments here are 0/1 (One, booleans, padding) */
    uint8_t *witness_val;
    zko_mat A, B, C;
} zko_cs;

zko_cs *zko_cs_new(int field_id);
void zko_cs_free(zko_cs *cs);
/* the reference's encrypt() synthesis: src/lib.rs:60-114 + 176-293 (message, key, circuit, public inputs) */
int zko_synth_aes(zko_cs *cs, const uint8_t *msg, size_t len, const uint8_t key[16], uint8_t *ct_out);
/* src/ops.rs:8-29 toy gates; return the u32 value */
uint32_t zko_synth_ops_xor(zko_cs *cs, uint32_t x, uint32_t y);
uint32_t zko_synth_ops_add(zko_cs *cs, uint32_t x, uint32_t y);
int zko_cs_is_satisfied(const zko_cs *cs);      /* 1 ok, else -(first failing row+1) */
size_t zko_cs_num_constraints(const zko_cs *cs);
/* ark-marlin padding: pad_input_for_indexer_and_prover + make_matrices_square */
void zko_cs_pad_for_marlin(zko_cs *cs);
static inline uint32_t zko_var_col(const zko_cs *cs, uint32_t var) {
    return var < ZKO_WIT_BASE ? var : cs->num_instance + (var - ZKO_WIT_BASE);
}

#endif
