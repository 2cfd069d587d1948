/* oracle/zko_marlin.h -- TEST INFRASTRUCTURE (CPU oracle): Marlin setup / index / prove. */
#ifndef ZKO_MARLIN_H
#define ZKO_MARLIN_H
#include "zko.h"

typedef struct { fr_t *c; size_t len; } zko_poly;   /* dense coefficients, low degree first */

typedef struct {
    const zko_curve *C;
    size_t max_degree;                 /* universal SRS degree (AHPForR1CS::max_degree of the setup literals) */
    fr_t beta;                         /* trapdoor; public because the reference seeds it with test_rng (F7) */
    g1a_t g, gamma_g;                  /* KZG10::setup's random base points (drawn after beta from the same rng) */
    size_t supported_degree;           /* committer key: powers_of_g[0..=supported_degree] */
    g1a_t *powers;
    g1a_t gamma_powers[3];             /* powers_of_gamma_g[0..=hiding_bound+1] */
    size_t bounds[2];                  /* enforced degree bounds, sorted: |H|-2, |K|-2 */
    size_t lowest_shift;               /* max_degree - bounds[1] */
    g1a_t *shifted_powers;             /* powers_of_g[lowest_shift..=max_degree] */
} zko_ck;

typedef struct {
    zko_cs *cs;                        /* padded constraint system (matrices) */
    size_t num_variables, num_constraints, num_non_zero, num_instance;
    zko_domain H, K, X;
    fr_t *row_evals, *col_evals, *row_col_evals, *val_a_evals, *val_b_evals, *val_c_evals;   /* on K */
    zko_poly row, col, row_col, val_a, val_b, val_c;                                        /* interpolated */
    g1a_t index_comms[6];              /* row, col, a_val, b_val, c_val, row_col (INDEXER_POLYNOMIALS order) */
    zko_ck ck;
    /* transposed copies of A,B,C for calculate_t are not kept: the oracle follows the row-major loop */
} zko_index;

typedef struct { g1a_t comm; int has_shifted; g1a_t shifted; } zko_commitment;
typedef struct {
    zko_commitment comms[9];           /* w z_a z_b mask_poly | t g_1 h_1 | g_2 h_2 */
    fr_t evals[4];                     /* g_1(beta), g_2(gamma), t(beta), z_b(beta)  (sorted by label) */
    g1a_t w_beta, w_gamma;
    fr_t random_v_beta;                /* pc_proof[0].random_v = Some(..); pc_proof[1].random_v = None */
    /* transcript values kept for tests */
    fr_t alpha, eta_a, eta_b, eta_c, beta, gamma, opening_challenge;
    zko_poly polys[9];                 /* prover oracles in the order above (kept for parity tests) */
} zko_proof;

size_t zko_ahp_max_degree(size_t num_constraints, size_t num_variables, size_t num_non_zero, const fr_params *F);
/* universal_setup(literals) + index_from_constraint_system(cs); takes ownership of a synthesized, UNPADDED cs */
zko_index *zko_marlin_index(zko_cs *cs, size_t srs_nc, size_t srs_nv, size_t srs_nnz);
void zko_index_free(zko_index *ix);
/* prove_from_constraint_system: cs = freshly synthesized (unpadded) system with the real assignment; padded in place.
 * zk_seed = 32-byte StdRng (ChaCha12) seed (NULL => ark_std::test_rng seed, as simpleworks::marlin::generate_rand). */
zko_proof *zko_marlin_prove(const zko_index *ix, zko_cs *cs, const uint8_t *zk_seed);
void zko_proof_free(zko_proof *p);
/* ark-serialize 0.3 compressed layout of ark_marlin::Proof; returns length, writes into out (cap bytes) */
size_t zko_proof_serialize(const zko_proof *p, const zko_curve *C, uint8_t *out, size_t cap);
/* ark-serialize image of ark_marlin::IndexProverKey minus its first field (index_vk), streamed to `path`; returns the bytes written (0 = cannot open) */
uint64_t zko_pk_serialize_ark_to_file(const zko_index *ix, const char *path);
/* uncompressed != 0: the serialize_uncompressed image (96-byte G1 points) that deserialize_unchecked reads */
uint64_t zko_pk_serialize_ark_to_file_mode(const zko_index *ix, const char *path, int uncompressed);
/* IndexVerifierKey pieces the product verifier needs are serialized by the product itself; for tests: */
void zko_commit_plain(g1a_t *out, const zko_ck *ck, const fr_t *coeffs, size_t len, size_t power_offset_in_shifted, int use_shifted);
#endif
