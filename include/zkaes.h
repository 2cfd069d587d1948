/* include/zkaes.h -- C ABI of libzkaes, the MI355X-native drop-in for the proving hot path of
 * lambdaclass/AES_zero_knowledge_proof_circuit (crate `zk-aes`).
 *
 * The reference has no FFI / plugin interface; its seam is three Rust functions in src/lib.rs (and the crate
 * forbids `unsafe`, src/lib.rs:2, so the binding lives in a sibling -sys crate: see INTEGRATION.md).  Each entry
 * point below names the reference item it replaces.  Conventions:
 *   - return value 0 = Ok, non-zero = Err; the message of the last error on the calling thread is
 *     zkaes_last_error() (mirrors anyhow::Result, src/lib.rs:45);
 *   - keys are opaque handles that own device-resident data (SRS powers, index polynomials, circuit tables,
 *     workspace); a handle is bound to the GPU that was current when it was created;
 *   - proofs cross the boundary as bytes in the ark-serialize 0.3 compressed layout of ark_marlin::Proof
 *     (what simpleworks' (de)serialize_proof reads/writes, re-exported at src/lib.rs:52);
 *   - byte buffers returned through `uint8_t**` are owned by the library: release with zkaes_bytes_free.
 *   - keys for a plaintext of 16 bytes or more use fixed-base window tables of the universal SRS (13 copies of 192-byte records of powers_of_g[0 ..= max_degree]: 31.4 GB for the
 *     reference's literals, held ONCE per process and device and shared by every key -- zkaes_pk_srs_info; skipped when the device is short of memory or with ZKAES_KEY_NO_TABLES): multi-proof calls (zkaes_encrypt_chunked / _batch) run their
 *     large MSMs (>= 100 k points) through them (13 balanced windows of 19-20 bits over ONE bucket set instead of 15 windows with their own buckets), and so does a
 *     lone zkaes_encrypt call, which additionally runs the independent commitments of each round on four MSM lanes (streams + host threads) side by side.  The
 *     SRS points are stored on BLS12-377's twisted Edwards model (7-product bucket additions): the prover's MSMs assume prime-order-subgroup bases, as KZG's are;
 *   - host threads of a multi-proof call wait for the GPU by polling with short sleeps (a fraction of a core per prover context); a lone zkaes_encrypt call
 *     spins, for latency.  ZKAES_WAIT=spin|sleep forces one policy;
 *   - when the library is loaded it exports GPU_MAX_HW_QUEUES=16 unless the variable is already set (one hardware queue per prover context; the ROCm default of 4
 *     lets the contexts' kernels queue behind each other).  It is read at the first HIP call of the process: export it yourself if HIP is initialised earlier.
 *     ZKAES_KEEP_ENV=1 makes the library leave the environment alone (for hosts whose other threads read it while libraries load);
 *   - the library needs a HIP device (gfx950) for key synthesis and proving and FAILS (non-zero + message) when
 *     none is present -- there is no CPU fallback; zkaes_verify_* runs on the host, as in the reference.
 */
#ifndef ZKAES_H
#define ZKAES_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkaes_pk zkaes_pk;   /* simpleworks::marlin::ProvingKey   (src/lib.rs:55) */
typedef struct zkaes_vk zkaes_vk;   /* simpleworks::marlin::VerifyingKey (src/lib.rs:55) */

/* thread-local message of the last failing call ("" if none) */
const char *zkaes_last_error(void);
void zkaes_bytes_free(uint8_t *p);
void zkaes_pk_free(zkaes_pk *pk);
void zkaes_vk_free(zkaes_vk *vk);
/* number of visible HIP devices (0 on a host without a GPU) */
int zkaes_device_count(void);
/* select the HIP device for the CALLING THREAD (hipSetDevice semantics): key handles synthesized afterwards live on it, and the kernel-level
 * entry points below run on it.  A proving key remembers its device: zkaes_encrypt* / zkaes_prove_ops re-select it on whatever thread calls them.
 * Intended deployment: one process per GPU (the NTT twiddle cache is per process). */
int zkaes_set_device(int ordinal);

/* ---- the reference's public API ------------------------------------------------------------------------------- */
/* replaces `pub fn synthesize_keys(plaintext_length: usize) -> Result<(ProvingKey, VerifyingKey)>` (src/lib.rs:138-174):
 * universal SRS for the literals (866_944, 513, 4_062_064) of src/lib.rs:141 + index of the AES circuit for a message of
 * `plaintext_length` bytes (multiple of 16). */
int zkaes_synthesize_keys(size_t plaintext_length, zkaes_pk **pk, zkaes_vk **vk);
/* replaces `pub fn encrypt(message: &[u8], secret_key: &[u8; 16], proving_key: ProvingKey) -> Result<MarlinProof>`
 * (src/lib.rs:60-114).  The proving key is borrowed, not consumed (callers of the reference clone it per call,
 * benches/benchmark_encrypt.rs:46).  Prover randomness = generate_rand() (fixed ark_std::test_rng seed), as src/lib.rs:65. */
int zkaes_encrypt(const uint8_t *message, size_t message_len, const uint8_t secret_key[16], const zkaes_pk *pk, uint8_t **proof, size_t *proof_len);
/* replaces `pub fn verify_encryption(verifying_key: VerifyingKey, proof: &MarlinProof, ciphertext: &[u8]) -> Result<bool>`
 * (src/lib.rs:116-136): ciphertext bytes -> 8 LSB-first field elements each (src/helpers/mod.rs:84-93) -> Marlin verify.
 * A wrong ciphertext is Ok(false): returns 0 with *accepted = 0 (tests/integration_tests.rs:336). */
int zkaes_verify_encryption(const zkaes_vk *vk, const uint8_t *proof, size_t proof_len, const uint8_t *ciphertext, size_t ciphertext_len, int *accepted);
/* replaces the re-export `deserialize_proof` (src/lib.rs:52) as a validity check + canonical re-serialization */
int zkaes_proof_roundtrip(const uint8_t *proof, size_t proof_len, uint8_t **out, size_t *out_len);

/* ---- extensions ------------------------------------------------------------------------------------------------ */
#define ZKAES_CIRCUIT_AES 0      /* src/lib.rs:176-293 */
#define ZKAES_CIRCUIT_OPS_XOR 1  /* src/ops.rs:8-18 (as a BLS12-377 Marlin circuit) */
#define ZKAES_CIRCUIT_OPS_ADD 2  /* src/ops.rs:20-29 */
/* as zkaes_synthesize_keys with an explicit circuit kind and universal-SRS literals (generate_universal_srs arguments) */
int zkaes_synthesize_keys_ex(int circuit_kind, size_t plaintext_length, size_t srs_num_constraints, size_t srs_num_variables, size_t srs_num_non_zero, zkaes_pk **pk,
                             zkaes_vk **vk);
/* the same with option flags.  ZKAES_KEY_NO_TABLES: this key does not use (and, if it is the first key over its SRS, does not build) the fixed-base window tables of the
 * universal SRS (31.4 GB of device memory for the reference's literals, shared by all keys); its multi-proof calls then use 15 per-window-bucket windows instead of
 * 13 table windows, ~9 % fewer proofs per second.  Unknown flag bits are an error. */
#define ZKAES_KEY_NO_TABLES 1u
int zkaes_synthesize_keys_ex2(int circuit_kind, size_t plaintext_length, size_t srs_num_constraints, size_t srs_num_variables, size_t srs_num_non_zero, unsigned flags,
                              zkaes_pk **pk, zkaes_vk **vk);
/* as zkaes_encrypt with an explicit 32-byte StdRng seed for the prover's zero-knowledge randomness (NULL = test_rng seed) */
int zkaes_encrypt_seeded(const uint8_t *message, size_t message_len, const uint8_t secret_key[16], const zkaes_pk *pk, const uint8_t *zk_seed32, uint8_t **proof,
                         size_t *proof_len);
/* chunked proving of a long ECB message: ceil(message_len / chunk_len) independent proofs with one key for chunk_len bytes
 * (the last chunk must be full).  proofs = concatenation, proof_lens[i] = length of proof i (caller array of n_chunks).
 * Zero-knowledge randomness: a FRESH 32-byte seed from the operating system per call (getrandom), proof i drawing from
 * StdRng(Blake2s(seed || i as u64 LE)) -- unlike the reference's encrypt(), whose every call draws from the fixed ark_std::test_rng() seed
 * (src/lib.rs:65), these extensions put hundreds of proofs under one AES key and must not share blinding factors.  The fixed stream for every proof (byte parity with the CPU
 * oracle in tests; not zero-knowledge across proofs) is reachable only explicitly: the *_seeded entry points with a NULL seed. */
int zkaes_encrypt_chunked(const uint8_t *message, size_t message_len, const uint8_t secret_key[16], const zkaes_pk *pk, uint8_t **proofs, size_t *proofs_len,
                          size_t *proof_lens, size_t n_chunks);
/* n independent (message_i, secret_key_i) pairs on one key / one SRS (BASELINE config 5: many small proofs): messages = n x plaintext
 * length bytes, secret_keys = n x 16 bytes.  Up to zkaes_pk_get_contexts(pk) proofs (zkaes_pk_set_contexts; default ZKAES_DEFAULT_CONTEXTS, the configuration bench.py
 * measures) are in flight per call, each on its own HIP stream.  Randomness as zkaes_encrypt_chunked.  This entry point cannot
 * check its buffer lengths: prefer zkaes_encrypt_batch_seeded. */
#define ZKAES_DEFAULT_CONTEXTS 12
int zkaes_encrypt_batch(size_t n, const uint8_t *messages, const uint8_t *secret_keys, const zkaes_pk *pk, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens);
/* Proofs in flight per multi-proof call (zkaes_encrypt_chunked* / _batch*) on THIS key: n = 1..64 prover contexts (own stream + ~0.9 / 4.7 GB of workspace each for the
 * 1- / 6-block key, created on first use), n = 0 restores the process default -- ZKAES_DEFAULT_CONTEXTS, or the ZKAES_CONTEXTS environment variable as it stood when the
 * library first needed it (read once; later changes of the environment are ignored).  Takes effect from the next call; calls already running keep their count. */
int zkaes_pk_set_contexts(zkaes_pk *pk, size_t n);
int zkaes_pk_get_contexts(const zkaes_pk *pk, size_t *n);
/* the chunked / batch calls with explicit buffer lengths (checked: messages_len == n x plaintext length, secret_keys_len == n x 16) and a
 * caller-supplied 32-byte seed.  Proof i draws from StdRng(Blake2s(zk_seed32 || (first_proof_index + i) as u64 LE)): a caller that splits ONE job over
 * several calls or ranks under one seed passes the job-global index of the call's first proof (the *_at variants; the plain ones use 0), so that no
 * two proofs of the job share blinding factors or the mask polynomial.  zk_seed32 == NULL selects the reference's fixed test_rng() stream for every
 * proof (explicit byte-parity mode: differences of hiding commitments across proofs are then unblinded -- tests only). */
int zkaes_encrypt_chunked_seeded(const uint8_t *message, size_t message_len, const uint8_t secret_key[16], const zkaes_pk *pk, const uint8_t *zk_seed32, uint8_t **proofs,
                                 size_t *proofs_len, size_t *proof_lens, size_t n_chunks);
int zkaes_encrypt_chunked_seeded_at(const uint8_t *message, size_t message_len, const uint8_t secret_key[16], const zkaes_pk *pk, const uint8_t *zk_seed32,
                                    uint64_t first_proof_index, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens, size_t n_chunks);
int zkaes_encrypt_batch_seeded(size_t n, const uint8_t *messages, size_t messages_len, const uint8_t *secret_keys, size_t secret_keys_len, const zkaes_pk *pk,
                               const uint8_t *zk_seed32, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens);
int zkaes_encrypt_batch_seeded_at(size_t n, const uint8_t *messages, size_t messages_len, const uint8_t *secret_keys, size_t secret_keys_len, const zkaes_pk *pk,
                                  const uint8_t *zk_seed32, uint64_t first_proof_index, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens);
/* src/ops.rs toy gates proven with Marlin (public input: none) */
int zkaes_prove_ops(const zkaes_pk *pk, uint32_t x, uint32_t y, const uint8_t *zk_seed32, uint8_t **proof, size_t *proof_len);
/* generic verify: public_input_bits = instance assignment without the leading One, one byte (0/1) per variable */
int zkaes_verify(const zkaes_vk *vk, const uint8_t *proof, size_t proof_len, const uint8_t *public_input_bits, size_t n_bits, int *accepted);
/* verifying-key transport, library-private layout v2 ("ZVK2", the unpadded public-input count, then the ark-serialize compressed image below): validated like the ark
 * path (canonical field elements, curve and prime-order-subgroup membership, index-info bounds) -- safe on untrusted bytes */
int zkaes_vk_serialize(const zkaes_vk *vk, uint8_t **out, size_t *out_len);
int zkaes_vk_deserialize(const uint8_t *bytes, size_t len, zkaes_vk **vk);
/* verifying-key transport in the ark-serialize 0.3 compressed layout of ark_marlin::IndexVerifierKey<Fr, MarlinKZG10<Bls12_377, _>>
 * (759 bytes for the AES keys): what `VerifyingKey::serialize` writes / `VerifyingKey::deserialize` reads on the Rust side of
 * src/lib.rs:116 -- SURVEY.md 8f item 1.  Layout restated from the published crates (the reference holds no VK bytes to pin it). */
int zkaes_vk_serialize_ark(const zkaes_vk *vk, uint8_t **out, size_t *out_len);
int zkaes_vk_deserialize_ark(const uint8_t *bytes, size_t len, zkaes_vk **vk);
/* the same key as serialize_uncompressed writes it (G1 as x || y = 96 B, G2 = 192 B, the infinity flag in the top bits of y's last byte; everything else identical;
 * 1,431 bytes): the form `IndexVerifierKey::deserialize_uncompressed` / `::deserialize_unchecked` read -- and the index_vk prefix of the uncompressed proving-key image */
int zkaes_vk_serialize_ark_uncompressed(const zkaes_vk *vk, uint8_t **out, size_t *out_len);

/* host-only: assemble a verifying key from an index made elsewhere over the SAME universal SRS, i.e. KZG10::setup replayed from
 * ark_std::test_rng() (the reference's generate_rand(), src/lib.rs:139): info = {num_variables, num_constraints, num_non_zero,
 * num_instance (padded), num_public_inputs, max_degree, supported_degree}; index_comms = 6 x 96 B affine Montgomery (row col a_val b_val
 * c_val row_col); beta = 32 B Montgomery Fr, must equal the replayed trapdoor (else error).  g, gamma_g, h are the replay's random curve
 * points, beta_h = beta*h, shift powers = beta^(max_degree - bound) * g. */
int zkaes_vk_from_trapdoor(const uint64_t info[7], const uint8_t *index_comms, const uint8_t *beta, zkaes_vk **vk);

/* ---- introspection used by tests / bench ----------------------------------------------------------------------- */
/* counters the reference logs through debug_constraint_system_status (src/helpers/mod.rs:73-81) and the Marlin index sizes:
 * out[0..11] = raw constraints, raw instance, raw witness, nnz A, nnz B, nnz C, padded constraints, padded instance,
 *              padded witness, joint nnz, |H|, |K| */
int zkaes_pk_info(const zkaes_pk *pk, uint64_t out[12]);
/* host-only circuit compilation (no GPU): same counters (joint nnz, |H|, |K| = 0) */
int zkaes_circuit_info(int circuit_kind, size_t plaintext_length, uint64_t out[12]);
/* host-only: CSR of matrix which (0 A, 1 B, 2 C) after padding.  Call with NULL arrays to get sizes. */
int zkaes_circuit_matrix(int circuit_kind, size_t plaintext_length, int which, uint64_t *n_rows, uint64_t *nnz, uint32_t *rowptr, uint32_t *col, int64_t *coeff);
/* copy an intermediate buffer of the LAST proof made with this key to the host ("z", "trace", "z_a_evals", "z_b_evals", "w", "z_a", "z_b",
 * "mask_poly", "t", "g_1", "h_1", "g_2", "h_2", index polys "row", "col", "a_val", "b_val", "c_val", "row_col" (+"_evals")).
 * Field elements are raw little-endian Montgomery limbs (32 B).  Returns the byte count in *len. */
int zkaes_pk_debug_fetch(const zkaes_pk *pk, const char *name, uint8_t **out, size_t *len);
/* per-phase wall times of the last proof: witness, round1, round2, round3, open, total (ms) */
int zkaes_pk_timings(const zkaes_pk *pk, double out[6]);
/* accumulated MSM statistics since the last reset, one entry per k_accumulate LAUNCH (a degree-bounded commitment = one prepared state finished against the plain and
 * the shifted powers = two launches, each booked with its own event pair, points and pairs): bucket-accumulation kernel ms (HIP events), total MSM wall ms, points,
 * launches, (point, window) pairs */
int zkaes_msm_stats(double out[5], int reset);
/* The library's ACTUAL op lists for one proof on this key ("recompute W + S + T + M from the actual op lists and print the lists"): proves `message` once with
 * the op recorder open -- throughput_path != 0: as a multi-proof call runs a proof (window tables, one lane), 0: as a lone zkaes_encrypt does -- and returns JSON (release
 * with zkaes_bytes_free): {"h","k","x","variables","constraints","nnz":[A,B,C],"blocks","path","ntt":[[points, transforms sharing the launch],...],
 * "msm":[[points,"buckets"|"second_bases"|"class_sum"],...]}.  The recorder is process-global: call it with no other proof in flight. */
int zkaes_pk_op_lists(const zkaes_pk *pk, const uint8_t *message, size_t message_len, const uint8_t secret_key[16], int throughput_path, uint8_t **json, size_t *json_len);
/* free / total device memory of the calling thread's current device (hipMemGetInfo) */
int zkaes_mem_info(uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- kernel-level entry points (parity tests + roofline measurement) -------------------------------------------- */
/* field_id: 377 or 381 (BLS12-377 / BLS12-381 scalar field).  data: n x 32 B Montgomery limbs, host memory, transformed in place.
 * n must be a power of two.  inverse != 0 -> IFFT (scaled by 1/n).  Natural order in and out. */
int zkaes_ntt(int field_id, uint8_t *data, size_t n, int inverse);
/* the same transform on the coset g D of the size-n domain D, g = W^coset_c with W the primitive 2^lg_big-th root (2^lg_big > n, 0 < coset_c < 2^lg_big / n): forward =
 * the values p(g w^i) of the coefficient vector, inverse = the coefficients from those values.  The prover's second round runs on such cosets of H inside the 4|H| domain. */
int zkaes_ntt_coset(int field_id, uint8_t *data, size_t n, int inverse, int coset_c, int lg_big);
/* `count` (1..12) transforms of one shape in shared launches, as the prover's rounds 1 and 2 issue them: data = count x n x 32 B, transformed in place; coset_c[i] = 0 for a
 * plain transform, > 0 for the coset W^coset_c[i] D (coset_c may be NULL: all plain; lg_big as in zkaes_ntt_coset, ignored when no job is a coset transform). */
int zkaes_ntt_batch(int field_id, uint8_t *data, size_t n, int count, int inverse, const int *coset_c, int lg_big);
/* curve_id 377 / 381.  bases: n x 96 B affine (x||y Montgomery), scalars: n x 32 B Montgomery Fr; out_xy 96 B, *out_inf = 1 if infinity */
int zkaes_msm(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n, uint8_t *out_xy, int *out_inf);
/* host-side sum of n affine points (n x 96 B, inf[i] != 0 marks the point at infinity; inf may be NULL): the local EC add that follows the
 * all-gather when ONE MSM is sharded by point range over ranks (SURVEY.md 8e "inside one proof"; the upstream analogue is the final fold of the
 * per-window sums in ark-ec's VariableBaseMSM::multi_scalar_mul).  Runs on the host: a few hundred field products. */
int zkaes_g1_sum(int curve_id, const uint8_t *points_xy, const int *inf, size_t n, uint8_t *out_xy, int *out_inf);
/* ---- ONE MSM sharded by point range over the GPUs of a node with a device-resident exchange (SURVEY.md 8e "inside one proof"; north_star's "single RCCL
 * all-reduce of bucket partials over xGMI" -- EC addition is not an RCCL reduction operator, hence all-gather + local fold).  Every rank:
 *   1. zkaes_msm_sharded_plan(curve, n_total, ...)           -> window plan of the WHOLE MSM + bytes of one rank's payload (n_windows x 192 B)
 *   2. zkaes_msm_window_sums_dev(curve, bases, scalars of ITS slice, n_local, n_total, dev_out, bytes): Pippenger over the slice, the n_windows
 *      XYZZ window sums are left IN DEVICE MEMORY at dev_out (e.g. row `rank` of a [world, bytes] torch.uint8 CUDA tensor)
 *   3. one all-gather of those rows over RCCL (HBM to HBM over xGMI; torch.distributed.all_gather_into_tensor)
 *   4. zkaes_msm_fold_window_sums_dev(curve, dev_in = the gathered [world, bytes] block, world, n_total, out): per-window sum over ranks on the
 *      device, Horner over the windows -> the MSM result.  aes_zero_knowledge_proof_circuit_amd/sharding.py msm_sharded_device is this sequence. */
int zkaes_msm_sharded_plan(int curve_id, size_t n_total, int *window_bits, int *n_windows, size_t *bytes_per_rank);
int zkaes_msm_window_sums_dev(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n_local, size_t n_total, void *dev_out, size_t dev_out_bytes);
int zkaes_msm_fold_window_sums_dev(int curve_id, const void *dev_in, int world, size_t n_total, uint8_t *out_xy, int *out_inf);
/* The same sharding on the PROVER'S OWN path, for commitments over a key's SRS (src/lib.rs:111 -> KZG10::commit): the key's powers_of_g on the curve's twisted Edwards model,
 * fixed-base window tables, ONE bucket set -- so a rank's share is ONE partial sum instead of one per window.  Every rank:
 *   1. zkaes_pk_msm_partial_dev(pk, scalars of ITS slice (n_local x 32 B Montgomery Fr), n_local, offset of the slice in powers_of_g, dev_out, 192): the slice's sum as one XYZZ
 *      point (192 B) left IN DEVICE MEMORY (row `rank` of a [world, 192] CUDA tensor); an empty share is the point at infinity
 *   2. one all-gather of the rows over RCCL
 *   3. zkaes_msm_fold_partials_dev(377, dev_in = the gathered block, world, out): sum over ranks on the device -> the commitment (affine).
 * Needs a key with tables (zkaes_pk_tables_built).  aes_zero_knowledge_proof_circuit_amd/sharding.py msm_sharded_srs_device is this sequence. */
int zkaes_pk_msm_partial_dev(const zkaes_pk *pk, const uint8_t *scalars, size_t n_local, size_t offset, void *dev_out, size_t dev_out_bytes);
int zkaes_msm_fold_partials_dev(int curve_id, const void *dev_in, int world, uint8_t *out_xy, int *out_inf);
/* The proving key as the reference would hold it (src/lib.rs:138-174 returns simpleworks' ProvingKey = ark_marlin::IndexProverKey BY VALUE): its ark-serialize 0.3 compressed
 * image -- index_vk, index_comm_rands, index (info, the padded matrices A, B, C, the six index polynomials with their evaluations on K), committer key (powers, shifted
 * powers, powers_of_gamma_g, degree bounds) -- streamed to `path` (0.65 GB for a 16-byte key).  A Rust caller reads it with
 * IndexProverKey::deserialize_unchecked(BufReader::new(File::open(path)?)) and can run the reference's own CPU encrypt() on a key that was synthesized on the GPU in seconds. */
int zkaes_pk_serialize_ark_to_file(const zkaes_pk *pk, const char *path, uint64_t *bytes_written);
/* the same with the point encoding chosen: uncompressed = 0 as above (48-byte points; `IndexProverKey::deserialize` takes a square root and a subgroup check per point --
 * tens of minutes for 16 M powers); uncompressed != 0: serialize_uncompressed's image (96-byte points, index_vk with 192-byte G2 elements; 1.25 GB for a 16-byte key),
 * which `IndexProverKey::deserialize_unchecked` -- ark-serialize 0.3: deserialize_unchecked defaults to the UNCOMPRESSED layout -- reads without any per-point work.
 * This is the image integration/rust_verify_harness/src/bin/encrypt_with_gpu_key.rs loads.  A failed write removes the partial file. */
int zkaes_pk_serialize_ark_to_file_ex(const zkaes_pk *pk, const char *path, int uncompressed, uint64_t *bytes_written);
/* *built = 1 when the key holds the fixed-base window tables of its SRS (they are skipped under ZKAES_KEY_NO_TABLES, or when device memory would not also hold the
 * default number of prover contexts); *table_bytes (may be NULL) = their size in device memory */
int zkaes_pk_tables_built(const zkaes_pk *pk, int *built, uint64_t *table_bytes);
/* The universal SRS behind the key.  src/lib.rs:139-141 builds ONE generate_universal_srs(866_944, 513, 4_062_064) for every circuit size; so does the library: per process,
 * device and SRS literals there is one array powers_of_g[0 ..= max_degree] (+ its 12 window-table copies), built by the first key that needs it and shared by all later
 * ones -- a key's plain and shifted powers are index ranges of it -- and released with the last key.  out = {max_degree, points per copy, copies (1 = no tables), device
 * bytes, keys sharing it now, bytes of the Lagrange-basis points (shared per |H|, |X|)}; secs (may be NULL) = {seconds THIS key's synthesis spent building the SRS
 * (~0 when it was shared), seconds of the whole synthesis}. */
int zkaes_pk_srs_info(const zkaes_pk *pk, uint64_t out[6], double secs[2]);
/* hold != 0: the library keeps every universal SRS (and Lagrange-basis SRS) it has built, or builds from now on, resident after the last key over it is freed -- for callers that
 * create and free keys in turn (one key per request size) and would otherwise rebuild 31.4 GB of window tables each time; hold == 0 releases them again (device memory goes
 * back once no key uses them).  Default: not held. */
/* the PROCESS default of a key's prover contexts (what ZKAES_CONTEXTS sets at start-up; 0 = back to that; at most 64).  Besides being what keys without their own
 * zkaes_pk_set_contexts use, it sizes the device memory key synthesis leaves free beside the window tables: lower it BEFORE synthesizing a key whose contexts are several times
 * larger than the reference sizes' (e.g. 28 blocks per proof over a universal SRS four times the reference's literal: ~19 GB per context, 126 GB of tables). */
int zkaes_set_default_contexts(size_t n);
int zkaes_srs_hold(int hold);
/* same sum through the precomputed-window layout the prover uses for the SRS (tables 2^(window offset j) P_i built on the fly here; one bucket set), on the
 * Weierstrass model with XYZZ buckets: correct for ANY curve points, both curves. */
int zkaes_msm_table(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n, int window_bits, uint8_t *out_xy, int *out_inf);
/* BLS12-377 only -- exactly the prover's SRS path: the tables on the curve's twisted Edwards model (7-product bucket additions; the law is unified but not complete).
 * PRECONDITION: every base lies in the prime-order subgroup (as every KZG SRS point does).  A base of order 2 or 4 is refused; any other point outside the subgroup
 * may silently give a wrong sum -- use zkaes_msm / zkaes_msm_table for untrusted points. */
int zkaes_msm_table_srs(const uint8_t *bases, const uint8_t *scalars, size_t n, int window_bits, uint8_t *out_xy, int *out_inf);
/* device-resident variant for benchmarking: repeats the MSM `reps` times over device copies, returns ms per MSM of the whole pipeline and of
 * the bucket-accumulation kernel alone */
int zkaes_msm_bench(int curve_id, const uint8_t *bases, const uint8_t *scalars, size_t n, int reps, double *ms_total, double *ms_accumulate);
/* BLS12-377, synthetic device-made bases (powers of a fixed scalar times the generator) and xorshift scalars: window_bits = 0 per-window
 * signed-digit buckets on the Weierstrass model (the generic zkaes_msm path), < 0 the same buckets on the curve's twisted Edwards model (the prover's
 * lone-call SRS path), > 0 the precomputed-table path (Edwards).  out_xy (96 B, may be NULL) receives the sum: all paths must agree. */
int zkaes_msm_bench_synth(size_t n, int window_bits, int reps, double *ms_total, double *ms_accumulate, uint8_t *out_xy);
/* stream-copy probe: copies `bytes` device-to-device `reps` times with a plain 16 B/lane kernel and returns read+write GB/s -- the measured
 * HBM peak bench.py prints beside the nominal 8 TB/s (SURVEY.md 8d "measure achievable with a stream-copy kernel and report both") */
int zkaes_stream_copy_bench(size_t bytes, int reps, double *gb_per_s);
/* per-box calibration of the hot kernel's integer roof (measurement only; ~`seconds` of GPU time, 0 < seconds <= 30).  out[0] = Fq377 reduced-radix Montgomery products per
 * second of the isolated product stream at four waves per SIMD (x 378 = v_mad_u64_u32 per second: the "peak" of roofline.int_multiplier), out[1] = the shader clock it ran at
 * in MHz (s_memtime ticks per second of s_memrealtime, median over waves and launches); out[2] = bucket additions per second of k_accumulate<EdwardsLaw>'s own loop at the
 * production launch shape over an L2-resident table (the kernel with its gathers made free), out[3] = its shader clock, out[4] = its shader cycles per addition per wave
 * (three waves share a SIMD); out[5] = rounds measured.  Medians over the rounds. */
int zkaes_int_rate_bench(double seconds, double out[8]);
/* AES witness only: fills z (padded instance + witness, one byte per variable) for a message under the key's circuit */
int zkaes_aes_witness(const zkaes_pk *pk, const uint8_t *message, size_t message_len, const uint8_t secret_key[16], uint8_t *z, size_t z_cap, size_t *z_len);

#ifdef __cplusplus
}
#endif
#endif
