// csrc/marlin.hpp -- Marlin (AHP for R1CS + MarlinKZG10) keys, GPU prover and CPU verifier of libzkaes.
//
// Mirrors what the reference reaches through simpleworks::marlin (not under /root/reference; restated from the published
// ark-marlin 0.3.0 / ark-poly-commit 0.3.0 algorithms):
//   synthesize_keys  -> generate_universal_srs + generate_proving_and_verifying_keys   (/root/reference/src/lib.rs:138-174)
//   prove            -> generate_proof                                                 (src/lib.rs:60-114)
//   verify           -> verify_proof                                                   (src/lib.rs:116-136)
// The prover's polynomial work (witness, SpMV, NTTs, MSMs, divisions, evaluations) runs on the GPU; the Fiat-Shamir
// transcript, the <=3-term hiding MSMs and the final window sums stay on the host.  The verifier is host-only.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "circuit.hpp"
#include "ec.cuh"
#include "pairing.hpp"

namespace zk {

using Fr = Fr377;
using G1A = Affine<Fq377>;

struct Commitment { G1A comm = G1A::inf(); bool has_shifted = false; G1A shifted = G1A::inf(); };

struct Proof {
    Commitment comms[9];   // w z_a z_b mask_poly | t g_1 h_1 | g_2 h_2
    Fr evals[4];           // g_1(beta) g_2(gamma) t(beta) z_b(beta)   (sorted by label)
    G1A w_beta, w_gamma;   // pc_proof.proof[0].w, [1].w
    Fr random_v_beta;      // pc_proof.proof[0].random_v = Some(..); [1].random_v = None
};
// ark-serialize 0.3 compressed layout of ark_marlin::Proof
std::vector<uint8_t> serialize_proof(const Proof &p);
// throws std::runtime_error on malformed input
Proof deserialize_proof(const uint8_t *bytes, size_t len);

struct SrsLiterals { size_t num_constraints = 866944, num_variables = 513, num_non_zero = 4062064; };   // src/lib.rs:141

struct VerifyingKey {
    uint64_t num_variables = 0, num_constraints = 0, num_non_zero = 0, num_instance = 0;   // IndexInfo (padded)
    size_t num_public_inputs = 0;     // unpadded instance count minus One (128 per block)
    G1A index_comms[6];               // row col a_val b_val c_val row_col
    G1A g, gamma_g;
    pairing::G2Affine h, beta_h;
    size_t degree_bounds[2] = {0, 0}; // |H|-2, |K|-2 sorted ascending
    G1A shift_powers[2];              // powers_of_g[max_degree - bound]
    size_t supported_degree = 0, max_degree = 0;
};

// the four draws of KZG10::setup from ark_std::test_rng(): trapdoor beta, base points g, gamma_g (G1) and h (G2)
void kzg_setup_points(Fr &beta, G1A &g, G1A &gamma_g, pairing::G2Affine &h);
// ark-serialize 0.3 compressed layout of ark_marlin::IndexVerifierKey (what a Rust caller gets from CanonicalSerialize::serialize)
// uncompressed = the serialize_uncompressed image (G1 96 B, G2 192 B): what deserialize_unchecked reads
std::vector<uint8_t> serialize_vk_ark(const VerifyingKey &vk, bool uncompressed = false);
VerifyingKey deserialize_vk_ark(const uint8_t *bytes, size_t len);

struct ProverTimings { double witness_ms = 0, round1_ms = 0, round2_ms = 0, round3_ms = 0, open_ms = 0, total_ms = 0; };

class ProvingKeyImpl;
class ProvingKey {
  public:
    ~ProvingKey();
    const VerifyingKey &vk() const;
    const Circuit &circuit() const;
    // encrypt(): message length must equal the length the key was synthesized for; zk_seed = 32-byte StdRng seed or nullptr for
    // ark_std::test_rng()'s (what simpleworks::marlin::generate_rand() returns)
    Proof prove_aes(const uint8_t *message, size_t len, const uint8_t key[16], const uint8_t *zk_seed);
    // ceil(len / chunk) independent chunk-proofs of a long ECB message, `n_contexts` proofs in flight on separate HIP streams
    // zk_seed: 32-byte seed, domain-separated per proof: proof i of the call draws from StdRng(Blake2s(seed || (index_offset + i) as u64 LE)), so callers that
    // split one job over several calls / ranks pass the job-global index of their first proof and reuse one seed.  nullptr = the reference's fixed
    // test_rng seed for every proof (byte-parity mode for tests; NOT zero-knowledge across proofs)
    std::vector<Proof> prove_aes_chunked(const uint8_t *message, size_t len, const uint8_t key[16], size_t n_contexts, const uint8_t *zk_seed = nullptr, uint64_t index_offset = 0);
    // n independent (message_i, key_i) pairs, each message of the key's plaintext length; keys = n x 16 bytes
    std::vector<Proof> prove_aes_batch(const uint8_t *messages, const uint8_t *keys, size_t n, size_t n_contexts, const uint8_t *zk_seed = nullptr, uint64_t index_offset = 0);
    Proof prove_ops(uint32_t x, uint32_t y, const uint8_t *zk_seed);
    // witness generation only (kernels aes_trace + witness_expand): z = padded instance || witness, one byte per variable
    std::vector<uint8_t> aes_witness(const uint8_t *message, size_t len, const uint8_t key[16]);
    const ProverTimings &last_timings() const;
    // one proof with the op recorder open: JSON of the transforms and MSMs the library actually launched (marlin.cpp; SURVEY.md 8d op lists)
    std::string op_lists_json(const uint8_t *message, size_t len, const uint8_t key[16], bool throughput_path);
    // test / parity hooks: copy an intermediate of the last proof to the host. names: "z" (bytes), "z_a_evals","z_b_evals",
    // polys "w","z_a","z_b","mask_poly","t","g_1","h_1","g_2","h_2" (Montgomery Fr), index "row","col","a_val","b_val","c_val","row_col"
    std::vector<uint8_t> debug_fetch(const std::string &name) const;
    // were the fixed-base window tables of the SRS built (they are skipped under KEY_NO_TABLES or when device memory is short)?  *bytes = their size
    bool tables_built(uint64_t *bytes = nullptr) const;
    // the universal SRS behind this key (one per process, device and SRS literals; shared by every key over it): out = {max_degree, points per copy, copies (1 = no
    // window tables), device bytes, keys sharing it now, bytes of the Lagrange-basis points (shared per |H|, |X|)}; secs = {seconds this key's synthesis spent BUILDING
    // the SRS (0 when it was shared), seconds of the whole synthesis}
    void srs_info(uint64_t out[6], double secs[2]) const;
    // proofs in flight per multi-proof call on this key (prove_aes_chunked / _batch with n_contexts = contexts()); 0 restores the process default
    size_t contexts() const;
    void set_contexts(size_t n);
    // ark-serialize image of the arkworks IndexProverKey this key corresponds to, streamed to `path`; returns the bytes written (marlin.cpp has the layout).
    // compressed (48-byte points): what IndexProverKey::serialize writes and ::deserialize reads (square root + subgroup check per point on the Rust side);
    // uncompressed (96-byte points): serialize_uncompressed's image, read by ::deserialize_uncompressed or -- without any check -- ::deserialize_unchecked
    uint64_t serialize_ark_to_file(const std::string &path, bool uncompressed = false) const;
    // ONE commitment-sized MSM sharded by point range over ranks, on the prover's own path (the key's SRS on the twisted Edwards model, window tables, one bucket set):
    // sum_i scalars[i] * powers_of_g[offset + i], i < n_local (scalars: host, n_local x 32 B Montgomery Fr), left as ONE XYZZ point (192 B) in device memory at dev_out
    // -- the rank's row of the all-gather; gpu::msm_fold_points_device adds the ranks' rows.  Needs a key with tables.
    void msm_powers_partial_device(const uint8_t *scalars, size_t n_local, size_t offset, void *dev_out);
    ProvingKeyImpl *impl;
};

// universal_setup(literals) + index; GPU required.  flags: KEY_NO_TABLES = do not build the fixed-base window tables of the SRS (13 copies, 10-42 GB per key):
// multi-proof calls then run 15 per-window-bucket windows instead of 13 table windows (~9 % fewer blocks/s), and the key fits a GPU that is short of memory.
// Without the flag the tables are built when memory allows (hipMemGetInfo) and silently skipped otherwise.
enum : unsigned { KEY_NO_TABLES = 1u };
std::unique_ptr<ProvingKey> synthesize_keys(int circuit_kind, size_t message_len, const SrsLiterals &srs, unsigned flags = 0);
// hold = true: every universal / Lagrange SRS built (or alive) from now on stays resident after its last key is freed; false: back to "freed with the last key"
void srs_hold(bool hold);
// process default of ProvingKey::contexts(): ZKAES_CONTEXTS from the environment (read once), else ZKAES_DEFAULT_CONTEXTS
size_t default_contexts();
void set_default_contexts(size_t n);       // 0 = back to the environment / built-in default; at most 64
// 32 bytes from the operating system (getrandom): the default zero-knowledge seed of the multi-proof entry points
void os_random_seed(uint8_t out[32]);

// public_input: the instance values WITHOUT the leading One (0/1 as field elements), e.g. ciphertext bits LSB-first per byte
bool verify(const VerifyingKey &vk, const std::vector<Fr> &public_input, const Proof &proof);
std::vector<Fr> ciphertext_to_public_input(const uint8_t *ct, size_t len);   // src/helpers/mod.rs:84-93 per byte

}  // namespace zk
