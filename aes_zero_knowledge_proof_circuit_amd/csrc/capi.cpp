// csrc/capi.cpp -- the entry points of include/zkaes.h that need a GPU: key synthesis, the prover, key-handle queries (host-only ones: capi_host.cpp)
#include "capi_common.hpp"
#include <algorithm>
#include "gpu.hpp"

using zk::capi::guard; using zk::capi::give; using zk::capi::fill_info; using zk::capi::next_pow2;

extern "C" {

void zkaes_pk_free(zkaes_pk *pk) { delete pk; }
int zkaes_device_count(void) { return zk::gpu::device_count(); }

int zkaes_synthesize_keys_ex2(int kind, size_t len, size_t nc, size_t nv, size_t nnz, unsigned flags, zkaes_pk **pk, zkaes_vk **vk) {
    return guard([&] {
        if (flags & ~(unsigned)ZKAES_KEY_NO_TABLES) throw std::invalid_argument("synthesize_keys: unknown flag bits");
        zk::SrsLiterals srs; srs.num_constraints = nc; srs.num_variables = nv; srs.num_non_zero = nnz;
        auto k = zk::synthesize_keys(kind, len, srs, (flags & ZKAES_KEY_NO_TABLES) ? (unsigned)zk::KEY_NO_TABLES : 0u);
        zkaes_vk *v = new zkaes_vk{k->vk()};
        zkaes_pk *p = new zkaes_pk{std::move(k)};
        if (pk) *pk = p; else delete p;
        if (vk) *vk = v; else delete v;
    });
}
int zkaes_synthesize_keys_ex(int kind, size_t len, size_t nc, size_t nv, size_t nnz, zkaes_pk **pk, zkaes_vk **vk) {
    return zkaes_synthesize_keys_ex2(kind, len, nc, nv, nnz, 0u, pk, vk);
}
int zkaes_synthesize_keys(size_t len, zkaes_pk **pk, zkaes_vk **vk) {
    zk::SrsLiterals d;
    return zkaes_synthesize_keys_ex(ZKAES_CIRCUIT_AES, len, d.num_constraints, d.num_variables, d.num_non_zero, pk, vk);
}
int zkaes_encrypt_seeded(const uint8_t *msg, size_t len, const uint8_t key[16], const zkaes_pk *pk, const uint8_t *seed, uint8_t **proof, size_t *proof_len) {
    return guard([&] {
        if (!pk || !proof || !proof_len) throw std::invalid_argument("null argument");
        zk::Proof p = pk->pk->prove_aes(msg, len, key, seed);
        auto b = zk::serialize_proof(p);
        *proof = give(b); *proof_len = b.size();
    });
}
int zkaes_encrypt(const uint8_t *msg, size_t len, const uint8_t key[16], const zkaes_pk *pk, uint8_t **proof, size_t *proof_len) {
    return zkaes_encrypt_seeded(msg, len, key, pk, nullptr, proof, proof_len);
}
static void pack_proofs(const std::vector<zk::Proof> &ps, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens) {
    std::vector<uint8_t> all;
    for (size_t i = 0; i < ps.size(); i++) {
        auto b = zk::serialize_proof(ps[i]);
        if (proof_lens) proof_lens[i] = b.size();
        all.insert(all.end(), b.begin(), b.end());
    }
    *proofs = give(all); *proofs_len = all.size();
}
// the unseeded multi-proof entry points draw a fresh seed from the OS per call.  The reference's fixed ark_std::test_rng() stream for every proof (byte parity with the
// oracle; NOT zero-knowledge across proofs) is reachable only explicitly, through the *_seeded entry points with a NULL seed -- no environment variable downgrades a caller.
int zkaes_encrypt_chunked_seeded_at(const uint8_t *msg, size_t len, const uint8_t key[16], const zkaes_pk *pk, const uint8_t *zk_seed32, uint64_t first_proof_index, uint8_t **proofs,
                                    size_t *proofs_len, size_t *proof_lens, size_t n_chunks) {
    return guard([&] {
        if (!pk || !proofs || !proofs_len || !key || (!msg && len)) throw std::invalid_argument("null argument");
        size_t chunk = pk->pk->circuit().n_blocks * 16;
        if (chunk == 0 || len % chunk || len / chunk != n_chunks) throw std::invalid_argument("message length must be n_chunks * the key's plaintext length");
        pack_proofs(pk->pk->prove_aes_chunked(msg, len, key, pk->pk->contexts(), zk_seed32, first_proof_index), proofs, proofs_len, proof_lens);
    });
}
int zkaes_encrypt_chunked_seeded(const uint8_t *msg, size_t len, const uint8_t key[16], const zkaes_pk *pk, const uint8_t *zk_seed32, uint8_t **proofs, size_t *proofs_len,
                                 size_t *proof_lens, size_t n_chunks) {
    return zkaes_encrypt_chunked_seeded_at(msg, len, key, pk, zk_seed32, 0, proofs, proofs_len, proof_lens, n_chunks);
}
int zkaes_encrypt_chunked(const uint8_t *msg, size_t len, const uint8_t key[16], const zkaes_pk *pk, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens, size_t n_chunks) {
    uint8_t seed[32];
    { int rc = guard([&] { zk::os_random_seed(seed); }); if (rc) return rc; }
    return zkaes_encrypt_chunked_seeded_at(msg, len, key, pk, seed, 0, proofs, proofs_len, proof_lens, n_chunks);
}
int zkaes_encrypt_batch_seeded_at(size_t n, const uint8_t *messages, size_t messages_len, const uint8_t *secret_keys, size_t secret_keys_len, const zkaes_pk *pk,
                                  const uint8_t *zk_seed32, uint64_t first_proof_index, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens) {
    return guard([&] {
        if (!pk || !proofs || !proofs_len || (n && (!messages || !secret_keys))) throw std::invalid_argument("null argument");
        size_t chunk = pk->pk->circuit().n_blocks * 16;
        if (messages_len != n * chunk) throw std::invalid_argument("messages must hold n x " + std::to_string(chunk) + " bytes (the key's plaintext length)");
        if (secret_keys_len != n * 16) throw std::invalid_argument("secret_keys must hold n x 16 bytes");
        pack_proofs(pk->pk->prove_aes_batch(messages, secret_keys, n, pk->pk->contexts(), zk_seed32, first_proof_index), proofs, proofs_len, proof_lens);
    });
}
int zkaes_encrypt_batch_seeded(size_t n, const uint8_t *messages, size_t messages_len, const uint8_t *secret_keys, size_t secret_keys_len, const zkaes_pk *pk,
                               const uint8_t *zk_seed32, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens) {
    return zkaes_encrypt_batch_seeded_at(n, messages, messages_len, secret_keys, secret_keys_len, pk, zk_seed32, 0, proofs, proofs_len, proof_lens);
}
int zkaes_encrypt_batch(size_t n, const uint8_t *messages, const uint8_t *secret_keys, const zkaes_pk *pk, uint8_t **proofs, size_t *proofs_len, size_t *proof_lens) {
    size_t chunk = pk ? pk->pk->circuit().n_blocks * 16 : 0;
    uint8_t seed[32];
    { int rc = guard([&] { zk::os_random_seed(seed); }); if (rc) return rc; }
    return zkaes_encrypt_batch_seeded_at(n, messages, n * chunk, secret_keys, n * 16, pk, seed, 0, proofs, proofs_len, proof_lens);
}
int zkaes_prove_ops(const zkaes_pk *pk, uint32_t x, uint32_t y, const uint8_t *seed, uint8_t **proof, size_t *proof_len) {
    return guard([&] {
        if (!pk || !proof || !proof_len) throw std::invalid_argument("null argument");
        auto b = zk::serialize_proof(pk->pk->prove_ops(x, y, seed));
        *proof = give(b); *proof_len = b.size();
    });
}
int zkaes_pk_info(const zkaes_pk *pk, uint64_t out[12]) {
    return guard([&] {
        fill_info(pk->pk->circuit(), out);
        out[9] = pk->pk->vk().num_non_zero; out[10] = next_pow2(pk->pk->vk().num_constraints); out[11] = next_pow2(pk->pk->vk().num_non_zero);
    });
}
int zkaes_pk_serialize_ark_to_file_ex(const zkaes_pk *pk, const char *path, int uncompressed, uint64_t *bytes_written) {
    return guard([&] {
        if (!pk || !path) throw std::invalid_argument("null argument");
        uint64_t n = pk->pk->serialize_ark_to_file(path, uncompressed != 0);
        if (bytes_written) *bytes_written = n;
    });
}
int zkaes_pk_serialize_ark_to_file(const zkaes_pk *pk, const char *path, uint64_t *bytes_written) { return zkaes_pk_serialize_ark_to_file_ex(pk, path, 0, bytes_written); }
int zkaes_set_default_contexts(size_t n) { return guard([&] { if (n > 64) throw std::invalid_argument("set_default_contexts: at most 64 prover contexts per key"); zk::set_default_contexts(n); }); }
int zkaes_srs_hold(int hold) { return guard([&] { zk::srs_hold(hold != 0); }); }
int zkaes_pk_set_contexts(zkaes_pk *pk, size_t n) {
    return guard([&] {
        if (!pk) throw std::invalid_argument("null argument");
        pk->pk->set_contexts(n);
    });
}
int zkaes_pk_get_contexts(const zkaes_pk *pk, size_t *n) {
    return guard([&] {
        if (!pk || !n) throw std::invalid_argument("null argument");
        *n = pk->pk->contexts();
    });
}
int zkaes_pk_srs_info(const zkaes_pk *pk, uint64_t out[6], double secs[2]) {
    return guard([&] {
        if (!pk || !out) throw std::invalid_argument("null argument");
        pk->pk->srs_info(out, secs);
    });
}
int zkaes_pk_op_lists(const zkaes_pk *pk, const uint8_t *msg, size_t len, const uint8_t key[16], int throughput_path, uint8_t **json, size_t *json_len) {
    return guard([&] {
        if (!pk || !json || !json_len || !key || (!msg && len)) throw std::invalid_argument("null argument");
        std::string o = pk->pk->op_lists_json(msg, len, key, throughput_path != 0);
        *json = give(std::vector<uint8_t>(o.begin(), o.end())); *json_len = o.size();
    });
}
int zkaes_pk_tables_built(const zkaes_pk *pk, int *built, uint64_t *table_bytes) {
    return guard([&] {
        if (!pk || !built) throw std::invalid_argument("null argument");
        *built = pk->pk->tables_built(table_bytes) ? 1 : 0;
    });
}
int zkaes_pk_msm_partial_dev(const zkaes_pk *pk, const uint8_t *scalars, size_t n_local, size_t offset, void *dev_out, size_t dev_out_bytes) {
    return guard([&] {
        if (!pk || (!scalars && n_local)) throw std::invalid_argument("null argument");
        if (!dev_out || dev_out_bytes < 192) throw std::invalid_argument("zkaes_pk_msm_partial_dev: device buffer too small for the partial sum (192 bytes)");
        pk->pk->msm_powers_partial_device(scalars, n_local, offset, dev_out);
    });
}
int zkaes_pk_debug_fetch(const zkaes_pk *pk, const char *name, uint8_t **out, size_t *len) {
    return guard([&] { auto b = pk->pk->debug_fetch(name); *out = give(b); *len = b.size(); });
}
int zkaes_aes_witness(const zkaes_pk *pk, const uint8_t *msg, size_t len, const uint8_t key[16], uint8_t *z, size_t z_cap, size_t *z_len) {
    return guard([&] {
        auto v = pk->pk->aes_witness(msg, len, key);
        if (z_len) *z_len = v.size();
        if (z) { if (z_cap < v.size()) throw std::invalid_argument("z buffer too small"); memcpy(z, v.data(), v.size()); }
    });
}
int zkaes_pk_timings(const zkaes_pk *pk, double out[6]) {
    return guard([&] { const auto &t = pk->pk->last_timings(); out[0] = t.witness_ms; out[1] = t.round1_ms; out[2] = t.round2_ms; out[3] = t.round3_ms; out[4] = t.open_ms; out[5] = t.total_ms; });
}
int zkaes_msm_stats(double out[5], int reset) {
    return guard([&] {
        auto s = zk::gpu::msm_stats(reset != 0);
        out[0] = s.accumulate_ms; out[1] = s.total_ms; out[2] = (double)s.points; out[3] = (double)s.launches; out[4] = (double)s.pairs;
    });
}

}  // extern "C"
