// csrc/capi.cpp -- the extern "C" boundary declared in include/zkaes.h
#include "../../include/zkaes.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>
#include "gpu.hpp"
#include "marlin.hpp"

struct zkaes_pk { std::unique_ptr<zk::ProvingKey> pk; };
struct zkaes_vk { zk::VerifyingKey vk; };

namespace {
thread_local std::string g_err;
template <class Fn> int guard(Fn &&fn) {
    try { g_err.clear(); fn(); return 0; }
    catch (const std::exception &e) { g_err = e.what(); return 1; }
    catch (...) { g_err = "unknown error"; return 1; }
}
uint8_t *give(const std::vector<uint8_t> &v) { uint8_t *p = (uint8_t *)malloc(v.size() ? v.size() : 1); memcpy(p, v.data(), v.size()); return p; }
void fill_info(const zk::Circuit &c, uint64_t out[12]) {
    out[0] = c.raw_constraints; out[1] = c.raw_instance; out[2] = c.raw_witness;
    out[3] = c.A.nnz(); out[4] = c.B.nnz(); out[5] = c.C.nnz();
    out[6] = c.num_constraints; out[7] = c.num_instance; out[8] = c.num_witness; out[9] = 0; out[10] = 0; out[11] = 0;
}
zk::Circuit compile(int kind, size_t len) { return kind == ZKAES_CIRCUIT_AES ? zk::compile_aes_circuit(len) : zk::compile_ops_circuit(kind); }
size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

// ---- VK transport (private layout v1)
struct W { std::vector<uint8_t> b; void put(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); } template <class T> void pod(const T &v) { put(&v, sizeof v); } };
struct R { const uint8_t *p; size_t n, off = 0; template <class T> void pod(T &v) { if (off + sizeof v > n) throw std::runtime_error("vk_deserialize: truncated"); memcpy(&v, p + off, sizeof v); off += sizeof v; } };
}  // namespace
namespace zk { void capi_set_error(const std::string &m) { g_err = m; } }

extern "C" {

const char *zkaes_last_error(void) { return g_err.c_str(); }
void zkaes_bytes_free(uint8_t *p) { free(p); }
void zkaes_pk_free(zkaes_pk *pk) { delete pk; }
void zkaes_vk_free(zkaes_vk *vk) { delete vk; }
int zkaes_device_count(void) { return zk::gpu::device_count(); }
int zkaes_set_device(int ordinal);   // defined in runtime glue below

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
int zkaes_verify(const zkaes_vk *vk, const uint8_t *proof, size_t proof_len, const uint8_t *bits, size_t n_bits, int *accepted) {
    return guard([&] {
        if (!vk || !proof || !accepted) throw std::invalid_argument("null argument");
        zk::Proof p = zk::deserialize_proof(proof, proof_len);
        std::vector<zk::Fr> pub(n_bits);
        for (size_t i = 0; i < n_bits; i++) pub[i] = bits[i] ? zk::Fr::one() : zk::Fr::zero();
        *accepted = zk::verify(vk->vk, pub, p) ? 1 : 0;
    });
}
int zkaes_verify_encryption(const zkaes_vk *vk, const uint8_t *proof, size_t proof_len, const uint8_t *ct, size_t ct_len, int *accepted) {
    return guard([&] {
        if (!vk || !proof || !accepted) throw std::invalid_argument("null argument");
        zk::Proof p = zk::deserialize_proof(proof, proof_len);
        *accepted = zk::verify(vk->vk, zk::ciphertext_to_public_input(ct, ct_len), p) ? 1 : 0;
    });
}
int zkaes_proof_roundtrip(const uint8_t *proof, size_t proof_len, uint8_t **out, size_t *out_len) {
    return guard([&] { auto b = zk::serialize_proof(zk::deserialize_proof(proof, proof_len)); *out = give(b); *out_len = b.size(); });
}
int zkaes_vk_serialize(const zkaes_vk *vk, uint8_t **out, size_t *out_len) {
    return guard([&] {
        W w;
        uint32_t magic = 0x314b565au;   // "ZVK1"
        w.pod(magic);
        static_assert(std::is_trivially_copyable<zk::VerifyingKey>::value, "VerifyingKey must be POD for the v1 transport");
        w.pod(vk->vk);
        *out = give(w.b); *out_len = w.b.size();
    });
}
int zkaes_vk_serialize_ark(const zkaes_vk *vk, uint8_t **out, size_t *out_len) {
    return guard([&] {
        if (!vk || !out || !out_len) throw std::invalid_argument("null argument");
        auto b = zk::serialize_vk_ark(vk->vk);
        *out = give(b); *out_len = b.size();
    });
}
int zkaes_vk_serialize_ark_uncompressed(const zkaes_vk *vk, uint8_t **out, size_t *out_len) {
    return guard([&] {
        if (!vk || !out || !out_len) throw std::invalid_argument("null argument");
        auto b = zk::serialize_vk_ark(vk->vk, true);
        *out = give(b); *out_len = b.size();
    });
}
int zkaes_vk_deserialize_ark(const uint8_t *bytes, size_t len, zkaes_vk **vk) {
    return guard([&] {
        if (!bytes || !vk) throw std::invalid_argument("null argument");
        *vk = new zkaes_vk{zk::deserialize_vk_ark(bytes, len)};
    });
}
int zkaes_vk_deserialize(const uint8_t *bytes, size_t len, zkaes_vk **vk) {
    return guard([&] {
        R r{bytes, len};
        uint32_t magic; r.pod(magic);
        if (magic != 0x314b565au) throw std::runtime_error("vk_deserialize: bad magic");
        zkaes_vk *v = new zkaes_vk();
        try { r.pod(v->vk); } catch (...) { delete v; throw; }
        if (r.off != len) { delete v; throw std::runtime_error("vk_deserialize: trailing bytes"); }
        *vk = v;
    });
}
int zkaes_vk_from_trapdoor(const uint64_t info[7], const uint8_t *index_comms, const uint8_t *beta_b, zkaes_vk **vk) {
    return guard([&] {
        if (!info || !index_comms || !beta_b || !vk) throw std::invalid_argument("null argument");
        zk::Fr beta_in, beta;
        memcpy(beta_in.l, beta_b, 32);
        zk::G1A g, gamma_g;
        zk::pairing::G2Affine h;
        zk::kzg_setup_points(beta, g, gamma_g, h);                   // this library's own replay of KZG10::setup's draws from test_rng
        if (!(beta == beta_in)) throw std::invalid_argument("vk_from_trapdoor: beta is not the first Fr draw of ark_std::test_rng()");
        zkaes_vk *v = new zkaes_vk();
        zk::VerifyingKey &k = v->vk;
        k.num_variables = info[0]; k.num_constraints = info[1]; k.num_non_zero = info[2]; k.num_instance = info[3];
        k.num_public_inputs = info[4]; k.max_degree = info[5]; k.supported_degree = info[6];
        for (int i = 0; i < 6; i++) { memcpy(k.index_comms[i].x.l, index_comms + 96 * i, 48); memcpy(k.index_comms[i].y.l, index_comms + 96 * i + 48, 48); }
        auto mulg = [&](const zk::Fr &s) { return zk::mul_fr(zk::XYZZ<zk::Fq377>::from_affine(g), s).to_affine(); };
        k.g = g; k.gamma_g = gamma_g; k.h = h;
        uint32_t raw[8]; beta.to_raw(raw);
        k.beta_h = zk::pairing::g2_mul_raw(k.h, raw, 8);
        size_t n = next_pow2(k.num_constraints), kk = next_pow2(k.num_non_zero);
        k.degree_bounds[0] = std::min(n - 2, kk - 2); k.degree_bounds[1] = std::max(n - 2, kk - 2);
        for (int i = 0; i < 2; i++) k.shift_powers[i] = mulg(beta.pow_u64(k.max_degree - k.degree_bounds[i]));
        *vk = v;
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
int zkaes_circuit_info(int kind, size_t len, uint64_t out[12]) { return guard([&] { fill_info(compile(kind, len), out); }); }
int zkaes_circuit_matrix(int kind, size_t len, int which, uint64_t *n_rows, uint64_t *nnz, uint32_t *rowptr, uint32_t *col, int64_t *coeff) {
    return guard([&] {
        zk::Circuit c = compile(kind, len);
        const zk::CsrMatrix &m = which == 0 ? c.A : which == 1 ? c.B : c.C;
        if (n_rows) *n_rows = m.rows();
        if (nnz) *nnz = m.nnz();
        if (rowptr) memcpy(rowptr, m.rowptr.data(), m.rowptr.size() * 4);
        if (col) memcpy(col, m.col.data(), m.col.size() * 4);
        if (coeff) memcpy(coeff, m.coeff.data(), m.coeff.size() * 8);
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
