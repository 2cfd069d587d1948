// csrc/capi_host.cpp -- the host-only entry points of include/zkaes.h: the verifier (src/lib.rs:116-136), proof / verifying-key (de)serialisation, circuit queries.
// No device, no HIP header: this file, marlin_codec.cpp and circuit.cpp are also built with -fsanitize=address,undefined,fuzzer (tests/fuzz_host.cpp).
#include "capi_common.hpp"
#include <algorithm>

namespace {
thread_local std::string g_err;
zk::Circuit compile(int kind, size_t len) { return kind == ZKAES_CIRCUIT_AES ? zk::compile_aes_circuit(len) : zk::compile_ops_circuit(kind); }
// ---- VK transport, library-private layout v2: "ZVK2", num_public_inputs (u64 LE), then the ark-serialize compressed image (marlin_codec.cpp).  Round 5's v1 was a memory
// image of the struct: reading it back from untrusted bytes put arbitrary limbs into field elements and an arbitrary byte into a bool, and skipped the curve / subgroup
// checks the ark path makes -- found while writing the fuzz target (tests/fuzz_host.cpp).  v2 goes through deserialize_vk_ark and inherits every check.
constexpr uint8_t VK_MAGIC[4] = {'Z', 'V', 'K', '2'};
}  // namespace
namespace zk { void capi_set_error(const std::string &m) { g_err = m; } }
using zk::capi::guard; using zk::capi::give; using zk::capi::fill_info; using zk::capi::next_pow2;

extern "C" {

const char *zkaes_last_error(void) { return g_err.c_str(); }
void zkaes_bytes_free(uint8_t *p) { free(p); }
void zkaes_vk_free(zkaes_vk *vk) { delete vk; }

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
        if (!vk || !out || !out_len) throw std::invalid_argument("null argument");
        std::vector<uint8_t> b(VK_MAGIC, VK_MAGIC + 4);
        for (int i = 0; i < 8; i++) b.push_back((uint8_t)((uint64_t)vk->vk.num_public_inputs >> (8 * i)));
        auto ark = zk::serialize_vk_ark(vk->vk);
        b.insert(b.end(), ark.begin(), ark.end());
        *out = give(b); *out_len = b.size();
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
        if (!bytes || !vk) throw std::invalid_argument("null argument");
        if (len < 12) throw std::runtime_error("vk_deserialize: truncated");
        if (memcmp(bytes, VK_MAGIC, 4) != 0) throw std::runtime_error("vk_deserialize: bad magic");
        uint64_t npub = 0;
        for (int i = 0; i < 8; i++) npub |= (uint64_t)bytes[4 + i] << (8 * i);
        std::unique_ptr<zkaes_vk> v(new zkaes_vk{zk::deserialize_vk_ark(bytes + 12, len - 12)});
        if (npub + 1 > v->vk.num_instance) throw std::runtime_error("vk_deserialize: more public inputs than instance variables");
        v->vk.num_public_inputs = (size_t)npub;
        *vk = v.release();
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

}  // extern "C"
