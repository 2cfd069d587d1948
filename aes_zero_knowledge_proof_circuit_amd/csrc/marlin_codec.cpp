// csrc/marlin_codec.cpp -- the host-only half of marlin.hpp: ark-serialize (de)coding of proofs and verifying keys, the replay of KZG10::setup's test_rng draws, and the
// Marlin verifier (src/lib.rs:116-136 -> simpleworks verify_proof -> ark-marlin verify).  Takes UNTRUSTED bytes (proofs, verifying keys): built a second time with
// -fsanitize=address,undefined,fuzzer and driven by tests/fuzz_host.cpp (tests/test_fuzz_host.py); no HIP header is included here.
#include "marlin_host.hpp"
#include <cstring>
#include <stdexcept>
#include "transcript.hpp"

namespace zk {
namespace {
using namespace hostx;

// Fq square root (Tonelli-Shanks; q - 1 = 2^46 * t) for point decompression
struct SqrtConsts {
    uint32_t t_limbs[12], tp1h_limbs[12], half_limbs[12];
    Fq377 z_t;   // nonresidue^t
    int S = 0;
    SqrtConsts() {
        uint32_t qm1[12];
        for (int i = 0; i < 12; i++) qm1[i] = FQ377_P[i];
        qm1[0] -= 1;
        auto shr1 = [](uint32_t *v) { for (int i = 0; i < 12; i++) v[i] = (v[i] >> 1) | (i < 11 ? v[i + 1] << 31 : 0); };
        memcpy(half_limbs, qm1, sizeof qm1); shr1(half_limbs);
        memcpy(t_limbs, qm1, sizeof qm1);
        while (!(t_limbs[0] & 1)) { shr1(t_limbs); S++; }
        memcpy(tp1h_limbs, t_limbs, sizeof t_limbs);
        tp1h_limbs[0] += 1;   // t odd -> no carry beyond limb 0 unless 0xffffffff
        shr1(tp1h_limbs);
        Fq377 minus_one = Fq377::one().neg();
        for (uint64_t c = 2;; c++) { Fq377 cand = Fq377::from_u64(c); if (cand.pow(half_limbs, 12) == minus_one) { z_t = cand.pow(t_limbs, 12); break; } }
    }
};
bool fq_sqrt(const Fq377 &a, Fq377 &out) {
    if (a.is_zero()) { out = a; return true; }
    static const SqrtConsts K;          // C++11 magic static: initialised exactly once, also when the first callers race (verifier-only processes, several threads)
    if (!(a.pow(K.half_limbs, 12) == Fq377::one())) return false;
    Fq377 c = K.z_t, x = a.pow(K.tp1h_limbs, 12), b = a.pow(K.t_limbs, 12);
    int m = K.S;
    while (!(b == Fq377::one())) {
        int i = 0;
        Fq377 bb = b;
        while (!(bb == Fq377::one())) { bb = bb.sqr(); i++; }
        Fq377 g = c;
        for (int k = 0; k < m - i - 1; k++) g = g.sqr();
        x = x * g; c = g.sqr(); b = b * c; m = i;
    }
    out = x;
    return true;
}

struct Reader {
    const uint8_t *p; size_t n, off = 0;
    void need(size_t k) { if (off + k > n) throw std::runtime_error("deserialize_proof: truncated input"); }
    uint8_t u8() { need(1); return p[off++]; }
    uint64_t u64() { need(8); uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[off + i] << (8 * i); off += 8; return v; }
    Fr fr() {
        need(32);
        uint32_t raw[8];
        for (int i = 0; i < 8; i++) raw[i] = (uint32_t)p[off + 4 * i] | (uint32_t)p[off + 4 * i + 1] << 8 | (uint32_t)p[off + 4 * i + 2] << 16 | (uint32_t)p[off + 4 * i + 3] << 24;
        off += 32;
        if (Fr::geq_mod(raw)) throw std::runtime_error("deserialize_proof: non-canonical field element");
        return Fr::from_raw(raw);
    }
    G1A g1() {
        need(48);
        uint8_t buf[48];
        memcpy(buf, p + off, 48); off += 48;
        bool inf = buf[47] & (1 << 6), y_gt = buf[47] & (1 << 7);
        if (inf && y_gt) throw std::runtime_error("deserialize_proof: invalid point flags (infinity and sign both set)");   // ark-serialize SWFlags::from_u8 -> None
        buf[47] &= 0x3f;
        uint32_t raw[12];
        for (int i = 0; i < 12; i++) raw[i] = (uint32_t)buf[4 * i] | (uint32_t)buf[4 * i + 1] << 8 | (uint32_t)buf[4 * i + 2] << 16 | (uint32_t)buf[4 * i + 3] << 24;
        if (Fq377::geq_mod(raw)) throw std::runtime_error("deserialize_proof: non-canonical x coordinate");
        if (inf) {
            for (int i = 0; i < 12; i++) if (raw[i]) throw std::runtime_error("deserialize_proof: point at infinity with a non-zero x coordinate");   // one encoding per point
            return G1A::inf();
        }
        G1A a; a.x = Fq377::from_raw(raw);
        Fq377 y;
        if (!fq_sqrt(a.x.sqr() * a.x + Bls377::b(), y)) throw std::runtime_error("deserialize_proof: x is not on the curve");
        uint32_t yr[12], nyr[12];
        y.to_raw(yr); y.neg().to_raw(nyr);
        bool gt = false;
        for (int i = 11; i >= 0; i--) if (yr[i] != nyr[i]) { gt = yr[i] > nyr[i]; break; }
        a.y = (gt == y_gt) ? y : y.neg();
        // ark-ec 0.3 GroupAffine::deserialize: is_in_correct_subgroup_assuming_on_curve, i.e. [r]P == O (the curve has cofactor (x-1)^2/3: a
        // small-order component would survive the pairing check and make proofs malleable)
        if (!XYZZ<Fq377>::from_affine(a).mul_raw(FR377_P, 8).is_inf()) throw std::runtime_error("deserialize_proof: point is not in the prime-order subgroup");
        return a;
    }
};


// Fq2 = Fq[u]/(u^2 + 5): lexicographic order of ark-ff's QuadExtField (c1 first, then c0) for the compressed-point sign flag
bool fq2_gt(const pairing::Fq2 &a, const pairing::Fq2 &b) {
    uint32_t x[12], y[12];
    a.c1.to_raw(x); b.c1.to_raw(y);
    for (int i = 11; i >= 0; i--) if (x[i] != y[i]) return x[i] > y[i];
    a.c0.to_raw(x); b.c0.to_raw(y);
    for (int i = 11; i >= 0; i--) if (x[i] != y[i]) return x[i] > y[i];
    return false;
}
// square root in Fq2 through the norm: a = a0 + a1 u, alpha = sqrt(a0^2 + 5 a1^2), delta = (a0 +- alpha)/2, c0 = sqrt(delta), c1 = a1/(2 c0)
bool fq2_sqrt(const pairing::Fq2 &a, pairing::Fq2 &out) {
    using pairing::Fq2;
    if (a.is_zero()) { out = a; return true; }
    Fq377 inv2 = Fq377::from_u64(2).inverse();
    if (a.c1.is_zero()) {
        Fq377 r;
        if (fq_sqrt(a.c0, r)) { out = Fq2{r, Fq377::zero()}; return true; }
        if (!fq_sqrt((a.c0 * Fq377::from_u64(5).inverse()).neg(), r)) return false;
        out = Fq2{Fq377::zero(), r};
        return true;
    }
    Fq377 alpha;
    if (!fq_sqrt(a.c0.sqr() + pairing::times5(a.c1.sqr()), alpha)) return false;
    Fq377 delta = (a.c0 + alpha) * inv2, c0;
    if (!fq_sqrt(delta, c0)) { delta = (a.c0 - alpha) * inv2; if (!fq_sqrt(delta, c0)) return false; }
    out = Fq2{c0, a.c1 * (c0.dbl()).inverse()};
    return out.sqr() == a;
}
void put_g2_compressed(Bytes &o, const pairing::G2Affine &p) {
    uint8_t buf[96] = {0};
    if (p.inf) { buf[95] |= 1 << 6; o.put(buf, 96); return; }
    uint32_t x0[12], x1[12];
    p.x.c0.to_raw(x0); p.x.c1.to_raw(x1);
    for (int i = 0; i < 48; i++) { buf[i] = (uint8_t)(x0[i / 4] >> (8 * (i % 4))); buf[48 + i] = (uint8_t)(x1[i / 4] >> (8 * (i % 4))); }
    if (fq2_gt(p.y, p.y.neg())) buf[95] |= 1 << 7;
    o.put(buf, 96);
}
void put_g2_uncompressed(Bytes &o, const pairing::G2Affine &p) {       // x (c0, c1), y (c0, c1 with the flags byte): 192 bytes
    const size_t at = o.b.size();
    if (p.inf) { for (int i = 0; i < 2; i++) o.field(Fq377::zero()); o.field(Fq377::one()); o.field(Fq377::zero()); o.b[at + 191] |= 1 << 6; return; }
    o.field(p.x.c0); o.field(p.x.c1); o.field(p.y.c0); o.field(p.y.c1);
}
pairing::G2Affine get_g2_compressed(Reader &r) {
    r.need(96);
    uint8_t buf[96];
    memcpy(buf, r.p + r.off, 96); r.off += 96;
    bool inf = buf[95] & (1 << 6), y_gt = buf[95] & (1 << 7);
    if (inf && y_gt) throw std::runtime_error("deserialize: invalid G2 point flags (infinity and sign both set)");
    buf[95] &= 0x3f;
    uint32_t raw[2][12];
    for (int h = 0; h < 2; h++) {
        for (int i = 0; i < 12; i++) { const uint8_t *q = buf + 48 * h + 4 * i; raw[h][i] = (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24; }
        if (Fq377::geq_mod(raw[h])) throw std::runtime_error("deserialize: non-canonical G2 x coordinate");
    }
    if (inf) {
        for (int h = 0; h < 2; h++) for (int i = 0; i < 12; i++) if (raw[h][i]) throw std::runtime_error("deserialize: G2 point at infinity with a non-zero x coordinate");
        return pairing::G2Affine::infinity();
    }
    pairing::G2Affine a;
    a.inf = false;
    a.x = pairing::Fq2{Fq377::from_raw(raw[0]), Fq377::from_raw(raw[1])};
    pairing::Fq2 y;
    if (!fq2_sqrt(a.x.sqr() * a.x + pairing::g2_b(), y)) throw std::runtime_error("deserialize: G2 x is not on the twist");
    a.y = (fq2_gt(y, y.neg()) == y_gt) ? y : y.neg();
    if (!pairing::g2_mul_raw(a, FR377_P, 8).inf) throw std::runtime_error("deserialize: G2 point is not in the prime-order subgroup");
    return a;
}


// canonical-integer comparison a < b (ark-ff's Ord on Fp)
bool fq_lt(const Fq377 &a, const Fq377 &b) {
    uint32_t x[12], y[12];
    a.to_raw(x); b.to_raw(y);
    for (int i = 11; i >= 0; i--) if (x[i] != y[i]) return x[i] < y[i];
    return false;
}
// ark-ec 0.3.0 `impl Distribution<GroupProjective<P>> for Standard` [RECALL]: loop { x = BaseField::rand; greatest = rng.gen::<bool>();
// get_point_from_x(x, greatest) } then scale_by_cofactor.  gen::<bool>() = top bit of one next_u32; y = (y < -y) ^ greatest ? y : -y.
G1A sample_g1(ChaChaRng &rng) {
    for (;;) {
        Fq377 x = rng.rand_field<Fq377>();
        bool greatest = (rng.next_u32() >> 31) != 0;
        Fq377 y;
        if (!fq_sqrt(x.sqr() * x + Bls377::b(), y)) continue;
        Fq377 ny = y.neg();
        G1A p; p.x = x; p.y = (fq_lt(y, ny) != greatest) ? y : ny;
        return XYZZ<Fq377>::from_affine(p).mul_raw(G1_377_COFACTOR, G1_377_COFACTOR_LIMBS).to_affine();
    }
}
pairing::G2Affine sample_g2(ChaChaRng &rng) {
    using pairing::Fq2;
    for (;;) {
        Fq2 x;
        x.c0 = rng.rand_field<Fq377>(); x.c1 = rng.rand_field<Fq377>();          // QuadExtField::rand: c0 then c1
        bool greatest = (rng.next_u32() >> 31) != 0;
        Fq2 y;
        if (!fq2_sqrt(x.sqr() * x + pairing::g2_b(), y)) continue;
        Fq2 ny = y.neg();
        pairing::G2Affine p; p.inf = false; p.x = x; p.y = (fq2_gt(ny, y) != greatest) ? y : ny;    // (y < -y) ^ greatest
        return pairing::g2_mul_raw(p, G2_377_COFACTOR, G2_377_COFACTOR_LIMBS);
    }
}

}  // namespace

// KZG10::setup's draws from ark_std::test_rng(), in upstream's order [RECALL ark-poly-commit 0.3.0]: beta, g, gamma_g, h
void kzg_setup_points(Fr &beta, G1A &g, G1A &gamma_g, pairing::G2Affine &h) {
    ChaChaRng rng(ark_test_rng_seed(), 12);
    beta = rng.rand_field<Fr>();
    g = sample_g1(rng);
    gamma_g = sample_g1(rng);
    h = sample_g2(rng);
}

// ark-serialize 0.3 compressed layout of ark_marlin::IndexVerifierKey<Fr, MarlinKZG10<Bls12_377, _>> [RECALL, SURVEY.md A.5]:
//   index_info   : num_variables, num_constraints, num_non_zero, num_instance_variables            (4 x u64 LE; PhantomData = 0 bytes)
//   index_comms  : u64 len, then per marlin_pc::Commitment: comm (G1 compressed 48 B), shifted_comm Option tag (0 = None)
//   verifier_key : marlin_pc::VerifierKey = kzg10::VerifierKey { g, gamma_g (G1 48 B each), h, beta_h (G2 compressed 96 B each); the
//                  prepared G2 elements are not serialized }, degree_bounds_and_shift_powers Option<Vec<(usize, G1)>> (tag, u64 len,
//                  (u64, 48 B) each), max_degree u64, supported_degree u64
// `uncompressed`: the image serialize_uncompressed writes (G1 96 B, G2 192 B, everything else identical) -- the form deserialize_unchecked reads
std::vector<uint8_t> serialize_vk_ark(const VerifyingKey &vk, bool uncompressed) {
    Bytes o;
    auto g2 = [&](const pairing::G2Affine &p) { if (uncompressed) put_g2_uncompressed(o, p); else put_g2_compressed(o, p); };
    o.u64(vk.num_variables); o.u64(vk.num_constraints); o.u64(vk.num_non_zero); o.u64(vk.num_instance);
    o.u64(6);
    for (int i = 0; i < 6; i++) { o.g1(vk.index_comms[i], uncompressed); o.u8(0); }
    o.g1(vk.g, uncompressed); o.g1(vk.gamma_g, uncompressed);
    g2(vk.h); g2(vk.beta_h);
    o.u8(1); o.u64(2);
    for (int i = 0; i < 2; i++) { o.u64(vk.degree_bounds[i]); o.g1(vk.shift_powers[i], uncompressed); }
    o.u64(vk.max_degree); o.u64(vk.supported_degree);
    return o.b;
}
VerifyingKey deserialize_vk_ark(const uint8_t *bytes, size_t len) {
    Reader r{bytes, len};
    VerifyingKey vk;
    vk.num_variables = r.u64(); vk.num_constraints = r.u64(); vk.num_non_zero = r.u64(); vk.num_instance = r.u64();
    if (r.u64() != 6) throw std::runtime_error("deserialize_vk: expected 6 index commitments");
    for (int i = 0; i < 6; i++) { vk.index_comms[i] = r.g1(); if (r.u8() != 0) throw std::runtime_error("deserialize_vk: index commitments carry no degree bound"); }
    vk.g = r.g1(); vk.gamma_g = r.g1();
    vk.h = get_g2_compressed(r); vk.beta_h = get_g2_compressed(r);
    if (r.u8() != 1 || r.u64() != 2) throw std::runtime_error("deserialize_vk: expected two enforced degree bounds");
    for (int i = 0; i < 2; i++) { vk.degree_bounds[i] = (size_t)r.u64(); vk.shift_powers[i] = r.g1(); }
    vk.max_degree = (size_t)r.u64(); vk.supported_degree = (size_t)r.u64();
    if (r.off != len) throw std::runtime_error("deserialize_vk: trailing bytes");
    if (vk.num_instance == 0 || vk.num_variables < vk.num_instance) throw std::runtime_error("deserialize_vk: inconsistent index info");
    // sizes a radix-2 domain of Fr (two-adicity 47) can hold; anything larger is not a key this library (or ark-marlin) could have produced, and the verifier's
    // domain arithmetic (next_pow2, roots of unity) is only defined below it
    const uint64_t LIM = (uint64_t)1 << 46;
    if (vk.num_variables > LIM || vk.num_constraints > LIM || vk.num_non_zero > LIM || vk.num_constraints == 0 || vk.num_non_zero == 0) throw std::runtime_error("deserialize_vk: index sizes out of range");
    if (vk.max_degree > 3 * LIM || vk.supported_degree > vk.max_degree || vk.degree_bounds[0] > vk.degree_bounds[1] || vk.degree_bounds[1] > vk.max_degree)
        throw std::runtime_error("deserialize_vk: inconsistent degree bounds");
    vk.num_public_inputs = (size_t)vk.num_instance - 1;   // the padded count; the verifier zero-pads shorter inputs the same way
    return vk;
}

std::vector<uint8_t> serialize_proof(const Proof &p) {
    Bytes o;
    static const int round_len[3] = {4, 3, 2};
    o.u64(3);
    int ci = 0;
    for (int r = 0; r < 3; r++) {
        o.u64(round_len[r]);
        for (int i = 0; i < round_len[r]; i++, ci++) {
            o.g1_compressed(p.comms[ci].comm);
            o.u8(p.comms[ci].has_shifted ? 1 : 0);
            if (p.comms[ci].has_shifted) o.g1_compressed(p.comms[ci].shifted);
        }
    }
    o.u64(4);
    for (int i = 0; i < 4; i++) o.field(p.evals[i]);
    o.u64(3);
    for (int i = 0; i < 3; i++) o.u8(0);                 // prover_messages: EmptyMessage -> Option::None
    o.u64(2);
    o.g1_compressed(p.w_beta); o.u8(1); o.field(p.random_v_beta);
    o.g1_compressed(p.w_gamma); o.u8(0);
    o.u8(0);                                             // pc_proof.evals: None
    return o.b;
}
Proof deserialize_proof(const uint8_t *bytes, size_t len) {
    Reader r{bytes, len};
    Proof p;
    static const uint64_t round_len[3] = {4, 3, 2};
    if (r.u64() != 3) throw std::runtime_error("deserialize_proof: expected 3 commitment rounds");
    int ci = 0;
    for (int rd = 0; rd < 3; rd++) {
        if (r.u64() != round_len[rd]) throw std::runtime_error("deserialize_proof: unexpected number of commitments");
        for (uint64_t i = 0; i < round_len[rd]; i++, ci++) {
            p.comms[ci].comm = r.g1();
            uint8_t tag = r.u8();
            if (tag > 1) throw std::runtime_error("deserialize_proof: bad Option tag");
            p.comms[ci].has_shifted = tag;
            if (tag) p.comms[ci].shifted = r.g1();
        }
    }
    if (r.u64() != 4) throw std::runtime_error("deserialize_proof: expected 4 evaluations");
    for (int i = 0; i < 4; i++) p.evals[i] = r.fr();
    if (r.u64() != 3) throw std::runtime_error("deserialize_proof: expected 3 prover messages");
    for (int i = 0; i < 3; i++) if (r.u8() != 0) throw std::runtime_error("deserialize_proof: non-empty prover message");
    if (r.u64() != 2) throw std::runtime_error("deserialize_proof: expected 2 opening proofs");
    p.w_beta = r.g1();
    if (r.u8() != 1) throw std::runtime_error("deserialize_proof: opening 0 must carry random_v");
    p.random_v_beta = r.fr();
    p.w_gamma = r.g1();
    if (r.u8() != 0) throw std::runtime_error("deserialize_proof: opening 1 must not carry random_v");
    if (r.u8() != 0) throw std::runtime_error("deserialize_proof: unexpected pc_proof.evals");
    if (r.off != len) throw std::runtime_error("deserialize_proof: trailing bytes");
    return p;
}

std::vector<Fr> ciphertext_to_public_input(const uint8_t *ct, size_t len) {
    std::vector<Fr> v;
    v.reserve(len * 8);
    for (size_t i = 0; i < len; i++) for (int b = 0; b < 8; b++) v.push_back(((ct[i] >> b) & 1) ? Fr::one() : Fr::zero());
    return v;
}

// =====================================================================================================================
// verifier (host only)
bool verify(const VerifyingKey &vk, const std::vector<Fr> &public_input_in, const Proof &proof) {
    using X = XYZZ<Fq377>;
    // pad the public input to |X| - 1 (ark-marlin verify)
    size_t m = next_pow2(public_input_in.size() + 1);
    std::vector<Fr> pub(public_input_in);
    pub.resize(std::max(public_input_in.size(), m - 1), Fr::zero());
    if (m != vk.num_instance) return false;               // InvalidPublicInputLength / instance does not match the index
    size_t n = next_pow2(vk.num_constraints), k = next_pow2(vk.num_non_zero);
    FiatShamirRng fs;
    {
        Bytes o;
        o.put("MARLIN-2019", 11);
        o.u64(vk.num_variables); o.u64(vk.num_constraints); o.u64(vk.num_non_zero);
        for (int i = 0; i < 6; i++) { Commitment ic; ic.comm = vk.index_comms[i]; o.commitment_tobytes(ic); }
        for (auto &v : pub) o.field(v);
        fs.initialize(o.b);
    }
    auto absorb_comms = [&](int from, int cnt) { Bytes o; for (int i = 0; i < cnt; i++) o.commitment_tobytes(proof.comms[from + i]); fs.absorb(o.b); };
    auto sample_outside_h = [&]() { Fr t; do { t = fs.rng().rand_field<Fr>(); } while (eval_vanishing(n, t).is_zero()); return t; };
    absorb_comms(0, 4);
    Fr alpha = sample_outside_h();
    Fr eta_a = fs.rng().rand_field<Fr>(), eta_b = fs.rng().rand_field<Fr>(), eta_c = fs.rng().rand_field<Fr>();
    absorb_comms(4, 3);
    Fr beta = sample_outside_h();
    absorb_comms(7, 2);
    Fr gamma = fs.rng().rand_field<Fr>();
    { Bytes o; for (auto &v : proof.evals) o.field(v); fs.absorb(o.b); }
    Fr ch;
    { uint64_t lo = fs.rng().next_u64(), hi = fs.rng().next_u64(); uint32_t raw[8] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), 0, 0, 0, 0}; ch = Fr::from_raw(raw); }
    // degree-bound shape checks (commitments g_1, g_2 carry shifted parts, the others must not)
    for (int i = 0; i < 9; i++) if (proof.comms[i].has_shifted != (i == 5 || i == 7)) return false;
    const Fr g1_b = proof.evals[0], g2_g = proof.evals[1], t_b = proof.evals[2], zb_b = proof.evals[3];
    // ---- construct_linear_combinations
    Fr vh_alpha = eval_vanishing(n, alpha), vh_beta = eval_vanishing(n, beta), vx_beta = eval_vanishing(m, beta), vk_gamma = eval_vanishing(k, gamma);
    Fr r_alpha_at_beta = (vh_alpha - vh_beta) * (alpha - beta).inverse();
    // x(beta) = sum_i L_i(beta) x_i over the X domain, x = [1, public inputs...]
    Fr x_at_beta = Fr::zero();
    {
        int lg_m = log2_exact(m);
        Fr gx = domain_gen(lg_m), m_inv = Fr::from_u64(m).inverse();
        if (vx_beta.is_zero()) {                                   // beta in X: the Lagrange basis is an indicator
            Fr e = Fr::one();
            for (size_t i = 0; i < m; i++) { if (e == beta) x_at_beta = i == 0 ? Fr::one() : pub[i - 1]; e = e * gx; }
        } else {
            // L_i(beta) = v_X(beta) * g^i / (m * (beta - g^i))
            std::vector<Fr> den(m);
            Fr e = Fr::one();
            for (size_t i = 0; i < m; i++) { den[i] = beta - e; e = e * gx; }
            // batch inversion
            std::vector<Fr> pre(m);
            Fr accp = Fr::one();
            for (size_t i = 0; i < m; i++) { pre[i] = accp; accp = accp * den[i]; }
            Fr inv = accp.inverse();
            for (size_t i = m; i-- > 0;) { Fr d = den[i]; den[i] = inv * pre[i]; inv = inv * d; }
            e = Fr::one();
            Fr common = vx_beta * m_inv;
            for (size_t i = 0; i < m; i++) {
                const Fr xi = i == 0 ? Fr::one() : pub[i - 1];
                if (!xi.is_zero()) x_at_beta = x_at_beta + common * e * den[i] * xi;
                e = e * gx;
            }
        }
    }
    Fr bmul = gamma * g2_g + t_b * Fr::from_u64(k).inverse();
    Fr vv = vh_alpha * vh_beta;
    // commitments by label
    const G1A &C_w = proof.comms[0].comm, &C_za = proof.comms[1].comm, &C_zb = proof.comms[2].comm, &C_mask = proof.comms[3].comm;
    const G1A &C_t = proof.comms[4].comm, &C_g1 = proof.comms[5].comm, &C_h1 = proof.comms[6].comm, &C_g2 = proof.comms[7].comm, &C_h2 = proof.comms[8].comm;
    const G1A *IX = vk.index_comms;   // row col a_val b_val c_val row_col
    auto add_scaled = [](X &acc, const G1A &c, const Fr &s) { if (s == Fr::one()) acc.madd(c); else acc.add(mul_fr(X::from_affine(c), s)); };
    // outer_sumcheck: commitment and expected evaluation (constants moved to the evaluation side, check_combinations)
    X outer = X::inf();
    add_scaled(outer, C_mask, Fr::one());
    add_scaled(outer, C_za, r_alpha_at_beta * (eta_a + eta_c * zb_b));
    add_scaled(outer, C_w, (t_b * vx_beta).neg());
    add_scaled(outer, C_h1, vh_beta.neg());
    Fr outer_eval = Fr::zero() - (r_alpha_at_beta * eta_b * zb_b) - ((t_b * x_at_beta).neg()) - ((beta * g1_b).neg());
    X inner = X::inf();
    add_scaled(inner, IX[2], eta_a * vv); add_scaled(inner, IX[3], eta_b * vv); add_scaled(inner, IX[4], eta_c * vv);
    add_scaled(inner, IX[0], alpha * bmul); add_scaled(inner, IX[1], beta * bmul); add_scaled(inner, IX[5], bmul.neg());
    add_scaled(inner, C_h2, vk_gamma.neg());
    Fr inner_eval = Fr::zero() - ((alpha * beta * bmul).neg());
    // ---- combine_and_normalize per query point (labels in BTreeSet order)
    Fr chp[5]; chp[0] = Fr::one(); for (int i = 1; i < 5; i++) chp[i] = chp[i - 1] * ch;
    auto shift_power = [&](size_t bound) -> const G1A & { return bound == vk.degree_bounds[0] ? vk.shift_powers[0] : vk.shift_powers[1]; };
    // beta: g_1 (ch^0, shifted ch^1), outer_sumcheck (ch^2), t (ch^3), z_b (ch^4)
    X comb_b = X::inf(); Fr val_b = Fr::zero();
    add_scaled(comb_b, C_g1, chp[0]); val_b = val_b + g1_b * chp[0];
    { X adj = X::from_affine(proof.comms[5].shifted); adj.add(mul_fr(X::from_affine(shift_power(n - 2)), g1_b).neg()); comb_b.add(mul_fr(adj, chp[1])); }
    comb_b.add(mul_fr(outer, chp[2])); val_b = val_b + outer_eval * chp[2];
    add_scaled(comb_b, C_t, chp[3]); val_b = val_b + t_b * chp[3];
    add_scaled(comb_b, C_zb, chp[4]); val_b = val_b + zb_b * chp[4];
    // gamma: g_2 (ch^0, shifted ch^1), inner_sumcheck (ch^2)
    X comb_g = X::inf(); Fr val_g = Fr::zero();
    add_scaled(comb_g, C_g2, chp[0]); val_g = val_g + g2_g * chp[0];
    { X adj = X::from_affine(proof.comms[7].shifted); adj.add(mul_fr(X::from_affine(shift_power(k - 2)), g2_g).neg()); comb_g.add(mul_fr(adj, chp[1])); }
    comb_g.add(mul_fr(inner, chp[2])); val_g = val_g + inner_eval * chp[2];
    // ---- KZG10::batch_check with 128-bit randomizers from generate_rand()
    ChaChaRng vrng(ark_test_rng_seed(), 12);
    X total_c = X::inf(), total_w = X::inf();
    Fr g_mult = Fr::zero(), gg_mult = Fr::zero(), randomizer = Fr::one();
    struct Item { X c; Fr z, v; const G1A *w; bool has_rv; Fr rv; } items[2] = {{comb_b, beta, val_b, &proof.w_beta, true, proof.random_v_beta}, {comb_g, gamma, val_g, &proof.w_gamma, false, Fr::zero()}};
    for (auto &it : items) {
        X tmp = mul_fr(X::from_affine(*it.w), it.z);
        tmp.add(it.c);
        g_mult = g_mult + randomizer * it.v;
        if (it.has_rv) gg_mult = gg_mult + randomizer * it.rv;
        total_c.add(mul_fr(tmp, randomizer));
        total_w.add(mul_fr(X::from_affine(*it.w), randomizer));
        uint64_t lo = vrng.next_u64(), hi = vrng.next_u64();
        uint32_t raw[8] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), 0, 0, 0, 0};
        randomizer = Fr::from_raw(raw);
    }
    total_c.add(mul_fr(X::from_affine(vk.g), g_mult).neg());
    total_c.add(mul_fr(X::from_affine(vk.gamma_g), gg_mult).neg());
    G1A Ps[2] = {total_w.neg().to_affine(), total_c.to_affine()};
    pairing::G2Affine Qs[2] = {vk.beta_h, vk.h};
    return pairing::pairing_product_is_one(Ps, Qs, 2);
}

}  // namespace zk
