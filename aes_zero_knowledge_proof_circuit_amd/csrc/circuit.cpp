// csrc/circuit.cpp -- see circuit.hpp.  Host-only C++; no device code.
#include "circuit.hpp"
#include <algorithm>
#include <array>
#include <stdexcept>
#include "trace_layout.h"

namespace zk {
namespace {

constexpr uint32_t WIT = 1u << 28;   // variable ids: 0 = One, i = Instance(i), WIT + j = Witness(j)  (ark Variable order)

// ---- Boolean literal: constant, or variable with optional negation (ark-r1cs-std Boolean::{Constant, Is, Not})
struct Bit {
    int32_t v;
    static Bit konst(bool b) { return Bit{b ? -1 : -2}; }
    static Bit is(uint32_t var) { return Bit{(int32_t)(var << 1)}; }
    bool is_const() const { return v < 0; }
    bool cval() const { return v == -1; }
    bool neg() const { return v & 1; }
    uint32_t var() const { return (uint32_t)v >> 1; }
    Bit operator!() const { return is_const() ? konst(!cval()) : Bit{v ^ 1}; }
};
using Byte = std::array<Bit, 8>;    // LSB first
using Word = std::array<Bit, 32>;

// ---- sorted sparse linear combination (ark-relations LinearCombination: sorted by Variable, duplicates merged)
struct LC {
    std::vector<std::pair<uint32_t, int64_t>> t;
    void add(int64_t c, uint32_t var) {
        auto it = std::lower_bound(t.begin(), t.end(), var, [](const std::pair<uint32_t, int64_t> &a, uint32_t b) { return a.first < b; });
        if (it != t.end() && it->first == var) it->second += c; else t.insert(it, {var, c});
    }
    void add(int64_t c, Bit b) {           // c * lc(b)
        if (b.is_const()) { if (b.cval()) add(c, 0u); }
        else if (!b.neg()) add(c, b.var());
        else { add(c, 0u); add(-c, b.var()); }
    }
};

struct RawMatrix {
    std::vector<uint32_t> rowptr{0};
    std::vector<uint32_t> var;
    std::vector<int64_t> coeff;
    void push(const LC &l) {
        for (auto &e : l.t) if (e.second != 0) { var.push_back(e.first); coeff.push_back(e.second); }   // make_row drops zero coefficients
        rowptr.push_back((uint32_t)var.size());
    }
};

class Builder {
  public:
    RawMatrix A, B, C;
    uint32_t n_instance = 1;              // Variable::One
    uint32_t n_witness = 0;
    std::vector<uint32_t> inst_desc{(WD_CONST << WD_KIND_SHIFT) | 1u};
    std::vector<uint32_t> wit_desc;
    std::vector<uint32_t> sbox_tmpl, sbox_in_off;

    uint32_t new_witness() { wit_desc.push_back(0xffffffffu); return WIT + n_witness++; }
    uint32_t new_input() { inst_desc.push_back(0xffffffffu); return n_instance++; }
    void enforce(const LC &a, const LC &b, const LC &c) { A.push(a); B.push(b); C.push(c); }
    void set_desc(uint32_t var, uint32_t d) { (var >= WIT ? wit_desc[var - WIT] : inst_desc[var]) = d; }
    // tag the variable behind `b` (if it was created at or after witness watermark `mark`) with a trace bit
    void tag_bytebit(Bit b, uint32_t mark, uint32_t off, int bit) {
        if (b.is_const() || b.var() < WIT + mark) return;
        set_desc(b.var(), (WD_BYTEBIT << WD_KIND_SHIFT) | (off << 4) | ((uint32_t)bit << 1) | (b.neg() ? 1u : 0u));
    }

    // AllocatedBool::new_variable: (1 - a) * a = 0
    Bit alloc(bool input, uint32_t off, int bit) {
        uint32_t v = input ? new_input() : new_witness();
        LC a, b, c;
        a.add(1, 0u); a.add(-1, v); b.add(1, v);
        enforce(a, b, c);
        set_desc(v, (WD_BYTEBIT << WD_KIND_SHIFT) | (off << 4) | ((uint32_t)bit << 1));
        return Bit::is(v);
    }
    uint32_t raw_xor(uint32_t a, uint32_t b) {        // (a + a) * b = a + b - c
        uint32_t r = new_witness();
        LC A_, B_, C_;
        A_.add(1, a); A_.add(1, a); B_.add(1, b); C_.add(1, a); C_.add(1, b); C_.add(-1, r);
        enforce(A_, B_, C_);
        return r;
    }
    Bit bxor(Bit a, Bit b) {
        if (a.is_const()) return a.cval() ? !b : b;
        if (b.is_const()) return b.cval() ? !a : a;
        if (a.neg() != b.neg()) { Bit is = a.neg() ? b : a, nt = a.neg() ? a : b; return !Bit::is(raw_xor(is.var(), nt.var())); }
        return Bit::is(raw_xor(a.var(), b.var()));
    }
    Bit band(Bit a, Bit b) {
        if (a.is_const()) return a.cval() ? b : Bit::konst(false);
        if (b.is_const()) return b.cval() ? a : Bit::konst(false);
        LC A_, B_, C_;
        if (!a.neg() && !b.neg()) { A_.add(1, a.var()); B_.add(1, b.var()); }                       // and
        else if (a.neg() && b.neg()) { A_.add(1, 0u); A_.add(-1, a.var()); B_.add(1, 0u); B_.add(-1, b.var()); }   // nor
        else { Bit is = a.neg() ? b : a, nt = a.neg() ? a : b; A_.add(1, is.var()); B_.add(1, 0u); B_.add(-1, nt.var()); }   // and_not
        uint32_t r = new_witness();
        C_.add(1, r);
        enforce(A_, B_, C_);
        return Bit::is(r);
    }
    Bit bor(Bit a, Bit b) {
        if (a.is_const()) return a.cval() ? Bit::konst(true) : b;
        if (b.is_const()) return b.cval() ? Bit::konst(true) : a;
        if (!a.neg() && !b.neg()) {                     // (1-a) * (1-b) = (1-c)
            uint32_t r = new_witness();
            LC A_, B_, C_;
            A_.add(1, 0u); A_.add(-1, a.var()); B_.add(1, 0u); B_.add(-1, b.var()); C_.add(1, 0u); C_.add(-1, r);
            enforce(A_, B_, C_);
            return Bit::is(r);
        }
        if (a.neg() && b.neg()) return !band(!b, !a);   // (b @ Not, a @ Not) => a.not().and(b.not()).not() with a = second operand
        Bit is = a.neg() ? b : a, nt = a.neg() ? a : b;
        return !band(!is, !nt);
    }
    Bit select(Bit cond, Bit t, Bit f) {
        if (cond.is_const()) return cond.cval() ? t : f;
        if (cond.neg()) return select(!cond, f, t);
        if (f.is_const() && !f.cval()) return band(cond, t);
        if (t.is_const() && !t.cval()) return band(!cond, f);
        if (t.is_const() && t.cval()) return bor(cond, f);
        if (f.is_const() && f.cval()) return bor(!cond, t);
        uint32_t r = new_witness();                     // cond * (t - f) = r - f
        LC A_, B_, C_;
        A_.add(1, cond); B_.add(1, t); B_.add(-1, f); C_.add(1, r); C_.add(-1, f);
        enforce(A_, B_, C_);
        return Bit::is(r);
    }
    void enforce_equal(Bit self, Bit other) {           // difference * 1 = 0
        if (self.is_const() && other.is_const()) return;
        LC d, one, z;
        if (self.is_const() || other.is_const()) {
            Bit c = self.is_const() ? self : other, x = self.is_const() ? other : self;
            bool one_minus = (c.cval() && !x.neg()) || (!c.cval() && x.neg());
            if (one_minus) { d.add(1, 0u); d.add(-1, x.var()); } else d.add(1, x.var());
        } else if (!self.neg() && !other.neg()) { d.add(1, other.var()); d.add(-1, self.var()); }
        else if (self.neg() && other.neg()) { d.add(1, self.var()); d.add(-1, other.var()); }
        else { Bit is = self.neg() ? other : self, nt = self.neg() ? self : other; d.add(1, 0u); d.add(-1, nt.var()); d.add(-1, is.var()); }
        one.add(1, 0u);
        enforce(d, one, z);
    }

    // ---- bytes / words
    static Byte const_byte(uint8_t v) { Byte r; for (int i = 0; i < 8; i++) r[i] = Bit::konst((v >> i) & 1); return r; }
    Byte alloc_byte(bool input, uint32_t off) { Byte r; for (int i = 0; i < 8; i++) r[i] = alloc(input, off, i); return r; }
    // xor whose result byte lives at trace offset `off`
    Byte xor_byte(const Byte &a, const Byte &b, uint32_t off) {
        Byte r;
        for (int i = 0; i < 8; i++) { uint32_t mark = n_witness; r[i] = bxor(a[i], b[i]); tag_bytebit(r[i], mark, off, i); }
        return r;
    }
    static Byte shl(const Byte &a, int n) { Byte r; for (int i = 0; i < 8; i++) r[i] = i >= n ? a[i - n] : Bit::konst(false); return r; }
    static Byte shr(const Byte &a, int n) { Byte r; for (int i = 0; i < 8; i++) r[i] = i + n < 8 ? a[i + n] : Bit::konst(false); return r; }

    // UInt8::conditionally_select_power_of_two_vector over the constant S-box table (src/aes_circuit.rs:243-248)
    Byte sbox(const Byte &x, const std::vector<Byte> &table, uint32_t in_off) {
        uint32_t inst = (uint32_t)sbox_in_off.size();
        sbox_in_off.push_back(in_off);
        bool record = sbox_tmpl.empty();
        uint32_t tix = 0;
        std::vector<Byte> cur(table), nxt;
        for (int lvl = 0; lvl < 8; lvl++) {
            nxt.assign(cur.size() / 2, Byte{});
            for (size_t j = 0; j < cur.size(); j += 2)
                for (int k = 0; k < 8; k++) {
                    uint32_t mark = n_witness;
                    Bit r = select(x[lvl], cur[j + 1][k], cur[j][k]);
                    nxt[j / 2][k] = r;
                    if (n_witness != mark) {
                        if (n_witness != mark + 1 || r.is_const() || r.var() != WIT + mark) throw std::logic_error("sbox: unexpected allocation pattern");
                        uint32_t entry = ((uint32_t)lvl << 12) | ((uint32_t)(j / 2) << 4) | ((uint32_t)k << 1);
                        if (record) sbox_tmpl.push_back(entry);
                        else if (tix >= sbox_tmpl.size() || sbox_tmpl[tix] != entry) throw std::logic_error("sbox: template differs between instances");
                        set_desc(r.var(), (WD_SBOX << WD_KIND_SHIFT) | (inst << 11) | (tix << 1) | (r.neg() ? 1u : 0u));
                        tix++;
                    }
                }
            cur.swap(nxt);
        }
        if (!record && tix != sbox_tmpl.size()) throw std::logic_error("sbox: template length differs");
        return cur[0];
    }
    // src/helpers/mod.rs:11-42
    Byte helpers_add(const Byte &augend, const Byte &addend) {
        Byte sum; Bit carry = Bit::konst(false);
        for (int i = 0; i < 8; i++) {
            Bit a = augend[i], b = addend[i];
            sum[i] = bxor(bxor(carry, a), b);
            carry = bor(band(!carry, band(a, b)), band(carry, bor(a, b)));
        }
        return sum;
    }
    // src/helpers/mod.rs:44-64 (multiplier is a constant here; the reference branches on its bit values)
    Byte helpers_multiply(const Byte &multiplicand, uint8_t multiplier) {
        Byte product = const_byte(0);
        for (int i = 0; i < 8; i++)
            if ((multiplier >> i) & 1) product = helpers_add(product, i ? shl(multiplicand, i) : multiplicand);
        return product;
    }
};

CsrMatrix finalize_matrix(const RawMatrix &m, uint32_t n_inst_padded, size_t rows_padded) {
    CsrMatrix o;
    o.rowptr = m.rowptr;
    o.rowptr.resize(rows_padded + 1, (uint32_t)m.var.size());
    o.col.resize(m.var.size());
    for (size_t i = 0; i < m.var.size(); i++) o.col[i] = m.var[i] < WIT ? m.var[i] : n_inst_padded + (m.var[i] - WIT);
    o.coeff = m.coeff;
    return o;
}

// ark-marlin padding (pad_input_for_indexer_and_prover + make_matrices_square) and final column numbering
Circuit finish(Builder &b, int kind, size_t n_blocks, size_t trace_bytes) {
    Circuit c;
    c.kind = kind; c.n_blocks = n_blocks; c.trace_bytes = trace_bytes;
    c.raw_constraints = b.A.rowptr.size() - 1; c.raw_instance = b.n_instance; c.raw_witness = b.n_witness;
    size_t ninst = 1;
    while (ninst < b.n_instance) ninst <<= 1;
    std::vector<uint32_t> desc(b.inst_desc);
    desc.resize(ninst, (WD_CONST << WD_KIND_SHIFT) | 0u);            // padded inputs are zero
    size_t nwit = b.n_witness, ncons = c.raw_constraints;
    if (ninst + nwit > ncons) ncons = ninst + nwit;                   // dummy 0*0=0 rows
    else nwit = ncons - ninst;                                        // dummy witnesses = F::one()
    desc.insert(desc.end(), b.wit_desc.begin(), b.wit_desc.end());
    desc.resize(ninst + nwit, (WD_CONST << WD_KIND_SHIFT) | 1u);
    for (uint32_t d : desc) if (d == 0xffffffffu) throw std::logic_error("circuit: variable without a witness descriptor");
    c.num_instance = ninst; c.num_witness = nwit; c.num_constraints = ncons;
    c.A = finalize_matrix(b.A, (uint32_t)ninst, ncons); c.B = finalize_matrix(b.B, (uint32_t)ninst, ncons); c.C = finalize_matrix(b.C, (uint32_t)ninst, ncons);
    c.desc.swap(desc);
    c.sbox_in_off = b.sbox_in_off; c.sbox_tmpl = b.sbox_tmpl;
    return c;
}

}  // namespace

uint8_t aes_sbox_value(uint8_t x) {   // algebraic S-box (FIPS-197 5.1.1; equals the 256 constants of src/aes_circuit.rs:433-694)
    auto mul = [](uint8_t a, uint8_t b) { uint8_t p = 0; for (int i = 0; i < 8; i++) { if (b & 1) p ^= a; bool h = a & 0x80; a <<= 1; if (h) a ^= 0x1B; b >>= 1; } return p; };
    uint8_t inv = 0;
    if (x) { inv = 1; for (int i = 0; i < 254; i++) inv = mul(inv, x); }
    uint8_t r = inv;
    for (int i = 1; i <= 4; i++) r ^= (uint8_t)((inv << i) | (inv >> (8 - i)));
    return r ^ 0x63;
}

Circuit compile_aes_circuit(size_t len) {
    if (len % 16) throw std::invalid_argument("Input must be 16 bytes length when adding round key");
    size_t nb = len / 16;
    Builder b;
    std::vector<Byte> table(256);
    for (int i = 0; i < 256; i++) table[i] = Builder::const_byte(aes_sbox_value((uint8_t)i));
    auto blk = [](size_t bi) { return (uint32_t)(TR_BLOCK0 + bi * TR_BLOCK_STRIDE); };
    // message then key witnesses (src/lib.rs:70-76, 82-88)
    std::vector<Byte> msg(len);
    for (size_t i = 0; i < len; i++) msg[i] = b.alloc_byte(false, blk(i / 16) + TR_BL_MSG + (uint32_t)(i % 16));
    std::array<Byte, 16> key;
    for (int i = 0; i < 16; i++) key[i] = b.alloc_byte(false, TR_KEY + i);
    // derive_keys (src/aes_circuit.rs:20-129): words are big-endian byte quadruples; UInt32::xor runs LSB-first over the
    // u32, i.e. byte 3 first (to_u32, :201-212)
    std::array<std::array<Byte, 4>, 44> w;
    for (int i = 0; i < 4; i++) for (int k = 0; k < 4; k++) w[i][k] = key[4 * i + k];
    static const uint8_t rc[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36};
    for (int i = 4; i < 44; i++) {
        if (i % 4 == 0) {
            int q = i / 4 - 1;
            std::array<Byte, 4> sub;
            for (int k = 0; k < 4; k++) {
                int src = (k + 1) % 4;                                            // rotate_word: rotate_left(1)
                sub[k] = b.sbox(w[i - 1][src], table, (uint32_t)(TR_KS_W + 4 * (i - 1) + src));
            }
            for (int k = 3; k >= 0; k--) w[i][k] = b.xor_byte(w[i - 4][k], sub[k], (uint32_t)(TR_KS_PRE + 4 * q + k));
            w[i][0] = b.xor_byte(w[i][0], Builder::const_byte(rc[q]), 0);         // Rcon: constant operand, free
        } else {
            for (int k = 3; k >= 0; k--) w[i][k] = b.xor_byte(w[i - 4][k], w[i - 1][k], (uint32_t)(TR_KS_W + 4 * i + k));
        }
    }
    // per block rounds (src/lib.rs:194-278)
    std::vector<Byte> ct(len);
    for (size_t bi = 0; bi < nb; bi++) {
        uint32_t base = blk(bi);
        std::array<Byte, 16> s, t, u;
        for (int i = 0; i < 16; i++) s[i] = b.xor_byte(msg[16 * bi + i], key[i], base + TR_BL_S + i);          // :196 raw key
        for (int r = 1; r <= 10; r++) {
            for (int i = 0; i < 16; i++) {                                                                      // substitute_bytes
                t[i] = b.sbox(s[i], table, base + TR_BL_S + 16 * (r - 1) + i);
            }
            for (int c = 0; c < 4; c++) for (int rr = 0; rr < 4; rr++) u[4 * c + rr] = t[4 * ((c + rr) % 4) + rr];   // shift_rows
            if (r <= 9) {                                                                                        // mix_columns
                for (int c = 0; c < 4; c++) {
                    std::array<Byte, 4> a{u[4 * c], u[4 * c + 1], u[4 * c + 2], u[4 * c + 3]}, xb;
                    for (int k = 0; k < 4; k++) {
                        Byte sh = Builder::shr(a[k], 7), h, one = Builder::const_byte(1);
                        for (int i = 0; i < 8; i++) h[i] = b.band(sh[i], one[i]);
                        Byte m = b.helpers_multiply(h, 0x1B);
                        xb[k] = b.xor_byte(Builder::shl(a[k], 1), m, base + TR_BL_XT + 16 * (r - 1) + 4 * c + k);
                    }
                    static const int order[4][5][2] = {{{1, 0}, {0, 3}, {0, 2}, {1, 1}, {0, 1}}, {{1, 1}, {0, 0}, {0, 3}, {1, 2}, {0, 2}},
                                                       {{1, 2}, {0, 1}, {0, 0}, {1, 3}, {0, 3}}, {{1, 3}, {0, 2}, {0, 1}, {1, 0}, {0, 0}}};
                    for (int o = 0; o < 4; o++) {
                        Byte acc = order[o][0][0] ? xb[order[o][0][1]] : a[order[o][0][1]];
                        for (int p = 1; p < 5; p++) {
                            const Byte &x = order[o][p][0] ? xb[order[o][p][1]] : a[order[o][p][1]];
                            acc = b.xor_byte(acc, x, base + TR_BL_MP + 64 * (r - 1) + 4 * (4 * c + o) + (p - 1));
                        }
                        t[4 * c + o] = acc;
                    }
                }
            } else {
                t = u;
            }
            for (int i = 0; i < 16; i++) {
                Byte rk;                                                          // round key r = words 4r..4r+3 as bytes
                rk = w[4 * r + i / 4][i % 4];
                s[i] = b.xor_byte(t[i], rk, base + TR_BL_S + 16 * r + i);
            }
        }
        for (int i = 0; i < 16; i++) ct[16 * bi + i] = s[i];
    }
    // public inputs + equality (src/lib.rs:282-286)
    for (size_t i = 0; i < len; i++) {
        Byte pi = b.alloc_byte(true, blk(i / 16) + TR_BL_S + 160 + (uint32_t)(i % 16));
        for (int k = 0; k < 8; k++) b.enforce_equal(pi[k], ct[i][k]);
    }
    return finish(b, CIRCUIT_AES, nb, TR_BLOCK0 + nb * TR_BLOCK_STRIDE);
}

// src/ops.rs:8-29.  Trace: x (4 B LE) | y (4 B LE) | result (8 B LE)
Circuit compile_ops_circuit(int kind) {
    Builder b;
    Word x, y;
    for (int i = 0; i < 32; i++) x[i] = b.alloc(false, (uint32_t)(i / 8), i % 8);
    for (int i = 0; i < 32; i++) y[i] = b.alloc(false, (uint32_t)(4 + i / 8), i % 8);
    if (kind == CIRCUIT_OPS_XOR) {
        for (int i = 0; i < 32; i++) { uint32_t mark = b.n_witness; Bit r = b.bxor(x[i], y[i]); b.tag_bytebit(r, mark, (uint32_t)(8 + i / 8), i % 8); }
    } else {
        // UInt32::addmany: 33 result bits (max_value = 2 * u32::MAX), then 0 * 0 = sum(2^i x_i) + sum(2^i y_i) - sum(2^i r_i)
        LC lc;
        int64_t coeff = 1;
        for (int i = 0; i < 32; i++) { lc.add(coeff, x[i]); coeff *= 2; }
        coeff = 1;
        for (int i = 0; i < 32; i++) { lc.add(coeff, y[i]); coeff *= 2; }
        coeff = 1;
        for (int i = 0; i < 33; i++) { Bit r = b.alloc(false, (uint32_t)(8 + i / 8), i % 8); lc.add(-coeff, r.var()); coeff *= 2; }
        LC z;
        b.enforce(z, z, lc);
    }
    return finish(b, kind, 0, 16);
}

}  // namespace zk
