/* oracle/zko_r1cs.c -- TEST INFRASTRUCTURE (CPU oracle).
 *
 * Gate-level restatement of the reference circuit:
 *   src/lib.rs:60-114 (encrypt: message witnesses, key witnesses, circuit), :176-293
 *   (encrypt_and_generate_constraints: lookup table, derive_keys, per-block rounds, public inputs),
 *   src/aes_circuit.rs:20-129 (derive_keys), :131-212 (substitute_word / rotate_word / to_bytes_be /
 *   to_u32), :214-241 (add_round_key), :243-266 (substitute_byte(s)), :268-334 (shift_rows),
 *   :336-427 (mix_columns / gmix_column), src/helpers/mod.rs:11-64 (add / multiply),
 *   src/ops.rs:8-29 (toy xor / add).
 * on top of the gadget semantics of ark-r1cs-std 0.3.1 / ark-relations 0.3.0 (Cargo.lock:267,283; NOT under
 * /root/reference -> restated from the published algorithms; "parity unpinned" for
 * variable order / matrices): Boolean = Constant | Is(var) | Not(var); UInt8 = 8 Booleans LSB first;
 * xor / and / or / conditionally_select constant-fold exactly as Boolean's match arms do; a linear
 * combination is a list sorted by Variable (One < Instance(i) < Witness(j)) with merged duplicates; matrix rows
 * drop zero coefficients (ConstraintSystem::make_row).
 * simpleworks shift_left / shift_right / rotate_left (source unavailable) are taken as free re-wiring with
 * Constant(false) fill.
 */
#include "zko.h"
#include <stdlib.h>

/* ------------------------------------------------------------------ constraint system */
#define LC_MAX 160
typedef struct { int n; uint32_t var[LC_MAX]; int64_t coeff[LC_MAX]; } lc_t;
static void lc_zero(lc_t *l) { l->n = 0; }
/* LinearCombination += (coeff, var): keep sorted by var, merge duplicates (ark-relations AddAssign) */
static void lc_add(lc_t *l, int64_t coeff, uint32_t var) {
    int i = 0;
    while (i < l->n && l->var[i] < var) i++;
    if (i < l->n && l->var[i] == var) { l->coeff[i] += coeff; return; }
    if (l->n >= LC_MAX) abort();
    for (int k = l->n; k > i; k--) { l->var[k] = l->var[k - 1]; l->coeff[k] = l->coeff[k - 1]; }
    l->var[i] = var; l->coeff[i] = coeff; l->n++;
}
static void lc_add_lc(lc_t *l, const lc_t *o, int64_t scale) {
    for (int i = 0; i < o->n; i++) lc_add(l, scale * o->coeff[i], o->var[i]);
}
static void mat_init(zko_mat *m) { memset(m, 0, sizeof *m); m->cap = 1024; m->rowptr = malloc((m->cap + 1) * sizeof(size_t)); m->rowptr[0] = 0; m->nzcap = 4096; m->var = malloc(m->nzcap * 4); m->coeff = malloc(m->nzcap * 8); }
static void mat_free(zko_mat *m) { free(m->rowptr); free(m->var); free(m->coeff); }
static void mat_push_row(zko_mat *m, const lc_t *l) {
    if (m->n == m->cap) { m->cap *= 2; m->rowptr = realloc(m->rowptr, (m->cap + 1) * sizeof(size_t)); }
    for (int i = 0; i < l->n; i++) {
        if (l->coeff[i] == 0) continue;          /* make_row filters zero coefficients */
        if (m->nnz == m->nzcap) { m->nzcap *= 2; m->var = realloc(m->var, m->nzcap * 4); m->coeff = realloc(m->coeff, m->nzcap * 8); }
        m->var[m->nnz] = l->var[i]; m->coeff[m->nnz] = l->coeff[i]; m->nnz++;
    }
    m->n++;
    m->rowptr[m->n] = m->nnz;
}
zko_cs *zko_cs_new(int field_id) {
    zko_cs *cs = calloc(1, sizeof *cs);
    cs->field_id = field_id;
    cs->icap = 1024; cs->wcap = 4096;
    cs->instance_val = malloc(cs->icap); cs->witness_val = malloc(cs->wcap);
    cs->instance_val[0] = 1; cs->num_instance = 1;   /* Variable::One */
    mat_init(&cs->A); mat_init(&cs->B); mat_init(&cs->C);
    return cs;
}
void zko_cs_free(zko_cs *cs) { if (!cs) return; mat_free(&cs->A); mat_free(&cs->B); mat_free(&cs->C); free(cs->instance_val); free(cs->witness_val); free(cs); }
size_t zko_cs_num_constraints(const zko_cs *cs) { return cs->A.n; }
static uint32_t cs_new_witness(zko_cs *cs, int val) {
    if (cs->num_witness == cs->wcap) { cs->wcap *= 2; cs->witness_val = realloc(cs->witness_val, cs->wcap); }
    cs->witness_val[cs->num_witness] = (uint8_t)val;
    return ZKO_WIT_BASE + cs->num_witness++;
}
static uint32_t cs_new_input(zko_cs *cs, int val) {
    if (cs->num_instance == cs->icap) { cs->icap *= 2; cs->instance_val = realloc(cs->instance_val, cs->icap); }
    cs->instance_val[cs->num_instance] = (uint8_t)val;
    return cs->num_instance++;
}
static void cs_enforce(zko_cs *cs, const lc_t *a, const lc_t *b, const lc_t *c) { mat_push_row(&cs->A, a); mat_push_row(&cs->B, b); mat_push_row(&cs->C, c); }
static int cs_val(const zko_cs *cs, uint32_t var) { return var < ZKO_WIT_BASE ? cs->instance_val[var] : cs->witness_val[var - ZKO_WIT_BASE]; }

int zko_cs_is_satisfied(const zko_cs *cs) {
    for (size_t r = 0; r < cs->A.n; r++) {
        __int128 s[3] = {0, 0, 0};
        const zko_mat *M[3] = {&cs->A, &cs->B, &cs->C};
        for (int k = 0; k < 3; k++)
            for (size_t i = M[k]->rowptr[r]; i < M[k]->rowptr[r + 1]; i++) s[k] += (__int128)M[k]->coeff[i] * cs_val(cs, M[k]->var[i]);
        if (s[0] * s[1] != s[2]) return -(int)(r + 1);
    }
    return 1;
}
/* ark-marlin constraint_systems.rs: pad_input_for_indexer_and_prover, make_matrices_square */
void zko_cs_pad_for_marlin(zko_cs *cs) {
    uint32_t padded = 1;
    while (padded < cs->num_instance) padded <<= 1;
    while (cs->num_instance < padded) cs_new_input(cs, 0);
    size_t nvars = (size_t)cs->num_instance + cs->num_witness, ncons = cs->A.n;
    lc_t z; lc_zero(&z);
    if (nvars > ncons) { for (size_t i = ncons; i < nvars; i++) cs_enforce(cs, &z, &z, &z); }
    else { for (size_t i = nvars; i < ncons; i++) cs_new_witness(cs, 1); }
}

/* ------------------------------------------------------------------ Boolean gadget */
enum { B_CONST = 0, B_IS = 1, B_NOT = 2 };
typedef struct { uint8_t kind, cval; uint32_t var; } zb;
typedef struct { zb b[8]; } zu8;
typedef struct { zb b[32]; } zu32;
static zb zb_const(int v) { zb r = {B_CONST, (uint8_t)(v != 0), 0}; return r; }
static int zb_val(const zko_cs *cs, zb a) { return a.kind == B_CONST ? a.cval : (a.kind == B_IS ? cs_val(cs, a.var) : !cs_val(cs, a.var)); }
static zb zb_not(zb a) { if (a.kind == B_CONST) return zb_const(!a.cval); a.kind = (a.kind == B_IS) ? B_NOT : B_IS; return a; }
static zb zb_is(uint32_t var) { zb r = {B_IS, 0, var}; return r; }
/* l += s * lc(b) */
static void lc_add_bool(lc_t *l, zb b, int64_t s) {
    if (b.kind == B_CONST) { if (b.cval) lc_add(l, s, 0); }
    else if (b.kind == B_IS) lc_add(l, s, b.var);
    else { lc_add(l, s, 0); lc_add(l, -s, b.var); }
}
/* AllocatedBool::new_variable: witness/input + booleanity (1 - a) * a = 0 */
static zb zb_alloc(zko_cs *cs, int val, int input) {
    uint32_t v = input ? cs_new_input(cs, val) : cs_new_witness(cs, val);
    lc_t a, b, c; lc_zero(&a); lc_zero(&b); lc_zero(&c);
    lc_add(&a, 1, 0); lc_add(&a, -1, v);
    lc_add(&b, 1, v);
    cs_enforce(cs, &a, &b, &c);
    return zb_is(v);
}
/* AllocatedBool::xor : (a + a) * b = a + b - c */
static uint32_t ab_xor(zko_cs *cs, uint32_t a, uint32_t b) {
    uint32_t r = cs_new_witness(cs, cs_val(cs, a) ^ cs_val(cs, b));
    lc_t A, B, C; lc_zero(&A); lc_zero(&B); lc_zero(&C);
    lc_add(&A, 1, a); lc_add(&A, 1, a);
    lc_add(&B, 1, b);
    lc_add(&C, 1, a); lc_add(&C, 1, b); lc_add(&C, -1, r);
    cs_enforce(cs, &A, &B, &C);
    return r;
}
static zb zb_xor(zko_cs *cs, zb a, zb b) {
    if (a.kind == B_CONST) return a.cval ? zb_not(b) : b;
    if (b.kind == B_CONST) return b.cval ? zb_not(a) : a;
    if (a.kind != b.kind) {                        /* is.xor(not.not()).not(): the Is operand is `self` */
        zb is = a.kind == B_IS ? a : b, nt = a.kind == B_IS ? b : a;
        return zb_not(zb_is(ab_xor(cs, is.var, nt.var)));
    }
    return zb_is(ab_xor(cs, a.var, b.var));
}
/* AllocatedBool::and / and_not / nor : lcA * lcB = c */
static zb zb_and(zko_cs *cs, zb a, zb b) {
    if (a.kind == B_CONST) return a.cval ? b : zb_const(0);
    if (b.kind == B_CONST) return b.cval ? a : zb_const(0);
    lc_t A, B, C; lc_zero(&A); lc_zero(&B); lc_zero(&C);
    int val;
    uint32_t r;
    if (a.kind == B_IS && b.kind == B_IS) {        /* a * b = c */
        val = cs_val(cs, a.var) & cs_val(cs, b.var);
        r = cs_new_witness(cs, val);
        lc_add(&A, 1, a.var); lc_add(&B, 1, b.var);
    } else if (a.kind == B_NOT && b.kind == B_NOT) { /* nor: (1-a) * (1-b) = c */
        val = !cs_val(cs, a.var) & !cs_val(cs, b.var);
        r = cs_new_witness(cs, val);
        lc_add(&A, 1, 0); lc_add(&A, -1, a.var); lc_add(&B, 1, 0); lc_add(&B, -1, b.var);
    } else {                                       /* is.and_not(not): a * (1-b) = c */
        zb is = a.kind == B_IS ? a : b, nt = a.kind == B_IS ? b : a;
        val = cs_val(cs, is.var) & !cs_val(cs, nt.var);
        r = cs_new_witness(cs, val);
        lc_add(&A, 1, is.var); lc_add(&B, 1, 0); lc_add(&B, -1, nt.var);
    }
    lc_add(&C, 1, r);
    cs_enforce(cs, &A, &B, &C);
    return zb_is(r);
}
static zb zb_or(zko_cs *cs, zb a, zb b) {
    if (a.kind == B_CONST) return a.cval ? zb_const(1) : b;
    if (b.kind == B_CONST) return b.cval ? zb_const(1) : a;
    if (a.kind == B_IS && b.kind == B_IS) {        /* AllocatedBool::or: (1-a) * (1-b) = (1-c) */
        uint32_t r = cs_new_witness(cs, cs_val(cs, a.var) | cs_val(cs, b.var));
        lc_t A, B, C; lc_zero(&A); lc_zero(&B); lc_zero(&C);
        lc_add(&A, 1, 0); lc_add(&A, -1, a.var); lc_add(&B, 1, 0); lc_add(&B, -1, b.var); lc_add(&C, 1, 0); lc_add(&C, -1, r);
        cs_enforce(cs, &A, &B, &C);
        return zb_is(r);
    }
    /* a OR b = NOT((NOT a) AND (NOT b)); in Boolean::or's arms the first factor is the Is operand when mixed,
     * and (b @ Not, a @ Not) binds a = second operand for Not/Not */
    if (a.kind == B_NOT && b.kind == B_NOT) return zb_not(zb_and(cs, zb_not(b), zb_not(a)));
    zb is = a.kind == B_IS ? a : b, nt = a.kind == B_IS ? b : a;
    return zb_not(zb_and(cs, zb_not(is), zb_not(nt)));
}
/* Boolean::conditionally_select */
static zb zb_select(zko_cs *cs, zb cond, zb t, zb f) {
    if (cond.kind == B_CONST) return cond.cval ? t : f;
    if (cond.kind == B_NOT) return zb_select(cs, zb_not(cond), f, t);
    if (f.kind == B_CONST && !f.cval) return zb_and(cs, cond, t);
    if (t.kind == B_CONST && !t.cval) return zb_and(cs, zb_not(cond), f);
    if (t.kind == B_CONST && t.cval) return zb_or(cs, cond, f);
    if (f.kind == B_CONST && f.cval) return zb_or(cs, zb_not(cond), t);
    int val = zb_val(cs, cond) ? zb_val(cs, t) : zb_val(cs, f);
    uint32_t r = cs_new_witness(cs, val);
    lc_t A, B, C; lc_zero(&A); lc_zero(&B); lc_zero(&C);          /* cond * (a - b) = r - b */
    lc_add_bool(&A, cond, 1);
    lc_add_bool(&B, t, 1); { lc_t tmp; lc_zero(&tmp); lc_add_bool(&tmp, f, 1); lc_add_lc(&B, &tmp, -1); }
    lc_add(&C, 1, r); { lc_t tmp; lc_zero(&tmp); lc_add_bool(&tmp, f, 1); lc_add_lc(&C, &tmp, -1); }
    cs_enforce(cs, &A, &B, &C);
    return zb_is(r);
}
/* Boolean::conditional_enforce_equal(self, other, TRUE): difference * 1 = 0 */
static void zb_enforce_equal(zko_cs *cs, zb self, zb other) {
    lc_t d, one, z; lc_zero(&d); lc_zero(&one); lc_zero(&z);
    if (self.kind == B_CONST && other.kind == B_CONST) return;
    if (self.kind == B_CONST || other.kind == B_CONST) {
        zb c = self.kind == B_CONST ? self : other, x = self.kind == B_CONST ? other : self;
        int one_minus = (c.cval && x.kind == B_IS) || (!c.cval && x.kind == B_NOT);
        if (one_minus) { lc_add(&d, 1, 0); lc_add(&d, -1, x.var); } else lc_add(&d, 1, x.var);
    } else if (self.kind == B_IS && other.kind == B_IS) { lc_add(&d, 1, other.var); lc_add(&d, -1, self.var); }
    else if (self.kind == B_NOT && other.kind == B_NOT) { lc_add(&d, 1, self.var); lc_add(&d, -1, other.var); }
    else { zb is = self.kind == B_IS ? self : other, nt = self.kind == B_IS ? other : self; lc_add(&d, 1, 0); lc_add(&d, -1, nt.var); lc_add(&d, -1, is.var); }
    lc_add(&one, 1, 0);
    cs_enforce(cs, &d, &one, &z);
}

/* ------------------------------------------------------------------ UInt8 / UInt32 */
static zu8 u8_const(uint8_t v) { zu8 r; for (int i = 0; i < 8; i++) r.b[i] = zb_const((v >> i) & 1); return r; }
static zu8 u8_alloc(zko_cs *cs, uint8_t v, int input) { zu8 r; for (int i = 0; i < 8; i++) r.b[i] = zb_alloc(cs, (v >> i) & 1, input); return r; }
static uint8_t u8_val(const zko_cs *cs, const zu8 *a) { uint8_t v = 0; for (int i = 0; i < 8; i++) v |= (uint8_t)(zb_val(cs, a->b[i]) << i); return v; }
static zu8 u8_xor(zko_cs *cs, const zu8 *a, const zu8 *b) { zu8 r; for (int i = 0; i < 8; i++) r.b[i] = zb_xor(cs, a->b[i], b->b[i]); return r; }
static zu8 u8_select(zko_cs *cs, zb cond, const zu8 *t, const zu8 *f) { zu8 r; for (int i = 0; i < 8; i++) r.b[i] = zb_select(cs, cond, t->b[i], f->b[i]); return r; }
static zu32 u32_xor(zko_cs *cs, const zu32 *a, const zu32 *b) { zu32 r; for (int i = 0; i < 32; i++) r.b[i] = zb_xor(cs, a->b[i], b->b[i]); return r; }
static zu32 u32_const(uint32_t v) { zu32 r; for (int i = 0; i < 32; i++) r.b[i] = zb_const((v >> i) & 1); return r; }
/* simpleworks BitwiseOperationGadget [source unavailable]: logical shifts, Constant(false) fill, no constraints */
static zu8 u8_shl(const zu8 *a, int n) { zu8 r; for (int i = 0; i < 8; i++) r.b[i] = i >= n ? a->b[i - n] : zb_const(0); return r; }
static zu8 u8_shr(const zu8 *a, int n) { zu8 r; for (int i = 0; i < 8; i++) r.b[i] = i + n < 8 ? a->b[i + n] : zb_const(0); return r; }

/* src/aes_circuit.rs:201-212 to_u32: bytes reversed then bits LE => byte 0 is the most significant */
static zu32 to_u32(const zu8 v[4]) { zu32 r; for (int k = 0; k < 4; k++) for (int i = 0; i < 8; i++) r.b[8 * k + i] = v[3 - k].b[i]; return r; }
/* :188-199 to_bytes_be */
static void to_bytes_be(zu8 out[4], const zu32 *w) { for (int k = 0; k < 4; k++) for (int i = 0; i < 8; i++) out[k].b[i] = w->b[8 * (3 - k) + i]; }

/* src/aes_circuit.rs:243-248 substitute_byte = UInt8::conditionally_select_power_of_two_vector(byte.to_bits_be(), table):
 * level i selects on position[n-1-i] = bit i (LSB first); pairs (values[j] false, values[j+1] true) */
static zu8 substitute_byte(zko_cs *cs, const zu8 *byte, const zu8 *table) {
    zu8 cur[256], nxt[128];
    memcpy(cur, table, sizeof cur);
    int size = 256;
    for (int lvl = 0; lvl < 8; lvl++) {
        for (int j = 0; j < size; j += 2) nxt[j / 2] = u8_select(cs, byte->b[lvl], &cur[j + 1], &cur[j]);
        size /= 2;
        memcpy(cur, nxt, size * sizeof(zu8));
    }
    return cur[0];
}
/* :214-241 */
static void add_round_key(zko_cs *cs, zu8 out[16], const zu8 in[16], const zu8 rk[16]) { for (int i = 0; i < 16; i++) out[i] = u8_xor(cs, &in[i], &rk[i]); }
/* :250-266 */
static void substitute_bytes(zko_cs *cs, zu8 out[16], const zu8 in[16], const zu8 *table) { for (int i = 0; i < 16; i++) out[i] = substitute_byte(cs, &in[i], table); }
/* :268-334 (rotate_left of the row arrays = re-wiring) */
static void shift_rows(zu8 out[16], const zu8 in[16]) {
    for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) out[4 * c + r] = in[4 * ((c + r) % 4) + r];
}
/* src/helpers/mod.rs:11-42 ripple-carry add on bits (to_bits_be, iterate LSB first) */
static zu8 helpers_add(zko_cs *cs, const zu8 *augend, const zu8 *addend) {
    zu8 sum; zb carry = zb_const(0);
    for (int i = 0; i < 8; i++) {                      /* i = LSB .. MSB */
        zb a = augend->b[i], b = addend->b[i];
        sum.b[i] = zb_xor(cs, zb_xor(cs, carry, a), b);
        zb t1 = zb_and(cs, zb_not(carry), zb_and(cs, a, b));
        zb t2 = zb_and(cs, carry, zb_or(cs, a, b));
        carry = zb_or(cs, t1, t2);
    }
    return sum;
}
/* src/helpers/mod.rs:44-64: branches on the VALUE of the multiplier bits */
static zu8 helpers_multiply(zko_cs *cs, const zu8 *multiplicand, const zu8 *multiplier) {
    zu8 product = u8_const(0);
    for (int i = 0; i < 8; i++) {
        if (zb_val(cs, multiplier->b[i])) {
            zu8 addend = i ? u8_shl(multiplicand, i) : *multiplicand;
            product = helpers_add(cs, &product, &addend);
        }
    }
    return product;
}
/* src/aes_circuit.rs:360-427 */
static void gmix_column(zko_cs *cs, zu8 out[4], const zu8 in[4]) {
    zu8 b[4];
    zu8 c1 = u8_const(1), c1b = u8_const(0x1B);
    for (int k = 0; k < 4; k++) {
        zu8 sh = u8_shr(&in[k], 7), h;
        for (int i = 0; i < 8; i++) h.b[i] = zb_and(cs, sh.b[i], c1.b[i]);
        zu8 partial = u8_shl(&in[k], 1);
        zu8 m = helpers_multiply(cs, &h, &c1b);
        b[k] = u8_xor(cs, &partial, &m);
    }
    static const int order[4][5][2] = {  /* {is_b, index}: left-assoc xor chains of :391-426 */
        {{1, 0}, {0, 3}, {0, 2}, {1, 1}, {0, 1}}, {{1, 1}, {0, 0}, {0, 3}, {1, 2}, {0, 2}},
        {{1, 2}, {0, 1}, {0, 0}, {1, 3}, {0, 3}}, {{1, 3}, {0, 2}, {0, 1}, {1, 0}, {0, 0}}};
    for (int o = 0; o < 4; o++) {
        zu8 acc = order[o][0][0] ? b[order[o][0][1]] : in[order[o][0][1]];
        for (int t = 1; t < 5; t++) { const zu8 *x = order[o][t][0] ? &b[order[o][t][1]] : &in[order[o][t][1]]; acc = u8_xor(cs, &acc, x); }
        out[o] = acc;
    }
}
static void mix_columns(zko_cs *cs, zu8 out[16], const zu8 in[16]) { for (int c = 0; c < 4; c++) gmix_column(cs, out + 4 * c, in + 4 * c); }

/* src/aes_circuit.rs:20-129 */
static void derive_keys(zko_cs *cs, zu8 rk[11][16], const zu8 key[16], const zu8 *table) {
    static const uint8_t rc[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36};
    zu32 w[44];
    for (int i = 0; i < 4; i++) w[i] = to_u32(key + 4 * i);
    for (int i = 4; i < 44; i++) {
        if (i % 4 == 0) {
            zu8 by[4], rot[4], sub[4];
            to_bytes_be(by, &w[i - 1]);
            for (int k = 0; k < 4; k++) rot[k] = by[(k + 1) % 4];                /* rotate_word: rotate_left(1) */
            for (int k = 0; k < 4; k++) sub[k] = substitute_byte(cs, &rot[k], table);
            zu32 sr = to_u32(sub);
            zu32 res = u32_xor(cs, &w[i - 4], &sr);
            zu32 rcon = u32_const((uint32_t)rc[i / 4 - 1] << 24);
            w[i] = u32_xor(cs, &res, &rcon);
        } else {
            w[i] = u32_xor(cs, &w[i - 4], &w[i - 1]);
        }
    }
    for (int r = 0; r < 11; r++) for (int k = 0; k < 4; k++) to_bytes_be(&rk[r][4 * k], &w[4 * r + k]);
}

int zko_synth_aes(zko_cs *cs, const uint8_t *msg, size_t len, const uint8_t key[16], uint8_t *ct_out) {
    if (len % 16) return -1;      /* "Input must be 16 bytes length when adding round key" (src/aes_circuit.rs:218-221) */
    zu8 *m = malloc((len ? len : 1) * sizeof(zu8)), *ct = malloc((len ? len : 1) * sizeof(zu8));
    zu8 k[16], table[256];
    for (size_t i = 0; i < len; i++) m[i] = u8_alloc(cs, msg[i], 0);               /* src/lib.rs:70-76 */
    for (int i = 0; i < 16; i++) k[i] = u8_alloc(cs, key[i], 0);                   /* :82-88 */
    for (int i = 0; i < 256; i++) table[i] = u8_const(zko_aes_substitute_byte((uint8_t)i));  /* lookup_table: src/aes_circuit.rs:433-694 */
    static zu8 rk[11][16];
    zu8 (*rkp)[16] = malloc(sizeof(zu8) * 11 * 16);
    (void)rk;
    derive_keys(cs, rkp, k, table);                                               /* :187 */
    for (size_t off = 0; off < len; off += 16) {                                   /* :194 */
        zu8 s[16], t[16];
        add_round_key(cs, s, m + off, k);                                          /* :196 raw key */
        for (int r = 1; r <= 9; r++) {
            substitute_bytes(cs, t, s, table); shift_rows(s, t); mix_columns(cs, t, s);
            add_round_key(cs, s, t, rkp[r]);
        }
        substitute_bytes(cs, t, s, table); shift_rows(s, t);
        add_round_key(cs, ct + off, s, rkp[10]);
    }
    for (size_t i = 0; i < len; i++) {                                             /* :282-286 */
        uint8_t v = u8_val(cs, &ct[i]);
        if (ct_out) ct_out[i] = v;
        zu8 pi = u8_alloc(cs, v, 1);
        for (int b = 0; b < 8; b++) zb_enforce_equal(cs, pi.b[b], ct[i].b[b]);
    }
    free(m); free(ct); free(rkp);
    return 0;
}

/* ------------------------------------------------------------------ src/ops.rs toy gates */
static zu32 u32_alloc(zko_cs *cs, uint32_t v) { zu32 r; for (int i = 0; i < 32; i++) r.b[i] = zb_alloc(cs, (v >> i) & 1, 0); return r; }
static uint32_t u32_val(const zko_cs *cs, const zu32 *a) { uint32_t v = 0; for (int i = 0; i < 32; i++) v |= (uint32_t)zb_val(cs, a->b[i]) << i; return v; }
uint32_t zko_synth_ops_xor(zko_cs *cs, uint32_t x, uint32_t y) {
    zu32 a = u32_alloc(cs, x), b = u32_alloc(cs, y);
    zu32 r = u32_xor(cs, &a, &b);
    return u32_val(cs, &r);
}
/* UInt32::addmany [ark-r1cs-std 0.3.1 bits/uint.rs]: result bits for max_value = n * u32::MAX, one 0*0 = lc row */
uint32_t zko_synth_ops_add(zko_cs *cs, uint32_t x, uint32_t y) {
    zu32 ops[2] = {u32_alloc(cs, x), u32_alloc(cs, y)};
    unsigned __int128 max_value = (unsigned __int128)2 * 0xffffffffu;
    uint64_t result_value = (uint64_t)x + y;
    lc_t lc; lc_zero(&lc);
    for (int o = 0; o < 2; o++) { int64_t coeff = 1; for (int i = 0; i < 32; i++) { lc_add_bool(&lc, ops[o].b[i], coeff); coeff *= 2; } }
    zu32 res; int i = 0; int64_t coeff = 1;
    while (max_value != 0) {
        zb b = zb_alloc(cs, (int)((result_value >> i) & 1), 0);
        lc_add(&lc, -coeff, b.var);
        if (i < 32) res.b[i] = b;
        max_value >>= 1; i++; coeff *= 2;
    }
    lc_t z; lc_zero(&z);
    cs_enforce(cs, &z, &z, &lc);
    return u32_val(cs, &res);
}
