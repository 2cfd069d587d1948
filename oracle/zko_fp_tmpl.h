/* oracle/zko_fp_tmpl.h -- TEST INFRASTRUCTURE (CPU oracle), never linked into the product.
 *
 * Montgomery prime-field arithmetic on 64-bit limbs (CIOS, unsigned __int128), instantiated by
 * including this file with FP_N (limb count) and FP_(name) (symbol prefix) defined.
 * Restates the arithmetic of ark-ff 0.3.0 `Fp256`/`Fp384` (Cargo.lock:159; source not under
 * /root/reference): elements are kept in Montgomery form, `rand` interprets masked raw limbs as
 * the Montgomery representation.
 */
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t l[FP_N]; } FP_(t);
typedef struct {
    uint64_t p[FP_N];    /* modulus */
    uint64_t one[FP_N];  /* R mod p */
    uint64_t r2[FP_N];   /* R^2 mod p */
    uint64_t inv;        /* -p^-1 mod 2^64 */
    int bits;            /* modulus bit length */
} FP_(params);

static inline int FP_(is_zero)(const FP_(t) *a) {
    uint64_t o = 0;
    for (int i = 0; i < FP_N; i++) o |= a->l[i];
    return o == 0;
}
static inline int FP_(eq)(const FP_(t) *a, const FP_(t) *b) {
    uint64_t o = 0;
    for (int i = 0; i < FP_N; i++) o |= a->l[i] ^ b->l[i];
    return o == 0;
}
static inline int FP_(geq_raw)(const uint64_t *a, const uint64_t *b) {
    for (int i = FP_N - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline void FP_(sub_raw)(uint64_t *r, const uint64_t *a, const uint64_t *b) {
    unsigned __int128 br = 0;
    for (int i = 0; i < FP_N; i++) {
        unsigned __int128 d = (unsigned __int128)a[i] - b[i] - (uint64_t)br;
        r[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline void FP_(add)(FP_(t) *r, const FP_(t) *a, const FP_(t) *b, const FP_(params) *P) {
    unsigned __int128 c = 0;
    uint64_t t[FP_N];
    for (int i = 0; i < FP_N; i++) {
        c += (unsigned __int128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || FP_(geq_raw)(t, P->p)) FP_(sub_raw)(t, t, P->p);
    memcpy(r->l, t, sizeof t);
}
static inline void FP_(sub)(FP_(t) *r, const FP_(t) *a, const FP_(t) *b, const FP_(params) *P) {
    uint64_t t[FP_N];
    if (FP_(geq_raw)(a->l, b->l)) {
        FP_(sub_raw)(t, a->l, b->l);
    } else {
        uint64_t u[FP_N];
        FP_(sub_raw)(u, P->p, b->l);
        unsigned __int128 c = 0;
        for (int i = 0; i < FP_N; i++) {
            c += (unsigned __int128)a->l[i] + u[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, sizeof t);
}
static inline void FP_(neg)(FP_(t) *r, const FP_(t) *a, const FP_(params) *P) {
    if (FP_(is_zero)(a)) { *r = *a; return; }
    FP_(sub_raw)(r->l, P->p, a->l);
}
static inline void FP_(dbl)(FP_(t) *r, const FP_(t) *a, const FP_(params) *P) { FP_(add)(r, a, a, P); }

static inline void FP_(mul)(FP_(t) *r, const FP_(t) *a, const FP_(t) *b, const FP_(params) *P) {
    uint64_t t[FP_N + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < FP_N; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < FP_N; j++) {
            c += (unsigned __int128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[FP_N];
        t[FP_N] = (uint64_t)c;
        t[FP_N + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * P->inv;
        c = (unsigned __int128)m * P->p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < FP_N; j++) {
            c += (unsigned __int128)m * P->p[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[FP_N];
        t[FP_N - 1] = (uint64_t)c;
        t[FP_N] = t[FP_N + 1] + (uint64_t)(c >> 64);
    }
    if (t[FP_N] || FP_(geq_raw)(t, P->p)) FP_(sub_raw)(t, t, P->p);
    memcpy(r->l, t, FP_N * sizeof(uint64_t));
}
static inline void FP_(sqr)(FP_(t) *r, const FP_(t) *a, const FP_(params) *P) { FP_(mul)(r, a, a, P); }

static inline void FP_(set_one)(FP_(t) *r, const FP_(params) *P) { memcpy(r->l, P->one, sizeof r->l); }
static inline void FP_(set_zero)(FP_(t) *r) { memset(r->l, 0, sizeof r->l); }
/* canonical integer (little-endian limbs) -> Montgomery */
static inline void FP_(from_raw)(FP_(t) *r, const uint64_t *raw, const FP_(params) *P) {
    FP_(t) a, r2;
    memcpy(a.l, raw, sizeof a.l);
    memcpy(r2.l, P->r2, sizeof r2.l);
    FP_(mul)(r, &a, &r2, P);
}
/* Montgomery -> canonical integer */
static inline void FP_(to_raw)(uint64_t *raw, const FP_(t) *a, const FP_(params) *P) {
    FP_(t) one, o;
    memset(one.l, 0, sizeof one.l);
    one.l[0] = 1;
    FP_(mul)(&o, a, &one, P);
    memcpy(raw, o.l, sizeof o.l);
}
static inline void FP_(from_u64)(FP_(t) *r, uint64_t v, const FP_(params) *P) {
    uint64_t raw[FP_N];
    memset(raw, 0, sizeof raw);
    raw[0] = v;
    FP_(from_raw)(r, raw, P);
}
static inline void FP_(from_i64)(FP_(t) *r, int64_t v, const FP_(params) *P) {
    if (v >= 0) { FP_(from_u64)(r, (uint64_t)v, P); }
    else { FP_(from_u64)(r, (uint64_t)(-v), P); FP_(neg)(r, r, P); }
}
/* r = a^e, e given as little-endian 64-bit limbs */
static inline void FP_(pow)(FP_(t) *r, const FP_(t) *a, const uint64_t *e, int elimbs, const FP_(params) *P) {
    FP_(t) acc, base = *a;
    FP_(set_one)(&acc, P);
    int top = elimbs * 64 - 1;
    while (top >= 0 && !((e[top / 64] >> (top % 64)) & 1)) top--;
    for (int i = top; i >= 0; i--) {
        FP_(sqr)(&acc, &acc, P);
        if ((e[i / 64] >> (i % 64)) & 1) FP_(mul)(&acc, &acc, &base, P);
    }
    *r = acc;
}
static inline void FP_(pow_u64)(FP_(t) *r, const FP_(t) *a, uint64_t e, const FP_(params) *P) { FP_(pow)(r, a, &e, 1, P); }
/* r = a^-1 (Fermat); inverse of zero is zero */
static inline void FP_(inv)(FP_(t) *r, const FP_(t) *a, const FP_(params) *P) {
    uint64_t e[FP_N];
    memcpy(e, P->p, sizeof e);
    e[0] -= 2; /* moduli here are odd and > 2, no borrow */
    FP_(pow)(r, a, e, FP_N, P);
}
/* in-place batch inversion (Montgomery's trick); zeros are left as zero (ark-ff batch_inversion) */
static inline void FP_(batch_inv)(FP_(t) *v, size_t n, FP_(t) *scratch, const FP_(params) *P) {
    FP_(t) acc;
    FP_(set_one)(&acc, P);
    for (size_t i = 0; i < n; i++) {
        scratch[i] = acc;
        if (!FP_(is_zero)(&v[i])) FP_(mul)(&acc, &acc, &v[i], P);
    }
    FP_(inv)(&acc, &acc, P);
    for (size_t i = n; i-- > 0;) {
        if (FP_(is_zero)(&v[i])) continue;
        FP_(t) t;
        FP_(mul)(&t, &acc, &scratch[i], P);
        FP_(mul)(&acc, &acc, &v[i], P);
        v[i] = t;
    }
}
