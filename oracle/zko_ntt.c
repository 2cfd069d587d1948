/* oracle/zko_ntt.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Radix-2 evaluation domains over Fr restating ark-poly 0.3.0 Radix2EvaluationDomain
 * (Cargo.lock:234; not under /root/reference): group_gen = root^(2^(s - log n)), in-order FFT/IFFT,
 * coset = multiplicative generator.
 */
#include "zko.h"
#include "zko_consts.h"
#include <stdlib.h>

int zko_domain_init(zko_domain *D, uint64_t min_size, const fr_params *F) {
    int lg = 0;
    while (((uint64_t)1 << lg) < min_size) lg++;
    int two_adicity = (F == &ZKO_FR381) ? FR381_TWO_ADICITY : FR377_TWO_ADICITY;
    if (lg > two_adicity) return -1;
    D->F = F; D->size = (uint64_t)1 << lg; D->log_size = lg;
    fr_t root;
    memcpy(root.l, (F == &ZKO_FR381) ? FR381_ROOT_MONT : FR377_ROOT_MONT, 32);
    memcpy(D->coset_gen.l, (F == &ZKO_FR381) ? FR381_GEN_MONT : FR377_GEN_MONT, 32);
    for (int i = lg; i < two_adicity; i++) fr_sqr(&root, &root, F);
    D->gen = root;
    fr_inv(&D->gen_inv, &root, F);
    fr_from_u64(&D->size_as_fe, D->size, F);
    fr_inv(&D->size_inv, &D->size_as_fe, F);
    return 0;
}
static inline uint64_t bitrev(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static void fft_core(fr_t *a, uint64_t n, int lg, const fr_t *w, const fr_params *F) {
    for (uint64_t i = 0; i < n; i++) {
        uint64_t j = bitrev(i, lg);
        if (i < j) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    /* twiddle table w^0..w^(n/2-1) */
    fr_t *tw = malloc((n / 2 ? n / 2 : 1) * sizeof(fr_t));
    fr_set_one(&tw[0], F);
    for (uint64_t i = 1; i < n / 2; i++) fr_mul(&tw[i], &tw[i - 1], w, F);
    for (uint64_t len = 2; len <= n; len <<= 1) {
        uint64_t half = len / 2, step = n / len;
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (uint64_t k = 0; k < n / 2; k++) {
            uint64_t blk = k / half, j = k % half;
            fr_t *lo = &a[blk * len + j], *hi = lo + half, t;
            fr_mul(&t, hi, &tw[j * step], F);
            fr_sub(hi, lo, &t, F);
            fr_add(lo, lo, &t, F);
        }
    }
    free(tw);
}
void zko_fft(const zko_domain *D, fr_t *a) { fft_core(a, D->size, D->log_size, &D->gen, D->F); }
void zko_ifft(const zko_domain *D, fr_t *a) {
    fft_core(a, D->size, D->log_size, &D->gen_inv, D->F);
#pragma omp parallel for schedule(static) if (D->size >= 4096)
    for (uint64_t i = 0; i < D->size; i++) fr_mul(&a[i], &a[i], &D->size_inv, D->F);
}
static void distribute_powers(fr_t *a, uint64_t n, const fr_t *g, const fr_params *F) {
    fr_t p; fr_set_one(&p, F);
    for (uint64_t i = 0; i < n; i++) { fr_mul(&a[i], &a[i], &p, F); fr_mul(&p, &p, g, F); }
}
void zko_coset_fft(const zko_domain *D, fr_t *a) { distribute_powers(a, D->size, &D->coset_gen, D->F); zko_fft(D, a); }
void zko_coset_ifft(const zko_domain *D, fr_t *a) {
    fr_t gi; fr_inv(&gi, &D->coset_gen, D->F);
    zko_ifft(D, a); distribute_powers(a, D->size, &gi, D->F);
}
void zko_domain_element(fr_t *r, const zko_domain *D, uint64_t i) { fr_pow_u64(r, &D->gen, i, D->F); }
void zko_domain_eval_vanishing(fr_t *r, const zko_domain *D, const fr_t *x) {
    fr_t one; fr_set_one(&one, D->F);
    fr_pow_u64(r, x, D->size, D->F);
    fr_sub(r, r, &one, D->F);
}
/* ark-poly EvaluationDomain::reindex_by_subdomain */
size_t zko_reindex_by_subdomain(const zko_domain *self, const zko_domain *other, size_t index) {
    size_t period = self->size / other->size;
    if (index < other->size) return index * period;
    size_t i = index - other->size, x = period - 1;
    return i + (i / x) + 1;
}
void zko_poly_eval(fr_t *r, const fr_t *c, size_t n, const fr_t *x, const fr_params *F) {
    fr_t acc; fr_set_zero(&acc);
    for (size_t i = n; i-- > 0;) { fr_mul(&acc, &acc, x, F); fr_add(&acc, &acc, &c[i], F); }
    *r = acc;
}
