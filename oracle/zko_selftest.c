/* oracle/zko_selftest.c -- TEST INFRASTRUCTURE: a small end-to-end exercise of the CPU oracle for the sanitizer build
 * (`make -C oracle selftest_asan`: -fsanitize=address,undefined; SURVEY.md section 5 "ASan/UBSan build of the CPU oracle").
 * It walks every layer once at small sizes: byte-level AES (FIPS-197 App. B vector, tests/integration_tests.rs:52-64), the gate-level circuit of one
 * block (satisfied; a flipped witness bit is not), NTT round trip, Pippenger MSM against double-and-add, and a complete Marlin index + proof of the
 * src/ops.rs xor gate.  Exit code 0 and "selftest ok" when everything agrees; any sanitizer report fails the run. */
#include "zko.h"
#include "zko_marlin.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

zko_cs *zko_api_synth_aes(int field_id, const uint8_t *msg, size_t len, const uint8_t *key, uint8_t *ct_out);
zko_cs *zko_api_synth_ops(int field_id, int which, uint32_t x, uint32_t y, uint32_t *out);
int zko_api_msm(int id, const uint8_t *bases, const fr_t *scalars, size_t n, uint8_t *out);
void zko_api_fixed_base(int id, const fr_t *scalars, size_t n, uint8_t *out);
int zko_api_ntt(int id, fr_t *data, size_t n, int kind);
int zko_api_g1_mul(int id, const uint8_t *xy, const fr_t *k, uint8_t *out);
int zko_api_g1_add(int id, const uint8_t *a, const uint8_t *b, uint8_t *out);
int zko_api_cs_set_witness(zko_cs *cs, size_t idx, int val);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(void) {
    static const uint8_t key[16] = {0x2b, 0x7e, 0x15, 0x16, 0x28, 0xae, 0xd2, 0xa6, 0xab, 0xf7, 0x15, 0x88, 0x09, 0xcf, 0x4f, 0x3c};
    static const uint8_t pt[16] = {0x32, 0x43, 0xf6, 0xa8, 0x88, 0x5a, 0x30, 0x8d, 0x31, 0x31, 0x98, 0xa2, 0xe0, 0x37, 0x07, 0x34};
    static const uint8_t ct_ref[16] = {0x39, 0x25, 0x84, 0x1d, 0x02, 0xdc, 0x09, 0xfb, 0xdc, 0x11, 0x85, 0x97, 0x19, 0x6a, 0x0b, 0x32};
    uint8_t ct[16];
    zko_aes_encrypt_ecb(ct, pt, 16, key);
    CHECK(memcmp(ct, ct_ref, 16) == 0);
    /* gate-level circuit of one block */
    uint8_t ct2[16];
    zko_cs *cs = zko_api_synth_aes(377, pt, 16, key, ct2);
    CHECK(cs != NULL && memcmp(ct2, ct_ref, 16) == 0);
    CHECK(zko_cs_is_satisfied(cs) == 1);
    CHECK(zko_api_cs_set_witness(cs, 1000, 1 - cs->witness_val[1000]) == 0);
    CHECK(zko_cs_is_satisfied(cs) < 0);
    zko_cs_free(cs);
    CHECK(zko_api_synth_aes(377, pt, 15, key, ct2) == NULL);                  /* ragged length is refused */
    /* NTT round trip, 377 and 381 */
    for (int id = 377; id <= 381; id += 4) {
        const fr_params *F = zko_fr_by_id(id);
        enum { N = 256 };
        fr_t a[N], b[N];
        for (int i = 0; i < N; i++) { fr_from_i64(&a[i], 3 * i + 1, F); b[i] = a[i]; }
        CHECK(zko_api_ntt(id, a, N, 0) == 0 && zko_api_ntt(id, a, N, 1) == 0);
        CHECK(memcmp(a, b, sizeof a) == 0);
    }
    /* MSM vs double-and-add: sum_i (i + 2) * (s_i G) */
    {
        enum { N = 70 };
        const fr_params *F = zko_fr_by_id(377);
        fr_t gen[N], sc[N];
        uint8_t bases[96 * N], got[96], acc[96], term[96];
        for (int i = 0; i < N; i++) { fr_from_i64(&gen[i], 1000 + 17 * i, F); fr_from_i64(&sc[i], i + 2, F); }
        zko_api_fixed_base(377, gen, N, bases);
        CHECK(zko_api_msm(377, bases, sc, N, got) == 0);
        int have = 0;
        for (int i = 0; i < N; i++) {
            CHECK(zko_api_g1_mul(377, bases + 96 * i, &sc[i], term) == 0);
            if (!have) { memcpy(acc, term, 96); have = 1; } else { uint8_t t[96]; CHECK(zko_api_g1_add(377, acc, term, t) == 0); memcpy(acc, t, 96); }
        }
        CHECK(memcmp(acc, got, 96) == 0);
    }
    /* Marlin: index + prove the xor gate of src/ops.rs over BLS12-377, serialize */
    {
        uint32_t out = 0;
        zko_cs *ics = zko_api_synth_ops(377, 0, 0, 0, &out);
        CHECK(ics != NULL);
        zko_index *ix = zko_marlin_index(ics, 200, 200, 600);
        CHECK(ix != NULL);
        zko_cs *pcs = zko_api_synth_ops(377, 0, 0xDEADBEEFu, 0x12345678u, &out);
        CHECK(pcs != NULL && out == (0xDEADBEEFu ^ 0x12345678u));
        zko_proof *pf = zko_marlin_prove(ix, pcs, NULL);
        CHECK(pf != NULL);
        uint8_t buf[2048];
        size_t n = zko_proof_serialize(pf, zko_curve_by_id(377), buf, sizeof buf);
        CHECK(n == 855);
        zko_proof_free(pf);
        zko_cs_free(pcs);
        zko_index_free(ix);
    }
    printf("selftest ok\n");
    return 0;
}
