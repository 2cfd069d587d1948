/* oracle/zko_fs.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Blake2s-256 (RFC 7693; blake2 0.9.2, Cargo.lock:414), the ChaCha block RNG of rand_chacha 0.3.1
 * (Cargo.lock:1303: ChaChaRng = 20 rounds, rand::StdRng = 12 rounds, 64-bit block counter, 4-block
 * buffer, next_u64 = two consecutive words) and ark-marlin's SimpleHashFiatShamirRng
 * (seed' = Blake2s(input || seed); rng = ChaCha20(seed')) -- SURVEY.md §A.4.  None of these sources
 * are under /root/reference; pinned here by RFC test vectors (tests/test_oracle_primitives.py).
 */
#include "zko.h"

static const uint32_t B2S_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static void b2s_compress(uint32_t h[8], const uint8_t block[64], uint64_t t, int last) {
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; i++) m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) | ((uint32_t)block[4 * i + 2] << 16) | ((uint32_t)block[4 * i + 3] << 24);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2S_IV[i]; }
    v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define G(a, b, c, d, x, y) \
    v[a] = v[a] + v[b] + (x); v[d] = rotr32(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr32(v[b] ^ v[c], 12); \
    v[a] = v[a] + v[b] + (y); v[d] = rotr32(v[d] ^ v[a], 8);  v[c] = v[c] + v[d]; v[b] = rotr32(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; r++) {
        const uint8_t *s = B2S_SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]) G(1, 5, 9, 13, m[s[2]], m[s[3]]) G(2, 6, 10, 14, m[s[4]], m[s[5]]) G(3, 7, 11, 15, m[s[6]], m[s[7]])
        G(0, 5, 10, 15, m[s[8]], m[s[9]]) G(1, 6, 11, 12, m[s[10]], m[s[11]]) G(2, 7, 8, 13, m[s[12]], m[s[13]]) G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
void zko_blake2s(uint8_t out[32], const uint8_t *in, size_t len) {
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = B2S_IV[i];
    h[0] ^= 0x01010020u; /* digest 32 bytes, no key, fanout 1, depth 1 */
    uint64_t t = 0;
    uint8_t block[64];
    while (len > 64) { t += 64; b2s_compress(h, in, t, 0); in += 64; len -= 64; }
    memset(block, 0, 64);
    if (len) memcpy(block, in, len);
    t += len;
    b2s_compress(h, block, t, 1);
    for (int i = 0; i < 8; i++) { out[4 * i] = h[i]; out[4 * i + 1] = h[i] >> 8; out[4 * i + 2] = h[i] >> 16; out[4 * i + 3] = h[i] >> 24; }
}

static void chacha_block(uint32_t out[16], const uint32_t key[8], uint64_t counter, int rounds) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), 0, 0};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
#define QR(a, b, c, d) \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    for (int r = 0; r < rounds; r += 2) {
        QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
        QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
    }
#undef QR
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
static void chacha_refill(zko_chacha *r) {
    for (int b = 0; b < 4; b++) chacha_block(r->buf + 16 * b, r->key, r->counter + b, r->rounds);
    r->counter += 4;
    r->idx = 0;
}
void zko_chacha_init(zko_chacha *r, const uint8_t seed[32], int rounds) {
    for (int i = 0; i < 8; i++) r->key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    r->counter = 0; r->rounds = rounds; r->idx = 64; /* empty: first use refills */
}
uint32_t zko_chacha_u32(zko_chacha *r) {
    if (r->idx >= 64) chacha_refill(r);
    return r->buf[r->idx++];
}
/* rand_core BlockRng::next_u64 */
uint64_t zko_chacha_u64(zko_chacha *r) {
    if (r->idx < 63) {
        uint64_t lo = r->buf[r->idx], hi = r->buf[r->idx + 1];
        r->idx += 2;
        return (hi << 32) | lo;
    } else if (r->idx >= 64) {
        chacha_refill(r);
        uint64_t lo = r->buf[0], hi = r->buf[1];
        r->idx = 2;
        return (hi << 32) | lo;
    } else {
        uint64_t lo = r->buf[63];
        chacha_refill(r);
        uint64_t hi = r->buf[0];
        r->idx = 1;
        return (hi << 32) | lo;
    }
}
/* ark-ff 0.3.0 `impl Distribution<Fp256<P>> for Standard`: 4 x next_u64, shave the top bits, reject if
 * >= modulus, and use the limbs AS the Montgomery representation. */
void zko_fr_rand(fr_t *out, zko_chacha *rng, const fr_params *F) {
    int shave = 256 - F->bits;
    for (;;) {
        for (int i = 0; i < 4; i++) out->l[i] = zko_chacha_u64(rng);
        out->l[3] &= (~(uint64_t)0) >> shave;
        if (!fr_geq_raw(out->l, F->p)) return;
    }
}
void zko_fs_init(zko_fsrng *fs, const uint8_t *bytes, size_t len) {
    zko_blake2s(fs->seed, bytes, len);
    zko_chacha_init(&fs->r, fs->seed, 20);
}
#include <stdlib.h>
void zko_fs_absorb(zko_fsrng *fs, const uint8_t *bytes, size_t len) {
    uint8_t *buf = malloc(len + 32);
    memcpy(buf, bytes, len);
    memcpy(buf + len, fs->seed, 32);
    zko_blake2s(fs->seed, buf, len + 32);
    free(buf);
    zko_chacha_init(&fs->r, fs->seed, 20);
}
