/* oracle/zko_aes.c -- TEST INFRASTRUCTURE (CPU oracle).
 * Byte-level AES-128 restating the reference's primitive implementation, /root/reference/src/aes.rs:
 *   add_round_key :10-18, substitute_byte :24-64 (algebraic S-box: p*=3, q/=3, affine transform),
 *   shift_rows :95-155 (column-major state, row r rotated left by r), gmix_column/mix_columns :157-199,
 *   derive_keys :206-247 (+ to_u32 :249-257 big-endian, rotate_word :259-269).
 * Pinned by the FIPS-197 vectors of tests/integration_tests.rs:52-64,67-276 and src/aes.rs:276-363.
 */
#include "zko.h"

void zko_aes_add_round_key(uint8_t out[16], const uint8_t in[16], const uint8_t key[16]) {
    for (int i = 0; i < 16; i++) out[i] = in[i] ^ key[i];
}
static uint8_t rotl8(uint8_t b, int n) { return (uint8_t)((b << n) | (b >> (8 - n))); }
uint8_t zko_aes_substitute_byte(uint8_t byte) {
    if (byte == 0) return 0x63;
    uint8_t p = 1, q = 1, sbox[256];
    memset(sbox, 0, sizeof sbox);
    do {
        p = (uint8_t)(p ^ (p << 1) ^ (((p >> 7) & 1) * 0x1B));      /* multiply p by 3 */
        q ^= (uint8_t)(q << 1); q ^= (uint8_t)(q << 2); q ^= (uint8_t)(q << 4);
        q ^= (uint8_t)(((q >> 7) & 1) * 0x09);                        /* divide q by 3 */
        uint8_t x = q ^ rotl8(q, 1) ^ rotl8(q, 2) ^ rotl8(q, 3) ^ rotl8(q, 4);
        sbox[p] = x ^ 0x63;
    } while (p != 1);
    return sbox[byte];
}
void zko_aes_substitute_bytes(uint8_t out[16], const uint8_t in[16]) {
    for (int i = 0; i < 16; i++) out[i] = zko_aes_substitute_byte(in[i]);
}
void zko_aes_shift_rows(uint8_t out[16], const uint8_t in[16]) {
    uint8_t m[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m[i][j] = in[i + 4 * j];
    for (int r = 0; r < 4; r++) {                 /* rotate row r left by r */
        uint8_t t[4];
        for (int j = 0; j < 4; j++) t[j] = m[r][(j + r) % 4];
        memcpy(m[r], t, 4);
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[i * 4 + j] = m[j][i];
}
static void gmix_column(uint8_t o[4], const uint8_t a[4]) {
    uint8_t b[4];
    for (int i = 0; i < 4; i++) {
        uint8_t h = (a[i] >> 7) & 1;
        b[i] = (uint8_t)((a[i] << 1) ^ (h * 0x1B));
    }
    o[0] = b[0] ^ a[3] ^ a[2] ^ b[1] ^ a[1];
    o[1] = b[1] ^ a[0] ^ a[3] ^ b[2] ^ a[2];
    o[2] = b[2] ^ a[1] ^ a[0] ^ b[3] ^ a[3];
    o[3] = b[3] ^ a[2] ^ a[1] ^ b[0] ^ a[0];
}
void zko_aes_mix_columns(uint8_t out[16], const uint8_t in[16]) {
    for (int c = 0; c < 4; c++) gmix_column(out + 4 * c, in + 4 * c);
}
void zko_aes_derive_keys(uint8_t out[11][16], const uint8_t key[16]) {
    static const uint8_t rc[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36};
    uint32_t w[44];
    for (int i = 0; i < 4; i++) w[i] = ((uint32_t)key[4 * i] << 24) | ((uint32_t)key[4 * i + 1] << 16) | ((uint32_t)key[4 * i + 2] << 8) | key[4 * i + 3];
    for (int i = 4; i < 44; i++) {
        if (i % 4 == 0) {
            uint32_t x = w[i - 1];
            uint8_t by[4] = {(uint8_t)(x >> 16), (uint8_t)(x >> 8), (uint8_t)x, (uint8_t)(x >> 24)};  /* rotate_word */
            uint32_t s = 0;
            for (int k = 0; k < 4; k++) s = (s << 8) | zko_aes_substitute_byte(by[k]);
            w[i] = w[i - 4] ^ s ^ ((uint32_t)rc[i / 4 - 1] << 24);
        } else {
            w[i] = w[i - 4] ^ w[i - 1];
        }
    }
    for (int r = 0; r < 11; r++) for (int k = 0; k < 4; k++) {
        uint32_t x = w[4 * r + k];
        out[r][4 * k] = x >> 24; out[r][4 * k + 1] = x >> 16; out[r][4 * k + 2] = x >> 8; out[r][4 * k + 3] = x;
    }
}
/* round structure of src/lib.rs:194-278 on plain bytes (round 0 uses the raw key, :196) */
void zko_aes_encrypt_ecb(uint8_t *out, const uint8_t *msg, size_t len, const uint8_t key[16]) {
    uint8_t rk[11][16], s[16], t[16];
    zko_aes_derive_keys(rk, key);
    for (size_t off = 0; off + 16 <= len; off += 16) {
        zko_aes_add_round_key(s, msg + off, key);
        for (int r = 1; r <= 9; r++) {
            zko_aes_substitute_bytes(t, s); zko_aes_shift_rows(s, t); zko_aes_mix_columns(t, s);
            zko_aes_add_round_key(s, t, rk[r]);
        }
        zko_aes_substitute_bytes(t, s); zko_aes_shift_rows(s, t);
        zko_aes_add_round_key(out + off, s, rk[10]);
    }
}
