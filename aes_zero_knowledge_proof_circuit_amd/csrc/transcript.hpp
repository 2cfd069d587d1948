// csrc/transcript.hpp -- host-side Fiat-Shamir transcript and prover/setup RNGs of libzkaes.
//
// Byte-exact roles (sources are upstream crates, not under /root/reference):
//   * Blake2s-256                        (blake2 0.9.2, Cargo.lock:414; RFC 7693)
//   * ChaCha block RNG, 64-word buffer   (rand_chacha 0.3.1, Cargo.lock:1303): ChaChaRng = 20 rounds for the
//     transcript, rand::StdRng = 12 rounds for `simpleworks::marlin::generate_rand()` = ark_std::test_rng()
//     (call sites /root/reference/src/lib.rs:65,134,139)
//   * SimpleHashFiatShamirRng<Blake2s, ChaChaRng>: seed <- Blake2s(input || seed); rng <- ChaCha20(seed)
// These run on the host: they are a few KB per proof and strictly sequential.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "ff.cuh"

namespace zk {

struct Blake2s {
    static void digest(uint8_t out[32], const uint8_t *in, size_t len) {
        static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        uint32_t h[8];
        for (int i = 0; i < 8; i++) h[i] = IV[i];
        h[0] ^= 0x01010020u;
        uint64_t t = 0;
        while (len > 64) { t += 64; compress(h, in, t, false, IV); in += 64; len -= 64; }
        uint8_t last[64] = {0};
        if (len) memcpy(last, in, len);
        t += len;
        compress(h, last, t, true, IV);
        for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(h[i] >> (8 * k));
    }

  private:
    static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    static void compress(uint32_t h[8], const uint8_t *blk, uint64_t t, bool last, const uint32_t IV[8]) {
        static const uint8_t S[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; i++) m[i] = (uint32_t)blk[4 * i] | (uint32_t)blk[4 * i + 1] << 8 | (uint32_t)blk[4 * i + 2] << 16 | (uint32_t)blk[4 * i + 3] << 24;
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
        v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
            v[a] += v[b] + x; v[d] = ror(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 12);
            v[a] += v[b] + y; v[d] = ror(v[d] ^ v[a], 8); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 7);
        };
        for (int r = 0; r < 10; r++) {
            const uint8_t *s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }
};

// rand_chacha's BlockRng<ChaChaXCore>: refills 4 blocks (64 words) at a time; 64-bit counter in words 12,13.
class ChaChaRng {
  public:
    ChaChaRng() : rounds_(20), counter_(0), idx_(64) { memset(key_, 0, sizeof key_); }
    ChaChaRng(const uint8_t seed[32], int rounds) { reseed(seed, rounds); }
    void reseed(const uint8_t seed[32], int rounds) {
        for (int i = 0; i < 8; i++) key_[i] = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 | (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24;
        rounds_ = rounds; counter_ = 0; idx_ = 64;
    }
    uint32_t next_u32() { if (idx_ >= 64) refill(); return buf_[idx_++]; }
    // position in the key stream, in 32-bit words (the next word next_u32 would return)
    uint64_t word_pos() const { return idx_ >= 64 ? counter_ * 16 : (counter_ - 4) * 16 + (uint64_t)idx_; }
    void set_word_pos(uint64_t pos) { counter_ = (pos / 64) * 4; refill(); idx_ = (int)(pos % 64); }
    const uint32_t *key_words() const { return key_; }
    int rounds() const { return rounds_; }
    uint64_t next_u64() {
        if (idx_ < 63) { uint64_t lo = buf_[idx_], hi = buf_[idx_ + 1]; idx_ += 2; return hi << 32 | lo; }
        if (idx_ >= 64) { refill(); idx_ = 2; return (uint64_t)buf_[1] << 32 | buf_[0]; }
        uint64_t lo = buf_[63]; refill(); idx_ = 1; return (uint64_t)buf_[0] << 32 | lo;
    }
    // ark-ff UniformRand for Fp: N/2 x next_u64, shave the unused top bits, reject >= p, limbs ARE the Montgomery form
    template <class F>
    F rand_field() {
        constexpr int shave = F::N * 32 - FieldBits<F>::value;
        F r;
        for (;;) {
            for (int i = 0; i < F::N; i += 2) { uint64_t v = next_u64(); r.l[i] = (uint32_t)v; r.l[i + 1] = (uint32_t)(v >> 32); }
            r.l[F::N - 1] &= 0xffffffffu >> shave;
            if (!F::geq_mod(r.l)) return r;
        }
    }
    template <class F> struct FieldBits;

  private:
    static uint32_t rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void block(uint32_t *out, uint64_t ctr) const {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key_[0], key_[1], key_[2], key_[3], key_[4], key_[5], key_[6], key_[7], (uint32_t)ctr, (uint32_t)(ctr >> 32), 0, 0};
        uint32_t x[16];
        memcpy(x, s, sizeof x);
        auto Q = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < rounds_; r += 2) { Q(0, 4, 8, 12); Q(1, 5, 9, 13); Q(2, 6, 10, 14); Q(3, 7, 11, 15); Q(0, 5, 10, 15); Q(1, 6, 11, 12); Q(2, 7, 8, 13); Q(3, 4, 9, 14); }
        for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
    }
    void refill() { for (int b = 0; b < 4; b++) block(buf_ + 16 * b, counter_ + b); counter_ += 4; idx_ = 0; }
    uint32_t key_[8];
    int rounds_;
    uint64_t counter_;
    uint32_t buf_[64];
    int idx_;
};
template <class P> struct ChaChaRng::FieldBits<Fp<P>> { static constexpr int value = P::BITS; };

// ark_std::test_rng() seed (simpleworks::marlin::generate_rand, [RECALL])
inline const uint8_t *ark_test_rng_seed() { return ARK_TEST_RNG_SEED; }

class FiatShamirRng {
  public:
    void initialize(const std::vector<uint8_t> &input) { Blake2s::digest(seed_, input.data(), input.size()); rng_.reseed(seed_, 20); }
    void absorb(const std::vector<uint8_t> &input) {
        std::vector<uint8_t> b(input);
        b.insert(b.end(), seed_, seed_ + 32);
        Blake2s::digest(seed_, b.data(), b.size());
        rng_.reseed(seed_, 20);
    }
    ChaChaRng &rng() { return rng_; }

  private:
    uint8_t seed_[32];
    ChaChaRng rng_;
};

}  // namespace zk
