// Host side of Rep3Rand's generators (rep3/rngs.rs:25-46): rand_chacha's ChaCha12Rng (12 rounds, 64-bit block counter in state words 12-13,
// stream id 0, output read as little-endian 32-bit words, next_u64 = two consecutive words, addressable by word: get_word_pos /
// set_word_pos) and ark-ff's `F::rand` (four next_u64 into the limbs, low first; the top 256 - MODULUS_BIT_SIZE bits cleared; redrawn until
// below the modulus; the bits are the Montgomery representation).  Used for the O(1) draws and short vectors; the m-element masking
// vectors are drawn by the backend's kernels (csrc/chacha_rand.hip) from the same (seed, position) pairs.
#pragma once
#include <cstdint>
#include <cstring>

namespace cgh {

struct ChaCha12 {
    uint32_t key[8]; uint64_t word_pos = 0;
    uint32_t blk[16]; uint64_t blk_index = ~0ull;
    ChaCha12() { memset(key, 0, sizeof key); }
    explicit ChaCha12(const uint8_t seed[32], uint64_t pos = 0) : word_pos(pos) { memcpy(key, seed, 32); }      // little-endian host
    static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    void refill(uint64_t counter) {
        const uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                                (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
        uint32_t x[16];
        memcpy(x, s, sizeof x);
#define CGH_QR(a, b, c, d) x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12); x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        for (int r = 0; r < 6; r++) {
            CGH_QR(0, 4, 8, 12) CGH_QR(1, 5, 9, 13) CGH_QR(2, 6, 10, 14) CGH_QR(3, 7, 11, 15)
            CGH_QR(0, 5, 10, 15) CGH_QR(1, 6, 11, 12) CGH_QR(2, 7, 8, 13) CGH_QR(3, 4, 9, 14)
        }
#undef CGH_QR
        for (int i = 0; i < 16; i++) blk[i] = x[i] + s[i];
        blk_index = counter;
    }
    uint32_t next_u32() { const uint64_t b = word_pos >> 4; if (b != blk_index) refill(b); return blk[word_pos++ & 15]; }
    uint64_t next_u64() { const uint64_t lo = next_u32(); return lo | (uint64_t)next_u32() << 32; }
    // F::rand for a four-limb scalar field
    void fr_rand(const uint64_t mod[4], int bits, uint64_t out[4]) {
        for (;;) {
            for (int i = 0; i < 4; i++) out[i] = next_u64();
            out[3] &= ~0ull >> (256 - bits);
            for (int i = 3; i >= 0; i--) { if (out[i] != mod[i]) { if (out[i] < mod[i]) return; break; } }
        }
    }
};

}  // namespace cgh
