// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
// The draws behind Rep3Rand (`/root/reference/mpc-core/src/protocols/rep3/rngs.rs:25-46`): `RngType = rand_chacha::ChaCha12Rng`
// (`mpc-core/src/lib.rs:10`), `F::rand(&mut rng)` per element.  Both algorithms live in third-party crates that are NOT under
// /root/reference (crates.io, not vendored): rand_chacha 0.3 (ChaCha with 12 rounds, 64-bit block counter in state words 12-13, 64-bit
// stream id 0 in words 14-15, output = successive blocks as little-endian u32 words; `next_u64` = two consecutive words, low first;
// `get_word_pos` / `set_word_pos` address the stream by 32-bit word) and ark-ff 0.4.2 (`impl Distribution<Fp<P, N>> for Standard`:
// loop { N x next_u64 into the limbs, low limb first; clear the top `64 N - MODULUS_BIT_SIZE` bits of the last limb; accept if below the
// modulus } — the accepted bits ARE the element's Montgomery representation).  Restated from their published algorithms.
// PARITY UNPINNED for the draw order (no reference test holds a drawn value); the block function is pinned to the published ChaCha
// known-answer vectors (RFC 7539 2.3.2 for 20 rounds, the all-zero-key vectors for 20 and 12 rounds) in tests/test_chacha_rand.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

// one 64-byte block: key = 8 words, counter and stream id 64 bits each
inline void chacha_block(int rounds, const uint32_t key[8], uint64_t counter, uint64_t stream, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = (uint32_t)stream; s[15] = (uint32_t)(stream >> 32);
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < rounds; r += 2) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

// ChaCha12Rng::from_seed(seed) positioned at a word: what set_word_pos / get_word_pos see
struct ChaCha12Stream {
    uint32_t key[8]; uint64_t word_pos = 0;
    uint32_t blk[16]; uint64_t blk_index = ~0ull;
    explicit ChaCha12Stream(const uint8_t seed[32], uint64_t pos = 0) : word_pos(pos) {
        for (int i = 0; i < 8; i++) key[i] = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 | (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24;
    }
    uint32_t next_u32() {
        const uint64_t b = word_pos >> 4;
        if (b != blk_index) { chacha_block(12, key, b, 0, blk); blk_index = b; }
        return blk[word_pos++ & 15];
    }
    uint64_t next_u64() { const uint64_t lo = next_u32(); return lo | (uint64_t)next_u32() << 32; }
};

// F::rand for a 4-limb scalar field: modulus as 4 x u64, `bits` = MODULUS_BIT_SIZE
inline void fr_rand(ChaCha12Stream& rng, const uint64_t mod[4], int bits, uint64_t out[4]) {
    const int shave = 256 - bits;
    for (;;) {
        for (int i = 0; i < 4; i++) out[i] = rng.next_u64();
        out[3] &= ~0ull >> shave;
        bool below = false;
        for (int i = 3; i >= 0; i--) { if (out[i] != mod[i]) { below = out[i] < mod[i]; break; } }
        if (below) return;
    }
}

}  // namespace orc
