// Rep3Rand's O(n) draws on the device: n x `F::rand(&mut ChaCha12Rng)` (mpc-core/src/protocols/rep3/rngs.rs:37-46 calls it once per
// stream and element; `RngType = rand_chacha::ChaCha12Rng`, mpc-core/src/lib.rs:10).  The reference draws them one after the other on one
// host thread — 4 x 2^22 rejection-sampled draws per 2^22-constraint proof, several times the GPU's whole prove.  A ChaCha stream is
// addressable by position, so every CANDIDATE draw is independent:
//   candidate k = the 8 stream words [word_pos + 8k, word_pos + 8k + 8) as four u64 limbs (low word first), top bits cleared (ark-ff 0.4.2
//   `Distribution<Fp> for Standard`), accepted when below the modulus; the i-th ACCEPTED candidate is the i-th draw.
// Three launches: k_chacha_candidates (one 64-byte ChaCha12 block per lane = two candidates, stored masked + per-tile acceptance counts),
// k_chacha_scan_tiles (exclusive scan of the tile counts, one workgroup), k_chacha_compact (rank inside the tile, ordered store of the
// first n accepted; the lane that holds draw n - 1 reports its candidate index: the caller's rng continues right behind it).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "common.hpp"

// (wave64 throughout: the compaction kernels rank with 64-lane ballots and shuffles; cg_ctx_create refuses a device whose wavefront is not 64 lanes)
namespace cg {

struct ChaChaArgs {
    uint32_t key[8];
    uint32_t mod[8];          // scalar-field modulus, 32-bit limbs
    uint32_t top_mask;        // limb 7 keeps MODULUS_BIT_SIZE - 224 bits
    uint32_t r;               // word_pos mod 8
    uint64_t block0;          // word_pos / 16: lane j of the launch owns stream block block0 + j
    uint32_t odd;             // (word_pos / 8) mod 2: the first candidate starts in the upper half of block0
    uint64_t n_pairs;         // lanes (blocks) in the launch
};

constexpr int CH_T = 256;                 // lanes per tile; a tile holds 2 * CH_T candidates

__device__ __forceinline__ uint32_t rotl(uint32_t x, int k) { return __builtin_rotateleft32(x, k); }
#define CG_QR(a, b, c, d) \
    a += b; d = rotl(d ^ a, 16); c += d; b = rotl(b ^ c, 12); a += b; d = rotl(d ^ a, 8); c += d; b = rotl(b ^ c, 7);

__device__ __forceinline__ void chacha12_block(const uint32_t (&key)[8], uint64_t counter, uint32_t (&out)[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};                 // stream id 0 (ChaCha12Rng::from_seed)
    uint32_t x[16];
    _Pragma("unroll") for (int i = 0; i < 16; i++) x[i] = s[i];
    _Pragma("unroll") for (int r = 0; r < 6; r++) {
        CG_QR(x[0], x[4], x[8], x[12]) CG_QR(x[1], x[5], x[9], x[13]) CG_QR(x[2], x[6], x[10], x[14]) CG_QR(x[3], x[7], x[11], x[15])
        CG_QR(x[0], x[5], x[10], x[15]) CG_QR(x[1], x[6], x[11], x[12]) CG_QR(x[2], x[7], x[8], x[13]) CG_QR(x[3], x[4], x[9], x[14])
    }
    _Pragma("unroll") for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

template <int R>
__device__ __forceinline__ void take(const uint32_t (&w)[24], uint32_t (&v0)[8], uint32_t (&v1)[8]) {
    _Pragma("unroll") for (int i = 0; i < 8; i++) { v0[i] = w[R + i]; v1[i] = w[8 + R + i]; }
}
__device__ __forceinline__ bool below(const uint32_t (&v)[8], const uint32_t (&m)[8]) {
    bool lt = false, decided = false;
    _Pragma("unroll") for (int i = 7; i >= 0; i--) { if (!decided && v[i] != m[i]) { lt = v[i] < m[i]; decided = true; } }
    return lt;
}
// the lane's two candidates (masked) and whether each exists and is accepted
__device__ __forceinline__ void lane_candidates(const ChaChaArgs& a, uint64_t j, uint32_t (&v0)[8], uint32_t (&v1)[8], bool& f0, bool& f1) {
    uint32_t w[24], b[16];
    chacha12_block(a.key, a.block0 + j, b);
    _Pragma("unroll") for (int i = 0; i < 16; i++) w[i] = b[i];
    if (a.r) { chacha12_block(a.key, a.block0 + j + 1, b); }                                 // uniform: an unaligned position reaches into the next block
    _Pragma("unroll") for (int i = 0; i < 8; i++) w[16 + i] = b[i];
    switch (a.r) {
        case 0: take<0>(w, v0, v1); break; case 1: take<1>(w, v0, v1); break; case 2: take<2>(w, v0, v1); break; case 3: take<3>(w, v0, v1); break;
        case 4: take<4>(w, v0, v1); break; case 5: take<5>(w, v0, v1); break; case 6: take<6>(w, v0, v1); break; default: take<7>(w, v0, v1); break;
    }
    v0[7] &= a.top_mask; v1[7] &= a.top_mask;
    f0 = below(v0, a.mod) && !(a.odd && j == 0);                                             // the half before word_pos is not part of the stream
    f1 = below(v1, a.mod);
}

__global__ void __launch_bounds__(CH_T) k_chacha_candidates(ChaChaArgs a, uint4* __restrict__ cand, uint32_t* __restrict__ tile_counts) {
    __shared__ uint32_t wave_cnt[CH_T / 64];
    const uint64_t j = (uint64_t)blockIdx.x * CH_T + threadIdx.x;
    uint32_t cnt = 0;
    if (j < a.n_pairs) {
        uint32_t v0[8], v1[8]; bool f0, f1;
        lane_candidates(a, j, v0, v1, f0, f1);
        uint4* dst = cand + 4 * j;
        dst[0] = make_uint4(v0[0], v0[1], v0[2], v0[3]); dst[1] = make_uint4(v0[4], v0[5], v0[6], v0[7]);
        dst[2] = make_uint4(v1[0], v1[1], v1[2], v1[3]); dst[3] = make_uint4(v1[4], v1[5], v1[6], v1[7]);
        cnt = (uint32_t)f0 + (uint32_t)f1;
    }
    _Pragma("unroll") for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < CH_T / 64; w++) t += wave_cnt[w]; tile_counts[blockIdx.x] = t; }
}

// exclusive scan of the tile counts in place; result[0] = total accepted
__global__ void __launch_bounds__(1024) k_chacha_scan_tiles(uint32_t* __restrict__ tile_counts, uint32_t n_tiles, unsigned long long* __restrict__ result) {
    __shared__ unsigned long long part[1024];
    const uint32_t per = (n_tiles + 1023) / 1024, lo = min(n_tiles, threadIdx.x * per), hi = min(n_tiles, lo + per);
    unsigned long long s = 0;
    for (uint32_t i = lo; i < hi; i++) s += tile_counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long add = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - s;                                           // exclusive prefix of this lane's segment
    if (run > 0xffffffffull) run = 0xffffffffull;                                             // offsets beyond 2^32 are never used (n < 2^32)
    for (uint32_t i = lo; i < hi; i++) { const uint32_t c = tile_counts[i]; tile_counts[i] = (uint32_t)min(run, 0xffffffffull); run += c; }
    if (threadIdx.x == 1023) result[0] = part[1023];
}

__global__ void __launch_bounds__(CH_T) k_chacha_compact(ChaChaArgs a, const uint4* __restrict__ cand, const uint32_t* __restrict__ tile_offsets, uint64_t n,
                                                      uint4* __restrict__ out, unsigned long long* __restrict__ result) {
    __shared__ uint32_t wave_cnt[CH_T / 64];
    const uint64_t j = (uint64_t)blockIdx.x * CH_T + threadIdx.x;
    uint4 q[4]; bool f0 = false, f1 = false;
    if (j < a.n_pairs) {
        _Pragma("unroll") for (int i = 0; i < 4; i++) q[i] = cand[4 * j + i];
        const uint32_t v0[8] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w}, v1[8] = {q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w};
        f0 = below(v0, a.mod) && !(a.odd && j == 0);
        f1 = below(v1, a.mod);
    }
    const uint32_t cnt = (uint32_t)f0 + (uint32_t)f1;
    uint32_t incl = cnt;                                                                    // inclusive scan across the wave
    const uint32_t lane = threadIdx.x & 63;
    _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if (lane >= (uint32_t)d) incl += up; }
    if (lane == 63) wave_cnt[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += wave_cnt[w];
    const uint64_t rank0 = (uint64_t)tile_offsets[blockIdx.x] + base + incl - cnt, rank1 = rank0 + (f0 ? 1 : 0);
    if (f0 && rank0 < n) { out[2 * rank0] = q[0]; out[2 * rank0 + 1] = q[1]; if (rank0 == n - 1) result[1] = 2 * j - a.odd; }
    if (f1 && rank1 < n) { out[2 * rank1] = q[2]; out[2 * rank1 + 1] = q[3]; if (rank1 == n - 1) result[1] = 2 * j + 1 - a.odd; }
}

// d_cand: 64 B x n_pairs; d_tiles: ceil(n_pairs / CH_T) words; d_result: two 64-bit words (accepted in all, candidate index of draw n - 1)
int chacha12_fr_rand_launch(hipStream_t st, const uint32_t* key8, const uint32_t* mod8, int modulus_bits, uint64_t word_pos, uint64_t n_pairs, uint64_t n,
                            void* d_cand, uint32_t* d_tiles, unsigned long long* d_result, void* d_out) {
    ChaChaArgs a;
    for (int i = 0; i < 8; i++) { a.key[i] = key8[i]; a.mod[i] = mod8[i]; }
    a.top_mask = modulus_bits >= 256 ? 0xffffffffu : (0xffffffffu >> (256 - modulus_bits));
    a.r = (uint32_t)(word_pos & 7); a.block0 = word_pos >> 4; a.odd = (uint32_t)((word_pos >> 3) & 1); a.n_pairs = n_pairs;
    const uint64_t tiles = (n_pairs + CH_T - 1) / CH_T;
    if (tiles == 0 || tiles > 0x7fffffffull) return fail(CG_ERR_ARG, "chacha12_fr_rand: size out of range");
    hipLaunchKernelGGL(k_chacha_candidates, dim3((unsigned)tiles), dim3(CH_T), 0, st, a, (uint4*)d_cand, d_tiles);
    hipLaunchKernelGGL(k_chacha_scan_tiles, dim3(1), dim3(1024), 0, st, d_tiles, (uint32_t)tiles, d_result);
    hipLaunchKernelGGL(k_chacha_compact, dim3((unsigned)tiles), dim3(CH_T), 0, st, a, (const uint4*)d_cand, (const uint32_t*)d_tiles, n, (uint4*)d_out, d_result);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace cg
