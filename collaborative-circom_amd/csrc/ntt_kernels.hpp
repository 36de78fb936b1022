// Radix-2 NTT over the 256-bit scalar fields (BN254 Fr, BLS12-381 Fr) for gfx950.
// Replaces ark-poly 0.4.2 `Radix2EvaluationDomain::{fft,ifft}_in_place` as called from
// `/root/reference/mpc-core/src/protocols/rep3.rs:893-921` (plain.rs:375-406, shamir.rs:826-871) with the generator
// supplied by the caller (`/root/reference/co-circom/co-groth16/src/groth16.rs:57-77` overrides group_gen).
//
// Structure: decimation-in-frequency passes, each pass = up to 11 butterfly stages done inside LDS on a 2^(k+t)-element
// tile (2^k strided rows x 2^t contiguous elements, 64 KiB), so a 2^22 transform is 3 HBM round trips instead of 22.
// Twiddles live in HBM in STAGE-MAJOR order (table s holds w^(j*2^s), j < m/2^(s+1), contiguous), so every stage reads a
// contiguous, coalesced run of its table.  The DIF passes leave the data bit-reversed; a final permutation pass restores
// natural order and fuses the 1/m scaling and the coset shift g^i (rep3.rs:681-688) into the same HBM round trip.
#pragma once
#include "field.hpp"
#include "lazy29.hpp"
#include "vec_kernels.hpp"

namespace cg {

constexpr int NTT_MAX_VECS = 8;
struct NttVecs { void* p[NTT_MAX_VECS]; };

constexpr int NTT_TILE_LOG = 11;                       // 2048 elements x 32 B = 64 KiB of LDS per workgroup
constexpr int NTT_THREADS = 256;
constexpr int NTT_TILE_LOG_LAZY = 10;                  // lazy passes: 1024 elements x 36 B = 36 KiB, four workgroups per CU
constexpr int BITREV_B_LAZY = 5;
// Product form of the butterflies (lazy29.hpp).  Default: by COLUMNS, the two independent products of a radix-4 step interleaved
// (-34 vector instructions per product against the row form; round 6).  -DNTT_FORM_ROWS restores the row form of rounds 2-5 for A/B runs.
#ifdef NTT_FORM_ROWS
#define NTT_MUL mul
#define NTT_MUL2(a, wa, b, wb, ra, rb) do { ra = L::mul(a, wa); rb = L::mul(b, wb); } while (0)
#define NTT_KEEP(x) (x).norm()
#else
#define NTT_MUL mul_cols
#define NTT_MUL2(a, wa, b, wb, ra, rb) L::mul2_cols(a, wa, b, wb, ra, rb)
#define NTT_KEEP(x) (x)
#endif

__host__ __device__ inline size_t tw_stage_offset(size_t m, int s) { return m - (m >> s); }

// tw_all[off(s) + j] = w^(j << s) via two-level tables lo[e & (2^log_lo-1)] * hi[e >> log_lo]
template <class F>
__global__ void __launch_bounds__(256) k_build_twiddles(F* __restrict__ tw_all, size_t m, int log_m, const F* __restrict__ lo, const F* __restrict__ hi, int log_lo) {
    const size_t total = m - 1;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        // find stage s with off(s) <= idx < off(s+1): m - idx - 1 has its top bit at position (log_m - s - 1)
        size_t rem = m - 1 - idx;                        // in [1, m-1]; off(s) = m - m/2^s
        int s = log_m - 1 - (63 - __builtin_clzll((unsigned long long)rem | 1ull));
        size_t j = idx - tw_stage_offset(m, s);
        size_t e = j << s;
        F w = ld_fp(lo + (e & (((size_t)1 << log_lo) - 1))) * ld_fp(hi + (e >> log_lo));
        st_fp(tw_all + idx, w);
    }
}

// One DIF pass: stages [s0, s0+k) on tiles of 2^k rows (stride 2^(log_m-s0-k)) x 2^t contiguous elements.
// grid.x = tiles per vector, grid.y = vector index.  src and dst may be the same buffers (in place) or different ones.
template <class F>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_dif_pass(NttVecs src_vecs, NttVecs dst_vecs, int log_m, int s0, int k, int t, const F* __restrict__ tw_all) {
    static_assert(F::N == 8, "NTT is specialised for 256-bit scalar fields");
    extern __shared__ uint4 lds[];                        // two planes (low/high 16 bytes) -> conflict-free 16-byte accesses
    const int E = 1 << (k + t);
    uint4* pl0 = lds;
    uint4* pl1 = lds + E;
    const F* src = reinterpret_cast<const F*>(src_vecs.p[blockIdx.y]);
    F* dst = reinterpret_cast<F*>(dst_vecs.p[blockIdx.y]);
    const size_t m = (size_t)1 << log_m;
    const int lo_bits = log_m - s0 - k;
    const size_t tiles_per_hi = (size_t)1 << (lo_bits - t);
    const size_t hi = blockIdx.x / tiles_per_hi;
    const size_t lo0 = (blockIdx.x % tiles_per_hi) << t;
    const size_t base = (hi << (log_m - s0)) + lo0;
    const int tmask = (1 << t) - 1;

    for (int idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        const size_t g = base + ((size_t)(idx >> t) << lo_bits) + (idx & tmask);
        const uint4* q = reinterpret_cast<const uint4*>(src + g);
        pl0[idx] = q[0]; pl1[idx] = q[1];
    }
    __syncthreads();
    for (int q = 0; q < k; q++) {
        const int pb = k - 1 - q;
        const int s = s0 + q;
        const F* tw = tw_all + tw_stage_offset(m, s);
        for (int u = threadIdx.x; u < E / 2; u += NTT_THREADS) {
            const int lo_local = u & tmask;
            const int mu = u >> t;
            const int mid0 = ((mu >> pb) << (pb + 1)) | (mu & ((1 << pb) - 1));
            const int i0 = (mid0 << t) | lo_local;
            const int i1 = i0 + (1 << (pb + t));
            const size_t j = ((size_t)(mid0 & ((1 << pb) - 1)) << lo_bits) + lo0 + lo_local;
            F a, b;
            { uint4 x = pl0[i0], y = pl1[i0]; a.v[0] = x.x; a.v[1] = x.y; a.v[2] = x.z; a.v[3] = x.w; a.v[4] = y.x; a.v[5] = y.y; a.v[6] = y.z; a.v[7] = y.w; }
            { uint4 x = pl0[i1], y = pl1[i1]; b.v[0] = x.x; b.v[1] = x.y; b.v[2] = x.z; b.v[3] = x.w; b.v[4] = y.x; b.v[5] = y.y; b.v[6] = y.z; b.v[7] = y.w; }
            F w = ld_fp(tw + j);
            F sum = a + b;
            F dif = (a - b) * w;
            pl0[i0] = make_uint4(sum.v[0], sum.v[1], sum.v[2], sum.v[3]); pl1[i0] = make_uint4(sum.v[4], sum.v[5], sum.v[6], sum.v[7]);
            pl0[i1] = make_uint4(dif.v[0], dif.v[1], dif.v[2], dif.v[3]); pl1[i1] = make_uint4(dif.v[4], dif.v[5], dif.v[6], dif.v[7]);
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        const size_t g = base + ((size_t)(idx >> t) << lo_bits) + (idx & tmask);
        uint4* q = reinterpret_cast<uint4*>(dst + g);
        q[0] = pl0[idx]; q[1] = pl1[idx];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lazy passes.  The DIF pass above is bound by VALU issue, not by HBM or LDS (521 instructions per butterfly, 162 of them
// multiplies: canonical add, canonical subtract, and a product that unpacks 8 x 32 -> 9 x 29 bits, reduces, and packs again).
// The passes below keep the elements as SIGNED LAZY 29-bit limbs (lazy29.hpp) from the first load to the final permutation:
//   * Cooley-Tukey butterflies on natural-order input, (u, v) -> (u + w v, u - w v): only v meets a multiplier, u just
//     accumulates (|value| grows by ~p per stage: 25 p after 24 stages, the limbs hold 169 p), so there is no reduction
//     and no conditional subtraction anywhere; the pairs (and therefore tiles, passes and the bit-reversed result) are those of
//     the DIF passes, the twiddle of a butterfly is w^bitrev(block index): ONE table of m/2 entries, contiguous per tile;
//   * twiddles are stored unpacked and pre-multiplied by 2^5 (the lazy core divides by 2^261, the ABI's Montgomery R is 2^256);
//   * between passes the elements stay in limb form (36 B each, three planes) in scratch; the final permutation multiplies by
//     32 * scale * coset power — which also brings the value back under 2p — and packs.
// ~350 instructions per butterfly.
struct Lazy29Planes { uint4* p0; uint4* p1; uint32_t* p2; };
__host__ __device__ inline size_t lazy29_bytes(size_t n) { return ((n * 36 + 255) / 256) * 256; }
__device__ __forceinline__ Lazy29Planes lazy29_planes(void* base, size_t n) {
    char* b = reinterpret_cast<char*>(base);
    return {reinterpret_cast<uint4*>(b), reinterpret_cast<uint4*>(b + n * 16), reinterpret_cast<uint32_t*>(b + n * 32)};
}
template <class L> __device__ __forceinline__ L lazy29_load(const uint4* p0, const uint4* p1, const uint32_t* p2, size_t i) {
    const uint4 x = p0[i], y = p1[i]; L r;
    r.l[0] = (int32_t)x.x; r.l[1] = (int32_t)x.y; r.l[2] = (int32_t)x.z; r.l[3] = (int32_t)x.w;
    r.l[4] = (int32_t)y.x; r.l[5] = (int32_t)y.y; r.l[6] = (int32_t)y.z; r.l[7] = (int32_t)y.w; r.l[8] = (int32_t)p2[i];
    return r;
}
template <class L> __device__ __forceinline__ void lazy29_store(uint4* p0, uint4* p1, uint32_t* p2, size_t i, const L& v) {
    p0[i] = make_uint4((uint32_t)v.l[0], (uint32_t)v.l[1], (uint32_t)v.l[2], (uint32_t)v.l[3]);
    p1[i] = make_uint4((uint32_t)v.l[4], (uint32_t)v.l[5], (uint32_t)v.l[6], (uint32_t)v.l[7]);
    p2[i] = (uint32_t)v.l[8];
}

// tw[i] = unpack(32 * w^bitrev(i)), i < m/2, bitrev over log_m - 1 bits; c32 = Montgomery form of 32
template <class F>
__global__ void __launch_bounds__(256) k_build_twiddles_lazy(void* tw_base, size_t half, int log_m, const F* __restrict__ lo, const F* __restrict__ hi, int log_lo, F c32) {
    typedef L29<F> L;
    const Lazy29Planes tw = lazy29_planes(tw_base, half);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = log_m > 1 ? (size_t)(__brevll((unsigned long long)i) >> (64 - (log_m - 1))) : 0;
        const F w = ld_fp(lo + (e & (((size_t)1 << log_lo) - 1))) * ld_fp(hi + (e >> log_lo)) * c32;
        lazy29_store<L>(tw.p0, tw.p1, tw.p2, i, L::template unpack<0>(w));
    }
}

// One Cooley-Tukey pass over stages [s0, s0+k): same tiles as k_ntt_dif_pass.  FIRST: src = the caller's packed vectors;
// otherwise src = dst = limb-form scratch (in place).
template <class F, bool FIRST>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_ct_pass(NttVecs src_vecs, NttVecs dst_vecs, int log_m, int s0, int k, int t, const void* tw_base) {
    static_assert(F::N == 8, "NTT is specialised for 256-bit scalar fields");
    typedef L29<F> L;
    extern __shared__ uint4 lds[];
    const int E = 1 << (k + t);
    uint4* pl0 = lds;
    uint4* pl1 = lds + E;
    uint32_t* pl2 = reinterpret_cast<uint32_t*>(lds + 2 * E);
    const size_t m = (size_t)1 << log_m;
    const Lazy29Planes tw = lazy29_planes(const_cast<void*>(tw_base), m / 2);
    const Lazy29Planes dst = lazy29_planes(dst_vecs.p[blockIdx.y], m);
    const int lo_bits = log_m - s0 - k;
    const size_t tiles_per_hi = (size_t)1 << (lo_bits - t);
    const size_t hi = blockIdx.x / tiles_per_hi;
    const size_t lo0 = (blockIdx.x % tiles_per_hi) << t;
    const size_t base = (hi << (log_m - s0)) + lo0;
    const int tmask = (1 << t) - 1;

    for (int idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        const size_t g = base + ((size_t)(idx >> t) << lo_bits) + (idx & tmask);
        L v;
        if constexpr (FIRST) v = L::template unpack<0>(ld_fp(reinterpret_cast<const F*>(src_vecs.p[blockIdx.y]) + g));
        else v = lazy29_load<L>(dst.p0, dst.p1, dst.p2, g);
        lazy29_store<L>(pl0, pl1, pl2, idx, v);
    }
    __syncthreads();
    int q = 0;
    // Two stages per trip through LDS (radix-4 step in registers): a lane takes the four elements that differ in bits pb, pb - 1 of the
    // tile index, runs the two butterflies of stage q (one twiddle, the pairs differ in bit pb) and the two of stage q + 1 (two
    // twiddles: blocks 2 blk and 2 blk + 1, the pairs differ in bit pb - 1).  Half the LDS traffic and barriers, 3 instead of 4
    // twiddle loads per four butterflies, and the two values that are only added to in stage q + 1 skip their carry normalisation.
    for (; q + 1 < k; q += 2) {
        const int pb = k - 1 - q;
        for (int u = threadIdx.x; u < E / 4; u += NTT_THREADS) {
            const int lo_local = u & tmask;
            const int mu = u >> t;
            const int mid0 = ((mu >> (pb - 1)) << (pb + 1)) | (mu & ((1 << (pb - 1)) - 1));
            const int i00 = (mid0 << t) | lo_local, d0 = 1 << (pb + t), d1 = 1 << (pb - 1 + t);
            const size_t blk = (hi << q) | (size_t)(mid0 >> (k - q));
            L a0 = lazy29_load<L>(pl0, pl1, pl2, i00), a1 = lazy29_load<L>(pl0, pl1, pl2, i00 + d1);
            const L a2 = lazy29_load<L>(pl0, pl1, pl2, i00 + d0), a3 = lazy29_load<L>(pl0, pl1, pl2, i00 + d0 + d1);
            const L w = lazy29_load<L>(tw.p0, tw.p1, tw.p2, blk);
            L v2, v3; NTT_MUL2(a2, w, a3, w, v2, v3);
            const L b0 = a0 + v2, b2 = NTT_KEEP(a0 - v2);                   // only added to below: their limbs stay unnormalised (|limb| < 2^30, < 2^31 after stage q + 1)
            const L b1 = (a1 + v3).norm(), b3 = (a1 - v3).norm();
            const L w0 = lazy29_load<L>(tw.p0, tw.p1, tw.p2, 2 * blk), w1 = lazy29_load<L>(tw.p0, tw.p1, tw.p2, 2 * blk + 1);
            L y1, y3; NTT_MUL2(b1, w0, b3, w1, y1, y3);
            lazy29_store<L>(pl0, pl1, pl2, i00, (b0 + y1).norm());
            lazy29_store<L>(pl0, pl1, pl2, i00 + d1, (b0 - y1).norm());
            lazy29_store<L>(pl0, pl1, pl2, i00 + d0, (b2 + y3).norm());
            lazy29_store<L>(pl0, pl1, pl2, i00 + d0 + d1, (b2 - y3).norm());
        }
        __syncthreads();
    }
    for (; q < k; q++) {
        const int pb = k - 1 - q;
        for (int u = threadIdx.x; u < E / 2; u += NTT_THREADS) {
            const int lo_local = u & tmask;
            const int mu = u >> t;
            const int mid0 = ((mu >> pb) << (pb + 1)) | (mu & ((1 << pb) - 1));
            const int i0 = (mid0 << t) | lo_local;
            const int i1 = i0 + (1 << (pb + t));
            const size_t blk = (hi << q) | (size_t)(mid0 >> (k - q));       // index of the butterfly's block at stage s0 + q
            const L a = lazy29_load<L>(pl0, pl1, pl2, i0), b = lazy29_load<L>(pl0, pl1, pl2, i1);
            const L w = lazy29_load<L>(tw.p0, tw.p1, tw.p2, blk);
            const L v = L::NTT_MUL(b, w);
            lazy29_store<L>(pl0, pl1, pl2, i0, (a + v).norm());
            lazy29_store<L>(pl0, pl1, pl2, i1, (a - v).norm());
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        const size_t g = base + ((size_t)(idx >> t) << lo_bits) + (idx & tmask);
        lazy29_store<L>(dst.p0, dst.p1, dst.p2, g, lazy29_load<L>(pl0, pl1, pl2, idx));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The other half of the pair: BIT-REVERSED input -> natural output, for `ifft_in_place; distribute_powers_and_mul_by_const(g, 1);
// fft_in_place` (groth16.rs:175-188, rep3.rs:893-921, :681-688) without the two permutation passes in the middle: the inverse transform
// runs the passes above (natural -> bit-reversed, limb-form scratch), the forward transform runs these.  Decimation in time: stage s
// (s = 0 first) pairs the elements whose indices differ in bit s, (u, v) -> (u + w v, u - w v) with w = omega^(j m / 2^(s+1)),
// j = index mod 2^s — the same lazy butterfly, so the value bounds of the passes above hold; twiddles come from ONE natural-order
// limb-form table tw[e] = 32 omega^e, e < m/2, read at e = j << (log_m - 1 - s).  Tiles are those of k_ntt_ct_pass taken in the opposite
// order (first the contiguous tile: stages 0 .. k_last - 1, then the strided ones), inside a tile the stages go from the low row bit up.
//   FIRST: the element at position g holds coefficient bitrev(g) of the inverse transform; it is multiplied by 32 (1/m) g^bitrev(g)
//          (the two coset tables of the permutation kernel), which also reduces it;
//   LAST:  the results are multiplied by 32 (-> the ABI's Montgomery form, reduced) and packed into the caller's vectors, natural order.
template <class F>
__global__ void __launch_bounds__(256) k_build_twiddles_lazy_natural(void* tw_base, size_t half, const F* __restrict__ lo, const F* __restrict__ hi, int log_lo, F c32) {
    typedef L29<F> L;
    const Lazy29Planes tw = lazy29_planes(tw_base, half);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < half; e += (size_t)gridDim.x * blockDim.x) {
        const F w = ld_fp(lo + (e & (((size_t)1 << log_lo) - 1))) * ld_fp(hi + (e >> log_lo)) * c32;
        lazy29_store<L>(tw.p0, tw.p1, tw.p2, e, L::template unpack<0>(w));
    }
}
template <class F, bool FIRST, bool LAST>
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_dit_pass(NttVecs out_vecs, NttVecs tmp_vecs, int log_m, int s0, int k, int t, const void* tw_base,
                                                             const F* __restrict__ cos_lo, const F* __restrict__ cos_hi, int log_lo, F c32) {
    static_assert(F::N == 8, "NTT is specialised for 256-bit scalar fields");
    typedef L29<F> L;
    extern __shared__ uint4 lds[];
    const int E = 1 << (k + t);
    uint4* pl0 = lds;
    uint4* pl1 = lds + E;
    uint32_t* pl2 = reinterpret_cast<uint32_t*>(lds + 2 * E);
    const size_t m = (size_t)1 << log_m;
    const Lazy29Planes tw = lazy29_planes(const_cast<void*>(tw_base), m / 2);
    const Lazy29Planes buf = lazy29_planes(tmp_vecs.p[blockIdx.y], m);
    const int lo_bits = log_m - s0 - k;                        // = first stage of this pass
    const size_t tiles_per_hi = (size_t)1 << (lo_bits - t);
    const size_t hi = blockIdx.x / tiles_per_hi;
    const size_t lo0 = (blockIdx.x % tiles_per_hi) << t;
    const size_t base = (hi << (log_m - s0)) + lo0;
    const int tmask = (1 << t) - 1;

    for (int idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        const size_t g = base + ((size_t)(idx >> t) << lo_bits) + (idx & tmask);
        L v = lazy29_load<L>(buf.p0, buf.p1, buf.p2, g);
        if constexpr (FIRST) {
            const size_t i = (size_t)(__brevll((unsigned long long)g) >> (64 - log_m));
            const F c = ld_fp(cos_lo + (i & (((size_t)1 << log_lo) - 1))) * ld_fp(cos_hi + (i >> log_lo));
            v = L::NTT_MUL(v, L::template unpack<0>(c));
        }
        lazy29_store<L>(pl0, pl1, pl2, idx, v);
    }
    __syncthreads();
    int q = 0;
    for (; q + 1 < k; q += 2) {                                 // two stages per trip through LDS
        const int s = lo_bits + q;
        for (int u = threadIdx.x; u < E / 4; u += NTT_THREADS) {
            const int lo_local = u & tmask;
            const int mu = u >> t;
            const int mid0 = ((mu >> q) << (q + 2)) | (mu & ((1 << q) - 1));
            const int i00 = (mid0 << t) | lo_local, d = 1 << (q + t);
            const size_t j = ((size_t)(mid0 & ((1 << q) - 1)) << lo_bits) + lo0 + lo_local;
            const size_t e0 = j << (log_m - 1 - s), e1 = j << (log_m - 2 - s);
            const L a00 = lazy29_load<L>(pl0, pl1, pl2, i00), a01 = lazy29_load<L>(pl0, pl1, pl2, i00 + d);
            const L a10 = lazy29_load<L>(pl0, pl1, pl2, i00 + 2 * d), a11 = lazy29_load<L>(pl0, pl1, pl2, i00 + 3 * d);
            const L w0 = lazy29_load<L>(tw.p0, tw.p1, tw.p2, e0);
            L v1, v3; NTT_MUL2(a01, w0, a11, w0, v1, v3);
            const L b00 = a00 + v1, b01 = a00 - v1;                  // only added to below: no carry normalisation
            const L b10 = (a10 + v3).norm(), b11 = (a10 - v3).norm();
            const L w1a = lazy29_load<L>(tw.p0, tw.p1, tw.p2, e1), w1b = lazy29_load<L>(tw.p0, tw.p1, tw.p2, e1 + (m >> 2));
            L y2, y3; NTT_MUL2(b10, w1a, b11, w1b, y2, y3);
            lazy29_store<L>(pl0, pl1, pl2, i00, (b00 + y2).norm());
            lazy29_store<L>(pl0, pl1, pl2, i00 + 2 * d, (b00 - y2).norm());
            lazy29_store<L>(pl0, pl1, pl2, i00 + d, (b01 + y3).norm());
            lazy29_store<L>(pl0, pl1, pl2, i00 + 3 * d, (b01 - y3).norm());
        }
        __syncthreads();
    }
    for (; q < k; q++) {
        const int s = lo_bits + q;
        for (int u = threadIdx.x; u < E / 2; u += NTT_THREADS) {
            const int lo_local = u & tmask;
            const int mu = u >> t;
            const int mid0 = ((mu >> q) << (q + 1)) | (mu & ((1 << q) - 1));
            const int i0 = (mid0 << t) | lo_local, i1 = i0 + (1 << (q + t));
            const size_t j = ((size_t)(mid0 & ((1 << q) - 1)) << lo_bits) + lo0 + lo_local;
            const L a = lazy29_load<L>(pl0, pl1, pl2, i0), b = lazy29_load<L>(pl0, pl1, pl2, i1);
            const L v = L::NTT_MUL(b, lazy29_load<L>(tw.p0, tw.p1, tw.p2, j << (log_m - 1 - s)));
            lazy29_store<L>(pl0, pl1, pl2, i0, (a + v).norm());
            lazy29_store<L>(pl0, pl1, pl2, i1, (a - v).norm());
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        const size_t g = base + ((size_t)(idx >> t) << lo_bits) + (idx & tmask);
        const L v = lazy29_load<L>(pl0, pl1, pl2, idx);
        if constexpr (LAST) st_fp(reinterpret_cast<F*>(out_vecs.p[blockIdx.y]) + g, L::pack_reduced(L::NTT_MUL(v, L::template unpack<0>(c32))));
        else lazy29_store<L>(buf.p0, buf.p1, buf.p2, g, v);
    }
}

// dst[bitrev(i)] = pack(src[i] * c(bitrev(i))), src in limb form: c = *scale (a constant that already contains the factor 32) or
// the product of the two coset tables (whose `lo` half contains 32 * scale).  Same LDS-tiled permutation as k_bitrev_scale.
template <class F>
__global__ void __launch_bounds__(256) k_bitrev_finish_lazy(NttVecs dst, NttVecs src, int log_m, const F* __restrict__ scale,
                                                            const F* __restrict__ cos_lo, const F* __restrict__ cos_hi, int log_lo) {
    typedef L29<F> L;
    constexpr int B = BITREV_B_LAZY, S = 1 << B;
    __shared__ uint4 tile0[S * (S + 1)];
    __shared__ uint4 tile1[S * (S + 1)];
    __shared__ uint32_t tile2[S * (S + 1)];
    const size_t m = (size_t)1 << log_m;
    const Lazy29Planes in = lazy29_planes(src.p[blockIdx.y], m);
    F* out = reinterpret_cast<F*>(dst.p[blockIdx.y]);
    const int mid_bits = log_m - 2 * B;
    const size_t mid = blockIdx.x;
    const size_t rmid = mid_bits > 0 ? (__brevll((unsigned long long)mid) >> (64 - mid_bits)) : 0;
    for (int e = threadIdx.x; e < S * S; e += 256) {
        const int h = e >> B, l = e & (S - 1);
        const size_t i = ((size_t)h << (log_m - B)) | (mid << B) | (size_t)l;
        tile0[h * (S + 1) + l] = in.p0[i]; tile1[h * (S + 1) + l] = in.p1[i]; tile2[h * (S + 1) + l] = in.p2[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * S; e += 256) {
        const int rl = e >> B, rh = e & (S - 1);
        const int l = __brev((unsigned)rl) >> (32 - B), h = __brev((unsigned)rh) >> (32 - B);
        const size_t o = ((size_t)rl << (log_m - B)) | (rmid << B) | (size_t)rh;
        const L v = lazy29_load<L>(tile0, tile1, tile2, (size_t)(h * (S + 1) + l));
        const F c = cos_lo ? ld_fp(cos_lo + (o & (((size_t)1 << log_lo) - 1))) * ld_fp(cos_hi + (o >> log_lo)) : ld_fp(scale);
        st_fp(out + o, L::pack_reduced(L::NTT_MUL(v, L::template unpack<0>(c))));
    }
}
template <class F>
__global__ void __launch_bounds__(256) k_bitrev_finish_lazy_small(NttVecs dst, NttVecs src, int log_m, const F* __restrict__ scale,
                                                                  const F* __restrict__ cos_lo, const F* __restrict__ cos_hi, int log_lo) {
    typedef L29<F> L;
    const size_t m = (size_t)1 << log_m;
    const Lazy29Planes in = lazy29_planes(src.p[blockIdx.y], m);
    F* out = reinterpret_cast<F*>(dst.p[blockIdx.y]);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = log_m ? (__brevll((unsigned long long)i) >> (64 - log_m)) : 0;
        const L v = lazy29_load<L>(in.p0, in.p1, in.p2, i);
        const F c = cos_lo ? ld_fp(cos_lo + (o & (((size_t)1 << log_lo) - 1))) * ld_fp(cos_hi + (o >> log_lo)) : ld_fp(scale);
        st_fp(out + o, L::pack_reduced(L::NTT_MUL(v, L::template unpack<0>(c))));
    }
}

// dst[bitrev(i)] = src[i] * (scale) * (coset power of bitrev(i))
//   scale: optional constant (1/m for the inverse transform), coset: optional two-level tables of g^j.
// Tiled through LDS so that both the read side and the write side move >= 512 contiguous bytes:
// i = (hi | mid | lo) with |hi| = |lo| = B bits; a workgroup takes one `mid` and all 2^B x 2^B (hi, lo) pairs.
constexpr int BITREV_B = 5;
template <class F>
__global__ void __launch_bounds__(256) k_bitrev_scale(NttVecs dst, NttVecs src, int log_m, const F* __restrict__ scale,
                                                      const F* __restrict__ cos_lo, const F* __restrict__ cos_hi, int log_lo) {
    static_assert(F::N == 8, "");
    __shared__ uint4 tile0[(1 << BITREV_B) * ((1 << BITREV_B) + 1)];
    __shared__ uint4 tile1[(1 << BITREV_B) * ((1 << BITREV_B) + 1)];
    const F* in = reinterpret_cast<const F*>(src.p[blockIdx.y]);
    F* out = reinterpret_cast<F*>(dst.p[blockIdx.y]);
    const int B = BITREV_B, S = 1 << B;
    const int mid_bits = log_m - 2 * B;
    const size_t mid = blockIdx.x;
    const size_t rmid = mid_bits > 0 ? (__brevll((unsigned long long)mid) >> (64 - mid_bits)) : 0;
    F sc = scale ? ld_fp(scale) : F::one();
    // read: rows = hi (S of them), contiguous in lo
    for (int e = threadIdx.x; e < S * S; e += 256) {
        const int h = e >> B, l = e & (S - 1);
        const size_t i = ((size_t)h << (log_m - B)) | (mid << B) | (size_t)l;
        const uint4* q = reinterpret_cast<const uint4*>(in + i);
        tile0[h * (S + 1) + l] = q[0]; tile1[h * (S + 1) + l] = q[1];
    }
    __syncthreads();
    // write: output index o = (rev(lo) | rev(mid) | rev(hi)); rows = rev(lo), contiguous in rev(hi)
    for (int e = threadIdx.x; e < S * S; e += 256) {
        const int rl = e >> B, rh = e & (S - 1);
        const int l = __brev((unsigned)rl) >> (32 - B), h = __brev((unsigned)rh) >> (32 - B);
        const size_t o = ((size_t)rl << (log_m - B)) | (rmid << B) | (size_t)rh;
        uint4 x = tile0[h * (S + 1) + l], y = tile1[h * (S + 1) + l];
        F v; v.v[0] = x.x; v.v[1] = x.y; v.v[2] = x.z; v.v[3] = x.w; v.v[4] = y.x; v.v[5] = y.y; v.v[6] = y.z; v.v[7] = y.w;
        if (cos_lo) {
            F w = ld_fp(cos_lo + (o & (((size_t)1 << log_lo) - 1))) * ld_fp(cos_hi + (o >> log_lo));
            if (scale) w = w * sc;
            v = v * w;
        } else if (scale) v = v * sc;
        st_fp(out + o, v);
    }
}

// small-m fallback for the permutation (log_m < 2*BITREV_B): one element per lane
template <class F>
__global__ void __launch_bounds__(256) k_bitrev_scale_small(NttVecs dst, NttVecs src, int log_m, const F* __restrict__ scale,
                                                            const F* __restrict__ cos_lo, const F* __restrict__ cos_hi, int log_lo) {
    const F* in = reinterpret_cast<const F*>(src.p[blockIdx.y]);
    F* out = reinterpret_cast<F*>(dst.p[blockIdx.y]);
    const size_t m = (size_t)1 << log_m;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = log_m ? (__brevll((unsigned long long)i) >> (64 - log_m)) : 0;
        F v = ld_fp(in + i);
        if (cos_lo) v = v * (ld_fp(cos_lo + (o & (((size_t)1 << log_lo) - 1))) * ld_fp(cos_hi + (o >> log_lo)));
        if (scale) v = v * ld_fp(scale);
        st_fp(out + o, v);
    }
}

}  // namespace cg
