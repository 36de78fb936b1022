// Scalar side of the MSM (depends on the scalar field only, NOT on the group): Montgomery -> canonical, signed c-bit digit
// decomposition, per-(window, bucket) histogram, exclusive scan, counting-sort scatter of point indices.
// The resulting schedule (offsets / counts / sorted) is shared by every MSM that uses the same scalar vector: in
// co-groth16 the aux-witness shares drive l_query, a_query, b_g1_query and b_g2_query (groth16.rs:251,267,284,298).
#pragma once
#include "field.hpp"
#include "vec_kernels.hpp"

namespace cg {

// digits[w*n + i] = signed digit of scalar i in window w; counts[w*nb + |d|-1]++
// shared != 0: all windows use ONE bucket set (per-window precomputed base tables, see k_precompute_window)
template <class Fr>
__global__ void __launch_bounds__(256) k_msm_digits(const Fr* __restrict__ scalars, size_t n, int c, int nwin, int shared,
                                                    int32_t* __restrict__ digits, uint32_t* __restrict__ counts) {
    const uint32_t nb = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr s = ld_fp(scalars + i).from_mont();
        uint32_t carry = 0;
        for (int w = 0; w < nwin; w++) {
            uint32_t d = (s.v[0] & mask) + carry;
            _Pragma("unroll") for (int l = 0; l < Fr::N; l++) {
                uint64_t two = ((uint64_t)(l + 1 < Fr::N ? s.v[l + 1] : 0u) << 32) | s.v[l];
                s.v[l] = (uint32_t)(two >> c);
            }
            int32_t dig;
            if (d > nb) { dig = (int32_t)d - (int32_t)(1u << c); carry = 1; } else { dig = (int32_t)d; carry = 0; }
            digits[(size_t)w * n + i] = dig;
            if (dig != 0) atomicAdd(&counts[(shared ? 0 : (size_t)w * nb) + (uint32_t)(dig < 0 ? -dig : dig) - 1], 1u);
        }
    }
}

// exclusive prefix sum of `total` counters in three launches: per-tile sums, scan of the tile sums, per-tile scan + base.
constexpr int SCAN_TILE = 2048;     // counters per workgroup (256 lanes x 8)
static __global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t* __restrict__ in, uint32_t* __restrict__ tile_sums, size_t total, uint32_t cap) {
    __shared__ uint32_t red[256];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
    for (int k = 0; k < SCAN_TILE / 256; k++) { size_t i = base + (size_t)k * 256 + threadIdx.x; if (i < total) s += min(in[i], cap); }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = red[0];
}
// single workgroup: in-place exclusive scan of up to 1024 * per-lane-chunk tile sums
static __global__ void __launch_bounds__(1024) k_scan_exclusive(const uint32_t* in, uint32_t* out, size_t total) {   // in may alias out
    __shared__ uint32_t part[1024];
    const size_t chunk = (total + 1023) / 1024;
    const size_t lo = (size_t)threadIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    uint32_t s = 0;
    for (size_t i = lo; i < hi; i++) s += in[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (size_t i = lo; i < hi; i++) { uint32_t v = in[i]; out[i] = run; run += v; }
}
static __global__ void __launch_bounds__(256) k_scan_tiles(const uint32_t* __restrict__ in, const uint32_t* __restrict__ tile_base, uint32_t* __restrict__ out, size_t total, uint32_t cap) {
    __shared__ uint32_t part[256];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * (SCAN_TILE / 256);
    uint32_t v[SCAN_TILE / 256]; uint32_t s = 0;
    for (int k = 0; k < SCAN_TILE / 256; k++) { size_t i = base + k; v[k] = i < total ? min(in[i], cap) : 0; s += v[k]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t t = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = tile_base[blockIdx.x] + (threadIdx.x ? part[threadIdx.x - 1] : 0);
    for (int k = 0; k < SCAN_TILE / 256; k++) { size_t i = base + k; if (i < total) out[i] = run; run += v[k]; }
}

// sorted[offsets[bucket] + k] = point index | sign << 31
// shared != 0: entry = window << 24 | point index (n <= 2^24), sign in bit 31
static __global__ void __launch_bounds__(256) k_msm_scatter(const int32_t* __restrict__ digits, size_t n, int c, int nwin, int shared, const uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ cursors, uint32_t* __restrict__ sorted) {
    const uint32_t nb = 1u << (c - 1);
    const size_t total = (size_t)nwin * n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int32_t dig = digits[idx];
        if (dig == 0) continue;
        const size_t w = idx / n;
        const uint32_t i = (uint32_t)(idx - w * n);
        const size_t bucket = (shared ? 0 : w * nb) + (uint32_t)(dig < 0 ? -dig : dig) - 1;
        const uint32_t pos = offsets[bucket] + atomicAdd(&cursors[bucket], 1u);
        sorted[pos] = (shared ? ((uint32_t)w << 24) | i : i) | (dig < 0 ? 0x80000000u : 0u);
    }
}

// Optimistic one-pass scatter: every bucket owns `cap` slots, so no histogram pass is needed — one atomic per (scalar, window)
// instead of two, no digit array.  counts[] ends up holding the true per-bucket counts; if any exceeds cap the entry is dropped
// and *overflow is set: the host then recomputes that MSM with the exact two-pass schedule (cg_msm_end), so results never
// depend on the guess.  Uniformly random scalars (REP3 shares) stay far below cap (mean + 25 % + 6 sigma).
template <class Fr>
__global__ void __launch_bounds__(256) k_msm_scatter_direct(const Fr* __restrict__ scalars, size_t n, int c, int nwin, int shared, uint32_t cap,
                                                            uint32_t* __restrict__ counts, uint32_t* __restrict__ sorted, uint32_t* __restrict__ overflow) {
    const uint32_t nb = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr s = ld_fp(scalars + i).from_mont();
        uint32_t carry = 0;
        for (int w = 0; w < nwin; w++) {
            uint32_t d = (s.v[0] & mask) + carry;
            _Pragma("unroll") for (int l = 0; l < Fr::N; l++) {
                uint64_t two = ((uint64_t)(l + 1 < Fr::N ? s.v[l + 1] : 0u) << 32) | s.v[l];
                s.v[l] = (uint32_t)(two >> c);
            }
            int32_t dig;
            if (d > nb) { dig = (int32_t)d - (int32_t)(1u << c); carry = 1; } else { dig = (int32_t)d; carry = 0; }
            if (dig == 0) continue;
            const size_t bucket = (shared ? 0 : (size_t)w * nb) + (uint32_t)(dig < 0 ? -dig : dig) - 1;
            const uint32_t pos = atomicAdd(&counts[bucket], 1u);
            if (pos < cap) sorted[bucket * cap + pos] = (shared ? ((uint32_t)w << 24) | (uint32_t)i : (uint32_t)i) | (dig < 0 ? 0x80000000u : 0u);
            else *overflow = 1u;
        }
    }
}

}  // namespace cg
