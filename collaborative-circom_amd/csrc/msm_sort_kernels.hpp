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

// The whole schedule of a SMALL scalar vector in ONE launch of one workgroup (shared bucket set, <= 2^13 buckets, <= 2^15 entries): digits ->
// LDS histogram -> LDS scan -> scatter through LDS cursors.  The six launches above cost a few-hundred-constraint circuit 85 us per scalar
// vector in launch gaps alone (four vectors per proof); here the digits are simply recomputed for the scatter.  The order of the entries
// inside a bucket differs from run to run (atomics), as it does for k_msm_scatter: the bucket sums do not depend on it.
// (one REP3 party, round 5: 20 k entries 1.25 ms with this kernel, 1.39 with the general schedule | 41 k: 1.47 / 1.42 | 82 k: 1.72 / 1.59 | 164 k: 2.15 / 1.85 —
// one workgroup pays for itself up to ~2^15 entries)
constexpr uint32_t SORT_SMALL_MAX_BUCKETS = 1u << 13, SORT_SMALL_MAX_ENTRIES = 1u << 15;
template <class Fr>
__global__ void __launch_bounds__(1024) k_msm_sort_small(const Fr* __restrict__ scalars, uint32_t n, int c, int nwin,
                                                         uint32_t* __restrict__ counts, uint32_t* __restrict__ offsets, uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t sort_lds[];                      // [nb counters] [nb cursors] [1024 partial sums]
    const uint32_t nb = 1u << (c - 1), mask = (1u << c) - 1, t = threadIdx.x;
    uint32_t* cnt = sort_lds; uint32_t* cur = sort_lds + nb; uint32_t* part = sort_lds + 2 * nb;
    for (uint32_t b = t; b < nb; b += 1024) cnt[b] = 0;
    __syncthreads();
    // signed digits of scalar i, least significant window first; f(w, bucket, negative) for every non-zero digit
    auto digits_of = [&](uint32_t i, auto&& f) {
        Fr s = ld_fp(scalars + i).from_mont();
        uint32_t carry = 0;
        for (int w = 0; w < nwin; w++) {
            const uint32_t d = (s.v[0] & mask) + carry;
            _Pragma("unroll") for (int l = 0; l < Fr::N; l++) {
                const uint64_t two = ((uint64_t)(l + 1 < Fr::N ? s.v[l + 1] : 0u) << 32) | s.v[l];
                s.v[l] = (uint32_t)(two >> c);
            }
            if (d > nb) { carry = 1; if (d != (1u << c)) f((uint32_t)w, (1u << c) - d - 1, true); }      // (d = 2^c: digit 0 with a carry)
            else { carry = 0; if (d) f((uint32_t)w, d - 1, false); }
        }
    };
    for (uint32_t i = t; i < n; i += 1024) digits_of(i, [&](uint32_t, uint32_t b, bool) { atomicAdd(&cnt[b], 1u); });
    __syncthreads();
    const uint32_t chunk = (nb + 1023) / 1024, lo = min(t * chunk, nb), hi = min(lo + chunk, nb);
    uint32_t mine = 0;
    for (uint32_t b = lo; b < hi; b++) mine += cnt[b];
    part[t] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {             // inclusive scan of the 1024 partial sums
        const uint32_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - mine;
    for (uint32_t b = lo; b < hi; b++) { const uint32_t k = cnt[b]; offsets[b] = run; cur[b] = run; counts[b] = k; run += k; }
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024)
        digits_of(i, [&](uint32_t w, uint32_t b, bool neg) { sorted[atomicAdd(&cur[b], 1u)] = (w << 24) | i | (neg ? 0x80000000u : 0u); });
}

// ---------------------------------------------------------------------------------------------------------------------
// MSD partition in front of the counting sort.  A direct counting sort of 54 M (window, scalar) entries into 524 k buckets
// issues 54 M scattered 4-byte stores (and as many scattered atomics): ~5 ms per schedule, all of it DRAM sector traffic.
// Partitioning the entries first by REGION (512 consecutive buckets = 2 KB of counters, ~53 k entries = a 212 KB slice of the
// output) makes the subsequent histogram + scatter L2-local: counters and destination lines stay cache-resident while a region
// is being processed, and DRAM sees whole lines.  The partition itself writes 8-byte items in runs of ~16 per (tile, region).
//   k_part_hist     tile histogram of region ids                    (tile = PART_TILE consecutive (window, scalar) entries)
//   k_part_colscan  per-region totals -> region bases; per-(tile, region) offsets
//   k_part_scatter  items[(bucket << 32) | payload] grouped by region (rank inside the tile via LDS atomics)
//   k_items_hist / k_items_scatter   the counting sort proper, reading the region-grouped items
constexpr int PART_TILE = 16384;
constexpr int PART_REGION_LOG = 9;      // 512 buckets per region
constexpr int PART_MAX_REGIONS = 4096;  // LDS histogram limit

__device__ __forceinline__ bool part_decode(int32_t dig, size_t w, uint32_t i, uint32_t nb, int shared, uint32_t* bucket, uint32_t* payload) {
    if (dig == 0) return false;
    *bucket = (uint32_t)((shared ? 0 : w * nb) + (uint32_t)(dig < 0 ? -dig : dig) - 1);
    *payload = (shared ? ((uint32_t)w << 24) | i : i) | (dig < 0 ? 0x80000000u : 0u);
    return true;
}
// digits only (no histogram): the partition passes read them
template <class Fr>
__global__ void __launch_bounds__(256) k_msm_digits_only(const Fr* __restrict__ scalars, size_t n, int c, int nwin, int32_t* __restrict__ digits) {
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
        }
    }
}
// per-region totals: tile-local LDS histogram, flushed with one global atomic per (tile, non-empty region)
static __global__ void __launch_bounds__(256) k_part_hist(const int32_t* __restrict__ digits, size_t n, int c, int nwin, int shared, uint32_t nregions,
                                                          uint32_t* __restrict__ region_total) {
    extern __shared__ uint32_t lds_cnt[];
    for (uint32_t r = threadIdx.x; r < nregions; r += 256) lds_cnt[r] = 0;
    __syncthreads();
    const uint32_t nb = 1u << (c - 1);
    const size_t total = (size_t)nwin * n, base = (size_t)blockIdx.x * PART_TILE;
    for (int k = 0; k < PART_TILE / 256; k++) {
        const size_t idx = base + (size_t)k * 256 + threadIdx.x;
        if (idx >= total) break;
        const size_t w = idx / n; uint32_t bucket, payload;
        if (part_decode(digits[idx], w, (uint32_t)(idx - w * n), nb, shared, &bucket, &payload)) atomicAdd(&lds_cnt[bucket >> PART_REGION_LOG], 1u);
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < nregions; r += 256) { const uint32_t v = lds_cnt[r]; if (v) atomicAdd(&region_total[r], v); }
}
// exclusive scan of the region totals (<= 4096) -> region_cursor (start of each region's item range); total -> *total_items
static __global__ void __launch_bounds__(1024) k_part_region_scan(const uint32_t* __restrict__ region_total, uint32_t nregions, uint32_t* __restrict__ region_cursor,
                                                                  uint32_t* __restrict__ total_items) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (nregions + 1023) / 1024;
    uint32_t v[4] = {0, 0, 0, 0}, mine = 0;
    for (uint32_t k = 0; k < per; k++) { const uint32_t r = threadIdx.x * per + k; if (r < nregions) { v[k] = region_total[r]; mine += v[k]; } }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t t = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    if (threadIdx.x == 1023) *total_items = part[1023];
    for (uint32_t k = 0; k < per; k++) { const uint32_t r = threadIdx.x * per + k; if (r < nregions) { region_cursor[r] = run; run += v[k]; } }
}
// tile: count per region (LDS), reserve one contiguous run per region with a global atomic, then write the items into the runs
static __global__ void __launch_bounds__(256) k_part_scatter(const int32_t* __restrict__ digits, size_t n, int c, int nwin, int shared, uint32_t nregions,
                                                             uint32_t* __restrict__ region_cursor, uint64_t* __restrict__ items) {
    extern __shared__ uint32_t lds_cnt[];
    for (uint32_t r = threadIdx.x; r < nregions; r += 256) lds_cnt[r] = 0;
    __syncthreads();
    const uint32_t nb = 1u << (c - 1);
    const size_t total = (size_t)nwin * n, base = (size_t)blockIdx.x * PART_TILE;
    for (int k = 0; k < PART_TILE / 256; k++) {
        const size_t idx = base + (size_t)k * 256 + threadIdx.x;
        if (idx >= total) break;
        const size_t w = idx / n; uint32_t bucket, payload;
        if (part_decode(digits[idx], w, (uint32_t)(idx - w * n), nb, shared, &bucket, &payload)) atomicAdd(&lds_cnt[bucket >> PART_REGION_LOG], 1u);
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < nregions; r += 256) { const uint32_t v = lds_cnt[r]; lds_cnt[r] = v ? atomicAdd(&region_cursor[r], v) : 0u; }
    __syncthreads();
    for (int k = 0; k < PART_TILE / 256; k++) {
        const size_t idx = base + (size_t)k * 256 + threadIdx.x;
        if (idx >= total) break;
        const size_t w = idx / n; uint32_t bucket, payload;
        if (part_decode(digits[idx], w, (uint32_t)(idx - w * n), nb, shared, &bucket, &payload)) {
            const uint32_t pos = atomicAdd(&lds_cnt[bucket >> PART_REGION_LOG], 1u);
            items[pos] = ((uint64_t)bucket << 32) | payload;
        }
    }
}
// Counting sort proper on region-grouped items.  A workgroup takes ITEM_TILE consecutive items; they belong to a few
// neighbouring regions, so their buckets fall into a window of ITEM_WINDOW consecutive bucket ids that is histogrammed in LDS
// and flushed / reserved with one global atomic per non-empty bucket (items outside the window take the direct global path).
constexpr int ITEM_TILE = 16384;
constexpr uint32_t ITEM_WINDOW = 2048;
// Workgroup i of a launch runs on XCD i % 8 and every XCD has its own L2.  Consecutive item tiles belong to the same region, i.e. they
// write into the same 212 KB slice of the output and bump the same 512 counters: XCD x takes the x-th eighth of the tiles, so that the
// tiles that share lines and counters share an L2 and the lines reach HBM whole (measured before: 1.41 GB written per launch for
// 218 MB of output, the 4-byte stores of one line coming from several XCDs).
__device__ __forceinline__ uint32_t xcd_tile(uint32_t block, uint32_t nblocks) {
    const uint32_t per = (nblocks + 7) / 8;
    return (block & 7u) * per + (block >> 3);           // may be >= nblocks for the last XCD: the caller returns
}
static __global__ void __launch_bounds__(256) k_items_hist(const uint64_t* __restrict__ items, const uint32_t* __restrict__ total_items, uint32_t* __restrict__ counts) {
    __shared__ uint32_t cnt[ITEM_WINDOW];
    const size_t total = *total_items, base = (size_t)xcd_tile(blockIdx.x, gridDim.x) * ITEM_TILE;
    if (base >= total) return;
    for (uint32_t r = threadIdx.x; r < ITEM_WINDOW; r += 256) cnt[r] = 0;
    const uint32_t b0 = (uint32_t)(items[base] >> 32) & ~((1u << PART_REGION_LOG) - 1);
    __syncthreads();
    for (int k = 0; k < ITEM_TILE / 256; k++) {
        const size_t e = base + (size_t)k * 256 + threadIdx.x;
        if (e >= total) break;
        const uint32_t bucket = (uint32_t)(items[e] >> 32), d = bucket - b0;
        if (d < ITEM_WINDOW) atomicAdd(&cnt[d], 1u); else atomicAdd(&counts[bucket], 1u);
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < ITEM_WINDOW; r += 256) { const uint32_t v = cnt[r]; if (v) atomicAdd(&counts[b0 + r], v); }
}
static __global__ void __launch_bounds__(256) k_items_scatter(const uint64_t* __restrict__ items, const uint32_t* __restrict__ total_items, const uint32_t* __restrict__ offsets,
                                                              uint32_t* __restrict__ cursors, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t cnt[ITEM_WINDOW];
    const size_t total = *total_items, base = (size_t)xcd_tile(blockIdx.x, gridDim.x) * ITEM_TILE;
    if (base >= total) return;
    for (uint32_t r = threadIdx.x; r < ITEM_WINDOW; r += 256) cnt[r] = 0;
    const uint32_t b0 = (uint32_t)(items[base] >> 32) & ~((1u << PART_REGION_LOG) - 1);
    __syncthreads();
    for (int k = 0; k < ITEM_TILE / 256; k++) {
        const size_t e = base + (size_t)k * 256 + threadIdx.x;
        if (e >= total) break;
        const uint32_t d = (uint32_t)(items[e] >> 32) - b0;
        if (d < ITEM_WINDOW) atomicAdd(&cnt[d], 1u);
    }
    __syncthreads();
    // reserve this tile's slots in every bucket of the window: cnt[r] becomes the first destination index
    for (uint32_t r = threadIdx.x; r < ITEM_WINDOW; r += 256) { const uint32_t v = cnt[r]; cnt[r] = v ? offsets[b0 + r] + atomicAdd(&cursors[b0 + r], v) : 0u; }
    __syncthreads();
    for (int k = 0; k < ITEM_TILE / 256; k++) {
        const size_t e = base + (size_t)k * 256 + threadIdx.x;
        if (e >= total) break;
        const uint64_t it = items[e];
        const uint32_t bucket = (uint32_t)(it >> 32), d = bucket - b0;
        const uint32_t pos = d < ITEM_WINDOW ? atomicAdd(&cnt[d], 1u) : offsets[bucket] + atomicAdd(&cursors[bucket], 1u);
        sorted[pos] = (uint32_t)it;
    }
}

// ---- staged variants: the tile is ranked in LDS, laid out there in destination order, and written out run by run.
// The kernels above store each entry where its LDS rank says — 64 lanes, 64 different lines per instruction, 4 or 8 bytes each, and
// the counters say what that costs: 1.56 GB written for 436 MB of items, 1.41 GB for 218 MB of indices (r02_pmc_traffic.json).  Here a
// wave writes consecutive entries of the same run (~16 items = 128 B per (tile, region), ~32 indices = 128 B per (tile, bucket)), and the
// tile's 16 entries per lane are loaded up front (16 independent loads in flight instead of one per loop trip).
constexpr int STAGE_THREADS = 1024;
constexpr int STAGE_PER_LANE = PART_TILE / STAGE_THREADS;       // 16
constexpr uint32_t STAGE_MAX_REGIONS = 2048;                     // cnt + delta + 128 KB of staged items must fit 160 KB of LDS
inline size_t part_staged_lds(uint32_t nregions) { return (size_t)(2 * nregions + 32) * 4 + (size_t)PART_TILE * 8; }
constexpr size_t ITEMS_STAGED_LDS = (size_t)(2 * ITEM_WINDOW + 32) * 4 + (size_t)ITEM_TILE * 6;
// exclusive prefix over the workgroup of one value per lane (1024 lanes); scratch: 16 dwords of LDS; ends with a barrier
__device__ __forceinline__ uint32_t stage_block_scan(uint32_t v, uint32_t* scratch, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    _Pragma("unroll") for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < STAGE_THREADS / 64 ? scratch[lane] : 0, winc = w;
        _Pragma("unroll") for (int off = 1; off < STAGE_THREADS / 64; off <<= 1) { const uint32_t t = __shfl_up(winc, off, 64); if (lane >= off) winc += t; }
        if (lane < STAGE_THREADS / 64) scratch[lane] = winc - w;
        if (lane == STAGE_THREADS / 64 - 1) scratch[STAGE_THREADS / 64] = winc;
    }
    __syncthreads();
    const uint32_t r = scratch[wave] + inc - v;
    *total = scratch[STAGE_THREADS / 64];
    __syncthreads();
    return r;
}
// LDS: cnt[nregions] | delta[nregions] | 17 dwords | staged items (PART_TILE x 8 B)
static __global__ void __launch_bounds__(STAGE_THREADS) k_part_scatter_staged(const int32_t* __restrict__ digits, size_t n, int c, int nwin, int shared, uint32_t nregions,
                                                                            uint32_t* __restrict__ region_cursor, uint64_t* __restrict__ items) {
    extern __shared__ uint32_t lds_u32[];
    uint32_t* cnt = lds_u32;
    uint32_t* delta = cnt + nregions;
    uint32_t* scratch = delta + nregions;
    uint64_t* stage = reinterpret_cast<uint64_t*>(scratch + 32);
    for (uint32_t r = threadIdx.x; r < nregions; r += STAGE_THREADS) cnt[r] = 0;
    const uint32_t nb = 1u << (c - 1);
    const size_t total = (size_t)nwin * n, base = (size_t)blockIdx.x * PART_TILE;
    int32_t dig[STAGE_PER_LANE];
    _Pragma("unroll") for (int k = 0; k < STAGE_PER_LANE; k++) {
        const size_t idx = base + (size_t)k * STAGE_THREADS + threadIdx.x;
        dig[k] = idx < total ? digits[idx] : 0;
    }
    __syncthreads();
    uint32_t bucket[STAGE_PER_LANE], payload[STAGE_PER_LANE], rank[STAGE_PER_LANE];
    _Pragma("unroll") for (int k = 0; k < STAGE_PER_LANE; k++) {
        const size_t idx = base + (size_t)k * STAGE_THREADS + threadIdx.x;
        const size_t w = idx / n;
        bucket[k] = 0xffffffffu; payload[k] = 0; rank[k] = 0;
        if (part_decode(dig[k], w, (uint32_t)(idx - w * n), nb, shared, &bucket[k], &payload[k])) rank[k] = atomicAdd(&cnt[bucket[k] >> PART_REGION_LOG], 1u);
        else bucket[k] = 0xffffffffu;
    }
    __syncthreads();
    // tile offsets (exclusive scan over the regions) and one global reservation per non-empty region
    const uint32_t per = (nregions + STAGE_THREADS - 1) / STAGE_THREADS;          // 1 or 2
    uint32_t v[2] = {0, 0}, mine = 0;
    for (uint32_t j = 0; j < per; j++) { const uint32_t r = threadIdx.x * per + j; if (r < nregions) { v[j] = cnt[r]; mine += v[j]; } }
    uint32_t tile_items;
    uint32_t run = stage_block_scan(mine, scratch, &tile_items);
    for (uint32_t j = 0; j < per; j++) {
        const uint32_t r = threadIdx.x * per + j;
        if (r < nregions) { cnt[r] = run; delta[r] = (v[j] ? atomicAdd(&region_cursor[r], v[j]) : 0u) - run; run += v[j]; }
    }
    __syncthreads();
    _Pragma("unroll") for (int k = 0; k < STAGE_PER_LANE; k++)
        if (bucket[k] != 0xffffffffu) stage[cnt[bucket[k] >> PART_REGION_LOG] + rank[k]] = ((uint64_t)bucket[k] << 32) | payload[k];
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < tile_items; s += STAGE_THREADS) {
        const uint64_t it = stage[s];
        items[delta[(uint32_t)(it >> 32) >> PART_REGION_LOG] + s] = it;
    }
}
// LDS: cnt[ITEM_WINDOW] | delta[ITEM_WINDOW] | 32 dwords | staged indices (ITEM_TILE x 4 B) | their window slots (ITEM_TILE x 2 B)
static __global__ void __launch_bounds__(STAGE_THREADS) k_items_scatter_staged(const uint64_t* __restrict__ items, const uint32_t* __restrict__ total_items,
                                                                             const uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursors, uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t lds_u32[];
    uint32_t* cnt = lds_u32;
    uint32_t* delta = cnt + ITEM_WINDOW;
    uint32_t* scratch = delta + ITEM_WINDOW;
    uint32_t* stage = scratch + 32;
    uint16_t* slot = reinterpret_cast<uint16_t*>(stage + ITEM_TILE);
    const size_t total = *total_items, base = (size_t)xcd_tile(blockIdx.x, gridDim.x) * ITEM_TILE;
    if (base >= total) return;
    for (uint32_t r = threadIdx.x; r < ITEM_WINDOW; r += STAGE_THREADS) cnt[r] = 0;
    uint64_t it[STAGE_PER_LANE];
    _Pragma("unroll") for (int k = 0; k < STAGE_PER_LANE; k++) {
        const size_t e = base + (size_t)k * STAGE_THREADS + threadIdx.x;
        it[k] = e < total ? items[e] : ~(uint64_t)0;
    }
    const uint32_t b0 = (uint32_t)(items[base] >> 32) & ~((1u << PART_REGION_LOG) - 1);
    __syncthreads();
    uint32_t rank[STAGE_PER_LANE];
    _Pragma("unroll") for (int k = 0; k < STAGE_PER_LANE; k++) {
        rank[k] = 0;
        if (it[k] == ~(uint64_t)0) continue;
        const uint32_t bucket = (uint32_t)(it[k] >> 32), d = bucket - b0;
        if (d < ITEM_WINDOW) rank[k] = atomicAdd(&cnt[d], 1u);
        else { sorted[offsets[bucket] + atomicAdd(&cursors[bucket], 1u)] = (uint32_t)it[k]; it[k] = ~(uint64_t)0; }   // outside the window (tiny regions): direct
    }
    __syncthreads();
    constexpr uint32_t per = ITEM_WINDOW / STAGE_THREADS;                          // 2
    uint32_t v[per], mine = 0;
    _Pragma("unroll") for (uint32_t j = 0; j < per; j++) { v[j] = cnt[threadIdx.x * per + j]; mine += v[j]; }
    uint32_t tile_items;
    uint32_t run = stage_block_scan(mine, scratch, &tile_items);
    _Pragma("unroll") for (uint32_t j = 0; j < per; j++) {
        const uint32_t r = threadIdx.x * per + j;
        cnt[r] = run;
        delta[r] = (v[j] ? offsets[b0 + r] + atomicAdd(&cursors[b0 + r], v[j]) : 0u) - run;
        run += v[j];
    }
    __syncthreads();
    _Pragma("unroll") for (int k = 0; k < STAGE_PER_LANE; k++) {
        if (it[k] == ~(uint64_t)0) continue;
        const uint32_t d = (uint32_t)(it[k] >> 32) - b0, at = cnt[d] + rank[k];
        stage[at] = (uint32_t)it[k]; slot[at] = (uint16_t)d;
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < tile_items; s += STAGE_THREADS) sorted[delta[slot[s]] + s] = stage[s];
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
