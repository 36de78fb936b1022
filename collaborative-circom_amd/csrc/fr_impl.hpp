// Launch side of the scalar-field kernels (vec_kernels.hpp, ntt_kernels.hpp); instantiated per field in fr_inst_*.hip.
#pragma once
#include "common.hpp"
#include "ntt_kernels.hpp"
#include "vec_kernels.hpp"
#include "msm_sort_kernels.hpp"

namespace cg {

template <class Fr> int launch_vec_binary(hipStream_t st, int op, Fr* out, const Fr* a, const Fr* b, size_t n) {
    if (!n) return 0;
    if (op == 0) hipLaunchKernelGGL((k_vec_binary<Fr, 0>), dim3(grid_for(n)), dim3(256), 0, st, out, a, b, n);
    else if (op == 1) hipLaunchKernelGGL((k_vec_binary<Fr, 1>), dim3(grid_for(n)), dim3(256), 0, st, out, a, b, n);
    else hipLaunchKernelGGL((k_vec_binary<Fr, 2>), dim3(grid_for(n)), dim3(256), 0, st, out, a, b, n);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_rep3_mul_local(hipStream_t st, Fr* out, const Fr* aa, const Fr* ab, const Fr* ba, const Fr* bb, const Fr* mask, size_t n) {
    if (!n) return 0;
    hipLaunchKernelGGL((k_rep3_mul_local<Fr>), dim3(grid_for(n)), dim3(256), 0, st, out, aa, ab, ba, bb, mask, n);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_vec_count_noncanonical(hipStream_t st, const Fr* v, size_t n, unsigned long long* n_bad) {
    if (!n) return 0;
    hipLaunchKernelGGL((k_vec_count_noncanonical<Fr>), dim3(grid_for(n)), dim3(256), 0, st, v, n, n_bad);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_distribute_powers(hipStream_t st, Fr* v, size_t n, const Fr* lo, const Fr* hi, int log_lo) {
    if (!n) return 0;
    hipLaunchKernelGGL((k_distribute_powers<Fr>), dim3(grid_for(n)), dim3(256), 0, st, v, n, lo, hi, log_lo);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_vec_affine(hipStream_t st, Fr* out, const Fr* a, size_t n, const Fr& c, const Fr& d) {
    if (!n) return 0;
    hipLaunchKernelGGL((k_vec_affine<Fr>), dim3(grid_for(n)), dim3(256), 0, st, out, a, n, c, d);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_vec_lincomb(hipStream_t st, Fr* out, long long out_off, long long out_stride, size_t n, const LincombArgs<Fr>& a) {
    if (!n) return 0;
    hipLaunchKernelGGL((k_vec_lincomb<Fr>), dim3(grid_for(n)), dim3(256), 0, st, out, out_off, out_stride, n, a);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_vec_gather_idx(hipStream_t st, Fr* out, const Fr* in, const uint32_t* idx, size_t n, uint32_t base) {
    if (!n) return 0;
    hipLaunchKernelGGL((k_vec_gather_idx<Fr>), dim3(grid_for(n)), dim3(256), 0, st, out, in, idx, n, base);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_vec_fill(hipStream_t st, Fr* v, size_t n, const Fr& value) {
    if (!n) return 0;
    hipLaunchKernelGGL((k_vec_fill<Fr>), dim3(grid_for(n)), dim3(256), 0, st, v, n, value);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_vec_gather_strided(hipStream_t st, Fr* out, const Fr* in, size_t n, size_t offset, size_t stride) {
    if (!n) return 0;
    hipLaunchKernelGGL((k_vec_gather_strided<Fr>), dim3(grid_for(n)), dim3(256), 0, st, out, in, n, offset, stride);
    HIPCHK(hipGetLastError());
    return 0;
}
// scratch: ceil(n / 2048) elements; op 0 = product, 1 = sum
template <class Fr, int OP> int launch_prefix_op(hipStream_t st, Fr* out, const Fr* in, size_t n, Fr* scratch) {
    const size_t tile = (size_t)256 * SCAN_ITEMS, ntiles = (n + tile - 1) / tile;
    hipLaunchKernelGGL((k_prefix_tiles<Fr, OP>), dim3((unsigned)ntiles), dim3(256), 0, st, out, in, n, scratch);
    if (ntiles > 1) {
        hipLaunchKernelGGL((k_prefix_totals<Fr, OP>), dim3(1), dim3(256), 0, st, scratch, ntiles);
        hipLaunchKernelGGL((k_prefix_fixup<Fr, OP>), dim3(grid_for(n)), dim3(256), 0, st, out, n, scratch);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_prefix_scan(hipStream_t st, int op, Fr* out, const Fr* in, size_t n, Fr* scratch) {
    if (!n) return 0;
    return op == 0 ? launch_prefix_op<Fr, 0>(st, out, in, n, scratch) : launch_prefix_op<Fr, 1>(st, out, in, n, scratch);
}
template <class Fr> int launch_vec_inverse(hipStream_t st, Fr* out, const Fr* in, size_t n) {
    if (!n) return 0;
    const size_t lanes = (n + INV_ITEMS - 1) / INV_ITEMS;
    hipLaunchKernelGGL((k_vec_inverse<Fr>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, st, out, in, n);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_spmv_csr(hipStream_t st, const uint32_t* row_ptr, const uint32_t* col, const Fr* coeff, size_t n_rows, const Fr* pub,
                                        uint32_t n_inputs, int party, const Fr* wit_a, const Fr* wit_b, Fr* out_a, Fr* out_b) {
    if (!n_rows) return 0;
    if (n_rows <= ((size_t)1 << 13)) hipLaunchKernelGGL((k_spmv_csr_wave<Fr>), dim3(grid_for(n_rows * 64)), dim3(256), 0, st, row_ptr, col, coeff, n_rows, pub, n_inputs, party, wit_a, wit_b, out_a, out_b);   // a wave per row
    else hipLaunchKernelGGL((k_spmv_csr<Fr>), dim3(grid_for(n_rows)), dim3(256), 0, st, row_ptr, col, coeff, n_rows, pub, n_inputs, party, wit_a, wit_b, out_a, out_b);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_build_twiddles(hipStream_t st, Fr* tw, size_t m, int log_m, const Fr* lo, const Fr* hi, int log_lo) {
    if (m > 1) hipLaunchKernelGGL((k_build_twiddles<Fr>), dim3(grid_for(m - 1)), dim3(256), 0, st, tw, m, log_m, lo, hi, log_lo);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_ntt_dif_pass(hipStream_t st, NttVecs src, NttVecs dst, int nvec, size_t n, int log_m, int s0, int k, int t, const Fr* tw) {
    static PerDeviceOnce attr_set;
    if (attr_set.pending()) { HIPCHK(hipFuncSetAttribute((const void*)k_ntt_dif_pass<Fr>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr_set.mark(); }
    const int E = 1 << (k + t);
    hipLaunchKernelGGL((k_ntt_dif_pass<Fr>), dim3((unsigned)(n / E), nvec), dim3(NTT_THREADS), (size_t)E * 32, st, src, dst, log_m, s0, k, t, tw);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_build_twiddles_lazy(hipStream_t st, void* tw, size_t m, int log_m, const Fr* lo, const Fr* hi, int log_lo, const Fr& c32) {
    if (m > 1) hipLaunchKernelGGL((k_build_twiddles_lazy<Fr>), dim3(grid_for(m / 2)), dim3(256), 0, st, tw, m / 2, log_m, lo, hi, log_lo, c32);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_ntt_ct_pass(hipStream_t st, bool first, NttVecs src, NttVecs dst, int nvec, size_t n, int log_m, int s0, int k, int t, const void* tw) {
    static PerDeviceOnce attr_set;
    if (attr_set.pending()) {
        HIPCHK(hipFuncSetAttribute((const void*)k_ntt_ct_pass<Fr, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 36 << NTT_TILE_LOG));
        HIPCHK(hipFuncSetAttribute((const void*)k_ntt_ct_pass<Fr, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 36 << NTT_TILE_LOG));
        attr_set.mark();
    }
    const int E = 1 << (k + t);
    if (first) hipLaunchKernelGGL((k_ntt_ct_pass<Fr, true>), dim3((unsigned)(n / E), nvec), dim3(NTT_THREADS), (size_t)E * 36, st, src, dst, log_m, s0, k, t, tw);
    else hipLaunchKernelGGL((k_ntt_ct_pass<Fr, false>), dim3((unsigned)(n / E), nvec), dim3(NTT_THREADS), (size_t)E * 36, st, src, dst, log_m, s0, k, t, tw);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_build_twiddles_lazy_natural(hipStream_t st, void* tw, size_t m, const Fr* lo, const Fr* hi, int log_lo, const Fr& c32) {
    if (m > 1) hipLaunchKernelGGL((k_build_twiddles_lazy_natural<Fr>), dim3(grid_for(m / 2)), dim3(256), 0, st, tw, m / 2, lo, hi, log_lo, c32);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_ntt_dit_pass(hipStream_t st, bool first, bool last, NttVecs out, NttVecs tmp, int nvec, size_t n, int log_m, int s0, int k, int t, const void* tw,
                                            const Fr* c_lo, const Fr* c_hi, int log_lo, const Fr& c32) {
    static PerDeviceOnce attr_set;
    if (attr_set.pending()) {
        HIPCHK(hipFuncSetAttribute((const void*)k_ntt_dit_pass<Fr, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 36 << NTT_TILE_LOG));
        HIPCHK(hipFuncSetAttribute((const void*)k_ntt_dit_pass<Fr, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 36 << NTT_TILE_LOG));
        HIPCHK(hipFuncSetAttribute((const void*)k_ntt_dit_pass<Fr, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 36 << NTT_TILE_LOG));
        HIPCHK(hipFuncSetAttribute((const void*)k_ntt_dit_pass<Fr, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 36 << NTT_TILE_LOG));
        attr_set.mark();
    }
    const int E = 1 << (k + t);
    const dim3 grid((unsigned)(n / E), nvec); const size_t lds = (size_t)E * 36;
    if (first && last) hipLaunchKernelGGL((k_ntt_dit_pass<Fr, true, true>), grid, dim3(NTT_THREADS), lds, st, out, tmp, log_m, s0, k, t, tw, c_lo, c_hi, log_lo, c32);
    else if (first) hipLaunchKernelGGL((k_ntt_dit_pass<Fr, true, false>), grid, dim3(NTT_THREADS), lds, st, out, tmp, log_m, s0, k, t, tw, c_lo, c_hi, log_lo, c32);
    else if (last) hipLaunchKernelGGL((k_ntt_dit_pass<Fr, false, true>), grid, dim3(NTT_THREADS), lds, st, out, tmp, log_m, s0, k, t, tw, c_lo, c_hi, log_lo, c32);
    else hipLaunchKernelGGL((k_ntt_dit_pass<Fr, false, false>), grid, dim3(NTT_THREADS), lds, st, out, tmp, log_m, s0, k, t, tw, c_lo, c_hi, log_lo, c32);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_bitrev_finish_lazy(hipStream_t st, NttVecs dst, NttVecs src, int nvec, size_t n, int log_m, const Fr* scale, const Fr* c_lo, const Fr* c_hi, int log_lo) {
    if (log_m >= 2 * BITREV_B_LAZY)
        hipLaunchKernelGGL((k_bitrev_finish_lazy<Fr>), dim3((unsigned)(n >> (2 * BITREV_B_LAZY)), nvec), dim3(256), 0, st, dst, src, log_m, scale, c_lo, c_hi, log_lo);
    else
        hipLaunchKernelGGL((k_bitrev_finish_lazy_small<Fr>), dim3(grid_for(n), nvec), dim3(256), 0, st, dst, src, log_m, scale, c_lo, c_hi, log_lo);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class Fr> int launch_bitrev_scale(hipStream_t st, NttVecs dst, NttVecs src, int nvec, size_t n, int log_m, const Fr* scale, const Fr* c_lo, const Fr* c_hi, int log_lo) {
    if (log_m >= 2 * BITREV_B)
        hipLaunchKernelGGL((k_bitrev_scale<Fr>), dim3((unsigned)(n >> (2 * BITREV_B)), nvec), dim3(256), 0, st, dst, src, log_m, scale, c_lo, c_hi, log_lo);
    else
        hipLaunchKernelGGL((k_bitrev_scale_small<Fr>), dim3(grid_for(n), nvec), dim3(256), 0, st, dst, src, log_m, scale, c_lo, c_hi, log_lo);
    HIPCHK(hipGetLastError());
    return 0;
}

// scalar-dependent half of the MSM: digits, histogram, scan, scatter.  Scratch layout (must match msm_sort_scratch_bytes):
// digits | sorted | counts | cursors | offsets.   evs (optional, 2 events) bracket the stage.
struct MsmSortPtrs { const uint32_t* sorted; const uint32_t* offsets; const uint32_t* counts; uint32_t cap; const uint32_t* overflow; };   // cap = 0: dense list
inline bool msm_sort_use_partition(size_t n, int c, int nwin, int shared) {
    const size_t nbuckets = (size_t)(shared ? 1 : nwin) << (c - 1);
    const size_t nregions = (nbuckets + ((size_t)1 << PART_REGION_LOG) - 1) >> PART_REGION_LOG;
    return (size_t)nwin * n >= ((size_t)1 << 21) && nregions >= 8 && nregions <= PART_MAX_REGIONS;
}
inline size_t msm_sort_scratch_bytes(size_t n, int c, int nwin) {   // sized for the per-window bucket sets (the shared-set mode needs less)
    const size_t nbuckets = (size_t)nwin << (c - 1);
    const size_t entries = (size_t)nwin * n, ntiles = (entries + PART_TILE - 1) / PART_TILE;
    return 2 * align_up(entries * 4) + 3 * align_up(nbuckets * 4) + align_up(((nbuckets + SCAN_TILE - 1) / SCAN_TILE) * 4) +
           align_up(entries * 8) + 2 * align_up(PART_MAX_REGIONS * 4) + 256;       // partition path: items + region totals/cursors + item count
}
template <class Fr> int msm_sort_launch(hipStream_t st, const Fr* d_scalars, size_t n, int c, int nwin, int shared, char* scratch, MsmSortPtrs* out, hipEvent_t* evs) {
    const size_t nbuckets = (size_t)(shared ? 1 : nwin) << (c - 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = scratch + off; off += align_up(bytes); return p; };
    int32_t* digits = (int32_t*)take((size_t)nwin * n * 4);
    uint32_t* sorted = (uint32_t*)take((size_t)nwin * n * 4);
    uint32_t* counts = (uint32_t*)take(nbuckets * 4);
    uint32_t* cursors = (uint32_t*)take(nbuckets * 4);
    uint32_t* offsets = (uint32_t*)take(nbuckets * 4);
    const size_t ntiles = (nbuckets + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t* tile_sums = (uint32_t*)take(ntiles * 4);
    if (evs) HIPCHK(hipEventRecord(evs[0], st));
    const bool no_small = !global_option(CG_GOPT_SORT_SMALL);                            // cg_set_option: the general six-launch schedule for every size
    // (the kernel asks for up to 68 KiB of dynamic LDS: a device whose workgroups cannot have that — the attribute call fails — takes the general
    // schedule below instead of failing the MSM: ADVICE r5)
    static PerDeviceOnce attr_small; static std::atomic<bool> small_unavailable{false};
    if (shared && !no_small && !small_unavailable.load(std::memory_order_relaxed) && attr_small.pending()) {
        if (hipFuncSetAttribute((const void*)k_msm_sort_small<Fr>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * SORT_SMALL_MAX_BUCKETS + 1024) * 4)) == hipSuccess) attr_small.mark();
        else { (void)hipGetLastError(); small_unavailable.store(true); }
    }
    if (shared && !no_small && !small_unavailable.load(std::memory_order_relaxed) && nbuckets <= SORT_SMALL_MAX_BUCKETS && (size_t)nwin * n <= SORT_SMALL_MAX_ENTRIES && n <= (1u << 24)) {   // small vectors: one launch
        const size_t lds = (2 * nbuckets + 1024) * 4;
        hipLaunchKernelGGL((k_msm_sort_small<Fr>), dim3(1), dim3(1024), lds, st, d_scalars, (uint32_t)n, c, nwin, counts, offsets, sorted);
        if (evs) HIPCHK(hipEventRecord(evs[1], st));
        HIPCHK(hipGetLastError());
        out->sorted = sorted; out->offsets = offsets; out->counts = counts; out->cap = 0; out->overflow = nullptr;
        return 0;
    }
    HIPCHK(hipMemsetAsync(counts, 0, align_up(nbuckets * 4) * 2, st));   // counts + cursors are adjacent
    if (msm_sort_use_partition(n, c, nwin, shared)) {
        const size_t entries = (size_t)nwin * n, ptiles = (entries + PART_TILE - 1) / PART_TILE;
        const uint32_t nregions = (uint32_t)((nbuckets + ((size_t)1 << PART_REGION_LOG) - 1) >> PART_REGION_LOG);
        uint64_t* items = (uint64_t*)take(entries * 8);
        uint32_t* region_total = (uint32_t*)take(PART_MAX_REGIONS * 4);
        uint32_t* region_cursor = (uint32_t*)take(PART_MAX_REGIONS * 4);
        uint32_t* total_items = (uint32_t*)take(4);
        const unsigned itiles = (unsigned)(((entries + ITEM_TILE - 1) / ITEM_TILE + 7) / 8 * 8);   // a multiple of 8: xcd_tile() deals the tiles to the XCDs
        HIPCHK(hipMemsetAsync(region_total, 0, nregions * 4, st));
        hipLaunchKernelGGL((k_msm_digits_only<Fr>), dim3(grid_for(n)), dim3(256), 0, st, d_scalars, n, c, nwin, digits);
        hipLaunchKernelGGL(k_part_hist, dim3((unsigned)ptiles), dim3(256), nregions * 4, st, digits, n, c, nwin, shared, nregions, region_total);
        hipLaunchKernelGGL(k_part_region_scan, dim3(1), dim3(1024), 0, st, region_total, nregions, region_cursor, total_items);
        const bool staged = global_option(CG_GOPT_SORT_STAGING) != 0;              // cg_set_option
        if (staged && nregions <= STAGE_MAX_REGIONS) {
            static PerDeviceOnce attr_set;
            if (attr_set.pending()) {
                HIPCHK(hipFuncSetAttribute((const void*)k_part_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_staged_lds(STAGE_MAX_REGIONS)));
                HIPCHK(hipFuncSetAttribute((const void*)k_items_scatter_staged, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ITEMS_STAGED_LDS));
                attr_set.mark();
            }
            hipLaunchKernelGGL(k_part_scatter_staged, dim3((unsigned)ptiles), dim3(STAGE_THREADS), part_staged_lds(nregions), st, digits, n, c, nwin, shared, nregions, region_cursor, items);
        } else
        hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)ptiles), dim3(256), nregions * 4, st, digits, n, c, nwin, shared, nregions, region_cursor, items);
        hipLaunchKernelGGL(k_items_hist, dim3(itiles), dim3(256), 0, st, items, total_items, counts);
        hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)ntiles), dim3(256), 0, st, counts, tile_sums, nbuckets, 0xffffffffu);
        hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, tile_sums, tile_sums, ntiles);
        hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(256), 0, st, counts, tile_sums, offsets, nbuckets, 0xffffffffu);
        if (staged && nregions <= STAGE_MAX_REGIONS) hipLaunchKernelGGL(k_items_scatter_staged, dim3(itiles), dim3(STAGE_THREADS), ITEMS_STAGED_LDS, st, items, total_items, offsets, cursors, sorted);
        else hipLaunchKernelGGL(k_items_scatter, dim3(itiles), dim3(256), 0, st, items, total_items, offsets, cursors, sorted);
        if (evs) HIPCHK(hipEventRecord(evs[1], st));
        HIPCHK(hipGetLastError());
        out->sorted = sorted; out->offsets = offsets; out->counts = counts; out->cap = 0; out->overflow = nullptr;
        return 0;
    }
    hipLaunchKernelGGL((k_msm_digits<Fr>), dim3(grid_for(n)), dim3(256), 0, st, d_scalars, n, c, nwin, shared, digits, counts);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)ntiles), dim3(256), 0, st, counts, tile_sums, nbuckets, 0xffffffffu);
    hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, tile_sums, tile_sums, ntiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(256), 0, st, counts, tile_sums, offsets, nbuckets, 0xffffffffu);
    hipLaunchKernelGGL(k_msm_scatter, dim3(grid_for((size_t)nwin * n)), dim3(256), 0, st, digits, n, c, nwin, shared, offsets, cursors, sorted);
    if (evs) HIPCHK(hipEventRecord(evs[1], st));
    HIPCHK(hipGetLastError());
    out->sorted = sorted; out->offsets = offsets; out->counts = counts; out->cap = 0; out->overflow = nullptr;
    return 0;
}
// optimistic one-pass variant: scratch layout  sorted[nbuckets * cap] | counts | offsets | tile sums | overflow flag
inline size_t msm_sort_direct_scratch_bytes(size_t n, int c, int nwin, int shared, uint32_t cap) {
    const size_t nbuckets = (size_t)(shared ? 1 : nwin) << (c - 1);
    return align_up(nbuckets * cap * 4) + 2 * align_up(nbuckets * 4) + align_up(((nbuckets + SCAN_TILE - 1) / SCAN_TILE) * 4) + 256;
}
template <class Fr> int msm_sort_direct_launch(hipStream_t st, const Fr* d_scalars, size_t n, int c, int nwin, int shared, uint32_t cap, char* scratch, MsmSortPtrs* out, hipEvent_t* evs) {
    const size_t nbuckets = (size_t)(shared ? 1 : nwin) << (c - 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = scratch + off; off += align_up(bytes); return p; };
    uint32_t* sorted = (uint32_t*)take(nbuckets * cap * 4);
    uint32_t* counts = (uint32_t*)take(nbuckets * 4);
    uint32_t* offsets = (uint32_t*)take(nbuckets * 4);
    const size_t ntiles = (nbuckets + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t* tile_sums = (uint32_t*)take(ntiles * 4);
    uint32_t* overflow = (uint32_t*)take(4);
    if (evs) HIPCHK(hipEventRecord(evs[0], st));
    HIPCHK(hipMemsetAsync(counts, 0, nbuckets * 4, st));
    HIPCHK(hipMemsetAsync(overflow, 0, 4, st));
    hipLaunchKernelGGL((k_msm_scatter_direct<Fr>), dim3(grid_for(n)), dim3(256), 0, st, d_scalars, n, c, nwin, shared, cap, counts, sorted, overflow);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)ntiles), dim3(256), 0, st, counts, tile_sums, nbuckets, cap);
    hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, tile_sums, tile_sums, ntiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)ntiles), dim3(256), 0, st, counts, tile_sums, offsets, nbuckets, cap);
    if (evs) HIPCHK(hipEventRecord(evs[1], st));
    HIPCHK(hipGetLastError());
    out->sorted = sorted; out->offsets = offsets; out->counts = counts; out->cap = cap; out->overflow = overflow;
    return 0;
}

}  // namespace cg

#define CG_INSTANTIATE_FR(Fr)                                                                                              \
    namespace cg {                                                                                                         \
    template int launch_vec_binary<Fr>(hipStream_t, int, Fr*, const Fr*, const Fr*, size_t);                               \
    template int launch_rep3_mul_local<Fr>(hipStream_t, Fr*, const Fr*, const Fr*, const Fr*, const Fr*, const Fr*, size_t); \
    template int launch_distribute_powers<Fr>(hipStream_t, Fr*, size_t, const Fr*, const Fr*, int);                        \
    template int launch_vec_count_noncanonical<Fr>(hipStream_t, const Fr*, size_t, unsigned long long*);                   \
    template int launch_vec_fill<Fr>(hipStream_t, Fr*, size_t, const Fr&);                                                 \
    template int launch_vec_gather_idx<Fr>(hipStream_t, Fr*, const Fr*, const uint32_t*, size_t, uint32_t);                \
    template int launch_vec_affine<Fr>(hipStream_t, Fr*, const Fr*, size_t, const Fr&, const Fr&);                         \
    template int launch_vec_gather_strided<Fr>(hipStream_t, Fr*, const Fr*, size_t, size_t, size_t);                       \
    template int launch_vec_lincomb<Fr>(hipStream_t, Fr*, long long, long long, size_t, const LincombArgs<Fr>&);           \
    template int launch_prefix_scan<Fr>(hipStream_t, int, Fr*, const Fr*, size_t, Fr*);                                    \
    template int launch_vec_inverse<Fr>(hipStream_t, Fr*, const Fr*, size_t);                                              \
    template int launch_spmv_csr<Fr>(hipStream_t, const uint32_t*, const uint32_t*, const Fr*, size_t, const Fr*, uint32_t, int, const Fr*, const Fr*, Fr*, Fr*); \
    template int launch_build_twiddles<Fr>(hipStream_t, Fr*, size_t, int, const Fr*, const Fr*, int);                      \
    template int launch_ntt_dif_pass<Fr>(hipStream_t, NttVecs, NttVecs, int, size_t, int, int, int, int, const Fr*);                \
    template int launch_build_twiddles_lazy<Fr>(hipStream_t, void*, size_t, int, const Fr*, const Fr*, int, const Fr&);             \
    template int launch_ntt_ct_pass<Fr>(hipStream_t, bool, NttVecs, NttVecs, int, size_t, int, int, int, int, const void*);         \
    template int launch_build_twiddles_lazy_natural<Fr>(hipStream_t, void*, size_t, const Fr*, const Fr*, int, const Fr&);          \
    template int launch_ntt_dit_pass<Fr>(hipStream_t, bool, bool, NttVecs, NttVecs, int, size_t, int, int, int, int, const void*, const Fr*, const Fr*, int, const Fr&); \
    template int launch_bitrev_finish_lazy<Fr>(hipStream_t, NttVecs, NttVecs, int, size_t, int, const Fr*, const Fr*, const Fr*, int); \
    template int launch_bitrev_scale<Fr>(hipStream_t, NttVecs, NttVecs, int, size_t, int, const Fr*, const Fr*, const Fr*, int); \
    template int msm_sort_launch<Fr>(hipStream_t, const Fr*, size_t, int, int, int, char*, MsmSortPtrs*, hipEvent_t*);          \
    template int msm_sort_direct_launch<Fr>(hipStream_t, const Fr*, size_t, int, int, int, uint32_t, char*, MsmSortPtrs*, hipEvent_t*); \
    }
