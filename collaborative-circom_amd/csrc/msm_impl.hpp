// Launch side of the MSM pipeline (see msm_kernels.hpp); explicitly instantiated once per (curve, group) in msm_inst_*.hip
// so that each instantiation is its own translation unit and the library builds in parallel.
#pragma once
#include "common.hpp"
#include "msm_kernels.hpp"

namespace cg {

// enqueue one MSM (one share component); window sums land in h_out (pinned) via an async copy
template <class F, class Fr>
// evs (optional, 6 events): sort [0,1], accumulate [2,3], reduce [4,5]
int msm_enqueue(hipStream_t st, const Affine<F>* d_bases, size_t n, const Fr* d_scalars, int c, int nwin, char* arena_base, XYZZ<F>* h_out, hipEvent_t* evs) {
    const uint32_t nb = 1u << (c - 1);
    const size_t nbuckets = (size_t)nwin * nb;
    const uint32_t seg_len = std::max<uint32_t>(1, nb / 2048);
    const uint32_t segs = nb / seg_len;
    // carve scratch (layout must match msm_scratch_bytes)
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = arena_base + off; off += align_up(bytes); return p; };
    int32_t* digits = (int32_t*)take((size_t)nwin * n * 4);
    uint32_t* sorted = (uint32_t*)take((size_t)nwin * n * 4);
    uint32_t* counts = (uint32_t*)take(nbuckets * 4);
    uint32_t* cursors = (uint32_t*)take(nbuckets * 4);
    uint32_t* offsets = (uint32_t*)take(nbuckets * 4);
    XYZZ<F>* buckets = (XYZZ<F>*)take(nbuckets * sizeof(XYZZ<F>));
    XYZZ<F>* partials = (XYZZ<F>*)take((size_t)nwin * segs * sizeof(XYZZ<F>));
    XYZZ<F>* wsums = (XYZZ<F>*)take((size_t)nwin * sizeof(XYZZ<F>));
    if (evs) HIPCHK(hipEventRecord(evs[0], st));
    HIPCHK(hipMemsetAsync(counts, 0, align_up(nbuckets * 4) * 2, st));   // counts + cursors are adjacent
    hipLaunchKernelGGL((k_msm_digits<Fr>), dim3(grid_for(n)), dim3(256), 0, st, d_scalars, n, c, nwin, digits, counts);
    hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, counts, offsets, nbuckets);
    hipLaunchKernelGGL(k_msm_scatter, dim3(grid_for((size_t)nwin * n)), dim3(256), 0, st, digits, n, c, nwin, offsets, cursors, sorted);
    if (evs) { HIPCHK(hipEventRecord(evs[1], st)); HIPCHK(hipEventRecord(evs[2], st)); }
    hipLaunchKernelGGL((k_msm_accumulate<F>), dim3((unsigned)((nbuckets + 255) / 256)), dim3(256), 0, st, d_bases, sorted, offsets, counts, nbuckets, buckets);
    if (evs) { HIPCHK(hipEventRecord(evs[3], st)); HIPCHK(hipEventRecord(evs[4], st)); }
    const size_t nseg_threads = (size_t)nwin * segs;
    hipLaunchKernelGGL((k_msm_reduce_segments<F>), dim3((unsigned)((nseg_threads + 63) / 64)), dim3(64), 0, st, buckets, nb, seg_len, nwin, partials);
    constexpr int WT = sizeof(XYZZ<F>) > 128 ? 128 : 256;
    hipLaunchKernelGGL((k_msm_window_sum<F, WT>), dim3(nwin), dim3(WT), WT * sizeof(XYZZ<F>), st, partials, segs, wsums);
    if (evs) HIPCHK(hipEventRecord(evs[5], st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_out, wsums, (size_t)nwin * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st));
    return 0;
}
template <class F>
size_t msm_scratch_bytes(size_t n, int c, int nwin) {
    const uint32_t nb = 1u << (c - 1);
    const size_t nbuckets = (size_t)nwin * nb;
    const uint32_t seg_len = std::max<uint32_t>(1, nb / 2048);
    const uint32_t segs = nb / seg_len;
    return 2 * align_up((size_t)nwin * n * 4) + 3 * align_up(nbuckets * 4) + align_up(nbuckets * sizeof(XYZZ<F>)) +
           align_up((size_t)nwin * segs * sizeof(XYZZ<F>)) + align_up((size_t)nwin * sizeof(XYZZ<F>));
}


template <class F>
int pack_bases_launch(hipStream_t st, const uint8_t* d_raw, size_t n, size_t stride, long inf_off, Affine<F>* d_dst) {
    hipLaunchKernelGGL((k_pack_bases<F>), dim3(grid_for(n)), dim3(256), 0, st, d_raw, n, stride, inf_off, d_dst);
    HIPCHK(hipGetLastError());
    return 0;
}

template <class F>
int synth_points_launch(hipStream_t st, const XYZZ<F>* d_lo, const XYZZ<F>* d_hi, int log_t, size_t n, Affine<F>* d_out) {
    if (n) hipLaunchKernelGGL((k_synth_points<F>), dim3(grid_for(n)), dim3(256), 0, st, d_lo, d_hi, log_t, n, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace cg

#define CG_INSTANTIATE_MSM(F, Fr)                                                                                          \
    namespace cg {                                                                                                         \
    template int msm_enqueue<F, Fr>(hipStream_t, const Affine<F>*, size_t, const Fr*, int, int, char*, XYZZ<F>*, hipEvent_t*); \
    template size_t msm_scratch_bytes<F>(size_t, int, int);                                                                \
    template int pack_bases_launch<F>(hipStream_t, const uint8_t*, size_t, size_t, long, Affine<F>*);                      \
    template int synth_points_launch<F>(hipStream_t, const XYZZ<F>*, const XYZZ<F>*, int, size_t, Affine<F>*);             \
    }
