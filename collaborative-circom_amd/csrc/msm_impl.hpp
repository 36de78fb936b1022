// Launch side of the group-dependent half of the MSM pipeline (bucket accumulation, merge, bucket reduction, per-window sums);
// explicitly instantiated once per (curve, group) in msm_inst_*.hip so that each is its own translation unit.
// The scalar-dependent half (digits + counting sort) is in fr_impl.hpp and is shared between MSMs over the same scalars.
#pragma once
#include <cstdlib>
#include "common.hpp"
#include "msm_kernels.hpp"

namespace cg {

// buckets per lane of k_msm_bitsum_partial and the resulting workgroups per bit
template <class B> constexpr int bitsum_items() { return BITSUM_ITEMS; }   // 2 for G2 measured slower (four times the LDS trees): 2^18 step 8.5 -> 10.1 ms
// TINY bucket sets (<= 2^10 buckets: circuits of a few hundred constraints) take ONE bucket per lane: with 8 a 128-bucket set is summed by 8 lanes in
// 8 serial additions + 3 tree levels (11 x 22 us for G2), with 1 by 64 lanes in 1 + 6
inline int bitsum_items_for(uint32_t nb) { return nb <= 1024u ? 1 : BITSUM_ITEMS; }
template <class B> uint32_t bitsum_groups(uint32_t nb) { const uint32_t it = (uint32_t)bitsum_items_for(nb); return std::max<uint32_t>(1, (nb / 2 + 256 * it - 1) / (256 * it)); }

// coordinates in the base field (G1: VGPR accumulator) or its quadratic extension (G2: LDS accumulator)
template <class F> struct IsFp2 { static constexpr bool value = false; };
template <class B> struct IsFp2<Fp2<B>> { static constexpr bool value = true; };
// lanes of the accumulation kernel resident at once on a 256-CU gfx950: G1 runs its workgroups of 256 in lock step — 3 per CU for BN254
// (32-byte coordinates, 167 VGPRs), 2 per CU for BLS12-381 (48-byte coordinates on 14 limbs, 233 VGPRs); the G2 workgroups (2 waves per
// SIMD, one of them favoured by the arbiter) do not, so no rounding there (0)
template <class F> constexpr size_t acc_resident_lanes() { return IsFp2<F>::value ? 0 : (size_t)256 * (sizeof(F) == 32 ? 3 : 2) * 256; }

template <class F>
size_t msm_acc_scratch_bytes(size_t n, int c, int nwin, bool shared, uint32_t chunk_request) {
    typedef typename BucketOf<F>::type B;
    MsmGeom g = msm_geom(n, c, nwin, shared, acc_resident_lanes<F>(), chunk_request, IsFp2<F>::value);
    g.bit_groups = bitsum_groups<B>(g.nb);
    return align_up(g.nbuckets * sizeof(B)) + align_up((size_t)g.nchunks * sizeof(B)) + align_up((size_t)g.nchunks * 4) +
           align_up(std::max({(size_t)g.nsets * g.segs, (size_t)64 * g.bit_groups, g.grid_partials()}) * sizeof(B)) + align_up((size_t)std::max(g.ngroups, 64) * sizeof(XYZZ<F>));
}

// scratch of one bucket set (one rotating slot of the context's arena): the same layout for every set of a launch geometry
template <class F>
struct AccScratch {
    typedef typename BucketOf<F>::type B;
    B* buckets; B* cont; uint32_t* cont_bucket; B* partials; XYZZ<F>* wsums;
    AccScratch(char* scratch, const MsmGeom& g) {
        size_t off = 0;
        auto take = [&](size_t bytes) { void* p = scratch + off; off += align_up(bytes); return p; };
        buckets = (B*)take(g.nbuckets * sizeof(B));
        cont = (B*)take((size_t)g.nchunks * sizeof(B));
        cont_bucket = (uint32_t*)take((size_t)g.nchunks * 4);
        partials = (B*)take(std::max({(size_t)g.nsets * g.segs, (size_t)64 * g.bit_groups, g.grid_partials()}) * sizeof(B));
        wsums = (XYZZ<F>*)take((size_t)std::max(g.ngroups, 64) * sizeof(XYZZ<F>));
    }
};
template <class F>
MsmGeom msm_geom_of(size_t n, int c, int nwin, bool shared, uint32_t chunk_request) {
    MsmGeom g = msm_geom(n, c, nwin, shared, acc_resident_lanes<F>(), chunk_request, IsFp2<F>::value);
    g.bit_groups = bitsum_groups<typename BucketOf<F>::type>(g.nb);
    return g;
}

// Bucket accumulation of UP TO ACC_MAX_SETS (bases, sorted schedule) pairs of one coordinate field and one launch geometry, each into
// the bucket set of its scratch slot, on `st`: one launch with blockIdx.y = set (a large call passes one set and fills the chip by
// itself; the tables and components of a small call go side by side).  table_stride != 0 selects the shared-bucket-set mode (bases =
// window-0 table of a [nwin][table_stride] precomputed block).  evs (optional, 2 events) bracket the accumulation KERNEL(s) alone.
template <class F>
int msm_accumulate_batch(hipStream_t st, const MsmAccSet* sets, int nsets, size_t n, int c, int nwin, bool shared, uint32_t cap, hipEvent_t* evs, uint32_t chunk_request, bool g2_slices) {
    if (nsets < 1 || nsets > ACC_MAX_SETS) return fail(CG_ERR_ARG, "internal: accumulation batch size");
    const MsmGeom g = msm_geom_of<F>(n, c, nwin, shared, chunk_request);
    typedef typename BucketOf<F>::type B;                 // limb-form points for the lazy pipelines, saturated XYZZ otherwise
    AccSets<F> S{};
    for (int i = 0; i < nsets; i++) {
        const AccScratch<F> sc(sets[i].scratch, g);
        S.bases[i] = (const Affine<F>*)sets[i].bases; S.sorted[i] = sets[i].sorted; S.offsets[i] = sets[i].offsets; S.counts[i] = sets[i].counts;
        S.buckets[i] = sc.buckets; S.cont[i] = sc.cont; S.cont_bucket[i] = sc.cont_bucket;
        S.table_stride[i] = (uint32_t)sets[i].table_stride; S.may_have_inf[i] = sets[i].may_have_inf ? 1u : 0u;
    }
    // all-zero = infinity (empty buckets are never written).  The sets of a small call sit in neighbouring scratch slots: ONE fill from the
    // first set's buckets to the end of the last set's (the slots' other arrays in between are written before they are read) instead of a
    // fill per set — six launches of 5 us in a row in front of a 46 us accumulation
    const ptrdiff_t stride = nsets > 1 ? sets[1].scratch - sets[0].scratch : 0;
    bool one_fill = stride > 0 && (size_t)(nsets - 1) * (size_t)stride + g.nbuckets * sizeof(B) <= ((size_t)8 << 20);
    for (int i = 2; i < nsets && one_fill; i++) one_fill = sets[i].scratch - sets[i - 1].scratch == stride;
    if (one_fill) HIPCHK(hipMemsetAsync(S.buckets[0], 0, (size_t)(nsets - 1) * (size_t)stride + g.nbuckets * sizeof(B), st));
    else for (int i = 0; i < nsets; i++) HIPCHK(hipMemsetAsync(S.buckets[i], 0, g.nbuckets * sizeof(B), st));
    if (evs) HIPCHK(hipEventRecord(evs[0], st));
    auto launch_acc = [&](auto kern, int T, size_t lds) -> int {                         // one set per launch (padded lists, CG_ACC_VARIANT=0)
        if (lds > 0) { if (int rc = ensure_dynamic_lds((const void*)kern, lds)) return rc; }   // once per (kernel, device), not per launch
        for (int i = 0; i < nsets; i++)
            hipLaunchKernelGGL(kern, dim3((g.nchunks + T - 1) / T), dim3(T), lds, st, S.bases[i], S.sorted[i], S.offsets[i], S.counts[i],
                               (uint32_t)g.nbuckets, g.chunk_len, g.nchunks, S.table_stride[i], cap, S.buckets[i], S.cont[i], S.cont_bucket[i], S.may_have_inf[i]);
        return 0;
    };
    // accumulator policy per coordinate field (lazy limbs everywhere: 9 x 29 bits for BN254, 14 x 28 bits for BLS12-381 Fq):
    //   G1 (Fq)    accumulator in VGPRs, 256-lane workgroups
    //   G2 (Fq2)   accumulator in LDS (72 / 112 dwords per lane), 128-lane workgroups
    int rc_acc;
    // Compact lists (cap == 0) of the lazy-limb fields run the software-pipelined kernel.  CG_ACC_VARIANT (tuning knob, read per
    // call; scripts/acc_variants.py): 0 = k_msm_accumulate, 1 = pipelined + L2 warm-up of the next record, 2 = pipelined + next
    // record in registers (one wave per SIMD less), 3 / unset = pipelined boundary reads + index list read 16 bytes at a time (the default:
    // the record prefetches measured no faster, the launch is bound by vector issue and not by the latency of the gather),
    // 10 = 4-byte index reads two iterations ahead
    const char* var_s = tune_env("CG_ACC_VARIANT");
    const int variant = var_s ? atoi(var_s) : 3;
    // CG_OPT_MSM_G2_SLICES: the G2 accumulation goes one chip-load of workgroups at a time: its workgroups hold 147 of the CU's 160 KB
    // of LDS, so nothing that needs LDS (an NTT pass: 72 KB) can start while the launch lasts — measured: a high-priority NTT pass
    // waited 11 ms, the whole launch.  Between the slices the chip drains and the waiting kernels go first; ~2 ms per launch when nothing waits.
    auto launch_pf = [&](auto kern, int T, size_t lds) -> int {
        if (lds > 0) { if (int rc = ensure_dynamic_lds((const void*)kern, lds)) return rc; }   // once per (kernel, device), not per launch
        const uint32_t groups = (g.nchunks + T - 1) / T;
        const uint32_t slice = lds > 0 && g2_slices ? 256u * (uint32_t)std::max<size_t>(1, ((size_t)160 << 10) / lds) : groups;
        for (uint32_t first = 0; first < groups; first += slice)
            hipLaunchKernelGGL(kern, dim3(std::min(slice, groups - first), (unsigned)nsets), dim3(T), lds, st, S, (uint32_t)g.nbuckets, g.chunk_len, g.nchunks, first * (uint32_t)T);
        return 0;
    };
    constexpr bool g2 = IsFp2<F>::value;
    constexpr int MINW1 = sizeof(F) == 32 ? 3 : 2;            // G1 waves per SIMD the register budget is set for (BN254: 167 VGPRs; BLS12-381: 14-limb values)
    const size_t lds2 = g2 ? (size_t)128 * 4 * sizeof(typename LazyOf<F>::type) : 0;
    if (variant && cap == 0) {
        if constexpr (!g2) {
            if (variant == 1) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 1>, 256, 256 * 4);
            else if (variant == 2) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1 - 1, 2>, 256, 0);
            else if (variant == 10) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 0>, 256, 0);
#ifdef CG_ACC_DEBUG_VARIANTS   // timing experiments with wrong results (scripts/acc_variants.py): where the launch spends its time
            else if (variant == 4) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 0, 1>, 256, 0);
            else if (variant == 5) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 0, 2>, 256, 0);
            else if (variant == 6) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 0, 3>, 256, 0);
            else if (variant == 7) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 0, 7>, 256, 0);
            else if (variant == 8) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 0, 4>, 256, 0);
            else if (variant == 9) rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 0, 8>, 256, 0);
#endif
            else rc_acc = launch_pf(k_msm_accumulate_pf<F, RegAcc29<F>, 256, MINW1, 3>, 256, 0);
        } else {
            if (variant == 1) rc_acc = launch_pf(k_msm_accumulate_pf<F, LdsAcc29<F>, 128, 1, 1>, 128, lds2 + 128 * 4);
            else if (variant == 2) rc_acc = launch_pf(k_msm_accumulate_pf<F, LdsAcc29<F>, 128, 1, 2>, 128, lds2);
            else if (variant == 10) rc_acc = launch_pf(k_msm_accumulate_pf<F, LdsAcc29<F>, 128, 1, 0>, 128, lds2);
            else rc_acc = launch_pf(k_msm_accumulate_pf<F, LdsAcc29<F>, 128, 1, 3>, 128, lds2);   // (256-lane workgroups, two per CU, were A/B'd in round 4: no gain beside a chain, +2 ms alone)
        }
    } else if constexpr (!g2) rc_acc = launch_acc(k_msm_accumulate<F, RegAcc29<F>, 256>, 256, 0);
    else rc_acc = launch_acc(k_msm_accumulate<F, LdsAcc29<F>, 128>, 128, lds2);
    if (rc_acc) return rc_acc;
    if (evs) HIPCHK(hipEventRecord(evs[1], st));
    HIPCHK(hipGetLastError());
    return 0;
}

// buckets -> partial sums for UP TO RED_MAX_SETS bucket sets of one coordinate field and one geometry in one sequence of launches on
// `st2` (the caller has made st2 wait for the accumulations): merge of the chunk-boundary pieces (then `ev_merged[0 .. n_merged)` are
// recorded: the schedules are no longer needed), bucket reduction, asynchronous copies of the g.ngroups sums of every set into its
// h_out.  The host ADDs / Horner-folds them (msm_end_impl).  Beside the accumulations of the next tables these few hundred waves cost
// the step their stand-alone duration, not their work: the sets of a call that share a field go through them together.
// evs (optional, 2 events) bracket the whole batch.
template <class F>
int msm_reduce_batch(hipStream_t st2, const MsmRedSet* sets, int nsets, size_t n, int c, int nwin, bool shared, uint32_t cap, hipEvent_t* ev_merged, int n_merged,
                     hipEvent_t* evs, uint32_t chunk_request) {
    if (nsets < 1 || nsets > RED_MAX_SETS) return fail(CG_ERR_ARG, "internal: reduction batch size");
    const MsmGeom g = msm_geom_of<F>(n, c, nwin, shared, chunk_request);
    typedef typename BucketOf<F>::type B;
    RedSets<B> S{};
    XYZZ<F>* wsums[RED_MAX_SETS];
    // The final kernel of every reduction kind only WRITES its sums, one struct per workgroup: it writes them straight into the caller's
    // page-locked result buffer (device-visible like all hipHostMalloc memory; the ticket's event, recorded behind this batch, orders the
    // host's reads) — no copy per set behind the reduction (eight copy launches per small proof).  CG_MSM_STAGED_OUT: A/B knob, the copies back.
    const bool direct_out = !global_option(CG_GOPT_MSM_STAGED_OUT);     // cg_set_option: 1 = results copied to the ticket buffer (one copy per bucket set)
    for (int i = 0; i < nsets; i++) {
        const AccScratch<F> sc(sets[i].scratch, g);
        S.buckets[i] = sc.buckets; S.cont[i] = sc.cont; S.cont_bucket[i] = sc.cont_bucket; S.partials[i] = sc.partials;
        wsums[i] = direct_out ? (XYZZ<F>*)sets[i].h_out : sc.wsums; S.wsums[i] = wsums[i];
        S.offsets[i] = sets[i].offsets; S.counts[i] = sets[i].counts;
    }
    const unsigned ys = (unsigned)nsets;
    if (evs) HIPCHK(hipEventRecord(evs[0], st2));
    auto merged = [&]() -> int { for (int i = 0; i < n_merged; i++) HIPCHK(hipEventRecord(ev_merged[i], st2)); return 0; };
    auto deliver = [&](size_t count) -> int {
        if (evs) HIPCHK(hipEventRecord(evs[1], st2));
        HIPCHK(hipGetLastError());
        if (!direct_out) for (int i = 0; i < nsets; i++) HIPCHK(hipMemcpyAsync(sets[i].h_out, wsums[i], count * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, st2));
        return 0;
    };
    // measurement only (results are WRONG): what the merges (1: skipped too) and the bucket reduction (2: only it) cost the step beside the accumulations
    // compiled only into -DCG_DEBUG_KNOBS builds (make EXTRA=-DCG_DEBUG_KNOBS OUT=...): the release library has no knob that changes results
#ifdef CG_DEBUG_KNOBS
    static const int skip_reduce = getenv("CG_DEBUG_NO_REDUCE") ? atoi(getenv("CG_DEBUG_NO_REDUCE")) : 0;
#else
    constexpr int skip_reduce = 0;
#endif
    auto fake_reduce = [&]() -> int {
        for (int i = 0; i < nsets; i++) HIPCHK(hipMemsetAsync(wsums[i], 0, (size_t)g.ngroups * sizeof(XYZZ<F>), st2));
        return deliver((size_t)g.ngroups);
    };
    if (skip_reduce == 1) { if (int rc = merged()) return rc; return fake_reduce(); }
    hipLaunchKernelGGL((k_msm_merge_direct<B>), dim3((unsigned)((g.nbuckets + 63) / 64), ys), dim3(64), 0, st2, S, (uint32_t)g.nbuckets, g.chunk_len, g.nchunks, cap);
    hipLaunchKernelGGL((k_msm_merge_cont_l1<B>), dim3((g.nchunks + 63) / 64, ys), dim3(64), 0, st2, S, g.nchunks);
    hipLaunchKernelGGL((k_msm_merge_cont<B>), dim3((g.nchunks + 63) / 64, ys), dim3(64), 0, st2, S, g.nchunks);
    if (int rc = merged()) return rc;                                               // the last readers of the sorted schedules (offsets / counts)
    if (skip_reduce == 2) return fake_reduce();
    if (g.bitsum) {
        static PerDeviceOnce attr_set2;
        if (attr_set2.pending()) {
            HIPCHK(hipFuncSetAttribute((const void*)k_msm_bitsum_partial<B, bitsum_items<B>()>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(256 * sizeof(B))));
            HIPCHK(hipFuncSetAttribute((const void*)k_msm_bitsum_partial<B, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(256 * sizeof(B))));
            HIPCHK(hipFuncSetAttribute((const void*)k_msm_bitsum_final<F, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * sizeof(B))));
            attr_set2.mark();
        }
        if (bitsum_items_for(g.nb) == 1) hipLaunchKernelGGL((k_msm_bitsum_partial<B, 1>), dim3((unsigned)(c * g.bit_groups), ys), dim3(256), 256 * sizeof(B), st2, S, g.nb, g.bit_groups);
        else hipLaunchKernelGGL((k_msm_bitsum_partial<B, bitsum_items<B>()>), dim3((unsigned)(c * g.bit_groups), ys), dim3(256), 256 * sizeof(B), st2, S, g.nb, g.bit_groups);
        hipLaunchKernelGGL((k_msm_bitsum_final<F, B>), dim3((unsigned)c, ys), dim3(64), 64 * sizeof(B), st2, S, g.bit_groups);
        return deliver((size_t)c);
    }
    if (g.grid) {
        static PerDeviceOnce attr_set3;
        if (attr_set3.pending()) {
            HIPCHK(hipFuncSetAttribute((const void*)k_msm_grid_partial<B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(256 * sizeof(B))));
            HIPCHK(hipFuncSetAttribute((const void*)k_msm_grid_bitsum<F, B, BITSUM_ITEMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(256 * sizeof(B))));
            attr_set3.mark();
        }
        const uint32_t H = 1u << g.log_h, L = 1u << g.log_l;
        hipLaunchKernelGGL((k_msm_grid_partial<B>), dim3((H / GRID_TR) * (L / GRID_TC), ys), dim3(256), 256 * sizeof(B), st2, S, (uint32_t)g.log_l, g.nb);
        hipLaunchKernelGGL((k_msm_grid_bitsum<F, B, BITSUM_ITEMS>), dim3((unsigned)g.ngroups, ys), dim3(256), 256 * sizeof(B), st2, S, (uint32_t)g.log_l, (uint32_t)g.log_h, g.gc, g.gr);
        return deliver((size_t)g.ngroups);
    }
    const size_t nseg_threads = (size_t)g.nsets * g.segs;
    hipLaunchKernelGGL((k_msm_reduce_segments<B>), dim3((unsigned)((nseg_threads + 63) / 64), ys), dim3(64), 0, st2, S, g.nb, g.seg_len, g.nsets);
    constexpr int WT = sizeof(B) > 160 ? 128 : 256;
    {
        static PerDeviceOnce attr_set;
        if (attr_set.pending()) { HIPCHK(hipFuncSetAttribute((const void*)k_msm_window_sum<F, B, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(WT * sizeof(B)))); attr_set.mark(); }
    }
    hipLaunchKernelGGL((k_msm_window_sum<F, B, WT>), dim3(g.ngroups, ys), dim3(WT), WT * sizeof(B), st2, S, g.group_segs);
    return deliver((size_t)g.ngroups);
}

template <class F>
int pack_bases_launch(hipStream_t st, const uint8_t* d_raw, size_t n, size_t stride, long inf_off, Affine<F>* d_dst) {
    hipLaunchKernelGGL((k_pack_bases<F>), dim3(grid_for(n)), dim3(256), 0, st, d_raw, n, stride, inf_off, d_dst);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class F>
int gather_points_launch(hipStream_t st, Affine<F>* d_dst, const Affine<F>* d_src, const uint32_t* d_idx, size_t n) {
    if (n) hipLaunchKernelGGL((k_gather_points<F>), dim3(grid_for(n)), dim3(256), 0, st, d_dst, d_src, d_idx, n);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class F>
int check_on_curve_launch(hipStream_t st, const Affine<F>* d_pts, size_t n, const F& b, unsigned long long* d_counters) {
    if (n) hipLaunchKernelGGL((k_check_on_curve<F>), dim3(grid_for(n)), dim3(256), 0, st, d_pts, n, b, d_counters, d_counters + 1);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class F, class Fr>
int check_subgroup_launch(hipStream_t st, const Affine<F>* d_pts, size_t n, unsigned long long* d_counters) {
    if (n) hipLaunchKernelGGL((k_check_subgroup<F, typename Fr::Params>), dim3((unsigned)std::min<size_t>((n + 127) / 128, 8192)), dim3(128), 0, st, d_pts, n, d_counters, d_counters + 1);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class F>
int check_subgroup_fast_launch(hipStream_t st, const Affine<F>* d_pts, size_t n, const FastSubgroup<F>& c, unsigned long long* d_counters) {
    if (n) hipLaunchKernelGGL((k_check_subgroup_fast<F>), dim3((unsigned)std::min<size_t>((n + 127) / 128, 8192)), dim3(128), 0, st, d_pts, n, c, d_counters, d_counters + 1);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class F>
int precompute_window_launch(hipStream_t st, const Affine<F>* d_src, Affine<F>* d_dst, size_t n, int c) {
    constexpr int K = sizeof(F) >= 64 ? 2 : 4;                     // points per lane sharing one inversion (G2 coordinates are twice as wide)
    if (n) hipLaunchKernelGGL((k_precompute_window<F, K>), dim3((unsigned)std::min<size_t>(((n + K - 1) / K + 255) / 256, 65535)), dim3(256), 0, st, d_src, d_dst, n, c);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class F>
int synth_points_launch(hipStream_t st, const XYZZ<F>* d_lo, const XYZZ<F>* d_hi, int log_t, size_t n, Affine<F>* d_out) {
    if (n) hipLaunchKernelGGL((k_synth_points<F>), dim3(grid_for(n)), dim3(256), 0, st, d_lo, d_hi, log_t, n, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

template <class F, class Fr>
int fixed_base_mul_launch(hipStream_t st, const Affine<F>& g, const Fr* d_scalars, size_t n, Affine<F>* d_tab, Affine<F>* d_out) {
    const int nwin = (Fr::Params::BITS + 7) / 8;
    hipLaunchKernelGGL((k_fixed_base_table<F>), dim3((nwin * 255 + 255) / 256), dim3(256), 0, st, g, nwin, d_tab);
    if (n) hipLaunchKernelGGL((k_fixed_base_mul<F, Fr>), dim3((unsigned)std::min<size_t>((n + 127) / 128, 16384)), dim3(128), 0, st, d_scalars, n, nwin, d_tab, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace cg

#define CG_INSTANTIATE_MSM(F, Fr)                                                                                          \
    namespace cg {                                                                                                         \
    template int msm_accumulate_batch<F>(hipStream_t, const MsmAccSet*, int, size_t, int, int, bool, uint32_t, hipEvent_t*, uint32_t, bool); \
    template int msm_reduce_batch<F>(hipStream_t, const MsmRedSet*, int, size_t, int, int, bool, uint32_t, hipEvent_t*, int, hipEvent_t*, uint32_t); \
    template size_t msm_acc_scratch_bytes<F>(size_t, int, int, bool, uint32_t);                                                      \
    template int precompute_window_launch<F>(hipStream_t, const Affine<F>*, Affine<F>*, size_t, int);                      \
    template int check_on_curve_launch<F>(hipStream_t, const Affine<F>*, size_t, const F&, unsigned long long*);           \
    template int check_subgroup_launch<F, Fr>(hipStream_t, const Affine<F>*, size_t, unsigned long long*);                 \
    template int check_subgroup_fast_launch<F>(hipStream_t, const Affine<F>*, size_t, const FastSubgroup<F>&, unsigned long long*); \
    template int pack_bases_launch<F>(hipStream_t, const uint8_t*, size_t, size_t, long, Affine<F>*);                      \
    template int gather_points_launch<F>(hipStream_t, Affine<F>*, const Affine<F>*, const uint32_t*, size_t);              \
    template int synth_points_launch<F>(hipStream_t, const XYZZ<F>*, const XYZZ<F>*, int, size_t, Affine<F>*);             \
    template int fixed_base_mul_launch<F, Fr>(hipStream_t, const Affine<F>&, const Fr*, size_t, Affine<F>*, Affine<F>*);   \
    }
