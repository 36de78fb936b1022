// Shared host-side helpers of the backend's translation units (error string, HIP error macro, launch geometry).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include "cogroth16_hip.h"

namespace cg {

inline thread_local std::string g_err;
inline int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return ::cg::fail(e_ == hipErrorOutOfMemory ? CG_ERR_OOM : CG_ERR_HIP,           \
                                                std::string(#expr) + ": " + hipGetErrorString(e_));            \
    } while (0)

// A/B knobs of the measurement scripts: environment variables in -DCG_DEBUG_KNOBS builds (make KNOBS=1 -> libcogroth16_hip_knobs.so), compiled
// out of the release library — every call site then folds to its default.  What a deployment or a test may want to move is an OPTION:
// per context (cg_ctx_set_option) or process-wide (cg_set_option, the table below).  Neither kind changes a result.
inline const char* tune_env(const char* name) {
#ifdef CG_DEBUG_KNOBS
    return getenv(name);
#else
    (void)name; return nullptr;
#endif
}
struct GlobalOptions {
    std::atomic<int64_t> v[CG_GOPT_COUNT];
    GlobalOptions() {
        for (auto& x : v) x.store(0);
        v[CG_GOPT_COMPACT_MIN_LOG].store(14); v[CG_GOPT_SORT_STAGING].store(1); v[CG_GOPT_SORT_SMALL].store(1); v[CG_GOPT_STREAM_PROBES].store(1);
    }
};
inline GlobalOptions g_options;
inline int64_t global_option(int id) { return g_options.v[id].load(std::memory_order_relaxed); }

constexpr int GRID_CAP = 2048;   // grid-stride kernels: 256 CUs x 8 workgroups
inline int grid_for(size_t n, int block = 256) { size_t g = (n + block - 1) / block; return (int)std::min<size_t>(std::max<size_t>(g, 1), GRID_CAP); }
// Launch attributes (the dynamic-LDS ceiling) belong to the (function, device) pair: call sites set them the first time they run on each
// device of the process (a multi-device session launches the same kernels on every GPU).  Several party threads reach the same call
// site at once (cgh_prove_rep3): a thread may launch only after the attribute calls have SUCCEEDED on its device, so the flag is set
// after them (`mark`), never before; two threads that both find it unset both set the attribute, which is harmless, and a failed
// set leaves the flag clear for the next call.
struct PerDeviceOnce {
    std::atomic<bool> done[64] = {};
    static int dev() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : -1; }
    bool pending() const { const int d = dev(); return d < 0 || !done[d].load(std::memory_order_acquire); }
    void mark() { const int d = dev(); if (d >= 0) done[d].store(true, std::memory_order_release); }
};
// the same for call sites that launch one of several kernels of ONE pointer type (the variants of k_msm_accumulate_pf): keyed by function and
// device, set under the lock (a second thread launches only after the first one's call has returned), raised when a larger size is asked for
inline int ensure_dynamic_lds(const void* fn, size_t bytes);
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
inline int ensure_dynamic_lds(const void* fn, size_t bytes) {
    static std::mutex mu; static std::map<std::pair<const void*, int>, size_t> done;
    int d = 0; if (hipGetDevice(&d) != hipSuccess) d = -1;
    std::lock_guard<std::mutex> l(mu);
    size_t& have = done[{fn, d}];
    if (have >= bytes) return 0;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return 0;
}
inline int log2_floor(size_t n) { int l = 0; while (((size_t)2 << l) <= n) l++; return l; }

// ---- MSM launch geometry: shared by the kernel launchers (msm_impl.hpp) and the host-side planner (capi.hip) --------------------------
constexpr int BITSUM_ITEMS = 8;   // buckets per lane in k_msm_bitsum_partial
#ifndef CG_GRID_TR
#define CG_GRID_TR 32
#endif
constexpr int GRID_LOG_L = 10, GRID_TR = CG_GRID_TR, GRID_TC = 64;   // k_msm_grid_partial: columns of the bucket grid, tile rows / columns per workgroup
struct MsmGeom {            // derived sizes shared by the host-side planner and the launchers
    uint32_t nb;            // buckets per bucket set = 2^(c-1)
    int nsets;              // bucket sets: nwin (classic) or 1 (shared: per-window precomputed tables)
    size_t nbuckets;        // nsets * nb
    uint32_t seg_len, segs; // bucket-reduction segments per bucket set
    int ngroups;            // partial sums handed to the host: nwin window sums (classic) or 16 plain groups (shared)
    uint32_t group_segs;    // segments summed per group
    uint32_t chunk_len, nchunks;
    bool bitsum;            // small shared bucket set: per-bit tree sums instead of the running-sum chain (ngroups = c bit sums)
    uint32_t bit_groups;    // workgroups per bit in k_msm_bitsum_partial
    // large shared bucket set: row / column sums + per-bit sums (k_msm_grid_partial / k_msm_grid_bitsum); ngroups = (log_l + 1) gc + log_h gr
    bool grid; int log_l, log_h; uint32_t gc, gr;
    size_t grid_partials() const { return grid ? ((size_t)1 << log_h) / GRID_TR * ((size_t)1 << log_l) + ((size_t)1 << log_h) * (((size_t)1 << log_l) / GRID_TC) : 0; }
};
constexpr int MSM_SHARED_GROUPS = 16;
// One bucket set of a reduction batch (msm_impl.hpp: msm_reduce_batch): its scratch slot, the schedule it was accumulated from, and where
// its sums go (pinned host memory).  Up to RED_MAX_SETS sets of one coordinate field and one launch geometry share the launches.
constexpr int RED_MAX_SETS = 8, ACC_MAX_SETS = 8;
// One accumulation of a batch (msm_impl.hpp: msm_accumulate_batch): table (window-0 records incl. the caller's offset; stride of the
// per-window copies, 0 = none), the schedule of its share component, its scratch slot.
struct MsmAccSet { const void* bases; size_t table_stride; const uint32_t* sorted; const uint32_t* offsets; const uint32_t* counts; char* scratch; bool may_have_inf; };
struct MsmRedSet { char* scratch; const uint32_t* offsets; const uint32_t* counts; void* h_out; };
// resident_lanes: lanes of the accumulation kernel the chip holds at once (0 = unknown).  Its workgroups do equal work and finish
// in lock step, so a launch of 2.16 residency rounds takes as long as 2.33 (the last 0.16 round runs one wave per SIMD, three
// times as fast, on a sixth of the chip): when the list is long enough the chunk length is chosen so that the chunks fill a whole
// number of rounds (2^22 points, 13 windows: 139 entries per lane, 2 rounds, instead of 128; measured 4.70 -> 4.51 ms per launch).
// chunk_request: entries per lane requested by the calling context (cg_msm_set_chunk; 0 = automatic): a context whose accumulations run
// BESIDE a latency-critical chain on another context uses shorter chunks — a workgroup then lives ~1 ms instead of ~2 and the chain's
// kernels, which can only start as workgroups retire, get onto the chip sooner.  Passed explicitly by every caller that sizes scratch
// or launches from the geometry (the planner in capi.hip, msm_acc_scratch_bytes, msm_accumulate_reduce): all three must agree on nchunks.
// g2: coordinates in the quadratic extension (LDS accumulators, no lock-stepped residency): chunks capped at CG_G2_CHUNK.
inline MsmGeom msm_geom(size_t n, int c, int nwin, bool shared, size_t resident_lanes = 0, uint32_t chunk_request = 0, bool g2 = false) {
    MsmGeom g;
    g.nb = 1u << (c - 1);
    g.nsets = shared ? 1 : nwin;
    g.nbuckets = (size_t)g.nsets * g.nb;
    const uint32_t want_segs = shared ? 32768u : 2048u;      // ~32k serial chains in total either way
    g.seg_len = std::max<uint32_t>(1, g.nb / want_segs);
    g.segs = g.nb / g.seg_len;
    g.ngroups = shared ? (g.segs >= (uint32_t)MSM_SHARED_GROUPS ? MSM_SHARED_GROUPS : 1) : nwin;
    g.group_segs = shared ? g.segs / g.ngroups : g.segs;
    static const bool no_bitsum = tune_env("CG_NO_BITSUM") != nullptr;                     // tuning knob
    g.bitsum = shared && g.nb <= (1u << 16) && g.nb >= 128 && !no_bitsum;                  // (from 2^7 buckets: the windows of circuits with a few hundred constraints)
    g.bit_groups = std::max<uint32_t>(1, (g.nb / 2 + 256 * BITSUM_ITEMS - 1) / (256 * BITSUM_ITEMS));
    if (g.bitsum) g.ngroups = c;
    static const bool no_grid = tune_env("CG_NO_GRID_REDUCE") != nullptr;                  // tuning knob (A/B against the running-sum chain)
    g.grid = shared && g.nb > (1u << 16) && !no_grid; g.log_l = 10; g.log_h = c - 1 - 10; g.gc = g.gr = 1;
    if (g.grid) {
        const size_t H = (size_t)1 << g.log_h, L = (size_t)1 << g.log_l, per_group = 256 * BITSUM_ITEMS;
        g.gc = (uint32_t)std::max<size_t>(1, (H / GRID_TR * (L / 2) + per_group - 1) / per_group);
        g.gr = (uint32_t)std::max<size_t>(1, (H / 2 * (L / GRID_TC) + per_group - 1) / per_group);
        g.ngroups = (int)((g.log_l + 1) * g.gc + g.log_h * g.gr);
    }
    const size_t entries = (size_t)nwin * n;
    static const size_t chunk_max = [] { const char* e = tune_env("CG_MSM_CHUNK"); return e ? (size_t)atoi(e) : (size_t)128; }();   // tuning knob
    static const size_t chunk_min = [] { const char* e = tune_env("CG_MSM_CHUNK_MIN"); return e ? (size_t)atoi(e) : (size_t)16; }();  // tuning knob (8 -> 16: 2^17-constraint step 7.3 -> 5.8 ms: half the continuation pieces)
    // (lists of at most 2^16 entries — circuits of a few hundred constraints — take chunks of 8: the launch is a handful of workgroups and lasts as
    // long as one lane's chain of additions, 14 us each in G2)
    const size_t chunk_floor = entries <= ((size_t)1 << 16) ? std::min<size_t>(chunk_min, 8) : chunk_min;
    g.chunk_len = (uint32_t)std::min<size_t>(chunk_max, std::max<size_t>(chunk_floor, entries / (256 * 1024)));
    // kernels without a lock-stepped residency (G2: two waves per SIMD, one of them favoured by the arbiter): shorter chunks let the
    // hardware's workgroup scheduler even out what the waves do not — 2^22 points: 12.4 -> 11.7 ms per launch alone, the step 71.5 -> 70.7 ms
    // (48: no better, the merge of twice as many boundary pieces takes it back).  CG_G2_CHUNK overrides (0 = no cap).
    static const size_t g2_chunk = [] { const char* e = tune_env("CG_G2_CHUNK"); return e ? (size_t)atoi(e) : (size_t)64; }();
    if (g2 && g2_chunk) g.chunk_len = (uint32_t)std::min<size_t>(g.chunk_len, std::max<size_t>(chunk_floor, g2_chunk));
    static const bool no_rounds = tune_env("CG_MSM_NO_ROUNDS") != nullptr;                  // tuning knob
    if (chunk_request) g.chunk_len = (uint32_t)std::min<size_t>(g.chunk_len, std::max<size_t>(chunk_floor, chunk_request));
    else if (resident_lanes && !no_rounds && entries >= resident_lanes * 2 * chunk_min) {      // from 32 entries per lane on (table slices of a multi-GPU plan:
        const size_t rounds = std::max<size_t>(1, (entries + resident_lanes * chunk_max / 2) / (resident_lanes * chunk_max));   // 2^20 points x 15 windows = one round of 80)
        g.chunk_len = (uint32_t)((entries + rounds * resident_lanes - 1) / (rounds * resident_lanes));   // between 2/3 and 3/2 of chunk_max (up to 192 entries): whole rounds matter more than the cap
    }
    g.nchunks = (uint32_t)std::max<size_t>(1, (entries + g.chunk_len - 1) / g.chunk_len);
    return g;
}

// arguments of k_vec_lincomb (vec_kernels.hpp), passed by value
constexpr int LINCOMB_MAX = 8;
template <class F>
struct LincombArgs { const F* src[LINCOMB_MAX]; long long off[LINCOMB_MAX]; long long stride[LINCOMB_MAX]; F coeff[LINCOMB_MAX]; int unit[LINCOMB_MAX]; int n_terms; };

}  // namespace cg
