// Internals shared by the translation units of the C ABI (capi.hip: tables, vector operations, point / field helpers, statistics, options;
// capi_streams.hip: stream pool, pipe map, contexts; capi_memory.hip: block caches, copies, peer copies, preflight; capi_msm.hip and
// capi_ntt.hip: the launch logic of the two kernel families).  Round 6: capi.hip was one 2 600-line unit; a pure move.
#pragma once
#include "common.hpp"
#include "curve.hpp"
#include "subgroup.hpp"
#include "host_ec64.hpp"
#include "ntt_kernels.hpp"   // NttVecs, plan constants (no kernels are instantiated in this translation unit)

#include <cmath>
#include <chrono>
#include <map>
#include <set>
#include <mutex>
#include <vector>

using namespace cg;

// launchers living in msm_inst_*.hip / fr_inst_*.hip (explicit instantiations)
namespace cg {
struct MsmSortPtrs { const uint32_t* sorted; const uint32_t* offsets; const uint32_t* counts; uint32_t cap; const uint32_t* overflow; };
template <class Fr> int msm_sort_launch(hipStream_t st, const Fr* d_scalars, size_t n, int c, int nwin, int shared, char* scratch, MsmSortPtrs* out, hipEvent_t* evs);
template <class Fr> int msm_sort_direct_launch(hipStream_t st, const Fr* d_scalars, size_t n, int c, int nwin, int shared, uint32_t cap, char* scratch, MsmSortPtrs* out, hipEvent_t* evs);
inline size_t msm_sort_direct_scratch_bytes(size_t n, int c, int nwin, int shared, uint32_t cap) {
    const size_t nbuckets = (size_t)(shared ? 1 : nwin) << (c - 1);
    return align_up(nbuckets * cap * 4) + 2 * align_up(nbuckets * 4) + align_up(((nbuckets + 2047) / 2048) * 4) + 256;
}
template <class F> int msm_accumulate_batch(hipStream_t st, const MsmAccSet* sets, int nsets, size_t n, int c, int nwin, bool shared, uint32_t cap, hipEvent_t* evs, uint32_t chunk_request, bool g2_slices);
template <class F> int msm_reduce_batch(hipStream_t st2, const MsmRedSet* sets, int nsets, size_t n, int c, int nwin, bool shared, uint32_t cap, hipEvent_t* ev_merged, int n_merged,
                                        hipEvent_t* evs, uint32_t chunk_request);
template <class F> size_t msm_acc_scratch_bytes(size_t n, int c, int nwin, bool shared, uint32_t chunk_request);
template <class F> int precompute_window_launch(hipStream_t st, const Affine<F>* d_src, Affine<F>* d_dst, size_t n, int c);
template <class F> int check_on_curve_launch(hipStream_t st, const Affine<F>* d_pts, size_t n, const F& b, unsigned long long* d_counters);
template <class F, class Fr> int check_subgroup_launch(hipStream_t st, const Affine<F>* d_pts, size_t n, unsigned long long* d_counters);
template <class F> int check_subgroup_fast_launch(hipStream_t st, const Affine<F>* d_pts, size_t n, const FastSubgroup<F>& c, unsigned long long* d_counters);
inline size_t msm_sort_scratch_bytes(size_t n, int c, int nwin) {   // must match fr_impl.hpp
    const size_t nbuckets = (size_t)nwin << (c - 1);
    const size_t entries = (size_t)nwin * n;
    return 2 * align_up(entries * 4) + 3 * align_up(nbuckets * 4) + align_up(((nbuckets + 2047) / 2048) * 4) + align_up(entries * 8) + 2 * align_up(4096 * 4) + 256;
}
template <class F> int pack_bases_launch(hipStream_t st, const uint8_t* d_raw, size_t n, size_t stride, long inf_off, Affine<F>* d_dst);
template <class F> int gather_points_launch(hipStream_t st, Affine<F>* d_dst, const Affine<F>* d_src, const uint32_t* d_idx, size_t n);
template <class Fr> int launch_vec_gather_idx(hipStream_t st, Fr* out, const Fr* in, const uint32_t* idx, size_t n, uint32_t base);
template <class F> int synth_points_launch(hipStream_t st, const XYZZ<F>* d_lo, const XYZZ<F>* d_hi, int log_t, size_t n, Affine<F>* d_out);
template <class F, class Fr> int fixed_base_mul_launch(hipStream_t st, const Affine<F>& g, const Fr* d_scalars, size_t n, Affine<F>* d_tab, Affine<F>* d_out);
template <class Fr> int launch_vec_binary(hipStream_t st, int op, Fr* out, const Fr* a, const Fr* b, size_t n);
int chacha12_fr_rand_launch(hipStream_t st, const uint32_t* key8, const uint32_t* mod8, int modulus_bits, uint64_t word_pos, uint64_t n_pairs, uint64_t n,
                            void* d_cand, uint32_t* d_tiles, unsigned long long* d_result, void* d_out);   // chacha_rand.hip
template <class Fr> int launch_rep3_mul_local(hipStream_t st, Fr* out, const Fr* aa, const Fr* ab, const Fr* ba, const Fr* bb, const Fr* mask, size_t n);
template <class Fr> int launch_distribute_powers(hipStream_t st, Fr* v, size_t n, const Fr* lo, const Fr* hi, int log_lo);
template <class Fr> int launch_vec_count_noncanonical(hipStream_t st, const Fr* v, size_t n, unsigned long long* n_bad);
template <class Fr> int launch_vec_fill(hipStream_t st, Fr* v, size_t n, const Fr& value);
template <class Fr> int launch_vec_affine(hipStream_t st, Fr* out, const Fr* a, size_t n, const Fr& c, const Fr& d);
template <class Fr> int launch_vec_gather_strided(hipStream_t st, Fr* out, const Fr* in, size_t n, size_t offset, size_t stride);
template <class Fr> int launch_vec_lincomb(hipStream_t st, Fr* out, long long out_off, long long out_stride, size_t n, const LincombArgs<Fr>& a);
template <class Fr> int launch_prefix_scan(hipStream_t st, int op, Fr* out, const Fr* in, size_t n, Fr* scratch);
template <class Fr> int launch_vec_inverse(hipStream_t st, Fr* out, const Fr* in, size_t n);
template <class Fr> int launch_spmv_csr(hipStream_t st, const uint32_t* row_ptr, const uint32_t* col, const Fr* coeff, size_t n_rows, const Fr* pub,
                                        uint32_t n_inputs, int party, const Fr* wit_a, const Fr* wit_b, Fr* out_a, Fr* out_b);
template <class Fr> int launch_build_twiddles(hipStream_t st, Fr* tw, size_t m, int log_m, const Fr* lo, const Fr* hi, int log_lo);
template <class Fr> int launch_build_twiddles_lazy(hipStream_t st, void* tw, size_t m, int log_m, const Fr* lo, const Fr* hi, int log_lo, const Fr& c32);
template <class Fr> int launch_ntt_ct_pass(hipStream_t st, bool first, NttVecs src, NttVecs dst, int nvec, size_t n, int log_m, int s0, int k, int t, const void* tw);
template <class Fr> int launch_build_twiddles_lazy_natural(hipStream_t st, void* tw, size_t m, const Fr* lo, const Fr* hi, int log_lo, const Fr& c32);
template <class Fr> int launch_ntt_dit_pass(hipStream_t st, bool first, bool last, NttVecs out, NttVecs tmp, int nvec, size_t n, int log_m, int s0, int k, int t, const void* tw,
                                            const Fr* c_lo, const Fr* c_hi, int log_lo, const Fr& c32);
template <class Fr> int launch_bitrev_finish_lazy(hipStream_t st, NttVecs dst, NttVecs src, int nvec, size_t n, int log_m, const Fr* scale, const Fr* c_lo, const Fr* c_hi, int log_lo);
template <class Fr> int launch_ntt_dif_pass(hipStream_t st, NttVecs src, NttVecs dst, int nvec, size_t n, int log_m, int s0, int k, int t, const Fr* tw);
template <class Fr> int launch_bitrev_scale(hipStream_t st, NttVecs dst, NttVecs src, int nvec, size_t n, int log_m, const Fr* scale, const Fr* c_lo, const Fr* c_hi, int log_lo);
}  // namespace cg

// every device allocation of the library outside cg_dev_alloc: hipMalloc that, when the device is out of memory, gives back the blocks
// parked by cg_dev_free on the current device (up to CG_DEV_CACHE_MB of them) and tries once more
hipError_t hip_malloc_flush(void** p, size_t bytes);
template <class T> hipError_t hip_malloc_flush(T** p, size_t bytes) { return hip_malloc_flush((void**)p, bytes); }

// (types that are members of cg_ctx: named namespace, not an anonymous one — cg_ctx is one type for every translation unit)
namespace cgi {

struct Arena {
    char* base = nullptr; size_t cap = 0, used = 0;
    void* take(size_t bytes) { void* p = base + used; used += align_up(bytes); return p; }
};

struct TwKey { int curve; int log_m; uint32_t gen[8]; int kind = 0;   // kind 0: stage-major packed tables (DIF passes), 1: bit-reversed limb-form table (lazy passes)
    bool operator<(const TwKey& o) const { if (curve != o.curve) return curve < o.curve; if (log_m != o.log_m) return log_m < o.log_m; if (kind != o.kind) return kind < o.kind; return memcmp(gen, o.gen, sizeof gen) < 0; } };
struct CosetKey { TwKey k; uint32_t scale[8]; bool operator<(const CosetKey& o) const { if (k < o.k) return true; if (o.k < k) return false; return memcmp(scale, o.scale, sizeof scale) < 0; } };
struct CosetTables { void* lo; void* hi; int log_lo; };

enum { TAG_MSM = 0, TAG_NTT, TAG_VEC, TAG_SPMV, TAG_SORT, TAG_ACC_G1, TAG_ACC_G2, TAG_REDUCE, TAG_COUNT };
struct EvPair { hipEvent_t a, b; int tag; };

struct MsmTicket {
    bool live = false;
    int curve = 0, group = 0, k = 0, c = 0, nwin = 0;
    // optimistic one-pass scatter: per-component overflow flags (pinned) + what is needed to redo the MSM exactly if one is set
    uint32_t* h_flags = nullptr; bool optimistic = false;
    const cg_bases* bases = nullptr; size_t offset = 0, n = 0; std::vector<const void*> scalars;
    int nsums = 0;            // partial sums per component delivered by the GPU
    bool plain_fold = false;  // true: add them (precomputed tables); false: Horner with c doublings (classic)
    bool bit_fold = false;    // the sums are the per-bit sums T_k of a small shared bucket set: Horner with ONE doubling per step
    bool grid_fold = false; int log_l = 0, log_h = 0; uint32_t gc = 1, gr = 1;   // row / column bit sums of a large shared bucket set (k_msm_grid_*)
    void* h_pinned = nullptr; size_t pinned_bytes = 0;   // k * nwin window sums (XYZZ)
    hipEvent_t done = nullptr;
};

}  // namespace cgi
using namespace cgi;


struct cg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    // second stream for the latency-bound bucket reductions, two rotating scratch slots, and the events that order them
    hipStream_t aux = nullptr;
    static constexpr int ACC_SLOTS_MAX = 8;                // rotating scratch slots of the accumulate / reduce pipeline (4 in use, see msm_begin_multi_impl)
    hipEvent_t ev_acc[ACC_SLOTS_MAX] = {}, ev_red[ACC_SLOTS_MAX] = {};
    bool slot_busy[ACC_SLOTS_MAX] = {};
    bool aux_pending = false; int last_slot = 0;
    // third stream for the scalar-side sort (HBM/latency bound): the schedule of component j+1 is built while component j is
    // accumulated (integer-VALU bound) on the main stream; two rotating schedule slots
    hipStream_t sortst = nullptr;
    hipEvent_t ev_in = nullptr, ev_sorted[2] = {nullptr, nullptr}, ev_sched_free[2] = {nullptr, nullptr};
    // the merge kernels on the aux stream are the last readers of a schedule: [slot] = the most recent one per schedule slot
    // ([reduction stream: 0 = aux, 1 = the sort stream (wide mode runs the G1 batch there beside the G2 batch on aux)][schedule slot]: one event
    // per stream, so that the later record of one batch cannot replace the other batch's mark)
    hipEvent_t ev_merged[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; bool merged_pending[2][2] = {{false, false}, {false, false}};
    // copy streams of the asynchronous host <-> device transfers (cg_dev_*_begin): MPC exchanges move under the compute
    static constexpr int COPY_TICKETS = 256;
    hipStream_t h2d = nullptr, d2h = nullptr;
    // ticket = running copy number (31 bits); slot = ticket % COPY_TICKETS holds its event.  A slot is recycled only after its
    // previous copy has completed (copy_begin waits for it), so a ticket older than the slot's current owner names a finished copy.
    hipEvent_t copy_ev[COPY_TICKETS] = {}; uint32_t copy_id[COPY_TICKETS] = {}; hipEvent_t ev_copy_order = nullptr; uint32_t copy_next = 0;
    static constexpr int MARKS = 16;                      // cg_stream_mark: points of the stream order that downloads can be ordered behind
    hipEvent_t mark_ev[MARKS] = {}; uint32_t mark_next = 0;
    // cg_chacha12_fr_rand_dev_begin / _finish: draws in flight (candidate buffers, the event behind the count's download, the page-locked count)
    struct RandDraw { bool live = false; void* d_cand = nullptr; void* d_small = nullptr; hipEvent_t ev = nullptr; uint64_t word_pos = 0; size_t n = 0; };
    static constexpr int RAND_DRAWS = 8;
    RandDraw rand_draw[RAND_DRAWS]; unsigned long long* rand_result = nullptr;   // [RAND_DRAWS][2] page-locked: accepted candidates, index of the last pair used
    // cg_msm_scalars_after: the scalar-side schedule of component j of the NEXT begin call waits for this event (an upload still in flight)
    hipEvent_t comp_after[4] = {};
    hipStream_t joinst = nullptr; hipEvent_t park_ev[5] = {};   // cg_dev_free: a work-free stream that joins the context's streams behind a released block
    // priority class of each stream: +1 high, 0 normal, -1 low (pooled_stream)
    int prio_main = 0, prio_side = 1, prio_copy = 0;
    uint32_t msm_chunk = 0;                               // cg_msm_set_chunk / CG_OPT_MSM_CHUNK
    int solo_log = 18;                                    // CG_OPT_MSM_SOLO_LOG: `solo` calls (msm_begin_multi_impl_) of at most 2^this entries
    int off_main_log = 22;                                // CG_MSM_OFF_MAIN_LOG: wide calls of at most 2^this entries keep their accumulations OFF the main stream (0 = never), see msm_begin_multi_impl_
    int one_stream_log = 0;                               // CG_MSM_ONE_STREAM_LOG: calls of at most 2^this entries run on the main stream alone (0 = never, the default: measured slower)
    int table_order = 0, g2_after = -1, g2_slices = 0, red_batch = 2, acc_slots = 4, wide_small = 22;   // CG_OPT_MSM_TABLE_ORDER / _G2_SLICES / _REDUCE_BATCH / _ACC_SLOTS (cg_ctx_set_option)
    hipEvent_t ev_peer = nullptr;                         // cg_dev_copy_peer: "source stream reached this point"
    Arena arena;
    Arena ntt_arena;                                      // limb-form scratch of the transforms: NOT the MSM arena (ensure_ntt_arena)
    Arena solo_arena;                                     // scratch of tiny single-field MSM calls that run in stream order on the main stream (`solo` in msm_begin_multi_impl_)
    std::vector<void*> retired;                        // outgrown arena blocks that enqueued kernels may still use
    void* gather_buf = nullptr; size_t gather_cap = 0;   // scalars gathered for compacted tables (see cg_bases::compact)
    bool sorts_unordered = false;                         // the last call's sorts ran off the main stream and the main stream has not waited for them (off_main): the next gather must
    std::map<TwKey, void*> twiddles;
    std::map<CosetKey, CosetTables> cosets;
    std::vector<MsmTicket> tickets;
    int msm_window = 0;
    int scatter_cap = -1;     // < 0 = exact two-pass sort (default: measured equally fast), 0 = optimistic one-pass scatter with automatic capacity, > 0 = forced capacity (tests)
    bool stats_on = false;
    cg_stage_times stats{};
    std::vector<EvPair> ev_live, ev_free;
};

struct cg_bases {
    int device, curve, group;
    size_t n, pt_bytes;
    void* d_pts;
    int pre_c = 0, pre_nwin = 0;   // per-window precomputed tables (cg_bases_precompute): d_pre = [pre_nwin][n] points, window 0 = d_pts copy
    void* d_pre = nullptr;
    // Real zkey queries are sparse in points: variables that occur in no B constraint leave the point at infinity in b_g1_query /
    // b_g2_query (34 % of the poseidon fixture).  When >= 1/8 of a table is infinity the MSMs run over a COMPACTED copy: `compact`
    // holds the non-infinity records, `h_live` / `d_live` their original indices (ascending), and the scalars are gathered to match.
    cg_bases* compact = nullptr;
    std::vector<uint32_t> h_live; uint32_t* d_live = nullptr; uint64_t live_sig = 0;
    bool no_inf = false;          // registration census found no point at infinity: the accumulate kernel skips its per-point test
};

namespace {

int ensure_arena(cg_ctx* ctx, size_t bytes) {
    ctx->arena.used = 0;
    if (ctx->aux_pending) {   // reductions of an earlier MSM may still be reading the arena on the aux stream
        for (int sl = 0; sl < cg_ctx::ACC_SLOTS_MAX; sl++) if (ctx->slot_busy[sl]) { HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_red[sl], 0)); ctx->slot_busy[sl] = false; }
        ctx->aux_pending = false;
    }
    if (bytes <= ctx->arena.cap) return 0;
    // grow WITHOUT draining the streams (a host that blocks here stalls the exchange pipeline of the drivers): kernels already
    // enqueued keep their pointers into the old block, which is retired and freed once the streams are idle
    const bool idle = hipStreamQuery(ctx->stream) == hipSuccess && (!ctx->aux || hipStreamQuery(ctx->aux) == hipSuccess) && (!ctx->sortst || hipStreamQuery(ctx->sortst) == hipSuccess);
    (void)hipGetLastError();                                // hipErrorNotReady from the queries is not an error
    if (idle) { for (void* p : ctx->retired) HIPCHK(hipFree(p)); ctx->retired.clear(); }
    if (ctx->arena.base) { if (idle) HIPCHK(hipFree(ctx->arena.base)); else ctx->retired.push_back(ctx->arena.base); }
    ctx->arena.base = nullptr; ctx->arena.cap = 0;
    size_t want = align_up(bytes + bytes / 8, 1 << 20);
    HIPCHK(hip_malloc_flush((void**)&ctx->arena.base, want));
    ctx->arena.cap = want;
    return 0;
}

// The transforms' scratch is a block of its own.  It used to be the front of the MSM arena, and a transform therefore had to wait for every bucket
// reduction still reading that arena on the side streams: a one-context party's witness map — constraint rows, product, TRANSFORMS, first
// exchange — stood still until the witness-independent MSMs it had started first were completely done (Poseidon fixture: the first exchange's
// download waited 0.2-0.46 ms of a 1.5 ms proof; with the reductions switched off it took 45 us).  Transforms run on the main stream only,
// so successive users of this block are ordered by the stream itself.
int ensure_main_stream_block(cg_ctx* ctx, Arena& a, size_t bytes) {
    if (bytes <= a.cap) return 0;
    const bool idle = hipStreamQuery(ctx->stream) == hipSuccess;
    (void)hipGetLastError();
    if (a.base) { if (idle) HIPCHK(hipFree(a.base)); else ctx->retired.push_back(a.base); }   // (enqueued kernels keep the old block: freed when the context is idle or goes away)
    a.base = nullptr; a.cap = 0;
    const size_t want = align_up(bytes + bytes / 8, 1 << 20);
    HIPCHK(hip_malloc_flush((void**)&a.base, want));
    a.cap = want;
    return 0;
}
int ensure_ntt_arena(cg_ctx* ctx, size_t bytes) { return ensure_main_stream_block(ctx, ctx->ntt_arena, bytes); }

// non-blocking timing: a pair of events per measured span, drained in cg_stats()
hipEvent_t ev_new(cg_ctx* ctx) { hipEvent_t e = nullptr; hipEventCreate(&e); return e; }
int ev_open(cg_ctx* ctx, int tag) {
    if (!ctx->stats_on) return -1;
    EvPair p;
    if (!ctx->ev_free.empty()) { p = ctx->ev_free.back(); ctx->ev_free.pop_back(); } else { p.a = ev_new(ctx); p.b = ev_new(ctx); }
    p.tag = tag;
    ctx->ev_live.push_back(p);
    return (int)ctx->ev_live.size() - 1;
}
struct StatScope {
    cg_ctx* ctx; int idx;
    StatScope(cg_ctx* c, int tag) : ctx(c), idx(ev_open(c, tag)) { if (idx >= 0) hipEventRecord(ctx->ev_live[idx].a, ctx->stream); }
    ~StatScope() { if (idx >= 0) hipEventRecord(ctx->ev_live[idx].b, ctx->stream); }
};

template <class Fn> int with_fr(int curve, Fn&& fn) {
    if (curve == CG_BN254) return fn(Bn254Fr{});
#if CG_WITH_BLS
    if (curve == CG_BLS12_381) return fn(Bls381Fr{});
#else
    if (curve == CG_BLS12_381) return fail(CG_ERR_ARG, "library built without BLS12-381 (make BLS=1)");
#endif
    return fail(CG_ERR_ARG, "unknown curve id");
}
template <class Fn> int with_fq(int curve, Fn&& fn) {
    if (curve == CG_BN254) return fn(Bn254Fq{});
#if CG_WITH_BLS
    if (curve == CG_BLS12_381) return fn(Bls381Fq{});
#else
    if (curve == CG_BLS12_381) return fail(CG_ERR_ARG, "library built without BLS12-381 (make BLS=1)");
#endif
    return fail(CG_ERR_ARG, "unknown curve id");
}
template <class Fn> int with_group(int curve, int group, Fn&& fn) {
    if (curve == CG_BN254 && group == CG_G1) return fn(Bn254Fq{}, Bn254Fr{});
    if (curve == CG_BN254 && group == CG_G2) return fn(Fp2<Bn254Fq>{}, Bn254Fr{});
#if CG_WITH_BLS
    if (curve == CG_BLS12_381 && group == CG_G1) return fn(Bls381Fq{}, Bls381Fr{});
    if (curve == CG_BLS12_381 && group == CG_G2) return fn(Fp2<Bls381Fq>{}, Bls381Fr{});
#else
    if (curve == CG_BLS12_381) return fail(CG_ERR_ARG, "library built without BLS12-381 (make BLS=1)");
#endif
    return fail(CG_ERR_ARG, "unknown curve/group id");
}

template <class F> void copy_in(F& dst, const void* src) { memcpy(dst.v, src, sizeof dst.v); }
}  // namespace

// ---- 64-bit-limb host arithmetic for the O(1) scalar multiplications of proof assembly (host_ec64.hpp)
namespace {
template <class P32, int N64> struct ModTag {
    static constexpr int N = N64;
    static const cg64::Mod<N64>& mod() { static const cg64::Mod<N64> m = [] { cg64::Mod<N64> x; x.init(P32::P); return x; }(); return m; }
};
typedef cg64::Fp<ModTag<Bn254Fq::Params, 4>> H64BnFq;
typedef cg64::Fp<ModTag<Bn254Fr::Params, 4>> H64BnFr;
#if CG_WITH_BLS
typedef cg64::Fp<ModTag<Bls381Fq::Params, 6>> H64BlsFq;
typedef cg64::Fp<ModTag<Bls381Fr::Params, 4>> H64BlsFr;
#endif
template <class Fn> int with_group64(int curve, int group, Fn&& fn) {
    if (curve == CG_BN254 && group == CG_G1) return fn(H64BnFq{}, H64BnFr{});
    if (curve == CG_BN254 && group == CG_G2) return fn(cg64::Fp2<H64BnFq>{}, H64BnFr{});
#if CG_WITH_BLS
    if (curve == CG_BLS12_381 && group == CG_G1) return fn(H64BlsFq{}, H64BlsFr{});
    if (curve == CG_BLS12_381 && group == CG_G2) return fn(cg64::Fp2<H64BlsFq>{}, H64BlsFr{});
#else
    if (curve == CG_BLS12_381) return fail(CG_ERR_ARG, "library built without BLS12-381 (make BLS=1)");
#endif
    return fail(CG_ERR_ARG, "unknown curve/group id");
}
}  // namespace

// ---- functions defined in one unit and used by another
// capi_ntt.hip: twiddle tables shared by the contexts of a device (cg_ctx_destroy gives its references back)
void shared_twiddles_release(int device, const TwKey& key);
// capi_streams.hip: a stream of priority class `cls` from the per-device pool (optionally on a measured pipe); a context's two copy streams
int pooled_stream(int device, int cls, hipStream_t* out, int want = -1);
int make_copy_streams(cg_ctx* c, int want_h2d = -1, int want_d2h = -1);

