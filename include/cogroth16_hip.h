/* cogroth16_hip.h — C ABI of the MI355X (gfx950) co-groth16 prover backend.
 *
 * This is the drop-in boundary for the ONE data-parallel hot path of TaceoLabs/collaborative-circom:
 * `CoGroth16::prove` (reference: co-circom/co-groth16/src/groth16.rs:113-326).  Every entry point below replaces an
 * arkworks call that one of the reference's MPC drivers makes from its implementation of the plug-in traits
 *   FFTProvider<F>            mpc-core/src/traits.rs:535-558
 *   MSMProvider<C>            mpc-core/src/traits.rs:561-568
 *   PrimeFieldMpcProtocol<F>  mpc-core/src/traits.rs:43-223   (the O(n) vector methods only)
 * and is what a Rust FFI shim (`extern "C"` block) would bind.  Network rounds, correlated
 * randomness (ChaCha12) and O(1) point algebra stay in the host driver, unchanged.
 *
 * DATA CONVENTIONS (identical to the reference's in-memory values, circom-types/src/traits.rs:57-67):
 *   field element  = N x u64 little-endian limbs in Montgomery form (R = 2^256; BLS12-381 Fq: R = 2^384), fully reduced.
 *                    N = 4 for BN254 Fr/Fq and BLS12-381 Fr, N = 6 for BLS12-381 Fq.
 *   G1 affine      = x || y ; G2 affine = x.c0 || x.c1 || y.c0 || y.c1.  Infinity: either a flag byte at
 *                    `infinity_offset` inside each point record (arkworks `Affine{x,y,infinity}`), or (0,0) when
 *                    infinity_offset < 0 (the packed zkey encoding, traits.rs:113-115).
 *   MSM result     = Jacobian (X, Y, Z) in Montgomery form, Z == 0 <=> infinity  (ark-ec short_weierstrass::Projective).
 *
 * ERRORS: every function returns 0 on success, non-zero otherwise; cg_last_error() gives the message (thread-local).
 *   There is NO CPU fallback: if no gfx950 device is present cg_ctx_create fails.
 * THREADING: a context is used by one thread at a time (mirrors `&mut self`); several contexts may share a GPU.
 * Pointers named d_* are DEVICE pointers (from cg_dev_alloc or any HIP allocation on the context's device);
 * pointers named h_* are HOST pointers owned by the caller.
 */
#ifndef COGROTH16_HIP_H
#define COGROTH16_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cg_ctx cg_ctx;       /* one per MPC party thread: HIP stream, scratch arena, twiddle cache */
typedef struct cg_bases cg_bases;   /* device-resident, immutable point table (a zkey query) */

enum { CG_BN254 = 0, CG_BLS12_381 = 1 };
enum { CG_G1 = 0, CG_G2 = 1 };
enum { CG_OK = 0, CG_ERR_ARG = 1, CG_ERR_HIP = 2, CG_ERR_NODEVICE = 3, CG_ERR_OOM = 4 };

/* ---- context ------------------------------------------------------------------------------------------------- */
int32_t cg_ctx_create(int32_t device, cg_ctx** out);
/* Contexts created on this thread between _begin and _end belong to ONE party (a chain context and the bulk context beside it): within each
 * priority class their streams are placed on hardware queues of their own while the class has queues left (four per class; a queue serves
 * its streams' packets in order, so a busy stream behind another busy stream's wait stands still with it).  Calls nest; without a group
 * a new context's streams go to the least used queues.  Placement only: results never depend on it. */
int32_t cg_stream_group_begin(void);
int32_t cg_stream_group_end(void);
/* flags bit 0 ("chain"): for the context that carries a dependency chain (the witness map with its party-to-party exchanges,
 * groth16.rs:141-204) while another context of the same party keeps the chip full with independent MSMs — high-priority main and copy
 * streams on hardware queues of their own.  flags bit 1 ("bulk"): the context next to it — low-priority main stream. */
int32_t cg_ctx_create_ex(int32_t device, uint32_t flags, cg_ctx** out);
int32_t cg_ctx_destroy(cg_ctx* ctx);
int32_t cg_ctx_sync(cg_ctx* ctx);
void*   cg_ctx_stream(cg_ctx* ctx);                 /* the hipStream_t every launch of this context goes to */
/* make the context launch on a stream owned by the host application (e.g. the stream its other GPU work runs on), so that
 * no cross-stream synchronisation is needed between the host's own kernels/copies and this library's. */
int32_t cg_ctx_set_stream(cg_ctx* ctx, void* hip_stream);
const char* cg_last_error(void);
const char* cg_version(void);

/* ---- device memory (thin wrappers so non-HIP hosts can keep vectors resident between calls) ------------------- */
/* cg_dev_free does not wait: the block is parked behind the work the context's streams hold at that moment and handed out again by a
 * later cg_dev_alloc (any context of the device) once that work has completed (a request of at most 1 MiB whose size is parked but not yet
 * released polls that release for up to 40 us — the tail of the previous proof of a small circuit — before asking the runtime for a new block).  Work on OTHER contexts that uses the block must have
 * completed before the call.  CG_DEV_CACHE_MB bounds the parked bytes per device (0: release at once, which waits for the device). */
int32_t cg_dev_alloc(cg_ctx* ctx, size_t bytes, void** d_ptr);
int32_t cg_dev_free(cg_ctx* ctx, void* d_ptr);
/* n blocks released together (NULL entries are skipped): one release mark behind the context's streams for all of them instead of one each —
 * what a prover does with the vectors of a finished witness map */
int32_t cg_dev_free_many(cg_ctx* ctx, void* const* d_ptrs, size_t n);
/* Gives the blocks parked by cg_dev_free on `device` back to the runtime (hipFree: stalls until the device is idle — call it when it is:
 * between proofs of different circuits, when a session closes).  Parked blocks are only reused by allocations of exactly their size;
 * without this a process that moves on to another circuit keeps up to CG_DEV_CACHE_MB of them, and once that bound is reached every
 * further cg_dev_free takes the synchronising path.  Returns the bytes released in *bytes (may be NULL). */
int32_t cg_dev_cache_trim(int32_t device, size_t* bytes);
int32_t cg_dev_upload(cg_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);     /* synchronous */
int32_t cg_dev_download(cg_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);   /* synchronous */
int32_t cg_dev_memset_zero(cg_ctx* ctx, void* d_dst, size_t bytes);
/* Several GPUs of one node: one context per device; a party's prover splits each MSM range over its contexts and
 * moves scalar slices with this copy — device to device between two contexts, enqueued on the DESTINATION context's stream behind
 * everything the source context's stream holds so far (xGMI peer copy between different devices).  cg_device_count() = visible GPUs. */
int32_t cg_dev_copy_peer(cg_ctx* dst, void* d_dst, cg_ctx* src, const void* d_src, size_t bytes);
int32_t cg_ctx_device(const cg_ctx* ctx);
int32_t cg_device_count(void);
/* Preflight of a party's device list (first contact with a multi-GPU node): distinct PCI bus ids, peer access between every pair, and one
 * 1 MiB peer copy per ordered pair checked word for word against the pattern its source was filled with.  Fails loudly (CG_ERR_ARG for a
 * repeated GPU, CG_ERR_HIP for a missing peer path or a corrupted copy; cg_last_error says which pair).  `report` (optional, NUL-terminated,
 * truncated to report_cap): one JSON object with the bus ids and every pair's peer access, copy time and rate.  Flags: ALLOW_SHARED lets a
 * device appear more than once (one-GPU tests; such pairs are local copies), ALLOW_STAGED accepts pairs without peer access.
 * cgh_session_open_multi runs it for n > 1 (include/cogroth16_host.h), bench.py --gpus N before it prints anything. */
#define CG_PREFLIGHT_ALLOW_SHARED 1u
#define CG_PREFLIGHT_ALLOW_STAGED 2u
int32_t cg_device_preflight(const int32_t* devices, int32_t n, uint32_t flags, char* report, size_t report_cap);
/* Page-locked staging buffers and asynchronous copies on the context's two copy streams, so that the MPC exchanges of mul_vec
 * (rep3.rs:650-670) and degree_reduce_vec (shamir.rs:302-384) can move in chunks under the compute.
 *   download_begin: the copy is ordered after everything enqueued on the context's stream so far.
 *   upload_begin:   after_stream != 0 orders it the same way (d_dst still in use by enqueued work); 0 = d_dst is not in use.
 *   A ticket names the copy until 256 further copies were begun on the context.  cg_copy_wait blocks the host until the copy is done;
 *   cg_copy_fence makes later launches on the context's stream wait for it without blocking the host.  Copies of one direction
 *   complete in the order they were begun.  Host buffers must come from cg_host_alloc and stay valid until the copy is done. */
int32_t cg_host_alloc(size_t bytes, void** h_ptr);
int32_t cg_host_free(void* h_ptr);
/* 1 if h_ptr lies in page-locked memory known to the runtime (cg_host_alloc or registered by the caller): such buffers can be the
 * source / destination of the asynchronous copies directly, without staging */
int32_t cg_host_is_pinned(const void* h_ptr);
int32_t cg_dev_download_begin(cg_ctx* ctx, void* h_dst_pinned, const void* d_src, size_t bytes, int32_t* ticket);
/* A download begun with cg_dev_download_begin is ordered behind EVERYTHING the context's stream holds when it is begun.  A caller that
 * streams a vector down chunk by chunk while it keeps enqueuing unrelated kernels (the mul_vec exchange under the transforms,
 * groth16.rs:174-188) marks the point where the vector was produced — cg_stream_mark, up to 16 marks alive per context — and begins the
 * chunks' downloads behind that mark: they then do not wait for the kernels enqueued after it. */
int32_t cg_stream_mark(cg_ctx* ctx, int32_t* mark);
int32_t cg_dev_download_begin_after(cg_ctx* ctx, void* h_dst_pinned, const void* d_src, size_t bytes, int32_t mark, int32_t* ticket);
int32_t cg_dev_upload_begin(cg_ctx* ctx, void* d_dst, const void* h_src_pinned, size_t bytes, int32_t after_stream, int32_t* ticket);
int32_t cg_copy_wait(cg_ctx* ctx, int32_t ticket);
int32_t cg_copy_fence(cg_ctx* ctx, int32_t ticket);

/* ---- MSMProvider::msm_public_points  (traits.rs:561-568; rep3.rs:934-947, shamir.rs:1027-1039, plain.rs:408-416) */
/* Upload a point table once (zkey a/b1/b2/l/h query, zkey.rs:48-71); it is reused by every proof.
 * `stride_bytes` = distance between records, `infinity_offset` = byte offset of the arkworks infinity flag or -1. */
int32_t cg_bases_register(cg_ctx* ctx, int32_t curve, int32_t group, const void* h_points, size_t n,
                          size_t stride_bytes, int64_t infinity_offset, cg_bases** out);
/* same, source already on the device in packed (x||y, (0,0)=inf) form; the table is copied */
int32_t cg_bases_register_device(cg_ctx* ctx, int32_t curve, int32_t group, const void* d_points_packed, size_t n, cg_bases** out);
int32_t cg_bases_release(cg_bases* bases);
/* zkey fast path: the point validation the reference's parser does per point on the CPU (circom-types/src/traits.rs:118-123,
 * 148-153 `is_on_curve`) as one device pass over a registered table.  *n_bad = number of non-infinity records off the curve,
 * *first_bad (optional) = index of the first one (UINT64_MAX if none). */
int32_t cg_bases_check_on_curve(cg_ctx* ctx, const cg_bases* bases, uint64_t* n_bad, uint64_t* first_bad);
/* second half of that validation (`is_in_correct_subgroup_assuming_on_curve`): counts the non-infinity records P with [r]P != 0,
 * r = scalar-field modulus.  Assumes the records are on the curve.  BN254 G1 has cofactor 1 and is accepted without work. */
int32_t cg_bases_check_subgroup(cg_ctx* ctx, const cg_bases* bases, uint64_t* n_bad, uint64_t* first_bad);
/* Optional, once per table: precompute 2^(c*j) * P_i for every window j (affine, resident: (254/c + 1) x the table size).
 * MSMs over such a table then use ONE bucket set for all windows: 254/c + 1 mixed additions per point with c up to 22 instead
 * of 16 at the default c = 16, and no doublings in the final fold.  The zkey queries are fixed for the life of the process
 * (zkey.rs:48-71), so this is part of registration, not of the proof.  Results are unchanged.
 * c = 0 picks the window by table size (20 above ~3 M points in G1 / ~1.5 M in G2, 16 up to 2^18 points, else 17); otherwise 8 <= c <= 22.
 * With c = 0 a table whose window copies do not fit in device memory simply stays without them (the call succeeds, MSMs over it use the
 * per-window bucket sets); an explicit c that does not fit fails with CG_ERR_OOM. */
int32_t cg_bases_precompute(cg_ctx* ctx, cg_bases* bases, int32_t c);
size_t  cg_bases_len(const cg_bases* bases);

/* out_jacobian[j] = sum_i scalars[j][i] * bases[offset + i], j < k (k = share components: REP3 2, Shamir/plain 1).
 * `offset` lets a caller pass a sub-slice exactly like `&query[1 + pub_len..]` (groth16.rs:221). n == 0 gives infinity. */
int32_t cg_msm(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n,
               const void* const* h_scalars, int32_t k, void* h_out_jacobian);
int32_t cg_msm_dev(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n,
                   const void* const* d_scalars, int32_t k, void* h_out_jacobian);
/* asynchronous form: enqueue now, collect later (lets the host overlap the O(1) Horner folds of several MSMs) */
int32_t cg_msm_dev_begin(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n,
                         const void* const* d_scalars, int32_t k, int32_t* ticket);
int32_t cg_msm_end(cg_ctx* ctx, int32_t ticket, void* h_out_jacobian);
/* Several base tables times the SAME k scalar vectors (calculate_coeff for a_query, b_g1_query, b_g2_query and the l_query MSM
 * all take `aux_assignment`, groth16.rs:251,267,284,298): the scalar decomposition and bucket sort are done once per vector and
 * reused by every table.  tables[i] is read from offsets[i] (offsets may be NULL = all zero); tickets[i] collects table i.
 * LIFETIME of d_scalars (every cg_msm_dev_begin* form): the vectors must stay untouched until cg_msm_end has returned for the call's
 * tickets — the digit / sort kernels read them on a side stream, and for small calls the context's main stream is NOT ordered behind
 * those reads (work enqueued on it after the begin call may run beside them). */
int32_t cg_msm_dev_begin_multi(cg_ctx* ctx, int32_t n_tables, const cg_bases* const* tables, const size_t* offsets, size_t n,
                               const void* const* d_scalars, int32_t k, int32_t* tickets);
/* Scalars still crossing PCIe: the NEXT cg_msm_dev_begin / _begin_multi on `ctx` lets the digit / sort schedule of share component
 * `component` (0 <= component < 4) wait on the device for the asynchronous upload `copy_ticket` of context `owner` (cg_dev_upload_begin)
 * instead of the host waiting for it — component a is scheduled and accumulated while component b is still on its way up
 * (`Rep3PrimeFieldShareVec{a, b}`, rep3/fieldshare.rs:233-236, arrives as two vectors).  Consumed by that one call; `owner` must stay alive
 * until that call has returned (the library keeps a reference to the copy's completion event, not to the context). */
int32_t cg_msm_scalars_after(cg_ctx* ctx, int32_t component, cg_ctx* owner, int32_t copy_ticket);
/* ---- environment variables.  The product needs none.  Complete list for libcogroth16_hip.so (tests/test_abi_surface.py checks it against the
 * sources): PROCESS-WIDE, none changes a result.
 *   resources     CG_DEV_CACHE_MB (32768)    device bytes parked by cg_dev_free per device before blocks are given back to the runtime (0: never park; read once)
 *                 CG_HOST_CACHE_MB (2048)    page-locked host bytes parked by cg_host_free (read once)
 *   diagnostics   CG_DEBUG_ALLOC             cg_dev_cache_trim prints how the device block cache fared since the last trim (read per call)
 *                 CG_DEBUG_STREAMS           one stderr line per stream handed out: priority class, hardware-queue slot, streams checked out per slot (read per call)
 * Everything else that was an environment variable until round 5 is now one of:
 *   * a per-context option (cg_ctx_set_option, table below) or a process-wide option (cg_set_option, table further down);
 *   * an A/B knob of the measurement scripts that exists ONLY in the planning build (make -C collaborative-circom_amd/csrc KNOBS=1 ->
 *     libcogroth16_hip_knobs.so, -DCG_DEBUG_KNOBS): CG_MSM_CHUNK, CG_MSM_CHUNK_MIN, CG_G2_CHUNK, CG_MSM_NO_ROUNDS, CG_ACC_VARIANT, CG_NO_BITSUM,
 *     CG_NO_GRID_REDUCE, CG_NTT_DIF, CG_NTT_NO_PAIR, CG_NTT_TILE, CG_BULK_CLASS, CG_NO_STREAM_PROBE, CG_NO_PIPE_MAP, the CG_MSM_* seeds of new contexts'
 *     option tables, and CG_DEBUG_NO_REDUCE (skips the bucket reductions: RESULTS ARE WRONG — what they cost a step).  The release library
 *     does not contain these names. */
/* ---- per-context tuning (never changes results).  One table instead of process-wide environment variables: every option belongs to the
 * context it is set on (a party's chain and bulk contexts differ), is read at the next call that uses it, and can be read back.
 *   option                         value                                                                                   default
 *   CG_OPT_MSM_CHUNK               entries of the sorted list one lane folds in the bucket accumulation (0 = automatic: ~128,    0
 *                                  whole residency rounds); shorter = shorter-lived workgroups, for a context whose MSMs run
 *                                  beside a dependency chain on another context                       (= cg_msm_set_chunk)
 *   CG_OPT_MSM_WINDOW              window size of tables without precomputed copies (0 = by size)     (= cg_msm_set_window)     0
 *   CG_OPT_MSM_SCATTER_CAP         see cg_msm_set_scatter_capacity                                                              -1
 *   CG_OPT_MSM_TABLE_ORDER         order of the tables of a cg_msm_dev_begin_multi call inside each share component:            0
 *                                  0 = the caller's order in every component; 1 = serpentine (odd components run the tables in
 *                                  reverse): with the G2 table LAST in the caller's order its two accumulations run back to back
 *                                  in the middle of the call — [a b1 l b2][b2 l b1 a] — after a neighbouring chain context has
 *                                  finished its transforms and before the call's tail, which then consists of G1 launches only
 *                                  2 = one launch order over (table, component) pairs, share components <= 2: the G1 pairs in serpentine
 *                                  order with the G2 pairs together after CG_OPT_MSM_G2_AFTER of them
 *   CG_OPT_MSM_G2_AFTER            with order 2: G1 accumulations launched before the first G2 one (-1 = all of them: G2 at the end)    -1
 *   CG_OPT_MSM_G2_SLICES           1 = a context with CG_OPT_MSM_CHUNK set launches a G2 accumulation one chip-load of               0
 *                                  workgroups at a time (its workgroups hold 147 of a CU's 160 KB of LDS: nothing that needs LDS,
 *                                  e.g. a transform pass of the chain context, can start while a launch lasts); costs ~2 ms per
 *                                  launch when nothing waits
 *   CG_OPT_MSM_REDUCE_BATCH        bucket sets merged and reduced together: 2 = per call and coordinate field, 1 = per share        2
 *                                  component and field, 0 = each on its own right behind its accumulation
 *   CG_OPT_MSM_ACC_SLOTS           rotating scratch slots of the accumulate / reduce pipeline (2 .. 8; batches take one per set)    4
 *   CG_OPT_MSM_WIDE_SMALL          10 .. 30 = calls of at most 2^value (point, window) entries and two share components launch all          22
 *                                  accumulations of a coordinate field side by side (one launch, one reduction batch per field; the two
 *                                  fields on two streams up to CG_OPT_MSM_OFF_MAIN_LOG entries); 1 = 2^20 entries (the bound of round 4); 0 = off
 *   CG_OPT_MSM_ONE_STREAM_LOG      MSM calls of at most 2^value (point, window) entries run schedule, accumulation and reduction in stream      0
 *                                  order on the context's main stream (0 = never: measured slower than three streams)
 *   CG_OPT_MSM_OFF_MAIN_LOG        wide calls of at most 2^value (point, window) entries accumulate their G2 sets on the context's aux          22
 *                                  stream and their G1 sets on its sort stream; the main stream stays free for the caller's next kernels
 *                                  (and is NOT ordered behind the reads of d_scalars: see cg_msm_dev_begin_multi); 0 = never
 *   CG_OPT_MSM_SOLO_LOG            wide single-field calls (one coordinate field, <= 2 components) of at most 2^value entries run as a closed   18
 *                                  sequence on the main stream with scratch of their own (the quotient MSM at the end of a SMALL proof: it
 *                                  then does not queue behind the aux call's G2 reductions); above it the call keeps the sort / accumulate /
 *                                  reduce overlap of three streams (it shared CG_OPT_MSM_OFF_MAIN_LOG's bound of 2^22 entries); 0 = never */
enum { CG_OPT_MSM_CHUNK = 1, CG_OPT_MSM_WINDOW = 2, CG_OPT_MSM_SCATTER_CAP = 3, CG_OPT_MSM_TABLE_ORDER = 4, CG_OPT_MSM_G2_SLICES = 5,
       CG_OPT_MSM_REDUCE_BATCH = 6, CG_OPT_MSM_ACC_SLOTS = 7, CG_OPT_MSM_G2_AFTER = 8, CG_OPT_MSM_WIDE_SMALL = 9, CG_OPT_MSM_ONE_STREAM_LOG = 10,
       CG_OPT_MSM_OFF_MAIN_LOG = 11, CG_OPT_MSM_SOLO_LOG = 12, CG_OPT_COUNT_ };
int32_t cg_ctx_set_option(cg_ctx* ctx, int32_t option, int64_t value);
int32_t cg_ctx_get_option(const cg_ctx* ctx, int32_t option, int64_t* value);
/* ---- process-wide options (policies that are not tied to a context; none changes a result; read at the call that uses them):
 *   option                         value                                                                                   default
 *   CG_GOPT_SUBGROUP_FULL          1 = subgroup checks by [r]P instead of the endomorphism tests (cg_bases_check_subgroup)       0
 *   CG_GOPT_COMPACT_MIN_LOG        log2 of the smallest table with >= 1/8 points at infinity that gets a compacted copy          14
 *                                  (cg_bases_register; 64 = no table does)
 *   CG_GOPT_SORT_STAGING           0 = unstaged partition / counting-sort scatters in the MSM schedule                            1
 *   CG_GOPT_SORT_SMALL             0 = small scalar vectors go through the general six-launch schedule instead of the             1
 *                                  one-workgroup kernel
 *   CG_GOPT_MSM_STAGED_OUT         1 = the sums of a bucket reduction are written to device scratch and copied to the ticket's   0
 *                                  page-locked buffer (one copy per bucket set) instead of being written there by the last kernel
 *   CG_GOPT_STREAM_PROBES          0 = NEW contexts take their streams from the pool by creation order, without the spin-kernel   1
 *                                  probes that measure which streams share a hardware queue / a pipe of the command processor (the
 *                                  probes rest on behaviour the runtime does not document: the switch for a runtime that changes it) */
enum { CG_GOPT_SUBGROUP_FULL = 1, CG_GOPT_COMPACT_MIN_LOG = 2, CG_GOPT_SORT_STAGING = 3, CG_GOPT_SORT_SMALL = 4, CG_GOPT_MSM_STAGED_OUT = 5,
       CG_GOPT_STREAM_PROBES = 6, CG_GOPT_COUNT = 7 };
int32_t cg_set_option(int32_t option, int64_t value);
int32_t cg_get_option(int32_t option, int64_t* value);
/* window size override (0 = automatic); tuning knob only, never changes results */
/* entries of the sorted list one lane folds in the bucket accumulation of THIS context's MSMs (0 = automatic: ~128, whole residency
 * rounds).  Shorter chunks = shorter-lived workgroups: for a context whose MSMs run beside a dependency chain on another context. */
int32_t cg_msm_set_chunk(cg_ctx* ctx, int32_t entries_per_lane);
int32_t cg_msm_set_window(cg_ctx* ctx, int32_t c);
/* Scalar-side schedule.  Default (cap < 0): exact two-pass counting sort.  cap == 0 selects an optimistic one-pass scatter into
 * fixed-capacity buckets sized for uniformly random scalars (secret shares); if a bucket overflows, cg_msm_end transparently
 * recomputes that MSM with the exact sort, so the scalars passed to cg_msm_dev_begin* must then stay valid until the matching
 * cg_msm_end.  cap > 0 forces a capacity (testing).  Measured equally fast on MI355X (the cost is the scattered stores, not the
 * atomics), hence opt-in.  Never changes results. */
int32_t cg_msm_set_scatter_capacity(cg_ctx* ctx, int32_t cap);

/* ---- FFTProvider::{fft,ifft}_in_place  (traits.rs:535-558; rep3.rs:893-921) ------------------------------------
 * k vectors of n = 2^j scalar-field elements, natural order in and out.  `h_group_gen` = domain.group_gen (the caller
 * may have overridden it: groth16.rs:63-70).  inverse != 0: uses group_gen^-1 and scales by n^-1.
 * `h_coset_gen` (optional, inverse only): additionally multiplies element i by coset_gen^i, i.e. fuses
 * distribute_powers_and_mul_by_const(v, g, 1) (traits.rs:177, rep3.rs:681-688) into the transform. */
int32_t cg_ntt(cg_ctx* ctx, int32_t curve, void* const* h_vecs, int32_t k, size_t n,
               const void* h_group_gen, int32_t inverse, const void* h_coset_gen);
int32_t cg_ntt_dev(cg_ctx* ctx, int32_t curve, void* const* d_vecs, int32_t k, size_t n,
                   const void* h_group_gen, int32_t inverse, const void* h_coset_gen);
/* The provers' sequence `ifft_in_place; distribute_powers_and_mul_by_const(g, 1); fft_in_place` (groth16.rs:175-188, rep3.rs:893-921,
 * :681-688) as ONE call: v <- NTT( g^i * iNTT(v)_i ), natural order in and out, same values as the two cg_ntt_dev calls.  The inverse
 * transform leaves its coefficients bit-reversed in the context's scratch and the forward transform (decimation in time) takes them from
 * there, so the two permutation passes in the middle — and their trips through HBM — do not happen. */
int32_t cg_ntt_coset_pair_dev(cg_ctx* ctx, int32_t curve, void* const* d_vecs, int32_t k, size_t n, const void* h_group_gen, const void* h_coset_gen);

/* ---- PrimeFieldMpcProtocol vector methods (device-resident operands) ------------------------------------------ */
/* add_vec traits.rs:161 / sub_assign_vec :67 / plain+Shamir local mul_vec :164 (plain.rs:219-224, shamir.rs:618-621) */
int32_t cg_vec_add_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_a, const void* d_b, size_t n);
int32_t cg_vec_sub_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_a, const void* d_b, size_t n);
int32_t cg_vec_mul_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_a, const void* d_b, size_t n);
/* REP3 mul_vec local part (rep3.rs:656-660): out = aa*ba + aa*bb + ab*ba + mask ; d_mask may be NULL */
int32_t cg_vec_rep3_mul_local_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_aa, const void* d_ab,
                                  const void* d_ba, const void* d_bb, const void* d_mask, size_t n);
/* The O(n) draws behind Rep3Rand::masking_field_element (rep3/rngs.rs:37-46; mul_vec draws n of them, rep3.rs:656-660) on the device:
 * n x `F::rand(&mut rng)` for rng = rand_chacha::ChaCha12Rng (mpc-core/src/lib.rs:10) with the given 32-byte seed (stream id 0), starting at
 * the 32-bit word position word_pos (ChaCha12Rng::get_word_pos; below 2^64) — ark-ff's rejection sampling (8 stream words per attempt, the
 * top 256 - MODULUS_BIT_SIZE bits cleared, accepted when below the modulus; the accepted bits are the Montgomery representation).
 * d_out receives the n elements in draw order; *word_pos_after is what the caller hands to ChaCha12Rng::set_word_pos so that its
 * next draw continues behind the last one taken here.  Synchronises the context's stream (the position is known only after the draws). */
int32_t cg_chacha12_fr_rand_dev(cg_ctx* ctx, int32_t curve, const uint8_t* seed32, uint64_t word_pos, size_t n, void* d_out, uint64_t* word_pos_after);
/* The same draws without the wait: _begin enqueues them on the context's stream (what follows on that stream sees d_out filled) and returns
 * a ticket (at most 8 in flight per context); _finish waits for the draw alone and reports the position.  The party's two mask vectors are
 * drawn this way before anything else of a proof is enqueued: the host goes on enqueuing while the generators' positions are still unknown
 * and asks for them before its next HOST draw (rep3.rs:595-598 follow :656-660 in the reference's order too).  The candidates are generated
 * with 12 standard deviations of surplus; a shortfall (never observed, ~1e-32 per call) makes _finish fail and leaves d_out incomplete. */
int32_t cg_chacha12_fr_rand_dev_begin(cg_ctx* ctx, int32_t curve, const uint8_t* seed32, uint64_t word_pos, size_t n, void* d_out, int32_t* ticket);
int32_t cg_chacha12_fr_rand_dev_finish(cg_ctx* ctx, int32_t ticket, uint64_t* word_pos_after);
/* distribute_powers_and_mul_by_const (traits.rs:177): v[i] *= c * g^i */
int32_t cg_vec_distribute_powers_dev(cg_ctx* ctx, int32_t curve, void* d_v, size_t n, const void* h_g, const void* h_c);
/* Single-component pointwise helpers (plain / Shamir shares, co-plonk round 2):
 *   affine:  out[i] = c * a[i] + d   (mul_with_public / add_with_public, plain.rs, shamir.rs:471-506; h_d may be NULL = 0)
 *   fill:    v[i] = value
 *   gather:  out[i] = in[offset + i * stride]                 (every 4th evaluation of a zkey polynomial, co-plonk round2.rs:196-206)
 *   prefix:  out[i] = in[0] * ... * in[i]                     (what array_prod_mul yields, round2.rs:18-41); out may equal in
 *   prefix sum: out[i] = in[0] + ... + in[i]                  (evaluate_poly_public and div_by_zerofier of co-plonk rounds 4/5 as scans)
 *   inverse: out[i] = in[i]^-1, 0 -> 0                        (inv_many; the reference errors on 0, callers check) */
int32_t cg_vec_affine_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_a, size_t n, const void* h_c, const void* h_d);
int32_t cg_vec_fill_dev(cg_ctx* ctx, int32_t curve, void* d_v, size_t n, const void* h_value);
int32_t cg_vec_gather_strided_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_in, size_t n, size_t offset, size_t stride);
/* Strided linear combination: out[out_off + i*out_stride] = sum_{j < n_terms} coeff[j] * src[j][src_off[j] + i*src_stride[j]], i < n.
 * Offsets and strides count elements; strides may be negative (reading a LIFO buffer backwards); 1 <= n_terms <= 8; h_coeffs =
 * n_terms Montgomery elements on the host.  One launch covers each step of the Shamir share algebra on device-resident vectors:
 * ShamirCore::share (shamir_core.rs:8-31), the Vandermonde step of buffer_triples (shamir.rs:904-921), the king's interpolation in
 * degree_reduce_vec (shamir.rs:330-345) and open_many (shamir.rs:581-601).  d_out must not overlap a source it reads differently. */
int32_t cg_vec_lincomb_dev(cg_ctx* ctx, int32_t curve, void* d_out, int64_t out_off, int64_t out_stride, size_t n, int32_t n_terms,
                           const void* const* d_src, const int64_t* src_off, const int64_t* src_stride, const void* h_coeffs);
int32_t cg_vec_prefix_prod_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_in, size_t n);
int32_t cg_vec_prefix_sum_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_in, size_t n);
int32_t cg_vec_inverse_dev(cg_ctx* ctx, int32_t curve, void* d_out, const void* d_in, size_t n);
/* evaluate_constraint over all rows (traits.rs:180; groth16.rs:159-166): CSR matrix, signal index < n_inputs = public.
 * party: -1 = single component (plain / Shamir), 0..2 = REP3 party id (add_with_public asymmetry, rep3.rs:600-608).
 * d_wit_b / d_out_b may be NULL when party == -1. */
int32_t cg_spmv_csr_dev(cg_ctx* ctx, int32_t curve, const uint32_t* d_row_ptr, const uint32_t* d_col, const void* d_coeff,
                        size_t n_rows, const void* d_pub, uint32_t n_inputs, int32_t party,
                        const void* d_wit_a, const void* d_wit_b, void* d_out_a, void* d_out_b);
/* host-buffer forms of the two elementwise products (what an unmodified `&mut Vec<F>` caller would use) */
int32_t cg_vec_mul(cg_ctx* ctx, int32_t curve, void* h_out, const void* h_a, const void* h_b, size_t n);
int32_t cg_vec_rep3_mul_local(cg_ctx* ctx, int32_t curve, void* h_out, const void* h_aa, const void* h_ab,
                              const void* h_ba, const void* h_bb, const void* h_mask, size_t n);

/* ---- O(1) point helpers on the host (EcMpcProtocol: scalar_mul_public_point rep3.rs:820-825, add_assign_points :780) */
/* Jacobian in/out, Montgomery. out = a + b ; out = k * a (k = Montgomery Fr) ; affine = normalise(a) (packed, (0,0)=inf) */
int32_t cg_point_add(int32_t curve, int32_t group, const void* h_a, const void* h_b, void* h_out);
int32_t cg_point_neg(int32_t curve, int32_t group, const void* h_a, void* h_out);
int32_t cg_point_scalar_mul(int32_t curve, int32_t group, const void* h_a, const void* h_k, void* h_out);
/* A base multiplied in every proof of a session (delta_1, delta_2 of groth16.rs:259-297, the generators behind the masking points of
 * rep3/rngs.rs:48-51, the public-input records of calculate_coeff groth16.rs:220): an 8-bit window table built once, then one mixed
 * addition per scalar byte.  Host arithmetic only (no device, no context); the table may be used from several threads at once. */
typedef struct cg_fixed_base cg_fixed_base;
int32_t cg_fixed_base_create(int32_t curve, int32_t group, const void* h_point_jacobian, cg_fixed_base** out);
int32_t cg_fixed_base_mul(const cg_fixed_base* table, const void* h_scalar, void* h_out_jacobian);
int32_t cg_fixed_base_destroy(cg_fixed_base* table);
int32_t cg_point_to_affine(int32_t curve, int32_t group, const void* h_a, void* h_out_affine);
int32_t cg_point_from_affine(int32_t curve, int32_t group, const void* h_affine, void* h_out);
/* What the reference checks when it deserialises a point or field elements received from a peer (ark-serialize, Validate::Yes: mpc-net's
 * recv paths, rep3/network.rs:137-176): coordinates below the modulus, on the curve, in the prime-order subgroup; limbs below the scalar
 * modulus.  Host arithmetic, for the O(1) values of a proof.  *ok = 1 / 0. */
int32_t cg_point_validate(int32_t curve, int32_t group, const void* h_affine, int32_t* ok);
int32_t cg_fr_is_canonical(int32_t curve, const void* h_elements, size_t n, int32_t* ok);
/* The same check for a vector on the device (the m-element messages of mul_vec): adds the number of elements that are not below the
 * modulus to the uint64 at d_count (device memory, zeroed by the caller); enqueued on the context's stream, 32 B read per element. */
int32_t cg_vec_check_canonical_dev(cg_ctx* ctx, int32_t curve, const void* d_vec, size_t n, void* d_count);
/* O(1) scalar-field helpers used by the host drivers (Montgomery in/out): op 0 add, 1 sub, 2 mul, 3 inverse(a) */
int32_t cg_fr_op(int32_t curve, int32_t op, const void* h_a, const void* h_b, void* h_out);
/* canonical little-endian integers (wtns values, circom-types/src/witness.rs:51-91) <-> Montgomery form; n elements, host.
 * cg_fr_from_canonical reduces mod r first (from_le_bytes_mod_order, traits.rs:50-54). */
int32_t cg_fr_from_canonical(int32_t curve, const void* h_in, void* h_out, size_t n);
int32_t cg_fr_to_canonical(int32_t curve, const void* h_in, void* h_out, size_t n);
/* base-field coordinates (32 B BN254 / 48 B BLS12-381 each), Montgomery <-> canonical little-endian: what the JSON encodings of
 * proofs and verification keys carry as decimal strings (circom-types/src/traits.rs:186-233).  from_canonical rejects values >= q. */
int32_t cg_fq_to_canonical(int32_t curve, const void* h_in, void* h_out, size_t n);
int32_t cg_fq_from_canonical(int32_t curve, const void* h_in, void* h_out, size_t n);
/* generator of G1/G2 as a Jacobian point (ark-bn254 / ark-bls12-381 constants) */
int32_t cg_point_generator(int32_t curve, int32_t group, void* h_out);

/* ---- tooling (bench / tests; not on the prover path) ----------------------------------------------------------- */
/* Builds, on the device, the table [(first + i) * G]_{i<n} of consecutive multiples of the group generator: valid,
 * pairwise distinct points with known discrete logs, used as synthetic zkey-sized bases. */
int32_t cg_bases_synth_multiples(cg_ctx* ctx, int32_t curve, int32_t group, uint64_t first, size_t n, cg_bases** out);
/* Fixed-base batch multiplication on the device: table[i] = s_i * G for n device-resident Montgomery scalars (G = the group
 * generator of cg_point_generator).  A zero scalar gives the point at infinity.  Builds the queries of a synthetic but VALID
 * Groth16 CRS from toxic-waste polynomial evaluations (SURVEY.md §8d "synthetic R1CS generator"; the reference has no setup
 * code — snarkjs produces its zkeys — so this replaces nothing on the prover path). */
int32_t cg_bases_from_scalars(cg_ctx* ctx, int32_t curve, int32_t group, const void* d_scalars, size_t n, cg_bases** out);
/* copies n packed affine points of a table back to the host */
int32_t cg_bases_download(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n, void* h_out_packed);

/* ---- statistics: GPU time per kernel class since the last reset, from HIP event pairs recorded on the context's stream
 * (non-blocking while running; cg_stats() synchronises the stream and drains them).  *_ms are sums, *_calls are counts. */
typedef struct cg_stage_times {
    double msm_ms, ntt_ms, vec_ms, spmv_ms;                 /* whole calls */
    uint64_t msm_calls, ntt_calls, vec_calls, spmv_calls;
    double msm_sort_ms, msm_acc_g1_ms, msm_acc_g2_ms, msm_reduce_ms;   /* inside the MSM: digits+scan+scatter / bucket accumulation / bucket reduction */
    uint64_t msm_sort_calls, msm_acc_g1_calls, msm_acc_g2_calls, msm_reduce_calls;
} cg_stage_times;
int32_t cg_stats_enable(cg_ctx* ctx, int32_t on);
int32_t cg_stats(cg_ctx* ctx, cg_stage_times* out, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* COGROTH16_HIP_H */
