// C ABI of the gfx950 co-groth16 backend (include/cogroth16_hip.h) — MSM launch logic (schedules, accumulations, reductions over the context's streams) and the cg_msm_* entry points
#include "capi_internal.hpp"

namespace {
// ------------------------------------------------------------------------------------------------ MSM
// Window size of the classic path (one bucket set per window), measured on MI355X for both groups (scripts/sweep_classic_window.py):
// what matters besides the add count is that the TOP window is nearly full — with bits = c*q + t it has only t (+1 carry) bits, all n
// entries of that window fall into 2^t buckets, and a tiny t (c = 14: t = 2) leaves a few huge buckets whose pieces are merged by
// few lanes.  c = 8 (t = 6), 13 (t = 7), 15 (t = 14) and 16 (t = 14) are the good choices for 254/255-bit scalars:
//   n <= 2^12: 8   |   2^13: 13   |   2^14 .. 2^18: 15   |   larger: 16        (2^16 points: 1.75 ms at c = 15 against 5.7 ms at c = 8 or 11)
int auto_window(size_t n, int bits) {
    const int lg = log2_floor(std::max<size_t>(n, 1));
    int c = lg <= 12 ? 8 : lg == 13 ? 13 : lg <= 18 ? 15 : 16;
    auto ok = [&](int w) { const int t = bits % w; return t != 0 && t >= w - 3 - (w >= 13 ? 6 : 0); };   // other scalar sizes: nudge to a window with a usable top
    if (!ok(c)) for (int d : {1, -1, 2, -2, 3, -3}) { if (c + d >= 3 && c + d <= 17 && ok(c + d)) { c += d; break; } }
    return c;
}

template <class F>
Jacobian<F> msm_fold_windows(const XYZZ<F>* w, int nwin, int c) {
    XYZZ<F> acc = w[nwin - 1];
    for (int i = nwin - 2; i >= 0; i--) {
        for (int d = 0; d < c; d++) acc = xyzz_dbl(acc);
        acc = xyzz_add(acc, w[i]);
    }
    return xyzz_to_jacobian(acc);
}

int ticket_slot(cg_ctx* ctx) {
    for (size_t i = 0; i < ctx->tickets.size(); i++) if (!ctx->tickets[i].live) return (int)i;
    ctx->tickets.emplace_back();
    return (int)ctx->tickets.size() - 1;
}

template <class Fn> int with_coord_field(int curve, int group, Fn&& fn) {   // group-only dispatch (the scalar field is fixed by the curve)
    return with_group(curve, group, [&](auto ftag, auto) -> int { return fn(ftag); });
}

// One digit/sort schedule per scalar vector, then one accumulate+reduce per base table: `nb` tables (same curve, any groups)
// multiplied by the SAME k scalar vectors.  tickets_out[b] collects the k results for table b.
int msm_begin_multi_impl_(cg_ctx* ctx, int nb, const cg_bases* const* bases, const size_t* offsets, size_t n, const void* const* d_scalars, int k, int* tickets_out, bool force_exact);
int msm_begin_multi_impl(cg_ctx* ctx, int nb, const cg_bases* const* bases, const size_t* offsets, size_t n, const void* const* d_scalars, int k, int* tickets_out, bool force_exact = false) {
    return msm_begin_multi_impl_(ctx, nb, bases, offsets, n, d_scalars, k, tickets_out, force_exact);
}
int msm_begin_multi_impl_(cg_ctx* ctx, int nb, const cg_bases* const* bases, const size_t* offsets, size_t n, const void* const* d_scalars, int k, int* tickets_out, bool force_exact) {
    const uint32_t chunk_request = ctx ? ctx->msm_chunk : 0;   // cg_msm_set_chunk: handed to every geometry computation of this call
    if (!ctx || !bases || !tickets_out || (n && !d_scalars)) return fail(CG_ERR_ARG, "null argument");
    if (nb < 1 || nb > 16) return fail(CG_ERR_ARG, "number of base tables out of range");
    if (k < 1 || k > 8) return fail(CG_ERR_ARG, "k out of range");
    for (int b = 0; b < nb; b++) {
        if (!bases[b]) return fail(CG_ERR_ARG, "null bases");
        if ((offsets ? offsets[b] : 0) + n > bases[b]->n) return fail(CG_ERR_ARG, "bases slice out of range");
        if (bases[b]->device != ctx->device) return fail(CG_ERR_ARG, "bases live on another device");
        if (bases[b]->curve != bases[0]->curve) return fail(CG_ERR_ARG, "all tables of one call must be on the same curve");
    }
    HIPCHK(hipSetDevice(ctx->device));
    {   // tables with a compacted copy: map (offset, n) into the compacted index space, gather the scalars, and run the groups of tables
        // that ended up with the same scalar set (same infinity pattern and range, e.g. b_g1_query and b_g2_query) as one schedule each
        bool any = false;
        for (int b = 0; b < nb; b++) any = any || bases[b]->compact != nullptr;
        if (any) {
            struct Grp { uint64_t sig; size_t off, cnt, caller_off; std::vector<int> members; };
            std::vector<Grp> groups;
            std::vector<size_t> off_c(nb), cnt_c(nb);
            for (int b = 0; b < nb; b++) {
                const size_t off = offsets ? offsets[b] : 0;
                uint64_t sig = 0; size_t o = off, cn = n;
                if (bases[b]->compact) {
                    const auto& lv = bases[b]->h_live;
                    o = (size_t)(std::lower_bound(lv.begin(), lv.end(), (uint32_t)off) - lv.begin());
                    cn = (size_t)(std::lower_bound(lv.begin(), lv.end(), (uint32_t)std::min<size_t>(off + n, 0xffffffffu)) - lv.begin()) - o;
                    sig = bases[b]->live_sig;
                }
                off_c[b] = o; cnt_c[b] = cn;
                bool placed = false;
                // one gather serves a group: same caller offset (the gather's index base), same compacted range, and the same live
                // indices inside it — compared element by element, the 64-bit signature only short-cuts the mismatch
                for (auto& g : groups) {
                    if (g.sig != sig || g.cnt != cn) continue;
                    if (sig != 0) {
                        if (g.off != o || g.caller_off != off) continue;
                        const auto& la = bases[g.members[0]]->h_live; const auto& lb = bases[b]->h_live;
                        if (memcmp(la.data() + o, lb.data() + o, cn * sizeof(uint32_t)) != 0) continue;
                    }
                    g.members.push_back(b); placed = true; break;
                }
                if (!placed) groups.push_back(Grp{sig, o, cn, off, {b}});
            }
            size_t need = 0;
            for (auto& g : groups) if (g.sig) need += align_up((size_t)k * g.cnt * 32);
            if (need > ctx->gather_cap) {
                HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->sortst));
                if (ctx->gather_buf) HIPCHK(hipFree(ctx->gather_buf));
                ctx->gather_buf = nullptr; ctx->gather_cap = 0;
                HIPCHK(hip_malloc_flush(&ctx->gather_buf, need)); ctx->gather_cap = need;
            }
            // gather_buf is rewritten from offset 0 by this call: an off-main call before it (no cg_msm_end in between) may still be reading it in
            // its digit / sort kernels, which the main stream no longer waits for — a stream wait, no host stall
            if (ctx->sorts_unordered) { for (int j = 0; j < 2; j++) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_sorted[j], 0)); ctx->sorts_unordered = false; }
            size_t used = 0;
            for (auto& g : groups) {
                std::vector<const cg_bases*> gb; std::vector<size_t> go; std::vector<const void*> gs(k);
                const int first = g.members[0];
                for (int m : g.members) { gb.push_back(bases[m]->compact ? bases[m]->compact : bases[m]); go.push_back(bases[m]->compact ? off_c[m] : (offsets ? offsets[m] : 0)); }
                if (g.sig) {
                    const size_t off = offsets ? offsets[first] : 0;
                    for (int j = 0; j < k; j++) {
                        void* dst = (char*)ctx->gather_buf + used + (size_t)j * g.cnt * 32;
                        int rc = with_fr(bases[first]->curve, [&](auto tag) -> int {
                            typedef decltype(tag) Fr;
                            return launch_vec_gather_idx<Fr>(ctx->stream, (Fr*)dst, (const Fr*)d_scalars[j], bases[first]->d_live + g.off, g.cnt, (uint32_t)off);
                        });
                        if (rc) return rc;
                        gs[j] = dst;
                    }
                    used += align_up((size_t)k * g.cnt * 32);
                } else for (int j = 0; j < k; j++) gs[j] = d_scalars[j];
                std::vector<int> tk(g.members.size());
                int rc = msm_begin_multi_impl(ctx, (int)g.members.size(), gb.data(), go.data(), g.cnt, gs.data(), k, tk.data(), true);
                if (rc) return rc;
                for (size_t i = 0; i < g.members.size(); i++) tickets_out[g.members[i]] = tk[i];
            }
            return 0;
        }
    }
    {   // a schedule depends on the window: tables precomputed with different windows (the automatic choice differs between G1 and
        // G2 for 1.5-3 M points), or a mix of precomputed and plain tables, run as one sub-call per window
        bool mixed = false;
        for (int b = 1; b < nb; b++) mixed = mixed || bases[b]->pre_c != bases[0]->pre_c;
        if (mixed) {
            std::vector<char> done(nb, 0);
            for (int b = 0; b < nb; b++) {
                if (done[b]) continue;
                std::vector<const cg_bases*> gb; std::vector<size_t> go; std::vector<int> idx;
                for (int m = b; m < nb; m++) if (!done[m] && bases[m]->pre_c == bases[b]->pre_c) { gb.push_back(bases[m]); go.push_back(offsets ? offsets[m] : 0); idx.push_back(m); done[m] = 1; }
                std::vector<int> tk(idx.size());
                int rc = msm_begin_multi_impl(ctx, (int)idx.size(), gb.data(), go.data(), n, d_scalars, k, tk.data(), force_exact);
                if (rc) return rc;
                for (size_t i = 0; i < idx.size(); i++) tickets_out[idx[i]] = tk[i];
            }
            return 0;
        }
    }
    const int curve = bases[0]->curve;
    const bool shared = bases[0]->pre_c != 0;          // per-window precomputed tables: one bucket set for all windows
    if (shared && n > ((size_t)1 << 24)) return fail(CG_ERR_ARG, "precomputed-table MSM supports at most 2^24 points per call");
    int bits = 0;
    { int rc = with_fr(curve, [&](auto tag) -> int { bits = decltype(tag)::Params::BITS; return 0; }); if (rc) return rc; }
    const int c = shared ? bases[0]->pre_c : (n ? (ctx->msm_window ? ctx->msm_window : auto_window(n, bits)) : 2);
    const int nwin = bits / c + 1;
    if (shared && nwin != bases[0]->pre_nwin) return fail(CG_ERR_ARG, "internal: window count mismatch");
    if ((uint64_t)nwin * n >= ((uint64_t)1 << 32))        // schedule positions are 32-bit
        return fail(CG_ERR_ARG, "MSM of more than 2^32 / windows points in one call (about 2^27): pass the table in slices and add the partial sums");
    // optimistic scatter capacity: expected heaviest bucket (regular windows + the narrower top window) + 25 % + 6 sigma
    uint32_t cap = 0;
    if (n && !force_exact && ctx->scatter_cap >= 0) {
        if (ctx->scatter_cap > 0) cap = (uint32_t)ctx->scatter_cap;
        else {
            const int t = bits % c;
            const double nbk = (double)((size_t)1 << (c - 1));
            const double top = t == 0 ? (double)n : (double)n / (double)((size_t)1 << std::min(c - 1, t));
            const double avg = shared ? (double)(nwin - 1) * (double)n / nbk + top : std::max((double)n / nbk, top);
            const double want = 1.25 * avg + 6.0 * std::sqrt(avg) + 16.0;
            if (want <= 4096.0) { cap = 16; while ((double)cap < want) cap <<= 1; }
        }
    }
    const MsmGeom geom = msm_geom(std::max<size_t>(n, 1), c, nwin, shared, 0, chunk_request);   // what is read here (sums per component, reduction kind) does not depend on the chunking
    const int nsums = geom.ngroups;
    // tickets + pinned result buffers
    std::vector<int> slots(nb);
    size_t acc_bytes = 0;
    for (int b = 0; b < nb; b++) {
        slots[b] = ticket_slot(ctx);
        MsmTicket& t = ctx->tickets[slots[b]];
        t.live = true;   // reserve before asking for the next slot
        t.curve = curve; t.group = bases[b]->group; t.k = k; t.c = c; t.nwin = nwin; t.nsums = nsums; t.plain_fold = shared && !geom.bitsum && !geom.grid; t.bit_fold = geom.bitsum;
        t.grid_fold = geom.grid; t.log_l = geom.log_l; t.log_h = geom.log_h; t.gc = geom.gc; t.gr = geom.gr;
        t.optimistic = cap != 0; t.bases = bases[b]; t.offset = offsets ? offsets[b] : 0; t.n = n; t.scalars.assign(d_scalars, d_scalars + (n ? k : 0));
        if (!t.h_flags) HIPCHK(hipHostMalloc((void**)&t.h_flags, 8 * sizeof(uint32_t), hipHostMallocDefault));
        for (int i = 0; i < 8; i++) t.h_flags[i] = 0;
        int rc = with_coord_field(curve, t.group, [&](auto ftag) -> int {
            typedef decltype(ftag) F;
            const size_t need = (size_t)k * nsums * sizeof(XYZZ<F>);
            if (t.pinned_bytes < need) {
                if (t.h_pinned) HIPCHK(hipHostFree(t.h_pinned));
                t.h_pinned = nullptr; t.pinned_bytes = 0;
                HIPCHK(hipHostMalloc(&t.h_pinned, need, hipHostMallocDefault));
                t.pinned_bytes = need;
            }
            if (n == 0) { XYZZ<F>* h = (XYZZ<F>*)t.h_pinned; for (int i = 0; i < k * nsums; i++) h[i] = XYZZ<F>::infinity(); }
            else acc_bytes = std::max(acc_bytes, msm_acc_scratch_bytes<F>(n, c, nwin, shared, chunk_request));
            return 0;
        });
        if (rc) return rc;
        if (!t.done) HIPCHK(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
    }
    if (n) {
        StatScope ss(ctx, TAG_MSM);
        const size_t sort_bytes = align_up(cap ? msm_sort_direct_scratch_bytes(n, c, nwin, shared ? 1 : 0, cap) : msm_sort_scratch_bytes(n, c, nwin));
        const size_t acc_slot = align_up(acc_bytes);
        const int nsched = k > 1 ? 2 : 1;                  // two schedule slots so that sort j+1 overlaps accumulate j
        // Four rotating scratch slots: an accumulation waits for the bucket reduction that used its slot, and beside the accumulations the
        // reduction chain of one MSM (merge, segment sums, window sums; 1 ms alone) takes 3-8 ms — with two slots the main stream stalled
        // on it (2^22 step: 71.0 -> 69.95 ms with four, no further gain with six or eight; CG_ACC_SLOTS = 2 .. 8 for A/B runs)
        const int acc_slots_min = std::min((int)cg_ctx::ACC_SLOTS_MAX, std::max(2, ctx->acc_slots));
        // Reduction batching (CG_OPT_MSM_REDUCE_BATCH): 2 (default) = the bucket sets of a call that share a coordinate field are merged and reduced TOGETHER,
        // after the last accumulation of that field in the call (with the G2 table first in every component, the G2 sets go while the
        // last component's G1 tables are still accumulated; only the G1 batch trails the call); 1 = per share component; 0 = every set on
        // its own right behind its accumulation (rounds 1-3).  Beside lock-stepped accumulations a reduction costs the step its stand-alone
        // duration whatever its width: 2^22 step with ten reductions 6.2 ms, with three.  A batch holds its sets' scratch slots until it has run: one slot per set.
        const int red_batch = ctx->red_batch;
        // CG_OPT_MSM_WIDE_SMALL: 0 = off, 1 = calls of at most 2^20 (point, window) entries, 10 .. 30 = log2 of that bound
        const uint64_t wide_max = ctx->wide_small == 0 ? 0 : (uint64_t)1 << (ctx->wide_small == 1 ? 20 : ctx->wide_small);
        const bool small_call = k <= 2 && (uint64_t)nwin * n <= wide_max;            // see `wide` below
        // A/B knob CG_MSM_ONE_STREAM_LOG (off by default): tiny calls with schedule, accumulation and reduction in stream order on the MAIN stream.
        // It takes the context's hardware-queue placement out of the picture — the same Poseidon-fixture party takes 1.9 to 3.7 ms from one
        // session of a process to the next with three streams, 2.5-2.8 ms with one — but the G2 reduction then no longer runs under the G1
        // accumulation, and the best placement is what the default keeps (profiles/r05_small_circuit_ab3.txt).
        // ... except for a tiny call whose tables all lie in ONE coordinate field (the quotient's MSM at the end of a small proof): nothing would run
        // beside anything, and on the main stream — another priority class than the side streams, so never on their hardware queues — its
        // schedule, accumulation and reduction do not queue behind the G2 reductions of the call before (the same Poseidon party waited 14 or
        // 200 us for this result, depending on where the two side streams had landed)
        bool single_field = true;
        for (int b = 1; b < nb; b++) single_field = single_field && bases[b]->group == bases[0]->group;
        const int acc_slots = red_batch || small_call ? std::min((int)cg_ctx::ACC_SLOTS_MAX, std::max(acc_slots_min, red_batch >= 2 || small_call ? nb * k : nb + 1)) : acc_slots_min;
        const bool wide = k <= 2 && nb * k <= std::min(acc_slots, (int)ACC_MAX_SETS) && (uint64_t)nwin * n <= wide_max;      // see the WIDE mode below
        // `solo`: such a call is a closed sequence on ONE stream — it takes its scratch from a block of its own (ordered by that stream alone) and
        // leaves the context's cross-stream bookkeeping (slot / schedule events of the shared arena) untouched: it neither waits for the
        // reductions of the call before, which still read the shared arena, nor hides them from the call after
        const bool solo = wide && small_call && single_field && ctx->solo_log > 0 && (uint64_t)nwin * n <= ((uint64_t)1 << ctx->solo_log);
        const bool one_stream = solo || (small_call && ctx->one_stream_log > 0 && (uint64_t)nwin * n <= ((uint64_t)1 << ctx->one_stream_log));
        const hipStream_t sortst = one_stream ? ctx->stream : ctx->sortst, auxst = one_stream ? ctx->stream : ctx->aux;
        { int rc = solo ? ensure_main_stream_block(ctx, ctx->solo_arena, nsched * sort_bytes + (size_t)acc_slots * acc_slot) : ensure_arena(ctx, nsched * sort_bytes + (size_t)acc_slots * acc_slot); if (rc) return rc; }
        char* const arena_base = solo ? ctx->solo_arena.base : ctx->arena.base;
        char* acc_scratch = arena_base + nsched * sort_bytes;
        if (!solo) {
        HIPCHK(hipEventRecord(ctx->ev_in, ctx->stream));   // scalars (and the arena) are ready once the main stream gets here
        HIPCHK(hipStreamWaitEvent(sortst, ctx->ev_in, 0));
        for (int rs = 0; rs < 2; rs++) for (int i = 0; i < 2; i++) if (ctx->merged_pending[rs][i]) { HIPCHK(hipStreamWaitEvent(sortst, ctx->ev_merged[rs][i], 0)); ctx->merged_pending[rs][i] = false; }   // ... and the previous call's merges have read the old schedules
        }
        std::vector<MsmSortPtrs> sps(k);
        auto launch_sort = [&](int j) -> int {             // scalar side: once per scalar vector, on the sort stream
            const int ss_ = j % nsched;
            if (j < 4 && ctx->comp_after[j]) HIPCHK(hipStreamWaitEvent(sortst, ctx->comp_after[j], 0));   // this component's scalars are still on their way up
            if (j >= nsched && !solo) {                          // accumulates and merges of component j-2 have consumed the slot
                HIPCHK(hipStreamWaitEvent(sortst, ctx->ev_sched_free[ss_], 0));
                for (int rs = 0; rs < 2; rs++) if (ctx->merged_pending[rs][ss_]) HIPCHK(hipStreamWaitEvent(sortst, ctx->ev_merged[rs][ss_], 0));
            }
            hipEvent_t evs[2]; hipEvent_t* pev = nullptr;
            if (ctx->stats_on) { const int i0 = ev_open(ctx, TAG_SORT); evs[0] = ctx->ev_live[i0].a; evs[1] = ctx->ev_live[i0].b; pev = evs; }
            char* sort_scratch = arena_base + (size_t)ss_ * sort_bytes;
            int rc = with_fr(curve, [&](auto tag) -> int {
                typedef decltype(tag) Fr;
                return cap ? msm_sort_direct_launch<Fr>(sortst, (const Fr*)d_scalars[j], n, c, nwin, shared ? 1 : 0, cap, sort_scratch, &sps[j], pev)
                           : msm_sort_launch<Fr>(sortst, (const Fr*)d_scalars[j], n, c, nwin, shared ? 1 : 0, sort_scratch, &sps[j], pev);
            });
            if (rc) return rc;
            if (cap) for (int b = 0; b < nb; b++) HIPCHK(hipMemcpyAsync(ctx->tickets[slots[b]].h_flags + j, sps[j].overflow, 4, hipMemcpyDeviceToHost, sortst));
            if (!solo) HIPCHK(hipEventRecord(ctx->ev_sorted[ss_], sortst));
            return 0;
        };
        int iter = 0;
        { int rc = launch_sort(0); if (rc) return rc; }
        // bucket sets accumulated but not yet merged / reduced, by coordinate field (group) of their table
        struct PendSet { MsmRedSet set; int slot, sched, table, comp; };
        std::vector<PendSet> pend[2];
        int tables_of_group[2] = {0, 0};
        for (int b = 0; b < nb; b++) tables_of_group[bases[b]->group == CG_G1 ? 0 : 1]++;
        std::vector<int> comps_left(nb, k);
        hipStream_t red_stream[2] = {auxst, auxst};      // reduction stream per field (wide mode: G1 on the idle sort stream, beside G2 on aux)
        hipStream_t acc_stream[2] = {ctx->stream, ctx->stream};   // accumulation stream per field (the main stream, except for tiny wide calls: see `off_main`)
        hipEvent_t last_acc[2] = {nullptr, nullptr};      // behind the last accumulation of a field's flushed batch
        auto flush = [&](int gi) -> int {
            std::vector<PendSet>& pd = pend[gi];
            if (pd.empty()) return 0;
            hipStream_t rst = red_stream[gi];
            // every accumulation of the batch sits on the main stream in front of this point: the reduction stream waits for the last one
            hipEvent_t ea = ctx->ev_acc[pd.back().slot];
            if (!solo) {
                HIPCHK(hipEventRecord(ea, acc_stream[gi]));
                if (rst != acc_stream[gi]) HIPCHK(hipStreamWaitEvent(rst, ea, 0));
                last_acc[gi] = ea;
            }
            hipEvent_t evs[2]; hipEvent_t* pev = nullptr;
            if (ctx->stats_on) { const int i2 = ev_open(ctx, TAG_REDUCE); evs[0] = ctx->ev_live[i2].a; evs[1] = ctx->ev_live[i2].b; pev = evs; }
            hipEvent_t evm[2]; int nm = 0; bool seen[2] = {false, false};
            std::vector<MsmRedSet> sets;
            const int rs = rst == sortst ? 1 : 0;
            for (const PendSet& ps : pd) { sets.push_back(ps.set); if (!solo && !seen[ps.sched]) { seen[ps.sched] = true; evm[nm++] = ctx->ev_merged[rs][ps.sched]; } }
            int rc = with_coord_field(curve, gi == 0 ? CG_G1 : CG_G2, [&](auto ftag) -> int {
                typedef decltype(ftag) F;
                return msm_reduce_batch<F>(rst, sets.data(), (int)sets.size(), n, c, nwin, shared, sps[pd[0].comp].cap, evm, nm, pev, chunk_request);
            });
            if (rc) return rc;
            if (!solo) for (const PendSet& ps : pd) {
                HIPCHK(hipEventRecord(ctx->ev_red[ps.slot], rst));
                ctx->slot_busy[ps.slot] = true; ctx->aux_pending = true; ctx->last_slot = ps.slot; ctx->merged_pending[rs][ps.sched] = true;
            }
            // a table's results are complete when the batch holding its LAST outstanding component has run (components may sit in different batches)
            for (const PendSet& ps : pd) if (--comps_left[ps.table] == 0) HIPCHK(hipEventRecord(ctx->tickets[slots[ps.table]].done, rst));
            pd.clear();
            return 0;
        };
        auto acc_set = [&](int b, int j, char* scratch) -> MsmAccSet {
            const char* pts = (const char*)(shared ? bases[b]->d_pre : bases[b]->d_pts) + (offsets ? offsets[b] : 0) * bases[b]->pt_bytes;
            return MsmAccSet{pts, shared ? bases[b]->n : 0, sps[j].sorted, sps[j].offsets, sps[j].counts, scratch, !bases[b]->no_inf};
        };
        auto red_set = [&](int b, int j, char* scratch) -> MsmRedSet {
            const MsmTicket& t = ctx->tickets[slots[b]];
            const size_t pinned_stride = (size_t)(t.group == CG_G1 ? 4 : 8) * (bases[b]->pt_bytes / (t.group == CG_G1 ? 2 : 4));     // sizeof(XYZZ<F>): four coordinates
            return MsmRedSet{scratch, sps[j].offsets, sps[j].counts, (char*)t.h_pinned + (size_t)j * nsums * pinned_stride};
        };
        // WIDE mode (small calls, <= 2 share components, one scratch slot per set): all accumulations of a coordinate field in ONE launch
        // (blockIdx.y = table x component), the G2 launch first and its reduction on the aux stream while the G1 launch runs, whose
        // reduction goes to the then idle sort stream.  A 2^16-point launch is 256 workgroups and lasts as long as one lane's chain of
        // additions; eight in a row cost eight chains (2^16 step: 3.2 ms), side by side one.
        // (measured, round 4: 2^14 step 2.63 -> 2.14 ms, 2^16 3.30 -> 3.09 ms and one REP3 party 5.85 -> 5.54 ms; from 2^17 points on — 2^21 entries — no gain)
        // Wide calls up to CG_MSM_OFF_MAIN_LOG entries (default 2^22: 2^18 points) keep the main stream free: the G2 sets are accumulated on the aux stream
        // and the G1 sets on the sort stream, each in front of its own reduction, and the main stream only marks where the scalars are
        // ready.  Such a call fills a fraction of the chip, so nothing is gained by queueing the caller's next kernels behind its accumulations
        // — a one-context party's witness map (a chain of short kernels and two host round trips) started 0.3 ms late behind the
        // witness-independent MSMs, and later still whenever their streams had fallen onto a shared hardware queue.  Beside a chain context the
        // gain is the two fields' accumulations running side by side instead of one after the other (one REP3 party, bounds 2^20 / 2^19 -> 2^22 / 2^22
        // entries for wide / off-main: 2^16 3.23 -> 2.78 ms, 2^17 4.97 -> 4.50, 2^18 7.3 -> 6.9.  Not beyond: with 2^24 the 2^19 / 2^20 parties stand
        // at 12.1 -> 12.0 / 21.8 -> 21.2 ms, but a party over four devices (2^20-point slices) goes from 23.0 to 24.5 ms, and 2^21 / 2^22 lose 0.3 / 1.8 ms).
        const bool off_main = wide && !one_stream && ctx->off_main_log > 0 && (uint64_t)nwin * n <= ((uint64_t)1 << ctx->off_main_log);
        if (wide) {
            if (k == 2) { int rc = launch_sort(1); if (rc) return rc; }
            if (off_main) { acc_stream[0] = sortst; acc_stream[1] = auxst; }
            if (!solo) for (int j = 0; j < k; j++) HIPCHK(hipStreamWaitEvent(acc_stream[1], ctx->ev_sorted[j], 0));      // (main stream, or aux; the sort stream is behind its own sorts anyway)
            red_stream[0] = sortst;
            for (int gi : {1, 0}) {
                std::vector<MsmAccSet> sets;
                for (int j = 0; j < k; j++) for (int b = 0; b < nb; b++) {
                    if ((bases[b]->group == CG_G1 ? 0 : 1) != gi) continue;
                    const int slot = iter++ % acc_slots;
                    if (!solo && ctx->slot_busy[slot]) HIPCHK(hipStreamWaitEvent(acc_stream[gi], ctx->ev_red[slot], 0));
                    char* scratch = acc_scratch + (size_t)slot * acc_slot;
                    sets.push_back(acc_set(b, j, scratch));
                    pend[gi].push_back(PendSet{red_set(b, j, scratch), slot, j, b, j});
                }
                if (sets.empty()) continue;
                hipEvent_t evs[2]; hipEvent_t* pev = nullptr;
                if (ctx->stats_on) { const int i1 = ev_open(ctx, gi == 0 ? TAG_ACC_G1 : TAG_ACC_G2); evs[0] = ctx->ev_live[i1].a; evs[1] = ctx->ev_live[i1].b; pev = evs; }
                int rc = with_coord_field(curve, gi == 0 ? CG_G1 : CG_G2, [&](auto ftag) -> int {
                    typedef decltype(ftag) F;
                    return msm_accumulate_batch<F>(acc_stream[gi], sets.data(), (int)sets.size(), n, c, nwin, shared, sps[0].cap, pev, chunk_request, false);
                });
                if (rc) return rc;
                if (gi == 0 && !solo) HIPCHK(hipStreamWaitEvent(sortst, ctx->ev_sorted[k - 1], 0));     // (the sort stream has nothing else left in this call)
                { int rc2 = flush(gi); if (rc2) return rc2; }
            }
            // the schedules are free once every accumulation has read them: behind them all on the main stream, or (off the main stream) on the
            // sort stream, which holds the G1 accumulations itself and waits here for the G2 ones
            if (off_main && last_acc[1]) HIPCHK(hipStreamWaitEvent(sortst, last_acc[1], 0));
            if (off_main) ctx->sorts_unordered = true;
            if (!solo) for (int j = 0; j < k; j++) HIPCHK(hipEventRecord(ctx->ev_sched_free[j], off_main ? sortst : ctx->stream));
        }
        // one accumulation: table b, share component j, into the next rotating scratch slot; its bucket set joins the batch of its field
        auto do_acc = [&](int b, int j) -> int {
            const MsmSortPtrs& sp = sps[j];
            MsmTicket& t = ctx->tickets[slots[b]];
            const int gi = t.group == CG_G1 ? 0 : 1;
            hipEvent_t evs[2]; hipEvent_t* pev = nullptr;
            if (ctx->stats_on) { const int i1 = ev_open(ctx, t.group == CG_G1 ? TAG_ACC_G1 : TAG_ACC_G2); evs[0] = ctx->ev_live[i1].a; evs[1] = ctx->ev_live[i1].b; pev = evs; }
            const int slot = iter++ % acc_slots;
            for (int g2 = 0; g2 < 2; g2++) {                 // the slot still holds a set that waits for its batch: run that batch now
                bool held = false;
                for (const PendSet& ps : pend[g2]) held = held || ps.slot == slot;
                if (held) { int rc = flush(g2); if (rc) return rc; }
            }
            if (ctx->slot_busy[slot]) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_red[slot], 0));   // slot's previous reduction must be done
            char* scratch = acc_scratch + (size_t)slot * acc_slot;
            const MsmAccSet as = acc_set(b, j, scratch);
            int rc = with_coord_field(curve, t.group, [&](auto ftag) -> int {
                typedef decltype(ftag) F;
                return msm_accumulate_batch<F>(ctx->stream, &as, 1, n, c, nwin, shared, sp.cap, pev, chunk_request, ctx->g2_slices != 0);
            });
            if (rc) return rc;
            pend[gi].push_back(PendSet{red_set(b, j, scratch), slot, j % nsched, b, j});
            return 0;
        };
        if (wide) {}
        else if (k <= 2 && ctx->table_order == 2) {
            // CG_OPT_MSM_TABLE_ORDER = 2: ONE launch order over (table, component) pairs — the G1 pairs in serpentine order, the G2 pairs together
            // after `g2_after` of them (CG_OPT_MSM_G2_AFTER; beyond the G1 count: at the end).  Both schedules are built up front.
            if (k == 2) { int rc = launch_sort(1); if (rc) return rc; }
            std::vector<std::pair<int, int>> g1o, g2o, order;
            for (int j = 0; j < k; j++) for (int bi = 0; bi < nb; bi++) {
                const int b = (j & 1) ? nb - 1 - bi : bi;
                (bases[b]->group == CG_G1 ? g1o : g2o).push_back({b, j});
            }
            const size_t at = ctx->g2_after < 0 ? g1o.size() : std::min<size_t>((size_t)ctx->g2_after, g1o.size());
            order.insert(order.end(), g1o.begin(), g1o.begin() + at); order.insert(order.end(), g2o.begin(), g2o.end()); order.insert(order.end(), g1o.begin() + at, g1o.end());
            bool waited[2] = {false, false};
            int left[2] = {(int)g1o.size(), (int)g2o.size()}, left_sched[2] = {0, 0};
            for (auto& pr : order) left_sched[pr.second]++;
            for (size_t i = 0; i < order.size(); i++) {
                const int b = order[i].first, j = order[i].second, gi = bases[b]->group == CG_G1 ? 0 : 1;
                if (!waited[j]) { HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_sorted[j], 0)); waited[j] = true; }
                { int rc = do_acc(b, j); if (rc) return rc; }
                const bool last_of_field = --left[gi] == 0;
                const bool comp_changes = i + 1 == order.size() || order[i + 1].second != j || (bases[order[i + 1].first]->group == CG_G1 ? 0 : 1) != gi;
                if (red_batch == 0 || (red_batch == 1 && comp_changes) || (last_of_field && red_batch != 3) || (int)pend[gi].size() == RED_MAX_SETS) { int rc3 = flush(gi); if (rc3) return rc3; }
                if (--left_sched[j] == 0) HIPCHK(hipEventRecord(ctx->ev_sched_free[j], ctx->stream));
            }
        } else
        for (int j = 0; j < k; j++) {
            HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_sorted[j % nsched], 0));
            // the next component's schedule is enqueued BEFORE this component's accumulates so that the two streams run side by side
            if (j + 1 < k && nsched == 2 && j + 1 < nsched) { int rc = launch_sort(j + 1); if (rc) return rc; }
            int left_in_comp[2] = {tables_of_group[0], tables_of_group[1]};
            for (int bi = 0; bi < nb; bi++) {   // group side: once per table, reusing the schedule
                const int b = (ctx->table_order >= 1 && (j & 1)) ? nb - 1 - bi : bi;      // serpentine: odd components run the tables in reverse
                const int gi = bases[b]->group == CG_G1 ? 0 : 1;
                { int rc = do_acc(b, j); if (rc) return rc; }
                const bool last_here = --left_in_comp[gi] == 0;                           // this field's last table of the component
                if (red_batch == 0 || (red_batch == 1 && last_here) || (last_here && j == k - 1 && red_batch != 3) || (int)pend[gi].size() == RED_MAX_SETS) { int rc3 = flush(gi); if (rc3) return rc3; }
            }
            HIPCHK(hipEventRecord(ctx->ev_sched_free[j % nsched], ctx->stream));
            if (j + 2 < k && nsched == 2) {                      // needs the schedule slot this component just released: its pending sets are merged first
                for (int g2 = 0; g2 < 2; g2++) { int rc = flush(g2); if (rc) return rc; }
                int rc = launch_sort(j + 2); if (rc) return rc;
            }
        }
        for (int g2 = 0; g2 < 2; g2++) { int rc = flush(g2); if (rc) return rc; }
    }
    for (int b = 0; b < nb; b++) { if (n == 0) HIPCHK(hipEventRecord(ctx->tickets[slots[b]].done, ctx->stream)); tickets_out[b] = slots[b]; }
    return 0;
}

int msm_begin_impl(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n, const void* const* d_scalars, int k, int* ticket_out) {
    if (!ticket_out) return fail(CG_ERR_ARG, "null argument");
    return msm_begin_multi_impl(ctx, 1, &bases, &offset, n, d_scalars, k, ticket_out);
}

int msm_end_impl(cg_ctx* ctx, int ticket, void* h_out);
int msm_end_impl(cg_ctx* ctx, int ticket, void* h_out) {
    if (!ctx || !h_out) return fail(CG_ERR_ARG, "null argument");
    if (ticket < 0 || ticket >= (int)ctx->tickets.size() || !ctx->tickets[ticket].live) return fail(CG_ERR_ARG, "bad MSM ticket");
    MsmTicket& t = ctx->tickets[ticket];
    HIPCHK(hipEventSynchronize(t.done));
    t.live = false;
    if (t.optimistic) {   // a bucket overflowed its guessed capacity (non-uniform scalars): redo this MSM with the exact schedule
        bool over = false;
        for (int j = 0; j < t.k; j++) over = over || t.h_flags[j] != 0;
        if (over) {
            const cg_bases* b = t.bases; const size_t off = t.offset, n = t.n; const int k = t.k;
            std::vector<const void*> sc = t.scalars;
            int t2 = -1;
            int rc = msm_begin_multi_impl(ctx, 1, &b, &off, n, sc.data(), k, &t2, true);
            if (rc) return rc;
            return msm_end_impl(ctx, t2, h_out);
        }
    }
    // the host's share of an MSM: ~100 point additions per result (the partial sums of the reduction kernels), on 64-bit limbs (host_ec64.hpp:
    // the same bytes as the kernels' 32-bit limbs; 3x the 32-bit host code, 0.2 ms less at the tail of a 2^22 proof, 0.5 ms per 2^16 proof)
    return with_group64(t.curve, t.group, [&](auto ftag, auto) -> int {
        typedef decltype(ftag) F;
        typedef cg64::Xyzz<F> X;
        const X* h = (const X*)t.h_pinned;
        cg64::Jac<F>* out = (cg64::Jac<F>*)h_out;
        for (int j = 0; j < t.k; j++) {
            const X* hs = h + (size_t)j * t.nsums;
            X acc;
            if (t.grid_fold) {
                // sum_b (b + 1) B_b = sum_k 2^k TC_k + 2^log_l sum_k 2^k TR_k: bit sums of the column side (k <= log_l, gc partial sums each)
                // then of the row side (k < log_h, gr each), merged into one sequence U_k and folded with one doubling per bit
                std::vector<X> U((size_t)t.log_l + t.log_h + 1, X::inf());
                size_t at = 0;
                for (int kk = 0; kk <= t.log_l; kk++) for (uint32_t g = 0; g < t.gc; g++) U[kk] = cg64::add(U[kk], hs[at++]);
                for (int kk = 0; kk < t.log_h; kk++) for (uint32_t g = 0; g < t.gr; g++) U[t.log_l + kk] = cg64::add(U[t.log_l + kk], hs[at++]);
                acc = U.back();
                for (size_t i = U.size() - 1; i-- > 0;) acc = cg64::add(cg64::dbl(acc), U[i]);
            } else if (t.bit_fold) {                            // sum_k 2^k T_k
                acc = hs[t.nsums - 1];
                for (int i = t.nsums - 2; i >= 0; i--) acc = cg64::add(cg64::dbl(acc), hs[i]);
            } else if (t.plain_fold) { acc = hs[0]; for (int i = 1; i < t.nsums; i++) acc = cg64::add(acc, hs[i]); }
            else {                                              // classic windows: Horner with c doublings per window
                acc = hs[t.nsums - 1];
                for (int i = t.nsums - 2; i >= 0; i--) { for (int d = 0; d < t.c; d++) acc = cg64::dbl(acc); acc = cg64::add(acc, hs[i]); }
            }
            const cg64::Jac<F> r = cg64::to_jac(acc);
            memcpy(out + j, &r, sizeof r);
        }
        return 0;
    });
}

}  // namespace

extern "C" {
int32_t cg_msm_set_scatter_capacity(cg_ctx* ctx, int32_t cap) {
    if (!ctx) return fail(CG_ERR_ARG, "null ctx");
    if (cap > 65536) return fail(CG_ERR_ARG, "capacity out of range");
    ctx->scatter_cap = cap;
    return 0;
}
int32_t cg_msm_set_chunk(cg_ctx* ctx, int32_t entries_per_lane) {
    if (!ctx) return fail(CG_ERR_ARG, "null ctx");
    if (entries_per_lane < 0 || entries_per_lane > 4096) return fail(CG_ERR_ARG, "chunk length out of range");
    ctx->msm_chunk = (uint32_t)entries_per_lane;
    return 0;
}
int32_t cg_msm_set_window(cg_ctx* ctx, int32_t c) {
    if (!ctx) return fail(CG_ERR_ARG, "null ctx");
    if (c != 0 && (c < 2 || c > 20)) return fail(CG_ERR_ARG, "window size must be 0 (auto) or in [2, 20]");
    ctx->msm_window = c;
    return 0;
}
int32_t cg_msm_scalars_after(cg_ctx* ctx, int32_t component, cg_ctx* owner, int32_t copy_ticket) {
    if (!ctx || !owner || component < 0 || component >= 4) return fail(CG_ERR_ARG, "bad argument");
    if (copy_ticket < 0 || !owner->copy_ev[copy_ticket % cg_ctx::COPY_TICKETS]) return fail(CG_ERR_ARG, "bad copy ticket");
    const int slot = copy_ticket % cg_ctx::COPY_TICKETS;
    ctx->comp_after[component] = owner->copy_id[slot] == (uint32_t)copy_ticket ? owner->copy_ev[slot] : nullptr;   // recycled: completed long ago
    return 0;
}
int32_t cg_msm_dev_begin(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n, const void* const* d_scalars, int32_t k, int32_t* ticket) {
    const int rc = msm_begin_impl(ctx, bases, offset, n, d_scalars, k, ticket);
    if (ctx) for (hipEvent_t& e : ctx->comp_after) e = nullptr;
    return rc;
}
int32_t cg_msm_dev_begin_multi(cg_ctx* ctx, int32_t n_tables, const cg_bases* const* bases, const size_t* offsets, size_t n,
                               const void* const* d_scalars, int32_t k, int32_t* tickets) {
    const int rc = msm_begin_multi_impl(ctx, n_tables, bases, offsets, n, d_scalars, k, tickets);
    if (ctx) for (hipEvent_t& e : ctx->comp_after) e = nullptr;
    return rc;
}
int32_t cg_msm_end(cg_ctx* ctx, int32_t ticket, void* h_out_jacobian) { return msm_end_impl(ctx, ticket, h_out_jacobian); }
int32_t cg_msm_dev(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n, const void* const* d_scalars, int32_t k, void* h_out) {
    int32_t t = -1;
    int rc = msm_begin_impl(ctx, bases, offset, n, d_scalars, k, &t);
    if (rc) return rc;
    return msm_end_impl(ctx, t, h_out);
}
int32_t cg_msm(cg_ctx* ctx, const cg_bases* bases, size_t offset, size_t n, const void* const* h_scalars, int32_t k, void* h_out) {
    if (!ctx || !bases || !h_scalars) return fail(CG_ERR_ARG, "null argument");
    if (k < 1 || k > 8) return fail(CG_ERR_ARG, "k out of range");
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<void*> d(k, nullptr);
    const size_t bytes = std::max<size_t>(n * 32, 16);
    for (int j = 0; j < k; j++) {
        HIPCHK(hip_malloc_flush(&d[j], bytes));
        if (n) HIPCHK(hipMemcpyAsync(d[j], h_scalars[j], n * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    int rc = cg_msm_dev(ctx, bases, offset, n, (const void* const*)d.data(), k, h_out);
    hipStreamSynchronize(ctx->stream);
    for (int j = 0; j < k; j++) hipFree(d[j]);
    return rc;
}

}  // extern "C"
