// C ABI of the gfx950 co-groth16 backend (include/cogroth16_hip.h) — stream pool, hardware-queue / pipe placement, contexts
#include "capi_internal.hpp"

extern "C" {
// Creating a HIP stream costs 4-10 ms on this platform (measured), a context has three to five of them: streams of destroyed
// contexts are parked per (device, priority class) and handed to the next context of the process (a prover that serves many proofs,
// the test-suite) instead of being destroyed.  A parked stream is idle: cg_ctx_destroy synchronises it first.
namespace {
std::mutex g_stream_pool_mu;
// cls: +1 high, 0 normal, -1 low.  The runtime keeps one set of (four) hardware queues per priority and hands a NEW stream the least used
// queue of its set, so the k-th stream this library creates in a class sits on queue k mod 4 of that class (other users of the process
// shift the numbering, not the spacing).  A queue serves its streams' packets in order — two busy streams on one queue wait for each
// other — so which parked stream a new context gets matters: last-in-first-out handed a process's second session pairs of streams on
// the same queue (a 2^22 resident step made after a session had come and gone: 70.6 ms against 66.3).  The pool therefore remembers each
// stream's slot (creation number mod 4) and hands out the idle stream whose slot has the fewest streams checked out.
constexpr int HWQ = 4;
struct StreamClassPool { std::vector<std::pair<hipStream_t, int>> idle; int created = 0; int out[HWQ] = {0, 0, 0, 0}; };
std::map<std::pair<int, int>, StreamClassPool> g_stream_pool;      // (device, priority class)
std::map<hipStream_t, int> g_stream_slot;                         // every stream made here -> its slot
// cg_stream_group_begin / _end (per thread): the contexts made in between belong to ONE party — within each priority class their streams
// get slots of their own as long as the class has any left (the streams they then share a queue with belong to somebody else's, mostly
// idle, contexts): a chain context's sort stream must not sit behind the bulk context's reduction batch and vice versa.
thread_local int g_group_depth = 0;
thread_local std::map<int, std::array<uint8_t, 3>> g_group_used_by_device;   // device -> [class + 1]: slots taken by the group so far
#define g_group_used (g_group_used_by_device[device])
int new_stream(int cls, hipStream_t* out) {
    if (cls == 0) { HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking)); return 0; }
    int prio_least = 0, prio_greatest = 0;                           // numerically: least >= greatest
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    HIPCHK(hipStreamCreateWithPriority(out, hipStreamNonBlocking, cls > 0 ? prio_greatest : prio_least));
    return 0;
}
// The queues of the three classes that carry the same index sit on one PIPE of the command processor, and streams on one pipe delay each other's
// dispatches by ~25 us even across classes (scripts/queue_map.hip: 140 us for two 120 us spin kernels side by side, 165 on one pipe, 260 on one queue).
// measured_pipe() finds the pipe of a new stream against idle reference streams (defined below, with the probe kernel); `want` asks for a stream on
// a given pipe: a parked one, or new ones until one lands there (the others are parked for later).
int measured_pipe(int device, int cls, hipStream_t st);
thread_local int g_group_rot = 0;                                     // pipes of this thread's stream group are rotated by this (parties of one process differ)
}  // namespace
} // extern "C" (C++ linkage for the two functions other units call)
int pooled_stream(int device, int cls, hipStream_t* out, int want) {
    std::lock_guard<std::mutex> l(g_stream_pool_mu);
    StreamClassPool& p = g_stream_pool[{device, cls}];
    if (want >= 0 && cls >= -1 && cls <= 1) {
        // (a pipe the group already uses in this class on this device — a second chain / bulk pair on the SAME device, as the tests' shared-device
        // sessions make them — would be the same hardware queue: the next pipe the class has left)
        if (g_group_depth > 0) for (int k = 0; k < HWQ && ((g_group_used[cls + 1] >> want) & 1u); k++) want = (want + 1) % HWQ;
        for (int tries = 0; tries < 2 * HWQ; tries++) {
            for (size_t i = 0; i < p.idle.size(); i++) if (p.idle[i].second == want) {
                *out = p.idle[i].first; p.out[want]++; if (g_group_depth > 0) g_group_used[cls + 1] |= (uint8_t)(1u << want);
                if (getenv("CG_DEBUG_STREAMS")) fprintf(stderr, "stream: class %d on pipe %d as asked (idle %zu)\n", cls, want, p.idle.size() - 1);
                p.idle.erase(p.idle.begin() + i);
                return 0;
            }
            hipStream_t st = nullptr;
            if (int rc = new_stream(cls, &st)) return rc;
            const int model = p.created++ % HWQ, seen = measured_pipe(device, cls, st);
            if (seen < 0) { g_stream_slot[st] = model; p.idle.push_back({st, model}); break; }      // no map on this device: the choice below
            g_stream_slot[st] = seen; p.idle.push_back({st, seen});
        }
    }
    const bool grp = g_group_depth > 0 && cls >= -1 && cls <= 1;
    uint8_t none = 0; uint8_t& used = grp ? g_group_used[cls + 1] : none;
    const bool slots_left = grp && used != (1u << HWQ) - 1;
    auto taken = [&](int slot) { return slots_left && ((used >> slot) & 1u); };
    for (int tries = 0; tries <= HWQ; tries++) {
        long best = -1;
        for (size_t i = 0; i < p.idle.size(); i++) {
            if (taken(p.idle[i].second)) continue;
            if (best < 0 || p.out[p.idle[i].second] < p.out[p.idle[best].second]) best = (long)i;                     // (ties: the longest parked)
        }
        if (best >= 0) {
            const int slot = p.idle[best].second;
            *out = p.idle[best].first; p.out[slot]++; if (grp) used |= (uint8_t)(1u << slot);
            if (getenv("CG_DEBUG_STREAMS")) fprintf(stderr, "stream: class %d slot %d (out %d %d %d %d, idle %zu%s)\n", cls, slot, p.out[0], p.out[1], p.out[2], p.out[3], p.idle.size() - 1, grp ? ", group" : "");
            p.idle.erase(p.idle.begin() + best);
            return 0;
        }
        hipStream_t st = nullptr;                                    // nothing suitable parked: a new stream joins the idle list and the choice is made again
        if (int rc = new_stream(cls, &st)) return rc;
        const int model = p.created++ % HWQ, seen = measured_pipe(device, cls, st);
        const int slot = seen >= 0 ? seen : model;
        g_stream_slot[st] = slot; p.idle.push_back({st, slot});
    }
    return fail(CG_ERR_HIP, "internal: stream pool");
}
namespace {
void park_stream(int device, int cls, hipStream_t st) {
    if (!st) return;
    std::lock_guard<std::mutex> l(g_stream_pool_mu);
    StreamClassPool& p = g_stream_pool[{device, cls}];
    auto it = g_stream_slot.find(st);
    if (it == g_stream_slot.end()) { hipStreamDestroy(st); return; }             // not one of ours
    if (p.out[it->second] > 0) p.out[it->second]--;
    if (p.idle.size() < 32) p.idle.push_back({st, it->second}); else { g_stream_slot.erase(it); hipStreamDestroy(st); }
}
}  // namespace
int make_copy_streams(cg_ctx* c, int want_h2d, int want_d2h) {
    { int rc = pooled_stream(c->device, c->prio_copy, &c->h2d, want_h2d); if (rc) return rc; }
    { int rc = pooled_stream(c->device, c->prio_copy, &c->d2h, want_d2h); if (rc) return rc; }
    HIPCHK(hipEventCreateWithFlags(&c->ev_copy_order, hipEventDisableTiming));
    return 0;
}
extern "C" {

// ---- which streams share a hardware queue?  Measured, not guessed.  The runtime maps streams onto a few hardware queues per priority class, and
// two busy streams on one queue wait for each other's packets — in particular for each other's WAITS: a context whose reduction stream shares
// a queue with its sort stream has the next schedule's kernels parked behind "wait for the accumulation" (the same small proof took 1.9 or
// 3.7 ms from one session of a process to the next).  The slot bookkeeping of the pool above is a model of the runtime's choice; the probe is
// the fact: two 120 us spin kernels, one per stream, started together — side by side they take 120 us, on one queue 240.
__global__ void k_probe_spin(unsigned long long ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } }
namespace {
// 120 us in ticks of wall_clock64 ON THE CURRENT DEVICE: the rate comes from the runtime (hipDeviceAttributeWallClockRate, kHz; 100 MHz on
// MI355X today) instead of being assumed — the probes' thresholds below are in microseconds of host time
unsigned long long probe_ticks() {
    static std::atomic<unsigned long long> cached[64] = {};
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 12000ull; }
    unsigned long long t = cached[dev].load(std::memory_order_relaxed);
    if (!t) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
        t = (unsigned long long)khz * 120ull / 1000ull;
        cached[dev].store(t, std::memory_order_relaxed);
    }
    return t;
}
bool streams_share_queue(hipStream_t a, hipStream_t b) {
    if (!a || !b || a == b) return false;
    double best = 1e9;
    for (int rep = 0; rep < 2 && best > 190.0; rep++) {                          // (a second try settles a launch hiccup)
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return false; }
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a, probe_ticks());  // 120 us
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, b, probe_ticks());
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return false; }
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return best > 190.0 && best < 420.0;        // (far beyond 240 us: the device is busy with somebody else's work and the probe says nothing)
}
// ---- the pipe of a stream (see pooled_stream).  Per device, once: four idle reference streams of the low class and four of the high class, each set on
// four different queues; the low set names the pipes, the high set is matched to it.  A new stream of the normal or high class is timed against the low
// set, one of the low class against the high set: the one pair that takes ~165 us instead of ~140 names its pipe.  Anything inconsistent (another party's
// work on the device, a runtime that maps differently) gives -1: the pool then falls back on its creation-order model.  CG_NO_PIPE_MAP: off.
struct PipeRefs { hipStream_t low[HWQ] = {nullptr, nullptr, nullptr, nullptr}, high[HWQ] = {nullptr, nullptr, nullptr, nullptr}; bool ok = false; int attempts = 0; };
std::map<int, PipeRefs> g_pipe_refs;
std::mutex g_pipe_mu;
double spin_pair_us(hipStream_t a, hipStream_t b) {
    double best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a, probe_ticks());
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, b, probe_ticks());
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return best;
}
// index of the ONE reference the stream is coupled to (same pipe: >= 152 us; same queue, 260 us, counts as well), -1 if none or several
int coupled_reference(const hipStream_t* refs, hipStream_t st) {
    int found = -1;
    for (int i = 0; i < HWQ; i++) {
        const double us = spin_pair_us(refs[i], st);
        if (us < 0 || us > 420.0) return -1;                                         // the device is busy: the probe says nothing
        if (us >= 152.0) { if (found >= 0) return -1; found = i; }
    }
    return found;
}
int measured_pipe(int device, int cls, hipStream_t st) {
    static const bool off = tune_env("CG_NO_PIPE_MAP") != nullptr || tune_env("CG_NO_STREAM_PROBE") != nullptr;
    if (off || !global_option(CG_GOPT_STREAM_PROBES) || cls < -1 || cls > 1) return -1;
    std::lock_guard<std::mutex> l(g_pipe_mu);
    PipeRefs& r = g_pipe_refs[device];
    if (!r.ok && r.attempts < 3) {                                                     // (an attempt made while somebody else's work held the device may fail: twice more, later)
        r.attempts++;
        for (hipStream_t& x : r.low) if (x) { hipStreamDestroy(x); x = nullptr; }
        for (hipStream_t& x : r.high) if (x) { hipStreamDestroy(x); x = nullptr; }
        bool ok = true;
        for (int i = 0; i < HWQ && ok; i++) ok = new_stream(-1, &r.low[i]) == 0;
        for (int i = 0; i < HWQ && ok; i++) ok = new_stream(1, &r.high[i]) == 0;
        for (int i = 0; i < HWQ && ok; i++) for (int j = i + 1; j < HWQ && ok; j++) {   // each set on four different queues
            const double a = spin_pair_us(r.low[i], r.low[j]), b = spin_pair_us(r.high[i], r.high[j]);
            ok = a > 0 && a < 152.0 && b > 0 && b < 152.0;
        }
        hipStream_t matched[HWQ] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < HWQ && ok; j++) {                                          // every high reference on the pipe of exactly one low reference, and all four used
            const int pipe = coupled_reference(r.low, r.high[j]);
            ok = pipe >= 0 && !matched[pipe];
            if (ok) matched[pipe] = r.high[j];
        }
        if (ok) for (int i = 0; i < HWQ; i++) r.high[i] = matched[i];
        r.ok = ok;
        if (getenv("CG_DEBUG_STREAMS")) fprintf(stderr, "pipe map of device %d: %s\n", device, ok ? "references in place" : "not available (the pool keeps its creation-order model)");
    }
    if (!r.ok) return -1;
    int pipe = coupled_reference(cls == -1 ? r.high : r.low, st);
    if (pipe < 0) pipe = coupled_reference(cls == -1 ? r.high : r.low, st);             // (a second try settles a launch hiccup)
    return pipe;
}
// make `moving` not share a queue with any of `fixed` (same priority class): streams that do are parked again and others tried
thread_local std::vector<hipStream_t> g_group_busy[3];                        // [class + 1]: streams of the contexts made so far in this thread's stream group
int separate_stream(int device, int cls, hipStream_t* moving, std::vector<hipStream_t> fixed) {
    static const bool off = tune_env("CG_NO_STREAM_PROBE") != nullptr;            // A/B knob
    if (off || !global_option(CG_GOPT_STREAM_PROBES)) return 0;
    std::vector<hipStream_t> rejected;
    for (int tries = 0; tries < 6; tries++) {
        bool clash = false;
        for (hipStream_t f : fixed) clash = clash || streams_share_queue(f, *moving);
        if (!clash) break;
        if (getenv("CG_DEBUG_STREAMS")) fprintf(stderr, "stream probe: class %d stream shares a hardware queue with another stream of its context: replaced (try %d)\n", cls, tries);
        rejected.push_back(*moving);                                              // (kept out of the pool until the choice is made)
        hipStream_t st = nullptr;
        if (int rc = pooled_stream(device, cls, &st)) { for (hipStream_t r : rejected) park_stream(device, cls, r); return rc; }   // a parked stream first, a new one (4-10 ms) only when none is left
        *moving = st;
    }
    for (hipStream_t r : rejected) park_stream(device, cls, r);
    return 0;
}
}  // namespace

int32_t cg_stream_group_begin(void) {
    static std::atomic<int> groups{0};
    if (g_group_depth++ == 0) { g_group_used_by_device.clear(); for (auto& v : g_group_busy) v.clear(); g_group_rot = groups.fetch_add(1) % HWQ; }
    return 0;
}
int32_t cg_stream_group_end(void) { if (g_group_depth > 0) g_group_depth--; return 0; }
int32_t cg_ctx_create(int32_t device, cg_ctx** out) { return cg_ctx_create_ex(device, 0, out); }
// flags bit 0 ("chain"): for the context that carries a dependency chain (witness map with its party-to-party exchanges) while another
// context of the same party keeps the chip full with independent bucket accumulations — main stream and copy streams (created here,
// one after the other: three different hardware queues) get high priority, the side streams normal priority.
// flags bit 1 ("bulk"): the context next to a chain context — main stream low priority, side streams normal: its kernels fill what
// the chain leaves free and share no hardware queue with it.
int32_t cg_ctx_create_ex(int32_t device, uint32_t flags, cg_ctx** out) {
    if (!out) return fail(CG_ERR_ARG, "null out");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(CG_ERR_NODEVICE, "no HIP device visible: this backend has no CPU fallback");
    if (device < 0 || device >= count) return fail(CG_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    {   // the kernels are written for 64-lane wavefronts (ballots, shuffles across 64 lanes, LDS tiles sized per wave): refuse anything else loudly
        int ws = 0; HIPCHK(hipDeviceGetAttribute(&ws, hipDeviceAttributeWarpSize, device));
        if (ws != 64) return fail(CG_ERR_NODEVICE, "device " + std::to_string(device) + " has " + std::to_string(ws) + "-lane wavefronts: this backend is written for wave64 (gfx950)");
    }
    cg_ctx* c = new cg_ctx();
    c->device = device;
    {   // planning builds (-DCG_DEBUG_KNOBS) only: environment variables seed the option table of new contexts (cg_ctx_set_option is the release interface)
        auto seed = [](const char* name, int lo, int hi, int& field) { if (const char* e = tune_env(name)) { const int v = atoi(e); if (v >= lo && v <= hi) field = v; } };
        seed("CG_MSM_TABLE_ORDER", 0, 2, c->table_order); seed("CG_MSM_G2_AFTER", -1, 64, c->g2_after); seed("CG_MSM_G2_SLICES", 0, 1, c->g2_slices);
        seed("CG_MSM_REDUCE_BATCH", 0, 3, c->red_batch); seed("CG_MSM_ACC_SLOTS", 2, cg_ctx::ACC_SLOTS_MAX, c->acc_slots); seed("CG_MSM_WIDE_SMALL", 0, 30, c->wide_small); seed("CG_MSM_ONE_STREAM_LOG", 0, 30, c->one_stream_log); seed("CG_MSM_OFF_MAIN_LOG", 0, 30, c->off_main_log); seed("CG_MSM_SOLO_LOG", 0, 30, c->solo_log);
    }
    if (flags & 1u) { c->prio_main = 1; c->prio_copy = 1; c->prio_side = 0; }
    else if (flags & 2u) { static const int bulk_cls = tune_env("CG_BULK_CLASS") ? atoi(tune_env("CG_BULK_CLASS")) : -1; c->prio_main = bulk_cls; c->prio_side = 0; }   // CG_BULK_CLASS: tuning knob
    // Inside a stream group (one party's contexts) every stream is asked for on a PIPE: the chain's main stream alone on one (the streams it shares it
    // with is idle while it works: its own sort stream), the bulk context's main, sort and reduction streams on the three others — the reduction stream
    // NOT on the pipe of the main stream, whose accumulations it runs beside at large sizes (one REP3 party, reduction stream on the main stream's pipe /
    // on its own: 2^22 73.2, 72.3 / 71.5, 71.6 ms, 2^20 23.3, 23.5 / 23.1, 23.1) — the copy streams beside the sort and reduction streams.  A party with
    // one context: main, aux and sort stream on three pipes.  (The first session of a process used to fall into this arrangement by the order in
    // which its streams were created — a 2^16 party 2.9 ms — and later ones did not: 3.2-3.5 ms.)
    const bool piped = g_group_depth > 0;
    auto pipe = [&](int k) { return piped ? (k + g_group_rot) % HWQ : -1; };
    const int w_main = (flags & 1u) ? pipe(0) : (flags & 2u) ? pipe(1) : pipe(0), w_aux = (flags & 1u) ? pipe(1) : (flags & 2u) ? pipe(3) : pipe(1), w_sort = (flags & 1u) ? pipe(0) : pipe(2);
    const int w_join = (flags & 1u) ? pipe(3) : (flags & 2u) ? pipe(2) : pipe(3);
    { int rc = pooled_stream(device, c->prio_main, &c->stream, w_main); if (rc) return rc; }
    if (flags & 1u) { int rc = make_copy_streams(c, pipe(3), pipe(2)); if (rc) return rc; }
    // the side streams carry short, latency-bound kernels the main stream's next accumulate waits for: let their workgroups
    // jump the backlog of accumulate workgroups (one priority class above the main stream's, except next to a chain)
    { int rc = pooled_stream(device, c->prio_side, &c->aux, w_aux); if (rc) return rc; }
    { int rc = pooled_stream(device, c->prio_side, &c->sortst, w_sort); if (rc) return rc; }
    // the work-free stream behind released blocks (cg_dev_free) is made here, not at the first release: inside a stream group it then gets a
    // queue apart from a bulk context's low-priority main stream (its packets are waits for OTHER streams' progress: nothing may queue behind them)
    { int rc = pooled_stream(device, -1, &c->joinst, w_join); if (rc) return rc; }
    // the context's busy streams of one priority class on hardware queues of their own (measured, see streams_share_queue): the two side
    // streams against each other and against whatever else of the context lives in their class; a chain context's copy streams against its main stream
    // Inside a stream group (one party's chain + bulk contexts) the streams of the contexts made before count as well: a class has four
    // hardware queues, a pair of contexts puts at most four streams into one class.
    {
        // one context at a time: three parties of one process make their contexts at the same moment, and two threads' spin kernels on one
        // queue read as "shared" (or as "busy") for both
        static std::mutex probe_mu;
        std::lock_guard<std::mutex> probing(probe_mu);
        const bool grp = g_group_depth > 0;
        auto others = [&](int cls, std::initializer_list<hipStream_t> own) {
            std::vector<hipStream_t> v;
            for (hipStream_t o : own) if (o) v.push_back(o);
            if (grp && cls >= -1 && cls <= 1) for (hipStream_t o : g_group_busy[cls + 1]) if (v.size() < (size_t)HWQ - 1) v.push_back(o);
            return v;
        };
        auto placed = [&](int cls, hipStream_t st) { if (grp && cls >= -1 && cls <= 1) g_group_busy[cls + 1].push_back(st); };
        if (int rc = separate_stream(device, c->prio_main, &c->stream, others(c->prio_main, {}))) return rc;
        placed(c->prio_main, c->stream);
        if (int rc = separate_stream(device, c->prio_side, &c->aux, others(c->prio_side, {c->prio_side == c->prio_main ? c->stream : nullptr}))) return rc;
        placed(c->prio_side, c->aux);
        if (int rc = separate_stream(device, c->prio_side, &c->sortst, others(c->prio_side, {c->aux, c->prio_side == c->prio_main ? c->stream : nullptr}))) return rc;
        placed(c->prio_side, c->sortst);
        if (c->h2d) {
            if (int rc = separate_stream(device, c->prio_copy, &c->h2d, others(c->prio_copy, {c->prio_copy == c->prio_main ? c->stream : nullptr}))) return rc;
            placed(c->prio_copy, c->h2d);
            if (int rc = separate_stream(device, c->prio_copy, &c->d2h, others(c->prio_copy, {c->h2d, c->prio_copy == c->prio_main ? c->stream : nullptr}))) return rc;
            placed(c->prio_copy, c->d2h);
        }
        // the work-free join stream carries only waits for the context's other streams: it must not sit in front of a BUSY stream of its class
        // (a bulk context's low-priority main stream)
        if (int rc = separate_stream(device, -1, &c->joinst, others(-1, {c->prio_main == -1 ? c->stream : nullptr}))) return rc;
        placed(-1, c->joinst);                                                      // (a later busy stream of the group keeps off its queue as well)
    }
    for (hipEvent_t& e : c->park_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) { HIPCHK(hipEventCreateWithFlags(&c->ev_sorted[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_sched_free[i], hipEventDisableTiming)); for (int rs = 0; rs < 2; rs++) HIPCHK(hipEventCreateWithFlags(&c->ev_merged[rs][i], hipEventDisableTiming)); }
    for (int i = 0; i < cg_ctx::ACC_SLOTS_MAX; i++) { HIPCHK(hipEventCreateWithFlags(&c->ev_acc[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_red[i], hipEventDisableTiming)); }
    *out = c;
    return 0;
}
int32_t cg_ctx_destroy(cg_ctx* ctx) {
    if (!ctx) return 0;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->aux);
    hipStreamSynchronize(ctx->sortst);
    for (int i = 0; i < cg_ctx::ACC_SLOTS_MAX; i++) { hipEventDestroy(ctx->ev_acc[i]); hipEventDestroy(ctx->ev_red[i]); }
    for (hipEvent_t e : ctx->mark_ev) if (e) hipEventDestroy(e);
    for (auto& d : ctx->rand_draw) { if (d.live) { cg_dev_free(ctx, d.d_cand); cg_dev_free(ctx, d.d_small); d.live = false; } if (d.ev) hipEventDestroy(d.ev); }   // (draws begun and never finished)
    if (ctx->rand_result) hipHostFree(ctx->rand_result);
    for (int i = 0; i < 2; i++) { hipEventDestroy(ctx->ev_sorted[i]); hipEventDestroy(ctx->ev_sched_free[i]); for (int rs = 0; rs < 2; rs++) hipEventDestroy(ctx->ev_merged[rs][i]); }
    hipEventDestroy(ctx->ev_in);
    if (ctx->h2d) {
        hipStreamSynchronize(ctx->h2d); hipStreamSynchronize(ctx->d2h);
        for (hipEvent_t e : ctx->copy_ev) if (e) hipEventDestroy(e);
        if (ctx->ev_peer) hipEventDestroy(ctx->ev_peer);
        hipEventDestroy(ctx->ev_copy_order);
        park_stream(ctx->device, ctx->prio_copy, ctx->h2d); park_stream(ctx->device, ctx->prio_copy, ctx->d2h);
    }
    park_stream(ctx->device, ctx->prio_side, ctx->aux);
    park_stream(ctx->device, ctx->prio_side, ctx->sortst);
    if (ctx->joinst) { hipStreamSynchronize(ctx->joinst); for (hipEvent_t e : ctx->park_ev) if (e) hipEventDestroy(e); park_stream(ctx->device, -1, ctx->joinst); }
    for (auto& kv : ctx->twiddles) shared_twiddles_release(ctx->device, kv.first);
    for (auto& kv : ctx->cosets) { hipFree(kv.second.lo); hipFree(kv.second.hi); }
    for (auto& t : ctx->tickets) { if (t.h_pinned) hipHostFree(t.h_pinned); if (t.h_flags) hipHostFree(t.h_flags); if (t.done) hipEventDestroy(t.done); }
    if (ctx->arena.base) hipFree(ctx->arena.base);
    if (ctx->ntt_arena.base) hipFree(ctx->ntt_arena.base);
    if (ctx->solo_arena.base) hipFree(ctx->solo_arena.base);
    for (void* p : ctx->retired) hipFree(p);
    if (ctx->gather_buf) hipFree(ctx->gather_buf);
    for (auto& p : ctx->ev_live) { if (p.a) hipEventDestroy(p.a); if (p.b) hipEventDestroy(p.b); }
    for (auto& p : ctx->ev_free) { if (p.a) hipEventDestroy(p.a); if (p.b) hipEventDestroy(p.b); }
    if (ctx->owns_stream) park_stream(ctx->device, ctx->prio_main, ctx->stream);
    delete ctx;
    return 0;
}
int32_t cg_ctx_sync(cg_ctx* ctx) {
    if (!ctx) return fail(CG_ERR_ARG, "null ctx");
    HIPCHK(hipStreamSynchronize(ctx->sortst)); HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->aux));
    if (ctx->h2d) { HIPCHK(hipStreamSynchronize(ctx->h2d)); HIPCHK(hipStreamSynchronize(ctx->d2h)); }
    return 0;
}
void* cg_ctx_stream(cg_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int32_t cg_ctx_set_stream(cg_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(CG_ERR_ARG, "null ctx");
    HIPCHK(hipStreamSynchronize(ctx->sortst));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->aux));
    if (ctx->owns_stream) park_stream(ctx->device, ctx->prio_main, ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->owns_stream = false;
    return 0;
}

}  // extern "C"
