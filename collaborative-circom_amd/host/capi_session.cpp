// C entry points: proving sessions (zkey resident on the device, one proof = what co-circom.rs:503-506 times)
#include "groth16.hpp"
#include "capi_common.hpp"
#include "chacha.hpp"

#include <map>
extern "C" {

// ---- proving sessions: the zkey is read, uploaded (and optionally given per-window precomputed tables) ONCE; proofs then cost
// what co-circom.rs:503-506 times.  A zkey is fixed for the life of a prover process (zkey.rs:48-71).
struct cgh_session {
    cgh::ZKey z; int device = 0; bool second_context = false;
    // one entry per GPU of the party (a plain session has one): the context the tables were registered with and the tables / table
    // slices it holds.  Contexts for proofs are kept between proofs, per device: their scratch arenas (GBs at 2^22) are allocated once
    std::vector<int> devices; std::vector<cg_ctx*> ctx0; std::vector<cgh::DeviceZKey> dzs;
    cg_ctx*& ctx0_ref() { return ctx0[0]; }
    // `chain` contexts have a high-priority main stream: they carry the witness map and its exchanges (a dependency chain) while the
    // party's second context fills the chip with the witness-independent MSMs
    std::mutex mu; std::vector<std::vector<cg_ctx*>> idle, idle_chain;
    bool bulk_second = false;                                                            // the non-chain contexts run next to a chain context
    bool additive_h = false;                                                             // open flag bit 1: REP3 additive-quotient variant
    bool counted = false;                                                                // the session was opened successfully (it counts as open until closed)
    cgh::SessionFixed fixed;                                                             // window tables of delta_1, delta_2 and the public-input records (host)
    // A free context is handed out in order of CREATION (lowest serial first), not of return: the pair made at session open serves a party
    // that proves alone in EVERY proof.  Contexts differ in how their streams fell onto the hardware queues; a first-returned-first-out
    // pool made a solo party rotate through the pairs of an earlier three-party run and its proof times cycle with them
    // (2^22: 75.4 / 72.9 / 71.9 ms, period 3).
    std::map<cg_ctx*, uint64_t> serial; uint64_t next_serial = 0;
    cg_ctx* take(int slot = 0, bool chain = false) {
        auto& pool = chain ? idle_chain : idle;
        {
            std::lock_guard<std::mutex> l(mu);
            if (!pool[slot].empty()) {
                size_t best = 0;
                for (size_t i = 1; i < pool[slot].size(); i++) if (serial[pool[slot][i]] < serial[pool[slot][best]]) best = i;
                cg_ctx* c = pool[slot][best]; pool[slot].erase(pool[slot].begin() + best); return c;
            }
        }
        static const uint32_t chain_flag = cgh::tune_env("CGH_CHAIN_FLAG") ? (uint32_t)atoi(cgh::tune_env("CGH_CHAIN_FLAG")) : 1u;     // tuning knobs (scripts/party_knobs_ab.sh)
        static const uint32_t bulk_flag = cgh::tune_env("CGH_BULK_FLAG") ? (uint32_t)atoi(cgh::tune_env("CGH_BULK_FLAG")) : 2u;
        cg_ctx* c = nullptr; if (cg_ctx_create_ex(devices[slot], chain ? chain_flag : (bulk_second ? bulk_flag : 0u), &c)) cgh::die("cg_ctx_create");
        if (const int64_t w = cgh::host_option(CGH_OPT_CTX_WIDE_LOG)) cg_ctx_set_option(c, CG_OPT_MSM_WIDE_SMALL, w);
        if (const int64_t o = cgh::host_option(CGH_OPT_CTX_OFF_MAIN_LOG)) cg_ctx_set_option(c, CG_OPT_MSM_OFF_MAIN_LOG, o);
        if (const int64_t o = cgh::host_option(CGH_OPT_CTX_SOLO_LOG)) cg_ctx_set_option(c, CG_OPT_MSM_SOLO_LOG, o);
        { std::lock_guard<std::mutex> l(mu); serial[c] = next_serial++; }
        return c;
    }
    // (a context comes back from a proof that read all its results: what its streams still hold are the release marks of freed blocks, and
    // the next proof is ordered behind them stream by stream — no synchronisation here; a failed proof's context is destroyed, not returned)
    void give(cg_ctx* c, int slot = 0, bool chain = false) { if (!c) return; std::lock_guard<std::mutex> l(mu); (chain ? idle_chain : idle)[slot].push_back(c); }
    void forget(cg_ctx* c) { std::lock_guard<std::mutex> l(mu); serial.erase(c); }
};
namespace {
std::atomic<int> g_open_sessions{0};                    // sessions handed to callers and not yet closed (session_destroy)
// the contexts made while one of these lives are one party's: their streams are spread over hardware queues of their own (cg_stream_group_begin)
struct StreamGroup { StreamGroup() { cg_stream_group_begin(); } ~StreamGroup() { cg_stream_group_end(); } StreamGroup(const StreamGroup&) = delete; StreamGroup& operator=(const StreamGroup&) = delete; };
// a context borrowed from the session: returned to the pool on success, destroyed when the proof failed (its streams may hold
// half-finished work)
struct Borrowed {
    cgh_session* s; cg_ctx* c = nullptr; bool ok = false; int slot; bool chain;
    Borrowed(cgh_session* ses, bool wanted = true, int device_slot = 0, bool chain_ctx = false) : s(ses), slot(device_slot), chain(chain_ctx) { if (wanted) c = ses->take(slot, chain); }
    ~Borrowed() { if (!c) return; if (ok) s->give(c, slot, chain); else { s->forget(c); cg_ctx_destroy(c); } }
    Borrowed(const Borrowed&) = delete; Borrowed& operator=(const Borrowed&) = delete;
};
// the zkey tables of the session with this proof's own public-input buffer (several proofs may run on one session at a time)
struct ProofZKey {
    cg_ctx* ctx; cgh::DeviceZKey dz;
    ProofZKey(cgh_session* s, cg_ctx* on, const std::vector<cgh::Fr>& pub) : ctx(on), dz(s->dzs[0]) {
        using namespace cgh;
        dz.pub_dev = nullptr;
        CG(cg_dev_alloc(ctx, pub.size() * 32, &dz.pub_dev));
        CG(cg_dev_upload(ctx, dz.pub_dev, pub.data(), pub.size() * 32));
    }
    ~ProofZKey() { if (dz.pub_dev) cg_dev_free(ctx, dz.pub_dev); }
    void* release_pub() { void* p = dz.pub_dev; dz.pub_dev = nullptr; return p; }     // handed to the driver's batch of deferred releases
    ProofZKey(const ProofZKey&) = delete; ProofZKey& operator=(const ProofZKey&) = delete;
};
// the further GPUs of a multi-device session for one proof: a borrowed context per device, bound to that device's table slices
struct ProofWorkers {
    std::vector<std::unique_ptr<Borrowed>> ctxs; cgh::MultiDevice md;
    explicit ProofWorkers(cgh_session* s) {
        for (size_t d = 1; d < s->devices.size(); d++) {
            ctxs.emplace_back(new Borrowed(s, true, (int)d));
            cgh::WorkerDevice w; w.ctx = ctxs.back()->c; w.dz = &s->dzs[d];
            ctxs.emplace_back(new Borrowed(s, true, (int)d, true));                   // its share of the witness map: a chain context of its own
            w.chain = ctxs.back()->c;
            md.workers.push_back(w);
        }
    }
    const cgh::MultiDevice* get() const { return md.workers.empty() ? nullptr : &md; }
    void ok() { for (auto& b : ctxs) b->ok = true; }
};
void session_destroy(cgh_session* s) {
    if (!s) return;
    for (auto& pool : s->idle) for (cg_ctx* c : pool) cg_ctx_destroy(c);
    for (auto& pool : s->idle_chain) for (cg_ctx* c : pool) cg_ctx_destroy(c);
    // the tables and matrices of device d are released through the context they were registered with (a session that failed while it
    // was being opened still has it) or, once the session is open, through one made for the purpose (the registration context's streams
    // went back to the pool at the end of cgh_session_open, see there)
    for (size_t d = 0; d < s->dzs.size(); d++) {
        cg_ctx* c = d < s->ctx0.size() ? s->ctx0[d] : nullptr;
        if (!c && cg_ctx_create(s->devices[d], &c)) continue;
        cgh::release_zkey(c, s->dzs[d]);
        cg_ctx_destroy(c);
    }
    // The parked blocks were sized for this circuit: back to the runtime with them — but only when this was the process's last open session.
    // The block cache is per device and process-wide, and giving blocks back is hipFree, which waits for the whole device: beside another
    // session's proofs it would stall them and empty the cache they are reusing.
    const bool last = s->counted && g_open_sessions.fetch_sub(1) == 1;
    if (last) for (int dev : s->devices) cg_dev_cache_trim(dev, nullptr);
    delete s;
}
}
// Several GPUs of one node for one party: devices[0] runs the witness map and slice 0 of every MSM, devices[i]
// slice i (table slices registered, validated and given their window tables on their own device); partial sums are folded on the
// host.  The prove calls below work on either kind of session.  The same device may be listed more than once only with
// CGH_SESSION_SHARED_DEVICES (tests, planning runs).
int32_t cgh_session_open_multi(const int32_t* devices, int32_t n_dev, int32_t curve, const char* zkey_path, int32_t precompute, uint32_t flags, void** out) {
    cgh_session* s = nullptr;
    try {
        using namespace cgh;
        if (!devices || n_dev < 1 || n_dev > 64) throw std::runtime_error("cgh_session_open_multi: bad device list");
        s = new cgh_session(); s->device = devices[0];
        s->devices.assign(devices, devices + n_dev); s->ctx0.assign(n_dev, nullptr); s->dzs.resize(n_dev); s->idle.resize(n_dev); s->idle_chain.resize(n_dev);
        // first contact with the node: the list names n DISTINCT GPUs that reach each other, or the session does not open (cg_device_preflight)
        // (pairs without peer access are accepted — cg_dev_copy_peer then goes through the host, slower but correct, and the checked copy has
        // shown that it arrives intact; a repeated GPU or a corrupted copy is not)
        if (n_dev > 1) CG(cg_device_preflight(devices, n_dev, CG_PREFLIGHT_ALLOW_STAGED | ((flags & CGH_SESSION_SHARED_DEVICES) ? CG_PREFLIGHT_ALLOW_SHARED : 0u), nullptr, 0));
        s->z = read_zkey(curve, zkey_path);
        std::vector<Fr> pub(s->z.n_public + 1);
        for (int d = 0; d < n_dev; d++) {
            if (cg_ctx_create(devices[d], &s->ctx0[d])) die("cg_ctx_create");
            s->dzs[d] = upload_zkey(s->ctx0[d], s->z, pub, (flags & 1u) ? 0 : -1, d, n_dev);
            s->dzs[d].z = &s->z;
        }
        // Window of the per-window tables.  Large tables: the library's choice by table size (c = 0).  SMALL circuits get ONE window for all five
        // tables (tables of one MSM call that differ in window run as separate sub-calls), chosen for latency: few buckets — the bucket
        // reduction is a tree of full additions, 12 us a level — and a top window that is nearly full (254 = 8 * 31 + 6 = 13 * 19 + 7: with
        // c = 12 the top window has 2 bits and a quarter of all points land in each of its 4 buckets).  Measured on the Poseidon fixture
        // (m = 256) and at 2^12, one REP3 party: c = 16 3.66 / 3.13 ms, c = 13 3.00 / 2.94, c = 8 2.84 / - (profiles/r05_small_circuit_ab2.txt).
        int window = precompute > 0 ? precompute : 0;
        if (precompute < 0) {
            const size_t nmax = std::max<size_t>(s->z.n_vars, s->z.domain_size) / (size_t)n_dev;
            // (round 5, after the small-proof work, one REP3 party: 2^8 1.07 ms at c = 8 | 2^9 1.88 at 8, 1.17 at 10 | 2^10 2.55 at 8, 1.34 at 13 |
            // 2^12 1.86 at 13, 1.99 at 15 | 2^14 2.23 at 15, 2.28 at 16 | c = 11 is always bad: its top window has one bit)
            if (nmax <= ((size_t)1 << 8)) window = 8; else if (nmax <= ((size_t)1 << 9)) window = 10; else if (nmax <= ((size_t)1 << 13)) window = 13;
        }
        if (precompute) for (int d = 0; d < n_dev; d++) for (cg_bases* b : {s->dzs[d].a, s->dzs[d].b1, s->dzs[d].b2, s->dzs[d].l, s->dzs[d].h})
            if (cg_bases_len(b)) CG(cg_bases_precompute(s->ctx0[d], b, window));
        {   // window tables of the bases every proof multiplies by a scalar, built side by side while the devices finish their set-up
            const ZKey& z = s->z; const Curve& c = z.curve;
            const size_t np = std::min<size_t>(z.n_public, SessionFixed::MAX_PUBLIC);
            s->fixed.a_pub.resize(np); s->fixed.b1_pub.resize(np); s->fixed.b2_pub.resize(np);
            generator_table(c, CG_G1); generator_table(c, CG_G2);                      // (before any thread exists: these may throw)
            // joined however this block is left: an exception past joinable threads would be std::terminate instead of an error code
            struct Joined { std::vector<std::thread> th; ~Joined() { for (auto& t : th) if (t.joinable()) t.join(); } } workers;
            std::vector<std::thread>& th = workers.th; std::vector<std::string> errs(2 + 3 * np);
            th.reserve(2 + 3 * np);
            auto job = [&](size_t slot, FixedTable* out, int group, const uint8_t* aff) {
                th.emplace_back([&errs, slot, out, group, aff, c] { try { *out = FixedTable(c, pt_from_affine(c, group, aff)); } catch (const std::exception& e) { errs[slot] = e.what(); } });
            };
            job(0, &s->fixed.delta_g1, CG_G1, z.delta_g1.data()); job(1, &s->fixed.delta_g2, CG_G2, z.delta_g2.data());
            for (size_t i = 0; i < np; i++) {
                job(2 + 3 * i, &s->fixed.a_pub[i], CG_G1, z.a_query.data() + (1 + i) * c.aff(CG_G1));
                job(3 + 3 * i, &s->fixed.b1_pub[i], CG_G1, z.b_g1_query.data() + (1 + i) * c.aff(CG_G1));
                job(4 + 3 * i, &s->fixed.b2_pub[i], CG_G2, z.b_g2_query.data() + (1 + i) * c.aff(CG_G2));
            }
            for (auto& t : th) t.join();
            for (const std::string& e : errs) if (!e.empty()) throw std::runtime_error(e);
            for (int d = 0; d < n_dev; d++) s->dzs[d].fixed = &s->fixed;
        }
        for (int d = 0; d < n_dev; d++) CG(cg_ctx_sync(s->ctx0[d]));
        // The registration contexts are done: their streams go back to the pool NOW, so that the proving contexts made below (and later)
        // take them over instead of sharing hardware queues with three streams that would sit idle for the life of the session — with
        // them held, the fifth normal-class stream of a party's pair landed on the queue of another BUSY stream of the pair whenever the
        // process had one more context of its own (a 2^16 party in such a process: 7.8 ms against 4.1).
        for (int d = 0; d < n_dev; d++) { s->dzs[d].owner = nullptr; cg_ctx_destroy(s->ctx0[d]); s->ctx0[d] = nullptr; }
        const int second_min = (int)cgh::host_option(CGH_OPT_SECOND_CONTEXT_MIN_LOG);    // tuning knob: log2 of the variables from which a proof uses two contexts
        s->second_context = s->z.n_vars >= ((size_t)1 << second_min) && !cgh::host_option(CGH_OPT_ONE_CONTEXT);
        s->bulk_second = s->second_context && !cgh::tune_env("CGH_NO_CHAIN_PRIORITY");
        s->additive_h = (flags & 2u) != 0;
        // The contexts of ONE party are made here, on this thread, in a fixed order: the runtime hands a new stream the least used hardware
        // queue of its priority class, so which streams end up sharing a queue — and with it a proof's time, by up to 4 ms at 2^22 — followed
        // from the order in which the first proofs' threads happened to create them (three co-located parties racing).  Further parties'
        // contexts are still made on demand.
        if (s->second_context) for (int d = 0; d < n_dev; d++) {
            StreamGroup one_party;
            cg_ctx* chain = s->take(d, !cgh::tune_env("CGH_NO_CHAIN_PRIORITY")); cg_ctx* bulk = s->take(d, false);
            s->give(chain, d, !cgh::tune_env("CGH_NO_CHAIN_PRIORITY")); s->give(bulk, d, false);
        }
        s->counted = true; g_open_sessions.fetch_add(1);
        *out = s;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); session_destroy(s); return 1; }
}
// flags: bit 0 = skip the point validation (the file was validated before, cgh_zkey_validate); bit 1 = REP3 proofs of this session run the
// additive-quotient variant (CoGroth16::prove: no vector exchange, MSMs on the own component, one five-point re-sharing round)
int32_t cgh_session_open_ex(int32_t device, int32_t curve, const char* zkey_path, int32_t precompute, uint32_t flags, void** out) {
    return cgh_session_open_multi(&device, 1, curve, zkey_path, precompute, flags, out);
}
int32_t cgh_session_open(int32_t device, int32_t curve, const char* zkey_path, int32_t precompute, void** out) {
    return cgh_session_open_ex(device, curve, zkey_path, precompute, 0, out);
}
int32_t cgh_session_close(void* h) { session_destroy((cgh_session*)h); return 0; }
// plain driver on an open session; seconds[0] (optional) = wall time of the prove
int32_t cgh_session_prove_plain(void* h, const uint64_t* full_witness, const uint64_t* r, const uint64_t* sc, uint64_t* out_proof, double* seconds) {
    cgh_session* s = (cgh_session*)h;
    try {
        using namespace cgh;
        const ZKey& z = s->z;
        const Fr* w = (const Fr*)full_witness;
        std::vector<Fr> pub(w, w + z.n_public + 1);
        static const bool no_prio = cgh::tune_env("CGH_NO_CHAIN_PRIORITY") != nullptr;          // tuning knob
        StreamGroup one_party;   // (contexts made here, when the pool has none, are this party's)
        Borrowed ctx(s, true, 0, s->second_context && !no_prio), second(s, s->second_context);
        ProofWorkers workers(s);
        ProofZKey pz(s, ctx.c, pub);
        const auto t0 = std::chrono::steady_clock::now();
        {
            HipDriver driver(ctx.c, z.curve, Mode::Plain, nullptr);
            driver.aux = second.c; driver.owns_aux = false; driver.md = workers.get();
            // several GPUs: the witness stays on the host here and goes up by rows, every device's rows over its own link (multidev.hpp)
            VecGuard wit(driver);
            if (DistributedWitnessMap::usable(driver, pz.dz)) { wit.v.n = z.n_vars - z.n_public - 1; driver.host_wit[0] = w + z.n_public + 1; }
            else wit.v = driver.upload_vec(w + z.n_public + 1, nullptr, z.n_vars - z.n_public - 1);
            FieldShare rs[2]; memcpy(rs[0].c[0].v, r, 32); rs[0].c[1] = rs[0].c[0]; memcpy(rs[1].c[0].v, sc, 32); rs[1].c[1] = rs[1].c[0];
            CoGroth16 prover(driver);
            Proof p = prover.prove(pz.dz, pub, wit.v, rs, nullptr);
            if (seconds) seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            store_proof(p, (uint8_t*)out_proof);
            driver.defer_free(pz.release_pub());
        }
        ctx.ok = second.ok = true; workers.ok();
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// ---- ONE REP3 party with the caller's network and randomness (include/cogroth16_host.h; co-circom.rs:484-506) ----------------------------
int32_t cgh_session_prove_rep3_party(void* h, const uint64_t* pub_in, const uint64_t* wit_a, const uint64_t* wit_b,
                                     const cgh_rep3_net* net_cb, const cgh_rep3_rand* rnd_cb, uint64_t* out_proof, double* seconds) {
    return cgh_session_prove_rep3_party_ex(h, pub_in, wit_a, wit_b, net_cb, rnd_cb, nullptr, out_proof, seconds);
}
// streams_cb != NULL: rng1 / rng2 are ChaCha12 generators the caller can position; the masking vectors of both mul_vec calls are drawn on the GPU
int32_t cgh_session_prove_rep3_party_ex(void* h, const uint64_t* pub_in, const uint64_t* wit_a, const uint64_t* wit_b, const cgh_rep3_net* net_cb,
                                        const cgh_rep3_rand* rnd_cb, const cgh_rep3_chacha* streams_cb, uint64_t* out_proof, double* seconds) {
    cgh_session* s = (cgh_session*)h;
    try {
        using namespace cgh;
        if (!s || !pub_in || !wit_a || !wit_b || !net_cb || !rnd_cb || !out_proof) throw std::runtime_error("cgh_session_prove_rep3_party: null argument");
        const ZKey& z = s->z;
        const size_t n_aux = z.n_vars - z.n_public - 1;
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        CallbackNetwork net(*net_cb);
        static const bool no_prio = cgh::tune_env("CGH_NO_CHAIN_PRIORITY") != nullptr;          // tuning knob
        StreamGroup one_party;   // (contexts made here, when the pool has none, are this party's)
        Borrowed ctx(s, true, 0, s->second_context && !no_prio), second(s, s->second_context);
        CallbackRand rnd(*rnd_cb);                                                       // (after the contexts: draws in flight are finished on a context that still exists)
        rnd.describe_streams(streams_cb);
        ProofWorkers workers(s);
        ProofZKey pz(s, ctx.c, pub);
        const auto t0 = std::chrono::steady_clock::now();
        {
            HipDriver driver(ctx.c, z.curve, Mode::Rep3, &net);
            driver.aux = second.c; driver.owns_aux = false; driver.md = workers.get();
            driver.rsrc = &rnd; driver.additive_h = s->additive_h;
            // One device: the masks of the witness map's two mul_vec calls are the first thing enqueued — the first draws of the proof in the
            // reference's order too (rep3.rs:656-660 precede :595-598).  Device draws are not waited for (CallbackRand::settle), so they cost
            // the host a few launches; shares that go up asynchronously are still crossing PCIe while the draws run on the idle chip, and
            // only then is the stream made to wait for them.
            const bool early_masks = workers.get() == nullptr;
            const bool unfenced = early_masks && n_aux >= driver.XCHG_ASYNC_MIN && cg_host_is_pinned(wit_a) && cg_host_is_pinned(wit_b);
            VecGuard wit(driver);
            if (DistributedWitnessMap::usable(driver, pz.dz)) { wit.v.n = n_aux; driver.host_wit[0] = (const Fr*)wit_a; driver.host_wit[1] = (const Fr*)wit_b; }   // up by rows (multidev.hpp)
            else wit.v = driver.upload_vec((const Fr*)wit_a, (const Fr*)wit_b, n_aux, !unfenced);
            if (early_masks) driver.prefetch_masks(2, groth16_domain(z.curve, z.pow, z.num_constraints, pub.size()).m);
            if (unfenced) driver.fence_uploads(wit.v);
            CoGroth16 prover(driver);
            Proof p = prover.prove(pz.dz, pub, wit.v, nullptr, nullptr);                 // groth16.rs:113-139
            rnd.settle();                                                                // the caller's generators stand behind the last draw when the call returns
            if (seconds) seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            store_proof(p, (uint8_t*)out_proof);
            driver.defer_free(pz.release_pub());
        }
        ctx.ok = second.ok = true; workers.ok();
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}

// ---- ONE Shamir party with the caller's network and randomness (co-circom.rs:507-527) -----------------------------------------------------
static int32_t shamir_party_impl(void* h, int32_t threshold, const uint64_t* pub_in, const uint64_t* wit_in, const cgh_shamir_net* net_cb,
                                 const cgh_shamir_rand* rnd_cb, const uint8_t* seed32, size_t preprocess, uint64_t* out_proof, double* seconds);
int32_t cgh_session_prove_shamir_party(void* h, int32_t threshold, const uint64_t* pub_in, const uint64_t* wit_in, const cgh_shamir_net* net_cb,
                                       const cgh_shamir_rand* rnd_cb, size_t preprocess, uint64_t* out_proof, double* seconds) {
    if (!rnd_cb) { g_host_err = "cgh_session_prove_shamir_party: null argument"; return 1; }
    return shamir_party_impl(h, threshold, pub_in, wit_in, net_cb, rnd_cb, nullptr, preprocess, out_proof, seconds);
}
int32_t cgh_session_prove_shamir_party_seeded(void* h, int32_t threshold, const uint64_t* pub_in, const uint64_t* wit_in, const cgh_shamir_net* net_cb,
                                              const uint8_t* seed32, size_t preprocess, uint64_t* out_proof, double* seconds) {
    if (!seed32) { g_host_err = "cgh_session_prove_shamir_party_seeded: null argument"; return 1; }
    return shamir_party_impl(h, threshold, pub_in, wit_in, net_cb, nullptr, seed32, preprocess, out_proof, seconds);
}
static int32_t shamir_party_impl(void* h, int32_t threshold, const uint64_t* pub_in, const uint64_t* wit_in, const cgh_shamir_net* net_cb,
                                 const cgh_shamir_rand* rnd_cb, const uint8_t* seed32, size_t preprocess, uint64_t* out_proof, double* seconds) {
    cgh_session* s = (cgh_session*)h;
    try {
        using namespace cgh;
        if (!s || !pub_in || !wit_in || !net_cb || (!rnd_cb && !seed32) || !out_proof) throw std::runtime_error("cgh_session_prove_shamir_party: null argument");
        if (rnd_cb && !rnd_cb->random_field_elements) throw std::runtime_error("cgh_shamir_rand: random_field_elements is required");
        const ZKey& z = s->z;
        const size_t n_aux = z.n_vars - z.n_public - 1;
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        CallbackShamirNet net(*net_cb);
        StreamGroup one_party;   // (contexts made here, when the pool has none, are this party's)
        Borrowed ctx(s, true, 0, false), second(s, s->second_context);
        ProofWorkers workers(s);
        ProofZKey pz(s, ctx.c, pub);
        const auto t0 = std::chrono::steady_clock::now();
        {
            HipDriver driver(ctx.c, z.curve, Mode::Shamir, nullptr);
            driver.aux = second.c; driver.owns_aux = false; driver.md = workers.get();
            driver.sh_rand = rnd_cb; driver.additive_h = s->additive_h;
            if (seed32) { driver.sh_gen = ChaCha12(seed32); driver.sh_gen_on = true; }
            driver.shamir_init(&net, threshold);                                        // ShamirProtocol::new, shamir.rs:211-246
            driver.preprocess(preprocess);
            VecGuard wit(driver, driver.upload_vec((const Fr*)wit_in, nullptr, n_aux));
            CoGroth16 prover(driver);
            Proof p = prover.prove(pz.dz, pub, wit.v, nullptr, nullptr);
            if (seconds) seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            store_proof(p, (uint8_t*)out_proof);
            driver.defer_free(pz.release_pub());
        }
        ctx.ok = second.ok = true; workers.ok();
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}

// ---- in-process transport behind the callback table (tests / bench / three parties on one box) ------------------------------------------
namespace {
struct Loopback {
    cgh::InProcHub hub;
    cgh::RecordedQueue rec[3][2];                                     // [party][0 = from prev, 1 = from next]
    std::vector<std::unique_ptr<cgh::Rep3Network>> nets;              // what the callback tables point at
    std::mutex mu;
};
#define NET_CALL(stmt) try { stmt; return 0; } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
int32_t lb_send_next(void* u, const void* d, size_t b) { NET_CALL(((cgh::Rep3Network*)u)->send_next(d, b)) }
int32_t lb_recv_prev(void* u, void* d, size_t b) { NET_CALL(((cgh::Rep3Network*)u)->recv_prev(d, b)) }
int32_t lb_send_prev(void* u, const void* d, size_t b) { NET_CALL(((cgh::Rep3Network*)u)->send_prev(d, b)) }
int32_t lb_recv_next(void* u, void* d, size_t b) { NET_CALL(((cgh::Rep3Network*)u)->recv_next(d, b)) }
#undef NET_CALL
const void* lb_recv_prev_pinned(void* u, size_t b) { try { return ((cgh::Rep3Network*)u)->recv_prev_pinned(b); } catch (const std::exception& e) { g_host_err = e.what(); return nullptr; } }
void fill_table(cgh_rep3_net* out, cgh::Rep3Network* n, bool pinned) {
    out->user = n; out->party_id = n->id();
    out->send_next = lb_send_next; out->recv_prev = lb_recv_prev; out->send_prev = lb_send_prev; out->recv_next = lb_recv_next;
    out->recv_prev_pinned = pinned ? lb_recv_prev_pinned : nullptr;
}
// owns an inner network next to the recorder that wraps it
struct OwningRecorder : cgh::RecordingNetwork {
    std::unique_ptr<cgh::Rep3Network> held;
    OwningRecorder(std::unique_ptr<cgh::Rep3Network> in, cgh::RecordedQueue* p, cgh::RecordedQueue* q) : cgh::RecordingNetwork(in.get(), p, q), held(std::move(in)) {}
};
}
int32_t cgh_loopback_create(void** out) { try { *out = new Loopback(); return 0; } catch (const std::exception& e) { g_host_err = e.what(); return 1; } }
int32_t cgh_loopback_net(void* hub, int32_t party, int32_t record, cgh_rep3_net* out) {
    Loopback* lb = (Loopback*)hub;
    try {
        if (!lb || !out || party < 0 || party > 2) throw std::runtime_error("cgh_loopback_net: bad argument");
        std::unique_ptr<cgh::Rep3Network> n(new cgh::InProcNetwork(&lb->hub, party));
        if (record) n.reset(new OwningRecorder(std::move(n), &lb->rec[party][0], &lb->rec[party][1]));
        std::lock_guard<std::mutex> l(lb->mu);
        fill_table(out, n.get(), false);
        lb->nets.push_back(std::move(n));
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_loopback_replay_net(void* hub, int32_t party, cgh_rep3_net* out) {
    Loopback* lb = (Loopback*)hub;
    try {
        if (!lb || !out || party < 0 || party > 2) throw std::runtime_error("cgh_loopback_replay_net: bad argument");
        std::unique_ptr<cgh::Rep3Network> n(new cgh::ReplayNetwork(party, &lb->rec[party][0], &lb->rec[party][1]));
        std::lock_guard<std::mutex> l(lb->mu);
        fill_table(out, n.get(), true);
        lb->nets.push_back(std::move(n));
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
/* a party died: wake the others out of their receives (they fail with "another party failed") */
int32_t cgh_loopback_abort(void* hub) { if (hub) ((Loopback*)hub)->hub.abort(); return 0; }
int32_t cgh_loopback_destroy(void* hub) { delete (Loopback*)hub; return 0; }

// ---- the Shamir twin of the loopback: n parties of one process joined by in-memory queues behind cgh_shamir_net tables ----------------------
namespace {
// what one party received, per sender, in order (large messages in page-locked memory): served again by ReplayShamirNet
struct ShamirRecord {
    struct Msg { void* p; size_t bytes; bool pinned; };
    std::vector<std::vector<Msg>> from;
    explicit ShamirRecord(int n) : from(n) {}
    ~ShamirRecord() { for (auto& q : from) for (Msg& m : q) { if (m.pinned) cg_host_free(m.p); else free(m.p); } }
    void keep(int sender, const void* d, size_t b) {
        Msg m{nullptr, b, false};
        if (b >= ((size_t)1 << 20) && cg_host_alloc(b, &m.p) == 0) m.pinned = true;
        else if (!(m.p = malloc(std::max<size_t>(b, 1)))) throw std::runtime_error("out of memory");
        memcpy(m.p, d, b); from[sender].push_back(m);
    }
};
struct RecordingShamirNet : cgh::ShamirNet {
    cgh::InProcShamirNet inner; ShamirRecord* rec;
    RecordingShamirNet(cgh::InProcShamirHub* h, int i, ShamirRecord* r) : inner(h, i), rec(r) {}
    int id() const override { return inner.id(); }
    int num_parties() const override { return inner.num_parties(); }
    void send(int to, const void* d, size_t b) override { inner.send(to, d, b); }
    void recv(int from, void* d, size_t b) override { inner.recv(from, d, b); if (rec) rec->keep(from, d, b); }
};
// the party ALONE: its sends are dropped, its receives are the recorded messages (a wrong size or an empty record is an error)
struct ReplayShamirNet : cgh::ShamirNet {
    int me, n; const ShamirRecord* rec; std::vector<size_t> next;
    ReplayShamirNet(int i, int np, const ShamirRecord* r) : me(i), n(np), rec(r), next(np, 0) {}
    int id() const override { return me; }
    int num_parties() const override { return n; }
    void send(int, const void*, size_t) override {}
    void recv(int from, void* d, size_t b) override {
        if (from < 0 || from >= n || next[from] >= rec->from[from].size()) throw std::runtime_error("replay: no recorded message left from that party");
        const ShamirRecord::Msg& m = rec->from[from][next[from]++];
        if (m.bytes != b) throw std::runtime_error("During execution of MPC: Invalid number of elements received");
        memcpy(d, m.p, b);
    }
};
struct ShamirLoopback {
    cgh::InProcShamirHub hub;
    std::vector<std::unique_ptr<cgh::ShamirNet>> nets; std::vector<std::unique_ptr<ShamirRecord>> recs; std::mutex mu;
    explicit ShamirLoopback(int n) : hub(n), recs(n) {}
};
int32_t sl_send(void* u, int32_t to, const void* d, size_t b) { try { ((cgh::ShamirNet*)u)->send(to, d, b); return 0; } catch (const std::exception& e) { g_host_err = e.what(); return 1; } }
int32_t sl_recv(void* u, int32_t from, void* d, size_t b) { try { ((cgh::ShamirNet*)u)->recv(from, d, b); return 0; } catch (const std::exception& e) { g_host_err = e.what(); return 1; } }
}
int32_t cgh_shamir_loopback_create(int32_t num_parties, void** out) {
    if (!out || num_parties < 3 || num_parties > 64) { g_host_err = "cgh_shamir_loopback_create: bad argument"; return 1; }
    try { *out = new ShamirLoopback(num_parties); return 0; } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_shamir_loopback_net(void* hub, int32_t party, int32_t record, cgh_shamir_net* out) {
    ShamirLoopback* lb = (ShamirLoopback*)hub;
    if (!lb || !out || party < 0 || party >= lb->hub.n) { g_host_err = "cgh_shamir_loopback_net: bad argument"; return 1; }
    try {
        std::lock_guard<std::mutex> l(lb->mu);
        if (record) lb->recs[party].reset(new ShamirRecord(lb->hub.n));
        lb->nets.emplace_back(new RecordingShamirNet(&lb->hub, party, record ? lb->recs[party].get() : nullptr));
        out->user = lb->nets.back().get(); out->party_id = party; out->num_parties = lb->hub.n; out->send = sl_send; out->recv = sl_recv;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_shamir_loopback_replay_net(void* hub, int32_t party, cgh_shamir_net* out) {
    ShamirLoopback* lb = (ShamirLoopback*)hub;
    if (!lb || !out || party < 0 || party >= lb->hub.n || !lb->recs[party]) { g_host_err = "cgh_shamir_loopback_replay_net: bad argument or nothing recorded for that party"; return 1; }
    try {
        std::lock_guard<std::mutex> l(lb->mu);
        lb->nets.emplace_back(new ReplayShamirNet(party, lb->hub.n, lb->recs[party].get()));
        out->user = lb->nets.back().get(); out->party_id = party; out->num_parties = lb->hub.n; out->send = sl_send; out->recv = sl_recv;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_shamir_loopback_abort(void* hub) { if (hub) ((ShamirLoopback*)hub)->hub.abort(); return 0; }
int32_t cgh_shamir_loopback_destroy(void* hub) { delete (ShamirLoopback*)hub; return 0; }

// ---- Rep3Rand over two pre-generated streams (rngs.rs:25-62 with the ChaCha draws done by the caller) -------------------------------------
namespace {
struct StreamRand {
    cgh::Curve curve; const cgh::Fr* rng1; const cgh::Fr* rng2; size_t len, cursor = 0;
    cgh::Fr* diff = nullptr; bool pinned = false;                      // rng1[k] - rng2[k]; page-locked when a device is present
    ~StreamRand() { if (diff) { if (pinned) cg_host_free(diff); else free(diff); } }
};
int32_t sr_masks(void* u, size_t n, uint64_t*, const uint64_t** out) {
    StreamRand* r = (StreamRand*)u;
    if (r->cursor + n > r->len) { g_host_err = "randomness stream exhausted"; return 1; }
    *out = r->diff[r->cursor].v; r->cursor += n;
    return 0;
}
int32_t sr_random_fes(void* u, uint64_t* a, uint64_t* b) {
    StreamRand* r = (StreamRand*)u;
    if (r->cursor >= r->len) { g_host_err = "randomness stream exhausted"; return 1; }
    memcpy(a, r->rng1[r->cursor].v, 32); memcpy(b, r->rng2[r->cursor].v, 32); r->cursor++;
    return 0;
}
int32_t sr_masking_ec(void* u, int32_t group, uint64_t* out) {
    StreamRand* r = (StreamRand*)u;
    try {
        using namespace cgh;
        if (r->cursor >= r->len) throw std::runtime_error("randomness stream exhausted");
        const Point m = pt_sub(r->curve, pt_mul_generator(r->curve, group, r->rng1[r->cursor]), pt_mul_generator(r->curve, group, r->rng2[r->cursor])); r->cursor++;
        memcpy(out, m.b.data(), m.b.size());
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
}
int32_t cgh_stream_rand_create(int32_t curve, const uint64_t* rng1, const uint64_t* rng2, size_t len, void** out_handle, cgh_rep3_rand* out) {
    StreamRand* r = nullptr;
    try {
        using namespace cgh;
        if (!rng1 || !rng2 || !out_handle || !out || (curve != CG_BN254 && curve != CG_BLS12_381)) throw std::runtime_error("cgh_stream_rand_create: bad argument");
        r = new StreamRand{Curve{curve}, (const Fr*)rng1, (const Fr*)rng2, len};
        void* p = nullptr;
        if (cg_host_alloc(std::max<size_t>(len, 1) * 32, &p) == 0) r->pinned = true;   // a source made where no device is visible (host-only tooling) keeps pageable memory
        else if (!(p = malloc(std::max<size_t>(len, 1) * 32))) throw std::runtime_error("cgh_stream_rand_create: out of memory");
        r->diff = (Fr*)p;
        const uint64_t* mod = MOD_R[curve];
        parallel_for(len, [&](size_t lo, size_t hi) {                 // a - b mod r on 4 x 64-bit limbs (Montgomery form subtracts like the canonical one)
            for (size_t i = lo; i < hi; i++) {
                const uint64_t* a = r->rng1[i].v; const uint64_t* b = r->rng2[i].v; uint64_t* d = r->diff[i].v;
                unsigned __int128 br = 0;
                for (int k = 0; k < 4; k++) { const unsigned __int128 t = (unsigned __int128)a[k] - b[k] - (uint64_t)br; d[k] = (uint64_t)t; br = (t >> 64) & 1; }
                if (br) { unsigned __int128 c = 0; for (int k = 0; k < 4; k++) { c += (unsigned __int128)d[k] + mod[k]; d[k] = (uint64_t)c; c >>= 64; } }
            }
        });
        out->user = r; out->masking_field_elements = sr_masks; out->random_fes = sr_random_fes; out->masking_ec_element = sr_masking_ec;
        *out_handle = r;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); delete r; return 1; }
}
int32_t cgh_stream_rand_destroy(void* handle) { delete (StreamRand*)handle; return 0; }

// ---- Rep3Rand over two ChaCha12 generators (rngs.rs:25-46), host side: chacha.hpp ---------------------------------------------------------
namespace {
struct ChaChaRand {
    cgh::Curve curve; cgh::ChaCha12 rng1, rng2; uint8_t seed1[32], seed2[32];
    const uint64_t* mod() const { return cgh::MOD_R[curve.id]; }
    int bits() const { return curve.id == CG_BN254 ? 254 : 255; }
};
int32_t cr_masks(void* u, size_t n, uint64_t* buf, const uint64_t** out) {
    ChaChaRand* r = (ChaChaRand*)u;
    const uint64_t* mod = r->mod(); const int bits = r->bits();
    for (size_t i = 0; i < n; i++) {                                   // masking_field_element: rand(rng1) - rand(rng2) (rngs.rs:37-46), one host thread like the reference
        uint64_t a[4], b[4]; uint64_t* d = buf + 4 * i;
        r->rng1.fr_rand(mod, bits, a); r->rng2.fr_rand(mod, bits, b);
        unsigned __int128 br = 0;
        for (int k = 0; k < 4; k++) { const unsigned __int128 t = (unsigned __int128)a[k] - b[k] - (uint64_t)br; d[k] = (uint64_t)t; br = (t >> 64) & 1; }
        if (br) { unsigned __int128 c = 0; for (int k = 0; k < 4; k++) { c += (unsigned __int128)d[k] + mod[k]; d[k] = (uint64_t)c; c >>= 64; } }
    }
    *out = buf;
    return 0;
}
int32_t cr_random_fes(void* u, uint64_t* a, uint64_t* b) {
    ChaChaRand* r = (ChaChaRand*)u;
    r->rng1.fr_rand(r->mod(), r->bits(), a); r->rng2.fr_rand(r->mod(), r->bits(), b);
    return 0;
}
int32_t cr_masking_ec(void* u, int32_t group, uint64_t* out) {
    ChaChaRand* r = (ChaChaRand*)u;
    try {
        using namespace cgh;
        Fr a, b;
        r->rng1.fr_rand(r->mod(), r->bits(), a.v); r->rng2.fr_rand(r->mod(), r->bits(), b.v);
        const Point m = pt_sub(r->curve, pt_mul_generator(r->curve, group, a), pt_mul_generator(r->curve, group, b));
        memcpy(out, m.b.data(), m.b.size());
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cr_get_state(void* u, uint8_t* s1, uint64_t* p1, uint8_t* s2, uint64_t* p2) {
    ChaChaRand* r = (ChaChaRand*)u;
    memcpy(s1, r->seed1, 32); memcpy(s2, r->seed2, 32); *p1 = r->rng1.word_pos; *p2 = r->rng2.word_pos;
    return 0;
}
int32_t cr_set_word_pos(void* u, uint64_t p1, uint64_t p2) { ChaChaRand* r = (ChaChaRand*)u; r->rng1.word_pos = p1; r->rng2.word_pos = p2; return 0; }
}
int32_t cgh_chacha_rand_create(int32_t curve, const uint8_t* seed1, const uint8_t* seed2, void** out_handle, cgh_rep3_rand* out, cgh_rep3_chacha* out_streams) {
    if (!seed1 || !seed2 || !out_handle || !out || (curve != CG_BN254 && curve != CG_BLS12_381)) { g_host_err = "cgh_chacha_rand_create: bad argument"; return 1; }
    ChaChaRand* r = new ChaChaRand{cgh::Curve{curve}, cgh::ChaCha12(seed1), cgh::ChaCha12(seed2), {}, {}};
    memcpy(r->seed1, seed1, 32); memcpy(r->seed2, seed2, 32);
    out->user = r; out->masking_field_elements = cr_masks; out->random_fes = cr_random_fes; out->masking_ec_element = cr_masking_ec;
    if (out_streams) { out_streams->user = r; out_streams->get_state = cr_get_state; out_streams->set_word_pos = cr_set_word_pos; }
    *out_handle = r;
    return 0;
}
int32_t cgh_chacha_rand_positions(void* handle, uint64_t* p1, uint64_t* p2) {
    if (!handle || !p1 || !p2) { g_host_err = "cgh_chacha_rand_positions: null argument"; return 1; }
    *p1 = ((ChaChaRand*)handle)->rng1.word_pos; *p2 = ((ChaChaRand*)handle)->rng2.word_pos;
    return 0;
}
int32_t cgh_chacha_rand_destroy(void* handle) { delete (ChaChaRand*)handle; return 0; }
int32_t cgh_chacha12_fr_rand_host(int32_t curve, const uint8_t* seed32, uint64_t word_pos, size_t n, uint64_t* out, uint64_t* word_pos_after) {
    if (!seed32 || (n && !out) || (curve != CG_BN254 && curve != CG_BLS12_381)) { g_host_err = "cgh_chacha12_fr_rand_host: bad argument"; return 1; }
    cgh::ChaCha12 rng(seed32, word_pos);
    for (size_t i = 0; i < n; i++) rng.fr_rand(cgh::MOD_R[curve], curve == CG_BN254 ? 254 : 255, out + 4 * i);
    if (word_pos_after) *word_pos_after = rng.word_pos;
    return 0;
}

// three REP3 parties on an open session: three threads, each calling the one-party entry above with a loopback transport and a stream
// randomness source (party i: rng1 = S_i, rng2 = S_(i-1)).  seconds (optional, 2 values): [0] = wall time of the three co-located
// parties; [1] = party 0 ALONE on the GPU, served the messages it received in the first run (its proof must repeat).
int32_t cgh_session_prove_rep3(void* h, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                               const uint64_t* const* streams, size_t stream_len, uint64_t* out_proofs, double* seconds) {
    cgh_session* s = (cgh_session*)h;
    void* hub = nullptr; void* rh[4] = {nullptr, nullptr, nullptr, nullptr};
    auto cleanup = [&] { for (void* r : rh) if (r) cgh_stream_rand_destroy(r); if (hub) cgh_loopback_destroy(hub); };
    try {
        using namespace cgh;
        const size_t psz = 8 * s->z.curve.fq();
        cgh_rep3_net nets[3]; cgh_rep3_rand rnd[4];
        if (cgh_loopback_create(&hub)) throw std::runtime_error(g_host_err);
        for (int i = 0; i < 3; i++) {
            if (cgh_loopback_net(hub, i, i == 0 && seconds, &nets[i])) throw std::runtime_error(g_host_err);
            if (cgh_stream_rand_create(s->z.curve.id, streams[i], streams[(i + 2) % 3], stream_len, &rh[i], &rnd[i])) throw std::runtime_error(g_host_err);
        }
        if (seconds && cgh_stream_rand_create(s->z.curve.id, streams[0], streams[2], stream_len, &rh[3], &rnd[3])) throw std::runtime_error(g_host_err);
        std::string errs[3];
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 3; i++) th.emplace_back([&, i] {
            if (cgh_session_prove_rep3_party(h, pub_in, wit_a[i], wit_b[i], &nets[i], &rnd[i], (uint64_t*)((uint8_t*)out_proofs + i * psz), nullptr)) { errs[i] = g_host_err; cgh_loopback_abort(hub); }
        });
        for (auto& t : th) t.join();
        if (report_party_errors(errs, 3)) { cleanup(); return 1; }
        if (seconds) {
            seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            Bytes solo(psz);
            cgh_rep3_net replay;
            if (cgh_loopback_replay_net(hub, 0, &replay)) throw std::runtime_error(g_host_err);
            if (cgh_session_prove_rep3_party(h, pub_in, wit_a[0], wit_b[0], &replay, &rnd[3], (uint64_t*)solo.data(), &seconds[1])) throw std::runtime_error(g_host_err);
            // (planning builds: CGH_EMULATE_DEVICE times one device's share of a multi-device proof on uninitialised stand-ins — nothing to compare)
            if (emulate_only_device() < 0 && memcmp(solo.data(), out_proofs, psz)) throw std::runtime_error("replayed party produced a different proof");
        }
        cleanup();
        return 0;
    } catch (const std::exception& e) { const std::string m = e.what(); cleanup(); g_host_err = m; return 1; }
}

}  // extern "C"
