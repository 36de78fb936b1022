// C entry points: proving sessions (zkey resident on the device, one proof = what co-circom.rs:503-506 times)
#include "groth16.hpp"
#include "capi_common.hpp"

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
    cg_ctx* take(int slot = 0, bool chain = false) {
        auto& pool = chain ? idle_chain : idle;
        { std::lock_guard<std::mutex> l(mu); if (!pool[slot].empty()) { cg_ctx* c = pool[slot].back(); pool[slot].pop_back(); return c; } }
        cg_ctx* c = nullptr; if (cg_ctx_create_ex(devices[slot], chain ? 1u : (bulk_second && slot == 0 ? 2u : 0u), &c)) cgh::die("cg_ctx_create"); return c;
    }
    void give(cg_ctx* c, int slot = 0, bool chain = false) { if (!c) return; cg_ctx_sync(c); std::lock_guard<std::mutex> l(mu); (chain ? idle_chain : idle)[slot].push_back(c); }
};
namespace {
// a context borrowed from the session: returned to the pool on success, destroyed when the proof failed (its streams may hold
// half-finished work)
struct Borrowed {
    cgh_session* s; cg_ctx* c = nullptr; bool ok = false; int slot; bool chain;
    Borrowed(cgh_session* ses, bool wanted = true, int device_slot = 0, bool chain_ctx = false) : s(ses), slot(device_slot), chain(chain_ctx) { if (wanted) c = ses->take(slot, chain); }
    ~Borrowed() { if (!c) return; if (ok) s->give(c, slot, chain); else cg_ctx_destroy(c); }
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
    ProofZKey(const ProofZKey&) = delete; ProofZKey& operator=(const ProofZKey&) = delete;
};
// the further GPUs of a multi-device session for one proof: a borrowed context per device, bound to that device's table slices
struct ProofWorkers {
    std::vector<std::unique_ptr<Borrowed>> ctxs; cgh::MultiDevice md;
    explicit ProofWorkers(cgh_session* s) {
        for (size_t d = 1; d < s->devices.size(); d++) {
            ctxs.emplace_back(new Borrowed(s, true, (int)d));
            md.workers.push_back(cgh::WorkerDevice{ctxs.back()->c, &s->dzs[d]});
        }
    }
    const cgh::MultiDevice* get() const { return md.workers.empty() ? nullptr : &md; }
    void ok() { for (auto& b : ctxs) b->ok = true; }
};
void session_destroy(cgh_session* s) {
    if (!s) return;
    for (auto& pool : s->idle) for (cg_ctx* c : pool) cg_ctx_destroy(c);
    for (auto& pool : s->idle_chain) for (cg_ctx* c : pool) cg_ctx_destroy(c);
    for (size_t d = 0; d < s->ctx0.size(); d++) if (s->ctx0[d]) { cgh::release_zkey(s->ctx0[d], s->dzs[d]); cg_ctx_destroy(s->ctx0[d]); }
    delete s;
}
}
// Several GPUs of one node for one party: devices[0] runs the witness map and slice 0 of every MSM, devices[i]
// slice i (table slices registered, validated and given their window tables on their own device); partial sums are folded on the
// host.  The prove calls below work on either kind of session.  The same device may be listed more than once (tests).
int32_t cgh_session_open_multi(const int32_t* devices, int32_t n_dev, int32_t curve, const char* zkey_path, int32_t precompute, uint32_t flags, void** out) {
    cgh_session* s = nullptr;
    try {
        using namespace cgh;
        if (!devices || n_dev < 1 || n_dev > 64) throw std::runtime_error("cgh_session_open_multi: bad device list");
        s = new cgh_session(); s->device = devices[0];
        s->devices.assign(devices, devices + n_dev); s->ctx0.assign(n_dev, nullptr); s->dzs.resize(n_dev); s->idle.resize(n_dev); s->idle_chain.resize(n_dev);
        s->z = read_zkey(curve, zkey_path);
        std::vector<Fr> pub(s->z.n_public + 1);
        for (int d = 0; d < n_dev; d++) {
            if (cg_ctx_create(devices[d], &s->ctx0[d])) die("cg_ctx_create");
            s->dzs[d] = upload_zkey(s->ctx0[d], s->z, pub, (flags & 1u) ? 0 : -1, d, n_dev);
            s->dzs[d].z = &s->z;
        }
        if (precompute) for (int d = 0; d < n_dev; d++) for (cg_bases* b : {s->dzs[d].a, s->dzs[d].b1, s->dzs[d].b2, s->dzs[d].l, s->dzs[d].h})
            if (cg_bases_len(b)) CG(cg_bases_precompute(s->ctx0[d], b, precompute > 0 ? precompute : 0));
        for (int d = 0; d < n_dev; d++) CG(cg_ctx_sync(s->ctx0[d]));
        s->second_context = s->z.n_vars >= ((size_t)1 << 19) && !getenv("CGH_ONE_CONTEXT");
        s->bulk_second = s->second_context && !getenv("CGH_NO_CHAIN_PRIORITY");
        *out = s;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); session_destroy(s); return 1; }
}
// flags: bit 0 = skip the point validation (the file was validated before, cgh_zkey_validate)
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
        static const bool no_prio = getenv("CGH_NO_CHAIN_PRIORITY") != nullptr;          // tuning knob
        Borrowed ctx(s, true, 0, s->second_context && !no_prio), second(s, s->second_context);
        ProofWorkers workers(s);
        ProofZKey pz(s, ctx.c, pub);
        const auto t0 = std::chrono::steady_clock::now();
        {
            HipDriver driver(ctx.c, z.curve, Mode::Plain, nullptr);
            driver.aux = second.c; driver.owns_aux = false; driver.md = workers.get();
            VecGuard wit(driver, driver.upload_vec(w + z.n_public + 1, nullptr, z.n_vars - z.n_public - 1));
            FieldShare rs[2]; memcpy(rs[0].c[0].v, r, 32); rs[0].c[1] = rs[0].c[0]; memcpy(rs[1].c[0].v, sc, 32); rs[1].c[1] = rs[1].c[0];
            CoGroth16 prover(driver);
            Proof p = prover.prove(pz.dz, pub, wit.v, rs, nullptr);
            if (seconds) seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            store_proof(p, (uint8_t*)out_proof);
        }
        ctx.ok = second.ok = true; workers.ok();
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// three REP3 parties on an open session (threads, in-process network).  seconds (optional, 2 values): [0] = wall time of the three
// co-located parties; [1] = party 0 ALONE on the GPU, replaying the messages it received in the first run (its proof must repeat).
int32_t cgh_session_prove_rep3(void* h, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                               const uint64_t* const* streams, size_t stream_len, uint64_t* out_proofs, double* seconds) {
    cgh_session* s = (cgh_session*)h;
    try {
        using namespace cgh;
        const ZKey& z = s->z;
        const size_t n_aux = z.n_vars - z.n_public - 1, psz = 8 * z.curve.fq();
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        RecordedQueue rec_prev, rec_next;
        auto party = [&](int i, Rep3Network* net, uint8_t* out) {
            static const bool no_prio = getenv("CGH_NO_CHAIN_PRIORITY") != nullptr;      // tuning knob
            Borrowed ctx(s, true, 0, s->second_context && !no_prio), second(s, s->second_context);
            ProofWorkers workers(s);
            ProofZKey pz(s, ctx.c, pub);
            {
                HipDriver driver(ctx.c, z.curve, Mode::Rep3, net);
                driver.aux = second.c; driver.owns_aux = false; driver.md = workers.get();
                driver.rng1 = (const Fr*)streams[i]; driver.rng2 = (const Fr*)streams[(i + 2) % 3]; driver.rng_len = stream_len;
                VecGuard wit(driver, driver.upload_vec((const Fr*)wit_a[i], (const Fr*)wit_b[i], n_aux));
                CoGroth16 prover(driver);
                Proof p = prover.prove(pz.dz, pub, wit.v, nullptr, nullptr);
                store_proof(p, out);
            }
            ctx.ok = second.ok = true; workers.ok();
        };
        InProcHub hub;
        std::string errs[3];
        std::vector<std::thread> th;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 3; i++) th.emplace_back([&, i] {
            try {
                InProcNetwork net(&hub, i);
                RecordingNetwork rec(&net, &rec_prev, &rec_next);
                party(i, i == 0 && seconds ? (Rep3Network*)&rec : (Rep3Network*)&net, (uint8_t*)out_proofs + i * psz);
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); }
        });
        for (auto& t : th) t.join();
        if (report_party_errors(errs, 3)) return 1;
        if (seconds) {
            seconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            Bytes solo(psz);
            ReplayNetwork replay(0, &rec_prev, &rec_next);
            const auto t1 = std::chrono::steady_clock::now();
            party(0, &replay, solo.data());
            seconds[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            if (memcmp(solo.data(), out_proofs, psz)) throw std::runtime_error("replayed party produced a different proof");
        }
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}

}  // extern "C"
