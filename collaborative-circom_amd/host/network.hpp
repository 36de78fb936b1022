// party-to-party channels of the drivers: Rep3Network (mpc-core/src/protocols/rep3/network.rs:13-64) and the Shamir any-to-any net (shamir/network.rs:17-59); in-process, recording / replaying and C-callback transports
#pragma once
#include "base.hpp"

namespace cgh {

// ---- network -----------------------------------------------------------------------------------------------------------
struct Rep3Network {   // rep3/network.rs:30-64
    virtual ~Rep3Network() {}
    virtual int id() const = 0;
    virtual void send_next(const void* data, size_t bytes) = 0;
    virtual void recv_prev(void* data, size_t bytes) = 0;
    virtual void send_prev(const void* data, size_t bytes) = 0;   // network.send(id.prev_id(), ..) (rep3.rs:746-753)
    virtual void recv_next(void* data, size_t bytes) = 0;
    // Optional zero-copy receive: the next message from the previous party, already in page-locked memory that stays valid until the
    // network object goes away (a transport that receives into registered buffers); nullptr = not available, use recv_prev.
    virtual const void* recv_prev_pinned(size_t bytes) { (void)bytes; return nullptr; }
};
struct InProcHub {
    std::mutex mu; std::condition_variable cv;
    std::deque<Bytes> q[3];    // q[i] = messages travelling from party i to party i+1
    std::deque<Bytes> qb[3];   // qb[i] = messages travelling from party i to party i-1
    bool failed = false;       // a party died: wake everybody up instead of waiting for messages that will never come
    void abort() { { std::lock_guard<std::mutex> l(mu); failed = true; } cv.notify_all(); }
};
struct InProcNetwork : Rep3Network {
    InProcHub* hub; int me;
    InProcNetwork(InProcHub* h, int i) : hub(h), me(i) {}
    int id() const override { return me; }
    void send_next(const void* data, size_t bytes) override {
        { std::lock_guard<std::mutex> l(hub->mu); hub->q[me].emplace_back((const uint8_t*)data, (const uint8_t*)data + bytes); }
        hub->cv.notify_all();
    }
    void recv_prev(void* data, size_t bytes) override {
        const int from = (me + 2) % 3;
        std::unique_lock<std::mutex> l(hub->mu);
        hub->cv.wait(l, [&] { return !hub->q[from].empty() || hub->failed; });
        if (hub->q[from].empty()) throw std::runtime_error("another party failed");
        Bytes m = std::move(hub->q[from].front()); hub->q[from].pop_front();
        if (m.size() != bytes) throw std::runtime_error("During execution of MPC: invalid number of bytes received");   // rep3.rs:663-668
        memcpy(data, m.data(), bytes);
    }
    void send_prev(const void* data, size_t bytes) override {
        { std::lock_guard<std::mutex> l(hub->mu); hub->qb[me].emplace_back((const uint8_t*)data, (const uint8_t*)data + bytes); }
        hub->cv.notify_all();
    }
    void recv_next(void* data, size_t bytes) override {
        const int from = (me + 1) % 3;
        std::unique_lock<std::mutex> l(hub->mu);
        hub->cv.wait(l, [&] { return !hub->qb[from].empty() || hub->failed; });
        if (hub->qb[from].empty()) throw std::runtime_error("another party failed");
        Bytes m = std::move(hub->qb[from].front()); hub->qb[from].pop_front();
        if (m.size() != bytes) throw std::runtime_error("During execution of MPC: invalid number of bytes received");
        memcpy(data, m.data(), bytes);
    }
};

// A party's incoming traffic recorded during a three-party run and replayed to the same party running alone: its messages depend
// only on the inputs and the randomness streams, so the solo run repeats the recorded one bit for bit.  Used to time ONE party with
// the GPU to itself, as in a deployment (each party on its own machine), without a second and third GPU.
// Large messages (the 4 MiB chunks of a mul_vec exchange) are recorded into page-locked memory, so that the replay can hand them to the
// driver where they lie (recv_prev_pinned) — a peer whose data is already in registered buffers, i.e. the network itself is excluded
// from the solo timing, as SURVEY §8d asks; small messages are copied as before.
struct RecordedMsg { Bytes small; void* pinned = nullptr; size_t n = 0; };
struct RecordedQueue {
    std::deque<RecordedMsg> q; std::vector<void*> owned;
    ~RecordedQueue() { for (void* p : owned) cg_host_free(p); }
    void add(const void* d, size_t b) {
        RecordedMsg m; m.n = b;
        if (b >= ((size_t)1 << 20) && cg_host_alloc(b, &m.pinned) == 0) { memcpy(m.pinned, d, b); owned.push_back(m.pinned); }
        else { m.pinned = nullptr; m.small.assign((const uint8_t*)d, (const uint8_t*)d + b); }
        q.push_back(std::move(m));
    }
};
struct RecordingNetwork : Rep3Network {
    Rep3Network* inner; RecordedQueue* from_prev; RecordedQueue* from_next;
    RecordingNetwork(Rep3Network* n, RecordedQueue* p, RecordedQueue* q) : inner(n), from_prev(p), from_next(q) {}
    int id() const override { return inner->id(); }
    void send_next(const void* d, size_t b) override { inner->send_next(d, b); }
    void send_prev(const void* d, size_t b) override { inner->send_prev(d, b); }
    void recv_prev(void* d, size_t b) override { inner->recv_prev(d, b); from_prev->add(d, b); }
    void recv_next(void* d, size_t b) override { inner->recv_next(d, b); from_next->add(d, b); }
};
struct ReplayNetwork : Rep3Network {      // reads the recorded queues without consuming them: every replay object starts at the first message
    int me; RecordedQueue* from_prev; RecordedQueue* from_next; size_t at_prev = 0, at_next = 0;
    ReplayNetwork(int i, RecordedQueue* p, RecordedQueue* q) : me(i), from_prev(p), from_next(q) {}
    int id() const override { return me; }
    void send_next(const void*, size_t) override {}
    void send_prev(const void*, size_t) override {}
    static const RecordedMsg& peek(RecordedQueue* q, size_t at, size_t b) {
        if (at >= q->q.size() || q->q[at].n != b) throw std::runtime_error("replay: message sequence differs from the recorded run");
        return q->q[at];
    }
    void recv_prev(void* d, size_t b) override { const RecordedMsg& m = peek(from_prev, at_prev, b); memcpy(d, m.pinned ? m.pinned : (const void*)m.small.data(), b); at_prev++; }
    void recv_next(void* d, size_t b) override { const RecordedMsg& m = peek(from_next, at_next, b); memcpy(d, m.pinned ? m.pinned : (const void*)m.small.data(), b); at_next++; }
    const void* recv_prev_pinned(size_t b) override {
        const void* p = peek(from_prev, at_prev, b).pinned;
        if (p) at_prev++;                                // the memory itself stays with the queue's owner list
        return p;
    }
};

// The caller's own transport behind the C callback table of cgh_session_prove_rep3_party (include/cogroth16_host.h): in the Rust binding the
// callbacks are closures over Rep3MpcNet::{send_bytes, recv_bytes} (rep3/network.rs:137-176).  A failing callback ends the proof with
// std::io::Error's role: an exception that names the call and the code.
struct CallbackNetwork : Rep3Network {
    cgh_rep3_net cb;
    explicit CallbackNetwork(const cgh_rep3_net& c) : cb(c) {
        if (cb.party_id < 0 || cb.party_id > 2) throw std::runtime_error("REP3 party id must be 0, 1 or 2");     // id.rs: PartyID::try_from
        if (!cb.send_next || !cb.recv_prev) throw std::runtime_error("cgh_rep3_net: send_next and recv_prev are required");
    }
    static void check(int32_t rc, const char* what) { if (rc) throw std::runtime_error(std::string("network: ") + what + " failed with code " + std::to_string(rc)); }
    int id() const override { return cb.party_id; }
    void send_next(const void* d, size_t b) override { check(cb.send_next(cb.user, d, b), "send_next"); }
    void recv_prev(void* d, size_t b) override { check(cb.recv_prev(cb.user, d, b), "recv_prev"); }
    void send_prev(const void* d, size_t b) override { if (!cb.send_prev) throw std::runtime_error("cgh_rep3_net: send_prev is not provided"); check(cb.send_prev(cb.user, d, b), "send_prev"); }
    void recv_next(void* d, size_t b) override { if (!cb.recv_next) throw std::runtime_error("cgh_rep3_net: recv_next is not provided"); check(cb.recv_next(cb.user, d, b), "recv_next"); }
    const void* recv_prev_pinned(size_t b) override { return cb.recv_prev_pinned ? cb.recv_prev_pinned(cb.user, b) : nullptr; }
};

// Rep3Rand (rep3/rngs.rs:25-62) as the driver sees it: masking vectors, replicated random shares, masking points.  The draws themselves
// belong to the caller (ChaCha12 streams agreed with the peers, rep3.rs:385-398).
struct Rep3RandSource {
    virtual ~Rep3RandSource() {}
    // n masking field elements; `buf` (page-locked, n elements) may be used for them or ignored; the result stays valid until the proof ends
    virtual const Fr* masking_field_elements(size_t n, Fr* buf) = 0;
    virtual void random_fes(Fr& a, Fr& b) = 0;
    virtual void masking_ec_element(int group, uint8_t* out_jacobian) = 0;
    // optional: the n masks drawn ON THE DEVICE into d_out (d_tmp: n elements of scratch), both generators advanced; false = not offered
    virtual bool masks_on_device(cg_ctx*, int /*curve*/, size_t /*n*/, void* /*d_out*/, void* /*d_tmp*/) { return false; }
    // device draws may still be in flight when masks_on_device returns: settle() waits for them and moves the caller's generators; the other
    // draws settle first by themselves, the owner settles once more when the proof is done
    virtual void settle() {}
};
struct CallbackRand : Rep3RandSource {
    cgh_rep3_rand cb;
    explicit CallbackRand(const cgh_rep3_rand& c) : cb(c) {
        if (!cb.masking_field_elements || !cb.random_fes || !cb.masking_ec_element) throw std::runtime_error("cgh_rep3_rand: all three callbacks are required");
    }
    static void check(int32_t rc, const char* what) { if (rc) throw std::runtime_error(std::string("randomness source: ") + what + " failed with code " + std::to_string(rc)); }
    const Fr* masking_field_elements(size_t n, Fr* buf) override {
        settle();
        const uint64_t* out = nullptr;
        check(cb.masking_field_elements(cb.user, n, (uint64_t*)buf, &out), "masking_field_elements");
        if (!out) throw std::runtime_error("randomness source: masking_field_elements returned no data");
        return (const Fr*)out;
    }
    void random_fes(Fr& a, Fr& b) override { settle(); check(cb.random_fes(cb.user, a.v, b.v), "random_fes"); }
    void masking_ec_element(int group, uint8_t* out) override { settle(); check(cb.masking_ec_element(cb.user, group, (uint64_t*)out), "masking_ec_element"); }
    // cgh_rep3_chacha: rng1 / rng2 are ChaCha12 streams the caller can position — the backend draws F::rand(rng1) - F::rand(rng2) itself
    cgh_rep3_chacha streams{}; bool has_streams = false;
    void describe_streams(const cgh_rep3_chacha* st) {
        if (!st) return;
        if (!st->get_state || !st->set_word_pos) throw std::runtime_error("cgh_rep3_chacha: both callbacks are required");
        streams = *st; has_streams = true;
    }
    // The two draws are enqueued and NOT waited for: the generators' positions are asked for (settle) before the next draw of any kind.
    // Masks made from stream words [p, p') may have LEFT the party (a local product sent to the next party) long before settle() runs, so
    // the caller's generators must end up behind those words on EVERY path out of the proof — a failed proof whose generators stayed at p
    // would hand the next proof the same masks, and the difference of two messages would then give away the difference of two local products.
    // finish_inflight moves them whatever happened: behind the accepted draws when the draw reported its position, behind every candidate
    // the draw can have generated when it did not (a device error, a shortfall).
    struct InFlight { cg_ctx* ctx = nullptr; int32_t t1 = -1, t2 = -1; uint64_t p1 = 0, p2 = 0; size_t n = 0; } inflight;
    // words past every candidate of an n-element draw: cg_chacha12_fr_rand_dev_begin generates (n + 12 sqrt(n) + 64) / accept + 6 candidates of
    // 8 words, accept >= 0.75 for both scalar fields — 2 n + 4096 candidates bound that for every n
    static uint64_t words_bound(size_t n) { return 8 * (2 * (uint64_t)n + 4096); }
    void finish_inflight(bool strict) {
        cg_ctx* c = inflight.ctx; inflight.ctx = nullptr;
        uint64_t a1 = 0, a2 = 0;
        const int32_t r1 = cg_chacha12_fr_rand_dev_finish(c, inflight.t1, &a1);
        const std::string m1 = r1 ? cg_last_error() : "";
        const int32_t r2 = cg_chacha12_fr_rand_dev_finish(c, inflight.t2, &a2);
        const std::string m2 = r2 ? cg_last_error() : "";
        if (r1) a1 = inflight.p1 + words_bound(inflight.n);
        if (r2) a2 = inflight.p2 + words_bound(inflight.n);
        const int32_t rs = streams.set_word_pos(streams.user, a1, a2);
        if (!strict) return;
        if (r1 || r2) throw std::runtime_error(std::string("masks on the device: ") + (r1 ? m1 : m2));
        check(rs, "set_word_pos");
    }
    void settle() override { if (inflight.ctx) finish_inflight(true); }
    ~CallbackRand() { if (inflight.ctx) { try { finish_inflight(false); } catch (...) {} } }   // a proof that died between the draws and its next settle()
    bool masks_on_device(cg_ctx* ctx, int curve, size_t n, void* d_out, void* d_tmp) override {
        if (!has_streams) return false;
        settle();
        uint8_t s1[32], s2[32]; uint64_t p1 = 0, p2 = 0;
        check(streams.get_state(streams.user, s1, &p1, s2, &p2), "get_state");
        int32_t t1 = -1, t2 = -1;
        int32_t rc = cg_chacha12_fr_rand_dev_begin(ctx, curve, s1, p1, n, d_out, &t1);
        if (rc == CG_ERR_OOM) return false;                            // no room for the candidates: the generators have not moved, the host callback draws instead
        if (!rc) { rc = cg_chacha12_fr_rand_dev_begin(ctx, curve, s2, p2, n, d_tmp, &t2); if (rc) cg_chacha12_fr_rand_dev_finish(ctx, t1, nullptr); }
        if (rc == CG_ERR_OOM) return false;
        if (rc) throw std::runtime_error(std::string("masks on the device: ") + cg_last_error());
        inflight.ctx = ctx; inflight.t1 = t1; inflight.t2 = t2; inflight.p1 = p1; inflight.p2 = p2; inflight.n = n;
        if (cg_vec_sub_dev(ctx, curve, d_out, d_out, d_tmp, n)) throw std::runtime_error(std::string("masks on the device: ") + cg_last_error());
        return true;
    }
};

// Shamir: any-to-any channels (shamir/network.rs:17-59)
struct ShamirNet {
    virtual ~ShamirNet() {}
    virtual int id() const = 0;
    virtual int num_parties() const = 0;
    virtual void send(int to, const void* data, size_t bytes) = 0;
    virtual void recv(int from, void* data, size_t bytes) = 0;
};
struct InProcShamirHub {
    int n;
    std::mutex mu; std::condition_variable cv;
    std::vector<std::deque<Bytes>> q;   // q[from * n + to]
    bool failed = false;
    explicit InProcShamirHub(int n_) : n(n_), q((size_t)n_ * n_) {}
    void abort() { { std::lock_guard<std::mutex> l(mu); failed = true; } cv.notify_all(); }
};
struct InProcShamirNet : ShamirNet {
    InProcShamirHub* hub; int me;
    InProcShamirNet(InProcShamirHub* h, int i) : hub(h), me(i) {}
    int id() const override { return me; }
    int num_parties() const override { return hub->n; }
    void send(int to, const void* data, size_t bytes) override {
        { std::lock_guard<std::mutex> l(hub->mu); hub->q[(size_t)me * hub->n + to].emplace_back((const uint8_t*)data, (const uint8_t*)data + bytes); }
        hub->cv.notify_all();
    }
    void recv(int from, void* data, size_t bytes) override {
        std::unique_lock<std::mutex> l(hub->mu);
        auto& qq = hub->q[(size_t)from * hub->n + me];
        hub->cv.wait(l, [&] { return !qq.empty() || hub->failed; });
        if (qq.empty()) throw std::runtime_error("another party failed");
        Bytes m = std::move(qq.front()); qq.pop_front();
        if (m.size() != bytes) throw std::runtime_error("During execution of MPC: Invalid number of elements received");   // shamir.rs:324-329
        memcpy(data, m.data(), bytes);
    }
};

// the caller's any-to-any transport behind cgh_shamir_net (cgh_session_prove_shamir_party): closures over ShamirMpcNet in the Rust binding
struct CallbackShamirNet : ShamirNet {
    cgh_shamir_net cb;
    explicit CallbackShamirNet(const cgh_shamir_net& c) : cb(c) {
        if (cb.num_parties < 3) throw std::runtime_error("Shamir protocol requires at least 3 parties");                       // shamir/network.rs:75-77
        if (cb.party_id < 0 || cb.party_id >= cb.num_parties) throw std::runtime_error("Shamir party id out of range");
        if (!cb.send || !cb.recv) throw std::runtime_error("cgh_shamir_net: send and recv are required");
    }
    int id() const override { return cb.party_id; }
    int num_parties() const override { return cb.num_parties; }
    void send(int to, const void* d, size_t b) override { if (const int32_t rc = cb.send(cb.user, to, d, b)) throw std::runtime_error("network: send failed with code " + std::to_string(rc)); }
    void recv(int from, void* d, size_t b) override { if (const int32_t rc = cb.recv(cb.user, from, d, b)) throw std::runtime_error("network: recv failed with code " + std::to_string(rc)); }
};

}  // namespace cgh
