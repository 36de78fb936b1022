// HipDriver: the MPC driver (modes Plain / Rep3 / Shamir) with the method names of mpc-core/src/traits.rs, every O(n) step on the GPU through the C ABI
#pragma once
#include "formats.hpp"
#include "network.hpp"
#include "chacha.hpp"
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <thread>

namespace cgh {

// Helper threads for the few independent host steps of a proof's tail (variable-base products, ~60 us each in G1): kept for the life of the
// process — starting a thread per product (std::async) cost ~20 us of the 60 it saved.  A product finds a waiting helper or, when all
// HELPERS_MAX are busy (several parties proving in one process), runs on the calling thread.  Never destroyed: the helpers sleep until exit.
class Helpers {
    static constexpr size_t HELPERS_MAX = 8;
    std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; size_t threads = 0, waiting = 0;
    void loop() {
        std::unique_lock<std::mutex> l(mu);
        for (;;) {
            waiting++; cv.wait(l, [this] { return !q.empty(); }); waiting--;
            std::function<void()> job = std::move(q.front()); q.pop_front();
            l.unlock(); job(); l.lock();
        }
    }
public:
    static Helpers& get() { static Helpers* h = new Helpers(); return *h; }
    template <class F> auto run(F f) -> std::future<decltype(f())> {
        typedef decltype(f()) R;
        auto task = std::make_shared<std::packaged_task<R()>>(std::move(f));
        std::future<R> fut = task->get_future();
        {
            std::unique_lock<std::mutex> l(mu);
            if (waiting <= q.size()) {                                               // nobody free for it
                if (threads >= HELPERS_MAX) { l.unlock(); (*task)(); return fut; }
                threads++; std::thread([this] { loop(); }).detach();
            }
            q.push_back([task] { (*task)(); });
        }
        cv.notify_one();
        return fut;
    }
};

// ---- driver ------------------------------------------------------------------------------------------------------------
struct ShareVec {   // device; REP3 uses c[0] = a, c[1] = b; plain only c[0]
    void* c[2] = {nullptr, nullptr}; size_t n = 0;
    int32_t up[2] = {-1, -1}; cg_ctx* up_ctx = nullptr;   // asynchronous uploads still filling the components (copy tickets of up_ctx)
    int32_t up_first = -1; size_t first_n = 0;            // component 0 went up in two pieces: the first first_n elements have landed at copy ticket up_first (upload_vec)
    bool ready = false;                                   // filled by copies that had completed when the vector was handed out (no stream holds work on it)
};
struct FieldShare { Fr c[2]; };
struct PointShare { Point c[2]; };
struct DeviceMatrix { uint32_t* row_ptr; uint32_t* col; void* coeff; size_t rows; };

struct DeviceZKey {   // bases uploaded once and reused by every proof / party (ownership of host buffers stays with ZKey)
    const ZKey* z;
    cg_bases *a = nullptr, *b1 = nullptr, *b2 = nullptr, *l = nullptr, *h = nullptr;
    DeviceMatrix mat[2] = {{nullptr, nullptr, nullptr, 0}, {nullptr, nullptr, nullptr, 0}};
    void* pub_dev = nullptr;
    cg_ctx* owner = nullptr;
    // several GPUs: this device holds the records [aux_lo, aux_lo + aux_n) of the four private-witness queries (counted
    // from the first private variable) and [h_lo, h_lo + h_n) of h_query, registered as tables of their own (offset 0)
    bool sliced = false; size_t aux_lo = 0, aux_n = 0, h_lo = 0, h_n = 0;
    // ... and the ROWS [h_lo, h_lo + h_n) of the two constraint matrices that are real constraints (rows_n of them; row_ptr rebased to 0):
    // evaluate_constraint (groth16.rs:156-166) runs by rows on every device (multidev.hpp)
    DeviceMatrix mat_rows[2] = {{nullptr, nullptr, nullptr, 0}, {nullptr, nullptr, nullptr, 0}};
    const struct SessionFixed* fixed = nullptr;          // a session's window tables of delta_1, delta_2 and the public-input records (host arithmetic)
};
// Fixed for the life of a zkey (zkey.rs:48-71) and multiplied by a scalar in EVERY proof (groth16.rs:220, :259-297): 8-bit window tables
struct SessionFixed {
    FixedTable delta_g1, delta_g2;
    std::vector<FixedTable> a_pub, b1_pub, b2_pub;       // query[1 + i], i < n_public
    static constexpr size_t MAX_PUBLIC = 16;             // circuits with more public inputs multiply the remaining records variable-base
};
struct WorkerDevice {     // one further GPU of a party: a context on it (its MSM slices) + its table slices; `chain`: a second, high-priority
    cg_ctx* ctx = nullptr; const DeviceZKey* dz = nullptr; cg_ctx* chain = nullptr;   // context for its share of the witness map (multidev.hpp)
};
struct MultiDevice { std::vector<WorkerDevice> workers; };

enum class Mode { Plain, Rep3, Shamir };

class HipDriver {
public:
    cg_ctx* ctx; Curve curve; Mode mode; Rep3Network* net;
    const Fr* rng1 = nullptr; const Fr* rng2 = nullptr; size_t rng_len = 0, cursor = 0;   // rngs.rs:25-46 streams (inputs)
    // REP3 with the caller's own Rep3Rand (cgh_session_prove_rep3_party): every draw goes through it instead of the two arrays above
    Rep3RandSource* rsrc = nullptr;
    std::vector<void*> mask_bufs;                                                          // page-locked scratch lent to rsrc, released at shutdown
    Fr* mask_scratch(size_t n) { void* p; CG(cg_host_alloc(n * 32, &p)); mask_bufs.push_back(p); return (Fr*)p; }
    int k() const { return k_override ? k_override : (mode == Mode::Rep3 ? 2 : 1); }
    // REP3 "additive quotient" variant (opt-in, NOT the reference's message sequence; see CoGroth16::prove): vector work on the own component only
    int k_override = 0; bool additive_h = false;
    struct Components { HipDriver& d; int old; Components(HipDriver& drv, int kk) : d(drv), old(drv.k_override) { d.k_override = kk; } ~Components() { d.k_override = old; } };
    // What arrives from a peer is checked the way the reference's deserialisation checks it (ark-serialize Validate::Yes behind
    // mpc-net's recv, rep3/network.rs:137-176 / shamir/network.rs:159-213): points on the curve and in the subgroup, field elements
    // below the modulus; failure = InvalidData.  (The 2 x m-element exchanges of mul_vec go to the device unchecked: DESIGN.md §4.)
    Point received_point(int group, const uint8_t* aff) const {
        int32_t ok = 0; CG(cg_point_validate(curve.id, group, aff, &ok));
        if (!ok) throw std::runtime_error("invalid data: a point received from a peer is not a valid curve point");
        return pt_from_affine(curve, group, aff);
    }
    // the m-element vectors a peer sends go from the transport's buffer to the device; the same range check runs there, behind the
    // upload, into a device counter that the provers read before their last opening (verify_received_vectors)
    void* d_bad = nullptr;
    void check_received_dev(const void* d_vec, size_t n) {
        if (!d_bad) { CG(cg_dev_alloc(ctx, 32, &d_bad)); CG(cg_dev_memset_zero(ctx, d_bad, 32)); }
        CG(cg_vec_check_canonical_dev(ctx, curve.id, d_vec, n, d_bad));
    }
    void verify_received_vectors() {
        if (!d_bad) return;
        uint64_t bad = 0; CG(cg_dev_download(ctx, &bad, d_bad, 8));
        if (bad) { CG(cg_dev_memset_zero(ctx, d_bad, 32)); throw std::runtime_error("invalid data: " + std::to_string(bad) + " field element(s) of a vector received from a peer are not below the modulus"); }
    }
    void check_received(const void* elements, size_t n) const {
        int32_t ok = 0; CG(cg_fr_is_canonical(curve.id, elements, n, &ok));
        if (!ok) throw std::runtime_error("invalid data: a field element received from a peer is not below the modulus");
    }
    int party() const { return mode == Mode::Rep3 ? net->id() : -1; }

    HipDriver(cg_ctx* c, Curve cv, Mode m, Rep3Network* n) : ctx(c), curve(cv), mode(m), net(n) {}
    // CGH_TIMING=1: wall-clock marks of the host-side protocol steps (stderr, microseconds since the previous mark)
    struct Marks {
        bool on; const char* what; std::chrono::steady_clock::time_point last; std::string line;
        Marks(const char* w, bool enabled) : on(enabled && getenv("CGH_TIMING")), what(w), last(std::chrono::steady_clock::now()) {}
        void mark(const char* name) {
            if (!on) return;
            const auto t = std::chrono::steady_clock::now(); char b[96];
            snprintf(b, sizeof b, " %s %.0f", name, std::chrono::duration<double, std::micro>(t - last).count()); line += b; last = t;
        }
        ~Marks() { if (on) fprintf(stderr, "%s [us]:%s\n", what, line.c_str()); }
    };

    // ---- Shamir state (shamir.rs:196-246): threshold, Lagrange tables, buffered double sharings; randomness = stream rng1
    ShamirNet* snet = nullptr; int sh_t = 0;
    std::vector<Fr> open_lagrange_t, open_lagrange_2t, mul_lagrange_2t;
    FrLazyVec sh_r_t, sh_r_2t;                                                           // LIFO pair buffers (shamir.rs:873-880); resize() does not touch the new elements
    // preprocessed pairs stay on the device: entries [pre_base, pre_base + pre_n) of the two buffers above are held in d_pre_* and
    // copied to the host only when a scalar pop or the lazy path needs them
    void* d_pre_rt = nullptr; void* d_pre_r2t = nullptr; size_t pre_base = 0, pre_n = 0; bool pre_on_host = true;
    void release_pre() { if (d_pre_rt) { cg_dev_free(ctx, d_pre_rt); cg_dev_free(ctx, d_pre_r2t); d_pre_rt = d_pre_r2t = nullptr; } pre_n = 0; pre_on_host = true; }
    void materialize_pre() {
        if (pre_on_host) return;
        const size_t live = std::min(pre_n, sh_r_t.size() > pre_base ? sh_r_t.size() - pre_base : 0);
        if (live) { CG(cg_dev_download(ctx, sh_r_t.data() + pre_base, d_pre_rt, live * 32)); CG(cg_dev_download(ctx, sh_r_2t.data() + pre_base, d_pre_r2t, live * 32)); }
        pre_on_host = true;
    }
    static constexpr size_t SHAMIR_BATCH = 1024;                                     // ShamirRng::BATCH_SIZE
    // Shamir with the caller's own RNG (cgh_session_prove_shamir_party): every draw goes through the callback instead of the stream rng1
    const cgh_shamir_rand* sh_rand = nullptr;
    // ... or from a ChaCha12 generator seeded by the caller for this proof (cgh_session_prove_shamir_party_seeded).  ShamirProtocol's RNG is
    // PRIVATE (`RngType::from_entropy()`, shamir.rs:211-246: no peer reproduces its draws), so a generator the library positions itself is
    // as good as the caller's: the amount * (1 + 3t) draws of `preprocess` are then made on the device, the short ones here, one stream.
    ChaCha12 sh_gen; bool sh_gen_on = false;
    void shamir_draw(size_t n, Fr* out) {
        if (sh_gen_on) { for (size_t i = 0; i < n; i++) sh_gen.fr_rand(MOD_R[curve.id], curve.id == CG_BN254 ? 254 : 255, out[i].v); return; }
        if (sh_rand) { if (const int32_t rc = sh_rand->random_field_elements(sh_rand->user, n, (uint64_t*)out)) throw std::runtime_error("randomness source: random_field_elements failed with code " + std::to_string(rc)); return; }
        if (cursor + n > rng_len) throw std::runtime_error("randomness stream exhausted");
        memcpy(out, rng1 + cursor, n * 32); cursor += n;
    }
    Fr next_rand() { Fr x; shamir_draw(1, &x); return x; }
    std::vector<Fr> lagrange_from_coeff(const std::vector<size_t>& pts) const {       // shamir_core.rs:56-75
        std::vector<Fr> res;
        for (size_t i : pts) {
            Fr num = fr_from_u64(curve, 1), den = num; const Fr fi = fr_from_u64(curve, i);
            for (size_t j : pts) if (i != j) { const Fr fj = fr_from_u64(curve, j); num = fr_mul(curve, num, fj); den = fr_mul(curve, den, fr_sub(curve, fj, fi)); }
            res.push_back(fr_mul(curve, num, fr_inv(curve, den)));
        }
        return res;
    }
    void shamir_init(ShamirNet* n, int threshold) {                                    // ShamirProtocol::new, shamir.rs:211-246
        snet = n; sh_t = threshold;
        const int np = n->num_parties(), id = n->id();
        if (2 * threshold + 1 > np) throw std::runtime_error("Threshold too large for number of parties");
        std::vector<size_t> p; for (int i = 0; i <= threshold; i++) p.push_back((size_t)((id + np - i) % np + 1));
        open_lagrange_t = lagrange_from_coeff(p);
        p.clear(); for (int i = 0; i <= 2 * threshold; i++) p.push_back((size_t)((id + np - i) % np + 1));
        open_lagrange_2t = lagrange_from_coeff(p);
        p.clear(); for (int i = 1; i <= 2 * threshold + 1; i++) p.push_back((size_t)i);
        mul_lagrange_2t = lagrange_from_coeff(p);
    }
    std::vector<Fr> shamir_share(const Fr& secret, int degree) {                       // shamir_core.rs:8-31
        const int np = snet->num_parties();
        std::vector<Fr> coeffs; for (int k = 0; k < degree; k++) coeffs.push_back(next_rand());
        std::vector<Fr> shares;
        for (int pidx = 1; pidx <= np; pidx++) {
            Fr sh = secret; const Fr x = fr_from_u64(curve, (uint64_t)pidx); Fr xp = x;
            for (const Fr& cf : coeffs) { sh = fr_add(curve, sh, fr_mul(curve, xp, cf)); xp = fr_mul(curve, xp, x); }
            shares.push_back(sh);
        }
        return shares;
    }
    void vandermonde_mul(const std::vector<Fr>& in, FrLazyVec& out) {            // shamir.rs:904-921 (appends t + 1 values)
        const int np = snet->num_parties();
        std::vector<Fr> row(np), cur(np);
        for (int i = 0; i < np; i++) { row[i] = fr_from_u64(curve, (uint64_t)i + 1); cur[i] = row[i]; }
        Fr s0 = fr_from_u64(curve, 0); for (const Fr& v : in) s0 = fr_add(curve, s0, v);
        out.push_back(s0);
        for (int k = 1; k <= sh_t; k++) {
            Fr acc = fr_from_u64(curve, 0);
            for (int i = 0; i < np; i++) { acc = fr_add(curve, acc, fr_mul(curve, cur[i], in[i])); cur[i] = fr_mul(curve, cur[i], row[i]); }
            out.push_back(acc);
        }
    }
    void buffer_triples(size_t amount) {                                               // shamir.rs:923-1010
        const int np = snet->num_parties(), me = snet->id();
        std::vector<Fr> rnd; for (size_t k = 0; k < amount; k++) rnd.push_back(next_rand());
        std::vector<std::vector<Fr>> send(np);
        for (const Fr& r : rnd) {
            auto a = shamir_share(r, sh_t), b = shamir_share(r, 2 * sh_t);
            for (int to = 0; to < np; to++) { send[to].push_back(a[to]); send[to].push_back(b[to]); }
        }
        for (int to = 0; to < np; to++) if (to != me) snet->send(to, send[to].data(), send[to].size() * 32);
        std::vector<std::vector<Fr>> got(np);
        for (int from = 0; from < np; from++) { if (from == me) got[from] = send[me]; else { got[from].resize(2 * amount); snet->recv(from, got[from].data(), 2 * amount * 32); check_received(got[from].data(), 2 * amount); } }
        for (size_t k = 0; k < amount; k++) {
            std::vector<Fr> in_t(np), in_2t(np);
            for (int from = 0; from < np; from++) { in_t[from] = got[from][2 * k]; in_2t[from] = got[from][2 * k + 1]; }
            vandermonde_mul(in_t, sh_r_t); vandermonde_mul(in_2t, sh_r_2t);
        }
    }
    // out[off + i*stride] = sum of terms on the device (cg_vec_lincomb_dev, at most 8 terms a launch: longer sums continue on `out`)
    struct Term { const void* src; int64_t off, stride; Fr coeff; };
    void lincomb(void* out, int64_t off, int64_t stride, size_t n, const std::vector<Term>& terms) {
        const Fr one = fr_from_u64(curve, 1);
        for (size_t at = 0; at < terms.size();) {
            std::vector<Term> part;
            if (at) part.push_back({out, off, stride, one});
            while (at < terms.size() && part.size() < 8) part.push_back(terms[at++]);
            const void* src[8]; int64_t so[8], ss[8]; Fr cf[8];
            for (size_t j = 0; j < part.size(); j++) { src[j] = part[j].src; so[j] = part[j].off; ss[j] = part[j].stride; cf[j] = part[j].coeff; }
            CG(cg_vec_lincomb_dev(ctx, curve.id, out, off, stride, n, (int32_t)part.size(), src, so, ss, cf));
        }
    }
    // ShamirProtocol::preprocess (shamir.rs:248-250) = buffer_triples(amount) (shamir.rs:923-1010) with the share algebra on the
    // device: the same draws from the stream in the same order (amount secrets, then per secret t + 2t coefficients), the same
    // values appended to the pair buffers, one message per peer.  The lazily refilled batches of 1024 (get_pair) stay on the host.
    void preprocess(size_t amount) {
        if (!amount) return;
        const int np = snet->num_parties(), me = snet->id(), t = sh_t;
        const size_t draws = amount * (size_t)(1 + 3 * t);
        if (!sh_rand && !sh_gen_on && cursor + draws > rng_len) throw std::runtime_error("randomness stream exhausted");
        Marks mk("shamir preprocess", me == 0);
        void* d_rnd = dalloc(draws * 32);
        if (sh_gen_on && draws >= DEVICE_MASKS_MIN) {                                  // the same stream, drawn where it is needed
            uint64_t after = 0;
            CG(cg_chacha12_fr_rand_dev(ctx, curve.id, (const uint8_t*)sh_gen.key, sh_gen.word_pos, draws, d_rnd, &after));
            sh_gen.word_pos = after;
        } else if (sh_rand || sh_gen_on) { std::vector<Fr> tmp(draws); shamir_draw(draws, tmp.data()); CG(cg_dev_upload(ctx, d_rnd, tmp.data(), draws * 32)); }
        else { CG(cg_dev_upload(ctx, d_rnd, rng1 + cursor, draws * 32)); cursor += draws; }
        mk.mark("upload draws");
        const Fr one = fr_from_u64(curve, 1);
        std::vector<void*> d_got(np);
        for (int from = 0; from < np; from++) d_got[from] = dalloc(2 * amount * 32);
        void* d_pairs = dalloc(2 * amount * 32);
        Fr* const buf = mask_scratch(2 * amount);                                      // page-locked staging (parked by the host cache between proofs): the copies are plain DMA
        for (int to = 0; to < np; to++) {                                              // ShamirCore::share for the receiver's point to + 1
            void* dst = to == me ? d_got[me] : d_pairs;
            const Fr x = fr_from_u64(curve, (uint64_t)to + 1);
            std::vector<Term> a{{d_rnd, 0, 1, one}}, b{{d_rnd, 0, 1, one}};
            Fr xp = x;
            for (int d = 0; d < 2 * t; d++) {
                if (d < t) a.push_back({d_rnd, (int64_t)amount + d, 3 * t, xp});
                b.push_back({d_rnd, (int64_t)amount + t + d, 3 * t, xp});
                xp = fr_mul(curve, xp, x);
            }
            lincomb(dst, 0, 2, amount, a); lincomb(dst, 1, 2, amount, b);
            if (to != me) { CG(cg_dev_download(ctx, buf, d_pairs, 2 * amount * 32)); snet->send(to, buf, 2 * amount * 32); }
        }
        mk.mark("share+send");
        for (int from = 0; from < np; from++) if (from != me) { snet->recv(from, buf, 2 * amount * 32); CG(cg_dev_upload(ctx, d_got[from], buf, 2 * amount * 32)); check_received_dev(d_got[from], 2 * amount); }
        mk.mark("recv+upload");
        // Vandermonde rows 1, x, .., x^t over the senders' points (shamir.rs:904-921): t + 1 outputs per secret
        const size_t outn = amount * (size_t)(t + 1);
        void* d_rt = dalloc(outn * 32); void* d_r2t = dalloc(outn * 32);
        std::vector<Fr> pw(np, one);
        for (int kk = 0; kk <= t; kk++) {
            std::vector<Term> a, b;
            for (int from = 0; from < np; from++) { a.push_back({d_got[from], 0, 2, pw[from]}); b.push_back({d_got[from], 1, 2, pw[from]}); }
            lincomb(d_rt, kk, t + 1, amount, a); lincomb(d_r2t, kk, t + 1, amount, b);
            for (int from = 0; from < np; from++) pw[from] = fr_mul(curve, pw[from], fr_from_u64(curve, (uint64_t)from + 1));
        }
        materialize_pre(); release_pre();                                              // an earlier preprocessed block moves to the host
        pre_base = sh_r_t.size(); pre_n = outn; d_pre_rt = d_rt; d_pre_r2t = d_r2t; pre_on_host = false;
        sh_r_t.resize(pre_base + outn); sh_r_2t.resize(pre_base + outn);
        for (void* q : d_got) CG(cg_dev_free(ctx, q));
        CG(cg_dev_free(ctx, d_rnd)); CG(cg_dev_free(ctx, d_pairs));
        mk.mark("vandermonde+free");
    }
    std::pair<Fr, Fr> get_pair() {                                                     // shamir.rs:1012-1025 (LIFO)
        if (sh_r_t.empty()) { release_pre(); buffer_triples(SHAMIR_BATCH); }
        const size_t idx = sh_r_t.size() - 1;
        if (!pre_on_host && idx >= pre_base && idx < pre_base + pre_n) {
            CG(cg_dev_download(ctx, &sh_r_t[idx], (const Fr*)d_pre_rt + (idx - pre_base), 32)); CG(cg_dev_download(ctx, &sh_r_2t[idx], (const Fr*)d_pre_r2t + (idx - pre_base), 32));
        }
        std::pair<Fr, Fr> pr{sh_r_t.back(), sh_r_2t.back()};
        sh_r_t.pop_back(); sh_r_2t.pop_back();
        return pr;
    }
    // degree_reduce_vec, shamir.rs:302-384.  `local` holds this party's products on the device and is consumed.
    ShareVec degree_reduce_vec(ShareVec local) {
        const int np = snet->num_parties(), me = snet->id();
        const size_t len = local.n;
        // the len pairs on top of the LIFO buffers, top first; read straight from the device when the preprocessed block holds them all
        const size_t top = sh_r_t.size();
        const bool on_dev = !pre_on_host && top >= len && top - len >= pre_base && top <= pre_base + pre_n;
        const Fr one = fr_from_u64(curve, 1);
        std::vector<Fr> rt, r2t;
        Marks mk(me == 0 ? "degree_reduce_vec king" : "degree_reduce_vec party 1", me <= 1);
        void* tmp = dalloc(len * 32);
        if (on_dev) {
            lincomb(local.c[0], 0, 1, len, {{local.c[0], 0, 1, one}, {d_pre_r2t, (int64_t)(top - 1 - pre_base), -1, one}});   // input += r_2t
        } else {
            materialize_pre();
            rt.resize(len); r2t.resize(len);
            for (size_t k = 0; k < len; k++) { auto pr = get_pair(); rt[k] = pr.first; r2t[k] = pr.second; }
            CG(cg_dev_upload(ctx, tmp, r2t.data(), len * 32));
            CG(cg_vec_add_dev(ctx, curve.id, local.c[0], local.c[0], tmp, len));      // input += r_2t
        }
        Fr* const buf = mask_scratch(len);                                             // page-locked staging of the messages to / from the king
        mk.mark("add r_2t");
        if (me == 0) {                                                                 // KING_ID: interpolate at 0 from parties 0..2t, re-share with degree t
            CG(cg_vec_affine_dev(ctx, curve.id, local.c[0], local.c[0], len, mul_lagrange_2t[0].v, nullptr));   // acc = input * lagrange_0
            for (int other = 1; other <= 2 * sh_t; other++) {
                snet->recv(other, buf, len * 32);
                CG(cg_dev_upload(ctx, tmp, buf, len * 32)); check_received_dev(tmp, len);
                CG(cg_vec_affine_dev(ctx, curve.id, tmp, tmp, len, mul_lagrange_2t[other].v, nullptr));
                CG(cg_vec_add_dev(ctx, curve.id, local.c[0], local.c[0], tmp, len));
            }
            mk.mark("recv+interpolate");
            // ShamirCore::share per element: coefficients are drawn element by element (t per element)
            std::vector<void*> d_coeff(sh_t);
            for (int d = 0; d < sh_t; d++) d_coeff[d] = dalloc(len * 32);
            if (sh_gen_on && len * (size_t)sh_t >= DEVICE_MASKS_MIN) {
                // the party's own ChaCha12 stream: draw k * t + d is coefficient d of element k — all len * t draws on the device, then de-interleaved
                void* d_all = dalloc(len * (size_t)sh_t * 32); uint64_t after = 0;
                CG(cg_chacha12_fr_rand_dev(ctx, curve.id, (const uint8_t*)sh_gen.key, sh_gen.word_pos, len * (size_t)sh_t, d_all, &after));
                sh_gen.word_pos = after;
                for (int d = 0; d < sh_t; d++) CG(cg_vec_gather_strided_dev(ctx, curve.id, d_coeff[d], d_all, len, (size_t)d, (size_t)sh_t));
                CG(cg_dev_free(ctx, d_all));
            } else {
                std::vector<std::vector<Fr>> coeff(sh_t, std::vector<Fr>(len));
                for (size_t k = 0; k < len; k++) for (int d = 0; d < sh_t; d++) coeff[d][k] = next_rand();
                for (int d = 0; d < sh_t; d++) CG(cg_dev_upload(ctx, d_coeff[d], coeff[d].data(), len * 32));
            }
            void* share = dalloc(len * 32); void* term = dalloc(len * 32);
            void* mine = dalloc(len * 32);
            for (int to = np - 1; to >= 0; to--) {                                     // any order: every share is a function of (acc, coeffs) only
                const Fr x = fr_from_u64(curve, (uint64_t)to + 1); Fr xp = x;
                // share = acc + sum_d coeff_d * x^(d+1)
                bool first = true;
                for (int d = 0; d < sh_t; d++) {
                    CG(cg_vec_affine_dev(ctx, curve.id, term, d_coeff[d], len, xp.v, nullptr));     // term = coeff_d * x^(d+1)
                    CG(cg_vec_add_dev(ctx, curve.id, share, first ? local.c[0] : share, term, len));
                    first = false; xp = fr_mul(curve, xp, x);
                }
                if (sh_t == 0) { CG(cg_dev_memset_zero(ctx, share, len * 32)); CG(cg_vec_add_dev(ctx, curve.id, share, share, local.c[0], len)); }
                if (to == 0) { CG(cg_dev_memset_zero(ctx, mine, len * 32)); CG(cg_vec_add_dev(ctx, curve.id, mine, mine, share, len)); }
                else { CG(cg_dev_download(ctx, buf, share, len * 32)); snet->send(to, buf, len * 32); }
            }
            CG(cg_dev_free(ctx, local.c[0])); local.c[0] = mine;
            for (void* p : d_coeff) CG(cg_dev_free(ctx, p));
            CG(cg_dev_free(ctx, share)); CG(cg_dev_free(ctx, term));
            mk.mark("reshare+send");
        } else {
            if (me <= 2 * sh_t) { CG(cg_dev_download(ctx, buf, local.c[0], len * 32)); snet->send(0, buf, len * 32); }   // only if my items are required
            mk.mark("download+send");
            snet->recv(0, buf, len * 32);
            mk.mark("wait for king");
            CG(cg_dev_upload(ctx, local.c[0], buf, len * 32)); check_received_dev(local.c[0], len);
            mk.mark("upload");
        }
        if (on_dev) {
            lincomb(tmp, 0, 1, len, {{d_pre_rt, (int64_t)(top - 1 - pre_base), -1, one}});
            sh_r_t.resize(top - len); sh_r_2t.resize(top - len);
        } else CG(cg_dev_upload(ctx, tmp, rt.data(), len * 32));
        CG(cg_vec_sub_dev(ctx, curve.id, local.c[0], local.c[0], tmp, len));          // share - r_t
        CG(cg_dev_free(ctx, tmp));
        mk.mark("sub r_t");
        return local;
    }
    Fr degree_reduce(Fr input) {                                                       // shamir.rs:252-300
        const int np = snet->num_parties(), me = snet->id();
        auto pr = get_pair();
        input = fr_add(curve, input, pr.second);
        Fr my_share;
        if (me == 0) {
            Fr acc = fr_mul(curve, input, mul_lagrange_2t[0]);
            for (int other = 1; other <= 2 * sh_t; other++) { Fr r; snet->recv(other, r.v, 32); check_received(r.v, 1); acc = fr_add(curve, acc, fr_mul(curve, r, mul_lagrange_2t[other])); }
            auto shares = shamir_share(acc, sh_t);
            for (int to = 0; to < np; to++) { if (to == me) my_share = shares[to]; else snet->send(to, shares[to].v, 32); }
        } else {
            if (me <= 2 * sh_t) snet->send(0, input.v, 32);
            snet->recv(0, my_share.v, 32); check_received(my_share.v, 1);
        }
        return fr_sub(curve, my_share, pr.first);
    }
    Point degree_reduce_point(Point input) {                                           // shamir.rs:386-436; C::rand stand-in: G * next_rand()
        const int np = snet->num_parties(), me = snet->id();
        const int g = input.group;
        auto pr = get_pair();
        const Point gen = pt_generator(curve, g);
        input = pt_add(curve, input, pt_mul_generator(curve, g, pr.second));
        Point my_share = pt_inf(curve, g);
        const size_t psz = curve.aff(g);
        if (me == 0) {
            Point acc = pt_mul(curve, input, mul_lagrange_2t[0]);
            for (int other = 1; other <= 2 * sh_t; other++) { Bytes a(psz); snet->recv(other, a.data(), psz); acc = pt_add(curve, acc, pt_mul(curve, received_point(g, a.data()), mul_lagrange_2t[other])); }
            std::vector<Point> coeffs; for (int d = 0; d < sh_t; d++) coeffs.push_back(pt_mul_generator(curve, g, next_rand()));
            for (int to = 0; to < np; to++) {
                Point sh = acc; const Fr x = fr_from_u64(curve, (uint64_t)to + 1); Fr xp = x;
                for (const Point& cf : coeffs) { sh = pt_add(curve, sh, pt_mul(curve, cf, xp)); xp = fr_mul(curve, xp, x); }
                if (to == me) my_share = sh; else { Bytes a = pt_to_affine(curve, sh); snet->send(to, a.data(), a.size()); }
            }
        } else {
            if (me <= 2 * sh_t) { Bytes a = pt_to_affine(curve, input); snet->send(0, a.data(), a.size()); }
            Bytes a(psz); snet->recv(0, a.data(), psz); my_share = received_point(g, a.data());
        }
        return pt_sub(curve, my_share, pt_mul_generator(curve, g, pr.first));
    }
    // broadcast_next(t + 1) + reconstruct_point (network.rs:233-266, shamir.rs:778-782)
    Point shamir_open_point(const Point& mine) {
        const int np = snet->num_parties(), me = snet->id();
        Bytes a = pt_to_affine(curve, mine);
        for (int sft = 1; sft <= sh_t; sft++) snet->send((me + sft) % np, a.data(), a.size());
        Point res = pt_mul(curve, mine, open_lagrange_t[0]);
        for (int r = 1; r <= sh_t; r++) { Bytes b(a.size()); snet->recv((me + np - r) % np, b.data(), b.size()); res = pt_add(curve, res, pt_mul(curve, received_point(mine.group, b.data()), open_lagrange_t[r])); }
        return res;
    }

    void* dalloc(size_t bytes) { void* p; CG(cg_dev_alloc(ctx, bytes, &p)); return p; }
    ShareVec alloc_vec(size_t n) { ShareVec v; v.n = n; for (int j = 0; j < k(); j++) { v.c[j] = dalloc(n * 32); CG(cg_dev_memset_zero(ctx, v.c[j], n * 32)); } return v; }
    void free_vec(ShareVec& v) { CG(cg_dev_free_many(ctx, v.c, 2)); v.c[0] = v.c[1] = nullptr; }
    // fence = false: the copies are started and this context's stream is NOT yet made to wait for them — the caller enqueues work that
    // does not read the shares (the masking draws of the two mul_vec calls) and then calls fence_uploads
    ShareVec upload_vec(const Fr* a, const Fr* b, size_t n, bool fence = true) {
        ShareVec v; v.n = n;
        if (n >= XCHG_ASYNC_MIN && cg_host_is_pinned(a) && (!b || k() < 2 || cg_host_is_pinned(b))) {   // page-locked shares: asynchronous DMA, the stream waits
            v.c[0] = dalloc(n * 32);
            // large proofs: component a in two pieces, so that the first MSM over it can start on the first piece while the rest is still
            // crossing PCIe (msm_begin_aux_split; 2.4 ms per component at 2^22)
            const size_t split_min = (size_t)host_option(CGH_OPT_SPLIT_FIRST_MSM_MIN);
            if (split_min && n >= split_min && cg_host_is_pinned(a)) {
                v.first_n = (n / 2 + 63) / 64 * 64;
                v.up_first = upload_staged(v.c[0], a, v.first_n);
                v.up[0] = upload_staged((uint8_t*)v.c[0] + v.first_n * 32, a + v.first_n, n - v.first_n);
            } else v.up[0] = upload_staged(v.c[0], a, n);
            if (b && k() == 2) { v.c[1] = dalloc(n * 32); v.up[1] = upload_staged(v.c[1], b, n); }
            v.up_ctx = ctx;
            if (fence) fence_uploads(v);
            return v;                                                                   // other readers: msm_begin_multi (per component, on the device)
        }
        v.c[0] = dalloc(n * 32); CG(cg_dev_upload(ctx, v.c[0], a, n * 32));
        if (k() == 2) { v.c[1] = dalloc(n * 32); CG(cg_dev_upload(ctx, v.c[1], b, n * 32)); }
        v.ready = true;
        return v;
    }
    void fence_uploads(const ShareVec& v) {                                           // this context's stream: behind the last copy (they complete in order)
        if (!v.up_ctx) return;
        const int32_t tk = v.up[1] >= 0 ? v.up[1] : v.up[0];
        if (tk >= 0) CG(cg_copy_fence(ctx, tk));
    }
    Fr draw(const Fr* s) const { if (cursor >= rng_len) throw std::runtime_error("randomness stream exhausted"); return s[cursor]; }

    // evaluate_constraint for every row (traits.rs:180; plain.rs:243-258, rep3.rs:690-708) into a zero-padded length-m vector
    ShareVec evaluate_constraints(const DeviceMatrix& mt, const void* d_pub, uint32_t n_inputs, const ShareVec& wit, size_t m) {
        ShareVec out = alloc_vec(m);
        CG(cg_spmv_csr_dev(ctx, curve.id, mt.row_ptr, mt.col, mt.coeff, mt.rows, d_pub, n_inputs, party(), wit.c[0], wit.c[1], out.c[0], out.c[1]));
        return out;
    }
    // promote_to_trivial_shares (fieldshare.rs:262-283) + clone_from_slice (rep3.rs:710-725)
    // d_pub (optional): the same values already on the device — the copy is then enqueued like a kernel, the host does not wait
    void clone_public_into(ShareVec& dst, size_t dst_off, const std::vector<Fr>& pub, const void* d_pub = nullptr) {
        const int holder = mode != Mode::Rep3 ? 0 : (party() == 0 ? 0 : party() == 1 ? 1 : -1);   // REP3: ID0 -> a, ID1 -> b, ID2 -> nothing; plain / Shamir: the value itself
        if (holder < 0) return;
        if (d_pub) CG(cg_dev_copy_peer(ctx, (uint8_t*)dst.c[holder] + dst_off * 32, ctx, d_pub, pub.size() * 32));
        else CG(cg_dev_upload(ctx, (uint8_t*)dst.c[holder] + dst_off * 32, pub.data(), pub.size() * 32));
    }
    // mul_vec (traits.rs:164): plain.rs:219-224 ; rep3.rs:650-670 (local product + mask, send to next, receive from prev)
    // ---- page-locked staging rings for the asynchronous exchanges: chunks of XCHG_CHUNK elements travel over the
    // context's copy streams while the compute stream keeps running; a slot is reused once the copy that used it has completed
    static constexpr size_t XCHG_CHUNK_MAX = (size_t)1 << 17;                           // 4 MiB of field elements
    static constexpr int XCHG_SLOTS = 8;
    // chunk length for a vector of n elements: about n / 8, a power of two in [4096, 2^17] (page-locking memory is slow: small proofs get small rings)
    static size_t xchg_chunk(size_t n) { size_t c = 4096; while (c < XCHG_CHUNK_MAX && c * XCHG_SLOTS < n) c <<= 1; return c; }
    struct PinRing { uint8_t* base = nullptr; size_t chunk = 0; int32_t busy[XCHG_SLOTS]; int next = 0; int last = 0; };
    PinRing ring_out, ring_in;
    uint8_t* ring_slot(PinRing& r, size_t chunk) {
        if (r.chunk < chunk) {                                                          // first use, or a longer vector than before
            if (r.base) { CG(cg_ctx_sync(ctx)); CG(cg_host_free(r.base)); }
            void* p; CG(cg_host_alloc(XCHG_SLOTS * chunk * 32, &p)); r.base = (uint8_t*)p; r.chunk = chunk; for (int32_t& b : r.busy) b = -1;
        }
        r.last = r.next++ % XCHG_SLOTS;
        if (r.busy[r.last] >= 0) { CG(cg_copy_wait(ctx, r.busy[r.last])); r.busy[r.last] = -1; }
        return r.base + (size_t)r.last * r.chunk * 32;
    }
    // (every copy that touches a slot holds that slot's ticket: waiting for the tickets still marked busy is enough, a whole-context
    // synchronisation — five streams — per ring cost a finished proof 0.1 ms of tear-down)
    void release_rings() {
        for (PinRing* r : {&ring_out, &ring_in}) if (r->base) {
            bool waited = true;
            for (int32_t& b : r->busy) if (b >= 0) { if (cg_copy_wait(ctx, b)) waited = false; b = -1; }
            if (!waited) cg_ctx_sync(ctx);
            cg_host_free(r->base); r->base = nullptr; r->chunk = 0;
        }
    }
    std::vector<void*> deferred;                                                        // device buffers freed at the next quiet point
    void defer_free(void* p) { if (p) deferred.push_back(p); }
    void defer_vec(ShareVec& v) { for (int j = 0; j < 2; j++) { defer_free(v.c[j]); v.c[j] = nullptr; } }
    void free_deferred() { if (!deferred.empty()) CG(cg_dev_free_many(ctx, deferred.data(), deferred.size())); deferred.clear(); }   // one release mark for all of them
    // host (pageable) -> device through the ring, asynchronous; returns the ticket of the last chunk
    int32_t upload_staged(void* d_dst, const Fr* src, size_t n) {
        int32_t tk = -1;
        if (n && cg_host_is_pinned(src)) {                                             // the caller keeps this vector page-locked: DMA straight from it
            CG(cg_dev_upload_begin(ctx, d_dst, src, n * 32, 0, &tk));
            return tk;
        }
        const size_t ch = xchg_chunk(n);
        for (size_t off = 0; off < n; off += ch) {
            const size_t len = std::min(ch, n - off);
            uint8_t* slot = ring_slot(ring_in, ch);
            memcpy(slot, src + off, len * 32);
            CG(cg_dev_upload_begin(ctx, (uint8_t*)d_dst + off * 32, slot, len * 32, 0, &tk));
            ring_in.busy[ring_in.last] = tk;
        }
        return tk;
    }
    // mul_vec (rep3.rs:650-670) in two halves, so that the caller can enqueue independent work between the local product and the
    // exchange: `begin` masks and multiplies on the device and starts streaming the local product to the host; `finish` sends it to
    // the next party chunk by chunk while receiving the previous party's chunks, which go straight back up.  Plain / Shamir: `begin`
    // is the whole operation.
    // shorter vectors: one synchronous message (setting up rings and copy streams costs more than it hides); CGH_XCHG_ASYNC_MIN overrides (A/B runs)
    const size_t XCHG_ASYNC_MIN = (size_t)host_option(CGH_OPT_XCHG_ASYNC_MIN);   // (2^19 until round 4; one REP3 party at 2^17: 8.1 -> 6.7 ms, at 2^16 the single message is faster)
    // masks of the coming mul_vec calls, uploaded ahead of time (only from page-locked randomness streams, where the copy is a plain
    // asynchronous DMA): the product kernel then never waits for PCIe
    struct MaskSet { void* m1; void* m2; int32_t tk; size_t n, at; void* block = nullptr; bool owns = true; };   // block: m1 lies inside a block drawn for several calls; the LAST of them releases it
    // a randomness source that describes its ChaCha12 generators has its masks drawn by the backend (no host draws, no upload); short vectors
    // are not worth three launches and a stream synchronisation — up to 2^10 elements: a host draw costs ~70 ns per element (two generators,
    // rejection sampling), 0.6 ms per mul_vec at 2^13 (one REP3 party there: 2.75 ms with host draws, 2.14 with device draws; 2^11: 1.52 -> 1.40)
    const size_t DEVICE_MASKS_MIN = (size_t)host_option(CGH_OPT_DEVICE_MASKS_MIN);   // (the override lets the small fixtures take the device path)
    bool masks_on_device(void* d_m, size_t n) {
        if (!rsrc || n < DEVICE_MASKS_MIN) return false;
        void* tmp = nullptr;
        if (const int32_t rc = cg_dev_alloc(ctx, n * 32, &tmp)) {                       // no room for the second stream's draws: the host callback draws instead (generators untouched)
            if (rc == CG_ERR_OOM) return false;
            throw std::runtime_error(cg_last_error());
        }
        bool done = false;
        try { done = rsrc->masks_on_device(ctx, curve.id, n, d_m, tmp); } catch (...) { cg_dev_free(ctx, tmp); throw; }
        defer_free(tmp);
        return done;
    }
    std::deque<MaskSet> prefetched;
    void prefetch_masks(int count, size_t n) {
        if (mode != Mode::Rep3) return;
        // short vectors are drawn where mul_vec asks for them: nothing to prepare — and nothing to allocate and release again (a released
        // block costs a mark on every stream of the context: the two attempts of a small proof took 0.2 ms of its 1.6)
        if (n < XCHG_ASYNC_MIN && (!rsrc || n < DEVICE_MASKS_MIN)) return;
        if (rsrc) {                                                                     // drawn now, in the reference's order (both mul_vec calls precede every other draw)
            // the masks of `count` consecutive mul_vec calls are count * n consecutive draws of each generator: ONE device draw per generator
            // (one stream synchronisation each instead of `count`), cut into the calls' vectors
            if (count > 1) {
                void* block = nullptr;
                if (cg_dev_alloc(ctx, (size_t)count * n * 32, &block) == 0) {
                    if (masks_on_device(block, (size_t)count * n)) {
                        for (int i = 0; i < count; i++) { MaskSet ms{(char*)block + (size_t)i * n * 32, nullptr, -1, n, 0}; ms.block = block; ms.owns = i == count - 1; prefetched.push_back(ms); }
                        return;
                    }
                    CG(cg_dev_free(ctx, block));
                }
            }
            for (int i = 0; i < count; i++) {
                MaskSet ms{dalloc(n * 32), nullptr, -1, n, 0};
                if (masks_on_device(ms.m1, n)) {}                                       // any length from DEVICE_MASKS_MIN on
                else if (n >= XCHG_ASYNC_MIN) ms.tk = upload_staged(ms.m1, rsrc->masking_field_elements(n, mask_scratch(n)), n);
                else { CG(cg_dev_free(ctx, ms.m1)); return; }                           // short vectors from the host callback: drawn where mul_vec asks for them
                prefetched.push_back(ms);
            }
            return;
        }
        if (n < XCHG_ASYNC_MIN || !rng1 || !rng2) return;
        size_t at = cursor;
        for (int i = 0; i < count && at + n <= rng_len; i++, at += n) {
            if (!cg_host_is_pinned(rng1 + at) || !cg_host_is_pinned(rng2 + at)) return;
            MaskSet ms{dalloc(n * 32), dalloc(n * 32), -1, n, at};
            upload_staged(ms.m1, rng1 + at, n);
            ms.tk = upload_staged(ms.m2, rng2 + at, n);
            prefetched.push_back(ms);
        }
    }
    struct Down { uint8_t* slot; int32_t tk; };
    struct PendingMul { ShareVec out; bool exchange = false; std::deque<Down> down; size_t issued = 0; int32_t mark = -1;    // mark: where the stream produced `out` (cg_stream_mark)
                        std::shared_ptr<void> stage; int32_t stage_tk = -1; };                                                // single staged message: page-locked block [local | received], ticket of the download
    // single-message exchanges are staged in page-locked memory from 2^12 elements on; in a party with two contexts (from 2^15 variables) and from 2^14
    // elements on they cross PCIe on the chain context's copy streams, beside the main stream (one REP3 party, never / with two contexts / from 2^12 on:
    // 2^13 1.63-1.79 / - / 1.86-2.20 ms | 2^14, one context 2.04-2.09 / - / 2.23-2.46 | 2^15 2.42-2.53 / 2.17-2.27 / 2.30-2.33 | 2^16 2.83-2.99 / 2.91-2.95 / 2.85-2.93)
    static constexpr size_t XCHG_STAGED_MIN = (size_t)1 << 12;
    const size_t XCHG_COPY_STREAM_MIN = (size_t)host_option(CGH_OPT_XCHG_COPY_STREAM_MIN);   // (A/B knob)
    // start streaming chunks of the local product to the host, as many as the ring has room for
    void issue_downloads(PendingMul& pm, size_t upto) {
        const size_t n = pm.out.n, ch = xchg_chunk(n), nch = (n + ch - 1) / ch;
        while (pm.issued < nch && pm.issued < upto) {
            const size_t off = pm.issued * ch, len = std::min(ch, n - off);
            Down d; d.slot = ring_slot(ring_out, ch);
            // behind the product kernel, NOT behind the transforms the prover has enqueued since (they used to hold chunks 8 .. 31 back for 20 ms)
            if (pm.mark >= 0) CG(cg_dev_download_begin_after(ctx, d.slot, (const uint8_t*)pm.out.c[0] + off * 32, len * 32, pm.mark, &d.tk));
            else CG(cg_dev_download_begin(ctx, d.slot, (const uint8_t*)pm.out.c[0] + off * 32, len * 32, &d.tk));
            ring_out.busy[ring_out.last] = d.tk;
            pm.down.push_back(d); pm.issued++;
        }
    }
    PendingMul mul_vec_begin(const ShareVec& a, const ShareVec& b, bool exchange = true) {   // exchange = false: the masked local product only (additive share)
        PendingMul pm; ShareVec& out = pm.out; out.n = a.n;
        out.c[0] = dalloc(a.n * 32);
        if (mode != Mode::Rep3) CG(cg_vec_mul_dev(ctx, curve.id, out.c[0], a.c[0], b.c[0], a.n));
        if (mode == Mode::Plain) return pm;
        if (mode == Mode::Shamir) { if (exchange) out = degree_reduce_vec(out); return pm; }   // shamir.rs:609-623 (exchange = false: the degree-2t products)
        void* m1 = nullptr; void* m2 = nullptr; void* m1_block = nullptr; bool m1_owned = true;
        if (rsrc) {
            if (!prefetched.empty() && prefetched.front().n == a.n) {
                const MaskSet ms = prefetched.front(); prefetched.pop_front();
                m1 = ms.m1; m1_owned = ms.owns; if (ms.block) m1_block = ms.block;
                if (ms.tk >= 0) CG(cg_copy_fence(ctx, ms.tk));
            } else {
                m1 = dalloc(a.n * 32);
                if (masks_on_device(m1, a.n)) {}
                else if (a.n < XCHG_ASYNC_MIN) { std::vector<Fr> buf(a.n); CG(cg_dev_upload(ctx, m1, rsrc->masking_field_elements(a.n, buf.data()), a.n * 32)); }
                else { const int32_t tk = upload_staged(m1, rsrc->masking_field_elements(a.n, mask_scratch(a.n)), a.n); if (tk >= 0) CG(cg_copy_fence(ctx, tk)); }
            }
        } else {
        if (cursor + a.n > rng_len) throw std::runtime_error("randomness stream exhausted");
        if (!prefetched.empty() && prefetched.front().at == cursor && prefetched.front().n == a.n) {
            const MaskSet ms = prefetched.front(); prefetched.pop_front();
            m1 = ms.m1; m2 = ms.m2;
            if (ms.tk >= 0) CG(cg_copy_fence(ctx, ms.tk));
        } else {
            m1 = dalloc(a.n * 32); m2 = dalloc(a.n * 32);
        if (a.n < XCHG_ASYNC_MIN) { CG(cg_dev_upload(ctx, m1, rng1 + cursor, a.n * 32)); CG(cg_dev_upload(ctx, m2, rng2 + cursor, a.n * 32)); }
        else {
            upload_staged(m1, rng1 + cursor, a.n);
            const int32_t tk = upload_staged(m2, rng2 + cursor, a.n);
            if (tk >= 0) CG(cg_copy_fence(ctx, tk));                                   // uploads complete in order: the last ticket covers both masks
        }
        }
        cursor += a.n;
        CG(cg_vec_sub_dev(ctx, curve.id, m1, m1, m2, a.n));                           // masking_field_element = rand(rng1) - rand(rng2)
        }
        CG(cg_vec_rep3_mul_local_dev(ctx, curve.id, out.c[0], a.c[0], a.c[1], b.c[0], b.c[1], m1, a.n));
        if (m1_owned) defer_free(m1_block ? m1_block : m1);                            // (a block drawn for several calls is released with the last of them)
        defer_free(m2);
        if (!exchange) return pm;
        out.c[1] = dalloc(a.n * 32);
        pm.exchange = true;
        if (a.n >= XCHG_ASYNC_MIN) CG(cg_stream_mark(ctx, &pm.mark));
        if (a.n >= XCHG_ASYNC_MIN) issue_downloads(pm, XCHG_SLOTS - 1);                // ordered right behind the product, ahead of whatever the caller enqueues next
        else if (a.n >= XCHG_COPY_STREAM_MIN && aux) {                                 // (a party with a second context: this one is its chain context, with copy streams of its own)
            // One message, but not on the main stream: the download starts behind the PRODUCT (a mark), on the copy stream — the synchronous copy it
            // replaces queued behind the transforms the prover enqueues next (a 2^16 party: the first exchange waited for the four transforms of a and b,
            // themselves slowed by the accumulations running beside them)
            void* p = nullptr; CG(cg_host_alloc(2 * a.n * 32, &p)); pm.stage.reset(p, [](void* q) { cg_host_free(q); });
            CG(cg_stream_mark(ctx, &pm.mark));
            CG(cg_dev_download_begin_after(ctx, p, out.c[0], a.n * 32, pm.mark, &pm.stage_tk));
        }
        return pm;
    }
    ShareVec mul_vec_finish(PendingMul& pm) {
        if (!pm.exchange) return pm.out;
        ShareVec& out = pm.out;
        pm.exchange = false;
        if (out.n < XCHG_ASYNC_MIN) {                                                  // rep3.rs:661-669 as one message
            Marks mk("  mul_vec_finish (one message)", party() <= 0);
            // From 2^12 elements on the message is staged in page-locked memory (parked blocks of the host cache) and copied on the copy streams;
            // the range check of what arrived runs on the device behind the upload, as for the chunked exchange.
            if (pm.stage) {                                                            // from XCHG_COPY_STREAM_MIN elements on (mul_vec_begin)
                Fr* local = (Fr*)pm.stage.get(); Fr* recv = local + out.n;
                CG(cg_copy_wait(ctx, pm.stage_tk));
                mk.mark("download");
                net->send_next(local, out.n * 32);
                const void* direct = net->recv_prev_pinned(out.n * 32);                // the transport holds it in page-locked memory already
                if (!direct) { net->recv_prev(recv, out.n * 32); direct = recv; }
                mk.mark("send + receive");
                int32_t up = -1;
                CG(cg_dev_upload_begin(ctx, out.c[1], direct, out.n * 32, 0, &up));
                CG(cg_copy_fence(ctx, up));                                            // later launches see the received component ...
                check_received_dev(out.c[1], out.n);                                   // ... the range check first (read before the last opening, verify_received_vectors)
                CG(cg_copy_wait(ctx, up));                                             // the staging block / the transport's buffer is free again
                pm.stage.reset();
                mk.mark("upload");
                return out;
            }
            const bool staged = out.n >= XCHG_STAGED_MIN;                                // page-locked, but in stream order on the main stream
            struct Pinned { void* p = nullptr; ~Pinned() { if (p) cg_host_free(p); } } stage;
            std::vector<Fr> pageable(staged ? 0 : 2 * out.n);
            if (staged) CG(cg_host_alloc(2 * out.n * 32, &stage.p));
            Fr* local = staged ? (Fr*)stage.p : pageable.data(); Fr* recv = local + out.n;
            CG(cg_dev_download(ctx, local, out.c[0], out.n * 32));
            mk.mark("download");
            net->send_next(local, out.n * 32);
            const void* direct = staged ? net->recv_prev_pinned(out.n * 32) : nullptr;      // the transport holds it in page-locked memory already
            if (!direct) { net->recv_prev(recv, out.n * 32); direct = recv; }
            if (!staged) check_received(direct, out.n);
            mk.mark("send + receive");
            CG(cg_dev_upload(ctx, out.c[1], direct, out.n * 32));
            if (staged) check_received_dev(out.c[1], out.n);
            mk.mark("upload");
            return out;
        }
        const size_t n = out.n, XCHG_CHUNK = xchg_chunk(n), nch = (n + XCHG_CHUNK - 1) / XCHG_CHUNK;
        std::deque<Down>& down = pm.down;
        int32_t up = -1;
        for (size_t c = 0; c < nch; c++) {
            issue_downloads(pm, c + XCHG_SLOTS - 1);                                   // keep the download stream ahead of the sender
            const size_t off = c * XCHG_CHUNK, len = std::min(XCHG_CHUNK, n - off);
            CG(cg_copy_wait(ctx, down.front().tk));
            net->send_next(down.front().slot, len * 32);                               // chunked send_next_many
            down.pop_front();
            if (const void* direct = net->recv_prev_pinned(len * 32)) {                    // the transport holds it in page-locked memory already
                CG(cg_dev_upload_begin(ctx, (uint8_t*)out.c[1] + off * 32, direct, len * 32, 0, &up));
            } else {
                uint8_t* slot = ring_slot(ring_in, XCHG_CHUNK);
                net->recv_prev(slot, len * 32);
                CG(cg_dev_upload_begin(ctx, (uint8_t*)out.c[1] + off * 32, slot, len * 32, 0, &up));
                ring_in.busy[ring_in.last] = up;
            }
        }
        if (up >= 0) CG(cg_copy_fence(ctx, up));                                       // later launches see the received component
        check_received_dev(out.c[1], n);
        pm.exchange = false;
        return out;
    }
    ShareVec mul_vec(const ShareVec& a, const ShareVec& b) { PendingMul pm = mul_vec_begin(a, b); ShareVec r = mul_vec_finish(pm); free_deferred(); return r; }
    // before the context goes away (idempotent; also run by the destructor when a party dies with an exception)
    void shutdown() {
        if (d_bad) { deferred.push_back(d_bad); d_bad = nullptr; }
        for (auto& ms : prefetched) { if (ms.owns) deferred.push_back(ms.block ? ms.block : ms.m1); if (ms.m2) deferred.push_back(ms.m2); }
        prefetched.clear();
        if (!deferred.empty()) cg_dev_free_many(ctx, deferred.data(), deferred.size());
        deferred.clear();
        if (!mask_bufs.empty()) { cg_ctx_sync(ctx); for (void* p : mask_bufs) cg_host_free(p); mask_bufs.clear(); }   // uploads from them may still be in flight
        release_rings(); release_pre();
        if (aux) { if (owns_aux) cg_ctx_destroy(aux); aux = nullptr; }
    }
    void sync_other_contexts() {                                                           // error paths: see VecGuard
        if (aux) cg_ctx_sync(aux);
        if (md) for (const WorkerDevice& w : md->workers) { if (w.ctx) cg_ctx_sync(w.ctx); if (w.chain) cg_ctx_sync(w.chain); }
    }
    void use_second_context(cg_ctx* second) { aux = second; }                           // owned from here on (shutdown destroys it)
    ~HipDriver() { shutdown(); }
    // ---- vector forms of rand / mul_open_many / open_many used by co-plonk (rep3.rs:544-558,595-598,620-628,738-757)
    int public_component() const { return mode != Mode::Rep3 ? 0 : (party() == 0 ? 0 : party() == 1 ? 1 : -1); }   // add_with_public: who holds a public addend
    // broadcast_next(num) of a vector + reconstruction with the given Lagrange table (shamir/network.rs:233-266, shamir.rs:581-601,684-711)
    std::vector<Fr> shamir_open_vec(const std::vector<Fr>& mine, const std::vector<Fr>& lagrange) {
        const int np = snet->num_parties(), me = snet->id(), num = (int)lagrange.size();
        const size_t n = mine.size();
        for (int sft = 1; sft < num; sft++) snet->send((me + sft) % np, mine.data(), n * 32);
        std::vector<Fr> out(n), got(n);
        for (size_t i = 0; i < n; i++) out[i] = fr_mul(curve, mine[i], lagrange[0]);
        for (int r = 1; r < num; r++) { snet->recv((me + np - r) % np, got.data(), n * 32); check_received(got.data(), n); for (size_t i = 0; i < n; i++) out[i] = fr_add(curve, out[i], fr_mul(curve, got[i], lagrange[r])); }
        return out;
    }
    ShareVec rand_vec(size_t n) {
        if (mode == Mode::Shamir) { std::vector<Fr> r(n); for (size_t i = 0; i < n; i++) r[i] = get_pair().first; return upload_vec(r.data(), nullptr, n); }   // shamir.rs:570-573
        if (mode != Mode::Rep3) throw std::runtime_error("rand_vec: REP3 / Shamir only");
        if (rsrc) { std::vector<Fr> a(n), b(n); for (size_t i = 0; i < n; i++) rsrc->random_fes(a[i], b[i]); return upload_vec(a.data(), b.data(), n); }
        if (cursor + n > rng_len) throw std::runtime_error("randomness stream exhausted");
        ShareVec v = upload_vec(rng1 + cursor, rng2 + cursor, n); cursor += n;
        return v;
    }
    // a * b opened: a public device vector (caller frees)
    void* mul_open_vec(const ShareVec& a, const ShareVec& b) {
        const size_t n = a.n;
        void* out = dalloc(n * 32);
        if (mode != Mode::Rep3) CG(cg_vec_mul_dev(ctx, curve.id, out, a.c[0], b.c[0], n));
        if (mode == Mode::Plain) return out;
        if (mode == Mode::Shamir) {                                                   // degree-2t product opened from 2t + 1 shares (shamir.rs:684-711)
            // broadcast_next(2t) + reconstruction (shamir/network.rs:233-266): the Lagrange combination runs on the device
            const int np = snet->num_parties(), me = snet->id(), num = (int)open_lagrange_2t.size();
            std::vector<Fr> buf(n); CG(cg_dev_download(ctx, buf.data(), out, n * 32));
            for (int sft = 1; sft < num; sft++) snet->send((me + sft) % np, buf.data(), n * 32);
            std::vector<Term> terms{{out, 0, 1, open_lagrange_2t[0]}};
            std::vector<void*> got;
            for (int r = 1; r < num; r++) {
                snet->recv((me + np - r) % np, buf.data(), n * 32); check_received(buf.data(), n);
                void* d = dalloc(n * 32); CG(cg_dev_upload(ctx, d, buf.data(), n * 32));
                got.push_back(d); terms.push_back({d, 0, 1, open_lagrange_2t[r]});
            }
            lincomb(out, 0, 1, n, terms);
            for (void* d : got) CG(cg_dev_free(ctx, d));
            return out;
        }
        void* m1 = dalloc(n * 32); void* m2 = dalloc(n * 32);
        if (rsrc) { std::vector<Fr> buf(n); CG(cg_dev_upload(ctx, m1, rsrc->masking_field_elements(n, buf.data()), n * 32)); }
        else {
            if (cursor + n > rng_len) throw std::runtime_error("randomness stream exhausted");
            CG(cg_dev_upload(ctx, m1, rng1 + cursor, n * 32)); CG(cg_dev_upload(ctx, m2, rng2 + cursor, n * 32)); cursor += n;
            CG(cg_vec_sub_dev(ctx, curve.id, m1, m1, m2, n));
        }
        CG(cg_vec_rep3_mul_local_dev(ctx, curve.id, out, a.c[0], a.c[1], b.c[0], b.c[1], m1, n));
        std::vector<Fr> mine(n), p(n), q(n);
        CG(cg_dev_download(ctx, mine.data(), out, n * 32));
        net->send_next(mine.data(), n * 32); net->send_prev(mine.data(), n * 32);
        net->recv_prev(p.data(), n * 32); net->recv_next(q.data(), n * 32); check_received(p.data(), n); check_received(q.data(), n);
        CG(cg_dev_upload(ctx, m1, p.data(), n * 32)); CG(cg_dev_upload(ctx, m2, q.data(), n * 32));
        CG(cg_vec_add_dev(ctx, curve.id, out, out, m1, n)); CG(cg_vec_add_dev(ctx, curve.id, out, out, m2, n));
        CG(cg_dev_free(ctx, m1)); CG(cg_dev_free(ctx, m2));
        return out;
    }
    std::vector<Fr> open_many(const std::vector<FieldShare>& a) {
        std::vector<Fr> out(a.size());
        if (mode == Mode::Plain) { for (size_t i = 0; i < a.size(); i++) out[i] = a[i].c[0]; return out; }
        if (mode == Mode::Shamir) { std::vector<Fr> mine(a.size()); for (size_t i = 0; i < a.size(); i++) mine[i] = a[i].c[0]; return shamir_open_vec(mine, open_lagrange_t); }   // shamir.rs:581-601
        std::vector<Fr> bs(a.size()), cs(a.size());
        for (size_t i = 0; i < a.size(); i++) bs[i] = a[i].c[1];
        net->send_next(bs.data(), bs.size() * 32); net->recv_prev(cs.data(), cs.size() * 32); check_received(cs.data(), cs.size());
        for (size_t i = 0; i < a.size(); i++) out[i] = fr_add(curve, fr_add(curve, a[i].c[0], a[i].c[1]), cs[i]);
        return out;
    }

    // FFTProvider (traits.rs:535-558): both share components in one launch
    void fft_in_place(ShareVec& v, const Fr& group_gen) { CG(cg_ntt_dev(ctx, curve.id, v.c, k(), v.n, group_gen.v, 0, nullptr)); }
    void ifft_in_place(ShareVec& v, const Fr& group_gen) { CG(cg_ntt_dev(ctx, curve.id, v.c, k(), v.n, group_gen.v, 1, nullptr)); }
    void distribute_powers_and_mul_by_const(ShareVec& v, const Fr& g, const Fr& c) { for (int j = 0; j < k(); j++) CG(cg_vec_distribute_powers_dev(ctx, curve.id, v.c[j], v.n, g.v, c.v)); }
    // fused ifft_in_place + distribute_powers_and_mul_by_const(g, 1): one HBM round trip less per vector
    void ifft_coset_in_place(ShareVec& v, const Fr& group_gen, const Fr& g) { CG(cg_ntt_dev(ctx, curve.id, v.c, k(), v.n, group_gen.v, 1, g.v)); }
    // ifft_in_place; distribute_powers_and_mul_by_const(g, 1); fft_in_place (groth16.rs:175-188) as one call: no permutation passes in between
    void ifft_coset_fft_in_place(ShareVec& v, const Fr& group_gen, const Fr& g) { CG(cg_ntt_coset_pair_dev(ctx, curve.id, v.c, k(), v.n, group_gen.v, g.v)); }
    // the same for two share vectors in ONE sequence of launches (a and b of groth16.rs:175-188: five kernels instead of ten — beside the
    // accumulations of another context every launch waits for workgroup slots, so the chain's latency goes with the number of kernels)
    void ifft_coset_fft_in_place2(ShareVec& u, ShareVec& v, const Fr& group_gen, const Fr& g) {
        void* p[4]; int m = 0;
        for (int j = 0; j < k(); j++) p[m++] = u.c[j];
        for (int j = 0; j < k(); j++) p[m++] = v.c[j];
        CG(cg_ntt_coset_pair_dev(ctx, curve.id, p, m, u.n, group_gen.v, g.v));
    }
    void sub_assign_vec(ShareVec& a, const ShareVec& b) { for (int j = 0; j < k(); j++) CG(cg_vec_sub_dev(ctx, curve.id, a.c[j], a.c[j], b.c[j], a.n)); }

    // MSMProvider::msm_public_points (traits.rs:561-568) on a sub-slice of a registered table
    PointShare msm_public_points(const cg_bases* bases, int group, size_t off, size_t n, const ShareVec& s) {
        Bytes out(curve.jac(group) * k());
        const void* sc[2] = {s.c[0], s.c[1]};
        CG(cg_msm_dev(ctx, bases, off, n, sc, k(), out.data()));
        PointShare r;
        for (int j = 0; j < k(); j++) r.c[j] = Point{Bytes(out.begin() + j * curve.jac(group), out.begin() + (j + 1) * curve.jac(group)), group};
        if (k() == 1) r.c[1] = pt_inf(curve, group);
        return r;
    }
    // The same MSMs started early and collected later (cg_msm_dev_begin_multi / cg_msm_end): the four queries over the private witness
    // (groth16.rs:251,267,284,298) share one scalar decomposition and run on a second context (`aux`, own streams) while the witness
    // map and its exchanges occupy the first.  MSMs involve no network, so the party-to-party message order is the reference's.
    cg_ctx* aux = nullptr; bool owns_aux = true;       // a session lends its contexts (owns_aux = false)
    struct PendingMsm {
        cg_ctx* on = nullptr; std::vector<int32_t> tickets; std::vector<int> groups; int k = 0;     // k: share components multiplied (0 = the driver's)
        struct Part { cg_ctx* on; std::vector<int32_t> tickets; void* sc[2]; };     // the same MSMs over the slices held by further GPUs
        std::vector<Part> parts;
        // msm_begin_aux_split: table `split_table` was multiplied in pieces — component a over two point ranges (tickets lo, hi), component b
        // on its own (ticket b, -1 for one-component drivers); tickets[split_table] is unused
        int split_table = -1; int32_t split_lo = -1, split_hi = -1, split_b = -1;
    };
    // Several GPUs: every MSM range is cut into one contiguous slice per device (the primary context's device holds slice 0); the scalar
    // slices travel device to device (cg_dev_copy_peer, xGMI), each device runs the bucket method on its slice, and the partial sums
    // — one Jacobian point per table, component and device — are added on the host (RCCL has no EC-add reduction, and a few hundred
    // bytes per proof need no collective).  MSMProvider::msm_public_points (rep3.rs:934-947) is linear in the (scalar, point) pairs.
    const MultiDevice* md = nullptr;
    // multi-device proofs whose entry point left the private witness on the HOST (ShareVec with null components): the distributed witness
    // map uploads rows of these vectors over every device's own link (multidev.hpp)
    const Fr* host_wit[2] = {nullptr, nullptr};
    PendingMsm msm_begin_sharded(const DeviceZKey& dz, bool aux_tables, const ShareVec& s) {
        const size_t lo = aux_tables ? dz.aux_lo : dz.h_lo, n = aux_tables ? dz.aux_n : dz.h_n;
        ShareVec mine; mine.n = n; for (int j = 0; j < k(); j++) mine.c[j] = (char*)s.c[j] + lo * 32;
        mine.up_ctx = s.up_ctx; mine.up[0] = s.up[0]; mine.up[1] = s.up[1];            // a slice of shares still crossing PCIe: the schedules wait on the device, not the host (it used to block 5 ms here at 2^22)
        PendingMsm p = aux_tables ? msm_begin_multi({dz.a, dz.b1, dz.b2, dz.l}, {0, 0, 0, 0}, {CG_G1, CG_G1, CG_G2, CG_G1}, n, mine, true)   // table order = CoGroth16::prove's AUX_* indices
                                  : msm_begin_multi({dz.h}, {0}, {CG_G1}, n, mine, false);
        if (!md) return p;
        if (emulate_only_device() >= 0) return p;                                         // planning builds only (base.hpp): the primary device's share alone
        for (const WorkerDevice& w : md->workers) {
            const DeviceZKey& wz = *w.dz;
            const size_t wlo = aux_tables ? wz.aux_lo : wz.h_lo, wn = aux_tables ? wz.aux_n : wz.h_n;
            PendingMsm::Part part{w.ctx, {}, {nullptr, nullptr}};
            for (int j = 0; j < k(); j++) {
                CG(cg_dev_alloc(w.ctx, std::max<size_t>(wn * 32, 32), &part.sc[j]));
                CG(cg_dev_copy_peer(w.ctx, part.sc[j], ctx, (const char*)s.c[j] + wlo * 32, wn * 32));
            }
            std::vector<const cg_bases*> tabs = aux_tables ? std::vector<const cg_bases*>{wz.a, wz.b1, wz.b2, wz.l} : std::vector<const cg_bases*>{wz.h};
            std::vector<size_t> offs(tabs.size(), 0);
            part.tickets.resize(tabs.size());
            const void* sc[2] = {part.sc[0], part.sc[1]};
            begin_multi_ordered(w.ctx, tabs, offs, aux_tables ? std::vector<int>{CG_G1, CG_G1, CG_G2, CG_G1} : std::vector<int>{CG_G1}, wn, sc, part.tickets);
            p.parts.push_back(part);
        }
        return p;
    }
    void msm_release(PendingMsm& p) {      // after the last msm_finish: the slices' scalar copies
        for (auto& part : p.parts) for (int j = 0; j < 2; j++) if (part.sc[j]) { cg_dev_free(part.on, part.sc[j]); part.sc[j] = nullptr; }
        p.parts.clear();
    }
    // One cg_msm_dev_begin_multi call with the launch order of its (table, share component) pairs chosen here (tickets come back in the
    // caller's table order).  Round 4: CG_OPT_MSM_TABLE_ORDER = 2 — the G1 pairs in serpentine order [a b1 l][l b1 a] with the two G2
    // accumulations TOGETHER after the first CGH_G2_AFTER of them (default 2: [a b1 | b2 b2 | l l b1 a]).  The G2 launches are not cut into
    // chip-loads any more (rounds 2-3: ~2 ms per launch, CG_OPT_MSM_G2_SLICES): they sit in the middle of the call, beside the chain
    // context's transforms, and the call's tail — where a launch runs alone on the chip — is G1 launches only.  Same-box sweeps of the
    // position (0 .. 6) all land within 1.5 ms of each other once the party's contexts are made in a fixed order (capi_session.cpp); the
    // round-3 order (G2 first in every component, sliced) is 3-5 ms slower (profiles/r04_entry_ab_*.txt).  CGH_G2_ORDER=first restores it for A/B runs.
    void begin_multi_ordered(cg_ctx* on, const std::vector<const cg_bases*>& tables, const std::vector<size_t>& offsets, const std::vector<int>& groups, size_t n,
                             const void* const* sc, std::vector<int32_t>& tickets) {
        static const bool g2_first = tune_env("CGH_G2_ORDER") && !strcmp(tune_env("CGH_G2_ORDER"), "first");   // A/B knob
        std::vector<size_t> ord;
        for (size_t i = 0; i < tables.size(); i++) if ((groups[i] == CG_G2) == g2_first) ord.push_back(i);
        for (size_t i = 0; i < tables.size(); i++) if ((groups[i] == CG_G2) != g2_first) ord.push_back(i);
        static const int g2_after = tune_env("CGH_G2_AFTER") ? atoi(tune_env("CGH_G2_AFTER")) : 2;   // A/B knob: G1 accumulations in front of the two G2 ones
        CG(cg_ctx_set_option(on, CG_OPT_MSM_TABLE_ORDER, g2_first ? 0 : 2));
        CG(cg_ctx_set_option(on, CG_OPT_MSM_G2_AFTER, g2_after));
        int64_t chunk = 0; CG(cg_ctx_get_option(on, CG_OPT_MSM_CHUNK, &chunk));
        CG(cg_ctx_set_option(on, CG_OPT_MSM_G2_SLICES, g2_first && chunk ? 1 : 0));
        std::vector<const cg_bases*> t(tables.size()); std::vector<size_t> o(tables.size()); std::vector<int32_t> tk(tables.size());
        for (size_t j = 0; j < ord.size(); j++) { t[j] = tables[ord[j]]; o[j] = offsets[ord[j]]; }
        CG(cg_msm_dev_begin_multi(on, (int32_t)t.size(), t.data(), o.data(), n, sc, k(), tk.data()));
        tickets.resize(tables.size());
        for (size_t j = 0; j < ord.size(); j++) tickets[ord[j]] = tk[j];
    }
    PendingMsm msm_begin_multi(const std::vector<const cg_bases*>& tables, const std::vector<size_t>& offsets, const std::vector<int>& groups, size_t n, const ShareVec& s, bool on_aux) {
        PendingMsm p; p.on = on_aux && aux ? aux : ctx; p.groups = groups; p.tickets.resize(tables.size()); p.k = k();
        // REP3 at sizes where the exchanges are asynchronous: these MSMs run beside the witness map's dependency chain (product -> down ->
        // peer -> up, twice) on the other context; shorter-lived workgroups let the chain's kernels onto the chip sooner (2^22: one party
        // alone 107 -> 97 ms)
        static const uint32_t bulk_chunk = tune_env("CGH_BULK_CHUNK") ? (uint32_t)atoi(tune_env("CGH_BULK_CHUNK")) : 64u;   // tuning knob
        static const uint32_t plain_chunk = tune_env("CGH_PLAIN_CHUNK") ? (uint32_t)atoi(tune_env("CGH_PLAIN_CHUNK")) : 0u;  // tuning knob
        if (p.on != ctx) CG(cg_msm_set_chunk(p.on, mode == Mode::Rep3 && n >= ((size_t)1 << 20) ? bulk_chunk : plain_chunk));   // (below 2^20 the shorter chunks cost more than they return: 2^19 15.9 -> 13.9 ms)
        const void* sc[2] = {s.c[0], s.c[1]};
        if (p.on != ctx) {
            if (s.up_ctx) {                                                             // fresh uploads: component j's schedule waits for ITS copy on the device,
                for (int j = 0; j < k(); j++) if (s.up[j] >= 0) CG(cg_msm_scalars_after(p.on, j, s.up_ctx, s.up[j]));   // a is accumulated while b is still crossing PCIe
            } else if (!s.ready) CG(cg_ctx_sync(ctx));                                  // the scalars were produced on this driver's stream
        }
        begin_multi_ordered(p.on, tables, offsets, groups, n, sc, p.tickets);
        return p;
    }
    // The aux MSMs of a LARGE proof whose witness is still crossing PCIe (ShareVec::up_first): the chip used to idle until component a had
    // landed completely and been scheduled (2.4 + 0.8 ms at 2^22: most of what the entry costs beyond the resident step).  Now
    // the FIRST table of the call is multiplied in pieces — component a over the first half of the points as soon as that half is on the
    // device, then over the second half, then (after the other tables) component b — while the other tables keep the one call with one
    // schedule per scalar vector.  Costs two half-size schedules, one more schedule of b and two more bucket sets (side streams), and the
    // 6 % a half-size accumulation is slower; buys ~1.6 ms of otherwise idle chip.  Same points: MSMs are linear in the (scalar, point) pairs.
    PendingMsm msm_begin_aux_split(const std::vector<const cg_bases*>& tables, const std::vector<size_t>& offsets, const std::vector<int>& groups, size_t n, const ShareVec& s, size_t first) {
        PendingMsm p; p.on = aux ? aux : ctx; p.groups = groups; p.tickets.assign(tables.size(), -1); p.k = k();
        static const uint32_t bulk_chunk = tune_env("CGH_BULK_CHUNK") ? (uint32_t)atoi(tune_env("CGH_BULK_CHUNK")) : 64u;
        if (p.on != ctx) CG(cg_msm_set_chunk(p.on, mode == Mode::Rep3 && n >= ((size_t)1 << 20) ? bulk_chunk : 0u));
        p.split_table = (int)first;
        const cg_bases* t0 = tables[first]; const size_t h = s.first_n;
        auto one = [&](size_t off, size_t cnt, const void* sc, int32_t after, int32_t* tk) {
            if (after >= 0) CG(cg_msm_scalars_after(p.on, 0, s.up_ctx, after));
            const size_t o = offsets[first] + off; const void* scs[1] = {sc};
            CG(cg_msm_dev_begin_multi(p.on, 1, &t0, &o, cnt, scs, 1, tk));
        };
        one(0, h, s.c[0], s.up_first, &p.split_lo);
        one(h, n - h, (const uint8_t*)s.c[0] + h * 32, s.up[0], &p.split_hi);
        {   // the other tables: one call, one schedule per component
            std::vector<const cg_bases*> t; std::vector<size_t> o; std::vector<int> g; std::vector<size_t> idx;
            for (size_t i = 0; i < tables.size(); i++) if (i != first) { t.push_back(tables[i]); o.push_back(offsets[i]); g.push_back(groups[i]); idx.push_back(i); }
            for (int j = 0; j < k(); j++) if (s.up[j] >= 0) CG(cg_msm_scalars_after(p.on, j, s.up_ctx, s.up[j]));
            const void* sc[2] = {s.c[0], s.c[1]};
            std::vector<int32_t> tk;
            begin_multi_ordered(p.on, t, o, g, n, sc, tk);
            for (size_t i = 0; i < idx.size(); i++) p.tickets[idx[i]] = tk[i];
        }
        if (k() == 2) one(0, n, s.c[1], s.up[1], &p.split_b);
        return p;
    }
    PointShare msm_finish(PendingMsm& p, size_t i) {
        const int group = p.groups[i], kk = p.k ? p.k : k();
        if ((int)i == p.split_table) {                                                  // the pieces of msm_begin_aux_split, folded (MSMs are linear in the pairs)
            Bytes one(curve.jac(group));
            auto take = [&](int32_t tk) { CG(cg_msm_end(p.on, tk, one.data())); return Point{one, group}; };
            PointShare r;
            r.c[0] = take(p.split_lo); r.c[0] = pt_add(curve, r.c[0], take(p.split_hi));
            r.c[1] = p.split_b >= 0 ? take(p.split_b) : pt_inf(curve, group);
            return r;
        }
        Bytes out(curve.jac(group) * kk);
        CG(cg_msm_end(p.on, p.tickets[i], out.data()));
        PointShare r;
        for (int j = 0; j < kk; j++) r.c[j] = Point{Bytes(out.begin() + j * curve.jac(group), out.begin() + (j + 1) * curve.jac(group)), group};
        for (auto& part : p.parts) {                                                    // slices on further GPUs: fold the partial sums
            CG(cg_msm_end(part.on, part.tickets[i], out.data()));
            for (int j = 0; j < kk; j++) r.c[j] = pt_add(curve, r.c[j], Point{Bytes(out.begin() + j * curve.jac(group), out.begin() + (j + 1) * curve.jac(group)), group});
        }
        if (kk == 1) r.c[1] = pt_inf(curve, group);
        return r;
    }
    // additive -> replicated for a handful of points in ONE round: every party sends its own component to the next party and takes the
    // previous party's as its second component (the pair (x_i, x_{i-1}) of rep3/pointshare.rs:14-17).  Used by the additive-quotient variant.
    void reshare_points(const std::vector<PointShare*>& pts) {
        if (mode != Mode::Rep3) return;
        Bytes msg;
        for (PointShare* ps : pts) { Bytes aff = pt_to_affine(curve, ps->c[0]); msg.insert(msg.end(), aff.begin(), aff.end()); }
        net->send_next(msg.data(), msg.size());
        Bytes got(msg.size()); net->recv_prev(got.data(), got.size());
        size_t at = 0;
        for (PointShare* ps : pts) { const int g = ps->c[0].group; ps->c[1] = received_point(g, got.data() + at); at += curve.aff(g); }
    }
    // rand (rep3.rs:595-598; plain: supplied by the caller)
    FieldShare rand() {
        if (mode == Mode::Shamir) { FieldShare f; f.c[0] = get_pair().first; f.c[1] = f.c[0]; return f; }   // shamir.rs:570-573
        FieldShare f;
        if (rsrc) { rsrc->random_fes(f.c[0], f.c[1]); return f; }
        f.c[0] = draw(rng1); f.c[1] = draw(rng2); cursor++; return f;
    }
    // mul (rep3.rs:503-511 / plain a*b)
    FieldShare mul(const FieldShare& a, const FieldShare& b) {
        FieldShare r;
        if (mode == Mode::Plain) { r.c[0] = fr_mul(curve, a.c[0], b.c[0]); r.c[1] = r.c[0]; return r; }
        if (mode == Mode::Shamir) { r.c[0] = degree_reduce(fr_mul(curve, a.c[0], b.c[0])); r.c[1] = r.c[0]; return r; }   // shamir.rs:481-488
        Fr local = fr_add(curve, fr_add(curve, fr_mul(curve, a.c[0], b.c[0]), fr_mul(curve, a.c[0], b.c[1])), fr_mul(curve, a.c[1], b.c[0]));
        if (rsrc) { Fr buf; local = fr_add(curve, local, *rsrc->masking_field_elements(1, &buf)); }
        else { local = fr_add(curve, local, fr_sub(curve, draw(rng1), draw(rng2))); cursor++; }
        net->send_next(local.v, 32);
        Fr prev; net->recv_prev(prev.v, 32); check_received(prev.v, 1);
        r.c[0] = local; r.c[1] = prev;
        return r;
    }
    PointShare scalar_mul_public_point(const Point& p, const FieldShare& s, const cg_fixed_base* tab = nullptr) {   // rep3.rs:820-825 (tab: p's window table, if the session holds one)
        PointShare r;
        if (!tab && k() == 2) {                                                          // two variable-base products: side by side (see scalar_mul)
            auto other = Helpers::get().run([&] { return pt_mul_fixed(curve, nullptr, p, s.c[1]); });
            try { r.c[0] = pt_mul_fixed(curve, nullptr, p, s.c[0]); } catch (...) { other.wait(); throw; }   // (the helper reads this frame)
            r.c[1] = other.get();
            return r;
        }
        for (int j = 0; j < 2; j++) r.c[j] = j < k() ? pt_mul_fixed(curve, tab, p, s.c[j]) : pt_inf(curve, p.group); return r;
    }
    PointShare scalar_mul(const PointShare& a, const FieldShare& b) {           // rep3.rs:835-847, pointshare.rs:117-124
        PointShare r;
        if (mode == Mode::Plain) { r.c[0] = pt_mul(curve, a.c[0], b.c[0]); r.c[1] = pt_inf(curve, a.c[0].group); return r; }
        if (mode == Mode::Shamir) { r.c[0] = degree_reduce_point(pt_mul(curve, a.c[0], b.c[0])); r.c[1] = pt_inf(curve, a.c[0].group); return r; }   // shamir.rs:769-776
        // three independent variable-base products (~60 us each in G1 on the host): two of them on helper threads (Helpers) — they sit between the
        // last MSM result and the proof, on every proof's tail
        auto p1 = Helpers::get().run([&] { return pt_mul(curve, a.c[1], b.c[0]); });
        auto p2 = Helpers::get().run([&] { return pt_mul(curve, a.c[0], b.c[1]); });
        Point p0; try { p0 = pt_mul(curve, a.c[0], b.c[0]); } catch (...) { p1.wait(); p2.wait(); throw; }   // (the helpers read this frame)
        p1.wait(); p2.wait();                                      // both done before either result (or exception) leaves this frame
        Point local = pt_add(curve, pt_add(curve, p0, p1.get()), p2.get());
        if (rsrc) { Point m{Bytes(curve.jac(a.c[0].group)), a.c[0].group}; rsrc->masking_ec_element(m.group, m.b.data()); local = pt_add(curve, local, m); }   // rngs.rs:48-51
        else {
            const int g = a.c[0].group;                            // masking_ec_element: G*rand(rng1) - G*rand(rng2)
            local = pt_add(curve, local, pt_sub(curve, pt_mul_generator(curve, g, draw(rng1)), pt_mul_generator(curve, g, draw(rng2)))); cursor++;
        }
        Bytes aff = pt_to_affine(curve, local);                    // points cross the wire in affine form (ark-serialize)
        net->send_next(aff.data(), aff.size());
        Bytes prev(aff.size()); net->recv_prev(prev.data(), prev.size());
        r.c[0] = local; r.c[1] = received_point(local.group, prev.data());
        return r;
    }
    void add_assign_points(PointShare& a, const PointShare& b) { for (int j = 0; j < k(); j++) a.c[j] = pt_add(curve, a.c[j], b.c[j]); }
    void sub_assign_points(PointShare& a, const PointShare& b) { for (int j = 0; j < k(); j++) a.c[j] = pt_sub(curve, a.c[j], b.c[j]); }
    void add_assign_points_public(PointShare& a, const Point& b) {              // rep3.rs:804-818: ID0 -> a, ID1 -> b, ID2 -> nothing
        if (mode != Mode::Rep3 || party() == 0) a.c[0] = pt_add(curve, a.c[0], b);       // Shamir: every party adds (shamir.rs:733-735)
        else if (party() == 1) a.c[1] = pt_add(curve, a.c[1], b);
    }
    Point open_point(const PointShare& a) {                                      // rep3.rs:849-853
        if (mode == Mode::Plain) return a.c[0];
        if (mode == Mode::Shamir) return shamir_open_point(a.c[0]);
        Bytes mine = pt_to_affine(curve, a.c[1]);
        net->send_next(mine.data(), mine.size());
        Bytes prev(mine.size()); net->recv_prev(prev.data(), prev.size());
        return pt_add(curve, pt_add(curve, a.c[0], a.c[1]), received_point(a.c[0].group, prev.data()));
    }
    std::pair<Point, Point> open_two_points(const PointShare& a, const PointShare& b) {   // rep3.rs:865-877
        if (mode == Mode::Plain) return {a.c[0], b.c[0]};
        if (mode == Mode::Shamir) {   // shamir.rs:808-824 (one message per point here)
            // both points are sent first, the G2 point's own term (a 254-bit product, as long as the whole G1 opening) runs on a helper under the G1
            // opening; the messages keep their order on every channel (G1 then G2)
            const int np = snet->num_parties(), me = snet->id();
            const Bytes m1 = pt_to_affine(curve, a.c[0]), m2 = pt_to_affine(curve, b.c[0]);
            for (int sft = 1; sft <= sh_t; sft++) { snet->send((me + sft) % np, m1.data(), m1.size()); snet->send((me + sft) % np, m2.data(), m2.size()); }
            auto own2 = Helpers::get().run([&] { return pt_mul(curve, b.c[0], open_lagrange_t[0]); });
            struct Joined { std::future<Point>& f; ~Joined() { if (f.valid()) f.wait(); } } joined{own2};     // (the helper reads this frame)
            Point r1 = pt_mul(curve, a.c[0], open_lagrange_t[0]);
            std::vector<Point> theirs2;
            for (int r = 1; r <= sh_t; r++) {
                Bytes b1(m1.size()), b2(m2.size());
                snet->recv((me + np - r) % np, b1.data(), b1.size()); snet->recv((me + np - r) % np, b2.data(), b2.size());
                theirs2.push_back(received_point(CG_G2, b2.data()));
                r1 = pt_add(curve, r1, pt_mul(curve, received_point(CG_G1, b1.data()), open_lagrange_t[r]));
            }
            Point r2 = own2.get();
            for (int r = 1; r <= sh_t; r++) r2 = pt_add(curve, r2, pt_mul(curve, theirs2[(size_t)r - 1], open_lagrange_t[r]));
            return {r1, r2};
        }
        Bytes m1 = pt_to_affine(curve, a.c[1]), m2 = pt_to_affine(curve, b.c[1]);
        Bytes msg(m1); msg.insert(msg.end(), m2.begin(), m2.end());
        net->send_next(msg.data(), msg.size());
        Bytes prev(msg.size()); net->recv_prev(prev.data(), prev.size());
        Point r1 = pt_add(curve, received_point(CG_G1, prev.data()), pt_add(curve, a.c[0], a.c[1]));
        Point r2 = pt_add(curve, received_point(CG_G2, prev.data() + m1.size()), pt_add(curve, b.c[0], b.c[1]));
        return {r1, r2};
    }
};

}  // namespace cgh
