// Host-side mirror of the reference's prover interface for the co-groth16 path, written ONLY against the C ABI
// (include/cogroth16_hip.h) — i.e. it does what a Rust driver crate bound to that ABI would do:
//
//   HipDriver  (modes Plain / Rep3)  ~  PlainDriver `mpc-core/src/protocols/plain.rs`, Rep3Protocol `mpc-core/src/protocols/rep3.rs`
//       method names, argument meaning and party-id asymmetries follow the traits of `mpc-core/src/traits.rs`
//       (PrimeFieldMpcProtocol :43, EcMpcProtocol :472, PairingEcMpcProtocol :525, FFTProvider :535, MSMProvider :561);
//       vectors are DEVICE-resident share vectors (SoA, like Rep3PrimeFieldShareVec `rep3/fieldshare.rs:233-236`).
//   CoGroth16::prove                 ~  `co-circom/co-groth16/src/groth16.rs:113-326` (same call sequence, line refs inline)
//   Rep3Network / InProcNetwork      ~  `mpc-core/src/protocols/rep3/network.rs:30-64` and the in-process test network
//                                       `tests/src/rep3_network.rs` (three parties on three threads, one queue per edge)
//   read_zkey / read_wtns            ~  `circom-types/src/groth16/zkey.rs:139-316`, `binfile.rs:52-97`, `witness.rs:51-91`
//
// Everything O(n) runs on the GPU through the ABI; this file only sequences calls, moves the two `mul_vec` messages and the
// O(1) points between parties, and does O(1) scalar/point algebra through the ABI's host helpers.  No CPU fallback exists.
#include "cogroth16_hip.h"
#include "cogroth16_host.h"

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <chrono>
#include <fcntl.h>
#include <functional>
#include <memory>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace cgh {

typedef std::vector<uint8_t> Bytes;
struct Fr { uint64_t v[4]; };

[[noreturn]] static void die(const std::string& what) { throw std::runtime_error(what + ": " + cg_last_error()); }
#define CG(call) do { if ((call) != 0) die(#call); } while (0)

struct Curve {
    int id;
    size_t fq() const { return id == CG_BLS12_381 ? 48 : 32; }
    size_t aff(int g) const { return fq() * (g == CG_G1 ? 2 : 4); }
    size_t jac(int g) const { return fq() * (g == CG_G1 ? 3 : 6); }
};

// ---- O(1) algebra through the ABI's host helpers ----------------------------------------------------------------
static Fr fr_op(const Curve& c, int op, const Fr& a, const Fr* b = nullptr) { Fr r; CG(cg_fr_op(c.id, op, a.v, b ? b->v : nullptr, r.v)); return r; }
static Fr fr_add(const Curve& c, const Fr& a, const Fr& b) { return fr_op(c, 0, a, &b); }
static Fr fr_sub(const Curve& c, const Fr& a, const Fr& b) { return fr_op(c, 1, a, &b); }
static Fr fr_mul(const Curve& c, const Fr& a, const Fr& b) { return fr_op(c, 2, a, &b); }
static Fr fr_inv(const Curve& c, const Fr& a) { return fr_op(c, 3, a); }
static Fr fr_from_u64(const Curve& c, uint64_t x) { Fr raw = {{x, 0, 0, 0}}, r; CG(cg_fr_from_canonical(c.id, raw.v, r.v, 1)); return r; }
static bool fr_eq(const Fr& a, const Fr& b) { return memcmp(a.v, b.v, 32) == 0; }
static Fr fr_pow(const Curve& c, Fr base, const uint64_t* e, int nlimbs) {
    Fr r = fr_from_u64(c, 1);
    for (int i = nlimbs * 64 - 1; i >= 0; i--) { r = fr_mul(c, r, r); if ((e[i / 64] >> (i % 64)) & 1) r = fr_mul(c, r, base); }
    return r;
}

struct Point { Bytes b; int group; };   // Jacobian, Montgomery
static Point pt_from_affine(const Curve& c, int g, const uint8_t* aff) { Point p{Bytes(c.jac(g)), g}; CG(cg_point_from_affine(c.id, g, aff, p.b.data())); return p; }
static Point pt_inf(const Curve& c, int g) { Bytes z(c.aff(g), 0); return pt_from_affine(c, g, z.data()); }
static Point pt_add(const Curve& c, const Point& a, const Point& b) { Point r{Bytes(a.b.size()), a.group}; CG(cg_point_add(c.id, a.group, a.b.data(), b.b.data(), r.b.data())); return r; }
static Point pt_neg(const Curve& c, const Point& a) { Point r{Bytes(a.b.size()), a.group}; CG(cg_point_neg(c.id, a.group, a.b.data(), r.b.data())); return r; }
static Point pt_sub(const Curve& c, const Point& a, const Point& b) { return pt_add(c, a, pt_neg(c, b)); }
static Point pt_mul(const Curve& c, const Point& a, const Fr& k) { Point r{Bytes(a.b.size()), a.group}; CG(cg_point_scalar_mul(c.id, a.group, a.b.data(), k.v, r.b.data())); return r; }
static Bytes pt_to_affine(const Curve& c, const Point& a) { Bytes r(c.aff(a.group)); CG(cg_point_to_affine(c.id, a.group, a.b.data(), r.data())); return r; }
static Point pt_generator(const Curve& c, int g) { Point p{Bytes(c.jac(g)), g}; CG(cg_point_generator(c.id, g, p.b.data())); return p; }

// ---- file formats ----------------------------------------------------------------------------------------------------
struct Cursor {
    const uint8_t* p; size_t n, off = 0;
    void need(size_t k) const { if (off > n || k > n - off) throw std::runtime_error("unexpected end of section"); }   // no wrap-around for a 64-bit length read from the file
    uint32_t u32() { need(4); uint32_t x; memcpy(&x, p + off, 4); off += 4; return x; }
    uint64_t u64() { need(8); uint64_t x; memcpy(&x, p + off, 8); off += 8; return x; }
    void bytes(void* d, size_t k) { need(k); memcpy(d, p + off, k); off += k; }
};
static Bytes slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    Bytes b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); throw std::runtime_error("short read"); }
    fclose(f);
    return b;
}
// zkey -> device fast path: the file is mapped, the point sections are handed to cg_bases_register where they lie (one
// host->device copy, no intermediate buffers); only the small header points and the coefficient section are decoded on the host.
struct MappedFile {
    const uint8_t* p = nullptr; size_t n = 0;
    explicit MappedFile(const std::string& path) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st; if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("cannot stat " + path); }
        n = (size_t)st.st_size;
        if (n) { void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if (m == MAP_FAILED) { close(fd); throw std::runtime_error("cannot map " + path); } p = (const uint8_t*)m; }
        close(fd);
    }
    ~MappedFile() { if (p) munmap((void*)p, n); }
    MappedFile(const MappedFile&) = delete; MappedFile& operator=(const MappedFile&) = delete;
};
struct View {   // a section of the mapped file, with the read-only part of the std::vector interface the prover uses
    const uint8_t* p = nullptr; size_t n = 0;
    const uint8_t* data() const { return p; } size_t size() const { return n; }
    const uint8_t* begin() const { return p; } const uint8_t* end() const { return p + n; }
};

static const uint64_t MOD_R[2][4] = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
                                     {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};
static const uint64_t MOD_Q[2][6] = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull, 0, 0},
                                     {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull}};

struct ZKey {   // zkey.rs:48-71; points kept in the packed on-disk form (x||y Montgomery, (0,0) = infinity) the ABI accepts directly
    Curve curve;
    size_t n_vars = 0, n_public = 0, domain_size = 0, pow = 0, num_constraints = 0;
    Bytes alpha_g1, beta_g1, delta_g1, beta_g2, gamma_g2, delta_g2;
    std::shared_ptr<MappedFile> file;                     // keeps the views below alive
    View ic, a_query, b_g1_query, b_g2_query, l_query, h_query;
    std::vector<uint32_t> row_ptr[2], col[2];
    std::vector<Fr> coeff[2];
};

static ZKey read_zkey(int curve_id, const std::string& path, bool header_only = false) {   // header_only: sizes and counts, no matrix values
    Curve c{curve_id};
    auto mf = std::make_shared<MappedFile>(path);
    struct { const uint8_t* p; size_t n; const uint8_t* data() const { return p; } size_t size() const { return n; } } buf{mf->p, mf->n};
    Cursor cur{buf.data(), buf.size()};
    char magic[5] = {0}; cur.bytes(magic, 4);
    if (std::string(magic) != "zkey") throw std::runtime_error("not a zkey file");
    cur.u32();
    uint32_t ns = cur.u32();
    std::map<uint32_t, std::pair<size_t, size_t>> sec;
    for (uint32_t i = 0; i < ns; i++) { uint32_t id = cur.u32(); uint64_t len = cur.u64(); cur.need(len); sec[id] = {cur.off, (size_t)len}; cur.off += len; }
    auto section = [&](uint32_t id) { auto it = sec.find(id); if (it == sec.end()) throw std::runtime_error("missing zkey section"); return Cursor{buf.data() + it->second.first, it->second.second}; };
    ZKey z; z.curve = c; z.file = mf;
    {   // header, zkey.rs:258-316
        Cursor h = section(2);
        if (h.u32() != c.fq()) throw std::runtime_error("unexpected base field byte size");
        uint64_t q[6] = {0}; h.bytes(q, c.fq());
        if (memcmp(q, MOD_Q[curve_id], c.fq())) throw std::runtime_error("invalid base prime in header");
        if (h.u32() != 32) throw std::runtime_error("unexpected scalar field byte size");
        uint64_t r[4]; h.bytes(r, 32);
        if (memcmp(r, MOD_R[curve_id], 32)) throw std::runtime_error("invalid scalar prime in header");
        z.n_vars = h.u32(); z.n_public = h.u32(); z.domain_size = h.u32();
        if (!z.domain_size || (z.domain_size & (z.domain_size - 1))) throw std::runtime_error("domain size must be a power of two");
        if (z.n_vars <= z.n_public) throw std::runtime_error("invalid data: n_vars must exceed n_public");
        while (((size_t)1 << z.pow) < z.domain_size) z.pow++;
        auto g = [&](int grp) { Bytes b(c.aff(grp)); h.bytes(b.data(), b.size()); return b; };
        z.alpha_g1 = g(CG_G1); z.beta_g1 = g(CG_G1); z.beta_g2 = g(CG_G2); z.gamma_g2 = g(CG_G2); z.delta_g1 = g(CG_G1); z.delta_g2 = g(CG_G2);
    }
    auto pts = [&](uint32_t id, size_t n, int grp) { Cursor s = section(id); s.need(n * c.aff(grp)); return View{s.p, n * c.aff(grp)}; };
    z.ic = pts(3, z.n_public + 1, CG_G1); z.a_query = pts(5, z.n_vars, CG_G1); z.b_g1_query = pts(6, z.n_vars, CG_G1);
    z.b_g2_query = pts(7, z.n_vars, CG_G2); z.l_query = pts(8, z.n_vars - z.n_public - 1, CG_G1); z.h_query = pts(9, z.domain_size, CG_G1);
    {   // section 4, zkey.rs:184-204: records (u32 matrix, u32 row, u32 signal, 32 B value); value on disk = v*R^2, one Montgomery
        // reduction gives the Montgomery form of v (traits.rs:65-67).  Three passes over the mapped records: last row, row counts,
        // CSR fill with the raw values; then the values are reduced in place, in parallel slices.
        Cursor s = section(4);
        const uint32_t ncoef = s.u32();
        s.need((size_t)ncoef * 44);
        const uint8_t* rec = s.p + s.off;
        auto word = [&](size_t i, int k) { uint32_t v; memcpy(&v, rec + i * 44 + 4 * k, 4); return v; };
        uint32_t max_row = 0;
        for (size_t i = 0; i < ncoef; i++) { if (word(i, 0) > 1) throw std::runtime_error("bad matrix id"); max_row = std::max(max_row, word(i, 1)); }
        if (ncoef == 0 || max_row < z.n_public) throw std::runtime_error("invalid data: coefficient section has no rows beyond the public inputs");
        z.num_constraints = (size_t)max_row - z.n_public;
        for (int m = 0; m < 2; m++) z.row_ptr[m].assign(z.num_constraints + 1, 0);
        for (size_t i = 0; i < ncoef; i++) { const uint32_t row = word(i, 1); if (row < z.num_constraints) z.row_ptr[word(i, 0)][row + 1]++; }
        std::vector<uint32_t> fill[2];
        for (int m = 0; m < 2; m++) {
            for (size_t i = 0; i < z.num_constraints; i++) z.row_ptr[m][i + 1] += z.row_ptr[m][i];
            z.col[m].resize(z.row_ptr[m].back()); z.coeff[m].resize(z.col[m].size());
            fill[m].assign(z.row_ptr[m].begin(), z.row_ptr[m].end() - 1);
        }
        if (!header_only) {
            for (size_t i = 0; i < ncoef; i++) {
                const uint32_t m = word(i, 0), row = word(i, 1);
                if (row >= z.num_constraints) continue;
                const uint32_t k = fill[m][row]++;
                if (word(i, 2) >= z.n_vars) throw std::runtime_error("invalid data: matrix column index beyond n_vars");   // the device mat-vec indexes the witness with it
                z.col[m][k] = word(i, 2); memcpy(z.coeff[m][k].v, rec + i * 44 + 12, 32);
            }
            const int nthreads = (int)std::min<size_t>(8, std::max<size_t>(1, ncoef / 65536));
            std::vector<std::thread> th; std::vector<int> rc(2 * nthreads, 0);
            for (int m = 0; m < 2; m++) for (int t = 0; t < nthreads; t++) th.emplace_back([&, m, t] {
                const size_t n = z.coeff[m].size(), lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
                if (hi > lo) rc[m * nthreads + t] = cg_fr_to_canonical(c.id, z.coeff[m].data() + lo, z.coeff[m].data() + lo, hi - lo);
            });
            for (auto& x : th) x.join();
            for (int r : rc) if (r) die("cg_fr_to_canonical");
        }
    }
    return z;
}

static std::vector<Fr> read_wtns(int curve_id, const std::string& path) {   // witness.rs:51-91
    Bytes buf = slurp(path);
    Cursor c{buf.data(), buf.size()};
    char magic[5] = {0}; c.bytes(magic, 4);
    if (std::string(magic) != "wtns") throw std::runtime_error("not a wtns file");
    if (c.u32() > 2) throw std::runtime_error("wtns version not supported");
    if (c.u32() > 2) throw std::runtime_error("invalid section number");
    c.u32(); c.u64();
    if (c.u32() != 32) throw std::runtime_error("wrong scalar field");
    uint64_t mod[4]; c.bytes(mod, 32);
    if (memcmp(mod, MOD_R[curve_id], 32)) throw std::runtime_error("wrong scalar field");
    uint32_t n = c.u32();
    c.u32(); c.u64();
    std::vector<Fr> raw(n), out(n);
    c.bytes(raw.data(), (size_t)n * 32);
    CG(cg_fr_from_canonical(curve_id, raw.data(), out.data(), n));
    return out;
}

// co-circom-snarks/src/lib.rs:208-221 + groth16.rs:57-77
struct Domain { size_t m; int log_m; Fr omega, coset_g; };
// roots[i] = primitive 2^i-th root of unity derived from the smallest quadratic non-residue (co-circom-snarks/src/lib.rs:208-221)
struct SnarkjsRoots { Fr q; std::vector<Fr> roots; int two_adicity; };
static SnarkjsRoots snarkjs_roots(const Curve& c) {
    const uint64_t* r = MOD_R[c.id];
    uint64_t t[4] = {r[0] - 1, r[1], r[2], r[3]};
    int s = 0;
    while (!(t[0] & 1)) { for (int i = 0; i < 4; i++) t[i] = (t[i] >> 1) | (i < 3 ? t[i + 1] << 63 : 0); s++; }
    uint64_t half[4] = {r[0] - 1, r[1], r[2], r[3]};
    for (int i = 0; i < 4; i++) half[i] = (half[i] >> 1) | (i < 3 ? half[i + 1] << 63 : 0);
    const Fr one = fr_from_u64(c, 1), minus_one = fr_sub(c, fr_from_u64(c, 0), one);
    Fr q = one;
    while (!fr_eq(fr_pow(c, q, half, 4), minus_one)) q = fr_add(c, q, one);       // smallest quadratic non-residue
    std::vector<Fr> roots(s + 1);
    roots[0] = fr_pow(c, q, t, 4);
    for (int i = 1; i <= s; i++) roots[i] = fr_mul(c, roots[i - 1], roots[i - 1]);
    return SnarkjsRoots{q, std::vector<Fr>(roots.rbegin(), roots.rend()), s};
}
static Domain groth16_domain(const Curve& c, size_t pow, size_t num_constraints, size_t num_inputs) {
    const SnarkjsRoots rt = snarkjs_roots(c);
    Domain d; d.m = 1; d.log_m = 0;
    while (d.m < num_constraints + num_inputs) { d.m <<= 1; d.log_m++; }
    d.omega = rt.roots[pow];
    d.coset_g = rt.two_adicity == d.log_m ? fr_mul(c, rt.q, rt.q) : rt.roots[d.log_m + 1];
    return d;
}

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
struct ReplayNetwork : Rep3Network {
    int me; RecordedQueue* from_prev; RecordedQueue* from_next;
    ReplayNetwork(int i, RecordedQueue* p, RecordedQueue* q) : me(i), from_prev(p), from_next(q) {}
    int id() const override { return me; }
    void send_next(const void*, size_t) override {}
    void send_prev(const void*, size_t) override {}
    static void pop(RecordedQueue* q, void* d, size_t b) {
        if (q->q.empty() || q->q.front().n != b) throw std::runtime_error("replay: message sequence differs from the recorded run");
        const RecordedMsg& m = q->q.front();
        memcpy(d, m.pinned ? m.pinned : (const void*)m.small.data(), b); q->q.pop_front();
    }
    void recv_prev(void* d, size_t b) override { pop(from_prev, d, b); }
    void recv_next(void* d, size_t b) override { pop(from_next, d, b); }
    const void* recv_prev_pinned(size_t b) override {
        if (from_prev->q.empty() || from_prev->q.front().n != b) throw std::runtime_error("replay: message sequence differs from the recorded run");
        const void* p = from_prev->q.front().pinned;
        if (p) from_prev->q.pop_front();                 // the memory itself stays with the queue's owner list
        return p;
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

// ---- driver ------------------------------------------------------------------------------------------------------------
struct ShareVec { void* c[2] = {nullptr, nullptr}; size_t n = 0; };   // device; REP3 uses c[0] = a, c[1] = b; plain only c[0]
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
};
struct WorkerDevice { cg_ctx* ctx = nullptr; const DeviceZKey* dz = nullptr; };     // one further GPU of a party: a context on it + its table slices
struct MultiDevice { std::vector<WorkerDevice> workers; };

enum class Mode { Plain, Rep3, Shamir };

class HipDriver {
public:
    cg_ctx* ctx; Curve curve; Mode mode; Rep3Network* net;
    const Fr* rng1 = nullptr; const Fr* rng2 = nullptr; size_t rng_len = 0, cursor = 0;   // rngs.rs:25-46 streams (inputs)
    int k() const { return mode == Mode::Rep3 ? 2 : 1; }
    int party() const { return mode == Mode::Rep3 ? net->id() : -1; }

    HipDriver(cg_ctx* c, Curve cv, Mode m, Rep3Network* n) : ctx(c), curve(cv), mode(m), net(n) {}
    // CGH_TIMING=1: wall-clock marks of the host-side protocol steps (stderr)
    struct Marks {
        bool on; const char* what; std::chrono::steady_clock::time_point last; std::string line;
        Marks(const char* w, bool enabled) : on(enabled && getenv("CGH_TIMING")), what(w), last(std::chrono::steady_clock::now()) {}
        void mark(const char* name) {
            if (!on) return;
            const auto t = std::chrono::steady_clock::now(); char b[96];
            snprintf(b, sizeof b, " %s %.1f", name, std::chrono::duration<double, std::milli>(t - last).count()); line += b; last = t;
        }
        ~Marks() { if (on) fprintf(stderr, "%s [ms]:%s\n", what, line.c_str()); }
    };

    // ---- Shamir state (shamir.rs:196-246): threshold, Lagrange tables, buffered double sharings; randomness = stream rng1
    ShamirNet* snet = nullptr; int sh_t = 0;
    std::vector<Fr> open_lagrange_t, open_lagrange_2t, mul_lagrange_2t, sh_r_t, sh_r_2t;
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
    Fr next_rand() { if (cursor >= rng_len) throw std::runtime_error("randomness stream exhausted"); return rng1[cursor++]; }
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
    void vandermonde_mul(const std::vector<Fr>& in, std::vector<Fr>& out) {            // shamir.rs:904-921 (appends t + 1 values)
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
        for (int from = 0; from < np; from++) { if (from == me) got[from] = send[me]; else { got[from].resize(2 * amount); snet->recv(from, got[from].data(), 2 * amount * 32); } }
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
        if (cursor + draws > rng_len) throw std::runtime_error("randomness stream exhausted");
        Marks mk("shamir preprocess", me == 0);
        void* d_rnd = dalloc(draws * 32);
        CG(cg_dev_upload(ctx, d_rnd, rng1 + cursor, draws * 32)); cursor += draws;
        mk.mark("upload draws");
        const Fr one = fr_from_u64(curve, 1);
        std::vector<void*> d_got(np);
        for (int from = 0; from < np; from++) d_got[from] = dalloc(2 * amount * 32);
        void* d_pairs = dalloc(2 * amount * 32);
        std::vector<Fr> buf(2 * amount);
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
            if (to != me) { CG(cg_dev_download(ctx, buf.data(), d_pairs, 2 * amount * 32)); snet->send(to, buf.data(), 2 * amount * 32); }
        }
        mk.mark("share+send");
        for (int from = 0; from < np; from++) if (from != me) { snet->recv(from, buf.data(), 2 * amount * 32); CG(cg_dev_upload(ctx, d_got[from], buf.data(), 2 * amount * 32)); }
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
        std::vector<Fr> buf(len);
        mk.mark("add r_2t");
        if (me == 0) {                                                                 // KING_ID: interpolate at 0 from parties 0..2t, re-share with degree t
            CG(cg_vec_affine_dev(ctx, curve.id, local.c[0], local.c[0], len, mul_lagrange_2t[0].v, nullptr));   // acc = input * lagrange_0
            for (int other = 1; other <= 2 * sh_t; other++) {
                snet->recv(other, buf.data(), len * 32);
                CG(cg_dev_upload(ctx, tmp, buf.data(), len * 32));
                CG(cg_vec_affine_dev(ctx, curve.id, tmp, tmp, len, mul_lagrange_2t[other].v, nullptr));
                CG(cg_vec_add_dev(ctx, curve.id, local.c[0], local.c[0], tmp, len));
            }
            mk.mark("recv+interpolate");
            // ShamirCore::share per element: coefficients are drawn element by element (t per element)
            std::vector<std::vector<Fr>> coeff(sh_t, std::vector<Fr>(len));
            for (size_t k = 0; k < len; k++) for (int d = 0; d < sh_t; d++) coeff[d][k] = next_rand();
            std::vector<void*> d_coeff(sh_t);
            for (int d = 0; d < sh_t; d++) { d_coeff[d] = dalloc(len * 32); CG(cg_dev_upload(ctx, d_coeff[d], coeff[d].data(), len * 32)); }
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
                else { CG(cg_dev_download(ctx, buf.data(), share, len * 32)); snet->send(to, buf.data(), len * 32); }
            }
            CG(cg_dev_free(ctx, local.c[0])); local.c[0] = mine;
            for (void* p : d_coeff) CG(cg_dev_free(ctx, p));
            CG(cg_dev_free(ctx, share)); CG(cg_dev_free(ctx, term));
            mk.mark("reshare+send");
        } else {
            if (me <= 2 * sh_t) { CG(cg_dev_download(ctx, buf.data(), local.c[0], len * 32)); snet->send(0, buf.data(), len * 32); }   // only if my items are required
            mk.mark("download+send");
            snet->recv(0, buf.data(), len * 32);
            mk.mark("wait for king");
            CG(cg_dev_upload(ctx, local.c[0], buf.data(), len * 32));
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
            for (int other = 1; other <= 2 * sh_t; other++) { Fr r; snet->recv(other, r.v, 32); acc = fr_add(curve, acc, fr_mul(curve, r, mul_lagrange_2t[other])); }
            auto shares = shamir_share(acc, sh_t);
            for (int to = 0; to < np; to++) { if (to == me) my_share = shares[to]; else snet->send(to, shares[to].v, 32); }
        } else {
            if (me <= 2 * sh_t) snet->send(0, input.v, 32);
            snet->recv(0, my_share.v, 32);
        }
        return fr_sub(curve, my_share, pr.first);
    }
    Point degree_reduce_point(Point input) {                                           // shamir.rs:386-436; C::rand stand-in: G * next_rand()
        const int np = snet->num_parties(), me = snet->id();
        const int g = input.group;
        auto pr = get_pair();
        const Point gen = pt_generator(curve, g);
        input = pt_add(curve, input, pt_mul(curve, gen, pr.second));
        Point my_share = pt_inf(curve, g);
        const size_t psz = curve.aff(g);
        if (me == 0) {
            Point acc = pt_mul(curve, input, mul_lagrange_2t[0]);
            for (int other = 1; other <= 2 * sh_t; other++) { Bytes a(psz); snet->recv(other, a.data(), psz); acc = pt_add(curve, acc, pt_mul(curve, pt_from_affine(curve, g, a.data()), mul_lagrange_2t[other])); }
            std::vector<Point> coeffs; for (int d = 0; d < sh_t; d++) coeffs.push_back(pt_mul(curve, gen, next_rand()));
            for (int to = 0; to < np; to++) {
                Point sh = acc; const Fr x = fr_from_u64(curve, (uint64_t)to + 1); Fr xp = x;
                for (const Point& cf : coeffs) { sh = pt_add(curve, sh, pt_mul(curve, cf, xp)); xp = fr_mul(curve, xp, x); }
                if (to == me) my_share = sh; else { Bytes a = pt_to_affine(curve, sh); snet->send(to, a.data(), a.size()); }
            }
        } else {
            if (me <= 2 * sh_t) { Bytes a = pt_to_affine(curve, input); snet->send(0, a.data(), a.size()); }
            Bytes a(psz); snet->recv(0, a.data(), psz); my_share = pt_from_affine(curve, g, a.data());
        }
        return pt_sub(curve, my_share, pt_mul(curve, gen, pr.first));
    }
    // broadcast_next(t + 1) + reconstruct_point (network.rs:233-266, shamir.rs:778-782)
    Point shamir_open_point(const Point& mine) {
        const int np = snet->num_parties(), me = snet->id();
        Bytes a = pt_to_affine(curve, mine);
        for (int sft = 1; sft <= sh_t; sft++) snet->send((me + sft) % np, a.data(), a.size());
        Point res = pt_mul(curve, mine, open_lagrange_t[0]);
        for (int r = 1; r <= sh_t; r++) { Bytes b(a.size()); snet->recv((me + np - r) % np, b.data(), b.size()); res = pt_add(curve, res, pt_mul(curve, pt_from_affine(curve, mine.group, b.data()), open_lagrange_t[r])); }
        return res;
    }

    void* dalloc(size_t bytes) { void* p; CG(cg_dev_alloc(ctx, bytes, &p)); return p; }
    ShareVec alloc_vec(size_t n) { ShareVec v; v.n = n; for (int j = 0; j < k(); j++) { v.c[j] = dalloc(n * 32); CG(cg_dev_memset_zero(ctx, v.c[j], n * 32)); } return v; }
    void free_vec(ShareVec& v) { for (int j = 0; j < 2; j++) if (v.c[j]) { CG(cg_dev_free(ctx, v.c[j])); v.c[j] = nullptr; } }
    ShareVec upload_vec(const Fr* a, const Fr* b, size_t n) {
        ShareVec v; v.n = n;
        if (n >= XCHG_ASYNC_MIN && cg_host_is_pinned(a) && (!b || k() < 2 || cg_host_is_pinned(b))) {   // page-locked shares: asynchronous DMA, the stream waits
            v.c[0] = dalloc(n * 32);
            int32_t tk = upload_staged(v.c[0], a, n);
            if (b && k() == 2) { v.c[1] = dalloc(n * 32); tk = upload_staged(v.c[1], b, n); }
            if (tk >= 0) CG(cg_copy_fence(ctx, tk));
            if (aux) CG(cg_copy_wait(ctx, tk));                                        // the second context reads the shares too
            return v;
        }
        v.c[0] = dalloc(n * 32); CG(cg_dev_upload(ctx, v.c[0], a, n * 32));
        if (k() == 2) { v.c[1] = dalloc(n * 32); CG(cg_dev_upload(ctx, v.c[1], b, n * 32)); }
        return v;
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
    void release_rings() { for (PinRing* r : {&ring_out, &ring_in}) if (r->base) { cg_ctx_sync(ctx); cg_host_free(r->base); r->base = nullptr; r->chunk = 0; } }
    std::vector<void*> deferred;                                                        // device buffers freed at the next quiet point
    void defer_free(void* p) { if (p) deferred.push_back(p); }
    void free_deferred() { for (void* p : deferred) CG(cg_dev_free(ctx, p)); deferred.clear(); }
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
    const size_t XCHG_ASYNC_MIN = getenv("CGH_XCHG_ASYNC_MIN") ? (size_t)atoll(getenv("CGH_XCHG_ASYNC_MIN")) : (size_t)1 << 19;
    // masks of the coming mul_vec calls, uploaded ahead of time (only from page-locked randomness streams, where the copy is a plain
    // asynchronous DMA): the product kernel then never waits for PCIe
    struct MaskSet { void* m1; void* m2; int32_t tk; size_t n, at; };
    std::deque<MaskSet> prefetched;
    void prefetch_masks(int count, size_t n) {
        if (mode != Mode::Rep3 || n < XCHG_ASYNC_MIN || !rng1 || !rng2) return;
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
    struct PendingMul { ShareVec out; bool exchange = false; std::deque<Down> down; size_t issued = 0; };
    // start streaming chunks of the local product to the host, as many as the ring has room for
    void issue_downloads(PendingMul& pm, size_t upto) {
        const size_t n = pm.out.n, ch = xchg_chunk(n), nch = (n + ch - 1) / ch;
        while (pm.issued < nch && pm.issued < upto) {
            const size_t off = pm.issued * ch, len = std::min(ch, n - off);
            Down d; d.slot = ring_slot(ring_out, ch);
            CG(cg_dev_download_begin(ctx, d.slot, (const uint8_t*)pm.out.c[0] + off * 32, len * 32, &d.tk));
            ring_out.busy[ring_out.last] = d.tk;
            pm.down.push_back(d); pm.issued++;
        }
    }
    PendingMul mul_vec_begin(const ShareVec& a, const ShareVec& b) {
        PendingMul pm; ShareVec& out = pm.out; out.n = a.n;
        out.c[0] = dalloc(a.n * 32);
        if (mode != Mode::Rep3) CG(cg_vec_mul_dev(ctx, curve.id, out.c[0], a.c[0], b.c[0], a.n));
        if (mode == Mode::Plain) return pm;
        if (mode == Mode::Shamir) { out = degree_reduce_vec(out); return pm; }         // shamir.rs:609-623
        if (cursor + a.n > rng_len) throw std::runtime_error("randomness stream exhausted");
        void* m1 = nullptr; void* m2 = nullptr;
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
        CG(cg_vec_rep3_mul_local_dev(ctx, curve.id, out.c[0], a.c[0], a.c[1], b.c[0], b.c[1], m1, a.n));
        defer_free(m1); defer_free(m2);
        out.c[1] = dalloc(a.n * 32);
        pm.exchange = true;
        if (a.n >= XCHG_ASYNC_MIN) issue_downloads(pm, XCHG_SLOTS - 1);                // ordered right behind the product, ahead of whatever the caller enqueues next
        return pm;
    }
    ShareVec mul_vec_finish(PendingMul& pm) {
        if (!pm.exchange) return pm.out;
        ShareVec& out = pm.out;
        pm.exchange = false;
        if (out.n < XCHG_ASYNC_MIN) {                                                  // rep3.rs:661-669 as one message
            std::vector<Fr> local(out.n), recv(out.n);
            CG(cg_dev_download(ctx, local.data(), out.c[0], out.n * 32));
            net->send_next(local.data(), out.n * 32);
            net->recv_prev(recv.data(), out.n * 32);
            CG(cg_dev_upload(ctx, out.c[1], recv.data(), out.n * 32));
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
        pm.exchange = false;
        return out;
    }
    ShareVec mul_vec(const ShareVec& a, const ShareVec& b) { PendingMul pm = mul_vec_begin(a, b); ShareVec r = mul_vec_finish(pm); free_deferred(); return r; }
    // before the context goes away (idempotent; also run by the destructor when a party dies with an exception)
    void shutdown() {
        for (void* p : deferred) cg_dev_free(ctx, p);
        deferred.clear();
        for (auto& ms : prefetched) { cg_dev_free(ctx, ms.m1); cg_dev_free(ctx, ms.m2); }
        prefetched.clear();
        release_rings(); release_pre();
        if (aux) { if (owns_aux) cg_ctx_destroy(aux); aux = nullptr; }
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
        for (int r = 1; r < num; r++) { snet->recv((me + np - r) % np, got.data(), n * 32); for (size_t i = 0; i < n; i++) out[i] = fr_add(curve, out[i], fr_mul(curve, got[i], lagrange[r])); }
        return out;
    }
    ShareVec rand_vec(size_t n) {
        if (mode == Mode::Shamir) { std::vector<Fr> r(n); for (size_t i = 0; i < n; i++) r[i] = get_pair().first; return upload_vec(r.data(), nullptr, n); }   // shamir.rs:570-573
        if (mode != Mode::Rep3) throw std::runtime_error("rand_vec: REP3 / Shamir only");
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
                snet->recv((me + np - r) % np, buf.data(), n * 32);
                void* d = dalloc(n * 32); CG(cg_dev_upload(ctx, d, buf.data(), n * 32));
                got.push_back(d); terms.push_back({d, 0, 1, open_lagrange_2t[r]});
            }
            lincomb(out, 0, 1, n, terms);
            for (void* d : got) CG(cg_dev_free(ctx, d));
            return out;
        }
        if (cursor + n > rng_len) throw std::runtime_error("randomness stream exhausted");
        void* m1 = dalloc(n * 32); void* m2 = dalloc(n * 32);
        CG(cg_dev_upload(ctx, m1, rng1 + cursor, n * 32)); CG(cg_dev_upload(ctx, m2, rng2 + cursor, n * 32)); cursor += n;
        CG(cg_vec_sub_dev(ctx, curve.id, m1, m1, m2, n));
        CG(cg_vec_rep3_mul_local_dev(ctx, curve.id, out, a.c[0], a.c[1], b.c[0], b.c[1], m1, n));
        std::vector<Fr> mine(n), p(n), q(n);
        CG(cg_dev_download(ctx, mine.data(), out, n * 32));
        net->send_next(mine.data(), n * 32); net->send_prev(mine.data(), n * 32);
        net->recv_prev(p.data(), n * 32); net->recv_next(q.data(), n * 32);
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
        net->send_next(bs.data(), bs.size() * 32); net->recv_prev(cs.data(), cs.size() * 32);
        for (size_t i = 0; i < a.size(); i++) out[i] = fr_add(curve, fr_add(curve, a[i].c[0], a[i].c[1]), cs[i]);
        return out;
    }

    // FFTProvider (traits.rs:535-558): both share components in one launch
    void fft_in_place(ShareVec& v, const Fr& group_gen) { CG(cg_ntt_dev(ctx, curve.id, v.c, k(), v.n, group_gen.v, 0, nullptr)); }
    void ifft_in_place(ShareVec& v, const Fr& group_gen) { CG(cg_ntt_dev(ctx, curve.id, v.c, k(), v.n, group_gen.v, 1, nullptr)); }
    void distribute_powers_and_mul_by_const(ShareVec& v, const Fr& g, const Fr& c) { for (int j = 0; j < k(); j++) CG(cg_vec_distribute_powers_dev(ctx, curve.id, v.c[j], v.n, g.v, c.v)); }
    // fused ifft_in_place + distribute_powers_and_mul_by_const(g, 1): one HBM round trip less per vector
    void ifft_coset_in_place(ShareVec& v, const Fr& group_gen, const Fr& g) { CG(cg_ntt_dev(ctx, curve.id, v.c, k(), v.n, group_gen.v, 1, g.v)); }
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
        cg_ctx* on = nullptr; std::vector<int32_t> tickets; std::vector<int> groups;
        struct Part { cg_ctx* on; std::vector<int32_t> tickets; void* sc[2]; };     // the same MSMs over the slices held by further GPUs
        std::vector<Part> parts;
    };
    // Several GPUs: every MSM range is cut into one contiguous slice per device (the primary context's device holds slice 0); the scalar
    // slices travel device to device (cg_dev_copy_peer, xGMI), each device runs the bucket method on its slice, and the partial sums
    // — one Jacobian point per table, component and device — are added on the host (RCCL has no EC-add reduction, and a few hundred
    // bytes per proof need no collective).  MSMProvider::msm_public_points (rep3.rs:934-947) is linear in the (scalar, point) pairs.
    const MultiDevice* md = nullptr;
    PendingMsm msm_begin_sharded(const DeviceZKey& dz, bool aux_tables, const ShareVec& s) {
        const size_t lo = aux_tables ? dz.aux_lo : dz.h_lo, n = aux_tables ? dz.aux_n : dz.h_n;
        ShareVec mine; mine.n = n; for (int j = 0; j < k(); j++) mine.c[j] = (char*)s.c[j] + lo * 32;
        PendingMsm p = aux_tables ? msm_begin_multi({dz.l, dz.a, dz.b1, dz.b2}, {0, 0, 0, 0}, {CG_G1, CG_G1, CG_G1, CG_G2}, n, mine, true)
                                  : msm_begin_multi({dz.h}, {0}, {CG_G1}, n, mine, false);
        if (!md) return p;
        static const bool primary_only = getenv("CGH_EMULATE_PRIMARY_ONLY") != nullptr;    // planning knob: time the primary device's share of an
        if (primary_only) return p;                                                       // N-device proof on one GPU (the proof is then wrong)
        for (const WorkerDevice& w : md->workers) {
            const DeviceZKey& wz = *w.dz;
            const size_t wlo = aux_tables ? wz.aux_lo : wz.h_lo, wn = aux_tables ? wz.aux_n : wz.h_n;
            PendingMsm::Part part{w.ctx, {}, {nullptr, nullptr}};
            for (int j = 0; j < k(); j++) {
                CG(cg_dev_alloc(w.ctx, std::max<size_t>(wn * 32, 32), &part.sc[j]));
                CG(cg_dev_copy_peer(w.ctx, part.sc[j], ctx, (const char*)s.c[j] + wlo * 32, wn * 32));
            }
            std::vector<const cg_bases*> tabs = aux_tables ? std::vector<const cg_bases*>{wz.l, wz.a, wz.b1, wz.b2} : std::vector<const cg_bases*>{wz.h};
            std::vector<size_t> offs(tabs.size(), 0);
            part.tickets.resize(tabs.size());
            const void* sc[2] = {part.sc[0], part.sc[1]};
            CG(cg_msm_dev_begin_multi(w.ctx, (int32_t)tabs.size(), tabs.data(), offs.data(), wn, sc, k(), part.tickets.data()));
            p.parts.push_back(part);
        }
        return p;
    }
    void msm_release(PendingMsm& p) {      // after the last msm_finish: the slices' scalar copies
        for (auto& part : p.parts) for (int j = 0; j < 2; j++) if (part.sc[j]) { cg_dev_free(part.on, part.sc[j]); part.sc[j] = nullptr; }
        p.parts.clear();
    }
    PendingMsm msm_begin_multi(const std::vector<const cg_bases*>& tables, const std::vector<size_t>& offsets, const std::vector<int>& groups, size_t n, const ShareVec& s, bool on_aux) {
        PendingMsm p; p.on = on_aux && aux ? aux : ctx; p.groups = groups; p.tickets.resize(tables.size());
        // REP3 at sizes where the exchanges are asynchronous: these MSMs run beside the witness map's dependency chain (product -> down ->
        // peer -> up, twice) on the other context; shorter-lived workgroups let the chain's kernels onto the chip sooner (2^22: one party
        // alone 107 -> 97 ms)
        static const uint32_t bulk_chunk = getenv("CGH_BULK_CHUNK") ? (uint32_t)atoi(getenv("CGH_BULK_CHUNK")) : 64u;   // tuning knob
        static const uint32_t plain_chunk = getenv("CGH_PLAIN_CHUNK") ? (uint32_t)atoi(getenv("CGH_PLAIN_CHUNK")) : 0u;  // tuning knob
        if (p.on != ctx) CG(cg_msm_set_chunk(p.on, mode == Mode::Rep3 && n >= XCHG_ASYNC_MIN ? bulk_chunk : plain_chunk));
        const void* sc[2] = {s.c[0], s.c[1]};
        if (p.on != ctx) CG(cg_ctx_sync(ctx));                                          // the scalars were produced on this driver's stream
        CG(cg_msm_dev_begin_multi(p.on, (int32_t)tables.size(), tables.data(), offsets.data(), n, sc, k(), p.tickets.data()));
        return p;
    }
    PointShare msm_finish(PendingMsm& p, size_t i) {
        const int group = p.groups[i];
        Bytes out(curve.jac(group) * k());
        CG(cg_msm_end(p.on, p.tickets[i], out.data()));
        PointShare r;
        for (int j = 0; j < k(); j++) r.c[j] = Point{Bytes(out.begin() + j * curve.jac(group), out.begin() + (j + 1) * curve.jac(group)), group};
        for (auto& part : p.parts) {                                                    // slices on further GPUs: fold the partial sums
            CG(cg_msm_end(part.on, part.tickets[i], out.data()));
            for (int j = 0; j < k(); j++) r.c[j] = pt_add(curve, r.c[j], Point{Bytes(out.begin() + j * curve.jac(group), out.begin() + (j + 1) * curve.jac(group)), group});
        }
        if (k() == 1) r.c[1] = pt_inf(curve, group);
        return r;
    }
    // rand (rep3.rs:595-598; plain: supplied by the caller)
    FieldShare rand() {
        if (mode == Mode::Shamir) { FieldShare f; f.c[0] = get_pair().first; f.c[1] = f.c[0]; return f; }   // shamir.rs:570-573
        FieldShare f; f.c[0] = draw(rng1); f.c[1] = draw(rng2); cursor++; return f;
    }
    // mul (rep3.rs:503-511 / plain a*b)
    FieldShare mul(const FieldShare& a, const FieldShare& b) {
        FieldShare r;
        if (mode == Mode::Plain) { r.c[0] = fr_mul(curve, a.c[0], b.c[0]); r.c[1] = r.c[0]; return r; }
        if (mode == Mode::Shamir) { r.c[0] = degree_reduce(fr_mul(curve, a.c[0], b.c[0])); r.c[1] = r.c[0]; return r; }   // shamir.rs:481-488
        Fr local = fr_add(curve, fr_add(curve, fr_mul(curve, a.c[0], b.c[0]), fr_mul(curve, a.c[0], b.c[1])), fr_mul(curve, a.c[1], b.c[0]));
        local = fr_add(curve, local, fr_sub(curve, draw(rng1), draw(rng2))); cursor++;
        net->send_next(local.v, 32);
        Fr prev; net->recv_prev(prev.v, 32);
        r.c[0] = local; r.c[1] = prev;
        return r;
    }
    PointShare scalar_mul_public_point(const Point& p, const FieldShare& s) {   // rep3.rs:820-825
        PointShare r; for (int j = 0; j < 2; j++) r.c[j] = j < k() ? pt_mul(curve, p, s.c[j]) : pt_inf(curve, p.group); return r;
    }
    PointShare scalar_mul(const PointShare& a, const FieldShare& b) {           // rep3.rs:835-847, pointshare.rs:117-124
        PointShare r;
        if (mode == Mode::Plain) { r.c[0] = pt_mul(curve, a.c[0], b.c[0]); r.c[1] = pt_inf(curve, a.c[0].group); return r; }
        if (mode == Mode::Shamir) { r.c[0] = degree_reduce_point(pt_mul(curve, a.c[0], b.c[0])); r.c[1] = pt_inf(curve, a.c[0].group); return r; }   // shamir.rs:769-776
        Point local = pt_add(curve, pt_add(curve, pt_mul(curve, a.c[0], b.c[0]), pt_mul(curve, a.c[1], b.c[0])), pt_mul(curve, a.c[0], b.c[1]));
        const Point gen = pt_generator(curve, a.c[0].group);       // masking_ec_element: G*rand(rng1) - G*rand(rng2)
        local = pt_add(curve, local, pt_sub(curve, pt_mul(curve, gen, draw(rng1)), pt_mul(curve, gen, draw(rng2)))); cursor++;
        Bytes aff = pt_to_affine(curve, local);                    // points cross the wire in affine form (ark-serialize)
        net->send_next(aff.data(), aff.size());
        Bytes prev(aff.size()); net->recv_prev(prev.data(), prev.size());
        r.c[0] = local; r.c[1] = pt_from_affine(curve, local.group, prev.data());
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
        return pt_add(curve, pt_add(curve, a.c[0], a.c[1]), pt_from_affine(curve, a.c[0].group, prev.data()));
    }
    std::pair<Point, Point> open_two_points(const PointShare& a, const PointShare& b) {   // rep3.rs:865-877
        if (mode == Mode::Plain) return {a.c[0], b.c[0]};
        if (mode == Mode::Shamir) { Point p1 = shamir_open_point(a.c[0]); return {p1, shamir_open_point(b.c[0])}; }   // shamir.rs:808-824 (one message per point here)
        Bytes m1 = pt_to_affine(curve, a.c[1]), m2 = pt_to_affine(curve, b.c[1]);
        Bytes msg(m1); msg.insert(msg.end(), m2.begin(), m2.end());
        net->send_next(msg.data(), msg.size());
        Bytes prev(msg.size()); net->recv_prev(prev.data(), prev.size());
        Point r1 = pt_add(curve, pt_from_affine(curve, CG_G1, prev.data()), pt_add(curve, a.c[0], a.c[1]));
        Point r2 = pt_add(curve, pt_from_affine(curve, CG_G2, prev.data() + m1.size()), pt_add(curve, b.c[0], b.c[1]));
        return {r1, r2};
    }
};

// ---- prover --------------------------------------------------------------------------------------------------------------
struct VecGuard {   // device share vector released when the entry point leaves, however it leaves
    HipDriver& d; ShareVec v;
    explicit VecGuard(HipDriver& drv) : d(drv) {}
    VecGuard(HipDriver& drv, ShareVec x) : d(drv), v(x) {}
    ~VecGuard() { try { d.free_vec(v); } catch (...) {} }
    VecGuard(const VecGuard&) = delete; VecGuard& operator=(const VecGuard&) = delete;
};
struct Proof { Bytes a, b, c; };   // packed affine, (0,0) = infinity  (Groth16Proof, groth16/proof.rs:8-29)

class CoGroth16 {
public:
    HipDriver& driver;
    explicit CoGroth16(HipDriver& d) : driver(d) {}

    // groth16.rs:141-204
    ShareVec witness_map_from_matrices(const DeviceZKey& dz, const std::vector<Fr>& public_inputs, const ShareVec& private_witness) {
        const ZKey& z = *dz.z;
        const size_t num_inputs = z.n_public + 1, num_constraints = z.num_constraints;
        const Domain dom = groth16_domain(driver.curve, z.pow, num_constraints, num_inputs);          // :150-153
        HipDriver::Marks mk("witness_map party 0", driver.party() <= 0);
        ShareVec a = driver.evaluate_constraints(dz.mat[0], dz.pub_dev, (uint32_t)num_inputs, private_witness, dom.m);   // :156-166
        ShareVec b = driver.evaluate_constraints(dz.mat[1], dz.pub_dev, (uint32_t)num_inputs, private_witness, dom.m);
        driver.clone_public_into(a, num_constraints, public_inputs, dz.pub_dev);                       // :168-171
        // The two mul_vec exchanges (:174, :190) run under the transforms that do not depend on them: the local product is started,
        // the independent NTTs are enqueued, then the party-to-party exchange proceeds while the GPU works (values as in the reference).
        if (driver.prefetched.empty()) driver.prefetch_masks(2, dom.m);                                // :174 and :190 draw next to each other
        mk.mark("spmv enqueue");
        auto c_pending = driver.mul_vec_begin(a, b);                                                   // :174
        mk.mark("mul_vec_begin");
        driver.ifft_coset_in_place(a, dom.omega, dom.coset_g);                                         // :175,177-181
        driver.ifft_coset_in_place(b, dom.omega, dom.coset_g);                                         // :176,182-186
        driver.fft_in_place(a, dom.omega); driver.fft_in_place(b, dom.omega);                          // :187-188
        mk.mark("ntt enqueue");
        ShareVec c = driver.mul_vec_finish(c_pending);
        mk.mark("mul_vec_finish");
        auto ab_pending = driver.mul_vec_begin(a, b);                                                  // :190
        mk.mark("mul_vec_begin");
        driver.ifft_coset_in_place(c, dom.omega, dom.coset_g);                                         // :194-199
        driver.fft_in_place(c, dom.omega);                                                             // :200
        mk.mark("ntt enqueue");
        ShareVec ab = driver.mul_vec_finish(ab_pending);
        mk.mark("mul_vec_finish");
        driver.sub_assign_vec(ab, c);                                                                  // :202
        driver.free_vec(a); driver.free_vec(b); driver.free_vec(c); driver.free_deferred();
        mk.mark("free (sync)");
        return ab;
    }

    // groth16.rs:206-235
    // priv_acc = msm_public_points(&query[1 + pub_len..], aux_assignment) (:221), started before the witness map (see prove)
    PointShare calculate_coeff(PointShare initial, const View& query_host, int group, const Bytes& vk_param,
                               const std::vector<Fr>& input_assignment, const PointShare& priv_acc) {
        const Curve& c = driver.curve;
        const size_t pub_len = input_assignment.size(), rec = c.aff(group);
        Point pub_acc = pt_inf(c, group);                                                              // :220 (tiny, plain scalars)
        for (size_t i = 0; i < pub_len; i++) pub_acc = pt_add(c, pub_acc, pt_mul(c, pt_from_affine(c, group, query_host.data() + (1 + i) * rec), input_assignment[i]));
        PointShare res = initial;
        driver.add_assign_points_public(res, pt_from_affine(c, group, query_host.data()));             // :227
        driver.add_assign_points_public(res, pt_from_affine(c, group, vk_param.data()));               // :228
        driver.add_assign_points_public(res, pub_acc);                                                 // :229
        driver.add_assign_points(res, priv_acc);                                                       // :230
        return res;
    }

    // groth16.rs:113-139 + :237-326
    Proof prove(const DeviceZKey& dz, const std::vector<Fr>& public_inputs, const ShareVec& private_witness, const FieldShare* rs_plain, ShareVec* h_out = nullptr) {
        const ZKey& z = *dz.z; const Curve& c = driver.curve;
        HipDriver::Marks mk("prove party 0", driver.party() <= 0);
        std::vector<Fr> input_assignment(public_inputs.begin() + 1, public_inputs.end());
        const size_t first_aux = 1 + input_assignment.size();
        // l (:251), a (:267 -> :221), b1 (:284), b2 (:298): one call, one scalar schedule, on the second context
        auto aux_msm = dz.sliced ? driver.msm_begin_sharded(dz, true, private_witness)
                                 : driver.msm_begin_multi({dz.l, dz.a, dz.b1, dz.b2}, {0, first_aux, first_aux, first_aux}, {CG_G1, CG_G1, CG_G1, CG_G2}, private_witness.n, private_witness, true);
        mk.mark("aux msm enqueued");
        // the masks of the witness map's two mul_vec calls (:174, :190) start their way to the device now: behind the witness shares and the
        // few small synchronous uploads of the MSM set-up (the copy engine serves its requests in order), ahead of everything else
        driver.prefetch_masks(2, groth16_domain(c, z.pow, z.num_constraints, public_inputs.size()).m);
        mk.mark("mask uploads enqueued");
        ShareVec h = witness_map_from_matrices(dz, public_inputs, private_witness);
        mk.mark("witness map");
        auto h_msm = dz.sliced ? driver.msm_begin_sharded(dz, false, h) : driver.msm_begin_multi({dz.h}, {0}, {CG_G1}, h.n, h, false);   // :248
        FieldShare r = rs_plain ? rs_plain[0] : driver.rand();                                         // :134-135
        FieldShare s = rs_plain ? rs_plain[1] : driver.rand();
        PointShare h_acc = driver.msm_finish(h_msm, 0);
        PointShare l_aux_acc = driver.msm_finish(aux_msm, 0);
        mk.mark("msm h + l");
        const Point delta_g1 = pt_from_affine(c, CG_G1, z.delta_g1.data());
        FieldShare rs = driver.mul(r, s);                                                              // :258
        PointShare r_s_delta_g1 = driver.scalar_mul_public_point(delta_g1, rs);                        // :259
        PointShare r_g1 = driver.scalar_mul_public_point(delta_g1, r);                                 // :265
        PointShare g_a = calculate_coeff(r_g1, z.a_query, CG_G1, z.alpha_g1, input_assignment, driver.msm_finish(aux_msm, 1));   // :267
        Point g_a_opened = driver.open_point(g_a);                                                     // :276
        PointShare s_g_a = driver.scalar_mul_public_point(g_a_opened, s);                              // :277
        PointShare s_g1 = driver.scalar_mul_public_point(delta_g1, s);                                 // :283
        PointShare g1_b = calculate_coeff(s_g1, z.b_g1_query, CG_G1, z.beta_g1, input_assignment, driver.msm_finish(aux_msm, 2));   // :284
        PointShare r_g1_b = driver.scalar_mul(g1_b, r);                                                // :291
        const Point delta_g2 = pt_from_affine(c, CG_G2, z.delta_g2.data());
        PointShare s_g2 = driver.scalar_mul_public_point(delta_g2, s);                                 // :297
        PointShare g2_b = calculate_coeff(s_g2, z.b_g2_query, CG_G2, z.beta_g2, input_assignment, driver.msm_finish(aux_msm, 3));   // :298
        PointShare g_c = s_g_a;                                                                        // :308-312
        driver.add_assign_points(g_c, r_g1_b);
        driver.sub_assign_points(g_c, r_s_delta_g1);
        driver.add_assign_points(g_c, l_aux_acc);
        driver.add_assign_points(g_c, h_acc);
        mk.mark("msm a, b1, b2 + scalar steps");
        auto opened = driver.open_two_points(g_c, g2_b);                                               // :316
        mk.mark("open");
        driver.msm_release(aux_msm); driver.msm_release(h_msm);
        if (h_out) *h_out = h; else driver.free_vec(h);
        return Proof{pt_to_affine(c, g_a_opened), pt_to_affine(c, opened.second), pt_to_affine(c, opened.first)};   // :319-325
    }
};

// The reference's parser validates every point while decoding (circom-types/src/traits.rs:107-155: is_on_curve, then
// is_in_correct_subgroup_assuming_on_curve; failure = SerializationError::InvalidData).  Here the packed sections go to the device
// as they are and the same two predicates run there, one pass per table.
static void validate_bases(cg_ctx* ctx, const cg_bases* b, const char* name) {
    uint64_t bad = 0, first = 0;
    CG(cg_bases_check_on_curve(ctx, b, &bad, &first));
    if (bad) throw std::runtime_error(std::string("invalid data: ") + name + "[" + std::to_string(first) + "] is not on the curve (" + std::to_string(bad) + " bad points)");
    CG(cg_bases_check_subgroup(ctx, b, &bad, &first));
    if (bad) throw std::runtime_error(std::string("invalid data: ") + name + "[" + std::to_string(first) + "] is not in the correct subgroup (" + std::to_string(bad) + " bad points)");
}

// The reference validates every zkey point while parsing (traits.rs:116-123, 147-153), so the prove entry points and
// cgh_session_open do too, by default.  Opt-out for callers that validated the file before (cgh_zkey_validate): the environment
// variable CGH_SKIP_ZKEY_VALIDATION or cgh_set_zkey_validation(0).
static std::atomic<int> g_validate_zkey{-1};
static bool validate_by_default() {
    int v = g_validate_zkey.load();
    if (v < 0) { v = getenv("CGH_SKIP_ZKEY_VALIDATION") ? 0 : 1; g_validate_zkey.store(v); }
    return v != 0;
}
struct DeviceZKeyGuard;
static void release_zkey(cg_ctx* ctx, DeviceZKey& d);
// slice `rank` of `world` of a range of n items (sizes differ by at most one)
static std::pair<size_t, size_t> slice_of(size_t n, int rank, int world) {
    const size_t base = n / world, rem = n % world, lo = (size_t)rank * base + std::min<size_t>((size_t)rank, rem);
    return {lo, lo + base + ((size_t)rank < rem ? 1 : 0)};
}
// rank/world: this device's share of a party's GPUs (world == 1: the whole zkey).  Rank 0 also holds the constraint matrices (the
// witness map runs there); every rank holds slice `rank` of the five queries.
static DeviceZKey upload_zkey(cg_ctx* ctx, const ZKey& z, const std::vector<Fr>& public_inputs, int validate_flag = -1, int rank = 0, int world = 1) {
    const bool validate = validate_flag < 0 ? validate_by_default() : validate_flag != 0;
    DeviceZKey d; d.z = &z; d.owner = ctx;
    struct Undo { cg_ctx* c; DeviceZKey* d; bool armed = true; ~Undo() { if (armed) release_zkey(c, *d); } } undo{ctx, &d};   // a failing table must not leak the ones before it
    const Curve& c = z.curve;
    auto reg = [&](const auto& pts, int group, const char* name = "") {
        cg_bases* b; CG(cg_bases_register(ctx, c.id, group, pts.data(), pts.size() / c.aff(group), c.aff(group), -1, &b));
        if (validate) { try { validate_bases(ctx, b, name); } catch (...) { cg_bases_release(b); throw; } }
        return b;
    };
    if (validate && rank == 0) {   // the O(1) verifying-key points and IC go through the same kernels
        Bytes g1 = z.alpha_g1; g1.insert(g1.end(), z.beta_g1.begin(), z.beta_g1.end()); g1.insert(g1.end(), z.delta_g1.begin(), z.delta_g1.end()); g1.insert(g1.end(), z.ic.begin(), z.ic.end());
        Bytes g2 = z.beta_g2; g2.insert(g2.end(), z.gamma_g2.begin(), z.gamma_g2.end()); g2.insert(g2.end(), z.delta_g2.begin(), z.delta_g2.end());
        cg_bases_release(reg(g1, CG_G1, "vk_g1/ic")); cg_bases_release(reg(g2, CG_G2, "vk_g2"));
    }
    auto up = [&](const void* src, size_t bytes) { void* p; CG(cg_dev_alloc(ctx, bytes, &p)); if (bytes) CG(cg_dev_upload(ctx, p, src, bytes)); return p; };
    if (world > 1) {
        const size_t first_aux = z.n_public + 1, n_aux = z.n_vars - first_aux;
        const auto ar = slice_of(n_aux, rank, world), hr = slice_of(z.domain_size, rank, world);
        d.sliced = true; d.aux_lo = ar.first; d.aux_n = ar.second - ar.first; d.h_lo = hr.first; d.h_n = hr.second - hr.first;
        auto cut = [&](const View& v, int group, size_t first, size_t count) { return View{v.data() + first * c.aff(group), count * c.aff(group)}; };
        if (validate && rank == 0) {   // the public-input records of a, b1, b2 stay on the host (calculate_coeff): checked here once
            cg_bases_release(reg(cut(z.a_query, CG_G1, 0, first_aux), CG_G1, "a_query")); cg_bases_release(reg(cut(z.b_g1_query, CG_G1, 0, first_aux), CG_G1, "b_g1_query"));
            cg_bases_release(reg(cut(z.b_g2_query, CG_G2, 0, first_aux), CG_G2, "b_g2_query"));
        }
        d.a = reg(cut(z.a_query, CG_G1, first_aux + d.aux_lo, d.aux_n), CG_G1, "a_query"); d.b1 = reg(cut(z.b_g1_query, CG_G1, first_aux + d.aux_lo, d.aux_n), CG_G1, "b_g1_query");
        d.b2 = reg(cut(z.b_g2_query, CG_G2, first_aux + d.aux_lo, d.aux_n), CG_G2, "b_g2_query");
        d.l = reg(cut(z.l_query, CG_G1, d.aux_lo, d.aux_n), CG_G1, "l_query"); d.h = reg(cut(z.h_query, CG_G1, d.h_lo, d.h_n), CG_G1, "h_query");
        if (rank != 0) { undo.armed = false; return d; }
    } else {
        d.a = reg(z.a_query, CG_G1, "a_query"); d.b1 = reg(z.b_g1_query, CG_G1, "b_g1_query"); d.b2 = reg(z.b_g2_query, CG_G2, "b_g2_query");
        d.l = reg(z.l_query, CG_G1, "l_query"); d.h = reg(z.h_query, CG_G1, "h_query");
    }
    for (int m = 0; m < 2; m++) {
        d.mat[m].row_ptr = (uint32_t*)up(z.row_ptr[m].data(), z.row_ptr[m].size() * 4);
        d.mat[m].col = (uint32_t*)up(z.col[m].data(), z.col[m].size() * 4);
        d.mat[m].coeff = up(z.coeff[m].data(), z.coeff[m].size() * 32);
        d.mat[m].rows = z.num_constraints;
    }
    d.pub_dev = up(public_inputs.data(), public_inputs.size() * 32);
    undo.armed = false;
    return d;
}
static void release_zkey(cg_ctx* ctx, DeviceZKey& d) {
    for (cg_bases** b : {&d.a, &d.b1, &d.b2, &d.l, &d.h}) { if (*b) cg_bases_release(*b); *b = nullptr; }
    for (int m = 0; m < 2; m++) {
        if (d.mat[m].row_ptr) cg_dev_free(ctx, d.mat[m].row_ptr); if (d.mat[m].col) cg_dev_free(ctx, d.mat[m].col); if (d.mat[m].coeff) cg_dev_free(ctx, d.mat[m].coeff);
        d.mat[m] = DeviceMatrix{nullptr, nullptr, nullptr, 0};
    }
    if (d.pub_dev) cg_dev_free(ctx, d.pub_dev);
    d.pub_dev = nullptr;
}
// scope guards of the C entry points: whatever a failing proof leaves behind on the device is released (a long-lived prover that
// hits "randomness stream exhausted" a few times must not run out of HBM)
struct DeviceZKeyGuard {
    cg_ctx* ctx; DeviceZKey dz; bool live = true;
    DeviceZKeyGuard(cg_ctx* c, DeviceZKey d) : ctx(c), dz(d) {}
    ~DeviceZKeyGuard() { if (live) release_zkey(ctx, dz); }
    DeviceZKeyGuard(const DeviceZKeyGuard&) = delete; DeviceZKeyGuard& operator=(const DeviceZKeyGuard&) = delete;
};
struct CtxGuard {
    cg_ctx* ctx = nullptr;
    ~CtxGuard() { if (ctx) cg_ctx_destroy(ctx); }
    cg_ctx* release() { cg_ctx* c = ctx; ctx = nullptr; return c; }
};
struct DevBufGuard { cg_ctx* ctx; void* p; ~DevBufGuard() { if (p) cg_dev_free(ctx, p); } };

// Second contexts for the witness-independent MSMs (HipDriver::aux).  Creating a context costs 15-25 ms (its streams), so they are
// made on a helper thread while the zkey is read and uploaded, and only for zkeys large enough (>= ~2^19 constraints) to gain.
struct SecondContexts {
    std::vector<cg_ctx*> made; std::thread worker;
    SecondContexts(int device, const char* zkey_path, int count) : made(count, nullptr) {
        struct stat st{};
        if (getenv("CGH_ONE_CONTEXT") || stat(zkey_path, &st) != 0 || st.st_size < (off_t)200 << 20) return;
        worker = std::thread([this, device] { for (auto& c : made) if (cg_ctx_create(device, &c) != 0) c = nullptr; });
    }
    void ready() { if (worker.joinable()) worker.join(); }
    cg_ctx* take(int i) { ready(); cg_ctx* c = made[i]; made[i] = nullptr; return c; }
    ~SecondContexts() { ready(); for (cg_ctx* c : made) if (c) cg_ctx_destroy(c); }
};

static void store_proof(const Proof& p, uint8_t* out) { memcpy(out, p.a.data(), p.a.size()); memcpy(out + p.a.size(), p.b.data(), p.b.size()); memcpy(out + p.a.size() + p.b.size(), p.c.data(), p.c.size()); }

static bool all_zero_bytes(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }
// ==================================================================================================== co-plonk, round 1
// First slice of the Plonk prover on the same kernels: [a]_1, [b]_1, [c]_1 = MSM(p_tau, blind(iNTT(wire values))).
// The reference pins the exact result for the blinding b_i = i (co-plonk/src/round1.rs:346-383).
struct PlonkZKey {   // circom-types/src/plonk/zkey.rs:18-42 (the fields round 1 reads)
    Curve curve;
    size_t n_vars = 0, n_public = 0, domain_size = 0, power = 0, n_additions = 0, n_constraints = 0;
    struct Addition { uint32_t id1, id2; Fr f1, f2; };
    std::vector<Addition> additions;
    std::vector<uint32_t> map[3];
    Bytes p_tau;        // domain_size + 6 packed G1 points
    Fr k1, k2;          // verifying key, zkey.rs:328-356
    Bytes vk_g1;        // qm, ql, qr, qo, qc, s1, s2, s3 (8 packed G1 points)
    std::vector<Fr> sigma_eval[3];   // 4 * domain_size evaluations of sigma1..3 (section 12, zkey.rs:116-135,170-180)
    std::vector<Fr> q_eval[5];       // qm, ql, qr, qo, qc on the extended domain (sections 7..11)
    std::vector<std::vector<Fr>> lagrange_eval;   // n_public polynomials on the extended domain (section 13)
    std::vector<Fr> q_coef[5], sigma_coef[3];     // coefficient forms (rounds 4 and 5)
};
static PlonkZKey read_plonk_zkey(int curve_id, const std::string& path) {   // zkey.rs:83-255, header :373-424
    Curve c{curve_id};
    Bytes buf = slurp(path);
    Cursor cur{buf.data(), buf.size()};
    char magic[5] = {0}; cur.bytes(magic, 4);
    if (std::string(magic) != "zkey") throw std::runtime_error("not a zkey file");
    cur.u32();
    uint32_t ns = cur.u32();
    std::map<uint32_t, std::pair<size_t, size_t>> sec;
    for (uint32_t i = 0; i < ns; i++) { uint32_t id = cur.u32(); uint64_t len = cur.u64(); cur.need(len); sec[id] = {cur.off, (size_t)len}; cur.off += len; }
    auto section = [&](uint32_t id) { auto it = sec.find(id); if (it == sec.end()) throw std::runtime_error("missing zkey section"); return Cursor{buf.data() + it->second.first, it->second.second}; };
    if (section(1).u32() != 2) throw std::runtime_error("not a plonk zkey");
    PlonkZKey z; z.curve = c;
    Cursor h = section(2);
    if (h.u32() != c.fq()) throw std::runtime_error("unexpected base field byte size");
    uint64_t q[6] = {0}; h.bytes(q, c.fq());
    if (memcmp(q, MOD_Q[curve_id], c.fq())) throw std::runtime_error("invalid base prime in header");
    if (h.u32() != 32) throw std::runtime_error("unexpected scalar field byte size");
    uint64_t r[4]; h.bytes(r, 32);
    if (memcmp(r, MOD_R[curve_id], 32)) throw std::runtime_error("invalid scalar prime in header");
    z.n_vars = h.u32(); z.n_public = h.u32(); z.domain_size = h.u32(); z.n_additions = h.u32(); z.n_constraints = h.u32();
    if (!z.domain_size || (z.domain_size & (z.domain_size - 1))) throw std::runtime_error("Invalid domain size. Must be power of 2");
    while (((size_t)1 << z.power) < z.domain_size) z.power++;
    h.bytes(z.k1.v, 32); h.bytes(z.k2.v, 32);
    z.vk_g1.resize(8 * c.aff(CG_G1)); h.bytes(z.vk_g1.data(), z.vk_g1.size());
    {
        Cursor sg = section(12);
        for (int k = 0; k < 3; k++) {
            z.sigma_coef[k].resize(z.domain_size); sg.bytes(z.sigma_coef[k].data(), z.domain_size * 32);
            z.sigma_eval[k].resize(4 * z.domain_size);
            sg.bytes(z.sigma_eval[k].data(), 4 * z.domain_size * 32);
        }
    }
    for (int k = 0; k < 5; k++) { Cursor q = section(7 + k); z.q_coef[k].resize(z.domain_size); q.bytes(z.q_coef[k].data(), z.domain_size * 32); z.q_eval[k].resize(4 * z.domain_size); q.bytes(z.q_eval[k].data(), 4 * z.domain_size * 32); }
    { Cursor l = section(13); z.lagrange_eval.resize(z.n_public); for (auto& v : z.lagrange_eval) { l.need(z.domain_size * 32); l.off += z.domain_size * 32; v.resize(4 * z.domain_size); l.bytes(v.data(), 4 * z.domain_size * 32); } }
    { Cursor a = section(3); z.additions.resize(z.n_additions); for (auto& e : z.additions) { e.id1 = a.u32(); e.id2 = a.u32(); a.bytes(e.f1.v, 32); a.bytes(e.f2.v, 32); } }
    for (int k = 0; k < 3; k++) { Cursor m = section(4 + k); z.map[k].resize(z.n_constraints); for (auto& v : z.map[k]) v = m.u32(); }
    { Cursor t = section(14); z.p_tau.resize((z.domain_size + 6) * c.aff(CG_G1)); t.bytes(z.p_tau.data(), z.p_tau.size()); }
    return z;
}

// Keccak-256 (pad 0x01) and the reference's transcript conventions (co-plonk/src/types.rs:122-176): big-endian canonical field
// bytes, 2 * byte_len zero bytes for the point at infinity, challenge = digest as a big-endian integer mod r
class Keccak256 {
    uint64_t a[25]; uint8_t blk[136]; size_t used = 0;
    static uint64_t rotl(uint64_t v, unsigned s) { return s ? (v << s) | (v >> (64 - s)) : v; }
    void f1600() {
        uint64_t lfsr = 1;
        for (int round = 0; round < 24; round++) {
            uint64_t col[5];
            for (int x = 0; x < 5; x++) col[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
            for (int x = 0; x < 5; x++) { const uint64_t d = col[(x + 4) % 5] ^ rotl(col[(x + 1) % 5], 1); for (int y = 0; y < 25; y += 5) a[y + x] ^= d; }
            // rho + pi along the standard lane walk
            int x = 1, y = 0; uint64_t cur = a[1];
            for (int t = 0; t < 24; t++) {
                const int nx = y, ny = (2 * x + 3 * y) % 5;
                const uint64_t nxt = a[nx + 5 * ny];
                a[nx + 5 * ny] = rotl(cur, ((t + 1) * (t + 2) / 2) % 64);
                cur = nxt; x = nx; y = ny;
            }
            for (int yy = 0; yy < 25; yy += 5) { uint64_t r[5]; for (int xx = 0; xx < 5; xx++) r[xx] = a[yy + xx]; for (int xx = 0; xx < 5; xx++) a[yy + xx] = r[xx] ^ (~r[(xx + 1) % 5] & r[(xx + 2) % 5]); }
            for (int j = 0; j < 7; j++) {                                               // iota via the degree-8 LFSR
                const bool bit = lfsr & 1; lfsr = (lfsr << 1) ^ ((lfsr >> 7) * 0x71); lfsr &= 0xff;
                if (bit) a[0] ^= (uint64_t)1 << ((1 << j) - 1);
            }
        }
    }
    void absorb() { for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); a[i] ^= w; } f1600(); used = 0; }
public:
    Keccak256() { memset(a, 0, sizeof a); }
    void update(const uint8_t* p, size_t n) { while (n--) { blk[used++] = *p++; if (used == sizeof blk) absorb(); } }
    void finish(uint8_t out[32]) { memset(blk + used, 0, sizeof blk - used); blk[used] ^= 0x01; blk[sizeof blk - 1] ^= 0x80; absorb(); memcpy(out, a, 32); }
};
class PlonkTranscript {
    const Curve& c; Keccak256 h;
    void be_bytes(const uint64_t* canonical, size_t nbytes) { std::vector<uint8_t> be(nbytes); for (size_t i = 0; i < nbytes; i++) be[nbytes - 1 - i] = (uint8_t)(canonical[i / 8] >> (8 * (i % 8))); h.update(be.data(), nbytes); }
public:
    explicit PlonkTranscript(const Curve& cv) : c(cv) {}
    void add_scalar(const Fr& s) { uint64_t can[4]; CG(cg_fr_to_canonical(c.id, s.v, can, 1)); be_bytes(can, 32); }
    void add_point(const uint8_t* aff) {                                                // packed affine G1, (0,0) = infinity
        if (all_zero_bytes(aff, c.aff(CG_G1))) { std::vector<uint8_t> z(2 * c.fq(), 0); h.update(z.data(), z.size()); return; }
        uint64_t can[12]; CG(cg_fq_to_canonical(c.id, aff, can, 2));
        be_bytes(can, c.fq()); be_bytes(can + c.fq() / 8, c.fq());
    }
    Fr get_challenge() {
        uint8_t d[32]; h.finish(d);
        Fr acc = fr_from_u64(c, 0); const Fr b = fr_from_u64(c, 256);                  // from_be_bytes_mod_order
        for (int i = 0; i < 32; i++) acc = fr_add(c, fr_mul(c, acc, b), fr_from_u64(c, d[i]));
        return acc;
    }
};

// ==================================================================================================== co-plonk, all rounds, any driver
// The five rounds (co-plonk/src/round1..5.rs) written once over share-vector operations: per-component kernels for everything
// linear, the driver's protocols for products of two shared vectors (`mul_vec`), for `array_prod_mul` / `inv_many` (round2.rs:18-41,
// rep3.rs:544-558, shamir.rs:521-535) and for openings.  Plain, REP3 and Shamir run the same code; values that every party reconstructs (commitments,
// evaluations) are functions of the witness and of the opened blinding values only.
class CoPlonk {
public:
    HipDriver& d; const PlonkZKey& z; const cg_bases* tau;
    const Curve c; cg_ctx* ctx; const size_t n, N; const int k;
    Fr zero, one, omega, omega4, w2r;
    FieldShare b[11];
    std::vector<Fr> pub;                       // n_public + 1 values, entry 0 forced to 0 (types.rs:107-109)
    ShareVec buf[3], poly[3], evl[3], poly_z, eval_z, tpart[3];
    Point commit[3], commit_z, commit_t[3], commit_wxi, commit_wxiw;
    Fr beta, gamma, alpha, xi, v[5], ev_a, ev_b, ev_c, ev_s1, ev_s2, ev_zw;
    std::vector<ShareVec> tmp_vecs; std::vector<void*> tmp_ptrs;

    CoPlonk(HipDriver& drv, const PlonkZKey& zk, const cg_bases* p_tau, const std::vector<Fr>& public_inputs, const FieldShare* blind)
        : d(drv), z(zk), tau(p_tau), c(drv.curve), ctx(drv.ctx), n(zk.domain_size), N(4 * zk.domain_size), k(drv.k()), pub(public_inputs) {
        if (pub.size() != z.n_public + 1) throw std::runtime_error("public input length does not match the zkey");
        zero = fr_from_u64(c, 0); one = fr_from_u64(c, 1);
        pub[0] = zero;
        const SnarkjsRoots rt = snarkjs_roots(c);
        omega = rt.roots[z.power]; omega4 = rt.roots[z.power + 2]; w2r = rt.roots[2];
        for (int i = 0; i < 11; i++) b[i] = blind[i];
    }
    ~CoPlonk() {
        release_tmp();
        for (ShareVec* sv : {&buf[0], &buf[1], &buf[2], &poly[0], &poly[1], &poly[2], &evl[0], &evl[1], &evl[2], &poly_z, &eval_z, &tpart[0], &tpart[1], &tpart[2]}) d.free_vec(*sv);
    }
    // ---- share-vector helpers ------------------------------------------------------------------------------------------------
    Fr neg(const Fr& v) const { return fr_sub(c, zero, v); }
    Fr M(const Fr& a, const Fr& x) const { return fr_mul(c, a, x); }
    Fr A(const Fr& a, const Fr& x) const { return fr_add(c, a, x); }
    static uint8_t* at(const ShareVec& s, int j, size_t off = 0) { return (uint8_t*)s.c[j] + off * 32; }
    ShareVec T(size_t len) { ShareVec v = d.alloc_vec(len); tmp_vecs.push_back(v); return v; }          // zeroed temporary, freed by release_tmp
    ShareVec keep(ShareVec v) { tmp_vecs.push_back(v); return v; }
    void* Tp(size_t len) { void* p = d.dalloc(len * 32); tmp_ptrs.push_back(p); return p; }
    void* upload(const std::vector<Fr>& h) { void* p = Tp(h.size()); CG(cg_dev_upload(ctx, p, h.data(), h.size() * 32)); return p; }
    void release_tmp() { for (auto& v : tmp_vecs) d.free_vec(v); tmp_vecs.clear(); for (void* p : tmp_ptrs) CG(cg_dev_free(ctx, p)); tmp_ptrs.clear(); }
    static ShareVec view(const ShareVec& s, size_t off, size_t len) { ShareVec v; v.n = len; for (int j = 0; j < 2; j++) v.c[j] = s.c[j] ? (uint8_t*)s.c[j] + off * 32 : nullptr; return v; }
    void copy(const ShareVec& o, const ShareVec& a, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_gather_strided_dev(ctx, c.id, o.c[j], a.c[j], len, 0, 1)); }
    void add(const ShareVec& o, const ShareVec& a, const ShareVec& x, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_add_dev(ctx, c.id, o.c[j], a.c[j], x.c[j], len)); }
    void sub(const ShareVec& o, const ShareVec& a, const ShareVec& x, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_sub_dev(ctx, c.id, o.c[j], a.c[j], x.c[j], len)); }
    void scale(const ShareVec& o, const ShareVec& a, const Fr& f, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_affine_dev(ctx, c.id, o.c[j], a.c[j], len, f.v, nullptr)); }   // mul_with_public
    void mulpub(const ShareVec& o, const ShareVec& a, const void* pv, size_t len) { for (int j = 0; j < k; j++) CG(cg_vec_mul_dev(ctx, c.id, o.c[j], a.c[j], pv, len)); }              // by a public vector
    void axpy(const ShareVec& o, const ShareVec& a, const Fr& f, size_t len) { ShareVec t = T(len); scale(t, a, f, len); add(o, o, t, len); }                                           // o += f * a
    void addpub_vec(const ShareVec& o, const ShareVec& a, const void* pv, size_t len) {                                               // add_with_public, element-wise
        const int pc = d.public_component();
        for (int j = 0; j < k; j++) { if (j == pc) CG(cg_vec_add_dev(ctx, c.id, o.c[j], a.c[j], pv, len)); else if (o.c[j] != a.c[j]) CG(cg_vec_gather_strided_dev(ctx, c.id, o.c[j], a.c[j], len, 0, 1)); }
    }
    void addpub_scalar(const ShareVec& o, const ShareVec& a, const Fr& f, size_t len) {
        const int pc = d.public_component();
        for (int j = 0; j < k; j++) { if (j == pc) CG(cg_vec_affine_dev(ctx, c.id, o.c[j], a.c[j], len, one.v, f.v)); else if (o.c[j] != a.c[j]) CG(cg_vec_gather_strided_dev(ctx, c.id, o.c[j], a.c[j], len, 0, 1)); }
    }
    void axpy_pub(const ShareVec& o, const void* pv, const Fr& f, size_t len) {                                                       // o += f * (public vector)
        const int pc = d.public_component(); if (pc < 0) return;
        void* t = Tp(len); CG(cg_vec_affine_dev(ctx, c.id, t, pv, len, f.v, nullptr)); CG(cg_vec_add_dev(ctx, c.id, o.c[pc], o.c[pc], t, len));
    }
    void affine_share(const ShareVec& o, const void* pv, const FieldShare& kk, const FieldShare& dd, size_t len) {                    // o = kk * (public vector) + dd, share-valued kk, dd
        for (int j = 0; j < k; j++) CG(cg_vec_affine_dev(ctx, c.id, o.c[j], pv, len, kk.c[j].v, dd.c[j].v));
    }
    FieldShare get(const ShareVec& s, size_t i) { FieldShare f; f.c[0] = f.c[1] = zero; for (int j = 0; j < k; j++) CG(cg_dev_download(ctx, f.c[j].v, at(s, j, i), 32)); return f; }
    void set(const ShareVec& s, size_t i, const FieldShare& f) { for (int j = 0; j < k; j++) CG(cg_dev_upload(ctx, at(s, j, i), f.c[j].v, 32)); }
    FieldShare fs_sub(const FieldShare& a, const FieldShare& x) const { FieldShare r; for (int j = 0; j < 2; j++) r.c[j] = fr_sub(c, a.c[j], x.c[j]); return r; }
    FieldShare fs_addpub(FieldShare a, const Fr& f) const { const int pc = d.public_component(); if (pc >= 0) a.c[pc] = fr_add(c, a.c[pc], f); return a; }
    ShareVec mul(const ShareVec& a, const ShareVec& x, size_t len) { ShareVec av = view(a, 0, len), xv = view(x, 0, len); return keep(d.mul_vec(av, xv)); }                // mul_vec / mul_many
    Point commit_open(const ShareVec& p, size_t len) {
        if (len > z.domain_size + 6) throw std::runtime_error("polynomial degree too large");
        return d.open_point(d.msm_public_points(tau, CG_G1, 0, len, p));
    }
    void ntt(const ShareVec& s, size_t len, const Fr& g, bool inverse) { void* ptrs[2] = {s.c[0], s.c[1]}; CG(cg_ntt_dev(ctx, c.id, ptrs, k, len, g.v, inverse ? 1 : 0, nullptr)); }
    // inv_many (rep3.rs:544-558 / plain): element-wise inverse of a shared vector
    ShareVec inv_many(const ShareVec& a, size_t len) {
        ShareVec out = T(len);
        if (d.mode == Mode::Plain) { CG(cg_vec_inverse_dev(ctx, c.id, out.c[0], a.c[0], len)); return out; }
        ShareVec r = keep(d.rand_vec(len));
        void* y = d.mul_open_vec(view(a, 0, len), r); tmp_ptrs.push_back(y);
        CG(cg_vec_inverse_dev(ctx, c.id, y, y, len));                                 // (a zero would make the reference fail with "cannot compute inverse of zero")
        mulpub(out, r, y, len);
        return out;
    }
    // array_prod_mul (round2.rs:18-41): shared prefix products in a constant number of rounds
    ShareVec array_prod_mul(const ShareVec& inp, size_t len) {
        if (d.mode == Mode::Plain) { ShareVec out = T(len); CG(cg_vec_prefix_prod_dev(ctx, c.id, out.c[0], inp.c[0], len)); return out; }
        ShareVec r = keep(d.rand_vec(len + 1));
        ShareVec r_inv = inv_many(r, len + 1);
        ShareVec r_inv0 = T(len);
        const FieldShare first = get(r_inv, 0);
        for (int j = 0; j < k; j++) CG(cg_vec_fill_dev(ctx, c.id, r_inv0.c[j], len, first.c[j].v));
        ShareVec unblind = mul(r_inv0, view(r, 1, len), len);
        ShareVec m = mul(view(r, 0, len), inp, len);
        void* open = d.mul_open_vec(m, view(r_inv, 1, len)); tmp_ptrs.push_back(open);
        CG(cg_vec_prefix_prod_dev(ctx, c.id, open, open, len));
        mulpub(unblind, unblind, open, len);
        return unblind;
    }
    Fr eval_pub_poly(const void* d_poly, size_t len, const Fr& x) {                    // Horner of a public polynomial as a scan
        void* t = Tp(len);
        CG(cg_vec_gather_strided_dev(ctx, c.id, t, d_poly, len, 0, 1));
        CG(cg_vec_distribute_powers_dev(ctx, c.id, t, len, x.v, one.v));
        CG(cg_vec_prefix_sum_dev(ctx, c.id, t, t, len));
        Fr r; CG(cg_dev_download(ctx, r.v, (const uint8_t*)t + (len - 1) * 32, 32));
        return r;
    }
    FieldShare eval_share_poly(const ShareVec& p, size_t len, const Fr& x) {           // evaluate_poly_public (rep3.rs:923-931)
        FieldShare f; f.c[0] = f.c[1] = zero;
        for (int j = 0; j < k; j++) f.c[j] = eval_pub_poly(p.c[j], len, x);
        return f;
    }
    void div_by_zerofier1(const ShareVec& p, size_t len, const Fr& point) {             // round5.rs:97-115 with n = 1; the caller drops the last entry
        const Fr pinv = fr_inv(c, point);
        for (int j = 0; j < k; j++) {
            CG(cg_vec_affine_dev(ctx, c.id, p.c[j], p.c[j], len, neg(pinv).v, nullptr));
            CG(cg_vec_distribute_powers_dev(ctx, c.id, p.c[j], len, point.v, one.v));
            CG(cg_vec_prefix_sum_dev(ctx, c.id, p.c[j], p.c[j], len));
            CG(cg_vec_distribute_powers_dev(ctx, c.id, p.c[j], len, pinv.v, one.v));
        }
    }
    void transcript_point(PlonkTranscript& t, const Point& p) { Bytes a = pt_to_affine(c, p); t.add_point(a.data()); }

    // ---- round 1 (round1.rs:118-312) ---------------------------------------------------------------------------------------------
    FieldShare trivial(const Fr& v) const { FieldShare f; f.c[0] = f.c[1] = zero; const int pc = d.public_component(); if (pc >= 0) f.c[pc] = v; return f; }
    ShareVec extend_witness(const ShareVec& wit) {                                       // calculate_additions (:208-238)
        const size_t n_priv = z.n_vars - z.n_additions - z.n_public - 1;
        std::vector<Fr> ext[2];
        for (int j = 0; j < k; j++) { ext[j].resize(n_priv + z.n_additions); if (n_priv) CG(cg_dev_download(ctx, ext[j].data(), wit.c[j], n_priv * 32)); }
        size_t have = n_priv;
        auto getw = [&](size_t idx) -> FieldShare {
            if (idx <= z.n_public) return trivial(pub[idx]);
            if (idx >= z.n_vars || idx - z.n_public - 1 >= have) throw std::runtime_error("Cannot index into witness " + std::to_string(idx));
            FieldShare f; f.c[0] = f.c[1] = zero; for (int j = 0; j < k; j++) f.c[j] = ext[j][idx - z.n_public - 1];
            return f;
        };
        for (const auto& a : z.additions) { FieldShare w1 = getw(a.id1), w2 = getw(a.id2); for (int j = 0; j < k; j++) ext[j][have] = A(M(a.f1, w1.c[j]), M(a.f2, w2.c[j])); have++; }
        return d.upload_vec(ext[0].data(), k == 2 ? ext[1].data() : nullptr, ext[0].size());
    }
    void round1(const ShareVec& private_witness) {
        const size_t nc = z.n_constraints;
        if (private_witness.n != z.n_vars - z.n_additions - z.n_public - 1) throw std::runtime_error("witness length does not match the zkey");
        ShareVec ext = z.n_additions ? keep(extend_witness(private_witness)) : private_witness;
        void* d_pub = upload(pub);
        std::vector<uint32_t> row_ptr(nc + 1); for (size_t i = 0; i <= nc; i++) row_ptr[i] = (uint32_t)i;
        uint32_t* d_rp = (uint32_t*)Tp((nc + 8) / 8 + 1); CG(cg_dev_upload(ctx, d_rp, row_ptr.data(), (nc + 1) * 4));
        void* d_one = Tp(std::max<size_t>(nc, 1)); CG(cg_vec_fill_dev(ctx, c.id, d_one, std::max<size_t>(nc, 1), one.v));
        uint32_t* d_col = (uint32_t*)Tp((nc + 8) / 8 + 1);
        for (int w = 0; w < 3; w++) {
            if (nc) CG(cg_dev_upload(ctx, d_col, z.map[w].data(), nc * 4));
            poly[w] = d.alloc_vec(n + 2);
            // the wire buffers are gathers: one-entry CSR rows with coefficient one reuse the constraint-evaluation kernel (get_witness' public /
            // private split with the REP3 party asymmetry, lib.rs:113-137)
            CG(cg_spmv_csr_dev(ctx, c.id, d_rp, d_col, d_one, nc, d_pub, (uint32_t)(z.n_public + 1), d.party(), ext.c[0], ext.c[1], poly[w].c[0], poly[w].c[1]));
            buf[w] = d.alloc_vec(n); copy(buf[w], poly[w], n);
            ntt(poly[w], n, omega, true);                                                // :170-172
            evl[w] = d.alloc_vec(N); copy(evl[w], poly[w], n); ntt(evl[w], N, omega4, false);   // :174-177
            const FieldShare &b_hi = b[2 * w], &b_lo = b[2 * w + 1];                     // blind_coefficients (lib.rs:140-158)
            set(poly[w], 0, fs_sub(get(poly[w], 0), b_lo)); set(poly[w], 1, fs_sub(get(poly[w], 1), b_hi));
            set(poly[w], n, b_lo); set(poly[w], n + 1, b_hi);
        }
        PointShare cm[3];
        for (int w = 0; w < 3; w++) cm[w] = d.msm_public_points(tau, CG_G1, 0, n + 2, poly[w]);   // :276-290
        for (int w = 0; w < 3; w++) commit[w] = d.open_point(cm[w]);                               // open_point_many (:292)
        release_tmp();
    }
    // ---- round 2 (round2.rs:146-298) ---------------------------------------------------------------------------------------------
    void round2() {
        {
            PlonkTranscript t(c);
            for (int i = 0; i < 8; i++) t.add_point(z.vk_g1.data() + i * c.aff(CG_G1));
            for (size_t i = 1; i < pub.size(); i++) t.add_scalar(pub[i]);
            for (int w = 0; w < 3; w++) transcript_point(t, commit[w]);
            beta = t.get_challenge();
            PlonkTranscript t2(c); t2.add_scalar(beta); gamma = t2.get_challenge();
        }
        void* betaw = Tp(n); CG(cg_vec_fill_dev(ctx, c.id, betaw, n, beta.v)); CG(cg_vec_distribute_powers_dev(ctx, c.id, betaw, n, omega.v, one.v));
        void* pv = Tp(n); void* sig = Tp(n);
        const Fr kk[3] = {one, z.k1, z.k2};
        ShareVec num, den;
        for (int w = 0; w < 3; w++) {                                                     // :162-216
            ShareVec f = T(n);
            CG(cg_vec_affine_dev(ctx, c.id, pv, betaw, n, kk[w].v, gamma.v));
            addpub_vec(f, buf[w], pv, n);
            num = w == 0 ? f : mul(num, f, n);
            void* d_sigma = upload(z.sigma_eval[w]);
            CG(cg_vec_gather_strided_dev(ctx, c.id, sig, d_sigma, n, 0, 4));
            CG(cg_vec_affine_dev(ctx, c.id, pv, sig, n, beta.v, gamma.v));
            ShareVec g = T(n);
            addpub_vec(g, buf[w], pv, n);
            den = w == 0 ? g : mul(den, g, n);
        }
        ShareVec num_p = array_prod_mul(num, n), den_p = array_prod_mul(den, n);           // :218-224
        ShareVec den_i = inv_many(den_p, n);                                               // :228
        ShareVec zb = mul(num_p, den_i, n);                                                // :229
        poly_z = d.alloc_vec(n + 3);
        if (n > 1) copy(view(poly_z, 1, n - 1), zb, n - 1);                                // rotate_right(1) (:230)
        copy(poly_z, view(zb, n - 1, 1), 1);
        ntt(poly_z, n, omega, true);                                                       // :235
        eval_z = d.alloc_vec(N); copy(eval_z, poly_z, n); ntt(eval_z, N, omega4, false);   // :238
        set(poly_z, 0, fs_sub(get(poly_z, 0), b[8])); set(poly_z, 1, fs_sub(get(poly_z, 1), b[7])); set(poly_z, 2, fs_sub(get(poly_z, 2), b[6]));
        set(poly_z, n, b[8]); set(poly_z, n + 1, b[7]); set(poly_z, n + 2, b[6]);
        commit_z = commit_open(poly_z, n + 3);                                             // :268-275
        release_tmp();
    }
    // ---- round 3 (round3.rs:234-527) ---------------------------------------------------------------------------------------------
    void round3() {
        if (z.lagrange_eval.empty()) throw std::runtime_error("round 3 needs at least one public input (lagrange[0])");
        { PlonkTranscript t(c); t.add_scalar(beta); t.add_scalar(gamma); transcript_point(t, commit_z); alpha = t.get_challenge(); }   // :498-503
        const Fr alpha2 = M(alpha, alpha), two = fr_from_u64(c, 2);
        const Fr Z1[4] = {zero, A(neg(one), w2r), neg(two), fr_sub(c, neg(one), w2r)};     // get_z1..3 (:203-232)
        const Fr m2w = M(neg(two), w2r);
        const Fr Z2[4] = {zero, m2w, M(two, two), neg(m2w)};
        const Fr tw = M(two, w2r);
        const Fr Z3[4] = {zero, A(two, tw), neg(M(M(two, two), two)), fr_sub(c, two, tw)};
        auto pattern = [&](const Fr* zz) { std::vector<Fr> h(N); for (size_t i = 0; i < N; i++) h[i] = zz[i & 3]; return upload(h); };
        void* z1p = pattern(Z1); void* z2p = pattern(Z2); void* z3p = pattern(Z3);
        const ShareVec &a = evl[0], &bb = evl[1], &cc = evl[2], &ez = eval_z;
        const FieldShare fzero = trivial(zero);
        // the blinding polynomials on the 4n-th roots of unity (:246-256, :307-322)
        void* pw = Tp(N); CG(cg_vec_fill_dev(ctx, c.id, pw, N, one.v)); CG(cg_vec_distribute_powers_dev(ctx, c.id, pw, N, omega4.v, one.v));
        void* pw2 = Tp(N); CG(cg_vec_mul_dev(ctx, c.id, pw2, pw, pw, N));
        void* pww = Tp(N); CG(cg_vec_affine_dev(ctx, c.id, pww, pw, N, omega.v, nullptr));
        void* pww2 = Tp(N); CG(cg_vec_mul_dev(ctx, c.id, pww2, pww, pww, N));
        ShareVec ap = T(N), bp = T(N), cp = T(N), zp = T(N), zwp = T(N), t0 = T(N);
        affine_share(ap, pw, b[0], b[1], N); affine_share(bp, pw, b[2], b[3], N); affine_share(cp, pw, b[4], b[5], N);
        affine_share(zp, pw2, b[6], b[8], N); affine_share(t0, pw, b[7], fzero, N); add(zp, zp, t0, N);
        affine_share(zwp, pww2, b[6], b[8], N); affine_share(t0, pww, b[7], fzero, N); add(zwp, zwp, t0, N);
        ShareVec zw = T(N);                                                                // z(X omega): eval_z rotated by 4 (:324-327)
        copy(zw, view(ez, 4, N - 4), N - 4); copy(view(zw, N - 4, 4), ez, 4);
        // gate constraint (:333-368)
        ShareVec a_b = mul(a, bb, N), a_bp = mul(a, bp, N), ap_b = mul(bb, ap, N), ap_bp = mul(ap, bp, N);
        ShareVec a0 = T(N); add(a0, a_bp, ap_b, N); mulpub(t0, ap_bp, z1p, N); add(a0, a0, t0, N);
        void* q[5]; for (int i = 0; i < 5; i++) q[i] = upload(z.q_eval[i]);
        ShareVec e1 = T(N), e1z = T(N);
        mulpub(e1, a_b, q[0], N); mulpub(t0, a, q[1], N); add(e1, e1, t0, N); mulpub(t0, bb, q[2], N); add(e1, e1, t0, N); mulpub(t0, cc, q[3], N); add(e1, e1, t0, N);
        addpub_vec(e1, e1, q[4], N);
        mulpub(e1z, a0, q[0], N); mulpub(t0, ap, q[1], N); add(e1z, e1z, t0, N); mulpub(t0, bp, q[2], N); add(e1z, e1z, t0, N); mulpub(t0, cp, q[3], N); add(e1z, e1z, t0, N);
        void* l1 = nullptr;
        for (size_t j = 0; j < z.lagrange_eval.size(); j++) {                              // public-input polynomial (:352-358): pi -= L_j * buffer_a[j]
            void* lj = upload(z.lagrange_eval[j]); if (j == 0) l1 = lj;
            const FieldShare aj = get(buf[0], j);
            for (int cpn = 0; cpn < k; cpn++) { CG(cg_vec_affine_dev(ctx, c.id, t0.c[cpn], lj, N, neg(aj.c[cpn]).v, nullptr)); }
            add(e1, e1, t0, N);
        }
        // permutation constraints (:370-418)
        auto mul4 = [&](const ShareVec& Av, const ShareVec& Bv, const ShareVec& Cv, const ShareVec& Dv, const ShareVec& Dp, ShareVec& r, ShareVec& rz) {   // mul4vec + mul4vec_post (:17-72)
            ShareVec AB = mul(Av, Bv, N);
            ShareVec S1 = mul(Av, bp, N); add(S1, S1, mul(ap, Bv, N), N);                  // A B' + A' B
            ShareVec CD = mul(Cv, Dv, N);
            ShareVec S2 = mul(Cv, Dp, N); add(S2, S2, mul(cp, Dv, N), N);                  // C D' + C' D
            ShareVec CpDp = mul(cp, Dp, N);
            r = mul(AB, CD, N);
            rz = mul(S1, CD, N); add(rz, rz, mul(AB, S2, N), N);
            ShareVec r1 = mul(ap_bp, CD, N); add(r1, r1, mul(S1, S2, N), N); add(r1, r1, mul(AB, CpDp, N), N);
            mulpub(t0, r1, z1p, N); add(rz, rz, t0, N);
            ShareVec r2 = mul(S1, CpDp, N); add(r2, r2, mul(ap_bp, S2, N), N);
            mulpub(t0, r2, z2p, N); add(rz, rz, t0, N);
            ShareVec r3 = mul(ap_bp, CpDp, N);
            mulpub(t0, r3, z3p, N); add(rz, rz, t0, N);
        };
        ShareVec e2, e2z, e3, e3z;
        {
            void* pvv = Tp(N);
            ShareVec fa = T(N), fb = T(N), fc = T(N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, pw, N, beta.v, gamma.v)); addpub_vec(fa, a, pvv, N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, pw, N, M(beta, z.k1).v, gamma.v)); addpub_vec(fb, bb, pvv, N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, pw, N, M(beta, z.k2).v, gamma.v)); addpub_vec(fc, cc, pvv, N);
            mul4(fa, fb, fc, ez, zp, e2, e2z);
            ShareVec ga = T(N), gb = T(N), gc = T(N);
            void* sg[3]; for (int i = 0; i < 3; i++) sg[i] = upload(z.sigma_eval[i]);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, sg[0], N, beta.v, gamma.v)); addpub_vec(ga, a, pvv, N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, sg[1], N, beta.v, gamma.v)); addpub_vec(gb, bb, pvv, N);
            CG(cg_vec_affine_dev(ctx, c.id, pvv, sg[2], N, beta.v, gamma.v)); addpub_vec(gc, cc, pvv, N);
            mul4(ga, gb, gc, zw, zwp, e3, e3z);
        }
        // t = e1 + alpha (e2 - e3) + alpha^2 L1 (z - 1), tz likewise from the blinding parts (:420-441)
        ShareVec Tv = T(N), TZ = T(N);
        sub(t0, e2, e3, N); scale(t0, t0, alpha, N); add(Tv, e1, t0, N);
        addpub_scalar(t0, ez, neg(one), N); mulpub(t0, t0, l1, N); scale(t0, t0, alpha2, N); add(Tv, Tv, t0, N);
        sub(t0, e2z, e3z, N); scale(t0, t0, alpha, N); add(TZ, e1z, t0, N);
        mulpub(t0, zp, l1, N); scale(t0, t0, alpha2, N); add(TZ, TZ, t0, N);
        ntt(Tv, N, omega4, true);                                                          // :442
        scale(view(Tv, 0, n), view(Tv, 0, n), neg(one), n);                                // neg_vec_in_place_limit (:443)
        for (int blk = 1; blk < 4; blk++) sub(view(Tv, blk * n, n), view(Tv, (blk - 1) * n, n), view(Tv, blk * n, n), n);   // division by X^n - 1 (:445-450)
        ntt(TZ, N, omega4, true);
        add(Tv, Tv, TZ, N);                                                                // :453
        const size_t len[3] = {n + 1, n + 1, n + 6};                                        // split (:455-470)
        for (int p = 0; p < 3; p++) { tpart[p] = d.alloc_vec(len[p]); copy(tpart[p], view(Tv, (size_t)p * n, p == 2 ? n + 6 : n), p == 2 ? n + 6 : n); }
        set(tpart[0], n, b[9]);
        set(tpart[1], 0, fs_sub(get(tpart[1], 0), b[9])); set(tpart[1], n, b[10]);
        set(tpart[2], 0, fs_sub(get(tpart[2], 0), b[10]));
        PointShare cm[3];
        for (int p = 0; p < 3; p++) { if (len[p] > z.domain_size + 6) throw std::runtime_error("polynomial degree too large"); cm[p] = d.msm_public_points(tau, CG_G1, 0, len[p], tpart[p]); }
        for (int p = 0; p < 3; p++) commit_t[p] = d.open_point(cm[p]);                     // :507-522
        release_tmp();
    }
    // ---- rounds 4 and 5 (round4.rs:115-160, round5.rs:143-365) --------------------------------------------------------------------
    void round4() {
        { PlonkTranscript t(c); t.add_scalar(alpha); for (int p = 0; p < 3; p++) transcript_point(t, commit_t[p]); xi = t.get_challenge(); }
        const Fr xiw = M(xi, omega);
        std::vector<FieldShare> sh = {eval_share_poly(poly[0], n + 2, xi), eval_share_poly(poly[1], n + 2, xi), eval_share_poly(poly[2], n + 2, xi), eval_share_poly(poly_z, n + 3, xiw)};
        const std::vector<Fr> opened = d.open_many(sh);                                    // :131
        ev_a = opened[0]; ev_b = opened[1]; ev_c = opened[2]; ev_zw = opened[3];
        ev_s1 = eval_pub_poly(upload(z.sigma_coef[0]), n, xi); ev_s2 = eval_pub_poly(upload(z.sigma_coef[1]), n, xi);
        release_tmp();
    }
    void round5() {
        { PlonkTranscript t(c); const Fr* items[7] = {&xi, &ev_a, &ev_b, &ev_c, &ev_s1, &ev_s2, &ev_zw}; for (const Fr* e : items) t.add_scalar(*e);
          v[0] = t.get_challenge(); for (int i = 1; i < 5; i++) v[i] = M(v[i - 1], v[0]); }                                          // :338-350
        const size_t len = n + 6;
        Fr xin = xi; for (size_t i = 0; i < z.power; i++) xin = M(xin, xin);               // lib.rs:160-184
        const Fr zh = fr_sub(c, xin, one);
        std::vector<Fr> l; { Fr wv = one; const Fr nn = fr_from_u64(c, (uint64_t)n); for (size_t i = 0; i < std::max<size_t>(1, z.n_public); i++) { l.push_back(M(M(wv, zh), fr_inv(c, M(nn, fr_sub(c, xi, wv))))); wv = M(wv, omega); } }
        Fr eval_pi = zero; for (size_t i = 1; i < pub.size() && i - 1 < l.size(); i++) eval_pi = fr_sub(c, eval_pi, M(l[i - 1], pub[i]));
        const Fr betaxi = M(beta, xi);
        const Fr e2 = M(M(M(A(A(ev_a, betaxi), gamma), A(A(ev_b, M(betaxi, z.k1)), gamma)), A(A(ev_c, M(betaxi, z.k2)), gamma)), alpha);
        const Fr e3 = M(M(M(A(A(ev_a, M(beta, ev_s1)), gamma), A(A(ev_b, M(beta, ev_s2)), gamma)), ev_zw), alpha);
        const Fr e4 = M(M(alpha, alpha), l[0]), e24 = A(e2, e4);
        ShareVec R = T(len);                                                               // compute_r (:143-260)
        axpy(R, poly_z, e24, n + 3);
        void* s_co[3]; for (int i = 0; i < 3; i++) s_co[i] = upload(z.sigma_coef[i]);
        { const Fr f[5] = {M(ev_a, ev_b), ev_a, ev_b, ev_c, one}; for (int i = 0; i < 5; i++) axpy_pub(R, upload(z.q_coef[i]), f[i], n); }
        axpy_pub(R, s_co[2], neg(M(e3, beta)), n);
        axpy(R, tpart[2], neg(M(zh, M(xin, xin))), n + 6); axpy(R, tpart[1], neg(M(zh, xin)), n + 1); axpy(R, tpart[0], neg(zh), n + 1);
        const Fr r0 = fr_sub(c, fr_sub(c, eval_pi, M(e3, A(ev_c, gamma))), e4);
        // compute_wxi (:263-311)
        for (int w = 0; w < 3; w++) axpy(R, poly[w], v[w], n + 2);
        axpy_pub(R, s_co[0], v[3], n); axpy_pub(R, s_co[1], v[4], n);
        const Fr corr = fr_sub(c, r0, A(A(A(A(M(v[0], ev_a), M(v[1], ev_b)), M(v[2], ev_c)), M(v[3], ev_s1)), M(v[4], ev_s2)));
        set(R, 0, fs_addpub(get(R, 0), corr));
        div_by_zerofier1(R, len, xi);
        // compute_wxiw (:314-327)
        ShareVec W = T(n + 3); copy(W, poly_z, n + 3);
        set(W, 0, fs_addpub(get(W, 0), neg(ev_zw)));
        div_by_zerofier1(W, n + 3, M(xi, omega));
        PointShare c1 = d.msm_public_points(tau, CG_G1, 0, len - 1, R), c2 = d.msm_public_points(tau, CG_G1, 0, n + 2, W);   // :351-358
        commit_wxi = d.open_point(c1); commit_wxiw = d.open_point(c2);
        release_tmp();
    }
};

// ---- JSON encodings of proofs and public inputs (circom-types/src/groth16/proof.rs:8-29, traits.rs:186-233, co-circom.rs:540,628) ----
static std::string limbs_to_dec(const uint64_t* limbs, int n) {          // canonical little-endian -> decimal
    std::vector<uint32_t> w(2 * n);
    for (int i = 0; i < n; i++) { w[2 * i] = (uint32_t)limbs[i]; w[2 * i + 1] = (uint32_t)(limbs[i] >> 32); }
    std::string out;
    while (true) {
        uint64_t rem = 0; bool nz = false;
        for (int i = (int)w.size() - 1; i >= 0; i--) { uint64_t cur = (rem << 32) | w[i]; w[i] = (uint32_t)(cur / 1000000000u); rem = cur % 1000000000u; nz = nz || w[i]; }
        char buf[16];
        if (nz) { snprintf(buf, sizeof buf, "%09u", (unsigned)rem); out.insert(0, buf); }
        else { snprintf(buf, sizeof buf, "%u", (unsigned)rem); out.insert(0, buf); break; }
    }
    return out;
}
static void dec_to_limbs(const std::string& sdec, uint64_t* limbs, int n) {   // decimal -> canonical little-endian (must fit)
    std::vector<uint32_t> w(2 * n, 0);
    if (sdec.empty()) throw std::runtime_error("empty number");
    for (char ch : sdec) {
        if (ch < '0' || ch > '9') throw std::runtime_error("invalid decimal digit");
        uint64_t carry = (uint64_t)(ch - '0');
        for (size_t i = 0; i < w.size(); i++) { uint64_t cur = (uint64_t)w[i] * 10u + carry; w[i] = (uint32_t)cur; carry = cur >> 32; }
        if (carry) throw std::runtime_error("number too large for the field");
    }
    for (int i = 0; i < n; i++) limbs[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
static std::string fq_dec(const Curve& c, const uint8_t* mont) {
    uint64_t can[6]; CG(cg_fq_to_canonical(c.id, mont, can, 1));
    return limbs_to_dec(can, (int)c.fq() / 8);
}
static bool all_zero(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) if (p[i]) return false; return true; }
static const char* curve_name(const Curve& c) { return c.id == CG_BN254 ? "bn128" : "bls12381"; }    // traits.rs:18,31
static std::string g1_json(const Curve& c, const uint8_t* aff) {
    if (all_zero(aff, c.aff(CG_G1))) return "[\"0\",\"1\",\"0\"]";                                       // traits.rs:190-192
    return "[\"" + fq_dec(c, aff) + "\",\"" + fq_dec(c, aff + c.fq()) + "\",\"1\"]";
}
static std::string g2_json(const Curve& c, const uint8_t* aff) {
    if (all_zero(aff, c.aff(CG_G2))) throw std::runtime_error("the point at infinity has no G2 JSON encoding (the reference unwraps xy(), traits.rs:227)");
    const size_t q = c.fq();
    return "[[\"" + fq_dec(c, aff) + "\",\"" + fq_dec(c, aff + q) + "\"],[\"" + fq_dec(c, aff + 2 * q) + "\",\"" + fq_dec(c, aff + 3 * q) + "\"],[\"1\",\"0\"]]";
}
static std::string proof_to_json(const Curve& c, const uint8_t* packed) {   // packed = A (G1) || B (G2) || C (G1)
    const uint8_t *a = packed, *b = packed + c.aff(CG_G1), *cc = b + c.aff(CG_G2);
    return "{\"pi_a\":" + g1_json(c, a) + ",\"pi_b\":" + g2_json(c, b) + ",\"pi_c\":" + g1_json(c, cc) + ",\"protocol\":\"groth16\",\"curve\":\"" + curve_name(c) + "\"}";
}
// the decimal strings of a JSON document, in order (the proof schema is fixed: keys pi_a, pi_b, pi_c carry 3 + 6 + 3 numbers)
static std::vector<std::string> json_numbers_after(const std::string& js, const char* key, size_t count) {
    size_t pos = js.find(std::string("\"") + key + "\"");
    if (pos == std::string::npos) throw std::runtime_error(std::string("missing key ") + key);
    pos = js.find(':', pos);
    std::vector<std::string> out;
    while (out.size() < count) {
        size_t q0 = js.find('"', pos + 1);
        if (q0 == std::string::npos) throw std::runtime_error("truncated proof JSON");
        size_t q1 = js.find('"', q0 + 1);
        if (q1 == std::string::npos) throw std::runtime_error("truncated proof JSON");
        out.push_back(js.substr(q0 + 1, q1 - q0 - 1));
        pos = q1;
    }
    return out;
}
static void proof_from_json(const Curve& c, const std::string& js, uint8_t* packed) {
    if (js.find(std::string("\"") + curve_name(c) + "\"") == std::string::npos) throw std::runtime_error("proof is for another curve");
    const int nl = (int)c.fq() / 8;
    auto put = [&](const std::string& d, uint8_t* dst) { uint64_t can[6] = {0}; dec_to_limbs(d, can, nl); CG(cg_fq_from_canonical(c.id, can, dst, 1)); };
    auto g1 = [&](const char* key, uint8_t* dst) {
        auto v = json_numbers_after(js, key, 3);
        if (v[2] == "0") { memset(dst, 0, c.aff(CG_G1)); return; }          // projective z = 0: infinity
        if (v[2] != "1") throw std::runtime_error("only z = 1 / z = 0 G1 encodings are produced by circom tools");
        put(v[0], dst); put(v[1], dst + c.fq());
    };
    g1("pi_a", packed);
    auto v = json_numbers_after(js, "pi_b", 6);
    if (v[4] != "1" || v[5] != "0") throw std::runtime_error("only z = (1, 0) G2 encodings are produced by circom tools");
    uint8_t* b = packed + c.aff(CG_G1);
    for (int i = 0; i < 4; i++) put(v[i], b + i * c.fq());
    g1("pi_c", b + c.aff(CG_G2));
}

// `.shared` witness files (co-circom.rs:330,400,449: `bincode::serialize_into(file, &SharedWitness)`).  Layout restated from the
// types, NOT pinned by a reference fixture (the snapshot ships no .shared file):
//   SharedWitness { public_inputs, witness } with both fields going through serde_compat::ark_se (co-circom-snarks/src/lib.rs:24-41,
//   serde_compat.rs:5-13) = serialize_bytes(ark-compressed value) = u64 LE byte length, then the bytes;
//   ark-compressed Vec<F> = u64 LE element count, then 32-byte canonical little-endian field elements;
//   Rep3PrimeFieldShareVec { a, b } (rep3/fieldshare.rs:232-236) = Vec a then Vec b; ShamirPrimeFieldShareVec { a } (shamir/fieldshare.rs:152-155) = Vec a.
static void put_u64(Bytes& o, uint64_t v) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put_vec(const Curve& c, Bytes& o, const Fr* v, size_t n) {
    put_u64(o, n);
    std::vector<Fr> can(n); if (n) CG(cg_fr_to_canonical(c.id, v, can.data(), n));
    const uint8_t* p = (const uint8_t*)can.data(); o.insert(o.end(), p, p + n * 32);
}
static std::vector<Fr> get_vec(const Curve& c, Cursor& cur) {
    const uint64_t n = cur.u64(); cur.need(n * 32);
    std::vector<Fr> raw(n), out(n); cur.bytes(raw.data(), n * 32);
    for (const Fr& e : raw) { for (int l = 3; l >= 0; l--) { if (e.v[l] < MOD_R[c.id][l]) break; if (e.v[l] > MOD_R[c.id][l] || l == 0) throw std::runtime_error("invalid data: field element not reduced"); } }
    if (n) CG(cg_fr_from_canonical(c.id, raw.data(), out.data(), n));
    return out;
}
static void write_shared_witness(const Curve& c, const std::string& path, const std::vector<Fr>& pub, const std::vector<Fr>& a, const std::vector<Fr>* b) {
    Bytes f1, f2, out;
    put_vec(c, f1, pub.data(), pub.size());
    put_vec(c, f2, a.data(), a.size()); if (b) put_vec(c, f2, b->data(), b->size());
    put_u64(out, f1.size()); out.insert(out.end(), f1.begin(), f1.end());
    put_u64(out, f2.size()); out.insert(out.end(), f2.begin(), f2.end());
    FILE* f = fopen(path.c_str(), "wb"); if (!f) throw std::runtime_error("cannot open " + path);
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size(); fclose(f);
    if (!ok) throw std::runtime_error("short write " + path);
}
static void read_shared_witness(const Curve& c, const std::string& path, bool rep3, std::vector<Fr>& pub, std::vector<Fr>& a, std::vector<Fr>& b) {
    Bytes buf = slurp(path);
    Cursor cur{buf.data(), buf.size()};
    const uint64_t l1 = cur.u64(); cur.need(l1);
    { Cursor f{buf.data() + cur.off, (size_t)l1}; pub = get_vec(c, f); if (f.off != l1) throw std::runtime_error("trailing bytes in public_inputs"); }
    cur.off += l1;
    const uint64_t l2 = cur.u64(); cur.need(l2);
    { Cursor f{buf.data() + cur.off, (size_t)l2}; a = get_vec(c, f); if (rep3) b = get_vec(c, f); if (f.off != l2) throw std::runtime_error("witness share does not match the protocol (REP3 has two vectors, Shamir one)"); }
    cur.off += l2;
    if (cur.off != buf.size()) throw std::runtime_error("trailing bytes after the shared witness");
    if (rep3 && a.size() != b.size()) throw std::runtime_error("REP3 share components differ in length");
}

// PlonkProof <-> JSON (circom-types/src/plonk/proof.rs:8-74): nine G1 points A, B, C, Z, T1, T2, T3, Wxi, Wxiw, six evaluations, tags
static const char* const PLONK_PT_KEYS[9] = {"A", "B", "C", "Z", "T1", "T2", "T3", "Wxi", "Wxiw"};
static const char* const PLONK_EV_KEYS[6] = {"eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw"};
static std::string plonk_proof_to_json(const Curve& c, const uint8_t* commits /* 9 packed G1 */, const Fr* evals /* 6 */) {
    std::string js = "{";
    for (int i = 0; i < 9; i++) js += std::string("\"") + PLONK_PT_KEYS[i] + "\":" + g1_json(c, commits + i * c.aff(CG_G1)) + ",";
    for (int i = 0; i < 6; i++) { uint64_t can[4]; CG(cg_fr_to_canonical(c.id, evals[i].v, can, 1)); js += std::string("\"") + PLONK_EV_KEYS[i] + "\":\"" + limbs_to_dec(can, 4) + "\","; }
    return js + "\"protocol\":\"plonk\",\"curve\":\"" + curve_name(c) + "\"}";
}
static void plonk_proof_from_json(const Curve& c, const std::string& js, uint8_t* commits, Fr* evals) {
    if (js.find(std::string("\"") + curve_name(c) + "\"") == std::string::npos) throw std::runtime_error("proof is for another curve");
    if (js.find("\"plonk\"") == std::string::npos) throw std::runtime_error("not a plonk proof");
    const int nl = (int)c.fq() / 8;
    for (int i = 0; i < 9; i++) {
        auto v = json_numbers_after(js, PLONK_PT_KEYS[i], 3);
        uint8_t* dst = commits + i * c.aff(CG_G1);
        if (v[2] == "0") { memset(dst, 0, c.aff(CG_G1)); continue; }
        if (v[2] != "1") throw std::runtime_error("only z = 1 / z = 0 G1 encodings are produced by circom tools");
        for (int k = 0; k < 2; k++) { uint64_t can[6] = {0}; dec_to_limbs(v[k], can, nl); CG(cg_fq_from_canonical(c.id, can, dst + k * c.fq(), 1)); }
    }
    for (int i = 0; i < 6; i++) { auto v = json_numbers_after(js, PLONK_EV_KEYS[i], 1); uint64_t can[4] = {0}; dec_to_limbs(v[0], can, 4); CG(cg_fr_from_canonical(c.id, can, evals[i].v, 1)); }
}

// ---- synthetic satisfiable circuit + valid Groth16 CRS (bench / test tooling; SURVEY.md §8d "synthetic R1CS generator") --------------
// Writes a snarkjs-format .zkey (sections 1-9, the layout read_zkey above parses: circom-types/src/groth16/zkey.rs:139-316) and a
// .wtns (witness.rs:51-91), so that sessions and file -> proof runs have a real file of any size to work on: the shipped fixtures stop
// at 213 constraints.  n_public = 1, num_constraints = m - 2, n_vars = m = domain size; constraint j:
//     (ca_j * w[j+1]) * (cb_j * w[sb_j]) = w[j+2],   sb_j = 1 + (7 j + 3) mod (j + 1)  (<= j + 1: the witness is computed forward).
// CRS from seeded toxic waste (tau, alpha, beta, gamma, delta): polynomial evaluations on the host (field arithmetic through the
// ABI's cg_fr_op, on a few threads), the five point tables by fixed-base batch multiplication on the GPU (cg_bases_from_scalars).
// Conventions the prover relies on (groth16.rs:141-204): section 4 carries the rows A[nc + i] = w_i for i <= n_public, and
//     h_query[i] = [ (tau^2m - 1) g w^i / (2 m delta (tau - g w^i)) ]_1,   g = w_2m:
// H = (AB - C)/Z is interpolated on the odd coset gH, where Z = g^m - 1 = -2, so the prover's h_i = (AB - C)(g w^i) needs no division.
static void parallel_for(size_t n, const std::function<void(size_t, size_t)>& fn) {
    const size_t T = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::thread::hardware_concurrency(), n / 4096 + 1}));
    if (T == 1) { fn(0, n); return; }
    std::vector<std::thread> th; std::vector<std::string> err(T);
    for (size_t t = 0; t < T; t++) th.emplace_back([&, t] { try { fn(n * t / T, n * (t + 1) / T); } catch (const std::exception& e) { err[t] = e.what(); } });
    for (auto& x : th) x.join();
    for (auto& e : err) if (!e.empty()) throw std::runtime_error(e);
}
static void batch_inverse(const Curve& c, std::vector<Fr>& v) {           // Montgomery's trick per slice; no zero elements
    parallel_for(v.size(), [&](size_t lo, size_t hi) {
        if (hi <= lo) return;
        std::vector<Fr> pre(hi - lo);
        Fr acc = fr_from_u64(c, 1);
        for (size_t i = lo; i < hi; i++) { pre[i - lo] = acc; acc = fr_mul(c, acc, v[i]); }
        Fr inv = fr_inv(c, acc);
        for (size_t i = hi; i-- > lo;) { const Fr t = fr_mul(c, inv, pre[i - lo]); inv = fr_mul(c, inv, v[i]); v[i] = t; }
    });
}
struct SplitMix { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); } };
static Fr random_nonzero_fr(const Curve& c, SplitMix& g) {
    for (;;) {
        Fr raw; for (int i = 0; i < 4; i++) raw.v[i] = g.next();
        raw.v[3] &= c.id == CG_BN254 ? 0x3fffffffffffffffull : 0x7fffffffffffffffull;
        bool lt = false, gt = false;
        for (int i = 3; i >= 0 && !lt && !gt; i--) { if (raw.v[i] < MOD_R[c.id][i]) lt = true; else if (raw.v[i] > MOD_R[c.id][i]) gt = true; }
        if (!lt || !(raw.v[0] | raw.v[1] | raw.v[2] | raw.v[3])) continue;
        Fr m; CG(cg_fr_from_canonical(c.id, raw.v, m.v, 1));
        return m;
    }
}
struct SectionWriter {     // sections are streamed: a 2^22-constraint zkey is 2 GB
    FILE* f;
    SectionWriter(const std::string& path, const char* magic, uint32_t version, uint32_t nsec) {
        f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        put(magic, 4); u32(version); u32(nsec);
    }
    ~SectionWriter() { if (f) fclose(f); }
    void put(const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) throw std::runtime_error("short write"); }
    void u32(uint32_t x) { put(&x, 4); }
    void u64(uint64_t x) { put(&x, 8); }
    void begin(uint32_t id, uint64_t bytes) { u32(id); u64(bytes); }
    void close() { if (f && fclose(f) != 0) { f = nullptr; throw std::runtime_error("close failed"); } f = nullptr; }
};
static void synth_circuit(int device, int curve_id, int log_m, uint64_t seed, const std::string& zkey_path, const std::string& wtns_path) {
    if (log_m < 2 || log_m > 26) throw std::runtime_error("log_m out of range");
    const Curve c{curve_id};
    const size_t m = (size_t)1 << log_m, nc = m - 2, n_pub = 1, n_vars = m, n_inp = n_pub + 1;
    SplitMix rng{seed * 0x2545f4914f6cdd1dull + 0x1234567};
    // circuit + witness (a serial chain by construction)
    std::vector<Fr> ca(nc), cb(nc), w(n_vars);
    std::vector<uint32_t> sb(nc);
    for (size_t j = 0; j < nc; j++) { ca[j] = random_nonzero_fr(c, rng); cb[j] = random_nonzero_fr(c, rng); sb[j] = (uint32_t)(1 + (7 * j + 3) % (j + 1)); }
    w[0] = fr_from_u64(c, 1); w[1] = random_nonzero_fr(c, rng);
    for (size_t j = 0; j < nc; j++) w[j + 2] = fr_mul(c, fr_mul(c, ca[j], w[j + 1]), fr_mul(c, cb[j], w[sb[j]]));
    // toxic waste, Lagrange values of the domain at tau
    const Fr tau = random_nonzero_fr(c, rng), alpha = random_nonzero_fr(c, rng), beta = random_nonzero_fr(c, rng), gamma = random_nonzero_fr(c, rng), delta = random_nonzero_fr(c, rng);
    const Domain dom = groth16_domain(c, (size_t)log_m, nc, n_inp);
    const Fr one = fr_from_u64(c, 1);
    Fr tau_m = tau; for (int i = 0; i < log_m; i++) tau_m = fr_mul(c, tau_m, tau_m);
    std::vector<Fr> wpow(m), lag(m), hexp(m);
    {   // w^j by slices: each slice starts from w^lo (square-and-multiply) and runs a product chain
        parallel_for(m, [&](size_t lo, size_t hi) {
            if (hi <= lo) return;
            uint64_t e[1] = {lo};
            Fr acc = fr_pow(c, dom.omega, e, 1);
            for (size_t j = lo; j < hi; j++) { wpow[j] = acc; acc = fr_mul(c, acc, dom.omega); }
        });
    }
    parallel_for(m, [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) { lag[j] = fr_sub(c, tau, wpow[j]); hexp[j] = fr_sub(c, tau, fr_mul(c, dom.coset_g, wpow[j])); } });
    batch_inverse(c, lag); batch_inverse(c, hexp);
    const Fr zt_over_m = fr_mul(c, fr_sub(c, tau_m, one), fr_inv(c, fr_from_u64(c, (uint64_t)m)));
    const Fr hfac = fr_mul(c, fr_mul(c, fr_sub(c, fr_mul(c, tau_m, tau_m), one), fr_inv(c, fr_mul(c, fr_from_u64(c, 2 * (uint64_t)m), delta))), dom.coset_g);
    parallel_for(m, [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) { lag[j] = fr_mul(c, fr_mul(c, zt_over_m, wpow[j]), lag[j]); hexp[j] = fr_mul(c, fr_mul(c, hfac, wpow[j]), hexp[j]); } });
    { std::vector<Fr>().swap(wpow); }
    // u_i = sum_j A[j][i] L_j(tau), v_i, and the C column (C[j][j+2] = 1)
    const Fr zero = fr_sub(c, one, one);
    std::vector<Fr> u(n_vars, zero), v(n_vars, zero), lic(n_vars);
    parallel_for(nc, [&](size_t lo, size_t hi) { for (size_t j = lo; j < hi; j++) u[j + 1] = fr_mul(c, ca[j], lag[j]); });
    for (size_t j = 0; j < nc; j++) v[sb[j]] = fr_add(c, v[sb[j]], fr_mul(c, cb[j], lag[j]));     // colliding targets: serial
    for (size_t i = 0; i < n_inp; i++) u[i] = fr_add(c, u[i], lag[nc + i]);
    const Fr ginv = fr_inv(c, gamma), dinv = fr_inv(c, delta);
    parallel_for(n_vars, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            Fr t = fr_add(c, fr_mul(c, beta, u[i]), fr_mul(c, alpha, v[i]));
            if (i >= 2) t = fr_add(c, t, lag[i - 2]);                                         // C column: w_i = L_(i-2) for i >= 2
            lic[i] = fr_mul(c, t, i <= n_pub ? ginv : dinv);
        }
    });
    // group elements on the GPU
    CtxGuard cg; if (cg_ctx_create(device, &cg.ctx)) die("cg_ctx_create");
    cg_ctx* ctx = cg.ctx;
    auto table = [&](const std::vector<Fr>& sc, int group) {
        DevBufGuard d{ctx, nullptr};
        CG(cg_dev_alloc(ctx, sc.size() * 32, &d.p));
        CG(cg_dev_upload(ctx, d.p, sc.data(), sc.size() * 32));
        cg_bases* b = nullptr; CG(cg_bases_from_scalars(ctx, c.id, group, d.p, sc.size(), &b));
        Bytes out(sc.size() * c.aff(group));
        const int rc = cg_bases_download(ctx, b, 0, sc.size(), out.data());
        cg_bases_release(b);
        if (rc) die("cg_bases_download");
        return out;
    };
    auto g1 = [&](const Fr& k) { return pt_to_affine(c, pt_mul(c, pt_generator(c, CG_G1), k)); };
    auto g2 = [&](const Fr& k) { return pt_to_affine(c, pt_mul(c, pt_generator(c, CG_G2), k)); };
    const uint32_t ncoef = (uint32_t)(2 * nc + n_inp);
    SectionWriter zk(zkey_path, "zkey", 1, 9);
    zk.begin(1, 4); zk.u32(1);                                                               // protocol: groth16
    zk.begin(2, 4 + c.fq() + 4 + 32 + 12 + 3 * c.aff(CG_G1) + 3 * c.aff(CG_G2));
    zk.u32((uint32_t)c.fq()); zk.put(MOD_Q[c.id], c.fq()); zk.u32(32); zk.put(MOD_R[c.id], 32);
    zk.u32((uint32_t)n_vars); zk.u32((uint32_t)n_pub); zk.u32((uint32_t)m);
    { Bytes a1 = g1(alpha), b1 = g1(beta), b2 = g2(beta), c2 = g2(gamma), d1 = g1(delta), d2 = g2(delta);
      zk.put(a1.data(), a1.size()); zk.put(b1.data(), b1.size()); zk.put(b2.data(), b2.size()); zk.put(c2.data(), c2.size()); zk.put(d1.data(), d1.size()); zk.put(d2.data(), d2.size()); }
    Bytes l_all = table(lic, CG_G1);
    { std::vector<Fr>().swap(lic); }
    zk.begin(3, n_inp * c.aff(CG_G1)); zk.put(l_all.data(), n_inp * c.aff(CG_G1));
    zk.begin(4, 4 + (uint64_t)ncoef * 44); zk.u32(ncoef);
    {   // value on disk = v * R^2: the Montgomery form of the Montgomery form (traits.rs:57-67 reduces once)
        auto rec = [&](uint32_t mat, uint32_t row, uint32_t sig, const Fr& val) { Fr d; CG(cg_fr_from_canonical(c.id, val.v, d.v, 1)); zk.u32(mat); zk.u32(row); zk.u32(sig); zk.put(d.v, 32); };
        for (size_t j = 0; j < nc; j++) { rec(0, (uint32_t)j, (uint32_t)(j + 1), ca[j]); rec(1, (uint32_t)j, sb[j], cb[j]); }
        for (size_t i = 0; i < n_inp; i++) rec(0, (uint32_t)(nc + i), (uint32_t)i, one);
    }
    { Bytes t = table(u, CG_G1); zk.begin(5, t.size()); zk.put(t.data(), t.size()); }
    { Bytes t = table(v, CG_G1); zk.begin(6, t.size()); zk.put(t.data(), t.size()); }
    { Bytes t = table(v, CG_G2); zk.begin(7, t.size()); zk.put(t.data(), t.size()); }
    zk.begin(8, (n_vars - n_inp) * c.aff(CG_G1)); zk.put(l_all.data() + n_inp * c.aff(CG_G1), (n_vars - n_inp) * c.aff(CG_G1));
    { Bytes t = table(hexp, CG_G1); zk.begin(9, t.size()); zk.put(t.data(), t.size()); }
    zk.close();
    SectionWriter wt(wtns_path, "wtns", 2, 2);
    wt.begin(1, 4 + 32 + 4); wt.u32(32); wt.put(MOD_R[c.id], 32); wt.u32((uint32_t)n_vars);
    wt.begin(2, (uint64_t)n_vars * 32);
    { std::vector<Fr> can(n_vars); CG(cg_fr_to_canonical(c.id, w.data(), can.data(), n_vars)); wt.put(can.data(), n_vars * 32); }
    wt.close();
}

}  // namespace cgh

// ==================================================================================================== C entry points (tests / tools)
static thread_local std::string g_host_err;
// first real failure among the parties (the others only report that somebody else died)
template <class Errs> static bool report_party_errors(const Errs& errs, int n) {
    int pick = -1;
    for (int i = 0; i < n; i++) if (!errs[i].empty() && (pick < 0 || (errs[pick] == "another party failed" && errs[i] != "another party failed"))) pick = i;
    if (pick < 0) return false;
    g_host_err = "party " + std::to_string(pick) + ": " + errs[pick];
    return true;
}
extern "C" {

const char* cgh_last_error(void) { return g_host_err.c_str(); }

// info: n_vars, n_public, domain_size, pow, num_constraints, nnzA, nnzB
int32_t cgh_zkey_info(int32_t curve, const char* path, size_t* info) {
    try {
        cgh::ZKey z = cgh::read_zkey(curve, path, true);
        info[0] = z.n_vars; info[1] = z.n_public; info[2] = z.domain_size; info[3] = z.pow; info[4] = z.num_constraints; info[5] = z.col[0].size(); info[6] = z.col[1].size();
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// zkey -> device with the parser's point validation done on the GPU (traits.rs:107-155); 0 = every point valid.  seconds[0] = file
// read + section decode (host), seconds[1] = upload + validation (device)
int32_t cgh_zkey_validate(int32_t device, int32_t curve, const char* path, double* seconds) {
    cg_ctx* ctx = nullptr;
    try {
        using namespace cgh;
        auto t0 = std::chrono::steady_clock::now();
        ZKey z = read_zkey(curve, path);
        auto t1 = std::chrono::steady_clock::now();
        if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
        std::vector<Fr> pub(z.n_public + 1);
        DeviceZKey dz = upload_zkey(ctx, z, pub, 1);
        release_zkey(ctx, dz);
        cg_ctx_destroy(ctx);
        auto t2 = std::chrono::steady_clock::now();
        if (seconds) { seconds[0] = std::chrono::duration<double>(t1 - t0).count(); seconds[1] = std::chrono::duration<double>(t2 - t1).count(); }
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); if (ctx) cg_ctx_destroy(ctx); return 1; }
}
static int32_t copy_out(const std::string& js, char* out, size_t cap) {
    if (js.size() + 1 > cap) { g_host_err = "buffer too small"; return 1; }
    memcpy(out, js.c_str(), js.size() + 1);
    return 0;
}
// Groth16Proof <-> JSON (proof.rs:8-29); proof = A || B || C packed affine Montgomery as returned by cgh_prove_*
int32_t cgh_proof_to_json(int32_t curve, const uint64_t* proof, char* out, size_t cap) {
    try { return copy_out(cgh::proof_to_json(cgh::Curve{curve}, (const uint8_t*)proof), out, cap); }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_proof_from_json(int32_t curve, const char* json, uint64_t* out_proof) {
    try { cgh::proof_from_json(cgh::Curve{curve}, json, (uint8_t*)out_proof); return 0; }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_plonk_proof_to_json(int32_t curve, const uint64_t* commits, const uint64_t* evals, char* out, size_t cap) {
    try { return copy_out(cgh::plonk_proof_to_json(cgh::Curve{curve}, (const uint8_t*)commits, (const cgh::Fr*)evals), out, cap); }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_plonk_proof_from_json(int32_t curve, const char* json, uint64_t* out_commits, uint64_t* out_evals) {
    try { cgh::plonk_proof_from_json(cgh::Curve{curve}, json, (uint8_t*)out_commits, (cgh::Fr*)out_evals); return 0; }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// .shared witness files; protocol: 0 = REP3 (components a, b), 1 = Shamir (a only).  All values Montgomery on this side of the call.
int32_t cgh_shared_witness_write(int32_t curve, const char* path, int32_t protocol, const uint64_t* pub, size_t n_pub, const uint64_t* a, const uint64_t* b, size_t n) {
    try {
        using namespace cgh;
        std::vector<Fr> p((const Fr*)pub, (const Fr*)pub + n_pub), va((const Fr*)a, (const Fr*)a + n), vb;
        if (protocol == 0) vb.assign((const Fr*)b, (const Fr*)b + n);
        write_shared_witness(Curve{curve}, path, p, va, protocol == 0 ? &vb : nullptr);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// sizes[0] = n_pub, sizes[1] = n; with out buffers NULL only the sizes are returned
int32_t cgh_shared_witness_read(int32_t curve, const char* path, int32_t protocol, size_t* sizes, uint64_t* pub, uint64_t* a, uint64_t* b) {
    try {
        using namespace cgh;
        std::vector<Fr> p, va, vb;
        read_shared_witness(Curve{curve}, path, protocol == 0, p, va, vb);
        sizes[0] = p.size(); sizes[1] = va.size();
        if (pub) memcpy(pub, p.data(), p.size() * 32);
        if (a) memcpy(a, va.data(), va.size() * 32);
        if (b && protocol == 0) memcpy(b, vb.data(), vb.size() * 32);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// public.json (co-circom.rs:620-628): the public signals without the leading constant 1, as decimal strings; pub = n Montgomery elements
int32_t cgh_public_to_json(int32_t curve, const uint64_t* pub, size_t n, char* out, size_t cap) {
    try {
        std::string js = "[";
        for (size_t i = 0; i < n; i++) {
            uint64_t can[4]; if (cg_fr_to_canonical(curve, pub + 4 * i, can, 1)) cgh::die("cg_fr_to_canonical");
            js += (i ? ",\"" : "\"") + cgh::limbs_to_dec(can, 4) + "\"";
        }
        return copy_out(js + "]", out, cap);
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// ShamirHipProtocol x n (n threads, in-process any-to-any network), threshold t.  wit[i] = party i's Shamir shares of the private
// witness; streams[i] = party i's private randomness (consumed in the order the reference draws values).  out_proofs = n proofs.
int32_t cgh_prove_shamir(int32_t device, int32_t curve, const char* zkey_path, int32_t n, int32_t t, const uint64_t* pub_in, const uint64_t* const* wit,
                         const uint64_t* const* streams, size_t stream_len, size_t preprocess, uint64_t* out_proofs, uint64_t* out_h) {
    try {
        using namespace cgh;
        if (n < 3) throw std::runtime_error("Shamir protocol requires at least 3 parties");        // shamir/network.rs:75-77
        SecondContexts second(device, zkey_path, n);
        ZKey z = read_zkey(curve, zkey_path);
        const size_t n_aux = z.n_vars - z.n_public - 1;
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        cg_ctx* ctx0 = nullptr;
        if (cg_ctx_create(device, &ctx0)) die("cg_ctx_create");
        DeviceZKey dz = upload_zkey(ctx0, z, pub);
        second.ready();
        InProcShamirHub hub(n);
        const size_t psz = 8 * z.curve.fq();
        std::vector<std::string> errs(n);
        std::vector<std::thread> th;
        for (int i = 0; i < n; i++) th.emplace_back([&, i] {
            cg_ctx* ctx = nullptr;
            try {
                if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
                InProcShamirNet net(&hub, i);
                HipDriver driver(ctx, z.curve, Mode::Shamir, nullptr);
                driver.use_second_context(second.take(i));
                driver.rng1 = (const Fr*)streams[i]; driver.rng_len = stream_len;
                driver.shamir_init(&net, t);
                const auto ta = std::chrono::steady_clock::now();
                driver.preprocess(preprocess);                                            // 0 = the reference's lazy batches of 1024
                const auto tb = std::chrono::steady_clock::now();
                ShareVec w = driver.upload_vec((const Fr*)wit[i], nullptr, n_aux);
                CoGroth16 prover(driver);
                ShareVec h;
                Proof p = prover.prove(dz, pub, w, nullptr, &h);
                if (i == 0 && getenv("CGH_TIMING"))
                    fprintf(stderr, "cgh_prove_shamir party 0: preprocess %.1f ms, prove %.1f ms\n", std::chrono::duration<double, std::milli>(tb - ta).count(),
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count());
                store_proof(p, (uint8_t*)out_proofs + i * psz);
                if (out_h && i == 0) CG(cg_dev_download(ctx, out_h, h.c[0], h.n * 32));
                driver.free_vec(h); driver.free_vec(w); driver.shutdown();
                cg_ctx_destroy(ctx);
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); if (ctx) cg_ctx_destroy(ctx); }
        });
        for (auto& x : th) x.join();
        release_zkey(ctx0, dz);
        cg_ctx_destroy(ctx0);
        if (report_party_errors(errs, n)) return 1;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// info: n_vars, n_public, domain_size, power, n_additions, n_constraints
int32_t cgh_plonk_zkey_info(int32_t curve, const char* path, size_t* info) {
    try {
        cgh::PlonkZKey z = cgh::read_plonk_zkey(curve, path);
        info[0] = z.n_vars; info[1] = z.n_public; info[2] = z.domain_size; info[3] = z.power; info[4] = z.n_additions; info[5] = z.n_constraints;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// ---- co-plonk entry points -------------------------------------------------------------------------------------------------------
namespace {
struct PlonkOut { uint64_t* commits; uint64_t* challenges; uint64_t* evals; uint64_t* t_polys; uint64_t* poly_z; };
// runs rounds 1..upto on `driver` and stores what has been computed (slot layout of cgh_plonk_prove_plain)
void plonk_run(cgh::HipDriver& driver, const cgh::PlonkZKey& z, const cg_bases* tau, const std::vector<cgh::Fr>& pub, const cgh::ShareVec& wit, const cgh::FieldShare* b, int upto, const PlonkOut& o) {
    using namespace cgh;
    const Curve& c = z.curve; const size_t psz = c.aff(CG_G1);
    auto put = [&](int slot, const Point& p) { if (!o.commits) return; Bytes a = pt_to_affine(c, p); memcpy((uint8_t*)o.commits + slot * psz, a.data(), psz); };
    auto putf = [&](uint64_t* dst, int slot, const Fr& f) { if (dst) memcpy(dst + 4 * slot, f.v, 32); };
    CoPlonk pk(driver, z, tau, pub, b);
    pk.round1(wit);
    for (int k = 0; k < 3; k++) put(k, pk.commit[k]);
    if (upto >= 2) {
        pk.round2(); put(3, pk.commit_z); putf(o.challenges, 0, pk.beta); putf(o.challenges, 1, pk.gamma);
        if (o.poly_z) CG(cg_dev_download(driver.ctx, o.poly_z, pk.poly_z.c[0], pk.poly_z.n * 32));
    }
    if (upto >= 3) {
        pk.round3(); for (int k = 0; k < 3; k++) put(4 + k, pk.commit_t[k]); putf(o.challenges, 2, pk.alpha);
        if (o.t_polys) { size_t off = 0; for (int k = 0; k < 3; k++) { CG(cg_dev_download(driver.ctx, o.t_polys + off * 4, pk.tpart[k].c[0], pk.tpart[k].n * 32)); off += pk.tpart[k].n; } }
    }
    if (upto >= 4) {
        pk.round4(); putf(o.challenges, 3, pk.xi);
        const Fr ev[6] = {pk.ev_a, pk.ev_b, pk.ev_c, pk.ev_s1, pk.ev_s2, pk.ev_zw};
        for (int i = 0; i < 6; i++) putf(o.evals, i, ev[i]);
    }
    if (upto >= 5) { pk.round5(); putf(o.challenges, 4, pk.v[0]); put(7, pk.commit_wxi); put(8, pk.commit_wxiw); }
}
}  // namespace
// PlainHipDriver through rounds 1..upto (<= 5).  full_witness = n_vars - n_additions Montgomery elements (Groth16-style, leading one);
// blind = 11 Fr; commits = 9 packed G1 (a, b, c, z, t1, t2, t3, wxi, wxiw; zero = not reached), challenges = beta, gamma, alpha, xi, v;
// evals = a, b, c, s1, s2, zw; optional: t_polys = t1 (n+1) | t2 (n+1) | t3 (n+6), poly_z (n+3)
int32_t cgh_plonk_prove_plain(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* full_witness, const uint64_t* blind, int32_t upto,
                              uint64_t* commits, uint64_t* challenges, uint64_t* evals, uint64_t* t_polys, uint64_t* poly_z) {
    cg_ctx* ctx = nullptr;
    try {
        using namespace cgh;
        PlonkZKey z = read_plonk_zkey(curve, zkey_path);
        if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
        const Curve& c = z.curve;
        if (commits) memset(commits, 0, 9 * c.aff(CG_G1)); if (challenges) memset(challenges, 0, 5 * 32); if (evals) memset(evals, 0, 6 * 32);
        cg_bases* tau = nullptr; CG(cg_bases_register(ctx, c.id, CG_G1, z.p_tau.data(), z.domain_size + 6, c.aff(CG_G1), -1, &tau));
        if (validate_by_default()) { try { validate_bases(ctx, tau, "p_tau"); } catch (...) { cg_bases_release(tau); throw; } }   // the zkey parser's per-point checks
        const Fr* w = (const Fr*)full_witness;
        std::vector<Fr> pub(w, w + z.n_public + 1);
        {
            HipDriver driver(ctx, c, Mode::Plain, nullptr);
            ShareVec wit = driver.upload_vec(w + z.n_public + 1, nullptr, z.n_vars - z.n_additions - z.n_public - 1);
            FieldShare b[11]; for (int i = 0; i < 11; i++) { memcpy(b[i].c[0].v, blind + 4 * i, 32); b[i].c[1] = b[i].c[0]; }
            plonk_run(driver, z, tau, pub, wit, b, upto, PlonkOut{commits, challenges, evals, t_polys, poly_z});
            driver.free_vec(wit);
        }
        cg_bases_release(tau); cg_ctx_destroy(ctx);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); if (ctx) cg_ctx_destroy(ctx); return 1; }
}
// Keccak256 transcript hook (tests): kinds[i] 0 = scalar (Fr), 1 = packed G1 point
int32_t cgh_plonk_transcript(int32_t curve, const int32_t* kinds, const uint64_t* const* payloads, int32_t n_items, uint64_t* out_challenge) {
    try {
        using namespace cgh;
        Curve c{curve};
        PlonkTranscript t(c);
        for (int i = 0; i < n_items; i++) { if (kinds[i] == 0) { Fr s; memcpy(s.v, payloads[i], 32); t.add_scalar(s); } else t.add_point((const uint8_t*)payloads[i]); }
        Fr r = t.get_challenge(); memcpy(out_challenge, r.v, 32);
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// ShamirHipProtocol x n (threshold t) through rounds 1..upto.  wit[i] / blind[i] = party i's Shamir shares of the private witness and of
// b_1..b_11; streams[i] = party i's private randomness.  Outputs as for cgh_plonk_prove_rep3, n parties.
int32_t cgh_plonk_prove_shamir(int32_t device, int32_t curve, const char* zkey_path, int32_t n, int32_t t, const uint64_t* pub_in, const uint64_t* const* wit,
                               const uint64_t* const* blind, const uint64_t* const* streams, size_t stream_len, int32_t upto,
                               uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges) {
    try {
        using namespace cgh;
        if (n < 3) throw std::runtime_error("Shamir protocol requires at least 3 parties");
        PlonkZKey z = read_plonk_zkey(curve, zkey_path);
        const Curve c = z.curve;
        const size_t n_priv = z.n_vars - z.n_additions - z.n_public - 1, psz = c.aff(CG_G1);
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        memset(out_commits, 0, (size_t)n * 9 * psz); if (out_evals) memset(out_evals, 0, (size_t)n * 6 * 32); if (out_challenges) memset(out_challenges, 0, (size_t)n * 5 * 32);
        cg_ctx* ctx0 = nullptr;
        if (cg_ctx_create(device, &ctx0)) die("cg_ctx_create");
        cg_bases* tau = nullptr; CG(cg_bases_register(ctx0, c.id, CG_G1, z.p_tau.data(), z.domain_size + 6, psz, -1, &tau));
        if (validate_by_default()) { try { validate_bases(ctx0, tau, "p_tau"); } catch (...) { cg_bases_release(tau); throw; } }   // the zkey parser's per-point checks
        InProcShamirHub hub(n);
        std::vector<std::string> errs(n);
        std::vector<std::thread> th;
        for (int i = 0; i < n; i++) th.emplace_back([&, i] {
            cg_ctx* ctx = nullptr;
            try {
                if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
                InProcShamirNet net(&hub, i);
                {
                    HipDriver driver(ctx, c, Mode::Shamir, nullptr);
                    driver.rng1 = (const Fr*)streams[i]; driver.rng_len = stream_len;
                    driver.shamir_init(&net, t);
                    ShareVec w = driver.upload_vec((const Fr*)wit[i], nullptr, n_priv);
                    FieldShare b[11]; for (int q = 0; q < 11; q++) { memcpy(b[q].c[0].v, blind[i] + 4 * q, 32); b[q].c[1] = b[q].c[0]; }
                    plonk_run(driver, z, tau, pub, w, b, upto, PlonkOut{(uint64_t*)((uint8_t*)out_commits + (size_t)i * 9 * psz), out_challenges ? out_challenges + i * 20 : nullptr,
                                                                        out_evals ? out_evals + i * 24 : nullptr, nullptr, nullptr});
                    driver.free_vec(w);
                }
                cg_ctx_destroy(ctx);
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); if (ctx) cg_ctx_destroy(ctx); }
        });
        for (auto& x : th) x.join();
        cg_bases_release(tau);
        cg_ctx_destroy(ctx0);
        if (report_party_errors(errs, n)) return 1;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// Rep3HipProtocol x 3 (three threads, in-process network) through rounds 1..upto.  blind_a[i] / blind_b[i] = party i's (a, b) shares of
// b_1..b_11; streams[i] = S_i (party i: rng1 = S_i, rng2 = S_{i-1}; rounds 2 and 3 consume masks and random shares).
// out_commits = 3 parties x 9 packed G1, out_evals = 3 x 6 Fr, out_challenges = 3 x 5 Fr (every party must report the same values)
int32_t cgh_plonk_prove_rep3(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                             const uint64_t* const* blind_a, const uint64_t* const* blind_b, const uint64_t* const* streams, size_t stream_len, int32_t upto,
                             uint64_t* out_commits, uint64_t* out_evals, uint64_t* out_challenges) {
    try {
        using namespace cgh;
        PlonkZKey z = read_plonk_zkey(curve, zkey_path);
        const Curve c = z.curve;
        const size_t n_priv = z.n_vars - z.n_additions - z.n_public - 1, psz = c.aff(CG_G1);
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        memset(out_commits, 0, 3 * 9 * psz); if (out_evals) memset(out_evals, 0, 3 * 6 * 32); if (out_challenges) memset(out_challenges, 0, 3 * 5 * 32);
        cg_ctx* ctx0 = nullptr;
        if (cg_ctx_create(device, &ctx0)) die("cg_ctx_create");
        cg_bases* tau = nullptr; CG(cg_bases_register(ctx0, c.id, CG_G1, z.p_tau.data(), z.domain_size + 6, psz, -1, &tau));
        if (validate_by_default()) { try { validate_bases(ctx0, tau, "p_tau"); } catch (...) { cg_bases_release(tau); throw; } }   // the zkey parser's per-point checks
        InProcHub hub;
        std::string errs[3];
        std::vector<std::thread> th;
        for (int i = 0; i < 3; i++) th.emplace_back([&, i] {
            cg_ctx* ctx = nullptr;
            try {
                if (cg_ctx_create(device, &ctx)) die("cg_ctx_create");
                InProcNetwork net(&hub, i);
                {
                    HipDriver driver(ctx, c, Mode::Rep3, &net);
                    if (streams) { driver.rng1 = (const Fr*)streams[i]; driver.rng2 = (const Fr*)streams[(i + 2) % 3]; driver.rng_len = stream_len; }
                    ShareVec wit = driver.upload_vec((const Fr*)wit_a[i], (const Fr*)wit_b[i], n_priv);
                    FieldShare b[11]; for (int t = 0; t < 11; t++) { memcpy(b[t].c[0].v, blind_a[i] + 4 * t, 32); memcpy(b[t].c[1].v, blind_b[i] + 4 * t, 32); }
                    plonk_run(driver, z, tau, pub, wit, b, upto, PlonkOut{(uint64_t*)((uint8_t*)out_commits + (size_t)i * 9 * psz), out_challenges ? out_challenges + i * 20 : nullptr,
                                                                          out_evals ? out_evals + i * 24 : nullptr, nullptr, nullptr});
                    driver.free_vec(wit);
                }
                cg_ctx_destroy(ctx);
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); if (ctx) cg_ctx_destroy(ctx); }
        });
        for (auto& t : th) t.join();
        cg_bases_release(tau);
        cg_ctx_destroy(ctx0);
        if (report_party_errors(errs, 3)) return 1;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
int32_t cgh_read_wtns(int32_t curve, const char* path, uint64_t* out, size_t cap, size_t* n) {
    try {
        auto w = cgh::read_wtns(curve, path);
        *n = w.size();
        if (out) { if (w.size() > cap) { g_host_err = "buffer too small"; return 1; } memcpy(out, w.data(), w.size() * 32); }
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// PlainHipDriver: full_witness = n_vars Montgomery elements; proof = A || B || C packed affine
int32_t cgh_prove_plain(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* full_witness, const uint64_t* r, const uint64_t* s, uint64_t* out_proof, uint64_t* out_h) {
    try {
        using namespace cgh;
        const bool timing = getenv("CGH_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t0 = now();
        SecondContexts second(device, zkey_path, 1);
        ZKey z = read_zkey(curve, zkey_path);
        const auto t1 = now();
        CtxGuard cg;                                           // destroyed last: everything below lives on it
        if (cg_ctx_create(device, &cg.ctx)) die("cg_ctx_create");
        cg_ctx* ctx = cg.ctx;
        const Fr* w = (const Fr*)full_witness;
        std::vector<Fr> pub(w, w + z.n_public + 1);
        DeviceZKeyGuard dzg(ctx, upload_zkey(ctx, z, pub));
        const auto t2 = now();
        std::chrono::steady_clock::time_point t3;
        {
            HipDriver driver(ctx, z.curve, Mode::Plain, nullptr);
            driver.use_second_context(second.take(0));
            VecGuard wit(driver, driver.upload_vec(w + z.n_public + 1, nullptr, z.n_vars - z.n_public - 1)), h(driver);
            FieldShare rs[2]; memcpy(rs[0].c[0].v, r, 32); rs[0].c[1] = rs[0].c[0]; memcpy(rs[1].c[0].v, s, 32); rs[1].c[1] = rs[1].c[0];
            CoGroth16 prover(driver);
            Proof p = prover.prove(dzg.dz, pub, wit.v, rs, &h.v);
            t3 = now();
            store_proof(p, (uint8_t*)out_proof);
            if (out_h) CG(cg_dev_download(ctx, out_h, h.v.c[0], h.v.n * 32));
        }
        if (timing) fprintf(stderr, "cgh_prove_plain: read+decode zkey %.1f ms, context + upload%s %.1f ms, prove %.1f ms, teardown %.1f ms\n", ms(t0, t1),
                            validate_by_default() ? " + point validation" : "", ms(t1, t2), ms(t2, t3), ms(t3, now()));
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}
// Rep3HipProtocol x 3 on three threads over the in-process network; streams[i] = S_i (party i: rng1 = S_i, rng2 = S_{i-1})
int32_t cgh_prove_rep3(int32_t device, int32_t curve, const char* zkey_path, const uint64_t* pub_in, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                       const uint64_t* const* streams, size_t stream_len, uint64_t* out_proofs, uint64_t* out_h) {
    try {
        using namespace cgh;
        SecondContexts second(device, zkey_path, 3);
        ZKey z = read_zkey(curve, zkey_path);
        const size_t n_aux = z.n_vars - z.n_public - 1;
        std::vector<Fr> pub((const Fr*)pub_in, (const Fr*)pub_in + z.n_public + 1);
        CtxGuard cg0;
        if (cg_ctx_create(device, &cg0.ctx)) die("cg_ctx_create");
        DeviceZKeyGuard dzg(cg0.ctx, upload_zkey(cg0.ctx, z, pub));   // one device-resident zkey shared by the three co-located parties
        const DeviceZKey& dz = dzg.dz;
        second.ready();
        InProcHub hub;
        const size_t psz = 8 * z.curve.fq();
        std::string errs[3];
        std::vector<std::thread> th;
        for (int i = 0; i < 3; i++) th.emplace_back([&, i] {
            try {
                CtxGuard cg;
                if (cg_ctx_create(device, &cg.ctx)) die("cg_ctx_create");
                InProcNetwork net(&hub, i);
                HipDriver driver(cg.ctx, z.curve, Mode::Rep3, &net);
                driver.use_second_context(second.take(i));
                driver.rng1 = (const Fr*)streams[i]; driver.rng2 = (const Fr*)streams[(i + 2) % 3]; driver.rng_len = stream_len;
                VecGuard wit(driver, driver.upload_vec((const Fr*)wit_a[i], (const Fr*)wit_b[i], n_aux)), h(driver);
                CoGroth16 prover(driver);
                Proof p = prover.prove(dz, pub, wit.v, nullptr, &h.v);
                store_proof(p, (uint8_t*)out_proofs + i * psz);
                if (out_h && i == 0) { CG(cg_dev_download(cg.ctx, out_h, h.v.c[0], h.v.n * 32)); CG(cg_dev_download(cg.ctx, out_h + h.v.n * 4, h.v.c[1], h.v.n * 32)); }
            } catch (const std::exception& e) { errs[i] = e.what(); hub.abort(); }
        });
        for (auto& t : th) t.join();
        if (report_party_errors(errs, 3)) return 1;
        return 0;
    } catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}

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
int32_t cgh_set_zkey_validation(int32_t on) { cgh::g_validate_zkey.store(on ? 1 : 0); return 0; }
// synthetic satisfiable circuit of 2^log_m - 2 constraints with a valid CRS, written as .zkey + .wtns (bench / test tooling)
int32_t cgh_synth_circuit(int32_t device, int32_t curve, int32_t log_m, uint64_t seed, const char* zkey_path, const char* wtns_path) {
    try { cgh::synth_circuit(device, curve, log_m, seed, zkey_path, wtns_path); return 0; }
    catch (const std::exception& e) { g_host_err = e.what(); return 1; }
}

}  // extern "C"
