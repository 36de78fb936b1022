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
#pragma once
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
// std::vector storage whose resize() leaves new elements uninitialised: the 2 x 268 MB host mirrors of the preprocessed Shamir pairs are only
// placeholders while the values live on the device, and value-initialising them costs ~100 ms of page faults per party at 2^22
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
typedef std::vector<Fr, NoInitAlloc<Fr>> FrLazyVec;

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
// bases multiplied in every proof: an 8-bit window table each (cg_fixed_base_*: host arithmetic on 64-bit limbs, ~10 us per G1 product
// against ~60 us for a variable base).  nullptr table = no table: the variable-base product.
struct FixedTable {
    cg_fixed_base* t = nullptr;
    FixedTable() {}
    FixedTable(const Curve& c, const Point& p) { CG(cg_fixed_base_create(c.id, p.group, p.b.data(), &t)); }
    ~FixedTable() { if (t) cg_fixed_base_destroy(t); }
    FixedTable(const FixedTable&) = delete; FixedTable& operator=(const FixedTable&) = delete;
    FixedTable(FixedTable&& o) noexcept : t(o.t) { o.t = nullptr; }
    FixedTable& operator=(FixedTable&& o) noexcept { if (this != &o) { if (t) cg_fixed_base_destroy(t); t = o.t; o.t = nullptr; } return *this; }
};
static Point pt_mul_fixed(const Curve& c, const cg_fixed_base* tab, const Point& base, const Fr& k) {
    if (!tab) return pt_mul(c, base, k);
    Point r{Bytes(base.b.size()), base.group}; CG(cg_fixed_base_mul(tab, k.v, r.b.data())); return r;
}
// the generators' tables, one per (curve, group) and process, built at first use (the masking points G * rand of the in-library
// randomness sources and of degree_reduce_point)
static const cg_fixed_base* generator_table(const Curve& c, int g) {
    static std::mutex mu; static FixedTable tabs[2][2];
    std::lock_guard<std::mutex> l(mu);
    FixedTable& t = tabs[c.id == CG_BLS12_381 ? 1 : 0][g == CG_G1 ? 0 : 1];
    if (!t.t) t = FixedTable(c, pt_generator(c, g));
    return t.t;
}
static Point pt_mul_generator(const Curve& c, int g, const Fr& k) { return pt_mul_fixed(c, generator_table(c, g), pt_generator(c, g), k); }


// Process-wide options of the host library (cgh_set_option / cgh_get_option, include/cogroth16_host.h): the thresholds and layout switches a
// deployment — and the tests — may want to move.  They were environment variables until round 5; none changes a proof.
struct HostOptions {
    std::atomic<int64_t> v[CGH_OPT_COUNT];
    HostOptions() {
        for (auto& x : v) x.store(0);
        v[CGH_OPT_XCHG_ASYNC_MIN].store((int64_t)1 << 17);        // (2^19 until round 4; one REP3 party at 2^17 / 2^18: chunked exchange 4.9 / 7.6 ms against 5.5 / 8.9 as one message)
        v[CGH_OPT_DEVICE_MASKS_MIN].store((int64_t)1 << 11);      // (a host draw is ~70 ns per element: 0.6 ms per mul_vec at 2^13)
        v[CGH_OPT_XCHG_COPY_STREAM_MIN].store((int64_t)1 << 14);
        v[CGH_OPT_SECOND_CONTEXT_MIN_LOG].store(15);
        v[CGH_OPT_DISTRIBUTED_MAP].store(1);
        v[CGH_OPT_ONE_CONTEXT].store(0);
        v[CGH_OPT_SPLIT_FIRST_MSM_MIN].store(0);                 // measured SLOWER (2^22: 73.4 against 70.9 ms, 2^20 23.7 / 23.1, 2^24 280.5 / 278.2; profiles/r06_split_first_msm_ab.txt): off
    }
};
inline HostOptions g_host_options;
inline int64_t host_option(int id) { return g_host_options.v[id].load(std::memory_order_relaxed); }
// A/B knobs of the measurement scripts (scripts/party_knobs_ab.sh ...): environment variables in -DCG_DEBUG_KNOBS builds, compiled out of
// the release library (every call site then folds to its default)
inline const char* tune_env(const char* name) {
#ifdef CG_DEBUG_KNOBS
    return getenv(name);
#else
    (void)name; return nullptr;
#endif
}
// Planning knob of scripts/multi_device_emulation.py, compiled ONLY into -DCG_DEBUG_KNOBS builds (make -C host KNOBS=1 -> libcogroth16_host_knobs.so):
// CGH_EMULATE_DEVICE=d makes every device but d of an N-device session a no-op, so that one GPU times device d's share of the proof.  The
// proof is then WRONG — which is why the release library does not contain the knob (tests/test_abi_surface.py checks that it ignores the
// variable).  -1 = off.
inline int emulate_only_device() {
#ifdef CG_DEBUG_KNOBS
    const char* e = getenv("CGH_EMULATE_DEVICE"); return e ? atoi(e) : -1;
#else
    return -1;
#endif
}
// fn(lo, hi) over slices of [0, n) on a few threads; the first exception of a slice is re-thrown
static void parallel_for(size_t n, const std::function<void(size_t, size_t)>& fn) {
    const size_t T = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::thread::hardware_concurrency(), n / 4096 + 1}));
    if (T == 1) { fn(0, n); return; }
    std::vector<std::thread> th; std::vector<std::string> err(T);
    for (size_t t = 0; t < T; t++) th.emplace_back([&, t] { try { fn(n * t / T, n * (t + 1) / T); } catch (const std::exception& e) { err[t] = e.what(); } });
    for (auto& x : th) x.join();
    for (auto& e : err) if (!e.empty()) throw std::runtime_error(e);
}

}  // namespace cgh
