// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// CPU baseline workload for bench.py's `cpu_baseline` leg: ONE REP3 party's compute of `CoGroth16::prove`
// (`/root/reference/co-circom/co-groth16/src/groth16.rs:141-204,237-326`) on a synthetic circuit, i.e. exactly the work the
// GPU bench times: 2 constraint mat-vecs, 2 local share products, 12 NTTs (3 iNTT->coset->NTT pipelines x 2 components) and
// 10 MSMs (h, l, a, b1 in G1 and b2 in G2, x 2 components, components processed serially as in rep3.rs:942-943).
// Parallelism mirrors what the shipped reference binary gets from arkworks+rayon: window-parallel MSM
// (at most ceil(254/c) useful threads) and data-parallel FFT stages / pointwise loops.  Network rounds are excluded on both sides.
#pragma once
#include "groth16.hpp"
#include "rngs.hpp"
#include "formats.hpp"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <functional>

namespace orc {

// Persistent worker pool (rayon's role in the reference binary): the workers sleep on a condition variable between jobs, a job is
// a range cut into slices that the workers claim with an atomic counter.  (Spawning threads per butterfly stage, as the first
// version of this file did, cost more than the stage.)
class Pool {
    std::vector<std::thread> workers; std::mutex mu; std::condition_variable cv_job, cv_done;
    const std::function<void(size_t, size_t)>* fn = nullptr; size_t n = 0, slice = 1; std::atomic<size_t> next{0};
    size_t generation = 0; int busy = 0; bool stop = false;
    void work() { for (;;) { const size_t lo = next.fetch_add(slice); if (lo >= n) return; (*fn)(lo, std::min(n, lo + slice)); } }
public:
    explicit Pool(int threads) {
        for (int t = 1; t < threads; t++) workers.emplace_back([this] {
            size_t seen = 0;
            for (;;) {
                { std::unique_lock<std::mutex> l(mu); cv_job.wait(l, [&] { return stop || generation != seen; }); if (stop) return; seen = generation; }
                work();
                { std::lock_guard<std::mutex> l(mu); if (--busy == 0) cv_done.notify_one(); }
            }
        });
    }
    ~Pool() { { std::lock_guard<std::mutex> l(mu); stop = true; } cv_job.notify_all(); for (auto& w : workers) w.join(); }
    int size() const { return (int)workers.size() + 1; }
    void run(size_t count, size_t min_slice, const std::function<void(size_t, size_t)>& f) {
        if (count == 0) return;
        if (workers.empty() || count <= min_slice) { f(0, count); return; }
        { std::lock_guard<std::mutex> l(mu); fn = &f; n = count; slice = std::max(min_slice, (count + 8 * (size_t)size() - 1) / (8 * (size_t)size())); next = 0; busy = (int)workers.size(); generation++; }
        cv_job.notify_all();
        work();
        std::unique_lock<std::mutex> l(mu); cv_done.wait(l, [&] { return busy == 0; });
    }
};
static inline void parallel_for(size_t n, int threads, const std::function<void(size_t, size_t)>& fn) {      // one-off jobs (setup code)
    if (threads <= 1 || n < 1024) { fn(0, n); return; }
    Pool p(threads); p.run(n, 256, fn);
}

// Cache-blocked radix-2 transforms, same arithmetic and results as poly.hpp.  The 2^BLK-point sub-transforms (1 MiB of field
// elements) are independent once the long-range stages are done (forward: decimation in frequency, long strides first; inverse:
// decimation in time, short strides first), so each is run through all of its stages by one worker while it sits in that core's L2;
// only the log2(n) - BLK long-range stages sweep the whole array.  The bit reversal is split over the workers as well.
constexpr int NTT_BLK = 15;
template <class F>
static void bitrev_permute_mt(F* a, size_t n, Pool& pool) {
    const int lg = log2_exact(n);
    pool.run(n, 4096, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) { const size_t j = bitrev(i, lg); if (i < j) std::swap(a[i], a[j]); } });
}
template <class F>
static void ntt_forward_mt(F* a, size_t n, const std::vector<F>& tw, Pool& pool) {
    const size_t blk = std::min<size_t>(n, (size_t)1 << NTT_BLK);
    size_t half = n / 2, step = 1;
    for (; 2 * half > blk; half >>= 1, step <<= 1)
        pool.run(n / 2, 4096, [&](size_t lo, size_t hi) {
            for (size_t u = lo; u < hi; u++) {
                const size_t b = (u / half) * 2 * half, j = u % half;
                const F x = a[b + j], y = a[b + j + half];
                a[b + j] = x + y; a[b + j + half] = (x - y) * tw[j * step];
            }
        });
    pool.run(n / blk, 1, [&](size_t lo, size_t hi) {
        for (size_t q = lo; q < hi; q++) {
            F* s = a + q * blk;
            for (size_t h = half, st = step; h >= 1; h >>= 1, st <<= 1)
                for (size_t b = 0; b < blk; b += 2 * h)
                    for (size_t j = 0; j < h; j++) { const F x = s[b + j], y = s[b + j + h]; s[b + j] = x + y; s[b + j + h] = (x - y) * tw[j * st]; }
        }
    });
    bitrev_permute_mt(a, n, pool);
}
template <class F>
static void ntt_inverse_mt(F* a, size_t n, const std::vector<F>& twi, const F& ninv, Pool& pool) {
    bitrev_permute_mt(a, n, pool);
    const size_t blk = std::min<size_t>(n, (size_t)1 << NTT_BLK);
    pool.run(n / blk, 1, [&](size_t lo, size_t hi) {
        for (size_t q = lo; q < hi; q++) {
            F* s = a + q * blk;
            for (size_t h = 1, st = n / 2; 2 * h <= blk; h <<= 1, st >>= 1)
                for (size_t b = 0; b < blk; b += 2 * h)
                    for (size_t j = 0; j < h; j++) { const F x = s[b + j], y = s[b + j + h] * twi[j * st]; s[b + j] = x + y; s[b + j + h] = x - y; }
        }
    });
    for (size_t half = blk, step = n / (2 * blk); half < n; half <<= 1, step >>= 1)
        pool.run(n / 2, 4096, [&](size_t lo, size_t hi) {
            for (size_t u = lo; u < hi; u++) {
                const size_t b = (u / half) * 2 * half, j = u % half;
                const F x = a[b + j], y = a[b + j + half] * twi[j * step];
                a[b + j] = x + y; a[b + j + half] = x - y;
            }
        });
    pool.run(n, 4096, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) a[i] = a[i] * ninv; });
}

struct XorShift { uint64_t s; uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; } };
template <class F> static F rand_fp(XorShift& r) {
    F x; for (int i = 0; i < F::N; i++) x.v[i] = r.next();
    x.v[F::N - 1] &= ((uint64_t)1 << ((F::K.bits - 2) % 64)) - 1;   // < 2^(bits-2) < p: a valid reduced representative
    return x;
}

// One party-0 prove compute at domain size m = 2^log_m.  Returns wall seconds with `threads` workers; stage[0..3] = spmv + pointwise,
// ntt, msm_g1, msm_g2.  threads_b > 0: the timed region is run a second time with threads_b workers (stage_b, *total_b) on the same
// inputs; the MSM stages are window-parallel (at most `windows` workers are ever busy, ark-ec msm_bigint), so when both settings
// have at least that many workers the second run re-times only the other stages and takes the MSM times over (*msm_shared = 1).
template <class C>
static double bench_rep3_party(int log_m, int threads, uint64_t seed, double* stage, int threads_b = 0, double* stage_b = nullptr, double* total_b = nullptr, int* msm_shared = nullptr) {
    typedef typename C::Fr Fr; typedef typename C::G1 G1; typedef typename C::G2 G2;
    C::init();
    const size_t m = (size_t)1 << log_m, nc = m - 2, n_aux = m - 2, n_inputs = 2;
    XorShift rng{seed | 1};
    // ---- untimed setup: bases = consecutive multiples of the generators (slices in parallel), CSR matrices, shares
    const int setup_threads = std::max(threads, threads_b);
    auto make_table = [&](auto gen_affine, auto jac_tag, size_t n, uint64_t first) {
        typedef decltype(jac_tag) J;
        std::vector<typename J::Affine> out(n);
        parallel_for(n, setup_threads, [&](size_t lo, size_t hi) {
            if (hi <= lo) return;
            uint64_t k[1] = {first + lo};
            J acc = J::from_affine(gen_affine).mul(k, 1);
            std::vector<J> jac(hi - lo);
            for (size_t i = lo; i < hi; i++) { jac[i - lo] = acc; acc = acc.add_affine(gen_affine); }
            for (size_t i = lo; i < hi; i++) out[i] = jac[i - lo].to_affine();
        });
        return out;
    };
    auto h_q = make_table(C::g1_generator(), G1::infinity(), m, 1), l_q = make_table(C::g1_generator(), G1::infinity(), n_aux, 3);
    auto a_q = make_table(C::g1_generator(), G1::infinity(), n_aux, 5), b1_q = make_table(C::g1_generator(), G1::infinity(), n_aux, 7);
    auto b2_q = make_table(C::g2_generator(), G2::infinity(), n_aux, 1);
    std::vector<uint32_t> rpA(nc + 1), colA(2 * nc), rpB(nc + 1), colB(nc);
    std::vector<Fr> coA(2 * nc), coB(nc);
    for (size_t i = 0; i < nc; i++) {
        rpA[i] = 2 * i; rpB[i] = i;
        colA[2 * i] = (uint32_t)(n_inputs + i); colA[2 * i + 1] = (uint32_t)(i == 0 ? 1 : n_inputs + i - 1);
        colB[i] = (uint32_t)(n_inputs + (i * 7 + 3) % n_aux);
        coA[2 * i] = rand_fp<Fr>(rng); coA[2 * i + 1] = rand_fp<Fr>(rng); coB[i] = rand_fp<Fr>(rng);
    }
    rpA[nc] = 2 * nc; rpB[nc] = nc;
    std::vector<Fr> pub = {Fr::one(), rand_fp<Fr>(rng)}, wa(n_aux), wb(n_aux), mask1(m), mask2(m), recv1(m), recv2(m);
    for (auto* v : {&wa, &wb, &mask1, &mask2, &recv1, &recv2}) for (auto& x : *v) x = rand_fp<Fr>(rng);
    auto dom = groth16_domain<Fr>((size_t)log_m, nc, n_inputs);
    std::vector<Fr> tw(m / 2), twi(m / 2);
    { Fr wi = dom.omega.inverse(); tw[0] = twi[0] = Fr::one(); for (size_t i = 1; i < m / 2; i++) { tw[i] = tw[i - 1] * dom.omega; twi[i] = twi[i - 1] * wi; } }
    Fr ninv = Fr::from_u64((uint64_t)m).inverse();
    const int windows = (Fr::K.bits + ark_window_size(m) - 1) / ark_window_size(m);

    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    auto run = [&](int nthreads, bool with_msm, double* st) {
        Pool pool(nthreads);
        auto t0 = now();
        std::vector<Fr> aa(m, Fr::zero()), ab(m, Fr::zero()), ba(m, Fr::zero()), bb(m, Fr::zero()), ca(m), cb, ha(m), hb;
        auto spmv = [&](const std::vector<uint32_t>& rp, const std::vector<uint32_t>& col, const std::vector<Fr>& co, std::vector<Fr>& oa, std::vector<Fr>& ob) {
            pool.run(nc, 1024, [&](size_t lo, size_t hi) {
                for (size_t r = lo; r < hi; r++) {
                    Fr xa = Fr::zero(), xb = Fr::zero();
                    for (uint32_t k = rp[r]; k < rp[r + 1]; k++) {
                        size_t idx = col[k];
                        if (idx < n_inputs) xa = xa + co[k] * pub[idx];                       // party 0: add_with_public -> component a
                        else { xa = xa + co[k] * wa[idx - n_inputs]; xb = xb + co[k] * wb[idx - n_inputs]; }
                    }
                    oa[r] = xa; ob[r] = xb;
                }
            });
        };
        spmv(rpA, colA, coA, aa, ab); spmv(rpB, colB, coB, ba, bb);
        for (size_t i = 0; i < n_inputs; i++) aa[nc + i] = pub[i];
        auto mul_local = [&](std::vector<Fr>& out, const std::vector<Fr>& mask) {
            pool.run(m, 1024, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) out[i] = aa[i] * ba[i] + aa[i] * bb[i] + ab[i] * ba[i] + mask[i]; });
        };
        mul_local(ca, mask1); cb = recv1;
        auto t1 = now();
        auto pipeline = [&](std::vector<Fr>& v) {
            ntt_inverse_mt(v.data(), m, twi, ninv, pool);
            Fr pw = Fr::one(); for (auto& x : v) { x = x * pw; pw = pw * dom.coset_g; }        // serial running power, as rep3.rs:681-688
            ntt_forward_mt(v.data(), m, tw, pool);
        };
        pipeline(aa); pipeline(ab); pipeline(ba); pipeline(bb);
        auto t2 = now();
        mul_local(ha, mask2); hb = recv2;
        auto t3 = now();
        pipeline(ca); pipeline(cb);
        auto t4 = now();
        pool.run(m, 1024, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) { ha[i] = ha[i] - ca[i]; hb[i] = hb[i] - cb[i]; } });
        auto t5 = now();
        st[0] = secs(t0, t1) + secs(t2, t3) + secs(t4, t5); st[1] = secs(t1, t2) + secs(t3, t4);
        if (!with_msm) return;
        G1 acc1 = G1::infinity();
        auto m1 = [&](const std::vector<typename G1::Affine>& q, const std::vector<Fr>& s, size_t n) { acc1 = acc1.add(msm_pippenger<G1, Fr>(q.data(), s.data(), n, nthreads)); };
        m1(h_q, ha, m); m1(h_q, hb, m); m1(l_q, wa, n_aux); m1(l_q, wb, n_aux); m1(a_q, wa, n_aux); m1(a_q, wb, n_aux); m1(b1_q, wa, n_aux); m1(b1_q, wb, n_aux);
        auto t6 = now();
        G2 acc2 = msm_pippenger<G2, Fr>(b2_q.data(), wa.data(), n_aux, nthreads).add(msm_pippenger<G2, Fr>(b2_q.data(), wb.data(), n_aux, nthreads));
        auto t7 = now();
        volatile bool sink = acc1.is_inf() || acc2.is_inf(); (void)sink;
        st[2] = secs(t5, t6); st[3] = secs(t6, t7);
    };
    double sa[4] = {0, 0, 0, 0};
    run(threads, true, sa);
    if (stage) for (int i = 0; i < 4; i++) stage[i] = sa[i];
    if (threads_b > 0 && stage_b && total_b) {
        const bool shared = threads >= windows && threads_b >= windows;
        double sb2[4] = {0, 0, sa[2], sa[3]};
        run(threads_b, !shared, sb2);
        for (int i = 0; i < 4; i++) stage_b[i] = sb2[i];
        *total_b = sb2[0] + sb2[1] + sb2[2] + sb2[3];
        if (msm_shared) *msm_shared = shared ? 1 : 0;
    }
    return sa[0] + sa[1] + sa[2] + sa[3];
}

// The masks of the two mul_vec calls of one proof as the reference draws them (rep3.rs:657-661 inside the serial izip map, rngs.rs:37-46):
// per element F::rand(rng1) - F::rand(rng2), on ONE thread — 4 x m ChaCha12 rejection-sampled draws per proof.  Seconds.
template <class C>
static double bench_mask_draws(size_t m, uint64_t seed, typename C::Fr* sink_out = nullptr) {
    typedef typename C::Fr Fr;
    C::init();
    uint8_t s1[32], s2[32];
    for (int i = 0; i < 32; i++) { s1[i] = (uint8_t)(seed >> (i % 8 * 8)) ^ (uint8_t)i; s2[i] = (uint8_t)~s1[i]; }
    ChaCha12Stream r1(s1), r2(s2);
    auto t0 = std::chrono::steady_clock::now();
    Fr acc = Fr::zero();
    for (int call = 0; call < 2; call++)
        for (size_t i = 0; i < m; i++) {
            Fr a, b; fr_rand(r1, Fr::K.p, Fr::K.bits, a.v); fr_rand(r2, Fr::K.p, Fr::K.bits, b.v);
            acc = acc + (a - b);
        }
    auto t1 = std::chrono::steady_clock::now();
    if (sink_out) *sink_out = acc;
    volatile uint64_t sink = acc.v[0]; (void)sink;
    return std::chrono::duration<double>(t1 - t0).count();
}

// ONE REP3 party (party 0) of `CoGroth16::prove` on a zkey + wtns pair — the CPU twin of the GPU bench's entry legs (the reference's own
// bench circuit, tests/benches/poseidon_hash2.rs:175-223, is such a pair).  Everything the party computes between witness shares in and
// proof shares out: constraint rows, both mul_vec products WITH their mask draws (serial, as above), the six transform pipelines, the ten
// MSMs + the public-input MSMs of calculate_coeff, r / s, the five scalar products of the tail.  What the peers send is taken as given
// (random vectors / points): network time excluded, like on the GPU side.  Seconds per proof (mean of `reps`), stage[0..4] = rows + products,
// transforms, MSM G1, MSM G2 + tail, mask draws.
template <class C>
static double bench_rep3_party_file(const std::string& zkey_path, const std::string& wtns_path, int threads, int reps, double* stage) {
    typedef typename C::Fr Fr; typedef typename C::G1 G1; typedef typename C::G2 G2;
    C::init();
    const ZKey<C> z = read_zkey<C>(zkey_path, false);
    const std::vector<Fr> w = read_wtns<Fr>(wtns_path);
    if (w.size() != z.n_vars) throw std::runtime_error("witness length does not match the zkey");
    const size_t n_inputs = z.n_public + 1, n_aux = z.n_vars - n_inputs, nc = z.num_constraints;
    auto dom = groth16_domain<Fr>(z.pow, nc, n_inputs);
    const size_t m = dom.m;
    XorShift rng{0x5eed5eedull};
    std::vector<Fr> pub(w.begin(), w.begin() + n_inputs), wa(n_aux), wb(n_aux), recv1(m), recv2(m);
    for (size_t i = 0; i < n_aux; i++) { wb[i] = rand_fp<Fr>(rng); wa[i] = w[n_inputs + i] - wb[i]; }      // a share pair of the real witness (the third share is the peers')
    for (auto* v : {&recv1, &recv2}) for (auto& x : *v) x = rand_fp<Fr>(rng);
    std::vector<Fr> tw(std::max<size_t>(1, m / 2)), twi(std::max<size_t>(1, m / 2));
    { Fr wi = dom.omega.inverse(); tw[0] = twi[0] = Fr::one(); for (size_t i = 1; i < m / 2; i++) { tw[i] = tw[i - 1] * dom.omega; twi[i] = twi[i - 1] * wi; } }
    const Fr ninv = Fr::from_u64((uint64_t)m).inverse();
    uint8_t s1[32], s2[32]; for (int i = 0; i < 32; i++) { s1[i] = (uint8_t)(7 * i + 1); s2[i] = (uint8_t)(11 * i + 3); }
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    double st[5] = {0, 0, 0, 0, 0};
    Pool pool(threads);
    const auto t_begin = now();
    for (int rep = 0; rep < reps; rep++) {
        ChaCha12Stream r1(s1, (uint64_t)rep << 40), r2(s2, (uint64_t)rep << 40);
        auto t0 = now();
        std::vector<Fr> aa(m, Fr::zero()), ab(m, Fr::zero()), ba(m, Fr::zero()), bb(m, Fr::zero()), ca(m), cb, ha(m), hb, mask(m);
        auto rows = [&](int mat, std::vector<Fr>& oa, std::vector<Fr>& ob) {
            const auto& rp = z.row_ptr[mat]; const auto& col = z.col[mat]; const auto& co = z.coeff[mat];
            pool.run(nc, 1024, [&](size_t lo, size_t hi) {
                for (size_t r = lo; r < hi; r++) {
                    Fr xa = Fr::zero(), xb = Fr::zero();
                    for (uint32_t k = rp[r]; k < rp[r + 1]; k++) {
                        const size_t idx = col[k];
                        if (idx < n_inputs) xa = xa + co[k] * pub[idx];
                        else { xa = xa + co[k] * wa[idx - n_inputs]; xb = xb + co[k] * wb[idx - n_inputs]; }
                    }
                    oa[r] = xa; ob[r] = xb;
                }
            });
        };
        rows(0, aa, ab); rows(1, ba, bb);
        for (size_t i = 0; i < n_inputs; i++) aa[nc + i] = pub[i];
        double t_draw = 0;
        auto draw = [&] { auto d0 = now(); for (size_t i = 0; i < m; i++) { Fr a, b; fr_rand(r1, Fr::K.p, Fr::K.bits, a.v); fr_rand(r2, Fr::K.p, Fr::K.bits, b.v); mask[i] = a - b; } t_draw += secs(d0, now()); };
        auto mul_local = [&](std::vector<Fr>& out) {
            draw();
            pool.run(m, 1024, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) out[i] = aa[i] * ba[i] + aa[i] * bb[i] + ab[i] * ba[i] + mask[i]; });
        };
        mul_local(ca); cb = recv1;
        auto t1 = now();
        auto pipeline = [&](std::vector<Fr>& v) {
            ntt_inverse_mt(v.data(), m, twi, ninv, pool);
            Fr pw = Fr::one(); for (auto& x : v) { x = x * pw; pw = pw * dom.coset_g; }
            ntt_forward_mt(v.data(), m, tw, pool);
        };
        pipeline(aa); pipeline(ab); pipeline(ba); pipeline(bb);
        auto t2 = now();
        mul_local(ha); hb = recv2;
        auto t3 = now();
        pipeline(ca); pipeline(cb);
        auto t4 = now();
        for (size_t i = 0; i < m; i++) { ha[i] = ha[i] - ca[i]; hb[i] = hb[i] - cb[i]; }
        auto t5 = now();
        G1 acc1 = G1::infinity();
        auto m1 = [&](const typename G1::Affine* q, const Fr* sc, size_t n) { acc1 = acc1.add(msm_auto<G1, Fr>(q, sc, n, threads)); };
        const size_t hn = std::min(m, z.h_query.size());
        m1(z.h_query.data(), ha.data(), hn); m1(z.h_query.data(), hb.data(), hn);
        m1(z.l_query.data(), wa.data(), n_aux); m1(z.l_query.data(), wb.data(), n_aux);
        for (const auto* q : {&z.a_query, &z.b_g1_query}) {
            m1(q->data() + 1, pub.data() + 1, z.n_public); m1(q->data() + n_inputs, wa.data(), n_aux); m1(q->data() + n_inputs, wb.data(), n_aux);
        }
        auto t6 = now();
        G2 acc2 = msm_auto<G2, Fr>(z.b_g2_query.data() + 1, pub.data() + 1, z.n_public, 1)
                      .add(msm_auto<G2, Fr>(z.b_g2_query.data() + n_inputs, wa.data(), n_aux, threads)).add(msm_auto<G2, Fr>(z.b_g2_query.data() + n_inputs, wb.data(), n_aux, threads));
        // tail (groth16.rs:258-297): r, s, r*s; delta_g1 * {rs, r, s} and g_a_open * s and g1_b * r (three terms) per component; delta_g2 * s
        Fr r_[2], s_[2], rs_[2];
        for (int j = 0; j < 2; j++) { fr_rand(j ? r2 : r1, Fr::K.p, Fr::K.bits, r_[j].v); fr_rand(j ? r2 : r1, Fr::K.p, Fr::K.bits, s_[j].v); }
        rs_[0] = r_[0] * s_[0] + r_[0] * s_[1] + r_[1] * s_[0]; rs_[1] = rand_fp<Fr>(rng);
        const G1 d1 = G1::from_affine(z.delta_g1); const G2 d2 = G2::from_affine(z.delta_g2);
        for (int j = 0; j < 2; j++) {
            acc1 = acc1.add(scalar_mul(d1, rs_[j])).add(scalar_mul(d1, r_[j])).add(scalar_mul(d1, s_[j])).add(scalar_mul(acc1, s_[j]));
            acc2 = acc2.add(scalar_mul(d2, s_[j]));
        }
        acc1 = acc1.add(scalar_mul(acc1, r_[0])).add(scalar_mul(acc1, r_[1])).add(scalar_mul(d1, r_[0]));     // local part of scalar_mul (rep3.rs:835-847)
        auto t7 = now();
        volatile bool sink = acc1.is_inf() || acc2.is_inf(); (void)sink;
        st[0] += secs(t0, t1) + secs(t2, t3) + secs(t4, t5) - t_draw; st[1] += secs(t1, t2) + secs(t3, t4); st[2] += secs(t5, t6); st[3] += secs(t6, t7); st[4] += t_draw;
    }
    const double total = secs(t_begin, now());
    if (stage) for (int i = 0; i < 5; i++) stage[i] = st[i] / reps;
    return total / reps;
}

}  // namespace orc
