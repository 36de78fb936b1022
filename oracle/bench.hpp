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
#include <chrono>
#include <functional>

namespace orc {

static inline void parallel_for(size_t n, int threads, const std::function<void(size_t, size_t)>& fn) {
    if (threads <= 1 || n < 1024) { fn(0, n); return; }
    std::vector<std::thread> pool;
    size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        size_t lo = (size_t)t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        pool.emplace_back([=, &fn] { fn(lo, hi); });
    }
    for (auto& th : pool) th.join();
}

// same transforms as poly.hpp, butterflies of each stage split across threads
template <class F>
static void ntt_forward_mt(F* a, size_t n, const std::vector<F>& tw, int threads) {
    for (size_t half = n / 2, step = 1; half >= 1; half >>= 1, step <<= 1)
        parallel_for(n / 2, threads, [&](size_t lo, size_t hi) {
            for (size_t u = lo; u < hi; u++) {
                size_t blk = (u / half) * 2 * half, j = u % half;
                F x = a[blk + j], y = a[blk + j + half];
                a[blk + j] = x + y; a[blk + j + half] = (x - y) * tw[j * step];
            }
        });
    bitrev_permute(a, n);
}
template <class F>
static void ntt_inverse_mt(F* a, size_t n, const std::vector<F>& twi, const F& ninv, int threads) {
    bitrev_permute(a, n);
    for (size_t half = 1, step = n / 2; half < n; half <<= 1, step >>= 1)
        parallel_for(n / 2, threads, [&](size_t lo, size_t hi) {
            for (size_t u = lo; u < hi; u++) {
                size_t blk = (u / half) * 2 * half, j = u % half;
                F x = a[blk + j], y = a[blk + j + half] * twi[j * step];
                a[blk + j] = x + y; a[blk + j + half] = x - y;
            }
        });
    parallel_for(n, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) a[i] = a[i] * ninv; });
}

struct XorShift { uint64_t s; uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; } };
template <class F> static F rand_fp(XorShift& r) {
    F x; for (int i = 0; i < F::N; i++) x.v[i] = r.next();
    x.v[F::N - 1] &= ((uint64_t)1 << ((F::K.bits - 2) % 64)) - 1;   // < 2^(bits-2) < p: a valid reduced representative
    return x;
}

// returns wall seconds of one party-0 prove compute at domain size m = 2^log_m; stage[0..3] = spmv+pointwise, ntt, msm_g1, msm_g2
template <class C>
static double bench_rep3_party(int log_m, int threads, uint64_t seed, double* stage) {
    typedef typename C::Fr Fr; typedef typename C::G1 G1; typedef typename C::G2 G2;
    C::init();
    const size_t m = (size_t)1 << log_m, nc = m - 2, n_aux = m - 2, n_inputs = 2;
    XorShift rng{seed | 1};
    // ---- untimed setup: bases = consecutive multiples of the generators, CSR matrices, shares
    auto make_g1 = [&](size_t n, uint64_t first) {
        std::vector<typename G1::Affine> out(n);
        uint64_t k[1] = {first};
        G1 acc = G1::from_affine(C::g1_generator()).mul(k, 1);
        std::vector<G1> jac(n);
        for (size_t i = 0; i < n; i++) { jac[i] = acc; acc = acc.add_affine(C::g1_generator()); }
        std::vector<typename C::Fq> zs(n);
        for (size_t i = 0; i < n; i++) zs[i] = jac[i].z;
        batch_inverse(zs.data(), n);
        parallel_for(n, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) { auto zi2 = zs[i].sqr(); out[i] = {jac[i].x * zi2, jac[i].y * zi2 * zs[i], false}; } });
        return out;
    };
    auto h_q = make_g1(m, 1), l_q = make_g1(n_aux, 3), a_q = make_g1(n_aux, 5), b1_q = make_g1(n_aux, 7);
    std::vector<typename G2::Affine> b2_q(n_aux);
    {
        G2 acc = G2::from_affine(C::g2_generator());
        std::vector<G2> jac(n_aux);
        for (size_t i = 0; i < n_aux; i++) { jac[i] = acc; acc = acc.add_affine(C::g2_generator()); }
        parallel_for(n_aux, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) b2_q[i] = jac[i].to_affine(); });
    }
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
    std::vector<Fr> tw(m / 2), twi(m / 2), gp(m);
    { Fr wi = dom.omega.inverse(); tw[0] = twi[0] = Fr::one(); for (size_t i = 1; i < m / 2; i++) { tw[i] = tw[i - 1] * dom.omega; twi[i] = twi[i - 1] * wi; } }
    Fr ninv = Fr::from_u64((uint64_t)m).inverse();

    // ---- timed region
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    auto t0 = now();
    std::vector<Fr> aa(m, Fr::zero()), ab(m, Fr::zero()), ba(m, Fr::zero()), bb(m, Fr::zero()), ca(m), cb, ha(m), hb;
    auto spmv = [&](const std::vector<uint32_t>& rp, const std::vector<uint32_t>& col, const std::vector<Fr>& co, std::vector<Fr>& oa, std::vector<Fr>& ob) {
        parallel_for(nc, threads, [&](size_t lo, size_t hi) {
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
        parallel_for(m, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) out[i] = aa[i] * ba[i] + aa[i] * bb[i] + ab[i] * ba[i] + mask[i]; });
    };
    mul_local(ca, mask1); cb = recv1;
    auto t1 = now();
    auto pipeline = [&](std::vector<Fr>& v) {
        ntt_inverse_mt(v.data(), m, twi, ninv, threads);
        Fr pw = Fr::one(); for (auto& x : v) { x = x * pw; pw = pw * dom.coset_g; }        // serial running power, as rep3.rs:681-688
        ntt_forward_mt(v.data(), m, tw, threads);
    };
    pipeline(aa); pipeline(ab); pipeline(ba); pipeline(bb);
    auto t2 = now();
    mul_local(ha, mask2); hb = recv2;
    auto t3 = now();
    pipeline(ca); pipeline(cb);
    auto t4 = now();
    parallel_for(m, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) { ha[i] = ha[i] - ca[i]; hb[i] = hb[i] - cb[i]; } });
    auto t5 = now();
    G1 acc1 = G1::infinity();
    auto m1 = [&](const std::vector<typename G1::Affine>& q, const std::vector<Fr>& s, size_t n) { acc1 = acc1.add(msm_pippenger<G1, Fr>(q.data(), s.data(), n, threads)); };
    m1(h_q, ha, m); m1(h_q, hb, m); m1(l_q, wa, n_aux); m1(l_q, wb, n_aux); m1(a_q, wa, n_aux); m1(a_q, wb, n_aux); m1(b1_q, wa, n_aux); m1(b1_q, wb, n_aux);
    auto t6 = now();
    G2 acc2 = msm_pippenger<G2, Fr>(b2_q.data(), wa.data(), n_aux, threads).add(msm_pippenger<G2, Fr>(b2_q.data(), wb.data(), n_aux, threads));
    auto t7 = now();
    volatile bool sink = acc1.is_inf() || acc2.is_inf(); (void)sink;
    if (stage) { stage[0] = secs(t0, t1) + secs(t2, t3) + secs(t4, t5); stage[1] = secs(t1, t2) + secs(t3, t4); stage[2] = secs(t5, t6); stage[3] = secs(t6, t7); }
    return secs(t0, t7);
}

}  // namespace orc
