// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// CPU restatement of the collaborative Groth16 prover, following line by line
//   `/root/reference/co-circom/co-groth16/src/groth16.rs:57-77`   root_of_unity_for_groth16
//   `/root/reference/co-circom/co-groth16/src/groth16.rs:141-204` witness_map_from_matrices
//   `/root/reference/co-circom/co-groth16/src/groth16.rs:206-235` calculate_coeff
//   `/root/reference/co-circom/co-groth16/src/groth16.rs:237-326` create_proof_with_assignment
// for two drivers:
//   plain  — `/root/reference/mpc-core/src/protocols/plain.rs:215-284,369-416`
//   REP3   — `/root/reference/mpc-core/src/protocols/rep3.rs:503-511,595-725,769-947`,
//            `rep3/fieldshare.rs:72-78,161-168,262-283`, `rep3/pointshare.rs:117-124`, `rep3/rngs.rs:37-60`
//            with the three parties run in lock-step in one process (every `send_next`/`recv_prev` pair becomes an
//            array hand-over), like the reference's own in-process test network `tests/src/rep3_network.rs`.
// Randomness (r, s, re-sharing masks) is an INPUT: party i consumes stream S_i as its `rng1` and S_{i-1} as its
// `rng2` (`rngs.rs:25-46`); the reference draws these from ChaCha12, and no reference test pins proof bytes.
#pragma once
#include "formats.hpp"
#include "poly.hpp"

namespace orc {

template <class C>
struct Proof {
    typename C::G1::Affine a;
    typename C::G2::Affine b;
    typename C::G1::Affine c;
    bool operator==(const Proof& o) const { return a == o.a && b == o.b && c == o.c; }
};

template <class Fr>
struct Groth16Domain { size_t m; int log_m; Fr omega; Fr coset_g; };

// groth16.rs:57-77 + :150-153 (domain size = next_pow2(num_constraints + num_inputs))
template <class Fr>
static Groth16Domain<Fr> groth16_domain(size_t pow, size_t num_constraints, size_t num_inputs) {
    Groth16Domain<Fr> d;
    size_t need = num_constraints + num_inputs;
    d.m = 1; d.log_m = 0;
    while (d.m < need) { d.m <<= 1; d.log_m++; }
    auto rt = roots_of_unity<Fr>();
    d.omega = rt.roots[pow];
    if (rt.two_adicity == d.log_m) d.coset_g = rt.q.sqr();
    else d.coset_g = rt.roots[d.log_m + 1];
    return d;
}

// plain.rs:226-234
template <class Fr>
static void distribute_powers(std::vector<Fr>& v, const Fr& g, const Fr& c) {
    Fr pw = c;
    for (auto& x : v) { x = x * pw; pw = pw * g; }
}

// ------------------------------------------------------------------------------------------------
// plain driver
// ------------------------------------------------------------------------------------------------
template <class C>
static std::vector<typename C::Fr> witness_map_plain(const ZKey<C>& z, const std::vector<typename C::Fr>& pub,
                                                     const std::vector<typename C::Fr>& wit) {
    typedef typename C::Fr Fr;
    const size_t num_inputs = z.n_public + 1, nc = z.num_constraints;
    auto dom = groth16_domain<Fr>(z.pow, nc, num_inputs);
    std::vector<Fr> a(dom.m, Fr::zero()), b(dom.m, Fr::zero());
    auto eval = [&](int m, size_t row) {   // plain.rs:243-258
        Fr acc = Fr::zero();
        for (uint32_t k = z.row_ptr[m][row]; k < z.row_ptr[m][row + 1]; k++) {
            size_t idx = z.col[m][k];
            acc = acc + z.coeff[m][k] * (idx < pub.size() ? pub[idx] : wit[idx - pub.size()]);
        }
        return acc;
    };
    for (size_t i = 0; i < nc; i++) { a[i] = eval(0, i); b[i] = eval(1, i); }
    for (size_t i = 0; i < num_inputs; i++) a[nc + i] = pub[i];          // groth16.rs:168-171
    std::vector<Fr> c(dom.m);
    for (size_t i = 0; i < dom.m; i++) c[i] = a[i] * b[i];                // :174
    ntt_inverse(a.data(), dom.m, dom.omega); ntt_inverse(b.data(), dom.m, dom.omega);
    distribute_powers(a, dom.coset_g, Fr::one()); distribute_powers(b, dom.coset_g, Fr::one());
    ntt_forward(a.data(), dom.m, dom.omega); ntt_forward(b.data(), dom.m, dom.omega);
    std::vector<Fr> ab(dom.m);
    for (size_t i = 0; i < dom.m; i++) ab[i] = a[i] * b[i];               // :190
    ntt_inverse(c.data(), dom.m, dom.omega);
    distribute_powers(c, dom.coset_g, Fr::one());
    ntt_forward(c.data(), dom.m, dom.omega);
    for (size_t i = 0; i < dom.m; i++) ab[i] = ab[i] - c[i];              // :202
    return ab;
}

template <class J, class Fr>
static J msm_auto(const typename J::Affine* bases, const Fr* sc, size_t n, int threads) {
    if (n == 0) return J::infinity();
    return msm_pippenger<J, Fr>(bases, sc, n, threads);
}
template <class J, class Fr>
static J scalar_mul(const J& p, const Fr& s) { uint64_t e[Fr::N]; s.to_canonical(e); return p.mul(e, Fr::N); }

template <class C>
static Proof<C> prove_plain(const ZKey<C>& z, const std::vector<typename C::Fr>& full_witness,
                            const typename C::Fr& r, const typename C::Fr& s, int threads = 1,
                            std::vector<typename C::Fr>* h_out = nullptr) {
    typedef typename C::Fr Fr; typedef typename C::G1 G1; typedef typename C::G2 G2;
    const size_t np1 = z.n_public + 1;
    std::vector<Fr> pub(full_witness.begin(), full_witness.begin() + np1);
    std::vector<Fr> wit(full_witness.begin() + np1, full_witness.end());
    std::vector<Fr> h = witness_map_plain<C>(z, pub, wit);
    if (h_out) *h_out = h;
    const Fr* inp = pub.data() + 1; const size_t pub_len = pub.size() - 1;
    // groth16.rs:248-251
    G1 h_acc = msm_auto<G1, Fr>(z.h_query.data(), h.data(), std::min(h.size(), z.h_query.size()), threads);
    G1 l_acc = msm_auto<G1, Fr>(z.l_query.data(), wit.data(), wit.size(), threads);
    G1 delta1 = G1::from_affine(z.delta_g1);
    Fr rs = r * s;
    G1 r_s_delta = scalar_mul(delta1, rs);
    G1 r_g1 = scalar_mul(delta1, r);
    auto coeff1 = [&](const G1& initial, const std::vector<typename G1::Affine>& q, const typename G1::Affine& vk) {
        G1 pub_acc = msm_auto<G1, Fr>(q.data() + 1, inp, pub_len, 1);
        G1 priv_acc = msm_auto<G1, Fr>(q.data() + 1 + pub_len, wit.data(), wit.size(), threads);
        return initial.add_affine(q[0]).add_affine(vk).add(pub_acc).add(priv_acc);
    };
    G1 g_a = coeff1(r_g1, z.a_query, z.alpha_g1);
    G1 s_g_a = scalar_mul(g_a, s);
    G1 s_g1 = scalar_mul(delta1, s);
    G1 g1_b = coeff1(s_g1, z.b_g1_query, z.beta_g1);
    G1 r_g1_b = scalar_mul(g1_b, r);
    G2 delta2 = G2::from_affine(z.delta_g2);
    G2 s_g2 = scalar_mul(delta2, s);
    G2 pub2 = msm_auto<G2, Fr>(z.b_g2_query.data() + 1, inp, pub_len, 1);
    G2 priv2 = msm_auto<G2, Fr>(z.b_g2_query.data() + 1 + pub_len, wit.data(), wit.size(), threads);
    G2 g2_b = s_g2.add_affine(z.b_g2_query[0]).add_affine(z.beta_g2).add(pub2).add(priv2);
    G1 g_c = s_g_a.add(r_g1_b).add(r_s_delta.neg()).add(l_acc).add(h_acc);
    return {g_a.to_affine(), g2_b.to_affine(), g_c.to_affine()};
}

// ------------------------------------------------------------------------------------------------
// REP3, three parties in lock-step
// ------------------------------------------------------------------------------------------------
template <class Fr> struct ShareVec { std::vector<Fr> a, b; };
template <class Fr> struct Share { Fr a, b; };
template <class J> struct PShare { J a, b; };

template <class C>
struct Rep3Sim {
    typedef typename C::Fr Fr; typedef typename C::G1 G1; typedef typename C::G2 G2;
    const ZKey<C>& z;
    std::vector<Fr> pub;                 // public inputs incl. the leading 1
    ShareVec<Fr> wit[3];                 // witness shares of party 0,1,2
    const std::vector<Fr>* stream[3];    // S_0,S_1,S_2 ; party i: rng1 = S_i, rng2 = S_{i-1}
    size_t cur[3] = {0, 0, 0};
    int threads = 1;

    Rep3Sim(const ZKey<C>& zk) : z(zk) {}

    Fr draw1(int i) { return (*stream[i])[cur[i]]; }
    Fr draw2(int i) { return (*stream[(i + 2) % 3])[cur[i]]; }
    // rngs.rs:37-46: mask = F::rand(rng1) - F::rand(rng2), one position consumed on both streams
    Fr masking(int i) { Fr m = draw1(i) - draw2(i); cur[i]++; return m; }
    Share<Fr> rand(int i) { Share<Fr> s = {draw1(i), draw2(i)}; cur[i]++; return s; }   // rep3.rs:595-598

    // rep3.rs:650-670
    void mul_vec(ShareVec<Fr> (&out)[3], const ShareVec<Fr> (&x)[3], const ShareVec<Fr> (&y)[3]) {
        std::vector<Fr> local[3];
        for (int i = 0; i < 3; i++) {
            size_t n = x[i].a.size();
            local[i].resize(n);
            for (size_t k = 0; k < n; k++)
                local[i][k] = x[i].a[k] * y[i].a[k] + x[i].a[k] * y[i].b[k] + x[i].b[k] * y[i].a[k] + masking(i);
        }
        for (int i = 0; i < 3; i++) { out[i].a = local[i]; out[i].b = local[(i + 2) % 3]; }   // send_next / recv_prev
    }
    // rep3.rs:690-708 with add_with_public :600-608
    Share<Fr> evaluate_constraint(int id, int m, size_t row) {
        Share<Fr> acc = {Fr::zero(), Fr::zero()};
        for (uint32_t k = z.row_ptr[m][row]; k < z.row_ptr[m][row + 1]; k++) {
            size_t idx = z.col[m][k];
            const Fr& co = z.coeff[m][k];
            if (idx < pub.size()) {
                Fr v = pub[idx] * co;
                if (id == 0) acc.a = acc.a + v; else if (id == 1) acc.b = acc.b + v;
            } else {
                acc.a = acc.a + co * wit[id].a[idx - pub.size()];
                acc.b = acc.b + co * wit[id].b[idx - pub.size()];
            }
        }
        return acc;
    }
    template <class J>
    PShare<J> msm_shared(const typename J::Affine* bases, const ShareVec<Fr>& sc, size_t n) {   // rep3.rs:934-947
        return {msm_auto<J, Fr>(bases, sc.a.data(), n, threads), msm_auto<J, Fr>(bases, sc.b.data(), n, threads)};
    }
    template <class J>
    static void add_public(int id, PShare<J>& p, const J& q) {   // rep3.rs:804-810
        if (id == 0) p.a = p.a.add(q); else if (id == 1) p.b = p.b.add(q);
    }
    template <class J>
    PShare<J> calc_coeff(int id, PShare<J> initial, const std::vector<typename J::Affine>& q, const typename J::Affine& vk) {
        const size_t pub_len = pub.size() - 1;
        J pub_acc = msm_auto<J, Fr>(q.data() + 1, pub.data() + 1, pub_len, 1);
        PShare<J> priv = msm_shared<J>(q.data() + 1 + pub_len, wit[id], wit[id].a.size());
        PShare<J> res = initial;
        add_public(id, res, J::from_affine(q[0]));
        add_public(id, res, J::from_affine(vk));
        add_public(id, res, pub_acc);
        res.a = res.a.add(priv.a); res.b = res.b.add(priv.b);
        return res;
    }

    void witness_map(ShareVec<Fr> (&h)[3]) {
        const size_t num_inputs = pub.size(), nc = z.num_constraints;
        auto dom = groth16_domain<Fr>(z.pow, nc, num_inputs);
        ShareVec<Fr> a[3], b[3], c[3], ab[3];
        for (int id = 0; id < 3; id++) {
            a[id].a.assign(dom.m, Fr::zero()); a[id].b.assign(dom.m, Fr::zero());
            b[id].a.assign(dom.m, Fr::zero()); b[id].b.assign(dom.m, Fr::zero());
            for (size_t i = 0; i < nc; i++) {
                auto ea = evaluate_constraint(id, 0, i), eb = evaluate_constraint(id, 1, i);
                a[id].a[i] = ea.a; a[id].b[i] = ea.b; b[id].a[i] = eb.a; b[id].b[i] = eb.b;
            }
            // promote_to_trivial_shares + clone_from_slice (fieldshare.rs:262-283, rep3.rs:710-725)
            for (size_t i = 0; i < num_inputs; i++) {
                a[id].a[nc + i] = id == 0 ? pub[i] : Fr::zero();
                a[id].b[nc + i] = id == 1 ? pub[i] : Fr::zero();
            }
        }
        mul_vec(c, a, b);
        auto pipeline = [&](std::vector<Fr>& v) {
            ntt_inverse(v.data(), dom.m, dom.omega);
            distribute_powers(v, dom.coset_g, Fr::one());
            ntt_forward(v.data(), dom.m, dom.omega);
        };
        for (int id = 0; id < 3; id++) { pipeline(a[id].a); pipeline(a[id].b); pipeline(b[id].a); pipeline(b[id].b); }
        mul_vec(ab, a, b);
        for (int id = 0; id < 3; id++) {
            pipeline(c[id].a); pipeline(c[id].b);
            for (size_t i = 0; i < dom.m; i++) { ab[id].a[i] = ab[id].a[i] - c[id].a[i]; ab[id].b[i] = ab[id].b[i] - c[id].b[i]; }
            h[id] = ab[id];
        }
    }

    // returns the three parties' proofs (the reference asserts they are equal, e2e_tests/mod.rs:70-71)
    void prove(Proof<C> (&out)[3], ShareVec<Fr> (*h_out)[3] = nullptr) {
        ShareVec<Fr> h[3];
        witness_map(h);
        if (h_out) for (int i = 0; i < 3; i++) (*h_out)[i] = h[i];
        Share<Fr> r[3], s[3];
        for (int i = 0; i < 3; i++) r[i] = rand(i);
        for (int i = 0; i < 3; i++) s[i] = rand(i);
        G1 delta1 = G1::from_affine(z.delta_g1); G2 delta2 = G2::from_affine(z.delta_g2);
        PShare<G1> h_acc[3], l_acc[3], r_s_delta[3], g_a[3], s_g_a[3], g1_b[3], r_g1_b[3], g_c[3];
        PShare<G2> g2_b[3];
        Share<Fr> rs[3];
        Fr rs_local[3];
        for (int i = 0; i < 3; i++) {
            h_acc[i] = msm_shared<G1>(z.h_query.data(), h[i], std::min(h[i].a.size(), z.h_query.size()));
            l_acc[i] = msm_shared<G1>(z.l_query.data(), wit[i], wit[i].a.size());
            rs_local[i] = r[i].a * s[i].a + r[i].a * s[i].b + r[i].b * s[i].a + masking(i);   // rep3.rs:503-511
        }
        for (int i = 0; i < 3; i++) rs[i] = {rs_local[i], rs_local[(i + 2) % 3]};
        G1 g_a_open[3];
        for (int i = 0; i < 3; i++) {
            r_s_delta[i] = {scalar_mul(delta1, rs[i].a), scalar_mul(delta1, rs[i].b)};
            PShare<G1> r_g1 = {scalar_mul(delta1, r[i].a), scalar_mul(delta1, r[i].b)};
            g_a[i] = calc_coeff<G1>(i, r_g1, z.a_query, z.alpha_g1);
        }
        for (int i = 0; i < 3; i++) g_a_open[i] = g_a[i].a.add(g_a[i].b).add(g_a[(i + 2) % 3].b);   // open_point rep3.rs:849-853
        G1 local_pt[3];
        for (int i = 0; i < 3; i++) {
            s_g_a[i] = {scalar_mul(g_a_open[i], s[i].a), scalar_mul(g_a_open[i], s[i].b)};
            PShare<G1> s_g1 = {scalar_mul(delta1, s[i].a), scalar_mul(delta1, s[i].b)};
            g1_b[i] = calc_coeff<G1>(i, s_g1, z.b_g1_query, z.beta_g1);
            // scalar_mul rep3.rs:835-847 with pointshare.rs:117-124 and masking_ec_element (here: G*(f1) - G*(f2))
            G1 gen = G1::from_affine(C::g1_generator());
            G1 mask = scalar_mul(gen, draw1(i)).add(scalar_mul(gen, draw2(i)).neg()); cur[i]++;
            local_pt[i] = scalar_mul(g1_b[i].a, r[i].a).add(scalar_mul(g1_b[i].b, r[i].a)).add(scalar_mul(g1_b[i].a, r[i].b)).add(mask);
        }
        for (int i = 0; i < 3; i++) r_g1_b[i] = {local_pt[i], local_pt[(i + 2) % 3]};
        for (int i = 0; i < 3; i++) {
            PShare<G2> s_g2 = {scalar_mul(delta2, s[i].a), scalar_mul(delta2, s[i].b)};
            g2_b[i] = calc_coeff<G2>(i, s_g2, z.b_g2_query, z.beta_g2);
            g_c[i].a = s_g_a[i].a.add(r_g1_b[i].a).add(r_s_delta[i].a.neg()).add(l_acc[i].a).add(h_acc[i].a);
            g_c[i].b = s_g_a[i].b.add(r_g1_b[i].b).add(r_s_delta[i].b.neg()).add(l_acc[i].b).add(h_acc[i].b);
        }
        for (int i = 0; i < 3; i++) {   // open_two_points rep3.rs:865-877
            int p = (i + 2) % 3;
            G1 c_open = g_c[i].a.add(g_c[i].b).add(g_c[p].b);
            G2 b_open = g2_b[i].a.add(g2_b[i].b).add(g2_b[p].b);
            out[i] = {g_a_open[i].to_affine(), b_open.to_affine(), c_open.to_affine()};
        }
    }
};

}  // namespace orc
