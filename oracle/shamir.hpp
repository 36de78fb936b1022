// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// Shamir-shared collaborative Groth16 (n parties, threshold t, 2t + 1 <= n), all parties simulated in lock-step.  Restates
//   `/root/reference/mpc-core/src/protocols/shamir/shamir_core.rs:8-31,56-75,97-105`   (share, lagrange_from_coeff, reconstruct)
//   `/root/reference/mpc-core/src/protocols/shamir.rs:211-246`    (Lagrange tables: opening reads shares "in circles" from the previous parties)
//   `:252-384,386-436`  (degree_reduce / _vec / _point: mask with r_2t, king = party 0 interpolates at 0, re-shares with degree t, parties subtract r_t)
//   `:471-473,609-623,625-632,642-659` (add_with_public = share + value on EVERY party, mul_vec, trivial shares, evaluate_constraint)
//   `:570-573,757-784,808-824`  (rand = the degree-t half of a double sharing, scalar_mul, open_point, open_two_points)
//   `:885-1025`  (ShamirRng: batches of 1024 random values, each shared with degree t and 2t, extraction by the (t+1) x n Vandermonde
//                 matrix, pairs handed out from the END of the buffer)
//   `/root/reference/mpc-core/src/protocols/shamir/network.rs:233-266` (broadcast_next: own value first, then party id-1, id-2, ...)
// Randomness: the reference draws from a private ChaCha stream per party (F::rand / C::rand).  Here party i consumes a caller-given
// stream S_i of field elements in the same ORDER the reference draws values (a batch: `amount` secrets, then per secret t + 2t
// polynomial coefficients; the king: t coefficients per re-shared element); a random curve point is G * next(S_i).  The reference
// pins no Shamir Groth16 output (no such test in the snapshot): parity = proofs verify, all parties agree, and the HIP host
// mirror reproduces these values on the same streams.
#pragma once
#include "groth16.hpp"

namespace orc {

template <class C>
struct ShamirSim {
    typedef typename C::Fr Fr; typedef typename C::G1 G1; typedef typename C::G2 G2;
    static constexpr size_t BATCH = 1024;
    const ZKey<C>& z;
    int n, t;
    std::vector<Fr> pub;
    std::vector<std::vector<Fr>> wit;                 // wit[i] = party i's shares of the private witness
    std::vector<const std::vector<Fr>*> stream;       // S_i
    std::vector<size_t> cur;
    std::vector<std::vector<Fr>> r_t, r_2t;           // per party: buffered double sharings
    std::vector<Fr> mul_lagrange_2t;                  // interpolation at 0 from parties 1..2t+1 (king)
    std::vector<std::vector<Fr>> open_lagrange_t;     // per party: from its own point and the t previous parties'
    int threads = 1;

    ShamirSim(const ZKey<C>& zk, int n_, int t_) : z(zk), n(n_), t(t_), wit(n_), stream(n_), cur(n_, 0), r_t(n_), r_2t(n_), open_lagrange_t(n_) {
        if (2 * t + 1 > n) throw std::runtime_error("Threshold too large for number of parties");
        std::vector<size_t> pts; for (int i = 1; i <= 2 * t + 1; i++) pts.push_back((size_t)i);
        mul_lagrange_2t = lagrange_from_coeff(pts);
        for (int id = 0; id < n; id++) {
            std::vector<size_t> p; for (int i = 0; i <= t; i++) p.push_back((size_t)((id + n - i) % n + 1));
            open_lagrange_t[id] = lagrange_from_coeff(p);
        }
    }
    static Fr from_u(size_t v) { uint64_t l[Fr::N] = {0}; l[0] = (uint64_t)v; return Fr::from_canonical(l); }
    static std::vector<Fr> lagrange_from_coeff(const std::vector<size_t>& coeffs) {   // shamir_core.rs:56-75
        std::vector<Fr> res;
        for (size_t i : coeffs) {
            Fr num = Fr::one(), den = Fr::one(); const Fr fi = from_u(i);
            for (size_t j : coeffs) if (i != j) { const Fr fj = from_u(j); num = num * fj; den = den * (fj - fi); }
            res.push_back(num * den.inverse());
        }
        return res;
    }
    Fr next(int i) { if (cur[i] >= stream[i]->size()) throw std::runtime_error("randomness stream exhausted"); return (*stream[i])[cur[i]++]; }
    std::vector<Fr> share(const Fr& secret, int degree, int who) {        // shamir_core.rs:8-31
        std::vector<Fr> coeffs; for (int k = 0; k < degree; k++) coeffs.push_back(next(who));
        std::vector<Fr> shares;
        for (int p = 1; p <= n; p++) { Fr sh = secret; const Fr x = from_u((size_t)p); Fr xp = x; for (const Fr& c : coeffs) { sh = sh + xp * c; xp = xp * x; } shares.push_back(sh); }
        return shares;
    }
    // shamir.rs:923-1010, every party at once
    void buffer_triples(size_t amount) {
        std::vector<std::vector<Fr>> sent_t(n), sent_2t(n);               // sent_*[from][k * n + to]
        for (int i = 0; i < n; i++) {
            std::vector<Fr> rnd; for (size_t k = 0; k < amount; k++) rnd.push_back(next(i));
            for (const Fr& r : rnd) { auto a = share(r, t, i), b = share(r, 2 * t, i); sent_t[i].insert(sent_t[i].end(), a.begin(), a.end()); sent_2t[i].insert(sent_2t[i].end(), b.begin(), b.end()); }
        }
        for (int me = 0; me < n; me++) {
            for (size_t k = 0; k < amount; k++) {
                std::vector<Fr> in_t(n), in_2t(n);
                for (int from = 0; from < n; from++) { in_t[from] = sent_t[from][k * n + me]; in_2t[from] = sent_2t[from][k * n + me]; }
                vandermonde_mul(in_t, r_t[me]); vandermonde_mul(in_2t, r_2t[me]);
            }
        }
    }
    void vandermonde_mul(const std::vector<Fr>& in, std::vector<Fr>& out) {   // :904-921, appends t + 1 values
        std::vector<Fr> row(n), curr(n);
        for (int p = 0; p < n; p++) { row[p] = from_u((size_t)p + 1); curr[p] = row[p]; }
        Fr s0 = Fr::zero(); for (const Fr& v : in) s0 = s0 + v;
        out.push_back(s0);
        for (int k = 1; k <= t; k++) { Fr acc = Fr::zero(); for (int p = 0; p < n; p++) { acc = acc + curr[p] * in[p]; curr[p] = curr[p] * row[p]; } out.push_back(acc); }
    }
    void preprocess(size_t amount) { if (amount) buffer_triples(amount); }                        // :248-250
    // one pair per party, taken in lock-step (:1012-1025)
    void get_pairs(std::vector<Fr>& rt, std::vector<Fr>& r2t) {
        if (r_t[0].empty()) buffer_triples(BATCH);
        rt.resize(n); r2t.resize(n);
        for (int i = 0; i < n; i++) { rt[i] = r_t[i].back(); r_t[i].pop_back(); r2t[i] = r_2t[i].back(); r_2t[i].pop_back(); }
    }
    std::vector<Fr> rand() { std::vector<Fr> a, b; get_pairs(a, b); return a; }                 // :570-573
    // :302-384; inputs[i] = party i's local products
    std::vector<std::vector<Fr>> degree_reduce_vec(std::vector<std::vector<Fr>> inputs) {
        const size_t len = inputs[0].size();
        std::vector<std::vector<Fr>> rts(n, std::vector<Fr>(len));
        for (size_t k = 0; k < len; k++) { std::vector<Fr> a, b; get_pairs(a, b); for (int i = 0; i < n; i++) { inputs[i][k] = inputs[i][k] + b[i]; rts[i][k] = a[i]; } }
        std::vector<std::vector<Fr>> out(n, std::vector<Fr>(len));
        for (size_t k = 0; k < len; k++) {
            Fr acc = Fr::zero();
            for (int p = 0; p <= 2 * t; p++) acc = acc + inputs[p][k] * mul_lagrange_2t[p];
            auto sh = share(acc, t, 0);
            for (int i = 0; i < n; i++) out[i][k] = sh[i] - rts[i][k];
        }
        return out;
    }
    std::vector<Fr> degree_reduce(const std::vector<Fr>& inputs) {                              // :252-300
        std::vector<std::vector<Fr>> v(n); for (int i = 0; i < n; i++) v[i] = {inputs[i]};
        auto r = degree_reduce_vec(v);
        std::vector<Fr> o(n); for (int i = 0; i < n; i++) o[i] = r[i][0];
        return o;
    }
    template <class J>
    std::vector<J> degree_reduce_point(std::vector<J> inputs, const J& gen) {                  // :386-436
        std::vector<Fr> a, b; get_pairs(a, b);
        for (int i = 0; i < n; i++) inputs[i] = inputs[i].add(scalar_mul(gen, b[i]));
        J acc = J::infinity();
        for (int p = 0; p <= 2 * t; p++) acc = acc.add(scalar_mul(inputs[p], mul_lagrange_2t[p]));
        std::vector<J> coeffs; for (int k = 0; k < t; k++) coeffs.push_back(scalar_mul(gen, next(0)));            // C::rand stand-in
        std::vector<J> out(n);
        for (int i = 0; i < n; i++) {
            J sh = acc; const Fr x = from_u((size_t)i + 1); Fr xp = x;
            for (const J& c : coeffs) { sh = sh.add(scalar_mul(c, xp)); xp = xp * x; }
            out[i] = sh.add(scalar_mul(gen, a[i]).neg());
        }
        return out;
    }
    template <class J>
    J open_point(const std::vector<J>& shares, int id) const {                                  // :778-782 as seen by party `id`
        J r = J::infinity();
        for (int i = 0; i <= t; i++) r = r.add(scalar_mul(shares[(id + n - i) % n], open_lagrange_t[id][i]));
        return r;
    }
    Fr evaluate_constraint(int id, int m, size_t row) const {                                    // :642-659
        Fr acc = Fr::zero();
        for (uint32_t k = z.row_ptr[m][row]; k < z.row_ptr[m][row + 1]; k++) {
            const size_t idx = z.col[m][k];
            acc = acc + z.coeff[m][k] * (idx < pub.size() ? pub[idx] : wit[id][idx - pub.size()]);
        }
        return acc;
    }
    std::vector<std::vector<Fr>> witness_map() {                                                 // groth16.rs:141-204 on single-component shares
        const size_t num_inputs = pub.size(), nc = z.num_constraints;
        auto dom = groth16_domain<Fr>(z.pow, nc, num_inputs);
        std::vector<std::vector<Fr>> a(n), b(n), prod(n);
        for (int id = 0; id < n; id++) {
            a[id].assign(dom.m, Fr::zero()); b[id].assign(dom.m, Fr::zero()); prod[id].resize(dom.m);
            for (size_t i = 0; i < nc; i++) { a[id][i] = evaluate_constraint(id, 0, i); b[id][i] = evaluate_constraint(id, 1, i); }
            for (size_t i = 0; i < num_inputs; i++) a[id][nc + i] = pub[i];                      // trivial share = the value itself (:625-632)
            for (size_t i = 0; i < dom.m; i++) prod[id][i] = a[id][i] * b[id][i];
        }
        auto c = degree_reduce_vec(prod);
        auto pipeline = [&](std::vector<Fr>& v) { ntt_inverse(v.data(), dom.m, dom.omega); distribute_powers(v, dom.coset_g, Fr::one()); ntt_forward(v.data(), dom.m, dom.omega); };
        for (int id = 0; id < n; id++) { pipeline(a[id]); pipeline(b[id]); for (size_t i = 0; i < dom.m; i++) prod[id][i] = a[id][i] * b[id][i]; }
        auto ab = degree_reduce_vec(prod);
        for (int id = 0; id < n; id++) { pipeline(c[id]); for (size_t i = 0; i < dom.m; i++) ab[id][i] = ab[id][i] - c[id][i]; }
        return ab;
    }
    template <class J>
    J calc_coeff(int id, const J& initial, const std::vector<typename J::Affine>& q, const typename J::Affine& vk) const {   // groth16.rs:206-235
        const size_t pub_len = pub.size() - 1;
        J pub_acc = msm_auto<J, Fr>(q.data() + 1, pub.data() + 1, pub_len, 1);
        J priv = msm_auto<J, Fr>(q.data() + 1 + pub_len, wit[id].data(), wit[id].size(), threads);
        return initial.add_affine(q[0]).add_affine(vk).add(pub_acc).add(priv);                  // add_assign_points_public on every party (:733-735)
    }
    // every party's proof (they must coincide)
    std::vector<Proof<C>> prove(std::vector<std::vector<Fr>>* h_out = nullptr) {
        auto h = witness_map();
        if (h_out) *h_out = h;
        auto r = rand(), s = rand();
        const G1 delta1 = G1::from_affine(z.delta_g1); const G2 delta2 = G2::from_affine(z.delta_g2);
        std::vector<Fr> rs_local(n); for (int i = 0; i < n; i++) rs_local[i] = r[i] * s[i];
        auto rs = degree_reduce(rs_local);
        std::vector<G1> h_acc(n), l_acc(n), g_a(n), g1_b(n), local(n);
        std::vector<G2> g2_b(n);
        for (int i = 0; i < n; i++) {
            h_acc[i] = msm_auto<G1, Fr>(z.h_query.data(), h[i].data(), std::min(h[i].size(), z.h_query.size()), threads);
            l_acc[i] = msm_auto<G1, Fr>(z.l_query.data(), wit[i].data(), wit[i].size(), threads);
            g_a[i] = calc_coeff<G1>(i, scalar_mul(delta1, r[i]), z.a_query, z.alpha_g1);
        }
        std::vector<Proof<C>> out(n);
        std::vector<G1> g_a_open(n); for (int i = 0; i < n; i++) g_a_open[i] = open_point<G1>(g_a, i);
        for (int i = 0; i < n; i++) {
            g1_b[i] = calc_coeff<G1>(i, scalar_mul(delta1, s[i]), z.b_g1_query, z.beta_g1);
            local[i] = scalar_mul(g1_b[i], r[i]);                                                   // scalar_mul local part (:769-776)
        }
        auto r_g1_b = degree_reduce_point<G1>(local, G1::from_affine(C::g1_generator()));
        std::vector<G1> g_c(n);
        for (int i = 0; i < n; i++) {
            g2_b[i] = calc_coeff<G2>(i, scalar_mul(delta2, s[i]), z.b_g2_query, z.beta_g2);
            g_c[i] = scalar_mul(g_a_open[i], s[i]).add(r_g1_b[i]).add(scalar_mul(delta1, rs[i]).neg()).add(l_acc[i]).add(h_acc[i]);
        }
        for (int i = 0; i < n; i++) out[i] = {g_a_open[i].to_affine(), open_point<G2>(g2_b, i).to_affine(), open_point<G1>(g_c, i).to_affine()};
        return out;
    }
};

}  // namespace orc
