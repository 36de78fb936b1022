// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// Synthetic satisfiable R1CS + Groth16 CRS generator (SURVEY.md §8d "synthetic R1CS generator"): writes a snarkjs-format
// `.zkey` (sections 1-9, `/root/reference/co-circom/circom-types/src/groth16/zkey.rs:139-316`) and `.wtns`
// (`witness.rs:51-91`) so that the format readers, the witness map and all five MSMs are exercised at sizes the shipped
// fixtures (<= 213 constraints) do not reach.  Circuit: n_public = 1 by default (any number on request), num_constraints = m - n_public - 1, n_vars = m (domain size exactly m);
// constraint j:  (a_j * w[j+1]) * (b_j * w[sb_j]) = w[j+2]   with sb_j <= j+1, so the witness is computed forward.
// CRS from seeded toxic waste (tau, alpha, beta, gamma, delta), in the snarkjs conventions the reference consumes:
//   * section 4 carries the extra rows  A[nc+i] = w_i (i <= n_public)  that the prover mirrors at groth16.rs:168-171
//   * h_query[i] = [ (tau^2m - 1) * g w^i / (2 m delta (tau - g w^i)) ]_1 : the quotient H = (AB - C)/Z is interpolated on the odd
//     coset g*H (g = w_2m, Z(g w^i) = -2), which is why witness_map_from_matrices needs no division (groth16.rs:141-204).
// A proof made from these files verifying under the pairing check is the evidence that the convention is right.
#pragma once
#include "bench.hpp"

namespace orc {

// fixed-base multiplication by 8-bit windows: table[w][d-1] = d * 2^(8w) * G (Jacobian)
template <class J>
struct FixedBase {
    std::vector<std::vector<J>> tab;
    explicit FixedBase(const J& g, int bits) {
        int nw = (bits + 7) / 8;
        tab.resize(nw);
        J base = g;
        for (int w = 0; w < nw; w++) {
            tab[w].resize(255);
            J acc = base;
            for (int d = 0; d < 255; d++) { tab[w][d] = acc; acc = acc.add(base); }
            base = acc;   // 256 * base
        }
    }
    template <class Fr> J mul(const Fr& s) const {
        uint64_t e[Fr::N]; s.to_canonical(e);
        J acc = J::infinity();
        for (size_t w = 0; w < tab.size(); w++) { unsigned d = (e[w / 8] >> (8 * (w % 8))) & 0xff; if (d) acc = acc.add(tab[w][d - 1]); }
        return acc;
    }
};

static inline void put32(std::vector<uint8_t>& o, uint32_t x) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(x >> (8 * i))); }
static inline void put64(std::vector<uint8_t>& o, uint64_t x) { for (int i = 0; i < 8; i++) o.push_back((uint8_t)(x >> (8 * i))); }
template <class F> static void put_mont(std::vector<uint8_t>& o, const F& x) { const uint8_t* p = (const uint8_t*)x.v; o.insert(o.end(), p, p + sizeof x.v); }
template <class Fq> static void put_g1(std::vector<uint8_t>& o, const AffineT<Fq>& a) {
    if (a.inf) { o.insert(o.end(), 2 * sizeof(Fq), 0); return; }
    put_mont(o, a.x); put_mont(o, a.y);
}
template <class Fq> static void put_g2(std::vector<uint8_t>& o, const AffineT<Fp2T<Fq>>& a) {
    if (a.inf) { o.insert(o.end(), 4 * sizeof(Fq), 0); return; }
    put_mont(o, a.x.c0); put_mont(o, a.x.c1); put_mont(o, a.y.c0); put_mont(o, a.y.c1);
}
static inline void write_sections(const std::string& path, const char* magic, uint32_t version, const std::vector<std::pair<uint32_t, std::vector<uint8_t>>>& secs) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::vector<uint8_t> hdr; hdr.insert(hdr.end(), magic, magic + 4); put32(hdr, version); put32(hdr, (uint32_t)secs.size());
    fwrite(hdr.data(), 1, hdr.size(), f);
    for (auto& s : secs) { std::vector<uint8_t> h; put32(h, s.first); put64(h, s.second.size()); fwrite(h.data(), 1, h.size(), f); fwrite(s.second.data(), 1, s.second.size(), f); }
    fclose(f);
}

template <class C>
static void make_synthetic(int log_m, uint64_t seed, const std::string& zkey_path, const std::string& wtns_path, int threads, size_t n_pub = 1) {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq; typedef typename C::G1 G1; typedef typename C::G2 G2;
    C::init();
    const size_t m = (size_t)1 << log_m, n_inp = n_pub + 1, nc = m - n_inp, n_vars = m;      // n_pub public inputs (default 1): still domain size exactly m
    if (log_m < 2 || n_pub < 1 || n_inp + 1 > m) throw std::runtime_error("log_m must be >= 2 and 1 <= n_public <= m - 2");
    XorShift rng{seed * 2654435761ull + 12345};
    auto rnd = [&] { Fr x; do { x = rand_fp<Fr>(rng); } while (x.is_zero()); return x; };
    // ---- circuit + witness
    std::vector<Fr> ca(nc), cb(nc), w(n_vars);
    std::vector<uint32_t> sb(nc);
    w[0] = Fr::one(); for (size_t i = 1; i <= n_pub; i++) w[i] = rnd();
    for (size_t j = 0; j < nc; j++) {                                        // constraint j defines w[n_pub + 1 + j] from its predecessor and an earlier signal
        ca[j] = rnd(); cb[j] = rnd();
        sb[j] = (uint32_t)(1 + (j * 2654435761ull + 7) % (j + n_pub));      // in [1, j + n_pub]
        w[j + n_pub + 1] = (ca[j] * w[j + n_pub]) * (cb[j] * w[sb[j]]);
    }
    // ---- toxic waste and Lagrange values on H
    Fr tau = rnd(), alpha = rnd(), beta = rnd(), gamma = rnd(), delta = rnd();
    auto dom = groth16_domain<Fr>((size_t)log_m, nc, n_inp);
    Fr tau_m = tau; for (int i = 0; i < log_m; i++) tau_m = tau_m.sqr();
    Fr minv = Fr::from_u64((uint64_t)m).inverse();
    std::vector<Fr> wpow(m), den(m), lag(m);
    wpow[0] = Fr::one(); for (size_t j = 1; j < m; j++) wpow[j] = wpow[j - 1] * dom.omega;
    for (size_t j = 0; j < m; j++) den[j] = tau - wpow[j];
    batch_inverse(den.data(), m);
    Fr zt = tau_m - Fr::one();
    for (size_t j = 0; j < m; j++) lag[j] = zt * minv * wpow[j] * den[j];
    // u_i = sum_j A[j][i] L_j, v_i, w_i (C matrix: C[j][j+2] = 1)
    std::vector<Fr> u(n_vars, Fr::zero()), v(n_vars, Fr::zero()), wc(n_vars, Fr::zero());
    for (size_t j = 0; j < nc; j++) { u[j + n_pub] += ca[j] * lag[j]; v[sb[j]] += cb[j] * lag[j]; wc[j + n_pub + 1] += lag[j]; }
    for (size_t i = 0; i < n_inp; i++) u[i] += lag[nc + i];
    // h exponents on the odd coset
    std::vector<Fr> hden(m), hexp(m);
    for (size_t i = 0; i < m; i++) hden[i] = tau - dom.coset_g * wpow[i];
    batch_inverse(hden.data(), m);
    Fr hfac = (tau_m.sqr() - Fr::one()) * (Fr::from_u64(2) * Fr::from_u64((uint64_t)m) * delta).inverse();
    for (size_t i = 0; i < m; i++) hexp[i] = hfac * dom.coset_g * wpow[i] * hden[i];
    Fr ginv = gamma.inverse(), dinv = delta.inverse();
    // ---- group elements
    FixedBase<G1> fb1(G1::from_affine(C::g1_generator()), Fr::K.bits);
    FixedBase<G2> fb2(G2::from_affine(C::g2_generator()), Fr::K.bits);
    auto g1vec = [&](const std::vector<Fr>& s) { std::vector<typename G1::Affine> o(s.size()); parallel_for(s.size(), threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) o[i] = fb1.mul(s[i]).to_affine(); }); return o; };
    std::vector<Fr> lic(n_vars);
    for (size_t i = 0; i < n_vars; i++) lic[i] = (beta * u[i] + alpha * v[i] + wc[i]) * (i <= n_pub ? ginv : dinv);
    auto a_q = g1vec(u), b1_q = g1vec(v), l_all = g1vec(lic), h_q = g1vec(hexp);
    std::vector<typename G2::Affine> b2_q(n_vars);
    parallel_for(n_vars, threads, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) b2_q[i] = fb2.mul(v[i]).to_affine(); });
    // ---- zkey
    std::vector<std::pair<uint32_t, std::vector<uint8_t>>> secs;
    { std::vector<uint8_t> s; put32(s, 1); secs.push_back({1, s}); }
    {
        std::vector<uint8_t> s;
        put32(s, Fq::N * 8); s.insert(s.end(), (const uint8_t*)Fq::K.p, (const uint8_t*)Fq::K.p + Fq::N * 8);
        put32(s, Fr::N * 8); s.insert(s.end(), (const uint8_t*)Fr::K.p, (const uint8_t*)Fr::K.p + Fr::N * 8);
        put32(s, (uint32_t)n_vars); put32(s, (uint32_t)n_pub); put32(s, (uint32_t)m);
        put_g1<Fq>(s, fb1.mul(alpha).to_affine()); put_g1<Fq>(s, fb1.mul(beta).to_affine());
        put_g2<Fq>(s, fb2.mul(beta).to_affine()); put_g2<Fq>(s, fb2.mul(gamma).to_affine());
        put_g1<Fq>(s, fb1.mul(delta).to_affine()); put_g2<Fq>(s, fb2.mul(delta).to_affine());
        secs.push_back({2, s});
    }
    { std::vector<uint8_t> s; for (size_t i = 0; i <= n_pub; i++) put_g1<Fq>(s, l_all[i]); secs.push_back({3, s}); }
    {
        std::vector<uint8_t> s;
        put32(s, (uint32_t)(2 * nc + n_inp));
        Fr r2 = Fr::from_mont_limbs(Fr::K.r2);
        auto coef = [&](uint32_t mat, uint32_t row, uint32_t sig, const Fr& val) { put32(s, mat); put32(s, row); put32(s, sig); put_mont(s, val * r2); };   // value * R^2 on disk
        for (size_t j = 0; j < nc; j++) { coef(0, (uint32_t)j, (uint32_t)(j + n_pub), ca[j]); coef(1, (uint32_t)j, sb[j], cb[j]); }
        for (size_t i = 0; i < n_inp; i++) coef(0, (uint32_t)(nc + i), (uint32_t)i, Fr::one());
        secs.push_back({4, s});
    }
    { std::vector<uint8_t> s; for (auto& p : a_q) put_g1<Fq>(s, p); secs.push_back({5, s}); }
    { std::vector<uint8_t> s; for (auto& p : b1_q) put_g1<Fq>(s, p); secs.push_back({6, s}); }
    { std::vector<uint8_t> s; for (auto& p : b2_q) put_g2<Fq>(s, p); secs.push_back({7, s}); }
    { std::vector<uint8_t> s; for (size_t i = n_pub + 1; i < n_vars; i++) put_g1<Fq>(s, l_all[i]); secs.push_back({8, s}); }
    { std::vector<uint8_t> s; for (auto& p : h_q) put_g1<Fq>(s, p); secs.push_back({9, s}); }
    write_sections(zkey_path, "zkey", 1, secs);
    // ---- wtns (canonical little-endian values)
    {
        std::vector<std::pair<uint32_t, std::vector<uint8_t>>> ws;
        std::vector<uint8_t> s1; put32(s1, Fr::N * 8); s1.insert(s1.end(), (const uint8_t*)Fr::K.p, (const uint8_t*)Fr::K.p + Fr::N * 8); put32(s1, (uint32_t)n_vars);
        std::vector<uint8_t> s2;
        for (size_t i = 0; i < n_vars; i++) { uint64_t c[Fr::N]; w[i].to_canonical(c); s2.insert(s2.end(), (const uint8_t*)c, (const uint8_t*)c + sizeof c); }
        ws.push_back({1, s1}); ws.push_back({2, s2});
        write_sections(wtns_path, "wtns", 2, ws);
    }
}

}  // namespace orc
