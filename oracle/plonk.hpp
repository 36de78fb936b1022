// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// co-plonk, round 1 (the wire-polynomial commitments): the first slice of SURVEY.md §8 f-2.  It exercises the SAME two hot
// operations as Groth16 (iNTT with the snarkjs root, variable-base MSM over `p_tau`) and — unlike Groth16 — the reference pins
// its exact output: `co-plonk/src/round1.rs:346-383` hard-codes [a]_1, [b]_1, [c]_1 for test_vectors/Plonk/bn254/multiplier2 with
// the deterministic blinding b_i = i.  Restates:
//   `/root/reference/co-circom/circom-types/src/plonk/zkey.rs:83-255,328-425`  (container sections 2..6 and 14, header)
//   `/root/reference/co-circom/co-plonk/src/types.rs:59-100,101-116`          (domain root = snarkjs roots[pow]; public_inputs[0] := 0)
//   `/root/reference/co-circom/co-plonk/src/lib.rs:113-158`                   (get_witness, blind_coefficients)
//   `/root/reference/co-circom/co-plonk/src/round1.rs:118-206,208-238,260-312` (wire polynomials, additions, commitments)
#pragma once
#include "formats.hpp"
#include "ec.hpp"
#include "pairing.hpp"

namespace orc {

// ---- Keccak-256 (original Keccak padding 0x01, as the `sha3::Keccak256` the reference's transcript uses, co-plonk/src/types.rs:20-25) ----
struct Keccak256 {
    uint64_t st[25]; uint8_t buf[136]; size_t fill = 0;
    Keccak256() { memset(st, 0, sizeof st); }
    static uint64_t rol(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }
    void permute() {
        static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                                        0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                                        0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                                        0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
        static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
        for (int r = 0; r < 24; r++) {
            uint64_t C[5], D[5], B[25];
            for (int x = 0; x < 5; x++) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
            for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1);
            for (int i = 0; i < 25; i++) st[i] ^= D[i % 5];
            for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rol(st[x + 5 * y], ROT[x + 5 * y]);
            for (int y = 0; y < 5; y++) for (int x = 0; x < 5; x++) st[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
            st[0] ^= RC[r];
        }
    }
    void absorb_block() { for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, buf + 8 * i, 8); st[i] ^= w; } permute(); fill = 0; }
    void update(const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) { buf[fill++] = p[i]; if (fill == 136) absorb_block(); } }
    void finalize(uint8_t out[32]) { memset(buf + fill, 0, 136 - fill); buf[fill] ^= 0x01; buf[135] ^= 0x80; absorb_block(); memcpy(out, st, 32); }
};

// Keccak256Transcript (co-plonk/src/types.rs:122-176): field elements enter as big-endian canonical bytes; the point at infinity as
// 2 x byte_len zero bytes; the challenge is the digest read big-endian, reduced mod r
template <class C>
struct PlonkTranscript {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq;
    Keccak256 h;
    template <class F> void add_field(const F& v) {
        uint64_t can[F::N]; v.to_canonical(can);
        uint8_t be[F::N * 8];
        for (int i = 0; i < F::N * 8; i++) be[F::N * 8 - 1 - i] = (uint8_t)(can[i / 8] >> (8 * (i % 8)));
        h.update(be, sizeof be);
    }
    void add_scalar(const Fr& s) { add_field(s); }
    void add_point(const AffineT<Fq>& p) {
        if (p.inf) { uint8_t z[2 * Fq::N * 8]; memset(z, 0, sizeof z); h.update(z, sizeof z); return; }
        add_field(p.x); add_field(p.y);
    }
    Fr get_challenge() {
        uint8_t d[32]; h.finalize(d);
        // from_be_bytes_mod_order: Horner over the 32 bytes
        Fr acc = Fr::zero(); const Fr b256 = Fr::from_u64(256);
        for (int i = 0; i < 32; i++) acc = acc * b256 + Fr::from_u64(d[i]);
        return acc;
    }
};

template <class C>
struct PlonkZKey {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq;
    size_t n_vars = 0, n_public = 0, domain_size = 0, power = 0, n_additions = 0, n_constraints = 0;
    struct Addition { uint32_t id1, id2; Fr f1, f2; };
    std::vector<Addition> additions;
    std::vector<uint32_t> map_a, map_b, map_c;
    std::vector<AffineT<Fq>> p_tau;          // domain_size + 6 points (zkey.rs:149-151)
    Fr k1, k2;                                // verifying key (zkey.rs:328-356)
    AffineT<Fq> vk_g1[8];                     // qm, ql, qr, qo, qc, s1, s2, s3
    AffineT<Fp2T<Fq>> x_2;                    // [tau]_2
    std::vector<Fr> sigma_eval[3];            // 4 * domain_size evaluations of sigma1..3 on the extended domain (section 12)
    std::vector<Fr> q_eval[5];                // qm, ql, qr, qo, qc on the extended domain (sections 7..11)
    std::vector<Fr> q_coef[5], sigma_coef[3]; // coefficient forms (round 5)
    std::vector<std::vector<Fr>> lagrange_eval;   // n_public polynomials, 4 * domain_size evaluations each (section 13)
};

template <class C>
static PlonkZKey<C> read_plonk_zkey(const std::string& path) {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq;
    BinSections bf = read_binfile(path);
    if (bf.magic != "zkey") throw std::runtime_error("not a zkey file");
    auto sec = [&](uint32_t id) -> const std::vector<uint8_t>& { auto it = bf.sec.find(id); if (it == bf.sec.end()) throw std::runtime_error("missing zkey section"); return it->second; };
    { Cursor c(sec(1)); if (c.u32() != 2) throw std::runtime_error("not a plonk zkey (protocol id != 2)"); }
    PlonkZKey<C> z;
    {   // header, zkey.rs:373-424
        Cursor h(sec(2));
        if (h.u32() != Fq::N * 8) throw std::runtime_error("unexpected base field byte size");
        uint64_t q[Fq::N]; h.bytes(q, sizeof q);
        if (raw_cmp<Fq::N>(q, Fq::K.p) != 0) throw std::runtime_error("invalid base prime in header");
        if (h.u32() != Fr::N * 8) throw std::runtime_error("unexpected scalar field byte size");
        uint64_t r[Fr::N]; h.bytes(r, sizeof r);
        if (raw_cmp<Fr::N>(r, Fr::K.p) != 0) throw std::runtime_error("invalid scalar prime in header");
        z.n_vars = h.u32(); z.n_public = h.u32(); z.domain_size = h.u32(); z.n_additions = h.u32(); z.n_constraints = h.u32();
        if (!z.domain_size || (z.domain_size & (z.domain_size - 1))) throw std::runtime_error("invalid domain size");
        while (((size_t)1 << z.power) < z.domain_size) z.power++;
        z.k1 = read_mont<Fr>(h); z.k2 = read_mont<Fr>(h);
        for (int i = 0; i < 8; i++) z.vk_g1[i] = read_g1<Fq>(h);
        z.x_2 = read_g2<Fq>(h);
    }
    {   // section 12 = sigma1 | sigma2 | sigma3, each domain_size coefficients followed by 4 * domain_size evaluations (zkey.rs:170-180,116-135)
        Cursor c(sec(12));
        for (int k = 0; k < 3; k++) {
            z.sigma_coef[k].resize(z.domain_size);
            for (auto& v : z.sigma_coef[k]) v = read_mont<Fr>(c);
            z.sigma_eval[k].resize(4 * z.domain_size);
            for (auto& v : z.sigma_eval[k]) v = read_mont<Fr>(c);
        }
    }
    for (int k = 0; k < 5; k++) {   // zkey.rs:116-135: coefficients then extended evaluations
        Cursor c(sec(7 + k));
        z.q_coef[k].resize(z.domain_size);
        for (auto& v : z.q_coef[k]) v = read_mont<Fr>(c);
        z.q_eval[k].resize(4 * z.domain_size);
        for (auto& v : z.q_eval[k]) v = read_mont<Fr>(c);
    }
    {
        Cursor c(sec(13));
        z.lagrange_eval.resize(z.n_public);
        for (auto& l : z.lagrange_eval) { for (size_t i = 0; i < z.domain_size; i++) read_mont<Fr>(c); l.resize(4 * z.domain_size); for (auto& v : l) v = read_mont<Fr>(c); }
    }
    { Cursor c(sec(3)); z.additions.resize(z.n_additions); for (auto& a : z.additions) { a.id1 = c.u32(); a.id2 = c.u32(); a.f1 = read_mont<Fr>(c); a.f2 = read_mont<Fr>(c); } }
    auto id_map = [&](uint32_t id) { Cursor c(sec(id)); std::vector<uint32_t> m(z.n_constraints); for (auto& v : m) v = c.u32(); return m; };
    z.map_a = id_map(4); z.map_b = id_map(5); z.map_c = id_map(6);
    { Cursor c(sec(14)); z.p_tau.resize(z.domain_size + 6); for (auto& p : z.p_tau) p = read_g1<Fq>(c); }
    return z;
}

// plain driver; `full_witness` is the Groth16-style witness (leading constant one); blind[0..6) = b_1..b_6 of round 1
template <class C>
static std::vector<AffineT<typename C::Fq>> plonk_round1_plain(const PlonkZKey<C>& z, const std::vector<typename C::Fr>& full_witness, const typename C::Fr* blind,
                                                                 std::vector<typename C::Fr>* polys_out = nullptr) {
    typedef typename C::Fr Fr; typedef typename C::G1 G1;
    if (full_witness.size() + z.n_additions != z.n_vars) throw std::runtime_error("witness length does not match the zkey");
    std::vector<Fr> w(full_witness);
    w[0] = Fr::zero();                                                        // types.rs:107-109: snarkjs writes 0 for the constant
    auto get = [&](size_t idx) -> Fr { if (idx >= z.n_vars || idx >= w.size()) throw std::runtime_error("corrupted witness index"); return w[idx]; };   // lib.rs:113-137
    for (const auto& a : z.additions) w.push_back(get(a.id1) * a.f1 + get(a.id2) * a.f2);                                 // round1.rs:208-238
    const size_t n = z.domain_size;
    const Fr omega = roots_of_unity<Fr>().roots[z.power];                      // types.rs:70-84
    std::vector<AffineT<typename C::Fq>> commits;
    const std::vector<uint32_t>* maps[3] = {&z.map_a, &z.map_b, &z.map_c};
    for (int k = 0; k < 3; k++) {
        std::vector<Fr> buf(n, Fr::zero());
        for (size_t i = 0; i < z.n_constraints; i++) buf[i] = get((*maps[k])[i]);                                         // round1.rs:134-157
        ntt_inverse(buf.data(), n, omega);                                                                                // :170-172
        std::vector<Fr> poly(buf);
        const Fr b_hi = blind[2 * k], b_lo = blind[2 * k + 1];                 // coeff_rev = [b_hi, b_lo]; reversed: b_lo first (lib.rs:140-158)
        poly[0] = poly[0] - b_lo; poly[1] = poly[1] - b_hi;
        poly.push_back(b_lo); poly.push_back(b_hi);
        if (poly.size() > z.p_tau.size()) throw std::runtime_error("polynomial degree too large");
        commits.push_back(msm_naive<G1>(z.p_tau.data(), poly.data(), poly.size()).to_affine());                          // round1.rs:276-290
        if (polys_out) polys_out->insert(polys_out->end(), poly.begin(), poly.end());
    }
    return commits;
}

// Round 2 (co-plonk/src/round2.rs:146-298) for a single-component driver: challenges beta, gamma from the transcript, the grand
// product z, its blinded coefficient form and [z]_1.  `round1_commits` = opened [a]_1, [b]_1, [c]_1; blind[6..9) = b_7, b_8, b_9.
template <class C>
struct PlonkRound2 { typename C::Fr beta, gamma; AffineT<typename C::Fq> commit_z; std::vector<typename C::Fr> poly_z; };
template <class C>
static PlonkRound2<C> plonk_round2_plain(const PlonkZKey<C>& z, const std::vector<typename C::Fr>& full_witness, const typename C::Fr* blind,
                                         const std::vector<AffineT<typename C::Fq>>& round1_commits) {
    typedef typename C::Fr Fr; typedef typename C::G1 G1;
    std::vector<Fr> w(full_witness);
    w[0] = Fr::zero();
    auto get = [&](size_t idx) -> Fr { if (idx >= w.size()) throw std::runtime_error("corrupted witness index"); return w[idx]; };
    for (const auto& a : z.additions) w.push_back(get(a.id1) * a.f1 + get(a.id2) * a.f2);
    const size_t n = z.domain_size;
    PlonkRound2<C> out;
    {   // round2.rs:243-263: vk points, public inputs without the leading entry, the round-1 commitments
        PlonkTranscript<C> t;
        for (int i = 0; i < 8; i++) t.add_point(z.vk_g1[i]);
        for (size_t i = 1; i <= z.n_public; i++) t.add_scalar(w[i]);
        for (const auto& cm : round1_commits) t.add_point(cm);
        out.beta = t.get_challenge();
        PlonkTranscript<C> t2; t2.add_scalar(out.beta);
        out.gamma = t2.get_challenge();
    }
    const Fr omega = roots_of_unity<Fr>().roots[z.power];
    std::vector<Fr> num(n), den(n);
    Fr wv = Fr::one();
    for (size_t i = 0; i < n; i++) {                                                        // :162-210
        const Fr a = i < z.n_constraints ? get(z.map_a[i]) : Fr::zero(), b = i < z.n_constraints ? get(z.map_b[i]) : Fr::zero(), c = i < z.n_constraints ? get(z.map_c[i]) : Fr::zero();
        const Fr betaw = out.beta * wv;
        num[i] = (a + betaw + out.gamma) * (b + z.k1 * betaw + out.gamma) * (c + z.k2 * betaw + out.gamma);
        den[i] = (a + out.beta * z.sigma_eval[0][4 * i] + out.gamma) * (b + out.beta * z.sigma_eval[1][4 * i] + out.gamma) * (c + out.beta * z.sigma_eval[2][4 * i] + out.gamma);
        wv = wv * omega;
    }
    for (size_t i = 1; i < n; i++) { num[i] = num[i] * num[i - 1]; den[i] = den[i] * den[i - 1]; }   // array_prod_mul (:18-41), single component
    std::vector<Fr> zb(n);
    for (size_t i = 0; i < n; i++) zb[(i + 1) % n] = num[i] * den[i].inverse();                         // :229-231 incl. rotate_right(1)
    ntt_inverse(zb.data(), n, omega);
    const Fr b6 = blind[6], b7 = blind[7], b8 = blind[8];                                               // coeff_rev = [b6, b7, b8] (lib.rs:140-158)
    zb[0] = zb[0] - b8; zb[1] = zb[1] - b7; zb[2] = zb[2] - b6;
    zb.push_back(b8); zb.push_back(b7); zb.push_back(b6);
    if (zb.size() > z.p_tau.size()) throw std::runtime_error("polynomial degree too large");
    out.commit_z = msm_naive<G1>(z.p_tau.data(), zb.data(), zb.size()).to_affine();
    out.poly_z = zb;
    return out;
}

// ---- the plain-driver prover, round by round, with the state the reference threads through Round1..Round5 ------------------------------
template <class C>
struct PlonkPlainProver {
    typedef typename C::Fr Fr; typedef typename C::Fq Fq; typedef typename C::G1 G1;
    const PlonkZKey<C>& z;
    size_t n;
    std::vector<Fr> w;                         // witness with the additions appended, w[0] = 0
    Fr b[11];                                  // Round1Challenges::b
    Fr omega, omega4;                          // roots[pow], roots[pow + 2] (types.rs:70-97)
    std::vector<Fr> buf[3], poly[3], eval[3];  // wire values, blinded coefficients (n + 2), evaluations of the UNBLINDED polynomial on 4n points
    std::vector<Fr> poly_z, eval_z;
    AffineT<Fq> commit[3], commit_z, commit_t[3];
    Fr beta, gamma, alpha, xi, v[5];
    std::vector<Fr> t1, t2, t3;
    Fr eval_a, eval_b, eval_c, eval_zw, eval_s1, eval_s2;
    std::vector<Fr> wxi, wxiw;
    AffineT<Fq> commit_wxi, commit_wxiw;

    PlonkPlainProver(const PlonkZKey<C>& zk, const std::vector<Fr>& full_witness, const Fr* blind) : z(zk), n(zk.domain_size), w(full_witness) {
        if (full_witness.size() + z.n_additions != z.n_vars) throw std::runtime_error("witness length does not match the zkey");
        for (int i = 0; i < 11; i++) b[i] = blind[i];
        w[0] = Fr::zero();
        for (const auto& a : z.additions) w.push_back(w.at(a.id1) * a.f1 + w.at(a.id2) * a.f2);
        auto rt = roots_of_unity<Fr>();
        omega = rt.roots[z.power]; omega4 = rt.roots[z.power + 2];
    }
    AffineT<Fq> commit_poly(const std::vector<Fr>& p) const {
        if (p.size() > z.p_tau.size()) throw std::runtime_error("polynomial degree too large");
        return msm_naive<G1>(z.p_tau.data(), p.data(), p.size()).to_affine();
    }
    std::vector<Fr> extended_eval(const std::vector<Fr>& coeffs) const { std::vector<Fr> e(coeffs); e.resize(4 * n, Fr::zero()); ntt_forward(e.data(), 4 * n, omega4); return e; }

    void round1() {                                                                                   // round1.rs:118-312
        const std::vector<uint32_t>* maps[3] = {&z.map_a, &z.map_b, &z.map_c};
        for (int k = 0; k < 3; k++) {
            buf[k].assign(n, Fr::zero());
            for (size_t i = 0; i < z.n_constraints; i++) buf[k][i] = w.at((*maps[k])[i]);
            std::vector<Fr> p(buf[k]);
            ntt_inverse(p.data(), n, omega);
            eval[k] = extended_eval(p);                                                               // :174-177 (before blinding)
            p[0] = p[0] - b[2 * k + 1]; p[1] = p[1] - b[2 * k]; p.push_back(b[2 * k + 1]); p.push_back(b[2 * k]);
            poly[k] = p;
            commit[k] = commit_poly(p);
        }
    }
    void round2() {                                                                                   // round2.rs:146-298
        {
            PlonkTranscript<C> t;
            for (int i = 0; i < 8; i++) t.add_point(z.vk_g1[i]);
            for (size_t i = 1; i <= z.n_public; i++) t.add_scalar(w[i]);
            for (int k = 0; k < 3; k++) t.add_point(commit[k]);
            beta = t.get_challenge();
            PlonkTranscript<C> t2; t2.add_scalar(beta); gamma = t2.get_challenge();
        }
        std::vector<Fr> num(n), den(n);
        Fr wv = Fr::one();
        for (size_t i = 0; i < n; i++) {
            const Fr bw = beta * wv;
            num[i] = (buf[0][i] + bw + gamma) * (buf[1][i] + z.k1 * bw + gamma) * (buf[2][i] + z.k2 * bw + gamma);
            den[i] = (buf[0][i] + beta * z.sigma_eval[0][4 * i] + gamma) * (buf[1][i] + beta * z.sigma_eval[1][4 * i] + gamma) * (buf[2][i] + beta * z.sigma_eval[2][4 * i] + gamma);
            wv = wv * omega;
        }
        for (size_t i = 1; i < n; i++) { num[i] = num[i] * num[i - 1]; den[i] = den[i] * den[i - 1]; }
        std::vector<Fr> p(n);
        for (size_t i = 0; i < n; i++) p[(i + 1) % n] = num[i] * den[i].inverse();
        ntt_inverse(p.data(), n, omega);
        eval_z = extended_eval(p);                                                                    // :238
        p[0] = p[0] - b[8]; p[1] = p[1] - b[7]; p[2] = p[2] - b[6]; p.push_back(b[8]); p.push_back(b[7]); p.push_back(b[6]);
        poly_z = p;
        commit_z = commit_poly(p);
    }
    // round3.rs:234-488, one component.  The quotient is split into its low part (from evaluations of the unblinded polynomials) and the
    // part produced by the blinding factors (the `...z` vectors), exactly as the reference does, so that the coefficient vectors coincide.
    void round3() {
        {
            PlonkTranscript<C> t; t.add_scalar(beta); t.add_scalar(gamma); t.add_point(commit_z);
            alpha = t.get_challenge();
        }
        const Fr alpha2 = alpha * alpha;
        const size_t N = 4 * n;
        const Fr one = Fr::one(), zero = Fr::zero(), two = one + one;
        const Fr w2r = roots_of_unity<Fr>().roots[2];                                                  // root_of_unity_2
        const Fr Z1[4] = {zero, zero - one + w2r, zero - two, zero - one - w2r};                      // get_z1..3 (:203-232)
        const Fr Z2[4] = {zero, (zero - two) * w2r, two * two, zero - (zero - two) * w2r};
        const Fr Z3[4] = {zero, two + two * w2r, zero - two * two * two, two - two * w2r};
        std::vector<Fr> tv(N), tzv(N);
        Fr wv = one;
        for (size_t i = 0; i < N; i++) {
            const Fr a = eval[0][i], bb = eval[1][i], c = eval[2][i], zz = eval_z[i], zw = eval_z[(i + 4) % N];
            const Fr ap = b[1] + b[0] * wv, bp = b[3] + b[2] * wv, cp = b[5] + b[4] * wv;
            const Fr w2 = wv * wv, zp = b[6] * w2 + b[7] * wv + b[8];
            const Fr ww = wv * omega, zwp = b[6] * ww * ww + b[7] * ww + b[8];
            const int mi = (int)(i % 4);
            const Fr a_b = a * bb, a_bp = a * bp, ap_b = bb * ap, ap_bp = ap * bp;
            Fr a0 = a_bp + ap_b + Z1[mi] * ap_bp;
            Fr e1 = z.q_eval[0][i] * a_b + a * z.q_eval[1][i] + bb * z.q_eval[2][i] + c * z.q_eval[3][i];
            Fr e1z = z.q_eval[0][i] * a0 + ap * z.q_eval[1][i] + bp * z.q_eval[2][i] + cp * z.q_eval[3][i];
            Fr pi = zero;
            for (size_t j = 0; j < z.lagrange_eval.size(); j++) pi = pi - z.lagrange_eval[j][i] * buf[0][j];
            e1 = e1 + pi + z.q_eval[4][i];
            const Fr bw = beta * wv;
            auto mul4 = [&](const Fr& A, const Fr& B, const Fr& Cc, const Fr& D, const Fr& Dp, Fr& r, Fr& rz) {   // mul4vec + mul4vec_post (:17-72)
                const Fr AB = A * B, ABp = A * bp, ApB = ap * B, ApBp = ap * bp, CD = Cc * D, CDp = Cc * Dp, CpD = cp * D, CpDp = cp * Dp;
                r = AB * CD;
                const Fr r0 = ApB * CD + ABp * CD + AB * CpD + AB * CDp;
                const Fr r1 = ApBp * CD + ApB * CpD + ApB * CDp + ABp * CpD + ABp * CDp + AB * CpDp;
                const Fr r2 = ABp * CpDp + ApB * CpDp + ApBp * CDp + ApBp * CpD;
                const Fr r3 = ApBp * CpDp;
                rz = r0 + Z1[mi] * r1 + Z2[mi] * r2 + Z3[mi] * r3;
            };
            Fr e2, e2z, e3, e3z;
            mul4(a + bw + gamma, bb + bw * z.k1 + gamma, c + bw * z.k2 + gamma, zz, zp, e2, e2z);
            mul4(a + z.sigma_eval[0][i] * beta + gamma, bb + z.sigma_eval[1][i] * beta + gamma, c + z.sigma_eval[2][i] * beta + gamma, zw, zwp, e3, e3z);
            const Fr l1 = z.lagrange_eval.at(0)[i];
            const Fr e4 = (zz - one) * l1 * alpha2, e4z = zp * l1 * alpha2;
            tv[i] = e1 + e2 * alpha - e3 * alpha + e4;
            tzv[i] = e1z + e2z * alpha - e3z * alpha + e4z;
            wv = wv * omega4;
        }
        ntt_inverse(tv.data(), N, omega4);
        for (size_t i = 0; i < n; i++) tv[i] = zero - tv[i];                                           // neg_vec_in_place_limit (:443)
        for (size_t i = n; i < N; i++) tv[i] = tv[i - n] - tv[i];                                      // division by X^n - 1 (:445-450)
        ntt_inverse(tzv.data(), N, omega4);
        for (size_t i = 0; i < N; i++) tv[i] = tv[i] + tzv[i];
        t1.assign(tv.begin(), tv.begin() + n); t2.assign(tv.begin() + n, tv.begin() + 2 * n); t3.assign(tv.begin() + 2 * n, tv.begin() + 3 * n + 6);
        t1.push_back(b[9]);
        t2[0] = t2[0] - b[9]; t2.push_back(b[10]);
        t3[0] = t3[0] - b[10];
        commit_t[0] = commit_poly(t1); commit_t[1] = commit_poly(t2); commit_t[2] = commit_poly(t3);
    }
    static Fr horner(const std::vector<Fr>& p, const Fr& x) { Fr acc = Fr::zero(); for (size_t i = p.size(); i-- > 0;) acc = acc * x + p[i]; return acc; }   // evaluate_poly_public
    void round4() {                                                                                   // round4.rs:115-160
        { PlonkTranscript<C> t; t.add_scalar(alpha); for (int k = 0; k < 3; k++) t.add_point(commit_t[k]); xi = t.get_challenge(); }
        eval_a = horner(poly[0], xi); eval_b = horner(poly[1], xi); eval_c = horner(poly[2], xi);
        eval_zw = horner(poly_z, xi * omega);
        eval_s1 = horner(z.sigma_coef[0], xi); eval_s2 = horner(z.sigma_coef[1], xi);
    }
    static void div_by_zerofier1(std::vector<Fr>& p, const Fr& beta_) {                               // round5.rs:97-115 with n = 1
        const Fr inv = beta_.inverse();
        p[0] = p[0] * (Fr::zero() - inv);
        for (size_t i = 1; i < p.size(); i++) p[i] = (p[i - 1] - p[i]) * inv;
        p.pop_back();
    }
    void round5() {                                                                                   // round5.rs:143-365
        {
            PlonkTranscript<C> t; t.add_scalar(xi); t.add_scalar(eval_a); t.add_scalar(eval_b); t.add_scalar(eval_c); t.add_scalar(eval_s1); t.add_scalar(eval_s2); t.add_scalar(eval_zw);
            v[0] = t.get_challenge(); for (int i = 1; i < 5; i++) v[i] = v[i - 1] * v[0];
        }
        const Fr one = Fr::one();
        Fr xin = xi; for (size_t i = 0; i < z.power; i++) xin = xin * xin;                             // lib.rs:160-184
        const Fr zh = xin - one;
        std::vector<Fr> l; { Fr wv = one; const Fr nn = Fr::from_u64((uint64_t)n); for (size_t i = 0; i < std::max<size_t>(1, z.n_public); i++) { l.push_back(wv * zh * (nn * (xi - wv)).inverse()); wv = wv * omega; } }
        Fr eval_pi = Fr::zero(); for (size_t i = 0; i < z.n_public && i < l.size(); i++) eval_pi = eval_pi - l[i] * w[i + 1];   // calculate_pi (:186-195)
        const Fr coef_ab = eval_a * eval_b, betaxi = beta * xi;
        const Fr e2 = (eval_a + betaxi + gamma) * (eval_b + betaxi * z.k1 + gamma) * (eval_c + betaxi * z.k2 + gamma) * alpha;
        const Fr e3 = (eval_a + beta * eval_s1 + gamma) * (eval_b + beta * eval_s2 + gamma) * eval_zw * alpha;
        const Fr e4 = alpha * alpha * l[0], e24 = e2 + e4;
        const size_t len = n + 6;
        std::vector<Fr> r(len, Fr::zero());
        for (size_t i = 0; i < poly_z.size(); i++) r[i] = e24 * poly_z[i];
        const Fr me3b = Fr::zero() - e3 * beta;
        for (size_t i = 0; i < n; i++) r[i] = r[i] + z.q_coef[0][i] * coef_ab + z.q_coef[1][i] * eval_a + z.q_coef[2][i] * eval_b + z.q_coef[3][i] * eval_c + z.q_coef[4][i] + z.sigma_coef[2][i] * me3b;
        const Fr xin2 = xin * xin;
        for (size_t i = 0; i < len; i++) {
            Fr tmp = (i < t3.size() ? t3[i] * xin2 : Fr::zero()) + (i < t2.size() ? t2[i] * xin : Fr::zero()) + (i < t1.size() ? t1[i] : Fr::zero());
            r[i] = r[i] - tmp * zh;
        }
        r[0] = r[0] + (eval_pi - e3 * (eval_c + gamma) - e4);
        wxi = r;                                                                                       // compute_wxi (:263-311)
        for (size_t i = 0; i < poly[0].size(); i++) wxi[i] = wxi[i] + v[0] * poly[0][i] + v[1] * poly[1][i] + v[2] * poly[2][i];
        for (size_t i = 0; i < n; i++) wxi[i] = wxi[i] + v[3] * z.sigma_coef[0][i] + v[4] * z.sigma_coef[1][i];
        wxi[0] = wxi[0] - v[0] * eval_a - v[1] * eval_b - v[2] * eval_c - v[3] * eval_s1 - v[4] * eval_s2;
        div_by_zerofier1(wxi, xi);
        wxiw = poly_z; wxiw[0] = wxiw[0] - eval_zw;                                                    // compute_wxiw (:314-327)
        div_by_zerofier1(wxiw, xi * omega);
        commit_wxi = commit_poly(wxi); commit_wxiw = commit_poly(wxiw);
    }
};

// The verifier (co-plonk/src/plonk.rs:41-131 challenges, :133-271 checks), with the verifying key taken from the zkey header.
// proof: nine commitments in the order a, b, c, z, t1, t2, t3, wxi, wxiw and the evaluations a, b, c, s1, s2, zw.
template <class C>
static bool plonk_verify(const PlonkZKey<C>& z, const AffineT<typename C::Fq>* cm, const typename C::Fr* ev, const std::vector<typename C::Fr>& public_inputs) {
    typedef typename C::Fr Fr; typedef typename C::G1 G1;
    if (public_inputs.size() != z.n_public) return false;
    for (int i = 0; i < 9; i++) if (!cm[i].inf && !G1::on_curve(cm[i])) return false;
    const Fr eval_a = ev[0], eval_b = ev[1], eval_c = ev[2], eval_s1 = ev[3], eval_s2 = ev[4], eval_zw = ev[5];
    Fr beta, gamma, alpha, xi, v[5], u;
    { PlonkTranscript<C> t; for (int i = 0; i < 8; i++) t.add_point(z.vk_g1[i]); for (const Fr& p : public_inputs) t.add_scalar(p); for (int i = 0; i < 3; i++) t.add_point(cm[i]); beta = t.get_challenge(); }
    { PlonkTranscript<C> t; t.add_scalar(beta); gamma = t.get_challenge(); }
    { PlonkTranscript<C> t; t.add_scalar(beta); t.add_scalar(gamma); t.add_point(cm[3]); alpha = t.get_challenge(); }
    { PlonkTranscript<C> t; t.add_scalar(alpha); for (int i = 4; i < 7; i++) t.add_point(cm[i]); xi = t.get_challenge(); }
    { PlonkTranscript<C> t; t.add_scalar(xi); for (int i = 0; i < 6; i++) t.add_scalar(ev[i]); v[0] = t.get_challenge(); for (int i = 1; i < 5; i++) v[i] = v[i - 1] * v[0]; }
    { PlonkTranscript<C> t; t.add_point(cm[7]); t.add_point(cm[8]); u = t.get_challenge(); }
    const Fr one = Fr::one();
    const Fr omega = roots_of_unity<Fr>().roots[z.power];
    Fr xin = xi; for (size_t i = 0; i < z.power; i++) xin = xin * xin;
    const Fr zh = xin - one;
    std::vector<Fr> l; { Fr wv = one; const Fr nn = Fr::from_u64((uint64_t)z.domain_size); for (size_t i = 0; i < std::max<size_t>(1, z.n_public); i++) { l.push_back(wv * zh * (nn * (xi - wv)).inverse()); wv = wv * omega; } }
    Fr pi = Fr::zero(); for (size_t i = 0; i < public_inputs.size(); i++) pi = pi - l[i] * public_inputs[i];
    // calculate_r0_d (:173-224)
    const Fr e2 = alpha * alpha * l[0];
    const Fr e3a = eval_a + eval_s1 * beta + gamma, e3b = eval_b + eval_s2 * beta + gamma;
    const Fr r0 = pi - e2 - e3a * e3b * (eval_c + gamma) * eval_zw * alpha;
    auto P = [&](int i) { return G1::from_affine(cm[i]); };
    auto VK = [&](int i) { return G1::from_affine(z.vk_g1[i]); };
    G1 d1 = scalar_mul(VK(0), eval_a * eval_b).add(scalar_mul(VK(1), eval_a)).add(scalar_mul(VK(2), eval_b)).add(scalar_mul(VK(3), eval_c)).add(VK(4));
    const Fr betaxi = beta * xi;
    const Fr d2a = (eval_a + betaxi + gamma) * (eval_b + betaxi * z.k1 + gamma) * (eval_c + betaxi * z.k2 + gamma) * alpha;
    G1 d2 = scalar_mul(P(3), d2a + e2 + u);
    G1 d3 = scalar_mul(VK(7), e3a * e3b * (alpha * beta * eval_zw));
    G1 d4 = scalar_mul(P(4).add(scalar_mul(P(5), xin)).add(scalar_mul(P(6), xin * xin)), zh);
    G1 d = d1.add(d2).add(d3.neg()).add(d4.neg());
    const Fr e_s = v[0] * eval_a + v[1] * eval_b + v[2] * eval_c + v[3] * eval_s1 + v[4] * eval_s2 + u * eval_zw - r0;   // calculate_e (:226-239)
    G1 e = scalar_mul(G1::from_affine(C::g1_generator()), e_s);
    G1 f = d.add(scalar_mul(P(0), v[0])).add(scalar_mul(P(1), v[1])).add(scalar_mul(P(2), v[2])).add(scalar_mul(VK(5), v[3])).add(scalar_mul(VK(6), v[4]));   // calculate_f
    // valid_pairing (:254-271): e(Wxi + u Wxiw, X_2) == e(xi Wxi + u xi omega Wxiw - E + F, G2)
    G1 a1 = P(7).add(scalar_mul(P(8), u));
    G1 b1 = scalar_mul(P(7), xi).add(scalar_mul(P(8), u * xi * omega)).add(e.neg()).add(f);
    auto negp = [](typename G1::Affine a) { if (!a.inf) a.y = -a.y; return a; };
    Fp12T<C> m = miller_tate<C>(a1.to_affine(), z.x_2) * miller_tate<C>(negp(b1.to_affine()), C::g2_generator());
    return final_exp<C>(m) == Fp12T<C>::one();
}

}  // namespace orc
