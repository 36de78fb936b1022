// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
// extern "C" surface over the CPU restatement so that tests/ (ctypes), __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can drive it.  Every buffer uses the product ABI's data convention:
// field elements = N x u64 little-endian Montgomery limbs; G1 affine = x||y, G2 affine = x.c0||x.c1||y.c0||y.c1
// with (0,0) = infinity (the packed zkey encoding, `/root/reference/co-circom/circom-types/src/traits.rs:107-155`).
#include "pairing.hpp"
#include "bench.hpp"
#include "synth.hpp"
#include "plonk.hpp"
#include "shamir.hpp"
#include "rngs.hpp"
#include <chrono>

using namespace orc;

namespace {
thread_local std::string g_err;
template <class F> F ld(const uint64_t* p) { return F::from_mont_limbs(p); }
template <class F> void st(uint64_t* p, const F& v) { memcpy(p, v.v, sizeof v.v); }

template <class Fq> AffineT<Fq> ld_g1(const uint64_t* p) {
    Fq x = ld<Fq>(p), y = ld<Fq>(p + Fq::N);
    if (x.is_zero() && y.is_zero()) return AffineT<Fq>::infinity();
    return {x, y, false};
}
template <class Fq> void st_g1(uint64_t* p, const AffineT<Fq>& a) {
    if (a.inf) { memset(p, 0, 2 * Fq::N * 8); return; }
    st(p, a.x); st(p + Fq::N, a.y);
}
template <class Fq> AffineT<Fp2T<Fq>> ld_g2(const uint64_t* p) {
    Fp2T<Fq> x = {ld<Fq>(p), ld<Fq>(p + Fq::N)}, y = {ld<Fq>(p + 2 * Fq::N), ld<Fq>(p + 3 * Fq::N)};
    if (x.is_zero() && y.is_zero()) return AffineT<Fp2T<Fq>>::infinity();
    return {x, y, false};
}
template <class Fq> void st_g2(uint64_t* p, const AffineT<Fp2T<Fq>>& a) {
    if (a.inf) { memset(p, 0, 4 * Fq::N * 8); return; }
    st(p, a.x.c0); st(p + Fq::N, a.x.c1); st(p + 2 * Fq::N, a.y.c0); st(p + 3 * Fq::N, a.y.c1);
}
template <class C> void st_proof(uint64_t* p, const Proof<C>& pf) {
    const int n = C::Fq::N;
    st_g1<typename C::Fq>(p, pf.a); st_g2<typename C::Fq>(p + 2 * n, pf.b); st_g1<typename C::Fq>(p + 6 * n, pf.c);
}
template <class C> Proof<C> ld_proof(const uint64_t* p) {
    const int n = C::Fq::N;
    return {ld_g1<typename C::Fq>(p), ld_g2<typename C::Fq>(p + 2 * n), ld_g1<typename C::Fq>(p + 6 * n)};
}

struct ZKeyHandle { int curve; ZKey<Bn254>* bn = nullptr; ZKey<Bls12_381>* bls = nullptr; };

#define DISPATCH(curve, ...)                                    \
    try {                                                       \
        if ((curve) == 0) { typedef Bn254 C; C::init(); __VA_ARGS__; } \
        else if ((curve) == 1) { typedef Bls12_381 C; C::init(); __VA_ARGS__; } \
        else { g_err = "bad curve id"; return -1; }             \
    } catch (const std::exception& e) { g_err = e.what(); return -2; }
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// which: 0 = Fr, 1 = Fq
int orc_field_limbs(int curve, int which) { return curve == 1 && which == 1 ? 6 : 4; }

int orc_from_dec(int curve, int which, const char* dec, uint64_t* out) {
    DISPATCH(curve, { if (which == 0) st(out, C::Fr::from_dec(dec)); else st(out, C::Fq::from_dec(dec)); });
    return 0;
}
int orc_to_dec(int curve, int which, const uint64_t* in, char* out, size_t cap) {
    DISPATCH(curve, {
        std::string s = which == 0 ? ld<typename C::Fr>(in).to_dec() : ld<typename C::Fq>(in).to_dec();
        if (s.size() + 1 > cap) { g_err = "buffer too small"; return -3; }
        memcpy(out, s.c_str(), s.size() + 1);
    });
    return 0;
}
// op: 0 add, 1 sub, 2 mul (elementwise over n elements of Fr (which=0) or Fq (which=1))
int orc_field_op(int curve, int which, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    DISPATCH(curve, {
        auto run = [&](auto tag) {
            typedef decltype(tag) F;
            for (size_t i = 0; i < n; i++) {
                F x = ld<F>(a + i * F::N), y = ld<F>(b + i * F::N);
                st(out + i * F::N, op == 0 ? x + y : op == 1 ? x - y : x * y);
            }
        };
        if (which == 0) run(typename C::Fr()); else run(typename C::Fq());
    });
    return 0;
}
int orc_field_inverse(int curve, int which, const uint64_t* a, uint64_t* out) {
    DISPATCH(curve, { if (which == 0) st(out, ld<typename C::Fr>(a).inverse()); else st(out, ld<typename C::Fq>(a).inverse()); });
    return 0;
}
// out_roots must hold (two_adicity+1) elements; returns two-adicity through *two_adicity
int orc_roots_of_unity(int curve, uint64_t* out_q, uint64_t* out_roots, int* two_adicity) {
    DISPATCH(curve, {
        auto rt = roots_of_unity<typename C::Fr>();
        st(out_q, rt.q); *two_adicity = rt.two_adicity;
        for (size_t i = 0; i < rt.roots.size(); i++) st(out_roots + i * 4, rt.roots[i]);
    });
    return 0;
}
int orc_groth16_domain(int curve, size_t pow, size_t num_constraints, size_t num_inputs, uint64_t* omega, uint64_t* coset_g, size_t* m) {
    DISPATCH(curve, {
        auto d = groth16_domain<typename C::Fr>(pow, num_constraints, num_inputs);
        st(omega, d.omega); st(coset_g, d.coset_g); *m = d.m;
    });
    return 0;
}
int orc_ntt(int curve, uint64_t* data, size_t n, const uint64_t* omega, int inverse) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr;
        Fr* v = reinterpret_cast<Fr*>(data);
        if (inverse) ntt_inverse(v, n, ld<Fr>(omega)); else ntt_forward(v, n, ld<Fr>(omega));
    });
    return 0;
}
int orc_dft_naive(int curve, const uint64_t* in, uint64_t* out, size_t n, const uint64_t* omega) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr;
        auto r = dft_naive(reinterpret_cast<const Fr*>(in), n, ld<Fr>(omega));
        memcpy(out, r.data(), n * sizeof(Fr));
    });
    return 0;
}
int orc_distribute_powers(int curve, uint64_t* data, size_t n, const uint64_t* g, const uint64_t* c) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr;
        Fr pw = ld<Fr>(c), gg = ld<Fr>(g);
        Fr* v = reinterpret_cast<Fr*>(data);
        for (size_t i = 0; i < n; i++) { v[i] = v[i] * pw; pw = pw * gg; }
    });
    return 0;
}
// group: 0 = G1, 1 = G2. algo: 0 = arkworks-style Pippenger, 1 = naive double-and-add. out = packed affine.
int orc_msm(int curve, int group, int algo, const uint64_t* points, const uint64_t* scalars, size_t n, int threads, uint64_t* out_affine) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        const Fr* sc = reinterpret_cast<const Fr*>(scalars);
        if (group == 0) {
            std::vector<typename C::G1::Affine> b(n);
            for (size_t i = 0; i < n; i++) b[i] = ld_g1<Fq>(points + i * 2 * Fq::N);
            auto r = n == 0 ? C::G1::infinity() : algo == 0 ? msm_pippenger<typename C::G1, Fr>(b.data(), sc, n, threads) : msm_naive<typename C::G1, Fr>(b.data(), sc, n);
            st_g1<Fq>(out_affine, r.to_affine());
        } else {
            std::vector<typename C::G2::Affine> b(n);
            for (size_t i = 0; i < n; i++) b[i] = ld_g2<Fq>(points + i * 4 * Fq::N);
            auto r = n == 0 ? C::G2::infinity() : algo == 0 ? msm_pippenger<typename C::G2, Fr>(b.data(), sc, n, threads) : msm_naive<typename C::G2, Fr>(b.data(), sc, n);
            st_g2<Fq>(out_affine, r.to_affine());
        }
    });
    return 0;
}
// Jacobian (X,Y,Z) -> packed affine; Z == 0 -> infinity. Used to normalise the product's MSM output before comparing.
int orc_jacobian_to_affine(int curve, int group, const uint64_t* jac, uint64_t* out_affine) {
    DISPATCH(curve, {
        typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2;
        if (group == 0) { typename C::G1 p = {ld<Fq>(jac), ld<Fq>(jac + Fq::N), ld<Fq>(jac + 2 * Fq::N)}; st_g1<Fq>(out_affine, p.to_affine()); }
        else {
            typename C::G2 p = {Fq2{ld<Fq>(jac), ld<Fq>(jac + Fq::N)}, Fq2{ld<Fq>(jac + 2 * Fq::N), ld<Fq>(jac + 3 * Fq::N)}, Fq2{ld<Fq>(jac + 4 * Fq::N), ld<Fq>(jac + 5 * Fq::N)}};
            st_g2<Fq>(out_affine, p.to_affine());
        }
    });
    return 0;
}
int orc_on_curve(int curve, int group, const uint64_t* pt) {
    DISPATCH(curve, {
        typedef typename C::Fq Fq;
        return group == 0 ? (int)C::G1::on_curve(ld_g1<Fq>(pt)) : (int)C::G2::on_curve(ld_g2<Fq>(pt));
    });
    return 0;
}
// generator * scalar (packed affine out); used by tests to make valid points
int orc_generator_mul(int curve, int group, const uint64_t* scalar, uint64_t* out_affine) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        Fr s = ld<Fr>(scalar);
        if (group == 0) st_g1<Fq>(out_affine, scalar_mul(C::G1::from_affine(C::g1_generator()), s).to_affine());
        else st_g2<Fq>(out_affine, scalar_mul(C::G2::from_affine(C::g2_generator()), s).to_affine());
    });
    return 0;
}
// out[i] = point[i] * scalar[i]  (packed affine in/out)
int orc_points_mul(int curve, int group, const uint64_t* pts, const uint64_t* scalars, size_t n, uint64_t* out_affine) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        for (size_t i = 0; i < n; i++) {
            Fr s = ld<Fr>(scalars + i * 4);
            if (group == 0) st_g1<Fq>(out_affine + i * 2 * Fq::N, scalar_mul(C::G1::from_affine(ld_g1<Fq>(pts + i * 2 * Fq::N)), s).to_affine());
            else st_g2<Fq>(out_affine + i * 4 * Fq::N, scalar_mul(C::G2::from_affine(ld_g2<Fq>(pts + i * 4 * Fq::N)), s).to_affine());
        }
    });
    return 0;
}
// out = a + b (packed affine)
int orc_point_add(int curve, int group, const uint64_t* a, const uint64_t* b, uint64_t* out_affine) {
    DISPATCH(curve, {
        typedef typename C::Fq Fq;
        if (group == 0) st_g1<Fq>(out_affine, C::G1::from_affine(ld_g1<Fq>(a)).add_affine(ld_g1<Fq>(b)).to_affine());
        else st_g2<Fq>(out_affine, C::G2::from_affine(ld_g2<Fq>(a)).add_affine(ld_g2<Fq>(b)).to_affine());
    });
    return 0;
}

// ---- zkey / wtns ------------------------------------------------------------------------------------
void* orc_zkey_open(int curve, const char* path) {
    try {
        auto* h = new ZKeyHandle{curve};
        if (curve == 0) { Bn254::init(); h->bn = new ZKey<Bn254>(read_zkey<Bn254>(path)); }
        else if (curve == 1) { Bls12_381::init(); h->bls = new ZKey<Bls12_381>(read_zkey<Bls12_381>(path)); }
        else { delete h; g_err = "bad curve id"; return nullptr; }
        return h;
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_zkey_close(void* hh) { auto* h = (ZKeyHandle*)hh; if (!h) return; delete h->bn; delete h->bls; delete h; }

#define ZK(h, ...) { auto* zh = (ZKeyHandle*)(h); try { if (zh->curve == 0) { typedef Bn254 C; auto& z = *zh->bn; __VA_ARGS__; } else { typedef Bls12_381 C; auto& z = *zh->bls; __VA_ARGS__; } } catch (const std::exception& e) { g_err = e.what(); return -2; } }

// info: n_vars, n_public, domain_size, pow, num_constraints, nnzA, nnzB
int orc_zkey_info(void* h, size_t* info) {
    ZK(h, { info[0] = z.n_vars; info[1] = z.n_public; info[2] = z.domain_size; info[3] = z.pow; info[4] = z.num_constraints; info[5] = z.col[0].size(); info[6] = z.col[1].size(); });
    return 0;
}
// which: 0 ic, 1 a_query, 2 b_g1_query, 3 b_g2_query, 4 l_query, 5 h_query, 6 (alpha1,beta1,delta1), 7 (beta2,gamma2,delta2)
int orc_zkey_points(void* h, int which, uint64_t* out) {
    ZK(h, {
        typedef typename C::Fq Fq;
        auto put1 = [&](const std::vector<typename C::G1::Affine>& v) { for (size_t i = 0; i < v.size(); i++) st_g1<Fq>(out + i * 2 * Fq::N, v[i]); };
        switch (which) {
            case 0: put1(z.ic); break; case 1: put1(z.a_query); break; case 2: put1(z.b_g1_query); break;
            case 3: for (size_t i = 0; i < z.b_g2_query.size(); i++) st_g2<Fq>(out + i * 4 * Fq::N, z.b_g2_query[i]); break;
            case 4: put1(z.l_query); break; case 5: put1(z.h_query); break;
            case 6: st_g1<Fq>(out, z.alpha_g1); st_g1<Fq>(out + 2 * Fq::N, z.beta_g1); st_g1<Fq>(out + 4 * Fq::N, z.delta_g1); break;
            case 7: st_g2<Fq>(out, z.beta_g2); st_g2<Fq>(out + 4 * Fq::N, z.gamma_g2); st_g2<Fq>(out + 8 * Fq::N, z.delta_g2); break;
            default: g_err = "bad selector"; return -1;
        }
    });
    return 0;
}
int orc_zkey_matrix(void* h, int m, uint32_t* row_ptr, uint32_t* col, uint64_t* coeff) {
    ZK(h, {
        memcpy(row_ptr, z.row_ptr[m].data(), z.row_ptr[m].size() * 4);
        memcpy(col, z.col[m].data(), z.col[m].size() * 4);
        memcpy(coeff, z.coeff[m].data(), z.coeff[m].size() * sizeof(typename C::Fr));
    });
    return 0;
}
int orc_wtns_read(int curve, const char* path, uint64_t* out, size_t cap, size_t* n) {
    DISPATCH(curve, {
        auto w = read_wtns<typename C::Fr>(path);
        *n = w.size();
        if (out) { if (w.size() > cap) { g_err = "buffer too small"; return -3; } memcpy(out, w.data(), w.size() * sizeof(typename C::Fr)); }
    });
    return 0;
}

// ---- co-plonk round 1 (plonk.hpp) ----------------------------------------------------------------------
// info: n_vars, n_public, domain_size, power, n_additions, n_constraints
int orc_plonk_zkey_info(int curve, const char* path, size_t* info) {
    DISPATCH(curve, {
        auto z = read_plonk_zkey<C>(path);
        info[0] = z.n_vars; info[1] = z.n_public; info[2] = z.domain_size; info[3] = z.power; info[4] = z.n_additions; info[5] = z.n_constraints;
    });
    return 0;
}
// maps: 3 x n_constraints u32 (a, b, c); additions: n_additions x (id1, id2) u32 and x (f1, f2) Fr; p_tau: (domain_size + 6) packed G1
int orc_plonk_zkey_data(int curve, const char* path, uint32_t* maps, uint32_t* add_ids, uint64_t* add_factors, uint64_t* p_tau) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        auto z = read_plonk_zkey<C>(path);
        for (size_t i = 0; i < z.n_constraints; i++) { maps[i] = z.map_a[i]; maps[z.n_constraints + i] = z.map_b[i]; maps[2 * z.n_constraints + i] = z.map_c[i]; }
        for (size_t i = 0; i < z.n_additions; i++) {
            add_ids[2 * i] = z.additions[i].id1; add_ids[2 * i + 1] = z.additions[i].id2;
            st<Fr>(add_factors + 2 * i * Fr::N, z.additions[i].f1); st<Fr>(add_factors + (2 * i + 1) * Fr::N, z.additions[i].f2);
        }
        for (size_t i = 0; i < z.p_tau.size(); i++) st_g1<Fq>(p_tau + i * 2 * Fq::N, z.p_tau[i]);
    });
    return 0;
}
// full_witness: n_vars - n_additions Montgomery elements (Groth16-style, leading one); blind: 6 Fr; out: 3 packed G1 (a, b, c);
// polys (optional): 3 x (domain_size + 2) blinded coefficient vectors
int orc_plonk_round1_plain(int curve, const char* path, const uint64_t* full_witness, const uint64_t* blind, uint64_t* out_commits, uint64_t* polys) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        auto z = read_plonk_zkey<C>(path);
        const Fr* w = reinterpret_cast<const Fr*>(full_witness);
        std::vector<Fr> fw(w, w + (z.n_vars - z.n_additions));
        Fr b[6]; for (int i = 0; i < 6; i++) b[i] = ld<Fr>(blind + i * Fr::N);
        std::vector<Fr> pl;
        auto cm = plonk_round1_plain<C>(z, fw, b, polys ? &pl : nullptr);
        for (int i = 0; i < 3; i++) st_g1<Fq>(out_commits + i * 2 * Fq::N, cm[i]);
        if (polys) memcpy(polys, pl.data(), pl.size() * sizeof(Fr));
    });
    return 0;
}

// transcript KAT hook: items = sequence of (kind, payload) with kind 0 = scalar (Fr), 1 = G1 point (packed affine, (0,0) = infinity)
int orc_plonk_transcript(int curve, const int* kinds, const uint64_t* const* payloads, int n_items, uint64_t* out_challenge) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        PlonkTranscript<C> t;
        for (int i = 0; i < n_items; i++) { if (kinds[i] == 0) t.add_scalar(ld<Fr>(payloads[i])); else t.add_point(ld_g1<Fq>(payloads[i])); }
        st<Fr>(out_challenge, t.get_challenge());
    });
    return 0;
}
// rounds 1 + 2 with the plain driver; blind: 9 Fr (b_1..b_9); out: beta, gamma (Fr), commit_z (packed G1), optional poly_z (domain_size + 3)
int orc_plonk_round2_plain(int curve, const char* path, const uint64_t* full_witness, const uint64_t* blind, uint64_t* out_beta_gamma, uint64_t* out_commit_z, uint64_t* poly_z) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        auto z = read_plonk_zkey<C>(path);
        const Fr* w = reinterpret_cast<const Fr*>(full_witness);
        std::vector<Fr> fw(w, w + (z.n_vars - z.n_additions));
        Fr b[9]; for (int i = 0; i < 9; i++) b[i] = ld<Fr>(blind + i * Fr::N);
        auto cm = plonk_round1_plain<C>(z, fw, b);
        auto r2 = plonk_round2_plain<C>(z, fw, b, cm);
        st<Fr>(out_beta_gamma, r2.beta); st<Fr>(out_beta_gamma + Fr::N, r2.gamma);
        st_g1<Fq>(out_commit_z, r2.commit_z);
        if (poly_z) memcpy(poly_z, r2.poly_z.data(), r2.poly_z.size() * sizeof(Fr));
    });
    return 0;
}

// the plain-driver prover up to round `upto` (1..5); blind = 11 Fr.  commits: 9 packed G1 (a, b, c, z, t1, t2, t3, wxi, wxiw; zeros when not
// reached), challenges: beta, gamma, alpha, xi, v; evals: a, b, c, s1, s2, zw.  t_polys (optional): t1 (n+1) | t2 (n+1) | t3 (n+6)
int orc_plonk_prove_plain(int curve, const char* path, const uint64_t* full_witness, const uint64_t* blind, int upto, uint64_t* commits, uint64_t* challenges, uint64_t* evals, uint64_t* t_polys) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        auto z = read_plonk_zkey<C>(path);
        const Fr* w = reinterpret_cast<const Fr*>(full_witness);
        std::vector<Fr> fw(w, w + (z.n_vars - z.n_additions));
        Fr b[11]; for (int i = 0; i < 11; i++) b[i] = ld<Fr>(blind + i * Fr::N);
        PlonkPlainProver<C> pr(z, fw, b);
        memset(commits, 0, 9 * 2 * Fq::N * 8); memset(challenges, 0, 5 * Fr::N * 8); memset(evals, 0, 6 * Fr::N * 8);
        pr.round1();
        for (int k = 0; k < 3; k++) st_g1<Fq>(commits + k * 2 * Fq::N, pr.commit[k]);
        if (upto >= 2) { pr.round2(); st_g1<Fq>(commits + 3 * 2 * Fq::N, pr.commit_z); st<Fr>(challenges, pr.beta); st<Fr>(challenges + Fr::N, pr.gamma); }
        if (upto >= 3) {
            pr.round3();
            for (int k = 0; k < 3; k++) st_g1<Fq>(commits + (4 + k) * 2 * Fq::N, pr.commit_t[k]);
            st<Fr>(challenges + 2 * Fr::N, pr.alpha);
            if (t_polys) { memcpy(t_polys, pr.t1.data(), pr.t1.size() * sizeof(Fr)); memcpy(t_polys + pr.t1.size() * Fr::N, pr.t2.data(), pr.t2.size() * sizeof(Fr));
                           memcpy(t_polys + (pr.t1.size() + pr.t2.size()) * Fr::N, pr.t3.data(), pr.t3.size() * sizeof(Fr)); }
        }
        if (upto >= 4) {
            pr.round4();
            st<Fr>(challenges + 3 * Fr::N, pr.xi);
            const Fr ev[6] = {pr.eval_a, pr.eval_b, pr.eval_c, pr.eval_s1, pr.eval_s2, pr.eval_zw};
            for (int i = 0; i < 6; i++) st<Fr>(evals + i * Fr::N, ev[i]);
        }
        if (upto >= 5) {
            pr.round5();
            st<Fr>(challenges + 4 * Fr::N, pr.v[0]);
            st_g1<Fq>(commits + 7 * 2 * Fq::N, pr.commit_wxi); st_g1<Fq>(commits + 8 * 2 * Fq::N, pr.commit_wxiw);
        }
    });
    return 0;
}

// Plonk verifier with the verifying key of the zkey: commits = 9 packed G1, evals = 6 Fr, pub = n_pub Fr.  1 = accept, 0 = reject.
// vk_out (optional): 8 packed G1 (Qm, Ql, Qr, Qo, Qc, S1, S2, S3) | packed G2 X_2 | k1, k2 — to cross-check verification_key.json
int orc_plonk_verify(int curve, const char* path, const uint64_t* commits, const uint64_t* evals, const uint64_t* pub, size_t n_pub, uint64_t* vk_out) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr; typedef typename C::Fq Fq;
        auto z = read_plonk_zkey<C>(path);
        if (vk_out) {
            for (int i = 0; i < 8; i++) st_g1<Fq>(vk_out + i * 2 * Fq::N, z.vk_g1[i]);
            st_g2<Fq>(vk_out + 16 * Fq::N, z.x_2);
            st<Fr>(vk_out + 20 * Fq::N, z.k1); st<Fr>(vk_out + 20 * Fq::N + Fr::N, z.k2);
        }
        if (!commits) return 1;
        AffineT<Fq> cm[9]; for (int i = 0; i < 9; i++) cm[i] = ld_g1<Fq>(commits + i * 2 * Fq::N);
        Fr ev[6]; for (int i = 0; i < 6; i++) ev[i] = ld<Fr>(evals + i * Fr::N);
        std::vector<Fr> pv(n_pub); for (size_t i = 0; i < n_pub; i++) pv[i] = ld<Fr>(pub + i * Fr::N);
        return plonk_verify<C>(z, cm, ev, pv) ? 1 : 0;
    });
    return 0;
}

// ---- prover ------------------------------------------------------------------------------------------
int orc_witness_map_plain(void* h, const uint64_t* full_witness, uint64_t* out_h) {
    ZK(h, {
        typedef typename C::Fr Fr;
        const Fr* w = reinterpret_cast<const Fr*>(full_witness);
        std::vector<Fr> pub(w, w + z.n_public + 1), wit(w + z.n_public + 1, w + z.n_vars);
        auto hv = witness_map_plain<C>(z, pub, wit);
        memcpy(out_h, hv.data(), hv.size() * sizeof(Fr));
    });
    return 0;
}
// proof layout: A (G1 packed) || B (G2 packed) || C (G1 packed) ; seconds (optional) = wall time of the prove call
int orc_prove_plain(void* h, const uint64_t* full_witness, const uint64_t* r, const uint64_t* s, int threads, uint64_t* out_proof, double* seconds) {
    ZK(h, {
        typedef typename C::Fr Fr;
        const Fr* w = reinterpret_cast<const Fr*>(full_witness);
        std::vector<Fr> fw(w, w + z.n_vars);
        auto t0 = std::chrono::steady_clock::now();
        auto pf = prove_plain<C>(z, fw, ld<Fr>(r), ld<Fr>(s), threads);
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        st_proof<C>(out_proof, pf);
    });
    return 0;
}
// shares: wit_a[i], wit_b[i] (i = party) each n_vars-n_public-1 elements; streams: s[i] each stream_len elements
// (needs 2*m + 4).  out_proofs = 3 proofs back to back.  out_h (optional) = party-0 h share: a then b (2*m elements).
int orc_prove_rep3(void* h, const uint64_t* pub, const uint64_t* const* wit_a, const uint64_t* const* wit_b,
                   const uint64_t* const* streams, size_t stream_len, int threads, uint64_t* out_proofs, uint64_t* out_h) {
    ZK(h, {
        typedef typename C::Fr Fr;
        const size_t n_aux = z.n_vars - z.n_public - 1;
        Rep3Sim<C> sim(z);
        sim.threads = threads;
        const Fr* p = reinterpret_cast<const Fr*>(pub);
        sim.pub.assign(p, p + z.n_public + 1);
        std::vector<Fr> st_[3];
        for (int i = 0; i < 3; i++) {
            const Fr* a = reinterpret_cast<const Fr*>(wit_a[i]); const Fr* b = reinterpret_cast<const Fr*>(wit_b[i]);
            sim.wit[i].a.assign(a, a + n_aux); sim.wit[i].b.assign(b, b + n_aux);
            const Fr* s = reinterpret_cast<const Fr*>(streams[i]);
            st_[i].assign(s, s + stream_len);
            sim.stream[i] = &st_[i];
        }
        size_t m = 1; while (m < z.num_constraints + z.n_public + 1) m <<= 1;
        if (stream_len < 2 * m + 4) { g_err = "randomness stream too short"; return -3; }
        Proof<C> out[3];
        ShareVec<Fr> hs[3];
        sim.prove(out, &hs);
        const int psz = 8 * C::Fq::N;
        for (int i = 0; i < 3; i++) st_proof<C>(out_proofs + i * psz, out[i]);
        if (out_h) { memcpy(out_h, hs[0].a.data(), m * sizeof(Fr)); memcpy(out_h + m * 4, hs[0].b.data(), m * sizeof(Fr)); }
    });
    return 0;
}
// Shamir (n parties, threshold t): wit[i] = party i's shares of the private witness, streams[i] = party i's private randomness.
// out_proofs = n proofs; out_h (optional) = party 0's h shares
int orc_prove_shamir(void* h, int n, int t, const uint64_t* pub, const uint64_t* const* wit, const uint64_t* const* streams, size_t stream_len,
                     size_t preprocess, int threads, uint64_t* out_proofs, uint64_t* out_h) {
    ZK(h, {
        typedef typename C::Fr Fr;
        const size_t n_aux = z.n_vars - z.n_public - 1;
        ShamirSim<C> sim(z, n, t);
        sim.threads = threads;
        const Fr* p = reinterpret_cast<const Fr*>(pub);
        sim.pub.assign(p, p + z.n_public + 1);
        std::vector<std::vector<Fr>> st_(n);
        for (int i = 0; i < n; i++) {
            const Fr* a = reinterpret_cast<const Fr*>(wit[i]);
            sim.wit[i].assign(a, a + n_aux);
            const Fr* s = reinterpret_cast<const Fr*>(streams[i]);
            st_[i].assign(s, s + stream_len);
            sim.stream[i] = &st_[i];
        }
        std::vector<std::vector<Fr>> hs;
        sim.preprocess(preprocess);
        auto out = sim.prove(&hs);
        const int psz = 8 * C::Fq::N;
        for (int i = 0; i < n; i++) st_proof<C>(out_proofs + i * psz, out[i]);
        if (out_h) memcpy(out_h, hs[0].data(), hs[0].size() * sizeof(Fr));
    });
    return 0;
}
// returns 1 = accept, 0 = reject, <0 error. ic has n_pub+1 packed G1 points.
int orc_verify(int curve, const uint64_t* alpha1, const uint64_t* beta2, const uint64_t* gamma2, const uint64_t* delta2,
               const uint64_t* ic, size_t n_pub, const uint64_t* pub, const uint64_t* proof) {
    DISPATCH(curve, {
        typedef typename C::Fq Fq; typedef typename C::Fr Fr;
        std::vector<typename C::G1::Affine> icv(n_pub + 1);
        for (size_t i = 0; i <= n_pub; i++) icv[i] = ld_g1<Fq>(ic + i * 2 * Fq::N);
        std::vector<Fr> pv(n_pub);
        for (size_t i = 0; i < n_pub; i++) pv[i] = ld<Fr>(pub + i * 4);
        return groth16_verify<C>(ld_g1<Fq>(alpha1), ld_g2<Fq>(beta2), ld_g2<Fq>(gamma2), ld_g2<Fq>(delta2), icv, pv, ld_proof<C>(proof)) ? 1 : 0;
    });
    return 0;
}
// bilinearity self-check of the pairing: t(aP, bQ) == t(P,Q)^(ab) tested as t(aP,Q) == t(P,aQ); returns 1 if it holds and t != 1
int orc_pairing_selfcheck(int curve, const uint64_t* scalar) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr;
        Fr a = ld<Fr>(scalar);
        auto P = C::g1_generator(); auto Q = C::g2_generator();
        auto aP = scalar_mul(C::G1::from_affine(P), a).to_affine();
        auto aQ = scalar_mul(C::G2::from_affine(Q), a).to_affine();
        auto e1 = tate_pairing<C>(aP, Q); auto e2 = tate_pairing<C>(P, aQ); auto e0 = tate_pairing<C>(P, Q);
        return (e1 == e2 && !(e0 == Fp12T<C>::one())) ? 1 : 0;
    });
    return 0;
}

// e(P, Q) as snarkjs / arkworks define it (pairing.hpp::optimal_ate_pairing), in the JSON layout of `vk_alphabeta_12`
// (circom-types/src/groth16/verification_key.rs:46-49): out[i][j][k] = coefficient c_i.c_j.c_k of Fq12 = Fq6[w]/(w^2 - v),
// Fq6 = Fq2[v]/(v^3 - xi), 12 base-field elements in Montgomery form
int orc_pairing(int curve, const uint64_t* g1_affine, const uint64_t* g2_affine, uint64_t* out) {
    DISPATCH(curve, {
        typedef typename C::Fq Fq;
        auto e = optimal_ate_pairing<C>(ld_g1<Fq>(g1_affine), ld_g2<Fq>(g2_affine));
        for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) {           // coefficient of w^(2j + i)
            st<Fq>(out + ((i * 3 + j) * 2 + 0) * Fq::N, e.c[2 * j + i].c0);
            st<Fq>(out + ((i * 3 + j) * 2 + 1) * Fq::N, e.c[2 * j + i].c1);
        }
        return 0;
    });
    return 0;
}

// bench.py cpu_baseline leg: seconds for one REP3 party's prove compute at m = 2^log_m (stage = 4 doubles, optional)
double orc_bench_rep3_party(int curve, int log_m, int threads, uint64_t seed, double* stage) {
    try {
        if (curve == 0) return bench_rep3_party<Bn254>(log_m, threads, seed, stage);
        if (curve == 1) return bench_rep3_party<Bls12_381>(log_m, threads, seed, stage);
        g_err = "bad curve id"; return -1.0;
    } catch (const std::exception& e) { g_err = e.what(); return -2.0; }
}

// the cache-blocked multi-threaded transforms of the CPU baseline against the plain ones of poly.hpp (1 = identical results)
int orc_bench_ntt_selfcheck(int curve, int log_n, int threads) {
    DISPATCH(curve, {
        typedef typename C::Fr Fr;
        const size_t n = (size_t)1 << log_n;
        auto dom = groth16_domain<Fr>((size_t)log_n, n - 2, 2);
        XorShift rng{77};
        std::vector<Fr> a(n); for (auto& x : a) x = rand_fp<Fr>(rng);
        std::vector<Fr> f1 = a, f2 = a, i1 = a, i2 = a;
        std::vector<Fr> tw(n / 2 ? n / 2 : 1), twi(n / 2 ? n / 2 : 1);
        Fr wi = dom.omega.inverse(); tw[0] = twi[0] = Fr::one(); for (size_t i = 1; i < n / 2; i++) { tw[i] = tw[i - 1] * dom.omega; twi[i] = twi[i - 1] * wi; }
        Pool pool(threads);
        ntt_forward(f1.data(), n, dom.omega); ntt_forward_mt(f2.data(), n, tw, pool);
        ntt_inverse(i1.data(), n, dom.omega); ntt_inverse_mt(i2.data(), n, twi, Fr::from_u64((uint64_t)n).inverse(), pool);
        for (size_t i = 0; i < n; i++) if (!(f1[i] == f2[i]) || !(i1[i] == i2[i])) return 0;
        return 1;
    });
    return 1;
}

// the same with a second thread setting on the same inputs (out: stage_b[4], total_b, msm_shared; see bench.hpp)
double orc_bench_rep3_party2(int curve, int log_m, int threads, int threads_b, uint64_t seed, double* stage, double* stage_b, double* total_b, int* msm_shared) {
    try {
        if (curve == 0) return bench_rep3_party<Bn254>(log_m, threads, seed, stage, threads_b, stage_b, total_b, msm_shared);
        if (curve == 1) return bench_rep3_party<Bls12_381>(log_m, threads, seed, stage, threads_b, stage_b, total_b, msm_shared);
        g_err = "bad curve id"; return -1.0;
    } catch (const std::exception& e) { g_err = e.what(); return -2.0; }
}

// the party's host mask draws of one proof (4 x m draws on one thread, rngs.rs:37-46), seconds
double orc_bench_mask_draws(int curve, size_t m, uint64_t seed) {
    try {
        if (curve == 0) return bench_mask_draws<Bn254>(m, seed);
        if (curve == 1) return bench_mask_draws<Bls12_381>(m, seed);
        g_err = "bad curve id"; return -1.0;
    } catch (const std::exception& e) { g_err = e.what(); return -2.0; }
}
// one REP3 party on a zkey + wtns pair, mask draws included (stage = 5 doubles, optional), seconds per proof
double orc_bench_rep3_party_file(int curve, const char* zkey_path, const char* wtns_path, int threads, int reps, double* stage) {
    try {
        if (curve == 0) return bench_rep3_party_file<Bn254>(zkey_path, wtns_path, threads, reps, stage);
        if (curve == 1) return bench_rep3_party_file<Bls12_381>(zkey_path, wtns_path, threads, reps, stage);
        g_err = "bad curve id"; return -1.0;
    } catch (const std::exception& e) { g_err = e.what(); return -2.0; }
}

// synthetic satisfiable circuit + valid CRS of domain size 2^log_m written as .zkey / .wtns (test tooling)
int orc_make_synthetic(int curve, int log_m, uint64_t seed, const char* zkey_path, const char* wtns_path, int threads) {
    DISPATCH(curve, { make_synthetic<C>(log_m, seed, zkey_path, wtns_path, threads); });
    return 0;
}
int orc_make_synthetic_pub(int curve, int log_m, uint64_t seed, const char* zkey_path, const char* wtns_path, int threads, uint64_t n_public) {
    DISPATCH(curve, { make_synthetic<C>(log_m, seed, zkey_path, wtns_path, threads, (size_t)n_public); });
    return 0;
}

// ---- randomness streams (rngs.hpp): rand_chacha ChaCha12Rng + ark-ff Fp::rand, restated ------------------------------------------------------
int orc_chacha_block(int rounds, const uint32_t* key8, uint64_t counter, uint64_t stream, uint32_t* out16) { chacha_block(rounds, key8, counter, stream, out16); return 0; }
// n x Fr::rand from ChaCha12Rng::from_seed(seed) positioned at word_pos; *word_pos_after = get_word_pos() after the last draw
int orc_chacha12_fr_rand(int curve, const uint8_t* seed32, uint64_t word_pos, size_t n, uint64_t* out, uint64_t* word_pos_after) {
    DISPATCH(curve, {
        ChaCha12Stream rng(seed32, word_pos);
        for (size_t i = 0; i < n; i++) fr_rand(rng, C::Fr::K.p, C::Fr::K.bits, out + 4 * i);
        if (word_pos_after) *word_pos_after = rng.word_pos;
    });
    return 0;
}
// Rep3Rand::masking_field_element x n (rngs.rs:37-46): rand(rng1) - rand(rng2), each stream advanced by its own rejections
int orc_rep3_masks_chacha12(int curve, const uint8_t* seed1, uint64_t* pos1, const uint8_t* seed2, uint64_t* pos2, size_t n, uint64_t* out) {
    DISPATCH(curve, {
        ChaCha12Stream r1(seed1, *pos1), r2(seed2, *pos2);
        for (size_t i = 0; i < n; i++) {
            uint64_t a[4], b[4];
            fr_rand(r1, C::Fr::K.p, C::Fr::K.bits, a); fr_rand(r2, C::Fr::K.p, C::Fr::K.bits, b);
            st(out + 4 * i, ld<typename C::Fr>(a) - ld<typename C::Fr>(b));
        }
        *pos1 = r1.word_pos; *pos2 = r2.word_pos;
    });
    return 0;
}

}  // extern "C"
