// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// Groth16 verifier. The reference wraps ark-groth16 =0.4.0 (`/root/reference/co-circom/co-groth16/src/verifier.rs:23-43`,
// `co-groth16/Cargo.toml:23`, crate not vendored): accept iff e(A,B) = e(alpha,beta) * e(vk_x,gamma) * e(C,delta),
// vk_x = IC[0] + sum pub_i IC[i+1].  Any non-degenerate bilinear pairing on G1 x G2 decides that equation identically;
// this file uses the reduced Tate pairing t(P,Q) = f_{r,P}(psi(Q))^((p^12-1)/r) with psi the untwisting map into
// E(Fp12), which needs no curve-specific Frobenius constants (both BN254, D-type twist, and BLS12-381, M-type twist).
// It is used to pin the oracle's proofs and the snarkjs `circom.proof` KATs
// (`co-groth16/src/lib.rs:56-73,104-140`, `tests/tests/circom/e2e_tests/mod.rs:85-100`).
// The numeric value `vk_alphabeta_12` of verification_key.json (`circom-types/src/groth16/verification_key.rs:46-49`: e(alpha, beta) as
// twelve decimal strings, written by snarkjs, parsed by the reference into `P::TargetField`) is an OPTIMAL-ATE value: it is pinned by
// optimal_ate_pairing below (tests/test_oracle_pinning.py, all four fixtures).
#pragma once
#include "curves.hpp"
#include "groth16.hpp"
#include "pairing_consts.hpp"

namespace orc {

// Fp12 = Fp2[w]/(w^6 - xi)
template <class C>
struct Fp12T {
    typedef typename C::Fq2 Fq2;
    Fq2 c[6];
    static Fp12T one() { Fp12T r; for (auto& x : r.c) x = Fq2::zero(); r.c[0] = Fq2::one(); return r; }
    bool operator==(const Fp12T& o) const { for (int i = 0; i < 6; i++) if (c[i] != o.c[i]) return false; return true; }
    Fp12T operator*(const Fp12T& o) const {
        Fq2 t[11];
        for (auto& x : t) x = Fq2::zero();
        for (int i = 0; i < 6; i++) {
            if (c[i].is_zero()) continue;
            for (int j = 0; j < 6; j++) { if (o.c[j].is_zero()) continue; t[i + j] = t[i + j] + c[i] * o.c[j]; }
        }
        Fq2 xi = C::xi();
        Fp12T r;
        for (int i = 0; i < 6; i++) r.c[i] = t[i];
        for (int i = 6; i < 11; i++) r.c[i - 6] = r.c[i - 6] + t[i] * xi;
        return r;
    }
    Fp12T pow_hex(const char* hex) const {
        std::string s(hex);
        Fp12T r = one();
        for (char ch : s) {
            int d = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
            for (int b = 3; b >= 0; b--) { r = r * r; if ((d >> b) & 1) r = r * (*this); }
        }
        return r;
    }
};

// Miller function f_{r,P} evaluated at the untwisted image of Q (vertical lines dropped: they lie in Fp6)
template <class C>
static Fp12T<C> miller_tate(const typename C::G1::Affine& P, const typename C::G2::Affine& Q) {
    typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2; typedef typename C::Fr Fr;
    Fp12T<C> f = Fp12T<C>::one();
    if (P.inf || Q.inf) return f;
    // psi(Q) = (xq * w^ex, yq * w^3) with D-type: xq = x', ex = 2 ; M-type: xq = x'/xi, ex = 4, yq = y'/xi
    Fq2 xq, yq; int ex;
    if (C::TWIST_D) { xq = Q.x; yq = Q.y; ex = 2; }
    else { Fq2 xi_inv = C::xi().inverse(); xq = Q.x * xi_inv; yq = Q.y * xi_inv; ex = 4; }
    auto line = [&](const Fq& lambda, const Fq& xt, const Fq& yt) {
        // l(Q) = (yQ - yT) - lambda (xQ - xT)
        Fp12T<C> l; for (auto& x : l.c) x = Fq2::zero();
        l.c[0] = {lambda * xt - yt, Fq::zero()};
        l.c[ex] = (-xq).mul_base(lambda);
        l.c[3] = l.c[3] + yq;
        return l;
    };
    Fq tx = P.x, ty = P.y; bool tinf = false;
    const uint64_t* r = Fr::K.p;
    int top = Fr::K.bits - 1;
    for (int i = top - 1; i >= 0; i--) {
        f = f * f;
        if (!tinf) {
            if (ty.is_zero()) { tinf = true; }
            else {
                Fq lambda = (tx.sqr() * Fq::from_u64(3)) * ty.dbl().inverse();
                f = f * line(lambda, tx, ty);
                Fq nx = lambda.sqr() - tx.dbl();
                Fq ny = lambda * (tx - nx) - ty;
                tx = nx; ty = ny;
            }
        }
        if ((r[i / 64] >> (i % 64)) & 1) {
            if (tinf) { tx = P.x; ty = P.y; tinf = false; }
            else if (tx == P.x) {
                if (ty == P.y) {   // doubling case (does not occur for prime-order P, kept for completeness)
                    Fq lambda = (tx.sqr() * Fq::from_u64(3)) * ty.dbl().inverse();
                    f = f * line(lambda, tx, ty);
                    Fq nx = lambda.sqr() - tx.dbl(); Fq ny = lambda * (tx - nx) - ty; tx = nx; ty = ny;
                } else tinf = true;   // T = -P: vertical line, dropped
            } else {
                Fq lambda = (ty - P.y) * (tx - P.x).inverse();
                f = f * line(lambda, tx, ty);
                Fq nx = lambda.sqr() - tx - P.x;
                Fq ny = lambda * (tx - nx) - ty;
                tx = nx; ty = ny;
            }
        }
    }
    return f;
}

template <class C>
static Fp12T<C> final_exp(const Fp12T<C>& f) {
    return f.pow_hex(C::ID == 0 ? FINAL_EXP_BN254 : FINAL_EXP_BLS12_381);
}

template <class C>
static Fp12T<C> tate_pairing(const typename C::G1::Affine& P, const typename C::G2::Affine& Q) {
    return final_exp<C>(miller_tate<C>(P, Q));
}

// ---- optimal ate pairing with the value conventions of snarkjs (ffjavascript / wasmcurves) and ark-ec 0.4.2 (`Bn::pairing`,
// `Bls12::pairing`; crates not vendored, restated from the published algorithms):
//   BN254:      f = f_{6x+2,Q}(P) * l_{[6x+2]Q, pi(Q)}(P) * l_{[6x+2]Q + pi(Q), -pi^2(Q)}(P)              (Vercauteren, optimal ate)
//               e = f^( 2x(6x^2+3x+1) * (p^12-1)/r )   — the hard part of Fuentes-Castaneda, Knapp, Rodriguez-Henriquez computes this
//               multiple of the plain exponent, and both libraries use that chain;
//   BLS12-381:  f = f_{|x|,Q}(P), x < 0  =>  conjugate;   e = conj(f)^( 3 * (p^12-1)/r )                  (Hayashida-Hayasaka-Teruya chain)
// T runs on the twist E'(Fp2); a line through T with slope lambda' (on E'), untwisted and evaluated at P = (xP, yP), is, up to a factor
// in Fp2 that the final exponentiation removes,
//   D-type (BN254, psi(x', y') = (x' w^2, y' w^3)):      yP - lambda' xP w + (lambda' x_T - y_T) w^3
//   M-type (BLS12-381, psi(x', y') = (x'/w^2, y'/w^3)):   xi yP + (lambda' x_T - y_T) w^3 - lambda' xP w^5
// pi on E' is (x', y') -> (conj(x') xi^((p-1)/3), conj(y') xi^((p-1)/2))  (untwist, Frobenius of Fp12, twist back).
template <class F2>
static F2 fp2_pow_hex(const F2& a, const char* hex) {
    F2 r = F2::one();
    for (const char* c = hex; *c; c++) {
        int d = *c <= '9' ? *c - '0' : (*c | 32) - 'a' + 10;
        for (int b = 3; b >= 0; b--) { r = r * r; if ((d >> b) & 1) r = r * a; }
    }
    return r;
}
template <class C>
static Fp12T<C> miller_ate(const typename C::G1::Affine& P, const typename C::G2::Affine& Q) {
    typedef typename C::Fq Fq; typedef typename C::Fq2 Fq2;
    Fp12T<C> f = Fp12T<C>::one();
    if (P.inf || Q.inf) return f;
    struct Pt { Fq2 x, y; };
    auto line = [&](const Fq2& lambda, const Pt& T) {
        Fp12T<C> l; for (auto& x : l.c) x = Fq2::zero();
        l.c[3] = lambda * T.x - T.y;
        if (C::TWIST_D) { l.c[0] = {P.y, Fq::zero()}; l.c[1] = (-lambda).mul_base(P.x); }
        else { l.c[0] = C::xi().mul_base(P.y); l.c[5] = (-lambda).mul_base(P.x); }
        return l;
    };
    auto dbl = [&](Pt& T) {                              // f *= l_{T,T}(P); T = 2T   (T never has order 2: Q has prime order r)
        Fq2 lambda = (T.x.sqr().dbl() + T.x.sqr()) * T.y.dbl().inverse();
        f = f * line(lambda, T);
        Fq2 nx = lambda.sqr() - T.x.dbl();
        T = Pt{nx, lambda * (T.x - nx) - T.y};
    };
    auto add = [&](Pt& T, const Pt& R) {                 // f *= l_{T,R}(P); T = T + R   (T != +-R inside the loop for a point of order r)
        Fq2 lambda = (T.y - R.y) * (T.x - R.x).inverse();
        f = f * line(lambda, T);
        Fq2 nx = lambda.sqr() - T.x - R.x;
        T = Pt{nx, lambda * (T.x - nx) - T.y};
    };
    const Pt Q0{Q.x, Q.y};
    Pt T = Q0;
    bool first = true;
    for (const char* c = C::ID == 0 ? ATE_LOOP_BN254 : ATE_LOOP_BLS12_381; *c; c++) {
        int d = *c <= '9' ? *c - '0' : (*c | 32) - 'a' + 10;
        for (int b = 3; b >= 0; b--) {
            if (first) { if ((d >> b) & 1) first = false; continue; }       // the leading one
            f = f * f;
            dbl(T);
            if ((d >> b) & 1) add(T, Q0);
        }
    }
    if (C::ID == 0) {
        const Fq2 g2 = fp2_pow_hex(C::xi(), BN254_P_MINUS_1_OVER_3), g3 = fp2_pow_hex(C::xi(), BN254_P_MINUS_1_OVER_2);
        auto frob = [&](const Pt& R) { return Pt{R.x.conj() * g2, R.y.conj() * g3}; };
        const Pt Q1 = frob(Q0);
        Pt Q2 = frob(Q1); Q2.y = -Q2.y;
        add(T, Q1); add(T, Q2);
    }
    return f;
}
template <class C>
static Fp12T<C> optimal_ate_pairing(const typename C::G1::Affine& P, const typename C::G2::Affine& Q) {
    Fp12T<C> f = miller_ate<C>(P, Q);
    if (C::ID != 0) for (int k = 1; k < 6; k += 2) f.c[k] = -f.c[k];       // x < 0: f^(p^6) = conjugation over Fp6 (odd powers of w change sign)
    return final_exp<C>(f).pow_hex(C::ID == 0 ? FINAL_EXP_FACTOR_BN254 : FINAL_EXP_FACTOR_BLS12_381);
}

// verifier.rs:23-43 / ark-groth16 verify_proof
template <class C>
static bool groth16_verify(const typename C::G1::Affine& alpha1, const typename C::G2::Affine& beta2,
                           const typename C::G2::Affine& gamma2, const typename C::G2::Affine& delta2,
                           const std::vector<typename C::G1::Affine>& ic, const std::vector<typename C::Fr>& pub,
                           const Proof<C>& pf) {
    typedef typename C::G1 G1;
    if (ic.size() != pub.size() + 1) return false;
    if (!G1::on_curve(pf.a) || !G1::on_curve(pf.c) || !C::G2::on_curve(pf.b)) return false;
    G1 vkx = G1::from_affine(ic[0]);
    for (size_t i = 0; i < pub.size(); i++) vkx = vkx.add(scalar_mul(G1::from_affine(ic[i + 1]), pub[i]));
    auto neg = [](typename G1::Affine a) { if (!a.inf) a.y = -a.y; return a; };
    Fp12T<C> f = miller_tate<C>(pf.a, pf.b);
    f = f * miller_tate<C>(neg(alpha1), beta2);
    f = f * miller_tate<C>(neg(vkx.to_affine()), gamma2);
    f = f * miller_tate<C>(neg(pf.c), delta2);
    return final_exp<C>(f) == Fp12T<C>::one();
}

}  // namespace orc
