// Short-Weierstrass point arithmetic, y^2 = x^3 + b (a = 0), generic over the coordinate field (Fq for G1, Fq2 for G2).
// Replaces the ark-ec 0.4.2 group operations the reference reaches through `C::msm_unchecked`
// (`/root/reference/mpc-core/src/protocols/rep3.rs:942-943`) and the O(1) point algebra of
// `/root/reference/co-circom/co-groth16/src/groth16.rs:227-231,308-312`.
//
// Accumulators use extended Jacobian "XYZZ" coordinates (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2:
// mixed addition is 8M + 2S with no inversion, which is what the bucket phase of Pippenger spends its time in.
// ZZ == 0 encodes infinity.  All exceptional cases (equal points, opposite points, infinity) are handled, because
// zkey queries do contain repeated points and points at infinity (e.g. b_g1_query of multiplier2).
#pragma once
#include "field.hpp"

namespace cg {

template <class F>
struct alignas(16) Affine {
    F x, y;   // (0, 0) = infinity (the packed zkey encoding, circom-types/src/traits.rs:113-115)
    CG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    CG_HD static Affine infinity() { return {F::zero(), F::zero()}; }
};

template <class F>
struct alignas(16) XYZZ {
    F x, y, zz, zzz;
    CG_HD static XYZZ infinity() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    CG_HD bool is_inf() const { return zz.is_zero(); }
    CG_HD static XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return infinity();
        return {p.x, p.y, F::one(), F::one()};
    }
    CG_HD XYZZ neg() const { return {x, y.neg(), zz, zzz}; }
};

// 2*(x1, y1) for an affine non-infinity point (mdbl-2008-s-1, a = 0)
template <class F>
CG_HD XYZZ<F> xyzz_dbl_affine(const F& x1, const F& y1) {
    if (y1.is_zero()) return XYZZ<F>::infinity();
    F U = y1.dbl(), V = U.sqr(), W = U * V, S = x1 * V;
    F xx = x1.sqr();
    F M = xx.dbl() + xx;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * y1;
    return {X3, Y3, V, W};
}

// dbl-2008-s-1, a = 0   (out of line: only the bucket-reduction / fold phases use it, and inlining it everywhere
// multiplies code size and build time without buying anything)
template <class F>
CG_HD_NOINLINE XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
    if (p.is_inf() || p.y.is_zero()) return XYZZ<F>::infinity();
    F U = p.y.dbl(), V = U.sqr(), W = U * V, S = p.x * V;
    F xx = p.x.sqr();
    F M = xx.dbl() + xx;
    F X3 = M.sqr() - S.dbl();
    F Y3 = M * (S - X3) - W * p.y;
    return {X3, Y3, V * p.zz, W * p.zzz};
}

// acc + (x2, y2)  (madd-2008-s), (x2, y2) affine and not infinity
template <class F>
CG_HD XYZZ<F> xyzz_madd(const XYZZ<F>& acc, const F& x2, const F& y2) {
    if (acc.is_inf()) return {x2, y2, F::one(), F::one()};
    F U2 = x2 * acc.zz, S2 = y2 * acc.zzz;
    F P = U2 - acc.x, R = S2 - acc.y;
    if (P.is_zero()) {
        if (R.is_zero()) return xyzz_dbl_affine(x2, y2);
        return XYZZ<F>::infinity();
    }
    F PP = P.sqr(), PPP = P * PP, Q = acc.x * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    F Y3 = R * (Q - X3) - acc.y * PPP;
    return {X3, Y3, acc.zz * PP, acc.zzz * PPP};
}

// add-2008-s   (out of line, see xyzz_dbl)
template <class F>
CG_HD_NOINLINE XYZZ<F> xyzz_add(const XYZZ<F>& a, const XYZZ<F>& b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    F U1 = a.x * b.zz, U2 = b.x * a.zz, S1 = a.y * b.zzz, S2 = b.y * a.zzz;
    F P = U2 - U1, R = S2 - S1;
    if (P.is_zero()) {
        if (R.is_zero()) return xyzz_dbl(a);
        return XYZZ<F>::infinity();
    }
    F PP = P.sqr(), PPP = P * PP, Q = U1 * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    F Y3 = R * (Q - X3) - S1 * PPP;
    return {X3, Y3, a.zz * b.zz * PP, a.zzz * b.zzz * PPP};
}

// XYZZ -> Jacobian (X', Y', Z') with x = X'/Z'^2, y = Y'/Z'^3: take Z' = ZZZ (z^3): X' = X*ZZ^2, Y' = Y*ZZZ^2.
// Z' == 0 <=> infinity, matching ark-ec `short_weierstrass::Projective` (result type of msm_public_points, traits.rs:561-568).
template <class F>
struct alignas(16) Jacobian { F x, y, z; };

template <class F>
CG_HD Jacobian<F> xyzz_to_jacobian(const XYZZ<F>& p) {
    if (p.is_inf()) return {F::one(), F::one(), F::zero()};
    return {p.x * p.zz.sqr(), p.y * p.zzz.sqr(), p.zzz};
}
template <class F>
CG_HD XYZZ<F> jacobian_to_xyzz(const Jacobian<F>& p) {
    if (p.z.is_zero()) return XYZZ<F>::infinity();
    F zz = p.z.sqr();
    return {p.x, p.y, zz, zz * p.z};
}
template <class F>
CG_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
    if (p.is_inf()) return Affine<F>::infinity();
    F izzz = fp_inverse(p.zzz);            // 1/z^3
    F iz = izzz * p.zz;                    // z^2/z^3 = 1/z
    F izz = iz.sqr();
    return {p.x * izz, p.y * izzz};
}

// k * P for a canonical little-endian scalar (O(1) host-side work: scalar_mul_public_point, rep3.rs:820-825)
template <class F>
CG_HD XYZZ<F> xyzz_scalar_mul(const XYZZ<F>& p, const uint32_t* k, int nlimbs) {
    XYZZ<F> r = XYZZ<F>::infinity();
    for (int i = nlimbs * 32 - 1; i >= 0; i--) {
        r = xyzz_dbl(r);
        if ((k[i / 32] >> (i % 32)) & 1u) r = xyzz_add(r, p);
    }
    return r;
}

}  // namespace cg
