// Subgroup membership by endomorphism instead of [r]P (the parser's `is_in_correct_subgroup_assuming_on_curve`, circom-types/src/traits.rs:
// 116-123,147-153; ark-ec's own implementations for these curves use the same tests).  For a point P ON THE CURVE:
//   BN254 G2      [x+1]P + psi([x]P) + psi^2([x]P) = psi^3([2x]P)         (x = 4965661367192848881; one 63-bit multiplication)
//   BLS12-381 G2  psi(P) = [x]P                                            (x = -0xd201000000010000)
//   BLS12-381 G1  sigma(P) = [-x^2]P,  sigma(x, y) = (beta x, y)
// psi = twist^-1 o Frobenius o twist: (x, y) -> (conj(x) gx, conj(y) gy).  Necessity: psi acts on G2 as [p], sigma on G1 as [lambda] with
// lambda^2 + lambda + 1 = 0 (mod r).  Sufficiency: psi satisfies X^2 - tX + p = 0 on all of E'(Fp2), so a point satisfying f(psi)P = 0 is
// killed by Res(f, X^2 - tX + p); that integer is a multiple of r and coprime to the cofactor for each of the three tests, so the order
// of P divides r (tests/test_host_mirror.py recomputes the three resultants and gcds with plain integers).  BN254 G1 has cofactor 1.
// The constants (gx, gy, beta) are computed on the host at first use (capi.hip: make_fast_subgroup) and chosen so that the group's
// generator passes; the [r]P kernel stays behind CG_SUBGROUP_FULL=1 and the tests run both on the same inputs.
#pragma once
#include "curve.hpp"

namespace cg {

template <class B> CG_HD Fp2<B> fp2_conj(const Fp2<B>& a) { return {a.c0, a.c1.neg()}; }

template <class F> CG_HD bool xyzz_same_point(const XYZZ<F>& a, const XYZZ<F>& b) {
    if (a.is_inf() || b.is_inf()) return a.is_inf() && b.is_inf();
    return a.x * b.zz == b.x * a.zz && a.y * b.zzz == b.y * a.zzz;
}
// [k]P, P affine and finite
template <class F> CG_HD XYZZ<F> xyzz_mul_u64(const F& x, const F& y, unsigned long long k) {
    XYZZ<F> r = XYZZ<F>::infinity();
    for (int i = 63; i >= 0; i--) { r = xyzz_dbl(r); if ((k >> i) & 1ull) r = xyzz_madd(r, x, y); }
    return r;
}
template <class F> CG_HD XYZZ<F> xyzz_mul_u64(const XYZZ<F>& p, unsigned long long k) {
    XYZZ<F> r = XYZZ<F>::infinity();
    for (int i = 63; i >= 0; i--) { r = xyzz_dbl(r); if ((k >> i) & 1ull) r = xyzz_add(r, p); }
    return r;
}

template <class F> struct FastSubgroup {
    static constexpr bool available = false;
    CG_HD bool contains(const Affine<F>&) const { return true; }
};

template <class B> struct PsiMap {
    Fp2<B> gx, gy;
    CG_HD XYZZ<Fp2<B>> operator()(const XYZZ<Fp2<B>>& p) const {
        if (p.is_inf()) return p;
        return {fp2_conj(p.x) * gx, fp2_conj(p.y) * gy, fp2_conj(p.zz), fp2_conj(p.zzz)};
    }
};

template <> struct FastSubgroup<Fp2<Bn254Fq>> {
    typedef Fp2<Bn254Fq> F;
    static constexpr bool available = true;
    static constexpr unsigned long long X = 4965661367192848881ull;
    PsiMap<Bn254Fq> psi;
    CG_HD bool contains(const Affine<F>& p) const {
        if (p.is_inf()) return true;
        XYZZ<F> e = xyzz_mul_u64(p.x, p.y, X);                 // [x]P
        XYZZ<F> lhs = xyzz_madd(e, p.x, p.y);                  // [x + 1]P
        e = psi(e); lhs = xyzz_add(lhs, e);
        e = psi(e); lhs = xyzz_add(lhs, e);
        e = psi(e);
        return xyzz_same_point(lhs, xyzz_dbl(e));
    }
};

#if CG_WITH_BLS
template <> struct FastSubgroup<Fp2<Bls381Fq>> {
    typedef Fp2<Bls381Fq> F;
    static constexpr bool available = true;
    static constexpr unsigned long long X_ABS = 0xd201000000010000ull;       // x is negative
    PsiMap<Bls381Fq> psi;
    CG_HD bool contains(const Affine<F>& p) const {
        if (p.is_inf()) return true;
        const XYZZ<F> q = xyzz_mul_u64(p.x, p.y, X_ABS);
        return xyzz_same_point(psi(XYZZ<F>::from_affine(p)), q.neg());
    }
};
template <> struct FastSubgroup<Bls381Fq> {
    typedef Bls381Fq F;
    static constexpr bool available = true;
    static constexpr unsigned long long X_ABS = 0xd201000000010000ull;
    F beta;
    CG_HD bool contains(const Affine<F>& p) const {
        if (p.is_inf()) return true;
        const XYZZ<F> q = xyzz_mul_u64(xyzz_mul_u64(p.x, p.y, X_ABS), X_ABS);   // [x^2]P
        return xyzz_same_point(XYZZ<F>{p.x * beta, p.y, F::one(), F::one()}, q.neg());
    }
};
#endif

}  // namespace cg
