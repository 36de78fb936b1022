// Host-side O(1) curve arithmetic on 64-bit limbs: the scalar multiplications of proof assembly (groth16.rs:258-312 — r*delta, s*delta,
// s*g_a, r*g1_b, the public-input terms of calculate_coeff :220, the masking points of scalar_mul rep3.rs:835-847).  A party makes ~18 of
// them per proof; on the 32-bit-limb field code the device kernels share with the host (field.hpp) they cost 2.3 + 2.4 ms per proof
// (2 x EPYC 9575F) — hidden under the MSMs at 2^22 constraints, HALF of the proof at 2^16.  Here: Montgomery products on N x 64-bit
// limbs (unsigned __int128), Jacobian coordinates for a = 0 curves (dbl-2009-l, add-2007-bl, madd-2007-bl), 4-bit fixed windows for a
// variable base, and 8-bit window tables (32 x 255 affine multiples, one mixed addition per scalar byte) for bases that are fixed for the
// life of a session: delta_1, delta_2, the generators, the public-input records of a / b1 / b2.
// Values are the ABI's: little-endian limbs in Montgomery form with R = 2^(64 N) (= 2^256 / 2^384: the same bytes as the 32-bit-limb
// representation), Jacobian (X, Y, Z) with Z = 0 for the point at infinity.  The group element computed is the same whatever the
// window; coordinates of the same point may differ from the 32-bit path's by a projective factor, as between any two implementations.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace cg64 {

typedef unsigned __int128 u128;

template <int N>
struct Mod {
    uint64_t p[N]; uint64_t inv;             // modulus, -p^-1 mod 2^64
    uint64_t one[N];                         // R mod p
    void init(const uint32_t* p32) {
        for (int i = 0; i < N; i++) p[i] = (uint64_t)p32[2 * i] | (uint64_t)p32[2 * i + 1] << 32;
        uint64_t x = 1;                      // Newton: x <- x (2 - p x), doubles the correct low bits each round
        for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
        inv = 0 - x;
        // R mod p = 2^(64 N) mod p by N * 64 doublings of 1
        uint64_t r[N]; for (int i = 0; i < N; i++) r[i] = i == 0;
        for (int b = 0; b < 64 * N; b++) {
            uint64_t top = r[N - 1] >> 63;
            for (int i = N - 1; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
            r[0] <<= 1;
            if (top || geq(r, p)) sub_n(r, r, p);
        }
        memcpy(one, r, sizeof r);
    }
    static bool geq(const uint64_t* a, const uint64_t* b) { for (int i = N - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; } return true; }
    static uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b) { uint64_t br = 0; for (int i = 0; i < N; i++) { const u128 t = (u128)a[i] - b[i] - br; r[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; } return br; }
    static uint64_t add_n(uint64_t* r, const uint64_t* a, const uint64_t* b) { u128 c = 0; for (int i = 0; i < N; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; } return (uint64_t)c; }
};

// prime field element bound to a modulus held by the tag type T (static const Mod<N>& mod())
template <class T>
struct Fp {
    static constexpr int N = T::N;
    uint64_t v[N];
    static Fp zero() { Fp r; for (int i = 0; i < N; i++) r.v[i] = 0; return r; }
    static Fp one() { Fp r; memcpy(r.v, T::mod().one, sizeof r.v); return r; }
    bool is_zero() const { uint64_t o = 0; for (int i = 0; i < N; i++) o |= v[i]; return o == 0; }
    bool operator==(const Fp& b) const { return memcmp(v, b.v, sizeof v) == 0; }
    Fp operator+(const Fp& b) const { Fp r; const uint64_t c = Mod<N>::add_n(r.v, v, b.v); if (c || Mod<N>::geq(r.v, T::mod().p)) Mod<N>::sub_n(r.v, r.v, T::mod().p); return r; }
    Fp operator-(const Fp& b) const { Fp r; if (Mod<N>::sub_n(r.v, v, b.v)) Mod<N>::add_n(r.v, r.v, T::mod().p); return r; }
    Fp neg() const { return is_zero() ? *this : zero() - *this; }
    Fp dbl() const { return *this + *this; }
    Fp operator*(const Fp& b) const {                                     // CIOS Montgomery product
        const Mod<N>& M = T::mod();
        uint64_t t[N + 2]; for (int i = 0; i < N + 2; i++) t[i] = 0;
        for (int i = 0; i < N; i++) {
            u128 c = 0;
            for (int j = 0; j < N; j++) { c += (u128)v[i] * b.v[j] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N] = (uint64_t)c; t[N + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * M.inv;
            c = (u128)m * M.p[0] + t[0]; c >>= 64;
            for (int j = 1; j < N; j++) { c += (u128)m * M.p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N - 1] = (uint64_t)c; t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        Fp r; memcpy(r.v, t, sizeof r.v);
        if (t[N] || Mod<N>::geq(r.v, M.p)) Mod<N>::sub_n(r.v, r.v, M.p);
        return r;
    }
    Fp sqr() const { return *this * *this; }
    Fp from_mont() const { Fp o = zero(); o.v[0] = 1; return *this * o; }
    Fp inverse() const {                                                  // a^(p-2)
        uint64_t e[N]; memcpy(e, T::mod().p, sizeof e); e[0] -= 2;        // p is odd and > 2: no borrow
        Fp r = one();
        for (int i = 64 * N - 1; i >= 0; i--) { r = r.sqr(); if ((e[i / 64] >> (i % 64)) & 1) r = r * *this; }
        return r;
    }
};
// Fp[u] / (u^2 + 1): the quadratic extension both supported curves build G2 on
template <class B>
struct Fp2 {
    B c0, c1;
    static Fp2 zero() { return {B::zero(), B::zero()}; }
    static Fp2 one() { return {B::one(), B::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2& b) const { return c0 == b.c0 && c1 == b.c1; }
    Fp2 operator+(const Fp2& b) const { return {c0 + b.c0, c1 + b.c1}; }
    Fp2 operator-(const Fp2& b) const { return {c0 - b.c0, c1 - b.c1}; }
    Fp2 neg() const { return {c0.neg(), c1.neg()}; }
    Fp2 conj() const { return {c0, c1.neg()}; }
    Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fp2 operator*(const Fp2& b) const { const B t0 = c0 * b.c0, t1 = c1 * b.c1; return {t0 - t1, (c0 + c1) * (b.c0 + b.c1) - t0 - t1}; }
    Fp2 sqr() const { return {(c0 + c1) * (c0 - c1), (c0 * c1).dbl()}; }
    Fp2 inverse() const { const B n = (c0.sqr() + c1.sqr()).inverse(); return {c0 * n, (c1 * n).neg()}; }
};

template <class F> struct Jac { F x, y, z; bool is_inf() const { return z.is_zero(); } static Jac inf() { return {F::one(), F::one(), F::zero()}; } };
template <class F> struct Aff { F x, y; };

template <class F>
Jac<F> dbl(const Jac<F>& p) {                                             // dbl-2009-l (a = 0): 2M + 5S
    if (p.is_inf() || p.y.is_zero()) return Jac<F>::inf();
    const F A = p.x.sqr(), B = p.y.sqr(), C = B.sqr();
    const F D = ((p.x + B).sqr() - A - C).dbl();
    const F E = A.dbl() + A, Fq = E.sqr();
    const F X3 = Fq - D.dbl();
    const F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
    return {X3, Y3, (p.y * p.z).dbl()};
}
template <class F>
Jac<F> add(const Jac<F>& p, const Jac<F>& q) {                            // add-2007-bl: 11M + 5S
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    const F Z1Z1 = p.z.sqr(), Z2Z2 = q.z.sqr();
    const F U1 = p.x * Z2Z2, U2 = q.x * Z1Z1;
    const F S1 = p.y * q.z * Z2Z2, S2 = q.y * p.z * Z1Z1;
    const F H = U2 - U1, rr = (S2 - S1).dbl();
    if (H.is_zero()) return rr.is_zero() ? dbl(p) : Jac<F>::inf();
    const F I = H.dbl().sqr(), J = H * I, V = U1 * I;
    const F X3 = rr.sqr() - J - V.dbl();
    const F Y3 = rr * (V - X3) - (S1 * J).dbl();
    return {X3, Y3, ((p.z + q.z).sqr() - Z1Z1 - Z2Z2) * H};
}
template <class F>
Jac<F> madd(const Jac<F>& p, const Aff<F>& q) {                           // madd-2007-bl (Z2 = 1): 7M + 4S
    if (p.is_inf()) return {q.x, q.y, F::one()};
    const F Z1Z1 = p.z.sqr();
    const F U2 = q.x * Z1Z1, S2 = q.y * p.z * Z1Z1;
    const F H = U2 - p.x, rr = (S2 - p.y).dbl();
    if (H.is_zero()) return rr.is_zero() ? dbl(p) : Jac<F>::inf();
    const F HH = H.sqr(), I = HH.dbl().dbl(), J = H * I, V = p.x * I;
    const F X3 = rr.sqr() - J - V.dbl();
    const F Y3 = rr * (V - X3) - (p.y * J).dbl();
    return {X3, Y3, (p.z + H).sqr() - Z1Z1 - HH};
}
// affine (x, y) = (X / Z^2, Y / Z^3), the point at infinity as (0, 0): what crosses the wire and what a proof holds
template <class F>
Aff<F> to_affine(const Jac<F>& p) { if (p.is_inf()) return {F::zero(), F::zero()}; const F iz = p.z.inverse(), iz2 = iz.sqr(); return {p.x * iz2, p.y * iz2 * iz}; }
template <class F>
Jac<F> neg(const Jac<F>& p) { return {p.x, p.y.neg(), p.z}; }
template <class F>
bool same_point(const Jac<F>& a, const Jac<F>& b) {
    if (a.is_inf() || b.is_inf()) return a.is_inf() && b.is_inf();
    const F za = a.z.sqr(), zb = b.z.sqr();
    return a.x * zb == b.x * za && a.y * zb * b.z == b.y * za * a.z;
}
// k * p for a 64-bit k, double-and-add from the top bit (the subgroup tests: k = the curve parameter)
template <class F>
Jac<F> mul_u64(const Aff<F>& p, uint64_t k) { Jac<F> r = Jac<F>::inf(); for (int i = 63; i >= 0; i--) { r = dbl(r); if ((k >> i) & 1) r = madd(r, p); } return r; }
template <class F>
Jac<F> mul_u64(const Jac<F>& p, uint64_t k) { Jac<F> r = Jac<F>::inf(); for (int i = 63; i >= 0; i--) { r = dbl(r); if ((k >> i) & 1) r = add(r, p); } return r; }
// XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; all-zero ZZ = infinity): what the reduction kernels hand the host, folded here
// (msm_end_impl: ~100 additions per MSM result) on the same 64-bit limbs
template <class F> struct Xyzz { F x, y, zz, zzz; bool is_inf() const { return zz.is_zero(); } static Xyzz inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; } };
template <class F>
Xyzz<F> dbl(const Xyzz<F>& p) {                                           // dbl-2008-s-1 (a = 0)
    if (p.is_inf() || p.y.is_zero()) return Xyzz<F>::inf();
    const F U = p.y.dbl(), V = U.sqr(), W = U * V, S = p.x * V;
    const F xx = p.x.sqr(), M = xx.dbl() + xx;
    const F X3 = M.sqr() - S.dbl();
    return {X3, M * (S - X3) - W * p.y, V * p.zz, W * p.zzz};
}
template <class F>
Xyzz<F> add(const Xyzz<F>& a, const Xyzz<F>& b) {                          // add-2008-s
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    const F U1 = a.x * b.zz, U2 = b.x * a.zz, S1 = a.y * b.zzz, S2 = b.y * a.zzz;
    const F P = U2 - U1, R = S2 - S1;
    if (P.is_zero()) return R.is_zero() ? dbl(a) : Xyzz<F>::inf();
    const F PP = P.sqr(), PPP = P * PP, Q = U1 * PP;
    const F X3 = R.sqr() - PPP - Q.dbl();
    return {X3, R * (Q - X3) - S1 * PPP, a.zz * b.zz * PP, a.zzz * b.zzz * PPP};
}
// Jacobian (X', Y', Z') with Z' = ZZZ: X' = X ZZ^2, Y' = Y ZZZ^2 (curve.hpp: xyzz_to_jacobian)
template <class F>
Jac<F> to_jac(const Xyzz<F>& p) { if (p.is_inf()) return Jac<F>::inf(); return {p.x * p.zz.sqr(), p.y * p.zzz.sqr(), p.zzz}; }

// k (canonical little-endian 64-bit words, NK of them) times p: 4-bit fixed windows, top down
template <class F>
Jac<F> scalar_mul(const Jac<F>& p, const uint64_t* k, int nk) {
    if (p.is_inf()) return p;
    Jac<F> tab[16];
    tab[1] = p; tab[2] = dbl(p);
    for (int i = 3; i < 16; i++) tab[i] = add(tab[i - 1], p);
    Jac<F> acc = Jac<F>::inf(); bool started = false;
    for (int w = nk * 16 - 1; w >= 0; w--) {
        const unsigned d = (unsigned)(k[w / 16] >> (4 * (w % 16))) & 15u;
        if (started) { acc = dbl(dbl(dbl(dbl(acc)))); }
        if (d) { acc = started ? add(acc, tab[d]) : tab[d]; started = true; }
    }
    return acc;
}
// fixed base: tab[w][d - 1] = d * 2^(8 w) * P in affine form, w < 4 nk... (8 windows per 64-bit word), d = 1 .. 255
template <class F>
struct FixedBase {
    int nk = 0; bool base_inf = false;
    std::vector<Aff<F>> tab;
    void build(const Jac<F>& p, int nk_) {
        nk = nk_; base_inf = p.is_inf();
        if (base_inf) return;
        const int W = 8 * nk;
        std::vector<Jac<F>> j((size_t)W * 255);
        Jac<F> base = p;
        for (int w = 0; w < W; w++) {
            Jac<F>* row = j.data() + (size_t)w * 255;
            row[0] = base;
            for (int d = 1; d < 255; d++) row[d] = add(row[d - 1], base);
            base = add(row[254], base);                               // 256 * base
        }
        // batched conversion to affine: one inversion for all (Montgomery's trick); a point at infinity in the table (a base of small
        // order: never for a subgroup point with a 254-bit order) is stored as (0, 0) and skipped by mul
        std::vector<F> pre(j.size());
        F run = F::one();
        for (size_t i = 0; i < j.size(); i++) { pre[i] = run; if (!j[i].is_inf()) run = run * j[i].z; }
        F inv = run.inverse();
        tab.resize(j.size());
        for (size_t i = j.size(); i-- > 0;) {
            if (j[i].is_inf()) { tab[i] = {F::zero(), F::zero()}; continue; }
            const F iz = inv * pre[i]; inv = inv * j[i].z;
            const F iz2 = iz.sqr();
            tab[i] = {j[i].x * iz2, j[i].y * iz2 * iz};
        }
    }
    Jac<F> mul(const uint64_t* k) const {
        Jac<F> acc = Jac<F>::inf();
        if (base_inf) return acc;
        for (int w = 0; w < 8 * nk; w++) {
            const unsigned d = (unsigned)(k[w / 8] >> (8 * (w % 8))) & 255u;
            if (!d) continue;
            const Aff<F>& e = tab[(size_t)w * 255 + (d - 1)];
            if (e.x.is_zero() && e.y.is_zero()) continue;
            acc = madd(acc, e);
        }
        return acc;
    }
};

}  // namespace cg64
