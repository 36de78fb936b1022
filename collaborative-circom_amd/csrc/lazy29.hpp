// Signed lazy-reduction field arithmetic on 9 limbs of 29 bits (256-bit fields), shared by the G1/G2 bucket accumulation
// (msm_kernels.hpp) and the NTT butterflies (ntt_kernels.hpp).
#pragma once
#include "field.hpp"

namespace cg {

template <class F> struct XYZZ;     // curve.hpp (only L29::dbl_affine needs it)

// ---------------------------------------------------------------------------------------------------------------------
// Signed lazy-reduction arithmetic on 9 limbs of 29 bits, used INSIDE the G1 bucket-accumulation kernel only.
//   value = sum l[k] * 2^(29k), limbs 0..7 nominally in [0, 2^29) ("normalised"), limb 8 signed; the value itself is only
//   known modulo p and lies in a small multiple of (-p, p) — no conditional subtractions anywhere in the mixed addition.
//   l29_mul(a, b) = (a*b + M*p) / 2^261 with 0 <= M < 2^261: result in (a*b/2^261, a*b/2^261 + p), normalised limbs.
//   Column accumulators are int64: <= 18 signed products of magnitude < 2^58 (+ carry) never overflow as long as every
//   multiplication operand has |limb| <= 2^29; differences of two normalised values satisfy that, sums of three do not and
//   are normalised first.  Internally values live in the 2^261 Montgomery domain (a loaded coordinate a*2^256 is unpacked
//   shifted left by 5 bits); l29_to_fp multiplies by 2^256/2^261 on the way out.
// Bounds for the mixed addition below (d = 0.2, all in units of p): products in (-d, 1+d); X in (-4, 2); Y in (-1.5, 1.5);
// P = U2 - X in (-2.3, 5.3); R = S2 - Y in (-1.8, 2.8); every |a*b| < 31 p^2 = 0.18 p * 2^261.
template <class F>
struct L29 {
    typedef typename F::Params P;
    static constexpr uint32_t MASK = (1u << 29) - 1;
    int32_t l[9];

    static constexpr int32_t pl(int k) { return (int32_t)F::p29(k); }
    // unpack a canonical field element (value < 2^256), optionally times 2^5
    template <int SHIFT>
    __device__ __forceinline__ static L29 unpack(const F& a) {
        L29 r;
        _Pragma("unroll") for (int k = 0; k < 9; k++) {
            const int bit = 29 * k - SHIFT;
            if (bit < 0) { r.l[k] = (int32_t)((a.v[0] << SHIFT) & MASK); continue; }
            const int w = bit >> 5, sh = bit & 31;
            uint64_t two = (uint64_t)a.v[w] | (w + 1 < F::N ? (uint64_t)a.v[w + 1] << 32 : 0);
            r.l[k] = (int32_t)((uint32_t)(two >> sh) & MASK);
        }
        if (SHIFT) r.l[8] = (int32_t)(a.v[7] >> (29 * 8 - SHIFT - 32 * 7));   // top limb keeps all remaining bits
        return r;
    }
    // floor(2^285 / p) (slightly under): quotient estimator of unpack_small.  p's top 64 bits give it to a relative 2^-62.
    static constexpr uint32_t quot_c() {
        unsigned __int128 top = ((unsigned __int128)P::P[F::N - 1] << 32) | P::P[F::N - 2];      // p >> (32 N - 64)
        return (uint32_t)((((unsigned __int128)1) << (285 - (32 * F::N - 64))) / (top + 1));
    }
    // The representative of 32*a (mod p) in [-p, p), normalised: a small accumulator coordinate straight from a table record
    // (a < p in the ABI's 2^256 Montgomery form), without a multiplication.  32a = x8 2^232 + low < 32p; k = floor(x8 c / 2^53)
    // with c = floor(2^285 / p) underestimates floor(32a / p) by at most one, so 32a - (k + 1) p lies in [-p, p).
    __device__ __forceinline__ static L29 unpack_small(const F& a) {
        static_assert(F::N == 8, "");
        const L29 x = unpack<5>(a);
        const int32_t k1 = (int32_t)(__umulhi((uint32_t)x.l[8], quot_c()) >> 21) + 1;           // <= 33
        L29 r; int64_t c = 0;
        _Pragma("unroll") for (int j = 0; j < 9; j++) {
            const int64_t t = (int64_t)x.l[j] - (int64_t)k1 * pl(j) + c;
            r.l[j] = j < 8 ? (int32_t)((uint32_t)t & MASK) : (int32_t)t;
            c = t >> 29;
        }
        return r;
    }
    __device__ __forceinline__ L29 operator+(const L29& b) const { L29 r; _Pragma("unroll") for (int k = 0; k < 9; k++) r.l[k] = l[k] + b.l[k]; return r; }
    __device__ __forceinline__ L29 operator-(const L29& b) const { L29 r; _Pragma("unroll") for (int k = 0; k < 9; k++) r.l[k] = l[k] - b.l[k]; return r; }
    __device__ __forceinline__ L29 neg() const { L29 r; _Pragma("unroll") for (int k = 0; k < 9; k++) r.l[k] = -l[k]; return r; }
    __device__ __forceinline__ L29 dbl() const { L29 r; _Pragma("unroll") for (int k = 0; k < 9; k++) r.l[k] = l[k] * 2; return r; }
    // carry propagation: limbs 0..7 into [0, 2^29), limb 8 signed
    __device__ __forceinline__ L29 norm() const {
        L29 r; int32_t c = 0;
        _Pragma("unroll") for (int k = 0; k < 8; k++) { int32_t t = l[k] + c; r.l[k] = t & (int32_t)MASK; c = t >> 29; }
        r.l[8] = l[8] + c;
        return r;
    }
    __device__ __forceinline__ static L29 mul(const L29& a, const L29& b) {
        int64_t T[18];
        _Pragma("unroll") for (int k = 0; k < 18; k++) T[k] = 0;
        _Pragma("unroll") for (int i = 0; i < 9; i++) {
            _Pragma("unroll") for (int j = 0; j < 9; j++) T[i + j] += (int64_t)a.l[i] * b.l[j];
            const int32_t m = (int32_t)(((uint32_t)T[i] * (P::INV & MASK)) & MASK);
            _Pragma("unroll") for (int j = 0; j < 9; j++) T[i + j] += (int64_t)m * pl(j);
            T[i + 1] += T[i] >> 29;                       // exact: T[i] is a multiple of 2^29
        }
        L29 r;
        _Pragma("unroll") for (int k = 0; k < 8; k++) { r.l[k] = (int32_t)((uint32_t)T[9 + k] & MASK); T[10 + k] += T[9 + k] >> 29; }
        r.l[8] = (int32_t)T[17];
        return r;
    }
    // cheap necessary condition for x = 0 (mod p) on an UNNORMALISED value: carries only travel upwards, so the lowest 29 bits of
    // limb 0 already are the normalised limb 0 and must equal limb 0 of one of the candidates k*p (a non-zero residue passes with
    // probability ~10 / 2^29).  Lets the hot loop skip the carry propagation it would otherwise do only for this test.
    __device__ __forceinline__ static bool maybe_zero_mod_p(const L29& x) {
        // x = k p with -3 <= k <= 6  =>  k = x p^-1 (mod 2^29) lies in that range; p^-1 = -INV (mod 2^29)
        const uint32_t k = ((uint32_t)x.l[0] * (0u - P::INV)) & MASK;
        return ((k + 3u) & MASK) < 10u;
    }
    // k*p in normalised limbs, k in [-3, 6]: the residues a normalised value in (-4p, 7p) takes when it is 0 mod p
    __device__ __forceinline__ static bool is_zero_mod_p(const L29& x /* normalised */) {
        // cheap filter on the lowest limb (a non-zero residue matches one of the candidates with probability ~10 / 2^29)
        bool maybe = false;
        _Pragma("unroll") for (int kk = -3; kk <= 6; kk++) maybe = maybe || (x.l[0] == (int32_t)(((int64_t)kk * pl(0)) & MASK));
        if (!maybe) return false;
        bool any = false;
        _Pragma("unroll") for (int kk = -3; kk <= 6; kk++) {
            // limbs of kk*p: carry-normalise kk * p29(j) on the fly (compile-time constants after unrolling)
            bool eq = true; int64_t c = 0;
            _Pragma("unroll") for (int j = 0; j < 9; j++) {
                int64_t t = (int64_t)kk * pl(j) + c;
                int32_t limb = j < 8 ? (int32_t)(t & MASK) : (int32_t)t;
                c = t >> 29;
                eq = eq && (x.l[j] == limb);
            }
            any = any || eq;
        }
        return any;
    }
    // (a*b - c*d + M p) / 2^261 with ONE reduction: 18 signed products + 9 reduction products per column (< 27 * 2^58 < 2^63)
    __device__ __forceinline__ static L29 mul_sub(const L29& a, const L29& b, const L29& c, const L29& d) {
        int64_t T[18];
        _Pragma("unroll") for (int k = 0; k < 18; k++) T[k] = 0;
        _Pragma("unroll") for (int i = 0; i < 9; i++) {
            _Pragma("unroll") for (int j = 0; j < 9; j++) T[i + j] += (int64_t)a.l[i] * b.l[j];
            _Pragma("unroll") for (int j = 0; j < 9; j++) T[i + j] += (int64_t)(-c.l[i]) * d.l[j];
            const int32_t m = (int32_t)(((uint32_t)T[i] * (P::INV & MASK)) & MASK);
            _Pragma("unroll") for (int j = 0; j < 9; j++) T[i + j] += (int64_t)m * pl(j);
            T[i + 1] += T[i] >> 29;
        }
        L29 r;
        _Pragma("unroll") for (int k = 0; k < 8; k++) { r.l[k] = (int32_t)((uint32_t)T[9 + k] & MASK); T[10 + k] += T[9 + k] >> 29; }
        r.l[8] = (int32_t)T[17];
        return r;
    }
    // squaring: the 36 off-diagonal products are taken once, doubled (45 instead of 81 multiplies before the reduction).
    // Column i is complete before round i uses it: a pair (x, y), x <= y, x + y = i is added in round x <= i/2.
    __device__ __forceinline__ static L29 sqr(const L29& a) {
        int64_t T[18];
        _Pragma("unroll") for (int k = 0; k < 18; k++) T[k] = 0;
        int32_t a2[9];
        _Pragma("unroll") for (int k = 0; k < 9; k++) a2[k] = a.l[k] * 2;
        _Pragma("unroll") for (int i = 0; i < 9; i++) {
            T[2 * i] += (int64_t)a.l[i] * a.l[i];
            _Pragma("unroll") for (int j = i + 1; j < 9; j++) T[i + j] += (int64_t)a2[i] * a.l[j];
            const int32_t m = (int32_t)(((uint32_t)T[i] * (P::INV & MASK)) & MASK);
            _Pragma("unroll") for (int j = 0; j < 9; j++) T[i + j] += (int64_t)m * pl(j);
            T[i + 1] += T[i] >> 29;
        }
        L29 r;
        _Pragma("unroll") for (int k = 0; k < 8; k++) { r.l[k] = (int32_t)((uint32_t)T[9 + k] & MASK); T[10 + k] += T[9 + k] >> 29; }
        r.l[8] = (int32_t)T[17];
        return r;
    }
    __device__ __forceinline__ static XYZZ<F> dbl_affine(const F& x, const F& y) { return xyzz_dbl_affine(x, y); }
    // small representative (0.5p .. 1.6p) of 1 in the 2^261 domain: (32 R1) * (32 R1) / 2^261 = 2^261 (mod p), R1 = 2^256 mod p
    __device__ __forceinline__ static L29 one() { const L29 o = unpack<5>(F::one()); return mul(o, o); }
    // the same residue in [0, p) without a product (constant-folded: F::one() is a compile-time constant)
    __device__ __forceinline__ static L29 one_small() {
        L29 o = unpack_small(F::one());
        L29 pp; _Pragma("unroll") for (int k = 0; k < 9; k++) pp.l[k] = pl(k);
        const L29 t = (o + pp).norm();
        if (o.l[8] < 0) o = t;
        return o;
    }
    // back to a canonical field element in the ABI's 2^256 Montgomery domain; |value| < 8p
    __device__ __forceinline__ static F to_fp(const L29& x) {
        L29 c; _Pragma("unroll") for (int k = 0; k < 9; k++) c.l[k] = 0;
        c.l[8] = 1 << 24;                                  // 2^256 as an integer: x * 2^256 / 2^261 = x / 32
        return pack_reduced(mul(x, c));                    // in (-0.3p, 1.3p), normalised
    }
    // normalised value in (-p, 2p) -> the canonical element with the same residue, packed into 8 x 32 bits
    __device__ __forceinline__ static F pack_reduced(L29 y) {
        // add p if negative, subtract p if >= p (sign of the top limb after normalisation decides)
        L29 pp; _Pragma("unroll") for (int k = 0; k < 9; k++) pp.l[k] = pl(k);
        L29 t = (y + pp).norm();
        if (y.l[8] < 0) y = t;
        t = (y - pp).norm();
        if (t.l[8] >= 0) y = t;
        F r;
        _Pragma("unroll") for (int w = 0; w < F::N; w++) {
            const int k = (32 * w) / 29, sh = 32 * w - 29 * k;
            r.v[w] = ((uint32_t)y.l[k] >> sh) | ((uint32_t)y.l[k + 1] << (29 - sh));
        }
        return r;
    }
};

}  // namespace cg
