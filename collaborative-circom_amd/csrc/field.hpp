// Prime-field arithmetic for gfx950 (and, for O(1) host-side work, the same code compiled for the host).
//
// Replaces what the reference gets from ark-ff 0.4.2 (`/root/reference/Cargo.toml:36`) on the co-groth16 path.
// Data convention = the reference's in-memory one (`/root/reference/co-circom/circom-types/src/traits.rs:57-67`):
// little-endian limbs in Montgomery form with R = 2^256 (BN254 Fr/Fq, BLS12-381 Fr) or 2^384 (BLS12-381 Fq), fully reduced.
// On the device an element is N x u32 (N = 8 or 12): CDNA4's integer multiplier is 32x32 (`v_mad_u64_u32`), there is no
// 64-bit multiply, so 32-bit limbs are the native width.  No MFMA, no floating point.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "field_consts.hpp"

#define CG_HD __host__ __device__ __forceinline__
#ifndef CG_MUL29
#define CG_MUL29 1
#endif
#define CG_HD_NOINLINE __host__ __device__ __attribute__((noinline))
// build-time experiment knobs (scripts/microbench.hip): how Fq2 products and base-field products are emitted
#if defined(CG_FP2_INLINE)
#define CG_FP2_OP CG_HD
#else
#define CG_FP2_OP CG_HD_NOINLINE
#endif
#if defined(CG_FPMUL_NOINLINE)
#define CG_FPMUL_OP CG_HD_NOINLINE
#else
#define CG_FPMUL_OP CG_HD
#endif

namespace cg {

template <class P>
struct alignas(16) Fp {
    static constexpr int N = P::N;
    typedef P Params;
    uint32_t v[N];

    CG_HD static Fp zero() { Fp r; _Pragma("unroll") for (int i = 0; i < N; i++) r.v[i] = 0; return r; }
    CG_HD static Fp one() { Fp r; _Pragma("unroll") for (int i = 0; i < N; i++) r.v[i] = P::R1[i]; return r; }
    CG_HD static Fp r2() { Fp r; _Pragma("unroll") for (int i = 0; i < N; i++) r.v[i] = P::R2[i]; return r; }

    CG_HD bool is_zero() const { uint32_t o = 0; _Pragma("unroll") for (int i = 0; i < N; i++) o |= v[i]; return o == 0; }
    CG_HD bool operator==(const Fp& b) const { uint32_t o = 0; _Pragma("unroll") for (int i = 0; i < N; i++) o |= v[i] ^ b.v[i]; return o == 0; }
    CG_HD bool operator!=(const Fp& b) const { return !(*this == b); }

    // r = a - p if a >= p else a   (a < 2p)
    CG_HD static Fp reduce_once(const uint32_t (&t)[N]) {
        uint32_t d[N]; uint32_t borrow = 0;
        _Pragma("unroll") for (int i = 0; i < N; i++) {
            uint64_t x = (uint64_t)t[i] - P::P[i] - borrow;
            d[i] = (uint32_t)x; borrow = (uint32_t)(x >> 32) & 1u;
        }
        Fp r;
        _Pragma("unroll") for (int i = 0; i < N; i++) r.v[i] = borrow ? t[i] : d[i];
        return r;
    }
    CG_HD Fp operator+(const Fp& b) const {
        uint32_t t[N]; uint32_t c = 0;
        _Pragma("unroll") for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)v[i] + b.v[i] + c; t[i] = (uint32_t)x; c = (uint32_t)(x >> 32); }
        return reduce_once(t);   // p < 2^(32N-1): no carry out of the top limb
    }
    CG_HD Fp operator-(const Fp& b) const {
        uint32_t t[N]; uint32_t borrow = 0;
        _Pragma("unroll") for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)v[i] - b.v[i] - borrow; t[i] = (uint32_t)x; borrow = (uint32_t)(x >> 32) & 1u; }
        Fp r; uint32_t c = 0;
        _Pragma("unroll") for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)t[i] + (borrow ? P::P[i] : 0u) + c; r.v[i] = (uint32_t)x; c = (uint32_t)(x >> 32); }
        return r;
    }
    CG_HD Fp neg() const { return zero() - *this; }
    CG_HD Fp dbl() const { return *this + *this; }

    // Montgomery product.  Two implementations with identical results (fully reduced in, fully reduced out):
    //  * 256-bit fields (N == 8): reduced-radix core, 9 limbs of 29 bits.  Every partial product is accumulated through
    //    v_mad_u64_u32's free 64-bit addend into 18 column accumulators that cannot overflow (<= 18 terms of < 2^58), so there
    //    are no carries and none of the pair-building v_mov traffic the saturated form needs on gfx950 (64-bit operands must sit
    //    in even-aligned register pairs): ~310 VALU instructions instead of ~520, 171 instead of 136 multiplies.
    //    The core reduces by 2^261; operand b is unpacked pre-multiplied by 2^5 so the result is a*b/2^256 as the ABI requires.
    //  * other widths (N == 12, BLS12-381 Fq): coarsely integrated operand scanning on 32-bit limbs.
    CG_FPMUL_OP Fp operator*(const Fp& b) const {
        if constexpr (N == 8 && CG_MUL29) return mul29(*this, b);
        else return mul_cios(*this, b);
    }

    static constexpr uint32_t MASK29 = (1u << 29) - 1;
    // k-th 29-bit limb of the modulus
    static constexpr uint32_t p29(int k) {
        const int bit = 29 * k, w = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)P::P[w] | (w + 1 < N ? (uint64_t)P::P[w + 1] << 32 : 0);
        return (uint32_t)(two >> sh) & MASK29;
    }
    CG_HD static Fp mul29(const Fp& a, const Fp& b) {
        uint32_t x[9], y[9];
        // a: limb k = bits [29k, 29k+29)
        _Pragma("unroll") for (int k = 0; k < 9; k++) {
            const int bit = 29 * k, w = bit >> 5, sh = bit & 31;
            uint64_t two = (uint64_t)a.v[w] | (w + 1 < N ? (uint64_t)a.v[w + 1] << 32 : 0);
            x[k] = (uint32_t)(two >> sh) & MASK29;
        }
        // 32*b: limb k = bits [29k-5, 29k+24) of b
        y[0] = (b.v[0] << 5) & MASK29;
        _Pragma("unroll") for (int k = 1; k < 9; k++) {
            const int bit = 29 * k - 5, w = bit >> 5, sh = bit & 31;
            uint64_t two = (uint64_t)b.v[w] | (w + 1 < N ? (uint64_t)b.v[w + 1] << 32 : 0);
            y[k] = (uint32_t)(two >> sh) & MASK29;
        }
        uint64_t T[18];
        _Pragma("unroll") for (int k = 0; k < 18; k++) T[k] = 0;
        _Pragma("unroll") for (int i = 0; i < 9; i++) {
            _Pragma("unroll") for (int j = 0; j < 9; j++) T[i + j] += (uint64_t)x[i] * y[j];
            const uint32_t m = ((uint32_t)T[i] * (P::INV & MASK29)) & MASK29;       // -p^-1 mod 2^29
            _Pragma("unroll") for (int j = 0; j < 9; j++) T[i + j] += (uint64_t)m * p29(j);
            T[i + 1] += T[i] >> 29;                                                   // T[i] is now a multiple of 2^29
        }
        uint32_t r[9];
        _Pragma("unroll") for (int k = 0; k < 9; k++) {
            r[k] = (uint32_t)T[9 + k] & MASK29;
            if (k < 8) T[10 + k] += T[9 + k] >> 29;
        }
        // (a * 32b + m p) / 2^261 < 2p < 2^255: repack into 8 x 32 and subtract p once if needed
        uint32_t t[N];
        _Pragma("unroll") for (int w = 0; w < N; w++) {
            const int k = (32 * w) / 29, sh = 32 * w - 29 * k;                        // sh <= 21: two limbs cover the word
            t[w] = (r[k] >> sh) | (r[k + 1] << (29 - sh));
        }
        return reduce_once(t);
    }
    CG_HD static Fp mul_cios(const Fp& a, const Fp& b) {
        // Invariant: the running value t stays < 2p < 2^(32N) between rounds (p < 2^(32N-1)), so N limbs hold it.
        uint32_t t[N];
        _Pragma("unroll") for (int j = 0; j < N; j++) t[j] = 0;
        _Pragma("unroll") for (int i = 0; i < N; i++) {
            uint64_t c = 0;
            _Pragma("unroll") for (int j = 0; j < N; j++) {
                c += (uint64_t)a.v[j] * b.v[i] + t[j];
                t[j] = (uint32_t)c; c >>= 32;
            }
            uint32_t top = (uint32_t)c;   // t + a*b_i < p*(2^32+1): one extra limb
            uint32_t m = t[0] * P::INV;
            uint64_t d = (uint64_t)m * P::P[0] + t[0];
            d >>= 32;
            _Pragma("unroll") for (int j = 1; j < N; j++) {
                d += (uint64_t)m * P::P[j] + t[j];
                t[j - 1] = (uint32_t)d; d >>= 32;
            }
            d += top;
            t[N - 1] = (uint32_t)d;   // (t + a*b_i + m*p) / 2^32 < 2p: no further carry
        }
        return reduce_once(t);
    }
    CG_HD Fp sqr() const { return (*this) * (*this); }

    // Montgomery -> canonical integer (multiplication by raw 1)
    CG_HD Fp from_mont() const { Fp o = zero(); o.v[0] = 1; return (*this) * o; }
    CG_HD Fp to_mont() const { return (*this) * r2(); }
};

typedef Fp<Bn254FrP> Bn254Fr;
typedef Fp<Bn254FqP> Bn254Fq;
typedef Fp<Bls381FrP> Bls381Fr;
typedef Fp<Bls381FqP> Bls381Fq;

// a^e, e = canonical little-endian 32-bit limbs (host-side O(1) helper; also usable on device)
template <class F>
CG_HD F fp_pow(const F& a, const uint32_t* e, int nlimbs) {
    F r = F::one();
    for (int i = nlimbs * 32 - 1; i >= 0; i--) {
        r = r.sqr();
        if ((e[i / 32] >> (i % 32)) & 1u) r = r * a;
    }
    return r;
}
template <class F>
CG_HD F fp_inverse(const F& a) {   // Fermat; inverse(0) = 0
    uint32_t e[F::N]; uint32_t borrow = 2;
    for (int i = 0; i < F::N; i++) { uint64_t x = (uint64_t)F::Params::P[i] - borrow; e[i] = (uint32_t)x; borrow = (uint32_t)(x >> 32) & 1u; }
    return fp_pow(a, e, F::N);
}

// Fp2 = Fp[u]/(u^2 + 1) (BN254 and BLS12-381)
template <class F>
struct alignas(16) Fp2 {
    typedef F Base;
    F c0, c1;
    CG_HD static Fp2 zero() { return {F::zero(), F::zero()}; }
    CG_HD static Fp2 one() { return {F::one(), F::zero()}; }
    CG_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    CG_HD bool operator==(const Fp2& b) const { return c0 == b.c0 && c1 == b.c1; }
    CG_HD bool operator!=(const Fp2& b) const { return !(*this == b); }
    CG_HD Fp2 operator+(const Fp2& b) const { return {c0 + b.c0, c1 + b.c1}; }
    CG_HD Fp2 operator-(const Fp2& b) const { return {c0 - b.c0, c1 - b.c1}; }
    CG_HD Fp2 neg() const { return {c0.neg(), c1.neg()}; }
    CG_HD Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    // Out of line (three base-field products per call; G2 formulas call this 8-12 times, inlining them all forces 1 wave/SIMD),
    // with operands passed BY VALUE so that they travel in VGPRs: by-reference operands of an out-of-line function have to
    // live in scratch memory, which costs ~2 KB of scratch traffic per G2 mixed addition.
    static CG_FP2_OP Fp2 mul_impl(Fp2 a, Fp2 b) {
        F t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
        F e = (a.c0 + a.c1) * (b.c0 + b.c1);
        return {t0 - t1, e - t0 - t1};
    }
    static CG_FP2_OP Fp2 sqr_impl(Fp2 a) {
        F s = a.c0 + a.c1, d = a.c0 - a.c1, m = a.c0 * a.c1;
        return {s * d, m.dbl()};
    }
    CG_HD Fp2 operator*(const Fp2& b) const { return mul_impl(*this, b); }
    CG_HD Fp2 sqr() const { return sqr_impl(*this); }
};

template <class F>
CG_HD Fp2<F> fp_inverse(const Fp2<F>& a) {
    F n = fp_inverse(a.c0.sqr() + a.c1.sqr());
    return {a.c0 * n, (a.c1 * n).neg()};
}

}  // namespace cg
