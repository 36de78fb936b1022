// Signed lazy-reduction field arithmetic on reduced-radix limbs, shared by the G1/G2 bucket accumulation (msm_kernels.hpp) and the NTT
// butterflies (ntt_kernels.hpp).  256-bit fields (BN254 Fq/Fr, BLS12-381 Fr): 9 limbs of 29 bits; BLS12-381 Fq (381 bits): 14 limbs of 28.
#pragma once
#include "field.hpp"

namespace cg {

template <class F> struct XYZZ;     // curve.hpp (only L29::dbl_affine needs it)

// ---------------------------------------------------------------------------------------------------------------------
// value = sum l[k] * 2^(W k), limbs 0..NL-2 nominally in [0, 2^W) ("normalised"), the top limb signed; the value itself is only
// known modulo p and lies in a small multiple of (-p, p) — no conditional subtractions anywhere in the mixed addition.
//   mul(a, b) = (a*b + M*p) / 2^(NL W) with 0 <= M < 2^(NL W): result in (a*b/2^(NL W), a*b/2^(NL W) + p), normalised limbs.
//   Column accumulators are int64: <= 2 NL signed products of magnitude < 2^(2W) (+ carry) never overflow as long as every
//   multiplication operand has |limb| <= 2^W (9 x 29: 18 * 2^58; 14 x 28: 28 * 2^56, and 42 * 2^56 for the fused two-product forms);
//   differences of two normalised values satisfy that, sums of three do not and are normalised first.  Internally values live in the
//   2^(NL W) Montgomery domain (2^261 / 2^392): a loaded coordinate a*2^(32N) is unpacked shifted left by SH = NL W - 32 N bits
//   (5 / 8); to_fp multiplies by 2^(32N)/2^(NL W) on the way out.
// Bounds for the mixed addition (d = 0.2, all in units of p): products in (-d, 1+d); X in (-4, 2); Y in (-1.5, 1.5);
// P = U2 - X in (-2.3, 5.3); R = S2 - Y in (-1.8, 2.8); every |a*b| < 31 p^2, far below p * 2^(NL W) (2^261 = 169 p for BN254,
// 2^392 = 2 521 p for BLS12-381 Fq).
template <class F>
struct L29 {
    typedef typename F::Params P;
    static constexpr int NL = F::N == 8 ? 9 : 14;                 // limbs
    static constexpr int W = F::N == 8 ? 29 : 28;                 // bits per limb
    static constexpr int SH = NL * W - 32 * F::N;                 // 2^(NL W) / 2^(32 N)
    static constexpr int TOPBIT = W * (NL - 1);
    static_assert(F::N == 8 || F::N == 12, "");
    static constexpr uint32_t MASK = (1u << W) - 1;
    int32_t l[NL];

    // k-th limb of the modulus
    static constexpr int32_t pl(int k) {
        const int bit = W * k, w = bit >> 5, sh = bit & 31;
        const uint64_t two = (w < F::N ? (uint64_t)P::P[w] : 0) | (w + 1 < F::N ? (uint64_t)P::P[w + 1] << 32 : 0);
        return (int32_t)((uint32_t)(two >> sh) & MASK);
    }
    // unpack a canonical field element (value < 2^(32N)), optionally times 2^SHIFT
    template <int SHIFT>
    __device__ __forceinline__ static L29 unpack(const F& a) {
        L29 r;
        _Pragma("unroll") for (int k = 0; k < NL; k++) {
            const int bit = W * k - SHIFT;
            if (bit < 0) { r.l[k] = (int32_t)((a.v[0] << SHIFT) & MASK); continue; }
            const int w = bit >> 5, sh = bit & 31;
            uint64_t two = (uint64_t)a.v[w] | (w + 1 < F::N ? (uint64_t)a.v[w + 1] << 32 : 0);
            r.l[k] = (k == NL - 1 && SHIFT) ? (int32_t)(uint32_t)(two >> sh)          // the top limb keeps all remaining bits
                                            : (int32_t)((uint32_t)(two >> sh) & MASK);
        }
        return r;
    }
    // floor(2^(TOPBIT + QS) / p) (slightly under): quotient estimator of unpack_small.  p's top 64 bits give it to a relative 2^-60.
    static constexpr int QS = F::N == 8 ? 53 : 48;
    static constexpr uint32_t quot_c() {
        unsigned __int128 top = ((unsigned __int128)P::P[F::N - 1] << 32) | P::P[F::N - 2];      // p >> (32 N - 64)
        return (uint32_t)((((unsigned __int128)1) << (TOPBIT + QS - (32 * F::N - 64))) / (top + 1));
    }
    // The representative of 2^SH * a (mod p) in [-p, p), normalised: a small accumulator coordinate straight from a table record
    // (a < p in the ABI's 2^(32N) Montgomery form), without a multiplication.  2^SH a = x_top 2^TOPBIT + low < 2^SH p;
    // k = floor(x_top c / 2^QS) with c = floor(2^(TOPBIT + QS) / p) underestimates floor(2^SH a / p) by at most one, so
    // 2^SH a - (k + 1) p lies in [-p, p).
    __device__ __forceinline__ static L29 unpack_small(const F& a) {
        const L29 x = unpack<SH>(a);
        const int32_t k1 = (int32_t)(((uint64_t)(uint32_t)x.l[NL - 1] * quot_c()) >> QS) + 1;     // <= 2^SH + 1
        L29 r; int64_t c = 0;
        _Pragma("unroll") for (int j = 0; j < NL; j++) {
            const int64_t t = (int64_t)x.l[j] - (int64_t)k1 * pl(j) + c;
            r.l[j] = j < NL - 1 ? (int32_t)((uint32_t)t & MASK) : (int32_t)t;
            c = t >> W;
        }
        return r;
    }
    __device__ __forceinline__ L29 operator+(const L29& b) const { L29 r; _Pragma("unroll") for (int k = 0; k < NL; k++) r.l[k] = l[k] + b.l[k]; return r; }
    __device__ __forceinline__ L29 operator-(const L29& b) const { L29 r; _Pragma("unroll") for (int k = 0; k < NL; k++) r.l[k] = l[k] - b.l[k]; return r; }
    __device__ __forceinline__ L29 neg() const { L29 r; _Pragma("unroll") for (int k = 0; k < NL; k++) r.l[k] = -l[k]; return r; }
    __device__ __forceinline__ L29 dbl() const { L29 r; _Pragma("unroll") for (int k = 0; k < NL; k++) r.l[k] = l[k] * 2; return r; }
    // carry propagation: limbs 0..NL-2 into [0, 2^W), the top limb signed
    __device__ __forceinline__ L29 norm() const {
        L29 r; int32_t c = 0;
        _Pragma("unroll") for (int k = 0; k < NL - 1; k++) { int32_t t = l[k] + c; r.l[k] = t & (int32_t)MASK; c = t >> W; }
        r.l[NL - 1] = l[NL - 1] + c;
        return r;
    }
    // reduction rounds + result limbs of the Montgomery cores below (T holds the 2 NL column sums of the products)
    __device__ __forceinline__ static void reduce_round(int64_t (&T)[2 * NL], int i) {
        const int32_t m = (int32_t)(((uint32_t)T[i] * (P::INV & MASK)) & MASK);
        _Pragma("unroll") for (int j = 0; j < NL; j++) T[i + j] += (int64_t)m * pl(j);
        T[i + 1] += T[i] >> W;                           // exact: T[i] is a multiple of 2^W
    }
    __device__ __forceinline__ static L29 upper_half(int64_t (&T)[2 * NL]) {
        L29 r;
        _Pragma("unroll") for (int k = 0; k < NL - 1; k++) { r.l[k] = (int32_t)((uint32_t)T[NL + k] & MASK); T[NL + 1 + k] += T[NL + k] >> W; }
        r.l[NL - 1] = (int32_t)T[2 * NL - 1];
        return r;
    }
    __device__ __forceinline__ static L29 mul(const L29& a, const L29& b) {
        int64_t T[2 * NL];
        _Pragma("unroll") for (int k = 0; k < 2 * NL; k++) T[k] = 0;
        _Pragma("unroll") for (int i = 0; i < NL; i++) {
            _Pragma("unroll") for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a.l[i] * b.l[j];
            reduce_round(T, i);
        }
        return upper_half(T);
    }
    // The same product by COLUMNS (product scanning, one running accumulator): column k takes its a_i b_(k-i) and m_i p_(k-i) terms, gives
    // m_k (lower half) or limb k - NL (upper half), and moves on with ONE 64-bit shift — the row form above pays shift + 64-bit add per
    // reduction round and mask + shift + 64-bit add per result limb (34 more vector instructions per product).  Same value, same bounds
    // (a column holds the same <= 2 NL products + carry).  The multiply-adds of a column form a dependency chain: used where a lane has
    // two or more independent products in flight (the NTT butterflies).
    __device__ __forceinline__ static L29 mul_cols(const L29& a, const L29& b) {
        // (the empty asm statements pin the ORDER of the additions: left alone, the compiler sums a column's products first and adds the shifted
        // carry last — one more 64-bit addition per column, which is exactly what this form is there to save)
#define CG_CHAIN(x) asm("" : "+v"(x))
        int32_t m[NL]; L29 r; int64_t T = 0;
        _Pragma("unroll") for (int k = 0; k < NL; k++) {
            _Pragma("unroll") for (int i = 0; i <= k; i++) { T += (int64_t)a.l[i] * b.l[k - i]; CG_CHAIN(T); }
            _Pragma("unroll") for (int i = 0; i < k; i++) { T += (int64_t)m[i] * pl(k - i); CG_CHAIN(T); }
            m[k] = (int32_t)(((uint32_t)T * (P::INV & MASK)) & MASK);
            T += (int64_t)m[k] * pl(0);
            T >>= W;                                     // exact: the column sum is a multiple of 2^W
            CG_CHAIN(T);
        }
        _Pragma("unroll") for (int k = NL; k < 2 * NL - 1; k++) {
            _Pragma("unroll") for (int i = k - NL + 1; i < NL; i++) { T += (int64_t)a.l[i] * b.l[k - i]; CG_CHAIN(T); }
            _Pragma("unroll") for (int i = k - NL + 1; i < NL; i++) { T += (int64_t)m[i] * pl(k - i); CG_CHAIN(T); }
            r.l[k - NL] = (int32_t)((uint32_t)T & MASK);
            T >>= W;
            CG_CHAIN(T);
        }
#undef CG_CHAIN
        r.l[NL - 1] = (int32_t)T;
        return r;
    }
    // two independent products by columns, their chains interleaved statement by statement (a dependent v_mad_u64_u32 pair costs a wait state:
    // the second chain fills it)
    __device__ __forceinline__ static void mul2_cols(const L29& a, const L29& b, const L29& c, const L29& d, L29& ab, L29& cd) {
#define CG_CHAIN(x) asm("" : "+v"(x))
        int32_t m[NL], n[NL]; int64_t T = 0, U = 0;
        _Pragma("unroll") for (int k = 0; k < NL; k++) {
            _Pragma("unroll") for (int i = 0; i <= k; i++) { T += (int64_t)a.l[i] * b.l[k - i]; CG_CHAIN(T); U += (int64_t)c.l[i] * d.l[k - i]; CG_CHAIN(U); }
            _Pragma("unroll") for (int i = 0; i < k; i++) { T += (int64_t)m[i] * pl(k - i); CG_CHAIN(T); U += (int64_t)n[i] * pl(k - i); CG_CHAIN(U); }
            m[k] = (int32_t)(((uint32_t)T * (P::INV & MASK)) & MASK); n[k] = (int32_t)(((uint32_t)U * (P::INV & MASK)) & MASK);
            T += (int64_t)m[k] * pl(0); U += (int64_t)n[k] * pl(0);
            T >>= W; U >>= W;
            CG_CHAIN(T); CG_CHAIN(U);
        }
        _Pragma("unroll") for (int k = NL; k < 2 * NL - 1; k++) {
            _Pragma("unroll") for (int i = k - NL + 1; i < NL; i++) { T += (int64_t)a.l[i] * b.l[k - i]; CG_CHAIN(T); U += (int64_t)c.l[i] * d.l[k - i]; CG_CHAIN(U); }
            _Pragma("unroll") for (int i = k - NL + 1; i < NL; i++) { T += (int64_t)m[i] * pl(k - i); CG_CHAIN(T); U += (int64_t)n[i] * pl(k - i); CG_CHAIN(U); }
            ab.l[k - NL] = (int32_t)((uint32_t)T & MASK); cd.l[k - NL] = (int32_t)((uint32_t)U & MASK);
            T >>= W; U >>= W;
            CG_CHAIN(T); CG_CHAIN(U);
        }
#undef CG_CHAIN
        ab.l[NL - 1] = (int32_t)T; cd.l[NL - 1] = (int32_t)U;
    }
    // ---- column engine: several INDEPENDENT Montgomery products run column by column, their multiply-adds interleaved chain by chain
    // (a dependent v_mad_u64_u32 needs two independent instructions in front of it: two chains leave one wait state per pair, three none).
    // A chain is one running accumulator; its column k takes the chain's own products (`prod`), the reduction terms m_i p_(k-i) (`red`) and
    // yields m_k or result limb k - NL (`finish`).  The order of the additions is pinned (see mul_cols).  Same values and bounds as the row forms.
#define CG_CHAIN(x) asm("" : "+v"(x))
    // (column and slot indices are template parameters: every condition below folds at compile time, every p_j is an immediate)
    struct ColBase {
        int64_t T = 0; int32_t m[NL]; L29 r;
        template <int K, int I> __device__ __forceinline__ void red() { if constexpr (I < K && I < NL && K - I < NL) { T += (int64_t)m[I] * pl(K - I); CG_CHAIN(T); } }
        template <int K, int I> __device__ __forceinline__ void prod2() {}               // second product of a slot (fused chains only)
        template <int K> __device__ __forceinline__ void finish() {
            if constexpr (K < NL) { m[K] = (int32_t)(((uint32_t)T * (P::INV & MASK)) & MASK); T += (int64_t)m[K] * pl(0); T >>= W; CG_CHAIN(T); }
            else if constexpr (K < 2 * NL - 2) { r.l[K - NL] = (int32_t)((uint32_t)T & MASK); T >>= W; CG_CHAIN(T); }
            else { r.l[K - NL] = (int32_t)((uint32_t)T & MASK); r.l[NL - 1] = (int32_t)(T >> W); }
        }
    };
    struct ColMul : ColBase {          // a * b
        const L29& a; const L29& b;
        __device__ __forceinline__ ColMul(const L29& a_, const L29& b_) : a(a_), b(b_) {}
        template <int K, int I> __device__ __forceinline__ void prod() { if constexpr (K - I >= 0 && K - I < NL) { this->T += (int64_t)a.l[I] * b.l[K - I]; CG_CHAIN(this->T); } }
    };
    struct ColSqr : ColBase {          // a * a, off-diagonal products once, doubled
        const L29& a; int32_t a2[NL];
        __device__ __forceinline__ ColSqr(const L29& a_) : a(a_) { _Pragma("unroll") for (int k = 0; k < NL; k++) a2[k] = a_.l[k] * 2; }
        template <int K, int I> __device__ __forceinline__ void prod() {
            if constexpr (K - I >= 0 && K - I < NL && I <= K - I) { this->T += I == K - I ? (int64_t)a.l[I] * a.l[I] : (int64_t)a2[I] * a.l[K - I]; CG_CHAIN(this->T); }
        }
    };
    struct ColMulSub : ColBase {       // a * b - c * d in ONE reduction
        const L29& a; const L29& b; const L29& c; const L29& d;
        __device__ __forceinline__ ColMulSub(const L29& a_, const L29& b_, const L29& c_, const L29& d_) : a(a_), b(b_), c(c_), d(d_) {}
        template <int K, int I> __device__ __forceinline__ void prod() {
            if constexpr (K - I >= 0 && K - I < NL) { this->T += (int64_t)a.l[I] * b.l[K - I]; CG_CHAIN(this->T); }
        }
        template <int K, int I> __device__ __forceinline__ void prod2() {
            if constexpr (K - I >= 0 && K - I < NL) { this->T += (int64_t)(-c.l[I]) * d.l[K - I]; CG_CHAIN(this->T); }
        }
    };
    struct ColMulAdd : ColBase {       // a * b + c * d in ONE reduction
        const L29& a; const L29& b; const L29& c; const L29& d;
        __device__ __forceinline__ ColMulAdd(const L29& a_, const L29& b_, const L29& c_, const L29& d_) : a(a_), b(b_), c(c_), d(d_) {}
        template <int K, int I> __device__ __forceinline__ void prod() {
            if constexpr (K - I >= 0 && K - I < NL) { this->T += (int64_t)a.l[I] * b.l[K - I]; CG_CHAIN(this->T); }
        }
        template <int K, int I> __device__ __forceinline__ void prod2() {
            if constexpr (K - I >= 0 && K - I < NL) { this->T += (int64_t)c.l[I] * d.l[K - I]; CG_CHAIN(this->T); }
        }
    };
    template <int K, int I, class... C> __device__ __forceinline__ static void col_prods(C&... c) { (c.template prod<K, I>(), ...); (c.template prod2<K, I>(), ...); if constexpr (I + 1 < NL) col_prods<K, I + 1>(c...); }
    template <int K, int I, class... C> __device__ __forceinline__ static void col_reds(C&... c) { (c.template red<K, I>(), ...); if constexpr (I + 1 < NL) col_reds<K, I + 1>(c...); }
    template <int K, class... C> __device__ __forceinline__ static void col_step(C&... c) {
        col_prods<K, 0>(c...); col_reds<K, 0>(c...); (c.template finish<K>(), ...);
        if constexpr (K + 1 < 2 * NL - 1) col_step<K + 1>(c...);
    }
    template <class... C> __device__ __forceinline__ static void run_cols(C&... c) { col_step<0>(c...); }
#undef CG_CHAIN
    // cheap necessary condition for x = 0 (mod p) on an UNNORMALISED value: carries only travel upwards, so the lowest W bits of
    // limb 0 already are the normalised limb 0 and must equal limb 0 of one of the candidates k*p (a non-zero residue passes with
    // probability ~10 / 2^W).  Lets the hot loop skip the carry propagation it would otherwise do only for this test.
    __device__ __forceinline__ static bool maybe_zero_mod_p(const L29& x) {
        // x = k p with -3 <= k <= 6  =>  k = x p^-1 (mod 2^W) lies in that range; p^-1 = -INV (mod 2^W)
        const uint32_t k = ((uint32_t)x.l[0] * (0u - P::INV)) & MASK;
        return ((k + 3u) & MASK) < 10u;
    }
    // k*p in normalised limbs, k in [-3, 6]: the residues a normalised value in (-4p, 7p) takes when it is 0 mod p
    __device__ __forceinline__ static bool is_zero_mod_p(const L29& x /* normalised */) {
        // cheap filter on the lowest limb (a non-zero residue matches one of the candidates with probability ~10 / 2^W)
        bool maybe = false;
        _Pragma("unroll") for (int kk = -3; kk <= 6; kk++) maybe = maybe || (x.l[0] == (int32_t)(((int64_t)kk * pl(0)) & MASK));
        if (!maybe) return false;
        bool any = false;
        _Pragma("unroll") for (int kk = -3; kk <= 6; kk++) {
            // limbs of kk*p: carry-normalise kk * pl(j) on the fly (compile-time constants after unrolling)
            bool eq = true; int64_t c = 0;
            _Pragma("unroll") for (int j = 0; j < NL; j++) {
                int64_t t = (int64_t)kk * pl(j) + c;
                int32_t limb = j < NL - 1 ? (int32_t)(t & MASK) : (int32_t)t;
                c = t >> W;
                eq = eq && (x.l[j] == limb);
            }
            any = any || eq;
        }
        return any;
    }
    // (a*b - c*d + M p) / 2^(NL W) with ONE reduction: 2 NL signed products + NL reduction products per column
    __device__ __forceinline__ static L29 mul_sub(const L29& a, const L29& b, const L29& c, const L29& d) {
        int64_t T[2 * NL];
        _Pragma("unroll") for (int k = 0; k < 2 * NL; k++) T[k] = 0;
        _Pragma("unroll") for (int i = 0; i < NL; i++) {
            _Pragma("unroll") for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a.l[i] * b.l[j];
            _Pragma("unroll") for (int j = 0; j < NL; j++) T[i + j] += (int64_t)(-c.l[i]) * d.l[j];
            reduce_round(T, i);
        }
        return upper_half(T);
    }
    // squaring: the off-diagonal products are taken once, doubled (NL (NL + 1) / 2 instead of NL^2 multiplies before the reduction).
    // Column i is complete before round i uses it: a pair (x, y), x <= y, x + y = i is added in round x <= i/2.
    __device__ __forceinline__ static L29 sqr(const L29& a) {
        int64_t T[2 * NL];
        _Pragma("unroll") for (int k = 0; k < 2 * NL; k++) T[k] = 0;
        int32_t a2[NL];
        _Pragma("unroll") for (int k = 0; k < NL; k++) a2[k] = a.l[k] * 2;
        _Pragma("unroll") for (int i = 0; i < NL; i++) {
            T[2 * i] += (int64_t)a.l[i] * a.l[i];
            _Pragma("unroll") for (int j = i + 1; j < NL; j++) T[i + j] += (int64_t)a2[i] * a.l[j];
            reduce_round(T, i);
        }
        return upper_half(T);
    }
    __device__ __forceinline__ static XYZZ<F> dbl_affine(const F& x, const F& y) { return xyzz_dbl_affine(x, y); }
    // small representative (0.5p .. 1.6p) of 1 in the 2^(NL W) domain: (2^SH R1)^2 / 2^(NL W) = 2^(NL W) (mod p), R1 = 2^(32N) mod p
    __device__ __forceinline__ static L29 one() { const L29 o = unpack<SH>(F::one()); return mul(o, o); }
    // the same residue in [0, p) without a product (constant-folded: F::one() is a compile-time constant)
    __device__ __forceinline__ static L29 one_small() {
        L29 o = unpack_small(F::one());
        L29 pp; _Pragma("unroll") for (int k = 0; k < NL; k++) pp.l[k] = pl(k);
        const L29 t = (o + pp).norm();
        if (o.l[NL - 1] < 0) o = t;
        return o;
    }
    // back to a canonical field element in the ABI's 2^(32N) Montgomery domain; |value| < 8p
    __device__ __forceinline__ static F to_fp(const L29& x) {
        L29 c; _Pragma("unroll") for (int k = 0; k < NL; k++) c.l[k] = 0;
        c.l[(32 * F::N) / W] = 1 << ((32 * F::N) % W);     // 2^(32N) as an integer: x * 2^(32N) / 2^(NL W) = x / 2^SH
        return pack_reduced(mul(x, c));                    // in (-0.3p, 1.3p), normalised
    }
    // normalised value in (-p, 2p) -> the canonical element with the same residue, packed into N x 32 bits
    __device__ __forceinline__ static F pack_reduced(L29 y) {
        // add p if negative, subtract p if >= p (sign of the top limb after normalisation decides)
        L29 pp; _Pragma("unroll") for (int k = 0; k < NL; k++) pp.l[k] = pl(k);
        L29 t = (y + pp).norm();
        if (y.l[NL - 1] < 0) y = t;
        t = (y - pp).norm();
        if (t.l[NL - 1] >= 0) y = t;
        F r;
        _Pragma("unroll") for (int w = 0; w < F::N; w++) {
            const int k = (32 * w) / W, sh = 32 * w - W * k;        // up to three limbs cover a word (W = 28, sh = 27: 1 + 28 + 3 bits)
            uint64_t v = (uint64_t)(uint32_t)y.l[k] >> sh;
            if (k + 1 < NL) v |= (uint64_t)(uint32_t)y.l[k + 1] << (W - sh);
            if (k + 2 < NL && 2 * W - sh < 32) v |= (uint64_t)(uint32_t)y.l[k + 2] << (2 * W - sh);
            r.v[w] = (uint32_t)v;
        }
        return r;
    }
};

}  // namespace cg
