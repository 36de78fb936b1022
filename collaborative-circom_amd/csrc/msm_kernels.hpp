// Variable-base multi-scalar multiplication (Pippenger bucket method) for gfx950, G1 and G2.
// Replaces `C::msm_unchecked(points, scalars)` of ark-ec 0.4.2 as called from
// `/root/reference/mpc-core/src/protocols/rep3.rs:942-943` (shamir.rs:1035, plain.rs:414).  The result is the same group
// element; which window size / bucket order is used cannot change it.
//
// Pipeline (all on one HIP stream, no host round trip until the per-window sums come back):
//   1. k_msm_digits      scalars: Montgomery -> canonical, signed c-bit digits (buckets 1..2^(c-1)), per-(window,bucket)
//                        histogram with global atomics.                                   [reads 32 B/scalar]
//   2. k_scan_exclusive  bucket offsets.
//   3. k_msm_scatter     counting sort of point indices by (window, bucket).
//   4. k_msm_accumulate  one lane per CHUNK of 128 consecutive sorted entries (balanced work per lane whatever the bucket sizes):
//                        gather the points (64/128 B each) and fold them with XYZZ mixed additions (8M+2S); pieces of a bucket
//                        that straddle chunks are merged by k_msm_merge_cont.  This is where the time goes: integer VALU bound.
//   5. k_msm_reduce_segments / k_msm_window_sum   running-sum bucket reduction, split into 2^15/L independent segments per
//                        window (segment result = sum (b-lo+1) B_b + lo * sum B_b), then a per-window tree sum in LDS.
//   6. host: Horner fold of the <= 64 window sums (c doublings each) — O(1) work, kept on the host.
// With uniformly random scalars (REP3 shares always are) every bucket receives n/2^(c-1) +- sqrt points: lanes are balanced.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "subgroup.hpp"
#include "curve.hpp"
#include "lazy29.hpp"
#include "vec_kernels.hpp"
#include "msm_sort_kernels.hpp"

namespace cg {

template <class A>
__device__ __forceinline__ A ld_struct(const A* p) {
    static_assert(sizeof(A) % 16 == 0, "");
    A r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4* d = reinterpret_cast<uint4*>(&r);
    _Pragma("unroll") for (int i = 0; i < (int)(sizeof(A) / 16); i++) d[i] = q[i];
    return r;
}
template <class A>
__device__ __forceinline__ void st_struct(A* p, const A& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const uint4* d = reinterpret_cast<const uint4*>(&r);
    _Pragma("unroll") for (int i = 0; i < (int)(sizeof(A) / 16); i++) q[i] = d[i];
}

// Accumulator storage policy of the bucket-accumulation kernel.
//  * RegAcc: the XYZZ accumulator lives in VGPRs (G1: 32 dwords, 3 waves/SIMD).
//  * LdsAcc: the accumulator lives in LDS, one 16-byte column per lane (G2: 64..96 dwords per lane would otherwise push the
//    kernel to 1 wave/SIMD; in LDS it costs ~32 ds_read/ds_write_b128 per mixed addition, nothing next to ~13k VALU ops).
template <class F>
struct RegAcc {
    typedef uint4 LdsT;
    static constexpr bool USES_LDS = false;
    XYZZ<F> v;
    bool inf;
    __device__ __forceinline__ void init(uint4*, int, int) { inf = true; }
    __device__ __forceinline__ F get(int f) const { return f == 0 ? v.x : f == 1 ? v.y : f == 2 ? v.zz : v.zzz; }
    __device__ __forceinline__ void set(int f, const F& x) { if (f == 0) v.x = x; else if (f == 1) v.y = x; else if (f == 2) v.zz = x; else v.zzz = x; }
};
template <class F>
struct LdsAcc {
    typedef uint4 LdsT;
    static constexpr bool USES_LDS = true;
    static constexpr size_t LDS_BYTES_PER_LANE = sizeof(XYZZ<F>);
    static constexpr int Q = sizeof(F) / 16;          // 16-byte quads per coordinate
    uint4* base; int stride;                           // quad e of this lane at base[e * stride]
    bool inf;
    __device__ __forceinline__ void init(uint4* lds, int tid, int nthreads) { base = lds + tid; stride = nthreads; inf = true; }
    __device__ __forceinline__ F get(int f) const {
        F r; uint4* d = reinterpret_cast<uint4*>(&r);
        _Pragma("unroll") for (int i = 0; i < Q; i++) d[i] = base[(f * Q + i) * stride];
        return r;
    }
    __device__ __forceinline__ void set(int f, const F& x) {
        const uint4* d = reinterpret_cast<const uint4*>(&x);
        _Pragma("unroll") for (int i = 0; i < Q; i++) base[(f * Q + i) * stride] = d[i];
    }
};

// acc += (x2, y2), formulas of xyzz_madd scheduled so that each accumulator coordinate is fetched right before its use
template <class F, class Acc>
__device__ __forceinline__ void acc_madd(Acc& acc, const F& x2, const F& y2) {
    if (acc.inf) { acc.set(0, x2); acc.set(1, y2); acc.set(2, F::one()); acc.set(3, F::one()); acc.inf = false; return; }
    F P = x2 * acc.get(2) - acc.get(0);
    F R = y2 * acc.get(3) - acc.get(1);
    if (P.is_zero()) {
        if (R.is_zero()) { XYZZ<F> d = xyzz_dbl_affine(x2, y2); acc.inf = d.is_inf(); acc.set(0, d.x); acc.set(1, d.y); acc.set(2, d.zz); acc.set(3, d.zzz); }
        else acc.inf = true;
        return;
    }
    F PP = P.sqr();
    acc.set(2, acc.get(2) * PP);
    F PPP = P * PP;
    acc.set(3, acc.get(3) * PPP);
    F Q = acc.get(0) * PP;
    F X3 = R.sqr() - PPP - Q.dbl();
    acc.set(0, X3);
    acc.set(1, R * (Q - X3) - acc.get(1) * PPP);
}
template <class F, class Acc>
__device__ __forceinline__ void acc_flush(Acc& acc, XYZZ<F>* dst) {
    XYZZ<F> r = acc.inf ? XYZZ<F>::infinity() : XYZZ<F>{acc.get(0), acc.get(1), acc.get(2), acc.get(3)};
    st_struct(dst, r);
    acc.inf = true;
}

// Fp2 = Fp[u]/(u^2+1) on lazy limbs.  Products are accumulated FUSED: c0 = a0 b0 - a1 b1 and c1 = a0 b1 + a1 b0 are each one
// column accumulation (18 signed products + 9 reduction products per column: < 27 * 2^58 < 2^63) followed by ONE reduction,
// i.e. the same 486 multiplies as Karatsuba but no pre/post additions and 2 instead of 3 reductions.
// Bounds (d = 0.45, units of p, per component): products in (-d, 1+d); X in (-4.8, 2.8); Y in (-1.9, 1.9); P in (-3.3, 6.3).
#ifndef CG_L29X2_INLINE
#define CG_L29X2_INLINE __forceinline__
#endif
template <class F2>
struct L29x2 {
    typedef typename F2::Base F;
    typedef L29<F> L;
    L c0, c1;
    template <int SHIFT> __device__ __forceinline__ static L29x2 unpack(const F2& a) { return {L::template unpack<SHIFT>(a.c0), L::template unpack<SHIFT>(a.c1)}; }
    __device__ __forceinline__ static L29x2 unpack_small(const F2& a) { return {L::unpack_small(a.c0), L::unpack_small(a.c1)}; }
    __device__ __forceinline__ static L29x2 one_small() { L z; _Pragma("unroll") for (int k = 0; k < L::NL; k++) z.l[k] = 0; return {L::one_small(), z}; }
    __device__ __forceinline__ L29x2 operator+(const L29x2& b) const { return {c0 + b.c0, c1 + b.c1}; }
    __device__ __forceinline__ L29x2 operator-(const L29x2& b) const { return {c0 - b.c0, c1 - b.c1}; }
    __device__ __forceinline__ L29x2 neg() const { return {c0.neg(), c1.neg()}; }
    __device__ __forceinline__ L29x2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    __device__ __forceinline__ L29x2 norm() const { return {c0.norm(), c1.norm()}; }
    // r = (a*b + s*c*d + M p) / 2^261, s = +1 or -1: the fused two-product Montgomery step
    template <int SIGN>
    __device__ __forceinline__ static L mul2(const L& a, const L& b, const L& c, const L& d) {
        constexpr int NL = L::NL;
        int64_t T[2 * NL];
        _Pragma("unroll") for (int k = 0; k < 2 * NL; k++) T[k] = 0;
        _Pragma("unroll") for (int i = 0; i < NL; i++) {
            _Pragma("unroll") for (int j = 0; j < NL; j++) T[i + j] += (int64_t)a.l[i] * b.l[j];
            _Pragma("unroll") for (int j = 0; j < NL; j++) T[i + j] += (int64_t)(SIGN > 0 ? c.l[i] : -c.l[i]) * d.l[j];
            L::reduce_round(T, i);
        }
        return L::upper_half(T);
    }
    static __device__ CG_L29X2_INLINE L29x2 mul(const L29x2& a, const L29x2& b) {
        return {mul2<-1>(a.c0, b.c0, a.c1, b.c1), mul2<+1>(a.c0, b.c1, a.c1, b.c0)};
    }
    static __device__ CG_L29X2_INLINE L29x2 sqr(const L29x2& a) {
        L s = (a.c0 + a.c1).norm(), d = (a.c0 - a.c1).norm();     // limb magnitude back to 2^29 before multiplying
        return {L::mul(s, d), L::mul(a.c0.dbl(), a.c1)};
    }
    // column-engine chains (L29::run_cols): an Fq2 product = two fused chains, a squaring = two plain ones
    struct Col2Mul {
        typename L::ColMulSub c0; typename L::ColMulAdd c1;
        __device__ __forceinline__ Col2Mul(const L29x2& a, const L29x2& b) : c0(a.c0, b.c0, a.c1, b.c1), c1(a.c0, b.c1, a.c1, b.c0) {}
        __device__ __forceinline__ L29x2 res() const { return {c0.r, c1.r}; }
    };
    struct Col2Sqr {
        L s, d, a0d; const L& a1; typename L::ColMul c0, c1;
        __device__ __forceinline__ Col2Sqr(const L29x2& a) : s((a.c0 + a.c1).norm()), d((a.c0 - a.c1).norm()), a0d(a.c0.dbl()), a1(a.c1), c0(s, d), c1(a0d, a1) {}
        __device__ __forceinline__ L29x2 res() const { return {c0.r, c1.r}; }
    };
    // a*b - c*d: kept as two products here (four signed products per column would not fit 63 bits)
    __device__ __forceinline__ static L29x2 mul_sub(const L29x2& a, const L29x2& b, const L29x2& c, const L29x2& d) { return (mul(a, b) - mul(c, d)).norm(); }
    __device__ __forceinline__ static bool is_zero_mod_p(const L29x2& x) { return L::is_zero_mod_p(x.c0) && L::is_zero_mod_p(x.c1); }
    __device__ __forceinline__ static bool maybe_zero_mod_p(const L29x2& x) { return L::maybe_zero_mod_p(x.c0) && L::maybe_zero_mod_p(x.c1); }
    __device__ __forceinline__ static L29x2 one() { L z; _Pragma("unroll") for (int k = 0; k < L::NL; k++) z.l[k] = 0; return {L::one(), z}; }
    __device__ __forceinline__ static F2 to_fp(const L29x2& x) { return {L::to_fp(x.c0), L::to_fp(x.c1)}; }
    __device__ __forceinline__ static XYZZ<F2> dbl_affine(const F2& x, const F2& y) { return xyzz_dbl_affine(x, y); }
};

template <class F> struct LazyOf { typedef L29<F> type; };
template <class B> struct LazyOf<Fp2<B>> { typedef L29x2<Fp2<B>> type; };
template <class L> struct LazyShift { static constexpr int value = L::SH; };                 // bits between the ABI's and the lazy Montgomery domain
template <class F2> struct LazyShift<L29x2<F2>> { static constexpr int value = L29x2<F2>::L::SH; };

// lazy accumulator in VGPRs (G1) or LDS (G2: 72 dwords per lane)
template <class F>
struct RegAcc29 {
    typedef uint32_t LdsT;
    typedef typename LazyOf<F>::type L;
    static constexpr bool USES_LDS = false;
    L c[4];
    bool inf;
    __device__ __forceinline__ void init(uint32_t*, int, int) { inf = true; }
    __device__ __forceinline__ L get(int f) const { return c[f]; }
    __device__ __forceinline__ void set(int f, const L& x) { c[f] = x; }
};
template <class F>
struct LdsAcc29 {
    typedef uint32_t LdsT;
    typedef typename LazyOf<F>::type L;
    static constexpr bool USES_LDS = true;
    static constexpr size_t LDS_BYTES_PER_LANE = 4 * sizeof(L);
    static constexpr int W = sizeof(L) / 4;               // dwords per coordinate
    uint32_t* base; int stride; bool inf;
    __device__ __forceinline__ void init(uint32_t* lds, int tid, int nthreads) { base = lds + tid; stride = nthreads; inf = true; }
    __device__ __forceinline__ L get(int f) const {
        L r; uint32_t* d = reinterpret_cast<uint32_t*>(&r);
        _Pragma("unroll") for (int i = 0; i < W; i++) d[i] = base[(f * W + i) * stride];
        return r;
    }
    __device__ __forceinline__ void set(int f, const L& x) {
        const uint32_t* d = reinterpret_cast<const uint32_t*>(&x);
        _Pragma("unroll") for (int i = 0; i < W; i++) base[(f * W + i) * stride] = d[i];
    }
};
template <class Acc> struct IsLazyAcc { static constexpr bool value = false; };
template <class F> struct IsLazyAcc<RegAcc29<F>> { static constexpr bool value = true; };
template <class F> struct IsLazyAcc<LdsAcc29<F>> { static constexpr bool value = true; };

// Buckets, continuation pieces and reduction partials of the lazy pipelines stay in LIMB form (XYZZL: 4 lazy coordinates;
// all limbs of ZZ zero = infinity, which a finite point can never show because its ZZ is non-zero modulo p), so nothing is
// converted between the accumulation and the final per-window sums.  Saturated XYZZ<F> is kept for fields without a lazy form.
template <class F>
struct alignas(16) XYZZL {
    typedef typename LazyOf<F>::type L;
    L c[4];   // X, Y, ZZ, ZZZ
};
template <class F> struct BucketOf { typedef XYZZ<F> type; };
template <class P> struct BucketOf<Fp<P>> { typedef XYZZL<Fp<P>> type; };                  // every field has a lazy form (9 x 29 or 14 x 28 limbs)
template <class B> struct BucketOf<Fp2<B>> { typedef XYZZL<Fp2<B>> type; };

template <class F> __device__ __forceinline__ bool bk_is_inf(const XYZZ<F>& p) { return p.is_inf(); }
template <class F> __device__ __forceinline__ bool bk_is_inf(const XYZZL<F>& p) {
    const int32_t* w = reinterpret_cast<const int32_t*>(&p.c[2]); int32_t o = 0;
    _Pragma("unroll") for (int i = 0; i < (int)(sizeof(p.c[2]) / 4); i++) o |= w[i];
    return o == 0;
}
template <class B> __device__ __forceinline__ B bk_inf() { B r; uint32_t* w = reinterpret_cast<uint32_t*>(&r); for (int i = 0; i < (int)(sizeof(B) / 4); i++) w[i] = 0; return r; }

// add-2008-s / dbl-2008-s-1 on lazy limbs (same value classes as the mixed addition: X in (-4.8, 2.8), Y in (-1.9, 1.9),
// ZZ, ZZZ in (-0.45, 1.45) times p; every product below stays under 72 p^2).  Out of line on purpose (latency-bound kernels).
template <class F>
__device__ __forceinline__ XYZZL<F> bk_dbl_inl(const XYZZL<F>& a) {
    typedef typename LazyOf<F>::type L;
    if (bk_is_inf(a) || L::is_zero_mod_p(a.c[1].norm())) return bk_inf<XYZZL<F>>();
    L U = a.c[1].dbl().norm(), V = L::sqr(U), W = L::mul(U, V), S = L::mul(a.c[0], V);
    L xx = L::sqr(a.c[0]);
    L M = (xx.dbl() + xx).norm();
    L X3 = (L::sqr(M) - S.dbl()).norm();
    XYZZL<F> r;
    r.c[0] = X3;
    r.c[1] = L::mul_sub(M, S - X3, W, a.c[1]);
    r.c[2] = L::mul(V, a.c[2]);
    r.c[3] = L::mul(W, a.c[3]);
    return r;
}
template <class F> __device__ __attribute__((noinline)) XYZZL<F> bk_dbl(const XYZZL<F>& a) { return bk_dbl_inl(a); }
// bk_add_inl: the same addition inlined into its caller — INCLUDING its doubling branch (equal operands; never taken on random points): a call
// there takes its operand's address, which put the caller's running sum into scratch memory on EVERY addition (round 4: 304 B / 592 B of private
// segment per lane in the merge and reduction kernels, ~30 scratch accesses on the hot path of each addition).  Out of line, both operands and the result travel through scratch memory
// (3 x 144 / 288 bytes per call and lane): in the reduction kernels, whose lanes run short serial chains of additions at one wave
// per SIMD, those round trips are on the critical path.
// (Round 6 measured the column form of acc_madd_lazy_cols for this full addition as well — groups of three independent chains, no wait states: the
// merge / reduction kernels go from 138-149 to 176-179 VGPRs and get no faster: reduce stage of a 2^22 G1 set 0.427-0.431 -> 0.438-0.454 ms, BLS12-381
// 0.770 -> 0.87; profiles/r06_acc_cols_ab.txt.  Row form kept.)
template <class F>
__device__ __forceinline__ XYZZL<F> bk_add_inl(const XYZZL<F>& a, const XYZZL<F>& b) {
    typedef typename LazyOf<F>::type L;
    if (bk_is_inf(a)) return b;
    if (bk_is_inf(b)) return a;
    L U1 = L::mul(a.c[0], b.c[2]), S1 = L::mul(a.c[1], b.c[3]);
    L P = L::mul(b.c[0], a.c[2]) - U1, R = L::mul(b.c[1], a.c[3]) - S1;
    if (L::is_zero_mod_p(P.norm())) {
        if (L::is_zero_mod_p(R.norm())) return bk_dbl_inl(a);
        return bk_inf<XYZZL<F>>();
    }
    L PP = L::sqr(P), PPP = L::mul(P, PP), Q = L::mul(U1, PP);
    L X3 = (L::sqr(R) - PPP - Q.dbl()).norm();
    XYZZL<F> r;
    r.c[0] = X3;
    r.c[1] = L::mul_sub(R, Q - X3, S1, PPP);
    r.c[2] = L::mul(L::mul(a.c[2], b.c[2]), PP);
    r.c[3] = L::mul(L::mul(a.c[3], b.c[3]), PPP);
    return r;
}
template <class F> __device__ __forceinline__ XYZZ<F> bk_add_inl(const XYZZ<F>& a, const XYZZ<F>& b) { return xyzz_add(a, b); }
template <class F>
__device__ __attribute__((noinline)) XYZZL<F> bk_add(const XYZZL<F>& a, const XYZZL<F>& b) {
    typedef typename LazyOf<F>::type L;
    if (bk_is_inf(a)) return b;
    if (bk_is_inf(b)) return a;
    L U1 = L::mul(a.c[0], b.c[2]), S1 = L::mul(a.c[1], b.c[3]);
    L P = L::mul(b.c[0], a.c[2]) - U1, R = L::mul(b.c[1], a.c[3]) - S1;
    if (L::is_zero_mod_p(P.norm())) {
        if (L::is_zero_mod_p(R.norm())) return bk_dbl(a);
        return bk_inf<XYZZL<F>>();
    }
    L PP = L::sqr(P), PPP = L::mul(P, PP), Q = L::mul(U1, PP);
    L X3 = (L::sqr(R) - PPP - Q.dbl()).norm();
    XYZZL<F> r;
    r.c[0] = X3;
    r.c[1] = L::mul_sub(R, Q - X3, S1, PPP);
    r.c[2] = L::mul(L::mul(a.c[2], b.c[2]), PP);
    r.c[3] = L::mul(L::mul(a.c[3], b.c[3]), PPP);
    return r;
}
template <class F> __device__ __forceinline__ XYZZ<F> bk_add(const XYZZ<F>& a, const XYZZ<F>& b) { return xyzz_add(a, b); }
template <class F> __device__ __forceinline__ XYZZ<F> bk_dbl(const XYZZ<F>& a) { return xyzz_dbl(a); }
template <class F> __device__ __forceinline__ XYZZ<F> bk_to_xyzz(const XYZZ<F>& a) { return a; }
template <class F> __device__ __forceinline__ XYZZ<F> bk_to_xyzz(const XYZZL<F>& a) {
    typedef typename LazyOf<F>::type L;
    if (bk_is_inf(a)) return XYZZ<F>::infinity();
    return {L::to_fp(a.c[0]), L::to_fp(a.c[1]), L::to_fp(a.c[2]), L::to_fp(a.c[3])};
}

// The same addition with the products run by COLUMNS and interleaved (L29::run_cols; round 6): -34 vector instructions per product, 153 instead of
// 164 VGPRs; 2^22-point launch 4.04-4.26 -> 3.90-4.10 ms (BN254 G1), 12.04 -> 11.6-11.7 (G2), 8.98-9.10 -> 8.85-8.87 (BLS12-381 G1) on the boxes of the
// round (profiles/r06_acc_cols_ab.txt).  Rounds 2-3 had measured product scanning with inline-asm multiply-adds and found nothing: left alone, the
// compiler re-associates the column sums (the saved 64-bit additions come back), and inline-asm multiply-adds each drag a wait state behind them;
// here the products stay C++ and only the ORDER of the additions is pinned (empty asm statements).  The nine products of madd-2008-s
// fall into four groups of independent ones — (U2, S2), (P^2, R^2), (P PP, X PP, ZZ PP), (ZZZ PPP, R (Q - X3) - Y PPP).
template <class L> struct HasColEngine { static constexpr bool value = false; };
template <class F> struct HasColEngine<L29<F>> { static constexpr bool value = true; };
template <class F, class Acc, class L>
__device__ __forceinline__ void acc_madd_lazy_cols(Acc& acc, const F& x2f, const F& y2f, const L& x2, const L& y2, bool negate) {
    const L X = acc.get(0), Y = acc.get(1), ZZ = acc.get(2), ZZZ = acc.get(3);
    typename L::ColMul u2(x2, ZZ), s2(y2, ZZZ);
    L::run_cols(u2, s2);
    const L P = u2.r - X, R = s2.r - Y;
    if (L::maybe_zero_mod_p(P) && L::is_zero_mod_p(P.norm())) {   // same x: doubling or cancellation (rare)
        if (L::is_zero_mod_p(R.norm())) {
            XYZZ<F> d = L::dbl_affine(x2f, negate ? y2f.neg() : y2f);
            acc.inf = d.is_inf();
            if (!acc.inf) {
                const L one = L::one();
                constexpr int SH = LazyShift<L>::value;
                acc.set(0, L::mul(L::template unpack<SH>(d.x), one)); acc.set(1, L::mul(L::template unpack<SH>(d.y), one));
                acc.set(2, L::mul(L::template unpack<SH>(d.zz), one)); acc.set(3, L::mul(L::template unpack<SH>(d.zzz), one));
            }
        } else acc.inf = true;
        return;
    }
    typename L::ColSqr pp(P), rr(R);
    L::run_cols(pp, rr);
    typename L::ColMul ppp(P, pp.r), q(X, pp.r), zz(ZZ, pp.r);
    L::run_cols(ppp, q, zz);
    const L X3 = (rr.r - ppp.r - q.r.dbl()).norm();
    const L QX = q.r - X3;
    typename L::ColMul zzz(ZZZ, ppp.r); typename L::ColMulSub y3(R, QX, Y, ppp.r);
    L::run_cols(zzz, y3);
    acc.set(2, zz.r); acc.set(3, zzz.r); acc.set(1, y3.r); acc.set(0, X3);
}
// ... and in Fq2: every product is two fused chains (c0 = a0 b0 - a1 b1, c1 = a0 b1 + a1 b0), interleaved product by product
// (BN254 only: on BLS12-381's 14-limb Fq2 the kernel already runs at one wave per SIMD with 321 registers, and the two chains' state pushes
// the mixed addition into scratch — a 2^22-point launch 25.9 -> 75.5 ms, measured)
template <class F2> struct HasColEngine<L29x2<F2>> { static constexpr bool value = F2::Base::N == 8; };
template <class F, class Acc, class F2>
__device__ __forceinline__ void acc_madd_lazy_cols(Acc& acc, const F& x2f, const F& y2f, const L29x2<F2>& x2, const L29x2<F2>& y2, bool negate) {
    typedef L29x2<F2> L; typedef typename L::L L1;
    auto mul = [](const L& a, const L& b) { typename L::Col2Mul c(a, b); L1::run_cols(c.c0, c.c1); return c.res(); };      // one Fq2 product = two interleaved chains
    auto sqr = [](const L& a) { typename L::Col2Sqr c(a); L1::run_cols(c.c0, c.c1); return c.res(); };
    const L P = mul(x2, acc.get(2)) - acc.get(0);
    const L R = mul(y2, acc.get(3)) - acc.get(1);
    if (L::maybe_zero_mod_p(P) && L::is_zero_mod_p(P.norm())) {   // same x: doubling or cancellation (rare)
        if (L::is_zero_mod_p(R.norm())) {
            XYZZ<F> d = L::dbl_affine(x2f, negate ? y2f.neg() : y2f);
            acc.inf = d.is_inf();
            if (!acc.inf) {
                const L one = L::one();
                constexpr int SH = LazyShift<L>::value;
                acc.set(0, L::mul(L::template unpack<SH>(d.x), one)); acc.set(1, L::mul(L::template unpack<SH>(d.y), one));
                acc.set(2, L::mul(L::template unpack<SH>(d.zz), one)); acc.set(3, L::mul(L::template unpack<SH>(d.zzz), one));
            }
        } else acc.inf = true;
        return;
    }
    const L PP = sqr(P);
    const L PPP = mul(P, PP);
    const L Q = mul(acc.get(0), PP);
    acc.set(2, mul(acc.get(2), PP));
    acc.set(3, mul(acc.get(3), PPP));
    const L X3 = (sqr(R) - PPP - Q.dbl()).norm();
    acc.set(1, (mul(R, Q - X3) - mul(acc.get(1), PPP)).norm());
    acc.set(0, X3);
}
// acc += (x2, y2): madd-2008-s on lazy signed limbs (see the bounds above); coordinates 0..3 = X, Y, ZZ, ZZZ
template <class F, class Acc>
__device__ __forceinline__ void acc_madd_lazy(Acc& acc, const F& x2f, const F& y2f, bool negate) {
    typedef typename LazyOf<F>::type L;
    L x2 = L::template unpack<LazyShift<L>::value>(x2f), y2 = L::template unpack<LazyShift<L>::value>(y2f);     // 2^SH * x2, 2^SH * y2: the 2^(NL W) domain
    if (negate) y2 = y2.neg();
    if (acc.inf) {
        // First point of a bucket: some lane of a wave is here on nearly every second iteration (64 lanes, ~100 entries per
        // bucket), and the whole wave waits for it, so this path must be short: the coordinates are brought into the small range
        // [-p, p) by subtracting a multiple of p (quotient estimated from the top limb) instead of multiplying by one.
        const L one = L::one_small();
        const L ys = L::unpack_small(y2f);
        acc.set(0, L::unpack_small(x2f)); acc.set(1, negate ? ys.neg().norm() : ys); acc.set(2, one); acc.set(3, one); acc.inf = false;
        return;
    }
#if !defined(CG_ACC_ROWS)                        // round 6 default: products by columns (-DCG_ACC_ROWS: the row form of rounds 2-5, for A/B runs)
    if constexpr (HasColEngine<L>::value) { acc_madd_lazy_cols<F>(acc, x2f, y2f, x2, y2, negate); return; }
#endif
    L P = L::mul(x2, acc.get(2)) - acc.get(0);
    L R = L::mul(y2, acc.get(3)) - acc.get(1);
    if (L::maybe_zero_mod_p(P) && L::is_zero_mod_p(P.norm())) {   // same x: doubling or cancellation (rare)
        if (L::is_zero_mod_p(R.norm())) {
            XYZZ<F> d = L::dbl_affine(x2f, negate ? y2f.neg() : y2f);
            acc.inf = d.is_inf();
            if (!acc.inf) {
                const L one = L::one();
                constexpr int SH = LazyShift<L>::value;
                acc.set(0, L::mul(L::template unpack<SH>(d.x), one)); acc.set(1, L::mul(L::template unpack<SH>(d.y), one));
                acc.set(2, L::mul(L::template unpack<SH>(d.zz), one)); acc.set(3, L::mul(L::template unpack<SH>(d.zzz), one));
            }
        } else acc.inf = true;
        return;
    }
    L PP = L::sqr(P);
    L PPP = L::mul(P, PP);
    L Q = L::mul(acc.get(0), PP);
    acc.set(2, L::mul(acc.get(2), PP));
    acc.set(3, L::mul(acc.get(3), PPP));
    L X3 = (L::sqr(R) - PPP - Q.dbl()).norm();
    acc.set(1, L::mul_sub(R, Q - X3, acc.get(1), PPP));        // one reduction for R*(Q - X3) - Y*PPP (result is normalised)
    acc.set(0, X3);
}
template <class F, class Acc>
__device__ __forceinline__ void acc_flush_lazy(Acc& acc, XYZZL<F>* dst) {
    XYZZL<F> r = bk_inf<XYZZL<F>>();
    if (!acc.inf) { r.c[0] = acc.get(0); r.c[1] = acc.get(1); r.c[2] = acc.get(2); r.c[3] = acc.get(3); }
    st_struct(dst, r);
    acc.inf = true;
}
// dispatch on the accumulator kind
template <class F, class Acc>
__device__ __forceinline__ void acc_madd(Acc& acc, const F& x2, const F& y2, bool negate) {
    if constexpr (IsLazyAcc<Acc>::value) acc_madd_lazy<F>(acc, x2, y2, negate);
    else acc_madd(acc, x2, negate ? y2.neg() : y2);
}
template <class F, class Acc, class B>
__device__ __forceinline__ void acc_store(Acc& acc, B* dst) {
    if constexpr (IsLazyAcc<Acc>::value) acc_flush_lazy<F>(acc, dst);
    else acc_flush(acc, dst);
}

// Bucket accumulation, chunk-balanced: lane q folds the L consecutive entries sorted[q*L, (q+1)*L) of the (window, bucket)-
// sorted index list, whatever buckets they belong to, so every lane of a wave does the same number of mixed additions
// (one lane per bucket wastes ~20 % of the wave on the Poisson spread of bucket sizes, and serialises skewed buckets).
//   * a bucket whose first entry lies in this chunk gets its partial sum written to buckets[b];
//   * the leading piece of the chunk that continues a bucket begun in an earlier chunk goes to cont[q] (tagged cont_bucket[q]);
// k_msm_merge_cont then adds the continuation pieces into their buckets (one lane per bucket run, no atomics).
template <class F, class Acc, int THREADS, int MINW = 1>
__global__ void __launch_bounds__(THREADS, MINW) k_msm_accumulate(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                            uint32_t nbuckets, uint32_t chunk_len, uint32_t nchunks, uint32_t table_stride, uint32_t cap,
                                                            typename BucketOf<F>::type* __restrict__ buckets, typename BucketOf<F>::type* __restrict__ cont,
                                                            uint32_t* __restrict__ cont_bucket, uint32_t may_have_inf) {
    // table_stride != 0: entries are (window << 24 | index) into per-window precomputed tables laid out [window][table_stride]
    // cap != 0: `sorted` is the padded layout of k_msm_scatter_direct (bucket b owns slots [b*cap, b*cap + min(count, cap)));
    //           positions (pos, offsets) are still those of the compact list, so the chunking is unchanged
    extern __shared__ uint4 acc_lds[];
    const uint32_t q = blockIdx.x * THREADS + threadIdx.x;
    if (q >= nchunks) return;
    const uint32_t capc = cap ? cap : 0xffffffffu;
    const uint32_t total = offsets[nbuckets - 1] + min(counts[nbuckets - 1], capc);
    uint32_t pos = q * chunk_len;
    if (pos >= total) { cont_bucket[q] = 0xffffffffu; return; }
    const uint32_t end = min(pos + chunk_len, total);
    // bucket containing entry `pos`: last b with offsets[b] <= pos (empty buckets share an offset with their successor; skip them)
    uint32_t lo = 0, hi = nbuckets - 1;
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (offsets[mid] <= pos) lo = mid; else hi = mid - 1; }
    uint32_t b = lo;
    uint32_t bend = offsets[b] + min(counts[b], capc);
    uint32_t kidx = pos - offsets[b];                       // index inside the bucket
    bool continuation = offsets[b] != pos;
    cont_bucket[q] = continuation ? b : 0xffffffffu;
    Acc acc;
    acc.init(reinterpret_cast<typename Acc::LdsT*>(acc_lds), threadIdx.x, THREADS);
    while (pos < end) {
        if (pos == bend) {                                   // finished bucket b inside this chunk
            if (continuation) { acc_store<F>(acc, cont + q); continuation = false; } else acc_store<F>(acc, buckets + b);
            do { b++; } while (counts[b] == 0);
            bend = offsets[b] + min(counts[b], capc);
            kidx = 0;
        }
        const uint32_t e = sorted[cap ? (size_t)b * cap + kidx : (size_t)pos];
        pos++; kidx++;
        const size_t at = table_stride ? (size_t)((e >> 24) & 0x7fu) * table_stride + (e & 0xffffffu) : (size_t)(e & 0x7fffffffu);
        Affine<F> p = ld_struct(bases + at);
        if (may_have_inf && p.is_inf()) continue;               // tables without a point at infinity (registration census) skip the test
        acc_madd(acc, p.x, p.y, (e >> 31) != 0);
    }
    if (continuation) acc_store<F>(acc, cont + q); else acc_store<F>(acc, buckets + b);
}

// Software-pipelined form of the same kernel for the compact list (cap == 0), written around where a wave of the kernel above
// spends the cycles it does not issue in (SQ_WAIT_ANY 19 % of wave time at 3 waves per SIMD): three dependent memory round trips
// per entry — sorted[pos] (L2), then the 64/128-byte gather bases[...] (HBM, random), and on nearly every second iteration of a
// WAVE (some lane of 64 crosses a bucket boundary) counts[b] / offsets[b] of the next bucket.  Here
//   * the sorted entries are read two iterations ahead (e0 current, e1 next, e2 in flight),
//   * the end of the NEXT bucket (offsets[b + 2]) is read when a bucket is entered, so a boundary costs no round trip,
//   * PF == 1: the record of the next entry is pulled into L2 while the current addition runs: one global_load_lds_dword per
//     64 bytes into a junk LDS slot (no VGPR destination, nothing to wait for), issued after the current record has arrived;
//   * PF == 2: the next record itself is loaded into registers before the current addition (for a 2-wave-per-SIMD build).
// Up to ACC_MAX_SETS accumulations of ONE launch geometry in one launch (blockIdx.y = set): the tables and share components of a small
// MSM call (2^16 constraints: a launch of 256 workgroups leaves three quarters of the chip idle and lasts as long as one lane's chain
// of additions, so eight launches in a row cost eight such chains; side by side they cost one).  Large calls pass one set.
template <class F>
struct AccSets {
    const Affine<F>* bases[ACC_MAX_SETS];                  // window-0 table (+ the caller's offset) of each set
    const uint32_t* sorted[ACC_MAX_SETS]; const uint32_t* offsets[ACC_MAX_SETS]; const uint32_t* counts[ACC_MAX_SETS];   // its schedule
    typename BucketOf<F>::type* buckets[ACC_MAX_SETS]; typename BucketOf<F>::type* cont[ACC_MAX_SETS]; uint32_t* cont_bucket[ACC_MAX_SETS];
    uint32_t table_stride[ACC_MAX_SETS]; uint32_t may_have_inf[ACC_MAX_SETS];
};
template <class Acc, int THREADS> __host__ __device__ constexpr size_t acc_lds_bytes() {
    if constexpr (Acc::USES_LDS) return (size_t)THREADS * Acc::LDS_BYTES_PER_LANE; else return 0;
}
template <class F, class Acc, int THREADS, int MINW, int PF, int DBG = 0>
__global__ void __launch_bounds__(THREADS, MINW) k_msm_accumulate_pf(const AccSets<F> S, uint32_t nbuckets, uint32_t chunk_len, uint32_t nchunks, uint32_t first_chunk) {
    const Affine<F>* __restrict__ bases = S.bases[blockIdx.y]; const uint32_t* __restrict__ sorted = S.sorted[blockIdx.y];
    const uint32_t* __restrict__ offsets = S.offsets[blockIdx.y]; const uint32_t* __restrict__ counts = S.counts[blockIdx.y];
    typename BucketOf<F>::type* __restrict__ buckets = S.buckets[blockIdx.y]; typename BucketOf<F>::type* __restrict__ cont = S.cont[blockIdx.y];
    uint32_t* __restrict__ cont_bucket = S.cont_bucket[blockIdx.y];
    const uint32_t table_stride = S.table_stride[blockIdx.y], may_have_inf = S.may_have_inf[blockIdx.y];
    extern __shared__ uint4 acc_lds[];        // [accumulators (LDS policies)] [PF == 1: THREADS junk dwords, see PF_JUNK_OFFSET]
    const uint32_t q = first_chunk + blockIdx.x * THREADS + threadIdx.x;   // first_chunk: a launch may cover a slice of the chunks (msm_accumulate_reduce)
    if (q >= nchunks) return;
    const uint32_t total = offsets[nbuckets - 1] + counts[nbuckets - 1];
    uint32_t pos = q * chunk_len;
    if (pos >= total) { cont_bucket[q] = 0xffffffffu; return; }
    const uint32_t end = min(pos + chunk_len, total);
    uint32_t lo = 0, hi = nbuckets - 1;
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (offsets[mid] <= pos) lo = mid; else hi = mid - 1; }
    uint32_t b = lo;
    auto end_of = [&](uint32_t bb) -> uint32_t { return bb + 1 < nbuckets ? offsets[bb + 1] : total; };   // exclusive scan: end of bucket bb
    uint32_t bend = end_of(b), bend_next = end_of(b + 1);
    bool continuation = offsets[b] != pos;
    cont_bucket[q] = continuation ? b : 0xffffffffu;
    Acc acc;
    acc.init(reinterpret_cast<typename Acc::LdsT*>(acc_lds), threadIdx.x, THREADS);
    // DBG (timing experiments only, results are wrong): 1 = cache-resident gather addresses, 2 = bucket boundaries ignored
    auto at_of = [&](uint32_t e) -> size_t {
        if constexpr (DBG & 1) return (size_t)(threadIdx.x + 256u * (e & 63u));
        if constexpr (DBG & 8) return (size_t)((e * 2654435761u) >> 12);                   // random over 2^20 records (64 MB: few pages, no cache reuse)
        return table_stride ? (size_t)((e >> 24) & 0x7fu) * table_stride + (e & 0xffffffu) : (size_t)(e & 0x7fffffffu);
    };
    auto entry = [&](uint32_t at) -> uint32_t { if constexpr (DBG & 4) return at * 2654435761u; else return sorted[at]; };   // DBG 4: no index loads
    uint32_t e0 = 0, e1 = 0;
    // PF == 3: the index list is read 16 bytes per lane every fourth iteration instead of 4 bytes every iteration:
    // a lane's entries are consecutive, but 64 lanes x 4 bytes touch 64 cache lines per load and each line comes back from L2 — or,
    // evicted by the gathers streaming through, from HBM — 32 times (1 GB of the 4.8 GB a 2^22 launch reads, r02_pmc_traffic.json)
    uint4 cur = make_uint4(0, 0, 0, 0);
    auto quad = [&](uint32_t at4) -> uint4 { return *reinterpret_cast<const uint4*>(sorted + at4); };    // reads past `end` stay inside the arena and are never used
    if constexpr (PF == 3) {
        cur = quad(pos & ~3u);
        for (uint32_t k = 0; k < (pos & 3u); k++) { cur.x = cur.y; cur.y = cur.z; cur.z = cur.w; }
    } else {
        e0 = entry(pos);
        e1 = pos + 1 < end ? entry(pos + 1) : e0;
    }
    Affine<F> pn;
    Affine<F> psyn;                                                                  // DBG 4: operands from registers, no gather
    if constexpr (DBG & 4) { uint32_t* w = reinterpret_cast<uint32_t*>(&psyn); for (int i = 0; i < (int)(sizeof(psyn) / 4); i++) w[i] = (threadIdx.x * 2654435761u + i * 40503u) & 0x0fffffffu; }
    if constexpr (PF == 2) pn = ld_struct(bases + at_of(e0));
    while (pos < end) {
        if (!(DBG & 2) && pos == bend) {                     // finished bucket b inside this chunk
            if (continuation) { acc_store<F>(acc, cont + q); continuation = false; } else acc_store<F>(acc, buckets + b);
            b++; bend = bend_next; bend_next = end_of(b + 1);
            while (bend == pos) { b++; bend = bend_next; bend_next = end_of(b + 1); }      // empty buckets (rare)
        }
        // load order matters: vmcnt counts in issue order, so the index read two iterations ahead is issued AFTER the record
        // loads — the wait for the current record then leaves it (and, PF == 2, the next record) in flight
        Affine<F> p;
        if constexpr (DBG & 4) { uint32_t* w = reinterpret_cast<uint32_t*>(&psyn); w[0] += 0x9e3779b9u; w[9] ^= w[0]; p = psyn; }
        else if constexpr (PF == 2) { p = pn; pn = ld_struct(bases + at_of(e1)); }
        else { if constexpr (PF == 3) e0 = cur.x; p = ld_struct(bases + at_of(e0)); }
        uint32_t e2 = 0;
        if constexpr (PF == 3) {
            cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
            if (((pos + 1) & 3u) == 0 && pos + 1 < end) cur = quad(pos + 1);   // issued behind the record loads: it arrives during this addition
        } else e2 = pos + 2 < end ? entry(pos + 2) : e1;
        if constexpr (PF == 1) {
            // after the last quad of p has arrived (the empty asm makes the address depend on it), so that the wait for p is not
            // extended to this load
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&p);
            uintptr_t a = reinterpret_cast<uintptr_t>(bases + at_of(e1));
            asm volatile("" : "+v"(a) : "v"(w[0]), "v"(w[sizeof(Affine<F>) / 4 - 1]), "v"(w[sizeof(Affine<F>) / 8]), "v"(w[sizeof(Affine<F>) / 8 - 1]));
            _Pragma("unroll") for (int k = 0; k < (int)(sizeof(Affine<F>) / 32); k++)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(a) + 8 * k,
                                                 (__attribute__((address_space(3))) uint32_t*)(reinterpret_cast<char*>(acc_lds) + acc_lds_bytes<Acc, THREADS>()), 4, 0, 0);
        }
        const bool neg = (e0 >> 31) != 0;
        pos++;
        if constexpr (PF != 3) { e0 = e1; e1 = e2; }
        if (may_have_inf && p.is_inf()) continue;
        acc_madd(acc, p.x, p.y, neg);
    }
    if (continuation) acc_store<F>(acc, cont + q); else acc_store<F>(acc, buckets + b);
}

// Continuation pieces of one bucket occupy consecutive chunks q0, q0+1, ... (same tag).  Two-level fold so that a bucket spread
// over thousands of chunks (skewed scalars, small top window) is not summed by a single lane:
//   level 1: every lane that is a GROUP head (run start, or q a multiple of MERGE_GROUP) sums the pieces up to the next group head;
//   level 2: the run-start lane adds the group sums of its run into buckets[b].
// With uniform scalars runs have 1-3 pieces and both levels are a handful of additions.
// Level 0, bucket-major: with a small shared bucket set (table slices of a multi-GPU plan: 2^16 buckets, ~230 entries each, chunks
// of ~57) nearly every chunk ends in a continuation piece and each bucket owns 3-5 of them; summing them chunk-major leaves three
// lanes in four idle.  Here lane b adds the pieces of bucket b, whose chunk range follows from offsets / counts alone, and retires
// them (tag -> none).  Buckets spread over more than MERGE_DIRECT_MAX chunks (skewed scalars) are left to the two levels below.
constexpr uint32_t MERGE_DIRECT_MAX = 12;
// The merge and reduction kernels below serve up to RED_MAX_SETS bucket sets of ONE launch geometry per launch (blockIdx.y = set): the
// bucket sets of the tables of one MSM call that share a coordinate field — l, a, b1 and both share components — whose accumulations
// have all finished.  Beside a lock-stepped accumulation a reduction costs the step its whole stand-alone DURATION (a chain of ~38
// dependent additions on a few hundred waves), whatever its width: six sets in one launch cost what one does.
template <class B>
struct RedSets {
    B* buckets[RED_MAX_SETS]; B* cont[RED_MAX_SETS]; uint32_t* cont_bucket[RED_MAX_SETS];
    const uint32_t* offsets[RED_MAX_SETS]; const uint32_t* counts[RED_MAX_SETS];     // the sorted schedule each set was accumulated from
    B* partials[RED_MAX_SETS]; void* wsums[RED_MAX_SETS];                               // reduction scratch, final sums (XYZZ<F>) of each set
};
template <class B>
__global__ void __launch_bounds__(64) k_msm_merge_direct(const RedSets<B> S, uint32_t nbuckets, uint32_t chunk_len, uint32_t nchunks, uint32_t cap) {
    B* __restrict__ buckets = S.buckets[blockIdx.y]; const B* __restrict__ cont = S.cont[blockIdx.y]; uint32_t* __restrict__ cont_bucket = S.cont_bucket[blockIdx.y];
    const uint32_t* __restrict__ offsets = S.offsets[blockIdx.y]; const uint32_t* __restrict__ counts = S.counts[blockIdx.y];
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    const uint32_t cnt = cap ? min(counts[b], cap) : counts[b];
    if (cnt == 0) return;
    const uint32_t first = offsets[b], last = first + cnt - 1;           // positions of the bucket's entries in the compact list
    const uint32_t q0 = first / chunk_len + 1, q1 = last / chunk_len;     // chunks that continue it
    if (q0 > q1 || q1 - q0 >= MERGE_DIRECT_MAX || q1 >= nchunks) return;
    B acc = ld_struct(buckets + b);
    for (uint32_t q = q0; q <= q1; q++) { acc = bk_add_inl(acc, ld_struct(cont + q)); cont_bucket[q] = 0xffffffffu; }
    st_struct(buckets + b, acc);
}
constexpr uint32_t MERGE_GROUP = 64;
template <class B>
__global__ void __launch_bounds__(64) k_msm_merge_cont_l1(const RedSets<B> S, uint32_t nchunks) {
    B* __restrict__ cont = S.cont[blockIdx.y]; const uint32_t* __restrict__ cont_bucket = S.cont_bucket[blockIdx.y];
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nchunks) return;
    const uint32_t b = cont_bucket[q];
    if (b == 0xffffffffu) return;
    const bool run_start = q == 0 || cont_bucket[q - 1] != b;
    if (!run_start && (q % MERGE_GROUP) != 0) return;
    const uint32_t next_head = (q / MERGE_GROUP + 1) * MERGE_GROUP;           // first aligned position after q
    if (q + 1 >= nchunks || q + 1 >= next_head || cont_bucket[q + 1] != b) return;   // single piece: nothing to fold
    B acc = ld_struct(cont + q);
    for (uint32_t r = q + 1; r < nchunks && r < next_head && cont_bucket[r] == b; r++) acc = bk_add_inl(acc, ld_struct(cont + r));
    st_struct(cont + q, acc);                                                   // only group heads are written; nobody else reads them in this launch
}
template <class B>
__global__ void __launch_bounds__(64) k_msm_merge_cont(const RedSets<B> S, uint32_t nchunks) {
    B* __restrict__ buckets = S.buckets[blockIdx.y]; const B* __restrict__ cont = S.cont[blockIdx.y]; const uint32_t* __restrict__ cont_bucket = S.cont_bucket[blockIdx.y];
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nchunks) return;
    const uint32_t b = cont_bucket[q];
    if (b == 0xffffffffu) return;
    if (q > 0 && cont_bucket[q - 1] == b) return;
    B acc = bk_add_inl(ld_struct(buckets + b), ld_struct(cont + q));
    for (uint32_t r = (q / MERGE_GROUP + 1) * MERGE_GROUP; r < nchunks && cont_bucket[r] == b; r += MERGE_GROUP) acc = bk_add_inl(acc, ld_struct(cont + r));
    st_struct(buckets + b, acc);
}

template <class B>
__device__ __forceinline__ B bk_mul_small(const B& p, uint32_t k) {
    B r = bk_inf<B>();
    if (k == 0) return r;
    for (int i = 31 - __builtin_clz(k); i >= 0; i--) {
        r = bk_dbl(r);
        if ((k >> i) & 1u) r = bk_add(r, p);
    }
    return r;
}

// lane (w, seg): sum_{b in [lo, lo+L)} (b+1) * B[w][b]  =  sum (b-lo+1) B_b  +  lo * sum B_b
template <class B>
__global__ void __launch_bounds__(64) k_msm_reduce_segments(const RedSets<B> S, uint32_t nb, uint32_t seg_len, int nwin) {
    const B* __restrict__ buckets = S.buckets[blockIdx.y]; B* __restrict__ partials = S.partials[blockIdx.y];
    const uint32_t segs = nb / seg_len;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nwin * segs) return;
    const uint32_t w = (uint32_t)(t / segs), seg = (uint32_t)(t % segs);
    const uint32_t lo = seg * seg_len;
    const B* Bk = buckets + (size_t)w * nb;
    B run = bk_inf<B>(), acc = bk_inf<B>();
    for (uint32_t b = lo + seg_len; b-- > lo;) {
        run = bk_add_inl(run, ld_struct(Bk + b));
        acc = bk_add_inl(acc, run);
    }
    acc = bk_add(acc, bk_mul_small(run, lo));
    st_struct(partials + t, acc);
}

// workgroup g: sums[g] = sum of its `segs` partials (strided serial sums, then an LDS tree), converted to canonical XYZZ for the host
template <class F, class B, int THREADS>
__global__ void __launch_bounds__(THREADS) k_msm_window_sum(const RedSets<B> S, uint32_t segs) {
    const B* __restrict__ partials = S.partials[blockIdx.y]; XYZZ<F>* __restrict__ window_sums = (XYZZ<F>*)S.wsums[blockIdx.y];
    extern __shared__ uint4 lds_raw[];
    B* sh = reinterpret_cast<B*>(lds_raw);
    const B* P = partials + (size_t)blockIdx.x * segs;
    B acc = bk_inf<B>();
    for (uint32_t s = threadIdx.x; s < segs; s += THREADS) acc = bk_add_inl(acc, ld_struct(P + s));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = THREADS / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) { acc = bk_add_inl(sh[threadIdx.x], sh[threadIdx.x + off]); }
        __syncthreads();
        if ((int)threadIdx.x < off) sh[threadIdx.x] = acc;
        __syncthreads();
    }
    if (threadIdx.x == 0) st_struct(window_sums + blockIdx.x, bk_to_xyzz<F>(sh[0]));
}

// Bucket reduction for SMALL bucket sets (shared bucket set with <= 2^16 buckets: shards of a multi-GPU plan, 2^16-constraint
// circuits), where the running-sum kernels above are a chain of ~45 dependent point additions on a handful of waves and that
// latency, not the work, is the whole cost of the MSM.  sum_w w B_{w-1} = sum_k 2^k T_k with T_k = sum of the buckets whose
// weight w has bit k set: c plain sums, each a log-depth tree, ~c/2 additions per bucket in total (nothing at this size).
//   k_msm_bitsum_partial: workgroup (k, g): 256 lanes x ITEMS buckets with bit k set, serial per lane + LDS tree
//   k_msm_bitsum_final:   one wave per bit folds the G workgroup results into T_k (canonical XYZZ for the host)
// The host finishes with one Horner pass over the c sums (one doubling per bit).
template <class B, int ITEMS>
__global__ void __launch_bounds__(256) k_msm_bitsum_partial(const RedSets<B> S, uint32_t nb, uint32_t groups) {
    const B* __restrict__ buckets = S.buckets[blockIdx.y]; B* __restrict__ partials = S.partials[blockIdx.y];
    extern __shared__ uint4 lds_raw[];
    B* sh = reinterpret_cast<B*>(lds_raw);
    const uint32_t k = blockIdx.x / groups, g = blockIdx.x % groups;
    const uint32_t j0 = (g * 256u + threadIdx.x) * ITEMS;
    B acc = bk_inf<B>();
    for (uint32_t i = 0; i < (uint32_t)ITEMS; i++) {
        const uint32_t j = j0 + i;                                            // j-th weight with bit k set
        const uint64_t w = (((uint64_t)j >> k) << (k + 1)) | ((uint64_t)1 << k) | (j & (((uint32_t)1 << k) - 1u));
        if (w >= 1 && w <= nb) acc = bk_add_inl(acc, ld_struct(buckets + (w - 1)));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) acc = bk_add_inl(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
        if ((int)threadIdx.x < off) sh[threadIdx.x] = acc;
        __syncthreads();
    }
    if (threadIdx.x == 0) st_struct(partials + blockIdx.x, sh[0]);
}
template <class F, class B>
__global__ void __launch_bounds__(64) k_msm_bitsum_final(const RedSets<B> S, uint32_t groups) {
    const B* __restrict__ partials = S.partials[blockIdx.y]; XYZZ<F>* __restrict__ bit_sums = (XYZZ<F>*)S.wsums[blockIdx.y];
    extern __shared__ uint4 lds_raw[];
    B* sh = reinterpret_cast<B*>(lds_raw);
    const B* P = partials + (size_t)blockIdx.x * groups;
    B acc = bk_inf<B>();
    for (uint32_t s = threadIdx.x; s < groups; s += 64) acc = bk_add_inl(acc, ld_struct(P + s));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 32; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) acc = bk_add_inl(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
        if ((int)threadIdx.x < off) sh[threadIdx.x] = acc;
        __syncthreads();
    }
    if (threadIdx.x == 0) st_struct(bit_sums + blockIdx.x, bk_to_xyzz<F>(sh[0]));
}

// Bucket reduction for LARGE shared bucket sets (per-window precomputed tables: 2^19 buckets at c = 20), two launches instead of a
// chain of ~90 dependent additions on half a wave per SIMD (k_msm_reduce_segments + k_msm_window_sum: 2.2 additions per bucket, 0.77 ms
// alone for G1, 1.97 ms for G2).  The bucket index is split b = hi * L + lo (L = 2^10 columns, H = nb / L rows), weight w = b + 1:
//     sum_b w B_b = L * sum_hi hi * R_hi + sum_lo (lo + 1) * C_lo,     R_hi = row sums, C_lo = column sums      (2 additions per bucket)
// and the two small weighted sums are done by bits, sum_w w X_w = sum_k 2^k T_k with T_k = the sum of the X_w whose weight has bit k
// set (plain log-depth tree sums), the host finishing with one Horner pass as for k_msm_bitsum_*.
//   k_msm_grid_partial: workgroup = 32 rows x 64 columns: 8 serial additions per lane in each direction + short LDS trees ->
//       colpart[H / 32][L] and rowpart[H][L / 64]                                   (2^19 buckets: 256 workgroups, 65 536 lanes)
//   k_msm_grid_bitsum:  workgroup (side, bit k, group g): 256 lanes x ITEMS partials whose weight has bit k set + LDS tree -> one
//       canonical XYZZ sum per workgroup for the host (62 of them at 2^19 buckets)
template <class B>
__global__ void __launch_bounds__(256) k_msm_grid_partial(const RedSets<B> S, uint32_t log_l, uint32_t nb) {
    const B* __restrict__ buckets = S.buckets[blockIdx.y];
    B* __restrict__ colpart = S.partials[blockIdx.y]; B* __restrict__ rowpart = colpart + (size_t)((nb >> log_l) / GRID_TR) * ((size_t)1 << log_l);   // [H / TR][L], then [H][L / TC]
    extern __shared__ uint4 lds_raw[];
    B* sh = reinterpret_cast<B*>(lds_raw);
    const uint32_t L = 1u << log_l, ncb = L / GRID_TC;
    const uint32_t cb = blockIdx.x % ncb, rb = blockIdx.x / ncb, t = threadIdx.x;
    const B* tile = buckets + (size_t)rb * GRID_TR * L + (size_t)cb * GRID_TC;
    constexpr uint32_t RPQ = GRID_TR / 4;                    // rows per lane in the column phase (four lanes share a column)
    constexpr uint32_t SEGS = 256 / GRID_TR, CPS = GRID_TC / SEGS;   // row phase: SEGS lanes share a row, CPS columns each
    {   // columns: lane = (row quarter, column); RPQ rows each, then 4 -> 1 through LDS
        const uint32_t col = t & 63u, rq = t >> 6;
        B acc = ld_struct(tile + (size_t)(rq * RPQ) * L + col);
        for (uint32_t i = 1; i < RPQ; i++) acc = bk_add_inl(acc, ld_struct(tile + (size_t)(rq * RPQ + i) * L + col));
        sh[t] = acc;
        __syncthreads();
        for (int off = 128; off >= 64; off >>= 1) {
            if ((int)t < off) acc = bk_add_inl(sh[t], sh[t + off]);
            __syncthreads();
            if ((int)t < off) sh[t] = acc;
            __syncthreads();
        }
        if (t < 64) st_struct(colpart + (size_t)rb * L + (size_t)cb * GRID_TC + t, acc);
        __syncthreads();
    }
    {   // rows: lane = (row, segment of CPS columns); SEGS -> 1 through LDS
        const uint32_t row = t / SEGS, seg = t % SEGS;
        const B* src = tile + (size_t)row * L + seg * CPS;
        B acc = ld_struct(src);
        for (uint32_t i = 1; i < CPS; i++) acc = bk_add_inl(acc, ld_struct(src + i));
        sh[t] = acc;
        __syncthreads();
        for (uint32_t off = SEGS / 2; off >= 1; off >>= 1) {
            if (seg < off) acc = bk_add_inl(sh[t], sh[t + off]);
            __syncthreads();
            if (seg < off) sh[t] = acc;
            __syncthreads();
        }
        if (seg == 0) st_struct(rowpart + (size_t)(rb * GRID_TR + row) * ncb + cb, acc);
    }
}
// j-th integer (j >= 0) whose bit k is set
__device__ __forceinline__ uint32_t nth_with_bit(uint32_t j, uint32_t k) { return ((j >> k) << (k + 1)) | (1u << k) | (j & ((1u << k) - 1u)); }
template <class F, class B, int ITEMS>
__global__ void __launch_bounds__(256) k_msm_grid_bitsum(const RedSets<B> S, uint32_t log_l, uint32_t log_h, uint32_t gc, uint32_t gr) {
    const B* __restrict__ colpart = S.partials[blockIdx.y]; const B* __restrict__ rowpart = colpart + (size_t)(((size_t)1 << log_h) / GRID_TR) * ((size_t)1 << log_l);
    XYZZ<F>* __restrict__ sums = (XYZZ<F>*)S.wsums[blockIdx.y];
    extern __shared__ uint4 lds_raw[];
    B* sh = reinterpret_cast<B*>(lds_raw);
    const uint32_t L = 1u << log_l, H = 1u << log_h, nrb = H / GRID_TR, ncb = L / GRID_TC, t = threadIdx.x;
    const uint32_t njc = (log_l + 1) * gc;
    B acc = bk_inf<B>();
    if (blockIdx.x < njc) {                                   // column side: element (rb, j), weight = column + 1 = j-th value with bit k, <= L
        const uint32_t k = blockIdx.x / gc, g = blockIdx.x % gc, half = L / 2;
        for (uint32_t i = 0; i < (uint32_t)ITEMS; i++) {
            const uint32_t e = (g * 256u + t) * ITEMS + i;
            if (e >= nrb * half) break;
            const uint32_t w = nth_with_bit(e % half, k);
            if (w <= L) acc = bk_add_inl(acc, ld_struct(colpart + (size_t)(e / half) * L + (w - 1)));
        }
    } else {                                                  // row side: element (j, cb), weight = row = j-th value with bit k, < H
        const uint32_t idx = blockIdx.x - njc, k = idx / gr, g = idx % gr;
        for (uint32_t i = 0; i < (uint32_t)ITEMS; i++) {
            const uint32_t e = (g * 256u + t) * ITEMS + i;
            if (e >= (H / 2) * ncb) break;
            const uint32_t w = nth_with_bit(e / ncb, k);
            if (w < H) acc = bk_add_inl(acc, ld_struct(rowpart + (size_t)w * ncb + (e % ncb)));
        }
    }
    sh[t] = acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)t < off) acc = bk_add_inl(sh[t], sh[t + off]);
        __syncthreads();
        if ((int)t < off) sh[t] = acc;
        __syncthreads();
    }
    if (t == 0) st_struct(sums + blockIdx.x, bk_to_xyzz<F>(sh[0]));
}

// arkworks in-memory affine (x, y, infinity flag at `inf_off`, arbitrary stride) or packed zkey points -> packed device layout
template <class F>
__global__ void __launch_bounds__(256) k_pack_bases(const uint8_t* __restrict__ src, size_t n, size_t stride, long inf_off, Affine<F>* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t* p = src + i * stride;
        Affine<F> a;
        uint32_t* w = reinterpret_cast<uint32_t*>(&a);
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
        for (int k = 0; k < (int)(sizeof(Affine<F>) / 4); k++) w[k] = q[k];
        if (inf_off >= 0 && p[inf_off]) a = Affine<F>::infinity();
        st_struct(dst + i, a);
    }
}


// dst[j] = src[idx[j]]   (compaction of a table whose records are partly the point at infinity)
template <class F>
__global__ void __launch_bounds__(256) k_gather_points(Affine<F>* __restrict__ dst, const Affine<F>* __restrict__ src, const uint32_t* __restrict__ idx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_struct(dst + i, ld_struct(src + idx[i]));
}

// zkey fast path: the reference's parser checks every point on the CPU (`circom-types/src/traits.rs:118-123,
// 148-153`: is_on_curve, then the subgroup check).  On-curve: y^2 == x^3 + b for every non-infinity record, counted on the
// device.  b is passed in Montgomery form (BN254 G1: 3, G2: 3/(9+u); BLS12-381 G1: 4, G2: 4(1+u)).
template <class F>
__global__ void __launch_bounds__(256) k_check_on_curve(const Affine<F>* __restrict__ pts, size_t n, F b, unsigned long long* __restrict__ n_bad, unsigned long long* __restrict__ first_bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Affine<F> p = ld_struct(pts + i);
        if (p.is_inf()) continue;
        F lhs = p.y.sqr(), rhs = p.x.sqr() * p.x + b;
        if (lhs != rhs) { atomicAdd(n_bad, 1ull); atomicMin(first_bad, (unsigned long long)i); }
    }
}

// The same predicate through the curve's endomorphisms (subgroup.hpp): 4 to 16 times fewer group operations than [r]P.
template <class F>
__global__ void __launch_bounds__(128) k_check_subgroup_fast(const Affine<F>* __restrict__ pts, size_t n, FastSubgroup<F> c, unsigned long long* __restrict__ n_bad, unsigned long long* __restrict__ first_bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Affine<F> p = ld_struct(pts + i);
        if (!c.contains(p)) { atomicAdd(n_bad, 1ull); atomicMin(first_bad, (unsigned long long)i); }
    }
}

// Subgroup check of the same parser step (`is_in_correct_subgroup_assuming_on_curve`): [r]P must be the point at infinity, r = the
// scalar-field modulus (a plain double-and-add over the 254/255 bits of r; every lane walks the same bits, so no divergence).
template <class F, class FrP>
__global__ void __launch_bounds__(128) k_check_subgroup(const Affine<F>* __restrict__ pts, size_t n, unsigned long long* __restrict__ n_bad, unsigned long long* __restrict__ first_bad) {
    uint32_t k[FrP::N];
    _Pragma("unroll") for (int j = 0; j < FrP::N; j++) k[j] = FrP::P[j];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Affine<F> p = ld_struct(pts + i);
        if (p.is_inf()) continue;
        XYZZ<F> r = XYZZ<F>::infinity();
        for (int b = FrP::BITS - 1; b >= 0; b--) {
            r = xyzz_dbl(r);
            if ((k[b >> 5] >> (b & 31)) & 1u) r = xyzz_madd(r, p.x, p.y);
        }
        if (!r.is_inf()) { atomicAdd(n_bad, 1ull); atomicMin(first_bad, (unsigned long long)i); }
    }
}

// Per-window precomputed tables: dst[i] = 2^c * src[i] in affine form (one inversion per point; run once per zkey table).
// The conversion back to affine costs one field inversion (~380 products by Fermat, twice the c doublings): a lane takes K points a grid
// stride apart and inverts the product of their zzz once (Montgomery's trick: 3 products per point + one inversion per K points).
template <class F, int K>
__global__ void __launch_bounds__(256) k_precompute_window(const Affine<F>* __restrict__ src, Affine<F>* __restrict__ dst, size_t n, int c) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += (size_t)K * stride) {
        XYZZ<F> a[K]; F pre[K];
        F run = F::one();
        _Pragma("unroll") for (int j = 0; j < K; j++) {
            const size_t i = i0 + (size_t)j * stride;
            a[j] = i < n ? XYZZ<F>::from_affine(ld_struct(src + i)) : XYZZ<F>::infinity();
            for (int k = 0; k < c; k++) a[j] = xyzz_dbl(a[j]);
            pre[j] = run;
            if (!a[j].is_inf()) run = run * a[j].zzz;
        }
        F inv = fp_inverse(run);                                   // 1 / (product of the finite points' zzz)
        _Pragma("unroll") for (int j = K - 1; j >= 0; j--) {
            const size_t i = i0 + (size_t)j * stride;
            if (i >= n) continue;
            if (a[j].is_inf()) { st_struct(dst + i, Affine<F>::infinity()); continue; }
            const F izzz = inv * pre[j];                           // 1 / zzz_j
            inv = inv * a[j].zzz;
            const F iz = izzz * a[j].zz;                           // zz / zzz = 1 / z
            const F izz = iz.sqr();
            st_struct(dst + i, Affine<F>{a[j].x * izz, a[j].y * izzz});
        }
    }
}

// Synthetic point tables (bench / test tooling, not on the prover path): out[i] = affine(hi[i >> log_t] + lo[i & (2^log_t-1)]).
// With lo[j] = (first+j)*P and hi[j] = (j << log_t)*P this yields the consecutive multiples (first+i)*P — valid, pairwise
// distinct curve points whose discrete logs are known, so an MSM over them can be checked at any size:
// sum_i s_i (first+i) P = (sum_i s_i (first+i)) P.
template <class F>
__global__ void __launch_bounds__(256) k_synth_points(const XYZZ<F>* __restrict__ lo, const XYZZ<F>* __restrict__ hi, int log_t, size_t n, Affine<F>* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        XYZZ<F> a = ld_struct(lo + (i & (((size_t)1 << log_t) - 1)));
        XYZZ<F> b = ld_struct(hi + (i >> log_t));
        st_struct(out + i, xyzz_to_affine(xyzz_add(a, b)));
    }
}

// Fixed-base batch multiplication (setup tooling, not on the prover path): out[i] = s_i * G for Montgomery scalars s_i, by 8-bit
// windows over the table tab[w][d - 1] = d * 2^(8w) * G (affine; 32 x 255 records, cache resident).  Used to build a synthetic but
// VALID Groth16 CRS on the device (the [u_i(tau)]_1, [v_i(tau)]_2, ... queries of a zkey are scalar multiples of the generators).
template <class F>
__global__ void __launch_bounds__(256) k_fixed_base_table(Affine<F> g, int nwin, Affine<F>* __restrict__ tab) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwin * 255) return;
    const int w = t / 255, d = t % 255 + 1;
    uint32_t k[9]; for (int i = 0; i < 9; i++) k[i] = 0;
    k[w / 4] = (uint32_t)d << (8 * (w % 4));                     // d * 2^(8w)
    st_struct(tab + t, xyzz_to_affine(xyzz_scalar_mul(XYZZ<F>::from_affine(g), k, 9)));
}
template <class F, class Fr>
__global__ void __launch_bounds__(128) k_fixed_base_mul(const Fr* __restrict__ scalars, size_t n, int nwin, const Affine<F>* __restrict__ tab, Affine<F>* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Fr s = ld_fp(scalars + i).from_mont();
        XYZZ<F> acc = XYZZ<F>::infinity();
        for (int w = 0; w < nwin; w++) {
            const uint32_t d = (s.v[w / 4] >> (8 * (w % 4))) & 0xffu;
            if (d) { const Affine<F> p = ld_struct(tab + (size_t)w * 255 + (d - 1)); acc = xyzz_madd(acc, p.x, p.y); }
        }
        st_struct(out + i, xyzz_to_affine(acc));
    }
}

}  // namespace cg
