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
#include "curve.hpp"
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
    static constexpr bool USES_LDS = false;
    XYZZ<F> v;
    bool inf;
    __device__ __forceinline__ void init(uint4*, int, int) { inf = true; }
    __device__ __forceinline__ F get(int f) const { return f == 0 ? v.x : f == 1 ? v.y : f == 2 ? v.zz : v.zzz; }
    __device__ __forceinline__ void set(int f, const F& x) { if (f == 0) v.x = x; else if (f == 1) v.y = x; else if (f == 2) v.zz = x; else v.zzz = x; }
};
template <class F>
struct LdsAcc {
    static constexpr bool USES_LDS = true;
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

// Bucket accumulation, chunk-balanced: lane q folds the L consecutive entries sorted[q*L, (q+1)*L) of the (window, bucket)-
// sorted index list, whatever buckets they belong to, so every lane of a wave does the same number of mixed additions
// (one lane per bucket wastes ~20 % of the wave on the Poisson spread of bucket sizes, and serialises skewed buckets).
//   * a bucket whose first entry lies in this chunk gets its partial sum written to buckets[b];
//   * the leading piece of the chunk that continues a bucket begun in an earlier chunk goes to cont[q] (tagged cont_bucket[q]);
// k_msm_merge_cont then adds the continuation pieces into their buckets (one lane per bucket run, no atomics).
template <class F, class Acc, int THREADS>
__global__ void __launch_bounds__(THREADS) k_msm_accumulate(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                            uint32_t nbuckets, uint32_t chunk_len, uint32_t nchunks,
                                                            XYZZ<F>* __restrict__ buckets, XYZZ<F>* __restrict__ cont, uint32_t* __restrict__ cont_bucket) {
    extern __shared__ uint4 acc_lds[];
    const uint32_t q = blockIdx.x * THREADS + threadIdx.x;
    if (q >= nchunks) return;
    const uint32_t total = offsets[nbuckets - 1] + counts[nbuckets - 1];
    uint32_t pos = q * chunk_len;
    if (pos >= total) { cont_bucket[q] = 0xffffffffu; return; }
    const uint32_t end = min(pos + chunk_len, total);
    // bucket containing entry `pos`: last b with offsets[b] <= pos (empty buckets share an offset with their successor; skip them)
    uint32_t lo = 0, hi = nbuckets - 1;
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (offsets[mid] <= pos) lo = mid; else hi = mid - 1; }
    uint32_t b = lo;
    uint32_t bend = offsets[b] + counts[b];
    bool continuation = offsets[b] != pos;
    cont_bucket[q] = continuation ? b : 0xffffffffu;
    Acc acc;
    acc.init(acc_lds, threadIdx.x, THREADS);
    while (pos < end) {
        if (pos == bend) {                                   // finished bucket b inside this chunk
            if (continuation) { acc_flush(acc, cont + q); continuation = false; } else acc_flush(acc, buckets + b);
            do { b++; } while (counts[b] == 0);
            bend = offsets[b] + counts[b];
        }
        const uint32_t e = sorted[pos++];
        Affine<F> p = ld_struct(bases + (e & 0x7fffffffu));
        if (p.is_inf()) continue;
        if (e >> 31) p.y = p.y.neg();
        acc_madd(acc, p.x, p.y);
    }
    if (continuation) acc_flush(acc, cont + q); else acc_flush(acc, buckets + b);
}

// lane q: if chunk q holds the FIRST continuation piece of its bucket, fold all consecutive continuation pieces of that bucket
// (chunks q, q+1, ... with the same tag) into buckets[b].  With uniform scalars a bucket spans at most 2-3 chunks.
template <class F>
__global__ void __launch_bounds__(64) k_msm_merge_cont(XYZZ<F>* __restrict__ buckets, const XYZZ<F>* __restrict__ cont, const uint32_t* __restrict__ cont_bucket, uint32_t nchunks) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nchunks) return;
    const uint32_t b = cont_bucket[q];
    if (b == 0xffffffffu) return;
    if (q > 0 && cont_bucket[q - 1] == b) return;
    XYZZ<F> acc = ld_struct(buckets + b);
    for (uint32_t r = q; r < nchunks && cont_bucket[r] == b; r++) acc = xyzz_add(acc, ld_struct(cont + r));
    st_struct(buckets + b, acc);
}

template <class F>
__device__ __forceinline__ XYZZ<F> xyzz_mul_small(const XYZZ<F>& p, uint32_t k) {
    XYZZ<F> r = XYZZ<F>::infinity();
    if (k == 0) return r;
    for (int i = 31 - __builtin_clz(k); i >= 0; i--) {
        r = xyzz_dbl(r);
        if ((k >> i) & 1u) r = xyzz_add(r, p);
    }
    return r;
}

// lane (w, seg): sum_{b in [lo, lo+L)} (b+1) * B[w][b]  =  sum (b-lo+1) B_b  +  lo * sum B_b
template <class F>
__global__ void __launch_bounds__(64) k_msm_reduce_segments(const XYZZ<F>* __restrict__ buckets, uint32_t nb, uint32_t seg_len, int nwin,
                                                            XYZZ<F>* __restrict__ partials) {
    const uint32_t segs = nb / seg_len;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nwin * segs) return;
    const uint32_t w = (uint32_t)(t / segs), seg = (uint32_t)(t % segs);
    const uint32_t lo = seg * seg_len;
    const XYZZ<F>* B = buckets + (size_t)w * nb;
    XYZZ<F> run = XYZZ<F>::infinity(), acc = XYZZ<F>::infinity();
    for (uint32_t b = lo + seg_len; b-- > lo;) {
        run = xyzz_add(run, ld_struct(B + b));
        acc = xyzz_add(acc, run);
    }
    acc = xyzz_add(acc, xyzz_mul_small(run, lo));
    st_struct(partials + t, acc);
}

// workgroup w: window_sums[w] = sum of its `segs` partials (strided serial sums, then an LDS tree)
template <class F, int THREADS>
__global__ void __launch_bounds__(THREADS) k_msm_window_sum(const XYZZ<F>* __restrict__ partials, uint32_t segs, XYZZ<F>* __restrict__ window_sums) {
    extern __shared__ uint4 lds_raw[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(lds_raw);
    const XYZZ<F>* P = partials + (size_t)blockIdx.x * segs;
    XYZZ<F> acc = XYZZ<F>::infinity();
    for (uint32_t s = threadIdx.x; s < segs; s += THREADS) acc = xyzz_add(acc, ld_struct(P + s));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = THREADS / 2; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) { acc = xyzz_add(sh[threadIdx.x], sh[threadIdx.x + off]); }
        __syncthreads();
        if ((int)threadIdx.x < off) sh[threadIdx.x] = acc;
        __syncthreads();
    }
    if (threadIdx.x == 0) st_struct(window_sums + blockIdx.x, sh[0]);
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

}  // namespace cg
