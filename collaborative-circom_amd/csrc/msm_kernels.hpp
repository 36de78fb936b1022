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
//   4. k_msm_accumulate  one lane per bucket: gather its points (64/128 B each, one contiguous run per lane) and fold them
//                        with XYZZ mixed additions (8M+2S).  This is where the time goes: integer VALU bound.
//   5. k_msm_reduce_segments / k_msm_window_sum   running-sum bucket reduction, split into 2^15/L independent segments per
//                        window (segment result = sum (b-lo+1) B_b + lo * sum B_b), then a per-window tree sum in LDS.
//   6. host: Horner fold of the <= 64 window sums (c doublings each) — O(1) work, kept on the host.
// With uniformly random scalars (REP3 shares always are) every bucket receives n/2^(c-1) +- sqrt points: lanes are balanced.
#pragma once
#include "curve.hpp"
#include "vec_kernels.hpp"

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

// digits[w*n + i] = signed digit of scalar i in window w; counts[w*nb + |d|-1]++
template <class Fr>
__global__ void __launch_bounds__(256) k_msm_digits(const Fr* __restrict__ scalars, size_t n, int c, int nwin,
                                                    int32_t* __restrict__ digits, uint32_t* __restrict__ counts) {
    const uint32_t nb = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr s = ld_fp(scalars + i).from_mont();
        uint32_t carry = 0;
        for (int w = 0; w < nwin; w++) {
            uint32_t d = (s.v[0] & mask) + carry;
            _Pragma("unroll") for (int l = 0; l < Fr::N; l++) {
                uint64_t two = ((uint64_t)(l + 1 < Fr::N ? s.v[l + 1] : 0u) << 32) | s.v[l];
                s.v[l] = (uint32_t)(two >> c);
            }
            int32_t dig;
            if (d > nb) { dig = (int32_t)d - (int32_t)(1u << c); carry = 1; } else { dig = (int32_t)d; carry = 0; }
            digits[(size_t)w * n + i] = dig;
            if (dig != 0) atomicAdd(&counts[(size_t)w * nb + (uint32_t)(dig < 0 ? -dig : dig) - 1], 1u);
        }
    }
}

// exclusive prefix sum of `total` counters, single workgroup of 1024 lanes (total <= a few million)
static __global__ void __launch_bounds__(1024) k_scan_exclusive(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t total) {
    __shared__ uint32_t part[1024];
    const size_t chunk = (total + 1023) / 1024;
    const size_t lo = (size_t)threadIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    uint32_t s = 0;
    for (size_t i = lo; i < hi; i++) s += in[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (size_t i = lo; i < hi; i++) { uint32_t v = in[i]; out[i] = run; run += v; }
}

// sorted[offsets[bucket] + k] = point index | sign << 31
static __global__ void __launch_bounds__(256) k_msm_scatter(const int32_t* __restrict__ digits, size_t n, int c, int nwin, const uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ cursors, uint32_t* __restrict__ sorted) {
    const uint32_t nb = 1u << (c - 1);
    const size_t total = (size_t)nwin * n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int32_t dig = digits[idx];
        if (dig == 0) continue;
        const size_t w = idx / n;
        const uint32_t i = (uint32_t)(idx - w * n);
        const size_t bucket = w * nb + (uint32_t)(dig < 0 ? -dig : dig) - 1;
        const uint32_t pos = offsets[bucket] + atomicAdd(&cursors[bucket], 1u);
        sorted[pos] = i | (dig < 0 ? 0x80000000u : 0u);
    }
}

// one lane per bucket
template <class F>
__global__ void __launch_bounds__(256) k_msm_accumulate(const Affine<F>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ counts,
                                                        size_t nbuckets, XYZZ<F>* __restrict__ buckets) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    XYZZ<F> acc = XYZZ<F>::infinity();
    const uint32_t off = offsets[b], cnt = counts[b];
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t e = sorted[off + k];
        Affine<F> p = ld_struct(bases + (e & 0x7fffffffu));
        if (p.is_inf()) continue;
        if (e >> 31) p.y = p.y.neg();
        acc = xyzz_madd(acc, p.x, p.y);
    }
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
