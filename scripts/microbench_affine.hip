// Go / no-go measurement for batched-AFFINE bucket accumulation ("Next round" item 2): lambda = dy * (dx)^-1 with the
// inversions shared by Montgomery's trick — per lane over B independent additions, across the 256 lanes of a workgroup through an LDS
// product tree, ONE field inversion per workgroup — against the XYZZ mixed addition the product ships (10 products, no inversion;
// scripts/microbench_clock.hip: 16.0-17.5 G additions/s with operands in registers, the "floor" of k_msm_accumulate_pf<G1>).
// Same arithmetic core as the product (csrc/lazy29.hpp, 9 x 29-bit signed lazy limbs), operands generated in registers (no gathers, no
// bucket boundaries: everything a real kernel adds on top — re-reading the points for the second pass, scratch for the tree levels —
// is NOT in these numbers).  Three figures:
//   1. affine core alone: the 5 products + 1 squaring per addition (1 prefix product, 2 for the back-substitution, lambda, lambda^2,
//      lambda * (x1 - x3)) with the inverse of the lane's product GIVEN — the ceiling of any batched-affine scheme;
//   2. the inversion: Fermat a^(p-2) on the same core, in product-equivalents;
//   3. the whole scheme, B = 2 / 4 / 8 additions per lane: per-lane products -> LDS up-sweep -> one inversion (lane 0) -> down-sweep ->
//      per-lane back-substitution and additions.  This is the number to hold against 17.5 G/s; the go threshold was 24 G/s.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../include scripts/microbench_affine.hip -o scripts/_build/microbench_affine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <utility>
#include "../collaborative-circom_amd/csrc/common.hpp"
#include "../collaborative-circom_amd/csrc/msm_kernels.hpp"
using namespace cg;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
typedef L29<Bn254Fq> L;
// compile-time loop: the index is a constant in the front end already, so the per-lane arrays below are plain registers
template <int N, class Fn, int... J> __device__ __forceinline__ void static_for_impl(Fn&& fn, std::integer_sequence<int, J...>) { (fn(std::integral_constant<int, J>{}), ...); }
template <int N, class Fn> __device__ __forceinline__ void static_for(Fn&& fn) { static_for_impl<N>(fn, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ uint32_t hash32(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }
// a pseudo-random normalised value (limbs below 2^29, top limb small): stands for a coordinate already in limb form
__device__ __forceinline__ L gen(uint32_t key) {
    L r;
    _Pragma("unroll") for (int k = 0; k < L::NL; k++) r.l[k] = (int32_t)(hash32(key * 16u + k) & L::MASK);
    r.l[L::NL - 1] &= 0x3fffff;
    return r;
}
// a^(p-2): square-and-multiply over the bits of p - 2 (253 squarings + popcount products)
__device__ __noinline__ L fermat_inverse(const L& a, int* nsqr, int* nmul) {
    typedef Bn254Fq::Params P;
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = P::P[i];
    e[0] -= 2;                                                   // p is odd and p[0] >= 2: no borrow
    L r = a; int s = 0, m = 0;
    bool started = false;
    for (int i = 255; i >= 0; i--) {
        const bool bit = (e[i >> 5] >> (i & 31)) & 1u;
        if (!started) { started = bit; continue; }
        r = L::sqr(r); s++;
        if (bit) { r = L::mul(r, a); m++; }
    }
    if (nsqr) { *nsqr = s; *nmul = m; }
    return r;
}

// One lane's share of a batch: pass 1 = prefix products of the B differences x2 - x1; pass 2 (backwards) = the individual inverses
// from the inverse of the lane's product, and the B additions.  ROLLED: the two passes stay loops and the B prefix products live in the
// lane's private memory (scratch, 36 bytes each); otherwise everything is unrolled and the compiler keeps what it can in VGPRs.
template <int B, bool ROLLED>
struct LaneBatch {
    L pre[B];
    __device__ __forceinline__ void step1(uint32_t key, int j) {
        const L dx = gen(key + j * 4u + 1) - gen(key + j * 4u);
        pre[j] = j ? L::mul(pre[j > 0 ? j - 1 : 0], dx) : dx.norm();
    }
    __device__ __forceinline__ void step2(uint32_t key, int j, L& I, uint32_t& sum) {
        const L x1 = gen(key + j * 4u), x2 = gen(key + j * 4u + 1), y1 = gen(key + j * 4u + 2), y2 = gen(key + j * 4u + 3);
        L inv = I;
        if (j > 0) { inv = L::mul(I, pre[j - 1]); I = L::mul(I, x2 - x1); }
        const L lam = L::mul(y2 - y1, inv);
        const L x3 = (L::sqr(lam) - x1 - x2).norm();
        const L y3 = L::mul(lam, (x1 - x3).norm()) - y1;
        sum ^= (uint32_t)x3.l[0] ^ (uint32_t)y3.l[3];
    }
    __device__ __forceinline__ void pass1(uint32_t key) {
        if constexpr (ROLLED) { _Pragma("nounroll") for (int j = 0; j < B; j++) step1(key, j); }
        else static_for<B>([&](auto jc) { step1(key, decltype(jc)::value); });
    }
    __device__ __forceinline__ void pass2(uint32_t key, L& I, uint32_t& sum) {
        if constexpr (ROLLED) { _Pragma("nounroll") for (int j = B - 1; j >= 0; j--) step2(key, j, I, sum); }
        else static_for<B>([&](auto jc) { step2(key, B - 1 - decltype(jc)::value, I, sum); });
    }
};

// 1. the affine core with the inverse given: B additions per iteration and lane
template <int B, int MINW, bool ROLLED>
__global__ void __launch_bounds__(256, MINW) k_affine_core(uint32_t* out, uint32_t seed, int iters) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    L I = gen(seed ^ lane);                                       // "inverse of the lane's product": any value, the cost is what is measured
    uint32_t sum = 0;
    LaneBatch<B, ROLLED> lb;
    for (int it = 0; it < iters; it++) {
        const uint32_t key = lane * 977u + it * 131u;
        lb.pass1(key);
        lb.pass2(key, I, sum);
        I.l[0] ^= (int32_t)(sum & 0xff);
    }
    out[lane] = sum;
}

// 2. inversions alone (every lane its own)
__global__ void __launch_bounds__(256, 3) k_fermat(uint32_t* out, uint32_t seed, int iters, int* counts) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    L a = gen(seed ^ lane);
    int s = 0, m = 0;
    for (int it = 0; it < iters; it++) a = fermat_inverse(a, &s, &m);
    uint32_t x = 0; for (int k = 0; k < L::NL; k++) x ^= (uint32_t)a.l[k];
    out[lane] = x;
    if (lane == 0) { counts[0] = s; counts[1] = m; }
}

// 3. the whole scheme
template <int B, int MINW, bool ROLLED>
__global__ void __launch_bounds__(256, MINW) k_affine_batched(uint32_t* out, uint32_t seed, int iters, uint32_t* check) {
    __shared__ L tree[511];                                        // level l starts at 512 - (512 >> l): 256 leaves, 128, ..., 1
    const uint32_t t = threadIdx.x, lane = blockIdx.x * 256 + t;
    uint32_t sum = 0, bad = 0;
    LaneBatch<B, ROLLED> lb;
    for (int it = 0; it < iters; it++) {
        const uint32_t key = lane * 977u + it * 131u;
        lb.pass1(key);
        tree[t] = lb.pre[B - 1];
        __syncthreads();
        int base = 0;
        for (int w = 128; w >= 1; w >>= 1) {                      // up-sweep: node(l + 1, i) = node(l, 2i) * node(l, 2i + 1)
            if ((int)t < w) tree[base + 2 * w + t] = L::mul(tree[base + 2 * t], tree[base + 2 * t + 1]);
            base += 2 * w;
            __syncthreads();
        }
        if (t == 0) {
            const L root = tree[510];
            const L inv = fermat_inverse(root, nullptr, nullptr);
            if (it == 0 && blockIdx.x == 0) { const Bn254Fq one = L::to_fp(L::mul(root, inv)); bad = !(one == Bn254Fq::one()); }
            tree[510] = inv;
        }
        __syncthreads();
        for (int w = 1; w <= 128; w <<= 1) {                      // down-sweep: inv(left) = inv(parent) * right, inv(right) = inv(parent) * left
            base -= 2 * w;
            if ((int)t < w) {
                const L I = tree[base + 2 * w + t], a = tree[base + 2 * t], b = tree[base + 2 * t + 1];
                tree[base + 2 * t] = L::mul(I, b); tree[base + 2 * t + 1] = L::mul(I, a);
            }
            __syncthreads();
        }
        L I = tree[t];
        if (it == 0 && blockIdx.x == 0 && t == 77) { const Bn254Fq one = L::to_fp(L::mul(lb.pre[B - 1], I)); bad |= !(one == Bn254Fq::one()) ? 2u : 0u; }
        lb.pass2(key, I, sum);
        __syncthreads();
    }
    out[lane] = sum;
    if (bad) atomicOr(check, bad);
}

template <class K> static int timed(const char* name, K launch, double adds, hipEvent_t e0, hipEvent_t e1) {
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (adds > 0) printf("%-64s rep %d: %8.3f ms  %6.2f G additions/s  (%.2f ms per 54.5 M)\n", name, rep, ms, adds / (ms * 1e-3) / 1e9, 54.5e6 / (adds / (ms * 1e-3)) * 1e3);
        else printf("%-64s rep %d: %8.3f ms\n", name, rep, ms);
    }
    return 0;
}

int main() {
    uint32_t* d; uint32_t* chk; int* cnt;
    const int WG = 1536 * 2;                                       // 4 full residency rounds of 3 workgroups per CU
    CHK(hipMalloc(&d, (size_t)WG * 256 * 4)); CHK(hipMalloc(&chk, 4)); CHK(hipMalloc(&cnt, 8)); CHK(hipMemset(chk, 0, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("== 1. affine addition core, inverse given (5 products + 1 squaring per addition), %d workgroups x 256 lanes\n", WG);
#define CORE(B, W, R, IT) if (timed("B = " #B ", " #W " waves/SIMD budget, " #R, [&] { hipLaunchKernelGGL((k_affine_core<B, W, R>), dim3(WG), dim3(256), 0, 0, d, 7u, IT); }, (double)WG * 256 * B * IT, e0, e1)) return 1;
    CORE(2, 3, false, 64) CORE(4, 3, false, 32) CORE(4, 2, false, 32) CORE(8, 2, false, 16) CORE(8, 3, true, 16) CORE(16, 3, true, 8)
    printf("== 2. Fermat inversion a^(p-2) on the same core, every lane its own, %d workgroups x 256 lanes x 4\n", WG);
    {
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fermat, dim3(WG), dim3(256), 0, 0, d, 9u, 4, cnt); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            int h[2]; CHK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
            const double inv_per_s = (double)WG * 256 * 4 / (ms * 1e-3);
            printf("rep %d: %8.3f ms  %6.3f G inversions/s  (%d squarings + %d products each; at 165 G products/s one inversion = %.0f product times)\n", rep, ms, inv_per_s / 1e9, h[0], h[1], 165e9 / inv_per_s);
        }
    }
    printf("== 3. batched affine, one inversion per 256-lane workgroup (LDS product tree), %d workgroups\n", WG);
#define FULL(B, W, R, IT) if (timed("B = " #B " additions per lane, " #W " waves/SIMD budget, " #R, [&] { hipLaunchKernelGGL((k_affine_batched<B, W, R>), dim3(WG), dim3(256), 0, 0, d, 7u, IT, chk); }, (double)WG * 256 * B * IT, e0, e1)) return 1;
    FULL(2, 3, false, 16) FULL(4, 3, false, 16) FULL(4, 2, false, 16) FULL(8, 2, false, 8) FULL(8, 3, true, 8) FULL(16, 3, true, 4) FULL(32, 3, true, 2)
    uint32_t bad; CHK(hipMemcpy(&bad, chk, 4, hipMemcpyDeviceToHost));
    printf("self-check (root * inverse == 1, lane product * its inverse == 1): %s\n", bad ? "FAILED" : "ok");
    printf("== reference: XYZZ mixed addition with operands in registers (scripts/microbench_clock.hip), same launch shape\n");
    {
        // reuse the product's accumulator policy: identical to k_madd_chain of microbench_clock.hip
        printf("   see profiles/r02_microbench_clock.txt: 17.1-17.6 G additions/s (3072 workgroups), 15.9-16.1 G/s (1660 workgroups)\n");
    }
    return bad ? 2 : 0;
}
