// Sustained-clock check for the integer multiplier (gfx950): v_mad_u64_u32 throughput and the shader clock actually held
// (s_memtime ticks / wall_clock64 at 100 MHz) for kernel durations from a 0.3 ms burst to ~100 ms.  The per-instruction peaks in
// profiles/microbench_r01.txt are 0.3 ms bursts; a 5 ms bucket-accumulation launch repeated back to back runs under the
// chip's power management, so the peak to price it against is the sustained one.
// Second part: instruction mix of a 52-bit-limb floating-point Montgomery product (5 x 5 limbs; per limb pair 2 v_fma_f64 +
// 1 v_add_f64 + two 64-bit integer accumulations, Emmart-style) against the 29-bit-limb integer product (9 x 9 limbs, 162
// v_mad_u64_u32 + carries) of csrc/lazy29.hpp — both as dependent chains of products, 3 waves per SIMD.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../include scripts/microbench_clock.hip -o scripts/_build/microbench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../collaborative-circom_amd/csrc/common.hpp"
#include "../collaborative-circom_amd/csrc/msm_kernels.hpp"
using namespace cg;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_mad_sustained(uint32_t* out, unsigned long long* clk, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
    uint32_t x = a + threadIdx.x, y = b;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) acc[i] = (uint64_t)x * (uint32_t)(y + i) + acc[i]; x += (uint32_t)acc[0]; }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

// dependent chain of lazy 29-bit Montgomery products (the product's own core)
__global__ void __launch_bounds__(256, 3) k_l29_chain(uint32_t* out, uint32_t seed, int iters) {
    typedef L29<Bn254Fq> L;
    L a, m;
    for (int k = 0; k < 9; k++) { a.l[k] = (int32_t)((threadIdx.x * 2654435761u + k * 40503u + seed) & 0x0fffffffu); m.l[k] = (int32_t)((seed * 7 + k * 977u) & 0x0fffffffu); }
    for (int it = 0; it < iters; it++) a = L::mul(a, m);
    uint32_t s = 0; for (int k = 0; k < 9; k++) s ^= (uint32_t)a.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same product, product scanning with hand-placed multiply-adds: the carry of column k-1 is the addend of the first
// multiply-add of column k (the compiler re-associates the sum and spends a 64-bit addition on it)
__device__ __forceinline__ int64_t madv(int32_t a, int32_t b, int64_t c) { int64_t d; uint64_t cy; asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ int64_t mads(int32_t a, int32_t b, int64_t c) { int64_t d; uint64_t cy; asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(b), "v"(c)); return d; }
template <class F>
__device__ __forceinline__ L29<F> mul_cols(const L29<F>& a, const L29<F>& b) {
    typedef L29<F> L; typedef typename F::Params P;
    int32_t m[9];
    int64_t T = 0;
    _Pragma("unroll") for (int k = 0; k < 9; k++) {
        _Pragma("unroll") for (int i = 0; i <= k; i++) T = madv(a.l[i], b.l[k - i], T);
        _Pragma("unroll") for (int i = 0; i < k; i++) T = mads(m[i], L::pl(k - i), T);
        m[k] = (int32_t)(((uint32_t)T * (P::INV & L::MASK)) & L::MASK);
        T = mads(m[k], L::pl(0), T);
        T >>= 29;
    }
    L r;
    _Pragma("unroll") for (int k = 9; k < 17; k++) {
        _Pragma("unroll") for (int i = k - 8; i < 9; i++) { T = madv(a.l[i], b.l[k - i], T); T = mads(m[i], L::pl(k - i), T); }
        r.l[k - 9] = (int32_t)((uint32_t)T & L::MASK);
        T >>= 29;
    }
    r.l[8] = (int32_t)T;
    return r;
}
__global__ void __launch_bounds__(256, 3) k_l29_chain_cols(uint32_t* out, uint32_t seed, int iters) {
    typedef L29<Bn254Fq> L;
    L a, m;
    for (int k = 0; k < 9; k++) { a.l[k] = (int32_t)((threadIdx.x * 2654435761u + k * 40503u + seed) & 0x0fffffffu); m.l[k] = (int32_t)((seed * 7 + k * 977u) & 0x0fffffffu); }
    for (int it = 0; it < iters; it++) a = mul_cols(a, m);
    uint32_t s = 0; for (int k = 0; k < 9; k++) s ^= (uint32_t)a.l[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 52-bit-limb floating-point product, instruction mix of one Montgomery multiplication: 25 limb pairs for a*b and 25 for q*p,
// each pair = hi = fma_rz(x, y, 2^104); lo = fma_rz(x, y, (2^104 + 2^52) - hi); two 64-bit integer additions of the raw bit
// patterns into column accumulators; per round one q = low 52 bits of (column * p') (2 fma + 1 add + bit fiddling) and one carry.
// The limb values fed back are re-derived from the columns so that the chain is dependent; the arithmetic VALUE is not a
// field product (no exponent bookkeeping, round-to-nearest instead of the round-toward-zero mode the real scheme sets once per
// kernel with s_setreg) — this measures issue cost only.
__device__ __forceinline__ void dpair(double x, double y, uint64_t& lo_col, uint64_t& hi_col) {
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    const double hi = __fma_rn(x, y, C1);
    const double lo = __fma_rn(x, y, C2 - hi);
    hi_col += (uint64_t)__double_as_longlong(hi);
    lo_col += (uint64_t)__double_as_longlong(lo);
}
__global__ void __launch_bounds__(256, 3) k_dfma_chain(uint32_t* out, uint32_t seed, int iters) {
    double a[5], m[5], p[5];
    for (int k = 0; k < 5; k++) { a[k] = (double)((threadIdx.x * 2654435761u + k * 40503u + seed) & 0x0fffffffu) * 1048576.0 + 3.0; m[k] = (double)((seed * 7 + k * 977u) & 0x0fffffffu) * 1048571.0 + 1.0; p[k] = (double)((seed * 13 + k * 31u) & 0x0fffffffu) * 1048573.0 + 5.0; }
    const double pinv = (double)((seed * 31 + 17) & 0x0fffffffu) * 1048575.0 + 7.0;
    const uint64_t M52 = (1ull << 52) - 1;
    for (int it = 0; it < iters; it++) {
        uint64_t col[11];
        _Pragma("unroll") for (int k = 0; k < 11; k++) col[k] = 0;
        _Pragma("unroll") for (int i = 0; i < 5; i++) {
            _Pragma("unroll") for (int j = 0; j < 5; j++) dpair(a[i], m[j], col[i + j], col[i + j + 1]);
            // q = low 52 bits of col[i] * p' as a double
            const double ci = __longlong_as_double((long long)((col[i] & M52) | 0x4330000000000000ull)) - 0x1p52;
            const double qh = __fma_rn(ci, pinv, 0x1p104);
            const double ql = __fma_rn(ci, pinv, (0x1p104 + 0x1p52) - qh);
            const double q = ql - 0x1p52;
            _Pragma("unroll") for (int j = 0; j < 5; j++) dpair(q, p[j], col[i + j], col[i + j + 1]);
            col[i + 1] += col[i] >> 52;
        }
        _Pragma("unroll") for (int k = 0; k < 5; k++) {
            a[k] = __longlong_as_double((long long)((col[5 + k] & M52) | 0x4330000000000000ull)) - 0x1p52;
            col[6 + k] += col[5 + k] >> 52;
        }
    }
    double s = 0; for (int k = 0; k < 5; k++) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(s);
}

// the bucket-accumulation inner operation on its own: mixed additions acc += (x, y) on lazy limbs with operands in registers
// (no gather, no bucket boundaries): the compute floor of k_msm_accumulate
template <class F, class Acc, int THREADS, int MINW>
__global__ void __launch_bounds__(THREADS, MINW) k_madd_chain(uint32_t* out, uint32_t seed, int iters) {
    extern __shared__ uint4 lds[];
    F x, y;
    uint32_t* xw = reinterpret_cast<uint32_t*>(&x); uint32_t* yw = reinterpret_cast<uint32_t*>(&y);
    // seed bit 31 set: full-width pseudo-random operands (what the MSM sees); clear: 28-bit words (fewer toggling multiplier bits,
    // the chip then holds a higher clock: the figures differ by the DVFS give-back, MI355X_MICROARCH.md)
    const uint32_t mask = (seed >> 31) ? 0xffffffffu : 0x0fffffffu;
    auto h = [](uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; };
    for (int i = 0; i < (int)(sizeof(F) / 4); i++) { xw[i] = h((blockIdx.x * THREADS + threadIdx.x) * 64u + i + seed) & mask; yw[i] = h((blockIdx.x * THREADS + threadIdx.x) * 64u + 32 + i + seed) & mask; }
    for (int i = 7; i < (int)(sizeof(F) / 4); i += 8) { xw[i] &= 0x1fffffffu; yw[i] &= 0x1fffffffu; }     // below p
    Acc acc; acc.init(reinterpret_cast<typename Acc::LdsT*>(lds), threadIdx.x, THREADS);
    for (int it = 0; it < iters; it++) {
        acc_madd(acc, x, y, (it & 1) != 0);
        _Pragma("unroll") for (int i = 0; i < (int)(sizeof(F) / 4); i += 2) { xw[i] = (xw[i] * 0x9e3779b1u + yw[i + 1]) & mask; yw[i] = (yw[i] ^ (xw[i] >> 3) ^ (xw[i] << 7)) & mask; }
    }
    typename BucketOf<F>::type b;
    acc_store<F>(acc, &b);
    const uint32_t* bw = reinterpret_cast<const uint32_t*>(&b);
    uint32_t s = 0; for (int i = 0; i < (int)(sizeof(b) / 4); i++) s ^= bw[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    uint32_t* d; unsigned long long* dc; CHK(hipMalloc(&d, (size_t)4096 * 256 * 4)); CHK(hipMalloc(&dc, 16));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("== v_mad_u64_u32, 2048 workgroups x 256 lanes, 8 independent chains per lane\n");
    for (int iters : {2048, 20000, 200000, 600000}) {
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mad_sustained, dim3(2048), dim3(256), 0, 0, d, dc, 3u, 5u, iters); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CHK(hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost));
            const double mads = 2048.0 * 256 * 8 * iters;
            printf("iters %7d rep %d: %8.3f ms  %7.2f Tmad/s  shader clock %.0f MHz (s_memtime ticks per 100 MHz wall tick x 100)\n", iters, rep, ms, mads / (ms * 1e-3) / 1e12, 100.0 * (double)h[0] / (double)h[1]);
        }
    }
    printf("== Montgomery product chains, 2048 workgroups x 256 lanes, 3 waves per SIMD\n");
    for (int which = 0; which < 3; which++) {
        const int iters = 4096;
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(k_l29_chain, dim3(2048), dim3(256), 0, 0, d, 11u, iters);
            else if (which == 2) hipLaunchKernelGGL(k_l29_chain_cols, dim3(2048), dim3(256), 0, 0, d, 11u, iters);
            else hipLaunchKernelGGL(k_dfma_chain, dim3(2048), dim3(256), 0, 0, d, 11u, iters);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-34s rep %d: %8.3f ms  %7.1f G products/s\n", which == 0 ? "lazy 29-bit integer (9x9 limbs)" : which == 2 ? "lazy 29-bit, product scanning (asm)" : "52-bit floating point (5x5 limbs)", rep, ms, 2048.0 * 256 * iters / (ms * 1e-3) / 1e9);
        }
    }
    printf("== mixed additions on lazy limbs, operands in registers (compute floor of k_msm_accumulate)\n");
    for (int which = 0; which < 6; which++) {
        const int iters = 128;
        const uint32_t seed = 11u | ((which & 1) ? 0x80000000u : 0u);
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0));
            if (which / 2 == 0) hipLaunchKernelGGL((k_madd_chain<Bn254Fq, RegAcc29<Bn254Fq>, 256, 3>), dim3(1660), dim3(256), 0, 0, d, seed, iters);
            else if (which / 2 == 1) hipLaunchKernelGGL((k_madd_chain<Bn254Fq, RegAcc29<Bn254Fq>, 256, 3>), dim3(1536 * 2), dim3(256), 0, 0, d, seed, iters);
            else hipLaunchKernelGGL((k_madd_chain<Fp2<Bn254Fq>, LdsAcc29<Fp2<Bn254Fq>>, 128, 1>), dim3(3320), dim3(128), 128 * 4 * sizeof(LazyOf<Fp2<Bn254Fq>>::type), 0, d, seed, iters);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            const double adds = (which / 2 == 1 ? 1536.0 * 2 * 256 : which / 2 == 0 ? 1660.0 * 256 : 3320.0 * 128) * iters;
            printf("%-42s %-10s rep %d: %8.3f ms  %6.2f G additions/s  (%.2f ms per 54.5 M)\n", which / 2 == 0 ? "G1, 1660 workgroups (= the 2^22 launch)" : which / 2 == 1 ? "G1, 3072 workgroups (4 full rounds)" : "G2, 3320 workgroups of 128",
                   (which & 1) ? "full-width" : "28-bit", rep, ms, adds / (ms * 1e-3) / 1e9, 54.5e6 / (adds / (ms * 1e-3)) * 1e3);
        }
    }
    printf("== sustained: 40 back-to-back launches of the G1 mixed-addition chain (1660 workgroups, full-width operands), ms per launch\n");
    {
        hipEvent_t ev[41]; for (int i = 0; i < 41; i++) CHK(hipEventCreate(&ev[i]));
        CHK(hipEventRecord(ev[0]));
        for (int i = 0; i < 40; i++) { hipLaunchKernelGGL((k_madd_chain<Bn254Fq, RegAcc29<Bn254Fq>, 256, 3>), dim3(1660), dim3(256), 0, 0, d, 0x8000000bu, 128); CHK(hipEventRecord(ev[i + 1])); }
        CHK(hipEventSynchronize(ev[40]));
        for (int i = 0; i < 40; i++) { float ms; CHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); printf("%.2f ", ms); }
        printf("\n");
    }
    return 0;
}
