// Instruction-throughput microbenchmarks that decide the big-integer strategy on gfx950 (run on the GPU box):
//   v_mad_u64_u32, v_mul_lo/hi_u32, 24-bit multiplies, v_fma_f64, carry chains, and the product's own Montgomery multiply.
// Output: one line per test, ops per clock per CU (assuming the reported clock) and chip-wide Gop/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../collaborative-circom_amd/csrc/curve.hpp"
using namespace cg;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

__global__ void k_mad64(uint32_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
    uint32_t x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) acc[i] = (uint64_t)x * (uint32_t)(y + i) + acc[i]; x += (uint32_t)acc[0]; }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
__global__ void k_mullo(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) acc[i] = acc[i] * (b + i); }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) acc[i] = __umulhi(acc[i] | 0x80000000u, b + i) + 1; }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul24(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) acc[i] = __umul24(acc[i], b + i) + 1; }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mulhi24(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { uint32_t r; asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(acc[i]), "v"(b + i)); acc[i] = r + 0x00ffff00u; } }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad24(uint32_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) { uint32_t r; asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(acc[i]), "v"(b + i), "v"(acc[(i + 1) & 7])); acc[i] = r; } }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(uint32_t* out, double a, double b) {
    double acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) acc[i] = __fma_rn(acc[i], a, b + i); }
    double s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(long long)s;
}
__global__ void k_fma32(uint32_t* out, float a, float b) {
    float acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) acc[i] = __fmaf_rn(acc[i], a, b + i); }
    float s = 0; for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(long long)s;
}
__global__ void k_addc(uint32_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;
    uint64_t inc = ((uint64_t)b << 32) | 0xfffffff1u;
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) acc[i] += inc + i; }
    uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <class F, int CHAINS>
__global__ void k_fpmul(uint32_t* out, uint32_t seed) {
    F acc[CHAINS], m;
    for (int c = 0; c < CHAINS; c++) for (int i = 0; i < F::N; i++) acc[c].v[i] = (threadIdx.x * 2654435761u + i * 40503u + c + seed) & 0x0fffffffu;
    for (int i = 0; i < F::N; i++) m.v[i] = (seed * 7 + i) & 0x0fffffffu;
    for (int it = 0; it < ITERS / 8; it++) { _Pragma("unroll") for (int c = 0; c < CHAINS; c++) acc[c] = acc[c] * m; }
    uint32_t s = 0; for (int c = 0; c < CHAINS; c++) for (int i = 0; i < F::N; i++) s ^= acc[c].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
__global__ void k_fpadd(uint32_t* out, uint32_t seed) {
    F acc[2], m;
    for (int c = 0; c < 2; c++) for (int i = 0; i < F::N; i++) acc[c].v[i] = (threadIdx.x * 2654435761u + i * 40503u + c + seed) & 0x0fffffffu;
    for (int i = 0; i < F::N; i++) m.v[i] = (seed * 7 + i) & 0x0fffffffu;
    for (int it = 0; it < ITERS; it++) { acc[0] = acc[0] + m; acc[1] = acc[1] - m; }
    uint32_t s = 0; for (int c = 0; c < 2; c++) for (int i = 0; i < F::N; i++) s ^= acc[c].v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
__global__ void __launch_bounds__(256) k_madd(uint32_t* out, uint32_t seed) {
    XYZZ<F> acc; F x, y;
    uint32_t* xw = reinterpret_cast<uint32_t*>(&x); uint32_t* yw = reinterpret_cast<uint32_t*>(&y);
    for (int i = 0; i < (int)(sizeof(F) / 4); i++) { xw[i] = (threadIdx.x * 2654435761u + i + seed) & 0x0fffffffu; yw[i] = (threadIdx.x * 40503u + i * 3 + seed) & 0x0fffffffu; }
    acc = {x, y, F::one(), F::one()};
    for (int it = 0; it < ITERS / 16; it++) { acc = xyzz_madd(acc, x, y); x = x + y; }
    const uint32_t* aw = reinterpret_cast<const uint32_t*>(&acc);
    uint32_t s = 0; for (int i = 0; i < (int)(sizeof(acc) / 4); i++) s ^= aw[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K, class... A>
int run(const char* name, double ops_per_thread, int blocks, int threads, K kern, A... args) {
    uint32_t* d; CHK(hipMalloc(&d, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, args...);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, args...); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double total = ops_per_thread * blocks * threads;
    double gops = total / (best * 1e-3) / 1e9;
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    double clk = p.clockRate * 1e3;   // Hz
    printf("%-28s blocks=%5d thr=%4d  %9.3f ms  %10.1f Gop/s  %7.2f op/clk/CU (lanes; @%.0f MHz, %d CUs)\n", name, blocks, threads, best, gops, gops * 1e9 / clk / p.multiProcessorCount, clk / 1e6, p.multiProcessorCount);
    CHK(hipFree(d));
    return 0;
}

int main() {
    const int B = 256 * 8, T = 256;
    const double n8 = 8.0 * ITERS;
#ifndef CG_MADD_ONLY
    run("v_mad_u64_u32", n8, B, T, k_mad64, 12345u, 777u);
    run("v_mul_lo_u32", n8, B, T, k_mullo, 12345u, 777u);
    run("v_mul_hi_u32(+add)", n8, B, T, k_mulhi, 12345u, 777u);
    run("v_mul_u32_u24(+add)", n8, B, T, k_mul24, 12345u, 777u);
    run("v_mul_hi_u32_u24(+add)", n8, B, T, k_mulhi24, 12345u, 777u);
    run("v_mad_u32_u24", n8, B, T, k_mad24, 12345u, 777u);
    run("v_fma_f64", n8, B, T, k_fma64, 1.0000001, 0.5);
    run("v_fma_f32", n8, B, T, k_fma32, 1.0000001f, 0.5f);
    run("u64 add (add_co+addc)", n8, B, T, k_addc, 12345u, 777u);
    for (int blocks : {256, 256 * 2, 256 * 4, 256 * 8}) {
        run("Fq254 mont_mul x1 chain", ITERS / 8, blocks, T, k_fpmul<Bn254Fq, 1>, 1u);
        run("Fq254 mont_mul x2 chains", 2.0 * (ITERS / 8), blocks, T, k_fpmul<Bn254Fq, 2>, 1u);
    }
    run("Fq254 mont_mul x4 chains", 4.0 * (ITERS / 8), B, T, k_fpmul<Bn254Fq, 4>, 1u);
    run("Fq381 mont_mul x1 chain", ITERS / 8, B, T, k_fpmul<Bls381Fq, 1>, 1u);
    run("Fq254 add+sub", 2.0 * ITERS, B, T, k_fpadd<Bn254Fq>, 1u);
#endif
    run("G1 bn254 xyzz_madd", ITERS / 16, B, T, k_madd<Bn254Fq>, 1u);
    run("G1 bn254 xyzz_madd (occ/2)", ITERS / 16, 256 * 2, T, k_madd<Bn254Fq>, 1u);
    run("G2 bn254 xyzz_madd", ITERS / 16, B, T, k_madd<Fp2<Bn254Fq>>, 1u);
    run("G1 bls381 xyzz_madd", ITERS / 16, B, T, k_madd<Bls381Fq>, 1u);
    return 0;
}
