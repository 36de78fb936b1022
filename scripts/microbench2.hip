// Per-instruction issue rates on gfx950 for the ops around the 29-bit Montgomery core (inline asm so nothing is folded away).
// 8 independent dependency chains per lane; output: wave-instruction issue cost in cycles per SIMD at the measured wall time
// (@2400 MHz nominal), i.e. 4.0 = full rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

#define KERNEL64(NAME, ASM)                                                                                     \
    __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                                               \
        uint64_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = ((uint64_t)(threadIdx.x + i) << 32) | (a + i);    \
        uint32_t x = a + threadIdx.x, y = b | 1;                                                                \
        for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(acc[i]) : "v"(x), "v"(y)); } \
        uint64_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                         \
    }
#define KERNEL32(NAME, ASM)                                                                                     \
    __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                                               \
        uint32_t acc[8]; for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i + a;                              \
        uint32_t x = a + threadIdx.x, y = b | 1;                                                                \
        for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(acc[i]) : "v"(x), "v"(y)); } \
        uint32_t s = 0; for (int i = 0; i < 8; i++) s ^= acc[i];                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                         \
    }

KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %1, %2, %0")
KERNEL64(k_ashr_i64, "v_ashrrev_i64 %0, 29, %0")
KERNEL64(k_lshr_b64, "v_lshrrev_b64 %0, 29, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0")
KERNEL64(k_mov_b64, "v_mov_b64 %0, %0")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %2")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %2")
KERNEL32(k_and_b32, "v_and_b32 %0, %0, %2")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %2")
KERNEL32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %2")
KERNEL32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 29")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mov_b32, "v_mov_b32 %0, %0")
KERNEL32(k_addc, "v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %0, vcc, %0, %1, vcc")

template <class K>
int run(const char* name, double insts_per_iter, K kern) {
    const int B = 256 * 8, T = 256;
    uint64_t* d; CHK(hipMalloc(&d, (size_t)B * T * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(B), dim3(T), 0, 0, d, 12345u, 777u); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(B), dim3(T), 0, 0, d, 12345u, 777u); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_insts = (double)B * T / 64.0 * ITERS * 8.0 * insts_per_iter;       // per chip
    const double per_simd = wave_insts / (256.0 * 4.0);
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-18s %8.3f ms   %6.2f cycles per wave-instruction per SIMD (4.00 = full rate)\n", name, ms, cycles / per_simd);
    CHK(hipFree(d));
    return 0;
}
int main() {
    run("v_mad_u64_u32", 1, k_mad_u64_u32); run("v_mad_i64_i32", 1, k_mad_i64_i32); run("v_ashrrev_i64", 1, k_ashr_i64); run("v_lshrrev_b64", 1, k_lshr_b64);
    run("v_lshl_add_u64", 1, k_lshl_add_u64); run("v_mov_b64", 1, k_mov_b64); run("v_mul_lo_u32", 1, k_mul_lo_u32); run("v_mul_hi_u32", 1, k_mul_hi_u32);
    run("v_and_b32", 1, k_and_b32); run("v_alignbit_b32", 1, k_alignbit); run("v_add_u32", 1, k_add_u32); run("v_add3_u32", 1, k_add3_u32);
    run("v_lshl_add_u32", 1, k_lshl_add_u32); run("v_bfe_u32", 1, k_bfe_u32); run("v_mad_u32_u24", 1, k_mad_u32_u24); run("v_mov_b32", 1, k_mov_b32);
    run("v_add_co+addc", 2, k_addc);
    return 0;
}
