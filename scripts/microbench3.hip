// Do the non-multiply VOP3 instructions of the Montgomery core (64-bit shifts / adds, v_mul_lo_u32, v_and) overlap with v_mad_u64_u32
// when interleaved, or do they cost their own issue slots?  8 independent mad chains per lane; variants add k other ops per 8 mads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

#define MAD(i) "v_mad_u64_u32 %" #i ", vcc, %16, %17, %" #i "\n"
#define MADS8 MAD(0) MAD(1) MAD(2) MAD(3) MAD(4) MAD(5) MAD(6) MAD(7)
#define KERNEL(NAME, EXTRA)                                                                                                  \
    __global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                                                            \
        uint64_t c[8]; for (int i = 0; i < 8; i++) c[i] = ((uint64_t)(threadIdx.x + i) << 32) | (a + i);                     \
        uint64_t e[4]; uint32_t f[4]; for (int i = 0; i < 4; i++) { e[i] = c[i] * 3 + 1; f[i] = a * (i + 3) + threadIdx.x; } \
        uint32_t x = a + threadIdx.x, y = b | 1;                                                                             \
        for (int it = 0; it < ITERS; it++)                                                                                   \
            asm volatile(MADS8 EXTRA : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), \
                                       "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) \
                         : "v"(x), "v"(y) : "vcc");                                                                          \
        uint64_t s = e[0] ^ e[1] ^ e[2] ^ e[3] ^ f[0] ^ f[1] ^ f[2] ^ f[3]; for (int i = 0; i < 8; i++) s ^= c[i];           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                      \
    }
KERNEL(k_mads, "")
KERNEL(k_mads_ashr4, "v_ashrrev_i64 %8, 29, %8\n v_ashrrev_i64 %9, 29, %9\n v_ashrrev_i64 %10, 29, %10\n v_ashrrev_i64 %11, 29, %11\n")
KERNEL(k_mads_add4, "v_lshl_add_u64 %8, %8, 0, %9\n v_lshl_add_u64 %9, %9, 0, %10\n v_lshl_add_u64 %10, %10, 0, %11\n v_lshl_add_u64 %11, %11, 0, %8\n")
KERNEL(k_mads_and4, "v_and_b32 %12, 0x1fffffff, %12\n v_and_b32 %13, 0x1fffffff, %13\n v_and_b32 %14, 0x1fffffff, %14\n v_and_b32 %15, 0x1fffffff, %15\n")
KERNEL(k_mads_mullo4, "v_mul_lo_u32 %12, %12, %17\n v_mul_lo_u32 %13, %13, %17\n v_mul_lo_u32 %14, %14, %17\n v_mul_lo_u32 %15, %15, %17\n")

template <class K>
int run(const char* name, K kern, double base_ms, double* out_ms) {
    const int B = 256 * 8, T = 256;
    uint64_t* d; CHK(hipMalloc(&d, (size_t)B * T * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(B), dim3(T), 0, 0, d, 12345u, 777u); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(B), dim3(T), 0, 0, d, 12345u, 777u); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd_iters = (double)B * T / 64.0 * ITERS / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / per_simd_iters;
    printf("%-16s %7.3f ms  %6.1f nominal cycles per iteration (8 mads%s)", name, ms, cyc, base_ms > 0 ? " + 4 extra" : "");
    if (base_ms > 0) printf("  -> %.2f cycles per extra op", (ms - base_ms) * 1e-3 * 2.4e9 / per_simd_iters / 4.0);
    printf("\n");
    if (out_ms) *out_ms = ms;
    CHK(hipFree(d));
    return 0;
}
int main() {
    double base = 0;
    run("8 mads", k_mads, 0, &base);
    run("+4 ashr_i64", k_mads_ashr4, base, nullptr);
    run("+4 lshl_add_u64", k_mads_add4, base, nullptr);
    run("+4 v_and_b32", k_mads_and4, base, nullptr);
    run("+4 v_mul_lo_u32", k_mads_mullo4, base, nullptr);
    return 0;
}
