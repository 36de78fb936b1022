// What a co-running kernel costs the bucket accumulation, by the co-runner's SHAPE: the real 2^22-point G1 accumulation (through the C ABI)
// is timed while a synthetic kernel sits on a high-priority stream — W waves in workgroups of T lanes, V allocated VGPRs, either busy
// (a dependent multiply-add chain, like the bucket reduction) or idle (s_sleep), for about D ms.  DESIGN.md §5 quotes the table.
// build: hipcc -O3 --offload-arch=gfx950 -I include scripts/microbench_corun.hip -L collaborative-circom_amd -lcogroth16_hip -Wl,-rpath,$PWD/collaborative-circom_amd -o scripts/_build/microbench_corun
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "cogroth16_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define CG(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s: %s\n", #x, cg_last_error()); exit(1); } } while (0)

__global__ void k_fill(uint64_t* p, size_t n64, uint64_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n64; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull ^ seed; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = (i & 3) == 3 ? (x >> 4) : x;          // top limb < 2^60: below the BN254 scalar modulus
    }
}
// V: highest VGPR touched (the allocation follows it); BUSY: dependent 64-bit multiply-adds, else sleep
template <int V, bool BUSY>
__global__ void k_corun(unsigned long long ticks, unsigned long long* sink) {
    if (V >= 190) asm volatile("v_mov_b32 v190, 0" ::: "v190");
    else if (V >= 160) asm volatile("v_mov_b32 v160, 0" ::: "v160");
    else if (V >= 96) asm volatile("v_mov_b32 v96, 0" ::: "v96");
    const unsigned long long t0 = wall_clock64();
    unsigned long long a = threadIdx.x + 1, b = 0x9E3779B97F4A7C15ull;
    while (wall_clock64() - t0 < ticks) {
        if (BUSY) { _Pragma("unroll") for (int i = 0; i < 64; i++) a = a * b + (a >> 7); }
        else __builtin_amdgcn_s_sleep(64);
    }
    if (a == 12345) *sink = a;
}

int main(int argc, char** argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 22;
    const size_t n = (size_t)1 << log_n;
    cg_ctx* ctx; CG(cg_ctx_create(0, &ctx));
    uint64_t *d_base, *d_sc; unsigned long long* d_sink;
    CK(hipMalloc(&d_base, n * 32)); CK(hipMalloc(&d_sc, n * 32)); CK(hipMalloc(&d_sink, 8));
    k_fill<<<2048, 256>>>(d_base, n * 4, 1); k_fill<<<2048, 256>>>(d_sc, n * 4, 2); CK(hipDeviceSynchronize());
    cg_bases* bases; CG(cg_bases_from_scalars(ctx, CG_BN254, CG_G1, d_base, n, &bases));
    CG(cg_bases_precompute(ctx, bases, 0));
    int prio_lo, prio_hi; CK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    hipStream_t hs; CK(hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, prio_hi));
    CG(cg_stats_enable(ctx, 1));
    std::vector<uint64_t> out(12 * 2);
    auto msm = [&](int reps) -> double {
        cg_stage_times st; CG(cg_stats(ctx, &st, 1));
        for (int r = 0; r < reps; r++) { const void* sc[1] = {d_sc}; int tk; CG(cg_msm_dev_begin(ctx, bases, 0, n, sc, 1, &tk)); CG(cg_msm_end(ctx, tk, out.data())); }
        CG(cg_stats(ctx, &st, 1));
        return st.msm_acc_g1_ms / reps;
    };
    msm(2);
    printf("accumulation alone: %.3f ms\n", msm(5));
    const double clk_ghz = 0.1;                            // s_memtime / readcyclecounter ticks at 100 MHz
    auto run = [&](const char* name, auto kern, int waves, int threads, double ms) {
        const unsigned long long ticks = (unsigned long long)(ms * 1e6 * clk_ghz);
        double acc = 0; const int reps = 12;
        for (int r = 0; r < reps; r++) {
            // the co-runner is on the chip first (nothing else running), then the MSM starts
            hipLaunchKernelGGL(kern, dim3(waves * 64 / threads), dim3(threads), 0, hs, ticks, d_sink);
            acc += msm(1);
            CK(hipStreamSynchronize(hs));
        }
        printf("%-44s waves %5d  wg %4d  ~%4.1f ms : accumulation %.3f ms\n", name, waves, threads, ms, acc / reps);
    };
    for (double ms : {2.0, 6.0}) {
        run("idle,  8 VGPRs", k_corun<8, false>, 512, 64, ms);
        run("idle, 96 VGPRs", k_corun<96, false>, 512, 64, ms);
        run("idle, 160 VGPRs", k_corun<160, false>, 512, 64, ms);
        run("idle, 190 VGPRs", k_corun<190, false>, 512, 64, ms);
        run("idle, 190 VGPRs, workgroups of 256", k_corun<190, false>, 512, 256, ms);
        run("idle, 160 VGPRs, workgroups of 256", k_corun<160, false>, 512, 256, ms);
        run("busy,  8 VGPRs", k_corun<8, true>, 512, 64, ms);
        run("busy, 190 VGPRs", k_corun<190, true>, 512, 64, ms);
        run("busy, 190 VGPRs, workgroups of 256", k_corun<190, true>, 512, 256, ms);
        run("busy, 190 VGPRs, 128 waves", k_corun<190, true>, 128, 64, ms);
        run("busy, 190 VGPRs, 2048 waves", k_corun<190, true>, 2048, 64, ms);
    }
    return 0;
}
