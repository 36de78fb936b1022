// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS repository's access patterns (MI355X_MICROARCH.md §HBM: the
// counters are exact only up to a pattern-dependent factor — FETCH_SIZE reports half the bytes of a wide coalesced stream — and ask for
// a calibration on a known byte count).  Three kernels with known traffic, far beyond the 256 MiB Infinity Cache:
//   k_cal_stream   coalesced 16 B/lane copy of 4 GiB                      (reads 4 GiB, writes 4 GiB)
//   k_cal_gather64 random 64-byte records, 4 x dwordx4 per lane from a 4 GiB table, 2^26 records; 4-byte result per lane
//                  (reads 2^26 x 64 B = 4 GiB of records + 256 MiB of indices, writes 256 MiB) — the bucket accumulation's gather
//   k_cal_gather128 the same with 128-byte records (G2), 2^25 records
// build: hipcc -O3 --offload-arch=gfx950 scripts/pmc_calibrate.hip -o scripts/_build/pmc_calibrate ; run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_cal_stream(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
template <int QUADS>
__global__ void __launch_bounds__(256) k_cal_gather(const uint4* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4* r = table + (size_t)idx[i] * QUADS;
        uint32_t s = 0;
        _Pragma("unroll") for (int q = 0; q < QUADS; q++) { uint4 v = r[q]; s ^= v.x ^ v.y ^ v.z ^ v.w; }
        out[i] = s;
    }
}
__global__ void k_cal_fill_idx(uint32_t* idx, size_t n, uint32_t mask) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v = (uint32_t)i * 2654435761u; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 13; idx[i] = v & mask;
    }
}
int main() {
    const size_t table_bytes = (size_t)4 << 30;
    uint4 *a, *b; uint32_t *idx, *out;
    CHK(hipMalloc(&a, table_bytes)); CHK(hipMalloc(&b, table_bytes)); CHK(hipMalloc(&idx, (size_t)256 << 20)); CHK(hipMalloc(&out, (size_t)256 << 20));
    CHK(hipMemset(a, 1, table_bytes));
    const size_t n64 = (size_t)1 << 26, n128 = (size_t)1 << 25;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_cal_stream, dim3(8192), dim3(256), 0, 0, a, b, table_bytes / 16);
        hipLaunchKernelGGL(k_cal_fill_idx, dim3(4096), dim3(256), 0, 0, idx, n64, (uint32_t)(table_bytes / 64 - 1));
        hipLaunchKernelGGL((k_cal_gather<4>), dim3(8192), dim3(256), 0, 0, a, idx, out, n64);
        hipLaunchKernelGGL(k_cal_fill_idx, dim3(4096), dim3(256), 0, 0, idx, n128, (uint32_t)(table_bytes / 128 - 1));
        hipLaunchKernelGGL((k_cal_gather<8>), dim3(8192), dim3(256), 0, 0, a, idx, out, n128);
        CHK(hipDeviceSynchronize());
    }
    printf("known bytes per launch: k_cal_stream read %zu write %zu; k_cal_gather<4> read %zu (+%zu idx) write %zu; k_cal_gather<8> read %zu (+%zu idx) write %zu\n",
           table_bytes, table_bytes, n64 * 64, n64 * 4, n64 * 4, n128 * 128, n128 * 4, n128 * 4);
    return 0;
}
