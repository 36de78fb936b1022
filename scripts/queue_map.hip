// Which pairs of HIP streams wait for each other?  Four streams per priority class (low / normal / high), a 120 us spin kernel on each stream of a pair,
// started together: ~120 us = side by side, ~240 us = one behind the other (a shared hardware queue, or whatever else serialises them).
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/_build/queue_map scripts/queue_map.hip ; run on the GPU box.  Output: the matrix in microseconds.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } }
int main(int argc, char** argv) {
    const int per = argc > 1 ? atoi(argv[1]) : 4;
    int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);             // lo = least priority (largest number)
    printf("priority range: least %d greatest %d\n", lo, hi);
    std::vector<hipStream_t> st; std::vector<int> cls;
    for (int c = 0; c < 3; c++) for (int i = 0; i < per; i++) {
        hipStream_t s; const int prio = c == 0 ? lo : c == 2 ? hi : (lo + hi) / 2;
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio) != hipSuccess) { printf("stream creation failed\n"); return 1; }
        st.push_back(s); cls.push_back(c - 1);
    }
    for (auto s : st) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000ull); hipStreamSynchronize(s); }   // warm
    const int n = (int)st.size();
    printf("rows / columns: class (-1 low, 0 normal, 1 high) and creation index within the class\n      ");
    for (int j = 0; j < n; j++) printf("%3d:%d ", cls[j], j % per);
    printf("\n");
    for (int i = 0; i < n; i++) {
        printf("%3d:%d ", cls[i], i % per);
        for (int j = 0; j < n; j++) {
            if (j <= i) { printf("    . "); continue; }
            double best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                hipStreamSynchronize(st[i]); hipStreamSynchronize(st[j]);
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], 12000ull);
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[j], 12000ull);
                hipStreamSynchronize(st[i]); hipStreamSynchronize(st[j]);
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (us < best) best = us;
            }
            printf("%5.0f ", best);
        }
        printf("\n");
    }
    return 0;
}
