// What a timing event costs between two kernels of one stream: plain launches, hipEventRecord between them, and
// hipExtLaunchKernelGGL with start / stop events (timestamps of the dispatch itself, no packet of their own).
// build: hipcc --offload-arch=gfx950 -O2 -o event_cost_probe event_cost_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k_small(int* p, int spin) {
    int x = 0;
    for (int i = 0; i < spin; ++i) x += __builtin_amdgcn_s_memtime() & 1;
    if (x == -1) p[0] = x;
}
int main() {
    int* d; hipMalloc(&d, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int NK = 6, IT = 300, SPIN = 2000;
    hipEvent_t ev[2 * NK + 2];
    for (auto& e : ev) hipEventCreate(&e);
    auto run = [&](int mode) {
        for (int warm = 0; warm < 2; ++warm) {
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < IT; ++it) {
                for (int k = 0; k < NK; ++k) {
                    if (mode == 1) hipEventRecord(ev[k], s);
                    if (mode == 2) hipExtLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, ev[2 * k], ev[2 * k + 1], 0, d, SPIN);
                    else hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, d, SPIN);
                }
                if (mode == 1) hipEventRecord(ev[NK], s);
                hipStreamSynchronize(s);
            }
            auto t1 = std::chrono::steady_clock::now();
            if (warm) {
                float ms = 0, ms2 = 0;
                if (mode == 1) hipEventElapsedTime(&ms, ev[0], ev[NK]);
                if (mode == 2) { hipEventElapsedTime(&ms, ev[0], ev[2 * NK - 1]); hipEventElapsedTime(&ms2, ev[2], ev[3]); }
                printf("mode %d: %.1f us per iteration of %d kernels; events say %.1f us span, %.1f us one kernel\n", mode,
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / IT, NK, ms * 1e3, ms2 * 1e3);
            }
        }
    };
    run(0); run(1); run(2); run(0);
    return 0;
}
