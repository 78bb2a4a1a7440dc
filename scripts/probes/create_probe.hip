// Where bzq_create's ~100 ms of a fresh process go: the calls it makes, timed one by one in a process of their own.
//   hipcc --offload-arch=gfx950 -O2 -o create_probe create_probe.hip && ./create_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop() {}
int main() {
    double t = now(), t0 = t;
    auto lap = [&](const char* what) { const double n = now(); printf("%-44s %8.2f ms\n", what, n - t); t = n; };
    int nd = 0; (void)hipGetDeviceCount(&nd); lap("hipGetDeviceCount (runtime start-up)");
    (void)hipSetDevice(0); lap("hipSetDevice");
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0); lap("hipGetDeviceProperties");
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); lap("hipDeviceGetStreamPriorityRange");
    hipStream_t s1, s2; (void)hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi); lap("hipStreamCreateWithPriority (first stream)");
    (void)hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi); lap("hipStreamCreateWithPriority (second)");
    void* d = nullptr; (void)hipMalloc(&d, 664); lap("hipMalloc(664)");
    void* d2 = nullptr; (void)hipMalloc(&d2, 16); lap("hipMalloc(16)");
    (void)hipMemsetAsync(d2, 0, 16, s1); lap("hipMemsetAsync");
    void* h = nullptr; (void)hipHostMalloc(&h, 664, hipHostMallocDefault); lap("hipHostMalloc(664)");
    hipEvent_t ev[9]; for (auto& e : ev) (void)hipEventCreate(&e); lap("9 x hipEventCreate");
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s1); lap("first kernel launch (code object load)");
    (void)hipStreamSynchronize(s1); lap("hipStreamSynchronize");
    void* big = nullptr; (void)hipMalloc(&big, 288u << 20); lap("hipMalloc(288 MiB)");
    void* big2 = nullptr; (void)hipMalloc(&big2, 288u << 20); lap("hipMalloc(288 MiB) again");
    printf("%-44s %8.2f ms\n", "total", now() - t0);
    return 0;
}
