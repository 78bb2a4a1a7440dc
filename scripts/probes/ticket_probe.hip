// What does a global ticket (one atomicAdd per workgroup) cost on MI355X?  194 k workgroups, like the emit kernel.
//   hipcc --offload-arch=gfx950 -O3 -o ticket_probe scripts/probes/ticket_probe.hip && ./ticket_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

template <int MODE>   // 0 none, 1 one counter, 2 one counter per XCD, 3 64 counters by blockIdx
__global__ __launch_bounds__(256) void k(unsigned long long* ctr, unsigned long long* out) {
    __shared__ unsigned long long s_t;
    if (threadIdx.x == 0) {
        unsigned long long t = blockIdx.x;
        if (MODE == 1) t = atomicAdd(&ctr[0], 1ull);
        if (MODE == 2) { const unsigned x = xcc_id() & 7; t = atomicAdd(&ctr[x * 16], 1ull) * 8 + x; }
        if (MODE == 3) { const unsigned x = blockIdx.x & 63; t = atomicAdd(&ctr[x * 16], 1ull) * 64 + x; }
        s_t = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[s_t] = s_t + 1;
}

template <int MODE> float run(unsigned long long* ctr, unsigned long long* out, int n, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipMemset(ctr, 0, 64 * 16 * 8);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, ctr, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-34s %8.3f ms  (%.1f ns per workgroup)\n", name, best, best * 1e6 / n);
    return best;
}

int main() {
    const int n = 194092;
    unsigned long long *ctr, *out;
    hipMalloc(&ctr, 64 * 16 * 8); hipMalloc(&out, (size_t)n * 8 * 2);
    run<0>(ctr, out, n, "no ticket");
    run<1>(ctr, out, n, "one global ticket counter");
    run<2>(ctr, out, n, "one counter per XCD (XCC_ID)");
    run<3>(ctr, out, n, "64 counters by blockIdx");
    // check mode 2 produced a permutation
    std::vector<unsigned long long> h(n * 2);
    hipMemset(out, 0, (size_t)n * 16); hipMemset(ctr, 0, 64 * 16 * 8);
    hipLaunchKernelGGL(k<2>, dim3(n), dim3(256), 0, 0, ctr, out);
    hipMemcpy(h.data(), out, (size_t)n * 16, hipMemcpyDeviceToHost);
    long long maxi = 0, filled = 0;
    for (int i = 0; i < 2 * n; ++i) if (h[i]) { ++filled; maxi = i; }
    unsigned long long c[8]; for (int x = 0; x < 8; ++x) hipMemcpy(&c[x], ctr + x * 16, 8, hipMemcpyDeviceToHost);
    printf("per-XCD tickets: filled %lld of %d, max index %lld; counters", filled, n, maxi);
    for (int x = 0; x < 8; ++x) printf(" %llu", c[x]);
    printf("\n");
    return 0;
}
