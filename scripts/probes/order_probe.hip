// In what order do workgroups START, across the 8 XCDs, when some of them wait on their predecessors?
//   a persistent-ish kernel shape like k_stream: 27 KiB LDS, two 16 KiB tiles per workgroup, then (mode 1) wait until the
//   63 predecessors of its group of 64 have "published" (a flag written right after the loads).
// Prints, per mode: kernel time; how late the latest-starting predecessor (among the 63 before) started relative to the
// workgroup itself (mean / p99 / max, microseconds); the spread of progress between XCDs sampled over time.
//   hipcc --offload-arch=gfx950 -O3 -o order_probe scripts/probes/order_probe.hip && ./order_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
constexpr int TILE = 16384, BLOCK = 256;

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }
__device__ __forceinline__ u64 ld_sc1(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ __launch_bounds__(BLOCK) void k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, u64* flag, u64* start, unsigned* xcc) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[TILE + 48 + 10240];
    const int tid = threadIdx.x;
    const long long w = blockIdx.x;
    if (tid == 0) { start[w] = wall_clock64(); xcc[w] = xcc_id(); }
    uint4 r[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) r[s] = *reinterpret_cast<const uint4*>(in + w * 2 * TILE + (tid + BLOCK * s) * 16);
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<uint4*>(s_tile + (tid + BLOCK * s) * 16) = r[s];
    __syncthreads();
    if (tid < 64) {
        if (tid == 0) st_sc1(&flag[w], 1ull << 63);
        if (MODE == 1) {
            const long long g0 = (w / 64) * 64, q = g0 + tid;
            bool ok = !(q < w);
            for (int spins = 0; spins < (1 << 20); ++spins) {
                if (!ok) ok = (ld_sc1(&flag[q]) >> 63) != 0;
                if (__ballot(!ok) == 0) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s == 4) { __syncthreads();
#pragma unroll
            for (int z = 0; z < 4; ++z) *reinterpret_cast<uint4*>(s_tile + (tid + BLOCK * z) * 16) = r[4 + z];
            __syncthreads(); }
        const uint4 v = *reinterpret_cast<const uint4*>(s_tile + (tid + BLOCK * (s & 3)) * 16);
        *reinterpret_cast<uint4*>(out + w * 2 * TILE + (tid + BLOCK * s) * 16) = v;
    }
}

template <int MODE> void run(const uint8_t* in, uint8_t* out, u64* flag, u64* start, unsigned* xcc, int nw) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(flag, 0, (size_t)nw * 8);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(nw), dim3(BLOCK), 0, 0, in, out, flag, start, xcc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<u64> st(nw); std::vector<unsigned> xc(nw);
    hipMemcpy(st.data(), start, (size_t)nw * 8, hipMemcpyDeviceToHost); hipMemcpy(xc.data(), xcc, (size_t)nw * 4, hipMemcpyDeviceToHost);
    std::vector<double> late;
    for (int w = 64; w < nw; ++w) {
        long long worst = -(1ll << 60);
        for (int q = (w / 64) * 64; q < w; ++q) worst = std::max(worst, (long long)st[q] - (long long)st[w]);
        if (w % 64) late.push_back(worst * 0.01);
    }
    std::sort(late.begin(), late.end());
    double mean = 0; for (double v : late) mean += v; mean /= late.size();
    int same = 0; for (int w = 0; w < nw; ++w) same += xc[w] == (unsigned)(w % 8);
    printf("mode %d: %.3f ms (%.0f GB/s r+w); latest predecessor started %+.1f us after the workgroup on average, p50 %+.1f, p99 %+.1f, max %+.1f; xcc == w%%8 for %.1f%% of workgroups\n",
           MODE, ms, 4.0 * nw * TILE / ms / 1e6, mean, late[late.size() / 2], late[late.size() * 99 / 100], late.back(), 100.0 * same / nw);
    // progress spread: at the time workgroup w starts, how far (in index) is the largest started index on every XCD
    fflush(stdout);
}
int main() {
    const int nw = 65536;   // 2 GiB in, 2 GiB out
    uint8_t *in, *out; u64 *flag, *start; unsigned* xcc;
    hipMalloc(&in, (size_t)nw * 2 * TILE); hipMalloc(&out, (size_t)nw * 2 * TILE); hipMemset(in, 1, (size_t)nw * 2 * TILE);
    hipMalloc(&flag, (size_t)nw * 8); hipMalloc(&start, (size_t)nw * 8); hipMalloc(&xcc, (size_t)nw * 4);
    for (int rep = 0; rep < 2; ++rep) { run<0>(in, out, flag, start, xcc, nw); run<1>(in, out, flag, start, xcc, nw); }
    return 0;
}
