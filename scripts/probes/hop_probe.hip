// What does ONE look-back hop cost inside a streaming kernel on MI355X, by how the descriptor is polled?
//
// Shape of the FASTQ emit kernel: one 256-thread workgroup per 16 KiB tile, ~27 KiB of LDS (6 workgroups per CU), every
// workgroup loads its tile (16 B per lane x 4), publishes an 8-byte {flag, timestamp} granule, waits for the granule(s) of
// its predecessor(s), then stores the tile.  Measured per polling method: kernel time, time from the own publish until the
// predecessors were seen (the hop), round trip of the last poll, polls per workgroup.
//
//   method 0  no hand-off at all (the streaming floor)
//   method 1  lane 0, relaxed agent-scope load (global_load sc1) of tile t-1
//   method 2  64 lanes, sc1 loads of tiles t-1 .. t-64 (a look-back window)
//   method 3  scalar load (s_dcache_inv + s_load_dwordx2 glc) of tile t-1
//   method 4  scalar loads of the 64-granule window (8 x s_load_dwordx16)
// each with the granules in ordinary device memory and in uncached device memory (hipDeviceMallocUncached).
//
//   hipcc --offload-arch=gfx950 -O3 -o hop_probe scripts/probes/hop_probe.hip && ./hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef uint32_t v16u __attribute__((ext_vector_type(16)));
constexpr int TILE = 16384, BLOCK = 256;
constexpr u64 FLAG = 1ull << 63;

struct __attribute__((packed, aligned(1))) U16B { uint32_t x, y, z, w; };

__device__ __forceinline__ u64 ld_sc1(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ u64 sload2(const u64* p) {
    u64 v;
    asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ v16u sload16(const void* p) {
    v16u v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

struct Stats { u64 wait_ticks, rtt_ticks, polls, n, max_wait; };

template <int METHOD>
__global__ __launch_bounds__(BLOCK) void k_probe(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, u64* desc, Stats* st,
                                                 int hops) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[TILE + 48];
    __shared__ uint8_t s_pad[10240];   // tables of the emit kernel: 6 workgroups per CU
    __shared__ u64 s_x;
    const int tid = threadIdx.x;
    const long long t = blockIdx.x;
    if (in == nullptr) s_pad[tid] = 1;
    const uint8_t* p = in + t * TILE + tid * 16;
    uint4 r[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) r[s] = *reinterpret_cast<const uint4*>(p + BLOCK * 16 * s);
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<uint4*>(s_tile + (tid + BLOCK * s) * 16) = r[s];
    __syncthreads();
    if (METHOD != 0 && tid < 64) {
        const int lane = tid;
        // hop 1: wait for the predecessors' granules; hop h > 1: wait for the granule the predecessor writes after ITS hop h-1
        for (int h = 0; h < hops; ++h) {
            u64* d = desc + (long long)h * gridDim.x;
            const u64 t_pub = wall_clock64();
            if (lane == 0) st_sc1(&d[t], FLAG | t_pub);
            u64 polls = 0, t0 = 0, t1 = 0;
            bool ok = t == 0;
            while (!ok) {
                t0 = wall_clock64();
                if (METHOD == 1) {
                    u64 v = lane == 0 ? ld_sc1(&d[t - 1]) : FLAG;
                    ok = __ballot(!(v & FLAG)) == 0;
                } else if (METHOD == 2) {
                    const long long q = t - 1 - lane;
                    u64 v = q >= 0 ? ld_sc1(&d[q]) : FLAG;
                    ok = __ballot(!(v & FLAG)) == 0;
                } else if (METHOD == 3) {
                    const u64 v = sload2(&d[t - 1]);
                    ok = (v & FLAG) != 0;
                } else {
                    const long long lo = t >= 64 ? t - 64 : 0;   // granules [lo, lo + 64): all must be flagged up to t-1
                    asm volatile("s_dcache_inv" ::: "memory");
                    bool all = true;
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const v16u v = sload16(d + lo + 8 * b);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (lo + 8 * b + e < t && !(v[2 * e + 1] & 0x80000000u)) all = false;
                    }
                    ok = all;
                }
                t1 = wall_clock64();
                ++polls;
                if (!ok) __builtin_amdgcn_s_sleep(2);
                if (polls > (1u << 20)) break;
            }
            if (lane == 0 && (t & 15) == 3 && t > 64) {
                const u64 w = t1 - t_pub;
                atomicAdd(&st[h].wait_ticks, w); atomicAdd(&st[h].rtt_ticks, t1 - t0); atomicAdd(&st[h].polls, polls); atomicAdd(&st[h].n, 1ull);
                atomicMax(&st[h].max_wait, w);
            }
        }
        if (lane == 0) s_x = 1;
    }
    __syncthreads();
    uint8_t* o = out + t * TILE + tid * 16 + 3;   // unaligned 16-byte stores like the emit kernel's
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint4 v = *reinterpret_cast<const uint4*>(s_tile + (tid + BLOCK * s) * 16);
        *reinterpret_cast<U16B*>(o + BLOCK * 16 * s) = U16B{v.x, v.y, v.z, v.w};
    }
}

template <int METHOD>
void run(const uint8_t* in, uint8_t* out, u64* desc, Stats* d_st, int nt, int hops, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    Stats hs[4];
    for (int r = 0; r < 4; ++r) {
        hipMemset(desc, 0, (size_t)nt * 8 * 4); hipMemset(d_st, 0, sizeof(Stats) * 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_probe<METHOD>, dim3(nt), dim3(BLOCK), 0, 0, in, out, desc, d_st, hops);
        hipEventRecord(b);
        if (hipEventSynchronize(b) != hipSuccess) { printf("%s: kernel failed\n", name); return; }
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) { best = ms; hipMemcpy(hs, d_st, sizeof(hs), hipMemcpyDeviceToHost); }
    }
    printf("%-46s %7.3f ms %6.0f GB/s", name, best, 2.0 * nt * TILE / best / 1e6);
    for (int h = 0; h < hops && METHOD; ++h)
        if (hs[h].n) printf(" | hop%d wait %5.2f us (max %6.1f) rtt %5.2f us polls %4.1f", h + 1, hs[h].wait_ticks * 0.01 / hs[h].n,
                            hs[h].max_wait * 0.01, hs[h].rtt_ticks * 0.01 / hs[h].n, (double)hs[h].polls / hs[h].n);
    printf("\n");
    fflush(stdout);
}

int main() {
    const int nt = 131072;   // 2 GiB in, 2 GiB out
    uint8_t *in, *out; u64 *desc, *desc_uc = nullptr; Stats* st;
    hipMalloc(&in, (size_t)nt * TILE + 64); hipMalloc(&out, (size_t)nt * TILE + 64);
    hipMemset(in, 7, (size_t)nt * TILE);
    hipMalloc(&desc, (size_t)nt * 8 * 4); hipMalloc(&st, sizeof(Stats) * 4);
    if (hipExtMallocWithFlags((void**)&desc_uc, (size_t)nt * 8 * 4, hipDeviceMallocUncached) != hipSuccess) { desc_uc = nullptr; (void)hipGetLastError(); }
    printf("wall_clock64 rate: %d kHz (ticks assumed 10 ns)\n", 100000);
    for (int hops = 1; hops <= 2; ++hops) {
        printf("---- %d hop(s) per workgroup\n", hops);
        run<0>(in, out, desc, st, nt, hops, "0 no hand-off");
        run<1>(in, out, desc, st, nt, hops, "1 vector sc1, predecessor");
        run<2>(in, out, desc, st, nt, hops, "2 vector sc1, 64-granule window");
        run<3>(in, out, desc, st, nt, hops, "3 scalar glc, predecessor");
        run<4>(in, out, desc, st, nt, hops, "4 scalar glc, 64-granule window");
        if (desc_uc) {
            run<1>(in, out, desc_uc, st, nt, hops, "1 vector sc1, predecessor      [uncached]");
            run<2>(in, out, desc_uc, st, nt, hops, "2 vector sc1, 64-granule window [uncached]");
            run<3>(in, out, desc_uc, st, nt, hops, "3 scalar glc, predecessor      [uncached]");
            run<4>(in, out, desc_uc, st, nt, hops, "4 scalar glc, 64-granule window [uncached]");
        } else printf("hipDeviceMallocUncached not available\n");
    }
    return 0;
}
