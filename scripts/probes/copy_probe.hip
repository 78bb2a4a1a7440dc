// How fast can a tile-shaped copy go on MI355X, by load / store flavour?  (the emit kernel moves 3.3 GB in + 3.56 GB out per
// launch and runs at the rate of variant "plain / unaligned+3" below)
//   hipcc --offload-arch=gfx950 -O3 -o copy_probe scripts/probes/copy_probe.hip && ./copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int TILE = 16384, BLOCK = 256;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U16B { uint32_t x, y, z, w; };

template <int LD, int ST, int SHIFT, int LDS_KB, int XCD = 0>
__global__ __launch_bounds__(BLOCK) void k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[LDS_KB * 1024];
    const int tid = threadIdx.x;
    // XCD = 1: workgroup b runs on XCD b % 8; every XCD takes one contiguous eighth of the tiles (the product's xcd_tile())
    const unsigned q8 = gridDim.x >> 3, r8 = gridDim.x & 7u, x8 = blockIdx.x & 7u;
    const long long t = XCD ? (long long)x8 * q8 + (x8 < r8 ? x8 : r8) + (blockIdx.x >> 3) : (long long)blockIdx.x;
    const uint8_t* p = in + t * TILE + tid * 16;
    uint4 r[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint4* q = reinterpret_cast<const uint4*>(p + BLOCK * 16 * s);
        if (LD == 0) r[s] = *q;
        else { const v4u w = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(q)); r[s] = make_uint4(w.x, w.y, w.z, w.w); }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<uint4*>(s_tile + (tid + BLOCK * s) * 16) = r[s];
    __syncthreads();
    uint8_t* o = out + t * TILE + tid * 16 + SHIFT;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint4 v = *reinterpret_cast<const uint4*>(s_tile + (tid + BLOCK * s) * 16);
        if (SHIFT == 0) {
            uint4* d = reinterpret_cast<uint4*>(o + BLOCK * 16 * s);
            if (ST == 0) *d = v; else { v4u w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<v4u*>(d)); }
        } else {
            if (ST == 0) *reinterpret_cast<U16B*>(o + BLOCK * 16 * s) = U16B{v.x, v.y, v.z, v.w};
            else { v4u w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(o + BLOCK * 16 * s), "v"(w) : "memory"); }
        }
    }
}
template <int LD, int ST, int SHIFT, int LDS_KB, int XCD = 0> void run(const uint8_t* in, uint8_t* out, int nt, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<LD, ST, SHIFT, LDS_KB, XCD>), dim3(nt), dim3(BLOCK), 0, 0, in, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-58s %7.3f ms  %6.0f GB/s read+write\n", name, best, 2.0 * nt * TILE / best / 1e6);
}
int main() {
    const int nt = 196608;   // 3 GiB in, 3 GiB out
    uint8_t *in, *out;
    hipMalloc(&in, (size_t)nt * TILE + 64); hipMalloc(&out, (size_t)nt * TILE + 64);
    hipMemset(in, 5, (size_t)nt * TILE);
    run<0, 0, 0, 27>(in, out, nt, "plain loads, plain aligned stores, 27 KiB LDS (6 WG/CU)");
    run<0, 0, 3, 27>(in, out, nt, "plain loads, plain stores at +3 (unaligned), 6 WG/CU");
    run<0, 1, 0, 27>(in, out, nt, "plain loads, nt aligned stores, 6 WG/CU");
    run<0, 1, 3, 27>(in, out, nt, "plain loads, nt stores at +3, 6 WG/CU");
    run<1, 0, 0, 27>(in, out, nt, "nt loads, plain aligned stores, 6 WG/CU");
    run<1, 1, 0, 27>(in, out, nt, "nt loads, nt aligned stores, 6 WG/CU");
    run<1, 1, 3, 27>(in, out, nt, "nt loads, nt stores at +3, 6 WG/CU");
    run<0, 0, 3, 20>(in, out, nt, "plain / unaligned, 20 KiB LDS (8 WG/CU)");
    run<1, 1, 3, 20>(in, out, nt, "nt / nt unaligned, 8 WG/CU");
    run<0, 0, 3, 40>(in, out, nt, "plain / unaligned, 40 KiB LDS (4 WG/CU)");
    printf("---- the same with one contiguous eighth of the tiles per XCD\n");
    run<0, 0, 0, 27, 1>(in, out, nt, "plain loads, plain aligned stores, 6 WG/CU   [XCD-contiguous]");
    run<0, 0, 3, 27, 1>(in, out, nt, "plain loads, plain stores at +3, 6 WG/CU     [XCD-contiguous]");
    run<1, 0, 3, 27, 1>(in, out, nt, "nt loads, plain stores at +3, 6 WG/CU        [XCD-contiguous]");
    run<1, 1, 3, 27, 1>(in, out, nt, "nt loads, nt stores at +3, 6 WG/CU           [XCD-contiguous]");
    run<0, 0, 3, 20, 1>(in, out, nt, "plain / unaligned, 8 WG/CU                   [XCD-contiguous]");
    return 0;
}
