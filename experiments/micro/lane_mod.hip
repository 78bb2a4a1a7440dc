// lane % d for d = 1..63 the way the symbol loops do it: lane - d * floor((lane + 0.5) * rcp(d))
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(uint32_t* out) {
    const uint32_t lane = threadIdx.x;
    const float lh = (float)lane + 0.5f;
    for (uint32_t d = 1; d < 64; ++d) {
        uint32_t vt, vsrc;
        const uint32_t sd = __builtin_amdgcn_readfirstlane(d);
        asm volatile("v_cvt_f32_u32 %0, %3\n\tv_rcp_f32 %0, %0\n\ts_nop 1\n\tv_mul_f32 %0, %2, %0\n\tv_cvt_u32_f32 %0, %0\n\tv_mul_lo_u32 %0, %0, %3\n\tv_sub_u32 %1, %4, %0"
                     : "=&v"(vt), "=&v"(vsrc) : "v"(lh), "s"(sd), "v"(lane));
        out[d * 64 + lane] = vsrc;
    }
}
int main() {
    uint32_t* o; (void)hipMalloc(&o, 64 * 64 * 4);
    k<<<1, 64>>>(o);
    uint32_t h[64 * 64]; (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int d = 1; d < 64; ++d) for (int i = 0; i < 64; ++i) if (h[d * 64 + i] != (uint32_t)(i % d)) { if (bad < 10) printf("d %d i %d: %u (want %d)\n", d, i, h[d * 64 + i], i % d); ++bad; }
    printf("bad = %d\n", bad);
    return 0;
}
