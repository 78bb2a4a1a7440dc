// What does global_load_lds_ushort write, and where?  (experiment behind the gzip decoder's LDS output buffer)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(const uint16_t* src, uint32_t* out, int base_bytes, int len, int stride) {
    __shared__ uint16_t buf[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) buf[i] = 0xEEEE;
    __syncthreads();
    const uint16_t* p = src + 100 + lane * stride;
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)buf + (uint32_t)base_bytes;
    if (lane < len) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_ushort %0, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "s"(lds) : "memory", "m0");
    }
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = ((const uint32_t*)buf)[i];
}
int main() {
    uint16_t h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (uint16_t)(0x1000 + i);
    uint16_t* d; uint32_t* o; hipMalloc(&d, sizeof h); hipMalloc(&o, 1024); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int base : {0, 2, 6, 64}) for (int len : {64, 5}) {
        k<<<1, 64>>>(d, o, base, len, 1);
        uint32_t r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("base %d len %d:", base, len);
        const uint16_t* u = (const uint16_t*)r;
        for (int i = 0; i < 80; ++i) printf(" %04x", u[i]);
        printf("\n");
    }
    return 0;
}
