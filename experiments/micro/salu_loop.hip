// What does one trip of a scalar decode loop cost on a CDNA4 CU?  (one wave, one workgroup: no contention)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int MODE>
__global__ void k(uint32_t* out, int n) {
    __shared__ uint32_t lut[1024];   // (4 KiB)
    for (int i = threadIdx.x; i < 1024; i += 64) lut[i] = (uint32_t)(i * 2654435761u) >> 22;   // next index
    __syncthreads();
    uint32_t vt = 0, lit = 0;
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lut;
    uint32_t acc = 0;
    if (MODE == 0)
        asm volatile("s_mov_b32 s40, %1\n1:\n\ts_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n\ts_mov_b32 %0, s40" : "=s"(acc) : "s"(n) : "s40", "scc");
    if (MODE == 1)   // + 10 dependent SALU
        asm volatile("s_mov_b32 s40, %1\n\ts_mov_b32 s41, 0\n1:\n\ts_add_i32 s41, s41, 1\n\ts_add_i32 s41, s41, 1\n\ts_add_i32 s41, s41, 1\n\ts_add_i32 s41, s41, 1\n\ts_add_i32 s41, s41, 1\n\t"
                     "s_add_i32 s41, s41, 1\n\ts_add_i32 s41, s41, 1\n\ts_add_i32 s41, s41, 1\n\ts_add_i32 s41, s41, 1\n\ts_add_i32 s41, s41, 1\n\t"
                     "s_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n\ts_mov_b32 %0, s41" : "=s"(acc) : "s"(n) : "s40", "s41", "scc");
    if (MODE == 2)   // the LDS round trip: index -> v_mov -> ds_read -> readfirstlane -> index
        asm volatile("s_mov_b32 s40, %2\n\ts_mov_b32 s41, 5\n1:\n\ts_and_b32 s42, s41, 0x3ff\n\ts_lshl2_add_u32 s42, s42, %3\n\tv_mov_b32 %1, s42\n\tds_read_b32 %1, %1\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 s41, %1\n\t"
                     "s_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n\ts_mov_b32 %0, s41" : "=s"(acc), "+v"(vt) : "s"(n), "s"(lds) : "s40", "s41", "s42", "scc", "memory");
    if (MODE == 3)   // two v_writelane through m0 + a 64-bit shift
        asm volatile("s_mov_b32 s40, %2\n\ts_mov_b32 s41, 5\n\ts_mov_b64 s[44:45], -1\n1:\n\ts_and_b32 s42, s40, 31\n\ts_mov_b32 m0, s42\n\tv_writelane_b32 %1, s41, m0\n\ts_add_i32 m0, s42, 1\n\tv_writelane_b32 %1, s40, m0\n\t"
                     "s_lshr_b64 s[44:45], s[44:45], 1\n\t"
                     "s_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n\ts_mov_b32 %0, s44" : "=s"(acc), "+v"(lit) : "s"(n), "s"(lds) : "s40", "s41", "s42", "s44", "s45", "m0", "scc");
    if (MODE == 4)   // a not-taken and a taken forward branch per trip
        asm volatile("s_mov_b32 s40, %1\n1:\n\ts_cmp_eq_u32 s40, 0x7fffffff\n\ts_cbranch_scc1 3f\n\ts_cmp_lg_u32 s40, 0x7ffffffe\n\ts_cbranch_scc1 2f\n\ts_nop 0\n\ts_nop 0\n2:\n\t"
                     "s_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n3:\n\ts_mov_b32 %0, s40" : "=s"(acc) : "s"(n) : "s40", "scc");
    if (MODE == 5)   // v_readlane with a scalar lane select feeding SALU
        asm volatile("s_mov_b32 s40, %2\n\ts_mov_b32 s41, 5\n1:\n\ts_and_b32 s42, s41, 63\n\tv_readlane_b32 s41, %1, s42\n\ts_add_i32 s41, s41, s40\n\t"
                     "s_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b\n\ts_mov_b32 %0, s41" : "=s"(acc), "+v"(vt) : "s"(n) : "s40", "s41", "s42", "scc");
    if (MODE == 6) {  // 10 dependent 32-bit VALU shifts
        uint32_t a = threadIdx.x + 77;
        asm volatile("s_mov_b32 s40, %1\n1:\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\t"
                     "v_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\tv_lshrrev_b32 %0, 1, %0\n\t"
                     "s_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b" : "+v"(a) : "s"(n) : "s40", "scc");
        vt = a;
    }
    if (MODE == 7) {  // 10 dependent 64-bit VALU shifts
        unsigned long long a = threadIdx.x + 77;
        asm volatile("s_mov_b32 s40, %1\n1:\n\tv_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %0, 1, %0\n\t"
                     "v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %0, 1, %0\n\t"
                     "s_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b" : "+v"(a) : "s"(n) : "s40", "scc");
        vt = (uint32_t)a;
    }
    if (MODE == 8) {  // 10 independent ds_read_u8 at scattered addresses + wait
        uint32_t a0 = (threadIdx.x * 97u) & 1023u, r0 = 0, r1 = 0;
        asm volatile("s_mov_b32 s40, %3\n1:\n\tds_read_u8 %1, %0\n\tds_read_u8 %2, %0 offset:1024\n\tds_read_u8 %1, %0 offset:2048\n\tds_read_u8 %2, %0 offset:3000\n\tds_read_u8 %1, %0 offset:77\n\t"
                     "ds_read_u8 %2, %0 offset:1111\n\tds_read_u8 %1, %0 offset:2222\n\tds_read_u8 %2, %0 offset:333\n\tds_read_u8 %1, %0 offset:444\n\tds_read_u8 %2, %0 offset:555\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "s_sub_i32 s40, s40, 1\n\ts_cmp_lg_u32 s40, 0\n\ts_cbranch_scc1 1b" : "+v"(a0), "+v"(r0), "+v"(r1) : "s"(n) : "s40", "scc", "memory");
        vt = r0 + r1;
    }
    if (threadIdx.x == 0) out[0] = acc + vt + lit;
}
template <int MODE> void run(uint32_t* o, const char* what, int extra) {
    const int n = 2000000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<1, 64>>>(o, 1000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<1, 64>>>(o, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %7.1f ns per trip (%d instructions)\n", what, ms * 1e6 / n, extra);
}
int main() {
    uint32_t* o; (void)hipMalloc(&o, 64);
    run<0>(o, "s_sub, s_cmp, taken s_cbranch", 3);
    run<1>(o, "+ 10 dependent s_add", 13);
    run<2>(o, "+ s_and, s_lshl2_add, v_mov, ds_read, waitcnt, readfirstlane", 9);
    run<3>(o, "+ s_and, 2 x (m0, v_writelane), s_lshr_b64", 9);
    run<4>(o, "+ cmp + not-taken branch, cmp + taken forward branch", 7);
    run<5>(o, "+ s_and, v_readlane (scalar select), s_add", 6);
    run<6>(o, "+ 10 dependent v_lshrrev_b32", 13);
    run<7>(o, "+ 10 dependent v_lshrrev_b64", 13);
    run<8>(o, "+ 10 independent ds_read_u8 (scattered) + wait", 14);
    return 0;
}
