// bzq_lookback.hpp -- decoupled look-backs of the single-read variant k_fused<..., LB = true> (round 1; correct, slower than two
// passes on this part: profiles/r2_single_read.md).  EXPERIMENTS build only (make -C blazeseq_amd/csrc exp); included by
// bzq_fused.hpp under BZQ_EXPERIMENTS, and used by bzq_single.hpp.
#pragma once
// (included INSIDE namespace bzq, behind wave_sum)

constexpr u64 DESC_A = 1ull << 62;              // granule holds this tile's own aggregate
constexpr u64 DESC_P = 2ull << 62;              // granule holds the inclusive prefix through this tile
constexpr u64 DESC_VMASK = (1ull << 62) - 1ull;
constexpr int64_t DESC_BIAS = 1ll << 44;        // prefixes may be slightly negative (shard head)
constexpr int SPIN_LIMIT = 1 << 21;

__device__ __forceinline__ u64 ld_agent(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Exclusive line prefix of tile t (wave 0, all 64 lanes).  Lane i inspects predecessor t-1-i.
__device__ inline int64_t lookback_lines(const u64* desc_c, int64_t t, int64_t P0, int lane, ChunkState* st) {
    if (t == 0) return P0;
    int64_t running = 0, base = t - 1;
    int spins = 0;
    for (;;) {
        const int64_t p = base - lane;
        const u64 g = p >= 0 ? ld_agent(&desc_c[p]) : (DESC_P | (u64)(P0 + DESC_BIAS));
        const int flag = (int)(g >> 62);
        const u64 pm = __ballot(flag == 2), xm = __ballot(flag == 0);
        const int f = pm ? __builtin_ctzll(pm) : 64;
        const u64 nearer = f >= 64 ? ~0ull : ((1ull << f) - 1ull);
        if (xm & nearer) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { if (lane == 0) st->lookback_timeout = 1; return running; }
            continue;
        }
        int64_t v = (int64_t)(g & DESC_VMASK);
        if (flag == 2) v -= DESC_BIAS;
        running += wave_sum(lane <= f ? v : 0);
        if (f < 64) return running;
        base -= 64;
    }
}

struct Cols { int64_t s, q, d; };

__device__ inline Cols lookback_cols(const u64* desc_agg, const u64* desc_pre, int64_t t, int64_t S0, int64_t Q0,
                                     int64_t I0, int lane, ChunkState* st) {
    if (t == 0) return Cols{S0, Q0, I0};
    Cols run{0, 0, 0};
    int64_t base = t - 1;
    int spins = 0;
    for (;;) {
        const int64_t p = base - lane;
        int flag = 0;
        int64_t vs = 0, vq = 0, vd = 0;
        if (p < 0) { flag = 2; vs = S0; vq = Q0; vd = I0; }
        else {
            const u64 a = ld_agent(&desc_pre[3 * p]), b = ld_agent(&desc_pre[3 * p + 1]), c = ld_agent(&desc_pre[3 * p + 2]);
            if ((a >> 62) == 2 && (b >> 62) == 2 && (c >> 62) == 2) {
                flag = 2;
                vs = (int64_t)(a & DESC_VMASK) - DESC_BIAS;
                vq = (int64_t)(b & DESC_VMASK) - DESC_BIAS;
                vd = (int64_t)(c & DESC_VMASK) - DESC_BIAS;
            } else {
                const u64 ag = ld_agent(&desc_agg[p]);
                if ((ag >> 62) == 1) {
                    flag = 1;
                    vs = (int64_t)(ag & 0xFFFFFull); vq = (int64_t)((ag >> 20) & 0xFFFFFull); vd = (int64_t)((ag >> 40) & 0xFFFFFull);
                }
            }
        }
        const u64 pm = __ballot(flag == 2), xm = __ballot(flag == 0);
        const int f = pm ? __builtin_ctzll(pm) : 64;
        const u64 nearer = f >= 64 ? ~0ull : ((1ull << f) - 1ull);
        if (xm & nearer) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) { if (lane == 0) st->lookback_timeout = 1; return run; }
            continue;
        }
        const bool use = lane <= f;
        run.s += wave_sum(use ? vs : 0);
        run.q += wave_sum(use ? vq : 0);
        run.d += wave_sum(use ? vd : 0);
        if (f < 64) return run;
        base -= 64;
    }
}

