// bzq_chain.hpp -- "last event wins" states over bit masks, shared by the FASTA kernels (bzq_fasta.hpp) and the validating
// views kernels (bzq_views.hpp).
//
// A state that is set by some bytes, cleared by others and kept by the rest (e.g. "a byte >= 0x80 since the last '\n'")
// is, over the 64 bytes a thread owns as bit masks, ONE 64-bit add: with a = ~clear and b = set, a set bit generates a
// carry, a clear bit kills it, anything else propagates it, so the carry out of bit p is the state after byte p.  The
// same adder applied to the wave's ballots carries the state across lanes, four LDS words across the waves of a tile.
#pragma once
#include "bzq_device.hpp"

namespace bzq {
namespace fa {

// 16-bit mask from 0x80-per-byte flags
__device__ __forceinline__ uint32_t flag_mask16(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    uint32_t lo = __builtin_amdgcn_udot4(a >> 7, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(b >> 7, 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(c >> 7, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(d >> 7, 0x80402010u, hi, false);
    return (hi << 8) | lo;
}

// ---- "last event wins" states ---------------------------------------------------------------------------------------
struct Chain { u64 incl, excl; };   // state after / before each byte
__device__ __forceinline__ Chain chain64(u64 set, u64 clear, uint32_t cin) {
    const u64 a = ~clear, b = set;
    const u64 s1 = a + b;
    const uint32_t c1 = s1 < a;
    const u64 s = s1 + cin;
    const uint32_t c2 = s < s1;
    Chain r;
    r.excl = s ^ a ^ b;   // carry INTO each bit
    r.incl = (r.excl >> 1) | ((u64)(c1 | c2) << 63);
    return r;
}
// what a whole thread / wave does to the state: 0 clears, 1 sets, 2 leaves it
__device__ __forceinline__ uint32_t chain_code(u64 set, u64 clear) { return (set | clear) ? (set > clear ? 1u : 0u) : 2u; }

struct WaveChain { u64 S, C; };
// ballots of one chain + this wave's code into s_slot[wave]; a __syncthreads() must follow before chain_cin.
// REV: the chain runs from the last byte to the first, so a thread's (wave's) verdict is its FIRST event.
template <bool REV>
__device__ __forceinline__ WaveChain chain_wave(u64 set, u64 clear, uint32_t* s_slot) {
    const uint32_t code = REV ? chain_code(__builtin_bitreverse64(set), __builtin_bitreverse64(clear)) : chain_code(set, clear);
    WaveChain w;
    w.S = __ballot(code == 1u);
    w.C = __ballot(code == 0u);
    if ((threadIdx.x & 63) == 0)
        s_slot[threadIdx.x >> 6] = REV ? chain_code(__builtin_bitreverse64(w.S), __builtin_bitreverse64(w.C)) : chain_code(w.S, w.C);
    return w;
}
template <bool REV>
__device__ __forceinline__ uint32_t chain_cin(const WaveChain& w, const uint32_t* s_slot, uint32_t tile_cin) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // everything up to the per-lane bit pick is wave-uniform: keep it on the scalar unit
    uint32_t c = tile_cin;
    if (!REV) {
#pragma unroll
        for (int i = 0; i < BLOCK / 64; ++i) { const uint32_t v = __builtin_amdgcn_readfirstlane(s_slot[i]); if (i < wave && v != 2u) c = v; }
        c = __builtin_amdgcn_readfirstlane(c);
        return (uint32_t)(chain64(w.S, w.C, c).excl >> lane) & 1u;
    } else {
#pragma unroll
        for (int i = BLOCK / 64 - 1; i >= 0; --i) { const uint32_t v = __builtin_amdgcn_readfirstlane(s_slot[i]); if (i > wave && v != 2u) c = v; }
        c = __builtin_amdgcn_readfirstlane(c);
        return (uint32_t)(chain64(__builtin_bitreverse64(w.S), __builtin_bitreverse64(w.C), c).excl >> (63 - lane)) & 1u;
    }
}
__device__ __forceinline__ Chain chain64_rev(u64 set, u64 clear, uint32_t cin) {
    Chain r = chain64(__builtin_bitreverse64(set), __builtin_bitreverse64(clear), cin);
    r.incl = __builtin_bitreverse64(r.incl);
    r.excl = __builtin_bitreverse64(r.excl);
    return r;
}

} // namespace fa
} // namespace bzq
