// bzq_inflate_ms.hpp -- BGZF blocks inflated on the GPU, EIGHT blocks per wave (round 4).
// An OPTION (inflate_ms = 1), not the default: correct on every test of tests/test_gpu_bgzf_inflate.py, and 2.7 x SLOWER than
// k_bgzf_inflate as compiled C++ -- profiles/r4_inflate_ms.md says why (instruction budget, compiler-placed vmcnt(0), occupancy).
//
// k_bgzf_inflate (bzq_inflate.hpp) gives a block a whole wave and runs the serial decode as UNIFORM code: 63 lanes idle in the
// symbol loop, and the CU's one scalar unit, shared by all its waves, is what bounds it (11 scalar instructions per literal
// lookup, ~50 per match; PMC: 1.03 scalar instructions per CU cycle).  Here a wave decodes MS_S = 8 blocks at once, MS_G = 8 lanes
// per block: the decoder state of a block (bit buffer, positions) lives in VECTOR registers, replicated in the eight lanes of its
// group, so one vector instruction advances eight bit buffers; the scalar unit only runs the loop control.  The eight lanes of a
// group are used where a block offers width: a match of up to 8 bytes is ONE load and one store per lane, two literals of one table
// entry leave through lanes 0 and 1.
//   * tables per block in LDS: a 9-bit literal / length table with up to two literals per entry and base + extra bits folded into
//     the length entries (build_lut2's format, 2 KiB), an 8-bit distance table of 16-bit one-symbol entries (0.5 KiB), and the two
//     canonical codes themselves (limit / first code / symbol offset per length, the sorted symbols: 0.8 KiB) for the codes the
//     tables' index does not reach: those are decoded INSIDE the loop, lane g of the group testing length 10 + g (9 + g for a
//     distance) -- one compare, one ballot.  (A first version handed them to the whole wave as uniform code: 1 % of the symbols and
//     1.6 % of the distances of the benchmark's FASTQ have such codes, and those 370 services per block were two thirds of its time);
//   * block headers (stored / fixed / dynamic: build_code and the table builders of bzq_inflate.hpp) are SERVED by the whole wave as
//     uniform code, one block at a time; the vector loop runs until some block reaches the end of a DEFLATE block;
//   * output goes straight to the chunk buffer; a match reads it back (same wave, in-order vector memory pipeline: a load issued
//     behind a store of the same wave sees it -- what k_bgzf_inflate's C++ paths rely on as well);
//   * ISIZE, every distance and every length are checked while decoding and the CRC-32 of every block afterwards (block_crc32,
//     the wave takes its eight blocks in turn).  A block that fails any of it fails the whole call.
#pragma once
#include "bzq_inflate.hpp"

namespace bzq {
namespace inf {

constexpr int MS_S = 8, MS_G = 8;          // blocks per wave, lanes per block
constexpr int MS_LBITS = 9, MS_DBITS = 8;  // index bits of the two tables
constexpr int MS_SCRATCH_WORDS = 256;      // per block: lim / first / offs of both codes (6 x 16 u32) + 320 sorted symbols (u16)

struct ArgsMs { const uint8_t* comp; uint64_t comp_bytes; const DevBlock* blocks; int64_t n_blocks; uint8_t* out; unsigned long long* first_bad; uint32_t* scratch; unsigned long long* stats; };   // stats: debug counters (BZQ_MS_STATS=1), else nullptr

enum { MS_DECODE = 0, MS_NEED_HDR = 1, MS_EOB = 2, MS_DONE = 5, MS_FAIL = 6 };

// uniform bit reader at an absolute bit position of a byte stream (the service paths); decode_sym only touches buf / cnt
struct UBits {
    const uint8_t* base;
    int64_t limit;   // bytes that may be read behind base; bits past them read as zero
    int64_t pos;     // next byte to enter buf
    u64 buf;
    int cnt;
    __device__ __forceinline__ void refill() {
        while (cnt <= 32) {
            uint32_t w = 0;
            if (pos + 4 <= limit) w = reinterpret_cast<const U32U*>(base + pos)->v;
            else for (int i = 0; i < 4; ++i) if (pos + i < limit) w |= (uint32_t)base[pos + i] << (8 * i);
            buf |= (u64)uni(w) << cnt;
            cnt += 32; pos += 4;
        }
        cnt = (int)uni((uint32_t)cnt);
        buf = ((u64)uni((uint32_t)(buf >> 32)) << 32) | uni((uint32_t)buf);
    }
    __device__ __forceinline__ void start(const uint8_t* p, int64_t lim, int64_t bitpos) {
        base = p; limit = lim; pos = bitpos >> 3; buf = 0; cnt = 0;
        refill();
        take((int)(bitpos & 7));
    }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = (uint32_t)buf & ((1u << n) - 1u); buf >>= n; cnt -= n; return v; }
    __device__ __forceinline__ int64_t bitpos() const { return pos * 8 - cnt; }
};

// build_lut2 (bzq_inflate.hpp) with the index width as a parameter
template <int BITS>
__device__ __forceinline__ void build_lut2_t(const Code& code, const uint16_t* symtab, const uint8_t* lens, uint32_t* lut2) {
    const int lane = threadIdx.x & 63;
    uint16_t* lut1 = reinterpret_cast<uint16_t*>(lut2) + (1 << BITS);
    build_lut<BITS>(code, symtab, lens, lut1);
    uint32_t ent[(1 << BITS) / 64];
#pragma unroll
    for (int k = 0; k < (1 << BITS) / 64; ++k) {
        const uint32_t i = (uint32_t)(k * 64 + lane);
        const uint32_t e1 = lut1[i];
        if (e1 & 0x100u) {
            const uint32_t sym = e1 & 0xFFFu;
            if (e1 != LUT_LONG && sym >= 257u && sym <= 285u) {   // RFC 1951 3.2.5
                const uint32_t c = sym - 257u, ext = (c < 8u || c >= 28u) ? 0u : (c - 4u) >> 2;
                const uint32_t base = c < 8u ? 3u + c : (c >= 28u ? 258u : 3u + ((4u + (c & 3u)) << ext));
                ent[k] = 0xC0000000u | (e1 >> 12) | (base << 5) | (ext << 16) | (((e1 >> 12) + ext) << 23);
            } else ent[k] = 0x80000000u | e1;
            continue;
        }
        const uint32_t l1 = e1 >> 12;
        const uint32_t e2 = lut1[i >> l1];
        const uint32_t l2 = e2 >> 12;
        const bool two = !(e2 & 0x100u) && l1 + l2 <= (uint32_t)BITS;
        ent[k] = (e1 & 0xFFu) | (l1 << 26) | (two ? ((e2 & 0xFFu) << 8) | (2u << 24) | ((l1 + l2) << 16) : (1u << 24) | (l1 << 16));   // (bits 26..29: the first literal's code length, for a block's last byte)
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < (1 << BITS) / 64; ++k) lut2[k * 64 + lane] = ent[k];
    __builtin_amdgcn_wave_barrier();
}

// out[pos + i] = out[pos - dist + i], i < len, by the whole wave (service paths); with dist < len the source repeats with period dist
__device__ __forceinline__ void ms_copy64(uint8_t* out, int pos, int len, int dist) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < len; i += 64) out[pos + i] = out[pos - dist + (dist >= len ? i : i % dist)];
}

static __global__ __launch_bounds__(64) void k_bgzf_inflate_ms(ArgsMs a) {
    __shared__ uint32_t s_lut[MS_S][1 << MS_LBITS];
    __shared__ uint16_t s_dlut[MS_S][1 << MS_DBITS];
    // the canonical codes of every block, lane-L values of Code as 16-bit words: [0..15] limit, [16..31] first code, [32..47] symbol
    // offset of the literal / length code, [48..95] the same of the distance code; and the sorted symbols of both (288 + 32)
    __shared__ uint16_t s_canon[MS_S][96];
    __shared__ uint16_t s_symtab[MS_S][288 + 32];
    __shared__ uint16_t s_sym[288 + 32];                        // sorted symbols of the block being served
    __shared__ __attribute__((aligned(4))) uint8_t s_lens[352]; // its code lengths
    __shared__ uint32_t s_crc_tab[256], s_x2n[32];
    {   // (crc_tables is written for a 256-thread workgroup: one wave fills the table in four trips)
        const int t0 = threadIdx.x;
        for (int t = t0; t < 256; t += 64) {
            uint32_t c = (uint32_t)t;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
            s_crc_tab[t] = c;
        }
        if (t0 == 0) {
            uint32_t p = 0x40000000u;
            for (int k = 0; k < 32; ++k) { s_x2n[k] = p; p = crc_mul(p, p); }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, grp = lane >> 3, g = lane & 7;
    const int64_t bi = (int64_t)xcd_tile() * MS_S + grp;   // (neighbouring blocks on one XCD)
    const bool have = bi < a.n_blocks;
    DevBlock blk{0, 0, 0, 0, 0, 0};
    if (have) blk = a.blocks[bi];
    const uint8_t* comp = a.comp + blk.coff;
    uint8_t* out = a.out + blk.uoff;
    // (said to the compiler as GLOBAL pointers: through generic ones every access of the loop becomes a FLAT instruction, which counts
    // against the LDS counter as well -- every table lookup then waits for the stores in flight)
    typedef __attribute__((address_space(1))) uint8_t g_u8;
    typedef __attribute__((address_space(1), aligned(1))) uint32_t g_u32u;
    const g_u8* const gcomp = (const g_u8*)(unsigned long long)(uintptr_t)comp;
    g_u8* const gout = (g_u8*)(unsigned long long)(uintptr_t)out;
    const int usize = (int)blk.usize;
    int64_t lim64 = (int64_t)(a.comp_bytes - blk.coff);
    if (lim64 > (int64_t)blk.csize + 8) lim64 = (int64_t)blk.csize + 8;   // (see inflate_block: bits behind the payload read as zero)
    const uint32_t limit = have ? (uint32_t)lim64 : 0u;

    // ---- the vector reader of a block: vbuf holds vcnt valid bits, voff = payload offset of the next dword to enter it, nw0 / nw1 =
    // the dwords at voff and voff + 4, fetched ahead
    u64 vbuf = 0;
    int vcnt = 0;
    uint32_t voff = 0, nw0 = 0, nw1 = 0;
    auto load32 = [&](uint32_t off) -> uint32_t {
        return off + 4u <= limit ? *(const g_u32u*)(gcomp + off) : 0u;   // (the last bytes of a payload: the 8 bytes of slack are the trailer)
    };
    auto vrefill = [&](bool on) {
        if (on && vcnt <= 32) {
            vbuf |= (u64)nw0 << vcnt; vcnt += 32; voff += 4u; nw0 = nw1;
            nw1 = load32(voff + 4u);
        }
    };
    auto vseek = [&](bool mine, int64_t bitpos) {   // the lanes of one group: continue at this bit of the payload
        if (mine) {
            voff = (uint32_t)(bitpos >> 3);
            vbuf = (u64)load32(voff); vcnt = 32;
            voff += 4u;
            nw0 = load32(voff); nw1 = load32(voff + 4u);
            const int sk = (int)(bitpos & 7);
            vbuf >>= sk; vcnt -= sk;
        }
    };
    int state = have ? MS_NEED_HDR : MS_DONE;
    int pos = 0;
    bool last = false;

    u64 tk_service = 0, tk_loop = 0;
    for (;;) {
        const u64 tk0 = a.stats ? wall_clock64() : 0;
        // ================================================================ service: one block at a time, the whole wave, uniform code
        for (;;) {
            const u64 need = __ballot(g == 0 && (state == MS_NEED_HDR || state == MS_EOB));
            if (!need) break;
            const int sl = __builtin_ctzll(need);          // leader lane of the block to serve
            const int s = sl >> 3;
            const bool mine = grp == s;
            const int st = (int)rdlane((uint32_t)state, sl);
            if (a.stats && lane == 0) atomicAdd(&a.stats[st], 1ull);
            int spos = (int)rdlane((uint32_t)pos, sl);
            const int su = (int)rdlane((uint32_t)usize, sl);
            bool slast = rdlane(last ? 1u : 0u, sl) != 0;
            const int64_t sbit = (int64_t)rdlane(voff, sl) * 8 - (int64_t)(int)rdlane((uint32_t)vcnt, sl);   // bits consumed so far
            const uint32_t c_lo = rdlane((uint32_t)(uintptr_t)comp, sl), c_hi = rdlane((uint32_t)((uintptr_t)comp >> 32), sl);
            const uint32_t o_lo = rdlane((uint32_t)(uintptr_t)out, sl), o_hi = rdlane((uint32_t)((uintptr_t)out >> 32), sl);
            const uint8_t* scomp = reinterpret_cast<const uint8_t*>(((uintptr_t)c_hi << 32) | c_lo);
            uint8_t* sout = reinterpret_cast<uint8_t*>(((uintptr_t)o_hi << 32) | o_lo);
            const int64_t slim = (int64_t)rdlane(limit, sl);
            const int scs = (int)rdlane(blk.csize, sl);
            uint32_t* lut = s_lut[s];
            uint16_t* dlut = s_dlut[s];
            UBits ub;
            ub.start(scomp, slim, sbit);
            int nst = MS_FAIL;
            if (st == MS_EOB && slast) nst = spos == su ? MS_DONE : MS_FAIL;
            else {   // block headers until one with a code (stored blocks are copied right here)
                for (;;) {
                    ub.refill();
                    slast = ub.take(1) != 0;
                    const uint32_t type = ub.take(2);
                    if (type == 3) break;
                    if (type == 0) {   // stored: to the next byte edge, LEN, ~LEN, LEN bytes
                        ub.take(ub.cnt & 7);
                        ub.refill();
                        const uint32_t len = ub.take(16), nlen = ub.take(16);
                        if ((len ^ nlen) != 0xFFFFu || spos + (int)len > su) break;
                        const int64_t src = ub.bitpos() >> 3;
                        if (src + (int64_t)len > (int64_t)scs) break;   // the bytes must lie inside this block's payload
                        for (int i = lane; i < (int)len; i += 64) sout[spos + i] = scomp[src + i];
                        spos += (int)len;
                        ub.start(scomp, slim, (src + (int64_t)len) * 8);
                        if (slast) { nst = spos == su ? MS_DONE : MS_FAIL; break; }
                        continue;
                    }
                    Code ll, dd;
                    bool ok = true;
                    if (type == 1) {   // fixed code (RFC 1951 3.2.6)
                        for (int q = lane; q < 288; q += 64) s_lens[q] = q < 144 ? 8 : (q < 256 ? 9 : (q < 280 ? 7 : 8));
                        if (lane < 32) s_lens[288 + lane] = 5;
                        __builtin_amdgcn_wave_barrier();
                        ok = build_code(s_lens, 288, s_sym, ll) && build_code(s_lens + 288, 30, s_sym + 288, dd);
                        if (ok) { build_lut2_t<MS_LBITS>(ll, s_sym, s_lens, lut); build_lut<MS_DBITS>(dd, s_sym + 288, s_lens + 288, dlut); }
                    } else {           // dynamic code (3.2.7)
                        const int hlit = (int)ub.take(5) + 257, hdist = (int)ub.take(5) + 1, hclen = (int)ub.take(4) + 4;
                        ok = hlit <= 286 && hdist <= 30;
                        if (ok) {
                            if (lane < 19) s_lens[lane] = 0;
                            for (int i = 0; i < hclen; ++i) {
                                ub.refill();
                                const uint32_t v = ub.take(3);
                                if (lane == 0) s_lens[CL_ORDER[i]] = (uint8_t)v;
                            }
                            __builtin_amdgcn_wave_barrier();
                            Code cl;
                            ok = build_code(s_lens, 19, s_sym + 288, cl);
                            int n = 0;
                            uint32_t prev = 0;
                            const int total = hlit + hdist;
                            while (ok && n < total) {
                                ub.refill();
                                const int q = decode_sym(ub, cl, s_sym + 288);
                                if (q < 0) { ok = false; break; }
                                uint32_t val = 0; int rep = 1;
                                if (q < 16) { val = (uint32_t)q; prev = val; }
                                else if (q == 16) { if (n == 0) { ok = false; break; } val = prev; rep = 3 + (int)ub.take(2); }
                                else if (q == 17) { rep = 3 + (int)ub.take(3); prev = 0; }
                                else { rep = 11 + (int)ub.take(7); prev = 0; }
                                if (n + rep > total) { ok = false; break; }
                                for (int i = lane; i < rep; i += 64) s_lens[32 + n + i] = (uint8_t)val;
                                n += rep;
                            }
                            __builtin_amdgcn_wave_barrier();
                            ok = ok && s_lens[32 + 256] != 0;   // no end-of-block code
                            ok = ok && build_code(s_lens + 32, hlit, s_sym, ll) && build_code(s_lens + 32 + hlit, hdist, s_sym + 288, dd);
                            if (ok) { build_lut2_t<MS_LBITS>(ll, s_sym, s_lens + 32, lut); build_lut<MS_DBITS>(dd, s_sym + 288, s_lens + 32 + hlit, dlut); }
                        }
                    }
                    if (!ok || (type == 2 && (!code_valid(ll, false, false) || !code_valid(dd, true, false)))) break;   // (the fixed distance code has 30 of its 32 codes)
                    // the canonical codes, for the codes the tables' index does not reach
                    if (lane < 16) {
                        uint16_t* cn = s_canon[s];
                        cn[lane] = (uint16_t)ll.lim; cn[16 + lane] = (uint16_t)ll.first; cn[32 + lane] = (uint16_t)ll.offs;
                        cn[48 + lane] = (uint16_t)dd.lim; cn[64 + lane] = (uint16_t)dd.first; cn[80 + lane] = (uint16_t)dd.offs;
                    }
                    for (int i = lane; i < 320; i += 64) s_symtab[s][i] = s_sym[i];
                    __builtin_amdgcn_wave_barrier();
                    nst = MS_DECODE;
                    break;
                }
            }
            // back into the block's lanes
            const int64_t nbit = ub.bitpos();
            if (nst == MS_DECODE) vseek(mine, nbit);
            if (mine) { state = nst; pos = spos; last = slast; }
            __builtin_amdgcn_wave_barrier();
        }
        const u64 tk1 = a.stats ? wall_clock64() : 0;
        tk_service += tk1 - tk0;
        if (!__ballot(state == MS_DECODE)) break;

        // ================================================================ the symbol loop: eight blocks per instruction
        for (;;) {
            const bool act = state == MS_DECODE;
            if (a.stats && lane == 0) atomicAdd(&a.stats[0], 1ull);
            vrefill(act);
            uint32_t e = s_lut[grp][(uint32_t)vbuf & ((1u << MS_LBITS) - 1u)];
            // a code longer than the table's index (or none): lane g of the group tests length MS_LBITS + 1 + g against the canonical
            // code's limits; the entry the table would have held is made up on the spot
            const bool lng = act && (e >> 30) == 2u && (e & 0xFFFFu) == LUT_LONG;
            if (__ballot(lng)) {
                const uint32_t c15 = __builtin_bitreverse32((uint32_t)vbuf) >> 17;
                const int L = MS_LBITS + 1 + g;
                const uint16_t* cn = s_canon[grp];
                const u64 hm = __ballot(lng && L <= 15 && c15 < (uint32_t)cn[L <= 15 ? L : 15]);
                const uint32_t gm = (uint32_t)(hm >> (grp * 8)) & 0xFFu;
                if (lng) {
                    e = 0x80000000u | 0xFFFu;   // no code: an invalid symbol, refused below
                    if (gm) {
                        const int Lh = MS_LBITS + 1 + __builtin_ctz(gm);
                        const uint32_t idx = (uint32_t)cn[32 + Lh] + (c15 >> (15 - Lh)) - (uint32_t)cn[16 + Lh];
                        const uint32_t sym = idx < 288u ? s_symtab[grp][idx] : 0xFFFu;
                        if (sym < 256u) e = sym | (1u << 24) | ((uint32_t)Lh << 16) | ((uint32_t)Lh << 26);
                        else if (sym >= 257u && sym <= 285u) {
                            const uint32_t c = sym - 257u, ext = (c < 8u || c >= 28u) ? 0u : (c - 4u) >> 2;
                            const uint32_t base = c < 8u ? 3u + c : (c >= 28u ? 258u : 3u + ((4u + (c & 3u)) << ext));
                            e = 0xC0000000u | (uint32_t)Lh | (base << 5) | (ext << 16) | (((uint32_t)Lh + ext) << 23);
                        } else e = 0x80000000u | sym | ((uint32_t)Lh << 12);   // end of block; 286 / 287: refused below
                    }
                }
            }
            bool has_len = false;
            int len = 0;
            if (act) {
                if (!(e >> 31)) {                         // one or two literals: lanes 0 and 1 of the group store them
                    int n = (int)((e >> 24) & 3u), bits = (int)((e >> 16) & 31u);
                    if (pos + n > usize && n == 2) { n = 1; bits = (int)((e >> 26) & 15u); }   // the block's last byte: the first of the two only
                    if (pos + n > usize) state = MS_FAIL;
                    else {
                        if (g < n) gout[pos + g] = (uint8_t)(e >> (8 * g));
                        pos += n;
                        vbuf >>= bits; vcnt -= bits;
                    }
                } else if (e & 0x40000000u) {             // a length symbol, base and extra bits folded into the entry
                    const int cb = (int)(e & 31u), xb = (int)((e >> 16) & 7u);
                    len = (int)((e >> 5) & 0x1FFu) + (int)(((uint32_t)(vbuf >> cb)) & ((1u << xb) - 1u));
                    const int tb = (int)((e >> 23) & 31u);
                    vbuf >>= tb; vcnt -= tb;
                    has_len = true;
                } else {                                  // end of block -- or no symbol at all
                    const uint32_t l = (e >> 12) & 0xFu;
                    if ((e & 0xFFFu) == 256u && l > 0u) { vbuf >>= l; vcnt -= (int)l; state = MS_EOB; }
                    else state = MS_FAIL;
                }
            }
            if (__ballot(has_len)) {
                vrefill(has_len);
                uint32_t d = s_dlut[grp][(uint32_t)vbuf & ((1u << MS_DBITS) - 1u)];
                const bool dlng = has_len && d == LUT_LONG;
                if (__ballot(dlng)) {   // a distance code of more than MS_DBITS bits: the same search, lengths MS_DBITS + 1 + g
                    const uint32_t c15 = __builtin_bitreverse32((uint32_t)vbuf) >> 17;
                    const int L = MS_DBITS + 1 + g;
                    const uint16_t* cn = s_canon[grp] + 48;
                    const u64 hm = __ballot(dlng && L <= 15 && c15 < (uint32_t)cn[L <= 15 ? L : 15]);
                    const uint32_t gm = (uint32_t)(hm >> (grp * 8)) & 0xFFu;
                    if (dlng) {
                        d = 0xFFFu;   // no code
                        if (gm) {
                            const int Lh = MS_DBITS + 1 + __builtin_ctz(gm);
                            const uint32_t idx = (uint32_t)cn[32 + Lh] + (c15 >> (15 - Lh)) - (uint32_t)cn[16 + Lh];
                            d = (idx < 32u ? (uint32_t)s_symtab[grp][288 + idx] : 0xFFFu) | ((uint32_t)Lh << 12);
                        }
                    }
                }
                int dist = 0;
                bool cp = false;
                if (has_len) {
                    const uint32_t ds = d & 0xFFFu;
                    if (ds > 29u) state = MS_FAIL;
                    else {
                        const int dl = (int)(d >> 12);
                        const int xb = ds < 4u ? 0 : (int)((ds - 2u) >> 1);
                        const int base = ds < 4u ? 1 + (int)ds : 1 + (int)((2u + (ds & 1u)) << xb);
                        dist = base + (int)(((uint32_t)(vbuf >> dl)) & ((1u << xb) - 1u));
                        vbuf >>= (dl + xb); vcnt -= dl + xb;
                        if (dist > pos || pos + len > usize) state = MS_FAIL;
                        else cp = true;
                    }
                }
                // the copy: lane g of the group takes bytes g, g + 8, ...; with dist < len the source repeats with period dist (only
                // bytes in front of pos are read)
                const bool rep = dist < len;
                const float rd = rep && dist > 0 ? 1.0f / (float)dist : 0.0f;
                for (int i = g; __ballot(cp && i < len); i += MS_G) {
                    if (cp && i < len) {
                        int si = i;
                        if (rep) {
                            int q = (int)((float)i * rd);
                            si = i - q * dist;
                            if (si < 0) si += dist;
                            if (si >= dist) si -= dist;
                        }
#ifdef BZQ_MS_NOLOAD   // experiment: what the loop costs without the load -> store dependency of a match (wrong output)
                        gout[pos + i] = (uint8_t)si;
#else
                        gout[pos + i] = gout[pos - dist + si];
#endif
                    }
                }
                if (cp) pos += len;
            }
            if (__ballot(state == MS_EOB) || !__ballot(state == MS_DECODE)) break;
        }
        if (a.stats) tk_loop += wall_clock64() - tk1;
    }
    const u64 tk2 = a.stats ? wall_clock64() : 0;
    // ---- verdicts: sizes and CRC-32 of every block (the wave takes them in turn; it reads back what it stored itself)
    bool bad = have && state != MS_DONE;
    for (int s = 0; s < MS_S; ++s) {
        const int sl = s * MS_G;
        if (!rdlane(have && state == MS_DONE ? 1u : 0u, sl)) continue;
        const uint32_t o_lo = rdlane((uint32_t)(uintptr_t)out, sl), o_hi = rdlane((uint32_t)((uintptr_t)out >> 32), sl);
        const uint8_t* sout = reinterpret_cast<const uint8_t*>(((uintptr_t)o_hi << 32) | o_lo);
        const uint32_t crc = block_crc32(sout, (int)rdlane((uint32_t)usize, sl), s_crc_tab, s_x2n);
        if (crc != rdlane(blk.crc, sl) && grp == s) bad = true;
    }
    if (bad && g == 0) atomicMin(a.first_bad, (unsigned long long)bi);
    if (a.stats && lane == 0) { atomicAdd(&a.stats[5], tk_service); atomicAdd(&a.stats[6], tk_loop); atomicAdd(&a.stats[7], wall_clock64() - tk2); }
}

// bytes of ArgsMs::scratch for n_blocks blocks (1 KiB per block slot of every wave)
inline size_t ms_scratch_bytes(int64_t n_blocks) { return (size_t)((n_blocks + MS_S - 1) / MS_S) * MS_S * MS_SCRATCH_WORDS * 4; }
inline void launch_bgzf_inflate_ms(const ArgsMs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_bgzf_inflate_ms, dim3((unsigned)((a.n_blocks + MS_S - 1) / MS_S)), dim3(64), 0, s, a);
}

} // namespace inf
} // namespace bzq
