// bzq_stream.hpp -- batch mode with ONE read of the input (k_stream).  EXPERIMENTS build only: correct on the whole parity
// suite, and 2x SLOWER than the two-pass path on MI355X (4.0 ms vs 1.9 ms per 3.18 GB; profiles/r2_single_read.md has the
// numbers and the probes).  Kept as the measured answer to "why does batch mode read its input twice".
//
// The two-pass path (k_tile_aggregate2 -> tile scan -> k_fused<LB=false>) reads every input byte twice because a tile's
// output positions depend on everything before it.  Both passes run at the rate of the XCD <-> memory fabric (~5.6-5.9 TB/s,
// reads + writes together, Infinity-Cache hits included: profiles/r2_emit_bound.md), so the second read costs its full
// 0.54 ms.  This kernel removes it:
//
//   * a workgroup owns a SUPER-TILE of ST consecutive 16 KiB tiles, all fetched into registers at once (16 VGPRs per tile:
//     the register file is three times the LDS and the emit kernel leaves two thirds of it idle), so that more bytes are in
//     flight per LDS byte while workgroups wait for each other;
//   * phase A (per tile, no LDS copy of the bytes): newline bitmap -> newline count and line lengths per line CLASS
//     (line index mod 4: the line phase is not known yet), assuming that no header line loses bytes to _strip_spaces --
//     "hypothesis H"; the ST summaries are merged and published as three 8-byte {flag, value} granules;
//   * workgroups are numbered by an atomic ticket: blockIdx order is NOT start order across the 8 XCDs (they drift apart by
//     10-40 us, scripts/probes/order_probe.hip; waiting for a blockIdx predecessor cost 60 us per workgroup);
//   * one decoupled look-back per SUPER-TILE (wave 0): level 1 over the <= 63 earlier workgroups of its group of 64, level 2
//     by the group's first workgroup over the groups before (each group = one 64-byte line holding its aggregate and, once
//     resolved, its inclusive prefix), the other 63 poll that one line.  Class-form aggregates are merged with the rotation
//     the line counts imply, so line phase and the three column offsets come out of the same walk;
//   * phase B (per tile, through the ONE 16 KiB LDS buffer of the workgroup): exactly the emit of the two-pass path
//     (emit_tile below is that code), now with known roles; header lines are measured exactly (header_kept), and if any
//     of them contradicts hypothesis H the chunk is flagged and the host repeats it on the two-pass kernels.  Only ids with
//     leading / trailing posix spaces do that (the reference strips them, utils.mojo:221-242; real data has none).
//
// Why it loses: a workgroup waits 9 us for its group's earlier workgroups (their loads land with that much jitter) and
// 13 us more for the group prefix; to hide 20+ us of waiting a CU would have to hold ~3x the tiles it can (LDS: 6 x 27 KiB),
// and staging them in registers means one workgroup works through its tiles one after the other at 16 waves per CU, where
// every barrier-separated step of the emit is latency-bound (8.5 us per tile instead of 1.5 us of CU time).
// Every spin is bounded; a timeout also falls back to the two-pass kernels.  No atomics on the data path but the ticket.
#pragma once
#include "bzq_fused.hpp"

namespace bzq {

#ifndef BZQ_STREAM_ST
#define BZQ_STREAM_ST 4
#endif
constexpr int ST = BZQ_STREAM_ST;   // 16 KiB tiles per workgroup
#ifndef BZQ_STREAM_SGRP
#define BZQ_STREAM_SGRP 64
#endif
constexpr int SGRP = BZQ_STREAM_SGRP;   // workgroups per look-back group (<= 64: one lane per workgroup in level 1)
constexpr int WD_WORDS = 4;         // u64 granules per workgroup descriptor (3 used)
constexpr int GD_WORDS = 8;         // u64 granules per group descriptor: 4 aggregate + 4 prefix = one 64-byte line
constexpr u64 F_SET = 2ull << 62;   // granule flag: value present

struct StreamArgs {
    FusedArgs f;
    u64* wd;          // [n_wg * WD_WORDS], zeroed before the launch
    u64* gd;          // [n_groups * GD_WORDS], zeroed before the launch
    u64* ticket;      // one word, zeroed before the launch
    int64_t n_wg;
};

// Everything the emit of one tile keeps in LDS (26.7 KiB: six workgroups per CU).
struct EmitShared {
    __attribute__((aligned(16))) uint8_t tile_raw[16 + TILE + 32];
    __attribute__((aligned(16))) uint16_t mask[PIECES];   // newline bitmap, 16 bits per 16-byte piece; later the line index per piece
    uint16_t nl[MAXL + 4];
    u64 seg[3][SEGS];      // per role slot (0 id, 1 sequence, 2 quality): tile offset | length << 16 | (column offset - tile offset) << 32
    u64 colbase[4];
    u64 w64[4];
    uint32_t w[4];
    int64_t bcast[8];
    u64 sum[2];
};

// ---- class-form summaries ---------------------------------------------------------------------------------------------
// c newlines; a[k] / d[k]: bytes / id bytes (hypothesis H) of the lines whose index relative to the summary's first line
// is k mod 4.  merge(x, y) = x followed by y.
struct ClassSum {
    int64_t c, a[4], d[4];
};
__device__ __forceinline__ ClassSum cs_zero() { return ClassSum{0, {0, 0, 0, 0}, {0, 0, 0, 0}}; }
// element (i & 3) of a 4-vector without runtime indexing (register arrays indexed at run time go to scratch memory)
__device__ __forceinline__ int64_t pick4(const int64_t (&v)[4], int i) {
    i &= 3;
    return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3];
}
// y's classes seen from a start `rot` lines earlier: class k of the result is class (k - rot) of y
__device__ __forceinline__ void cs_rotate(int64_t (&v)[4], int rot) {
    const int64_t v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    rot &= 3;
    v[0] = rot == 0 ? v0 : rot == 1 ? v3 : rot == 2 ? v2 : v1;
    v[1] = rot == 0 ? v1 : rot == 1 ? v0 : rot == 2 ? v3 : v2;
    v[2] = rot == 0 ? v2 : rot == 1 ? v1 : rot == 2 ? v0 : v3;
    v[3] = rot == 0 ? v3 : rot == 1 ? v2 : rot == 2 ? v1 : v0;
}
__device__ __forceinline__ ClassSum cs_merge(const ClassSum& x, const ClassSum& y) {
    ClassSum r = y;
    const int rot = (int)(x.c & 3);
    cs_rotate(r.a, rot); cs_rotate(r.d, rot);
    r.c = x.c + y.c;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.a[k] += x.a[k]; r.d[k] += x.d[k]; }
    return r;
}

// One tile of the batch path with known prefixes: roles, segment table, per-record outputs, scatter, validation.  This is
// the body of k_fused<LB=false> as a function (same code, same LDS layout); `r` holds the tile (four 16-byte pieces per
// thread).  CHECK_H: flag header lines whose exact kept range differs from hypothesis H.
template <bool CA, bool CQ, bool OFFS, bool CHECK_H>
__device__ __forceinline__ void emit_tile(const FusedArgs& a, EmitShared& sh, int64_t t, const uint4 (&r)[4], int64_t P, int64_t S, int64_t Q,
                                          int64_t I, uint32_t prevb, ErrAcc& err, bool& overflow, bool& h_bad) {
    uint8_t* s_tile = sh.tile_raw + 16;
    uint16_t* s_pline = sh.mask;
    const int tid = threadIdx.x;
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    const bool first_starts = prevb == 10u;
    tile_stage<true>(r, valid, sh.mask, s_tile);
    ByteSrc bs{a.g, a.n, a.prev_byte, s_tile, t0, valid};
    if (a.walk_limit > 0) bs.walk_limit = a.walk_limit;
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(sh.mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), sh.w, c);
    const bool dense = ((int)c > MAXL) || a.force_dense;
    reinterpret_cast<uint32_t*>(&sh.seg[0][tid])[0] = 0u; reinterpret_cast<uint32_t*>(&sh.seg[1][tid])[0] = 0u;
    reinterpret_cast<uint32_t*>(&sh.seg[2][tid])[0] = 0u;
    if (!dense) {
        const uint32_t l0 = excl, l1 = l0 + (uint32_t)__popc((uint32_t)m64 & 0xFFFFu),
                       l2 = l0 + (uint32_t)__popc((uint32_t)m64), l3 = l0 + (uint32_t)__popcll(m64 & 0xFFFFFFFFFFFFull);
        *reinterpret_cast<u64*>(&s_pline[4 * tid]) = (u64)l0 | ((u64)l1 << 16) | ((u64)l2 << 32) | ((u64)l3 << 48);
        u64 m = m64;
        int idx = 0;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            sh.nl[excl + idx] = (uint16_t)(tid * 64 + bit);
            ++idx;
        }
    }
    __syncthreads();
    const int ph = (int)(P & 3);

    if (dense) {
        // any input: one thread walks every line of the tile (records of a few bytes; adversarial inputs)
        if (tid == 0) {
            int64_t rs = S, rq = Q, ri = I;
            int j = 0, line_start = 0;
            bool start_in = first_starts;
            auto handle = [&](int start, int end, bool end_in) {
                const int64_t L = P + j;
                const int role = (int)(L & 3);
                const int64_t rec = L >> 2;
                const int64_t ls = t0 + start, le = t0 + end;
                const bool sin = start_in && start < valid;
                if (role == 0) {
                    if (sin) {
                        if (s_tile[start] != 64) err.structure(rec, 1);
                        if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_hdr[rec] = ls;
                    }
                    int64_t lo = ls, hi = ls;
                    if (end > start) header_kept(bs, ls, le, start_in, end_in, t0 + valid, lo, hi);
                    if (CHECK_H && (hi - lo) != (int64_t)(end - start) - ((start_in && end > start) ? 1 : 0)) h_bad = true;
                    for (int64_t p = lo; p < hi; ++p) {
                        const uint8_t ch = s_tile[p - t0];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (ri + (p - lo) >= 0) a.col_id[ri + (p - lo)] = ch;
                    }
                    ri += hi - lo;
                    if (end_in && rec >= 0) { if (rec < a.rec_cap) a.id_ends[rec] = ri; else overflow = true; }
                } else if (role == 1) {
                    if (sin && OFFS && rec >= 0 && rec < a.rec_cap) a.o_seq[rec] = ls;
                    for (int p = start; p < end; ++p) {
                        const uint8_t ch = s_tile[p];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (rs + (p - start) >= 0) a.col_seq[rs + (p - start)] = ch;
                    }
                    rs += end - start;
                } else if (role == 2) {
                    if (sin) {
                        if (s_tile[start] != 43) err.structure(rec, 2);
                        if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_sep[rec] = ls;
                    }
                } else {
                    if (sin && OFFS && rec >= 0 && rec < a.rec_cap) a.o_qual[rec] = ls;
                    for (int p = start; p < end; ++p) {
                        const uint8_t ch = s_tile[p];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (CQ && (uint32_t)((ch - a.q_lower) & 0xFFu) > (a.q_upper - a.q_lower)) err.valid(rec, 5);
                        if (rq + (p - start) >= 0) a.col_qual[rq + (p - start)] = ch;
                    }
                    rq += end - start;
                    if (end_in && rec >= 0) {
                        if (rec < a.rec_cap) { a.ends[rec] = rq; a.rec_end[rec] = le; } else overflow = true;
                        if (rs != rq) err.structure(rec, 3);
                    }
                }
            };
            for (int w = 0; w < BLOCK; ++w) {
                u64 m = s_mask64[w];
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const int nl = w * 64 + bit;
                    handle(line_start, nl, true);
                    line_start = nl + 1;
                    start_in = true;
                    ++j;
                }
            }
            handle(line_start, valid, false);
            atomicAdd((u64*)&a.st->dense_tiles, 1ull);
        }
        return;
    }

    // ---- line pass: one line per thread; line j has role (ph+j)&3 and is segment k = j>>2 of that role
    for (int j = tid; j <= (int)c; j += BLOCK) {
        const int role = (ph + j) & 3;
        const int k = j >> 2;
        const int start = j ? (int)sh.nl[j - 1] + 1 : 0;
        const bool end_in = j < (int)c;
        const int end = end_in ? (int)sh.nl[j] : valid;
        const int64_t rec = (P + j) >> 2;
        const bool sknown = j > 0 ? true : first_starts;
        const bool sin = sknown && start < valid;
        const int64_t ls = t0 + start;
        if (role == 0) {
            if (sin) {
                if (s_tile[start] != 64) err.structure(rec, 1);   // '@', utils.mojo:454
                if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_hdr[rec] = ls;
            }
            int64_t lo = ls, hi = ls;
            if (end > start) header_kept(bs, ls, t0 + end, sknown, end_in, t0 + valid, lo, hi);
            if (CHECK_H && (hi - lo) != (int64_t)(end - start) - ((sknown && end > start) ? 1 : 0)) h_bad = true;
            reinterpret_cast<uint32_t*>(&sh.seg[0][k])[0] = (uint32_t)(lo - t0) | ((uint32_t)(hi - lo) << 16);
        } else if (role == 2) {
            if (sin) {
                if (s_tile[start] != 43) err.structure(rec, 2);   // '+', utils.mojo:456
                if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_sep[rec] = ls;
            }
        } else {
            const int slot = role == 1 ? 1 : 2;
            if (sin && OFFS && rec >= 0 && rec < a.rec_cap) (role == 1 ? a.o_seq : a.o_qual)[rec] = ls;
            reinterpret_cast<uint32_t*>(&sh.seg[slot][k])[0] = (uint32_t)start | ((uint32_t)(end - start) << 16);
        }
    }
    __syncthreads();
    const uint32_t g0 = reinterpret_cast<const uint32_t*>(&sh.seg[0][tid])[0], g1 = reinterpret_cast<const uint32_t*>(&sh.seg[1][tid])[0],
                   g2 = reinterpret_cast<const uint32_t*>(&sh.seg[2][tid])[0];
    const uint32_t lh = g0 >> 16, lsq = g1 >> 16, lq = g2 >> 16;
    const u64 packed = (u64)lh | ((u64)lsq << 21) | ((u64)lq << 42);
    u64 tot = 0;
    const u64 ex = block_exclusive_scan<u64, 4>(packed, sh.w64, tot);
    const int dh = (int)(ex & 0x1FFFFFull), ds = (int)((ex >> 21) & 0x1FFFFFull), dq = (int)((ex >> 42) & 0x1FFFFFull);
    reinterpret_cast<int32_t*>(&sh.seg[0][tid])[1] = dh - (int)(g0 & 0xFFFFu);
    reinterpret_cast<int32_t*>(&sh.seg[1][tid])[1] = ds - (int)(g1 & 0xFFFFu);
    reinterpret_cast<int32_t*>(&sh.seg[2][tid])[1] = dq - (int)(g2 & 0xFFFFu);
    if (tid < 4) sh.colbase[tid] = (u64)(tid == 0 ? a.col_id + I : (tid == 1 ? a.col_seq + S : a.col_qual + Q));
    __syncthreads();

    // ---- per-record outputs of lines that END in this tile
    const int jh = (0 - ph) & 3, jq = (3 - ph) & 3;
    {
        const int j = 4 * tid + jh;           // this thread's header line
        const int64_t rec = (P + j) >> 2;
        if (j < (int)c && rec >= 0) {
            if (rec < a.rec_cap) a.id_ends[rec] = I + (int64_t)(dh + (int)lh);
            else overflow = true;
        }
    }
    {
        const int j = 4 * tid + jq;           // this thread's quality line
        const int64_t rec = (P + j) >> 2;
        if (j < (int)c && rec >= 0) {
            const int64_t qe = Q + (int64_t)(dq + (int)lq);
            const int64_t se = S + (int64_t)(jq >= 2 ? ds + (int)lsq : ds);
            if (rec < a.rec_cap) { a.ends[rec] = qe; a.rec_end[rec] = t0 + (int64_t)sh.nl[j]; }
            else overflow = true;
            if (se != qe) err.structure(rec, 3); // utils.mojo:458-461 as a cumulative test
        }
    }
    // ---- scatter: whole 16-byte source pieces inside one kept line straight from registers; first and last 16 bytes of
    // every line from LDS (unaligned both sides); shorter lines byte-exact.  Lines with a negative index belong to the
    // previous shard: not written.
    const int jmin = P < 0 ? (int)(-P) : 0;
    {
        uint32_t jj[4];
        u64 sg[4], cb[4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) jj[sidx] = s_pline[tid + BLOCK * sidx];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const uint32_t role = ((uint32_t)ph + jj[sidx]) & 3u;
            sg[sidx] = sh.seg[role < 2u ? role : 2u][jj[sidx] >> 2];
            cb[sidx] = sh.colbase[role];
        }
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int pos = (tid + BLOCK * sidx) * 16;
            const uint32_t role = ((uint32_t)ph + jj[sidx]) & 3u;
            const int src = (int)((uint32_t)sg[sidx] & 0xFFFFu), len = (int)((uint32_t)sg[sidx] >> 16);
            const int delta = (int)(sg[sidx] >> 32);
            if (role != 2u && pos >= src && pos + 16 <= src + len && (int)jj[sidx] >= jmin) {
                const int64_t rec = (P + (int64_t)jj[sidx]) >> 2;
                const uint4 pv = r[sidx];
                if (CA && any_non_ascii(pv.x | pv.y | pv.z | pv.w)) err.valid(rec, 4);
                if (CQ && role == 3u &&
                    (any_out_of_range(pv.x, a.q_lower, a.q_upper) | any_out_of_range(pv.y, a.q_lower, a.q_upper) |
                     any_out_of_range(pv.z, a.q_lower, a.q_upper) | any_out_of_range(pv.w, a.q_lower, a.q_upper)))
                    err.valid(rec, 5);
                const U16B v{pv.x, pv.y, pv.z, pv.w};
                *reinterpret_cast<U16B*>(reinterpret_cast<uint8_t*>(cb[sidx]) + (int64_t)(delta + pos)) = v;
            }
        }
    }
    {
        const int nseg = ((int)c + 4) >> 2; // segment indices in use
        for (int pidx = tid; pidx < 6 * nseg; pidx += BLOCK) {
            const int side = pidx & 1, sk = pidx >> 1;
            const int k = sk / 3, slot = sk - 3 * k;
            const u64 sg = sh.seg[slot][k];
            const int len = (int)((uint32_t)sg >> 16);
            const int role = slot == 2 ? 3 : slot;
            const int j = 4 * k + ((role - ph) & 3);
            if (len == 0 || j < jmin) continue;
            const int src = (int)((uint32_t)sg & 0xFFFFu), delta = (int)(sg >> 32);
            uint8_t* col = reinterpret_cast<uint8_t*>(sh.colbase[role]);
            const int64_t rec = (P + j) >> 2;
            if (len >= 16) {
                const int off = side ? src + len - 16 : src;
                if (((side ? src + len : src) & 15) == 0) continue;   // that end is a whole piece already
                copy16<CA, CQ>(col, (int64_t)(delta + off), off, s_tile, rec, slot == 2, a.q_lower, a.q_upper, err);
            } else if (side == 0) {
                emit_part_rt<CA, CQ>(col, (int64_t)(delta + src), src, len, s_tile, rec, slot == 2, a.q_lower, a.q_upper, err);
            }
        }
    }
}

// ---- phase A: summary of one tile from its newline bitmap alone ---------------------------------------------------------
// Returns (in every thread) c and the packed class sums pa / pd (4 x 16 bits: bytes / id bytes under hypothesis H per line
// class).  A tile with more than MAXL_A newlines is counted by one thread from the bitmap (any input).
__device__ __forceinline__ void tile_summary(const uint4 (&r)[4], int valid, bool first_starts, EmitShared& sh, uint32_t& c_out, u64& pa_out,
                                             u64& pd_out) {
    const int tid = threadIdx.x, lane = tid & 63;
    tile_stage<false>(r, valid, sh.mask, nullptr);
    if (tid < 2) sh.sum[tid] = 0;
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(sh.mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), sh.w, c);
    u64 pa = 0, pd = 0;
    auto add_line = [&](int j, int start, int end) {
        const int len = end - start;
        if (len <= 0) return;
        const bool sknown = j > 0 ? true : first_starts;
        pa += (u64)len << (16 * (j & 3));
        pd += (u64)(len - (sknown ? 1 : 0)) << (16 * (j & 3));   // hypothesis H: only the '@' position is dropped
    };
    if ((int)c <= MAXL) {
        u64 m = m64;
        int idx = 0;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            sh.nl[excl + idx] = (uint16_t)(tid * 64 + bit);
            ++idx;
        }
        __syncthreads();
        for (int j = tid; j <= (int)c; j += BLOCK) add_line(j, j ? (int)sh.nl[j - 1] + 1 : 0, j < (int)c ? (int)sh.nl[j] : valid);
    } else if (tid == 0) {
        int j = 0, line_start = 0;
        for (int w = 0; w < BLOCK; ++w) {
            u64 m = s_mask64[w];
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                add_line(j, line_start, w * 64 + bit);
                line_start = w * 64 + bit + 1;
                ++j;
            }
        }
        add_line(j, line_start, valid);
    }
    pa = wave_sum_u64(pa); pd = wave_sum_u64(pd);   // every field total <= 16384: no carry between fields
    if (lane == 0) { atomicAdd(&sh.sum[0], pa); atomicAdd(&sh.sum[1], pd); }
    __syncthreads();
    c_out = c; pa_out = sh.sum[0]; pd_out = sh.sum[1];
    __syncthreads();
}

__device__ __forceinline__ u64 uniform64(u64 v) {
    return ((u64)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (u64)__builtin_amdgcn_readfirstlane((uint32_t)v);
}
__device__ __forceinline__ ClassSum cs_from_packed(uint32_t c, u64 pa, u64 pd) {
    ClassSum s;
    s.c = c;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s.a[k] = (int64_t)((pa >> (16 * k)) & 0xFFFFull); s.d[k] = (int64_t)((pd >> (16 * k)) & 0xFFFFull); }
    return s;
}

// ---- descriptors ---------------------------------------------------------------------------------------------------------
// workgroup (super-tile of ST tiles, <= 2^16 bytes): G0 = c | a0 << 20 | a1 << 40, G1 = a2 | d0 << 20 | d1 << 40, G2 = d2 | d3 << 20
// (a3 = bytes - c - a0 - a1 - a2).  group (64 workgroups, <= 2^22 bytes): words 0..3 aggregate {c | a0 << 31}, {a1 | a2 << 31},
// {d0 | d1 << 31}, {d2 | d3 << 31} (a3 derived), words 4..7 inclusive prefix P, S, Q, I (biased by DESC_BIAS).
__device__ __forceinline__ bool g_set(u64 g) { return (g >> 62) == 2; }
__device__ __forceinline__ u64 g_val(u64 g) { return g & DESC_VMASK; }

__device__ __forceinline__ ClassSum wd_unpack(u64 g0, u64 g1, u64 g2, int64_t bytes) {
    ClassSum s;
    const u64 v0 = g_val(g0), v1 = g_val(g1), v2 = g_val(g2);
    s.c = (int64_t)(v0 & 0xFFFFFull); s.a[0] = (int64_t)((v0 >> 20) & 0xFFFFFull); s.a[1] = (int64_t)((v0 >> 40) & 0xFFFFFull);
    s.a[2] = (int64_t)(v1 & 0xFFFFFull); s.d[0] = (int64_t)((v1 >> 20) & 0xFFFFFull); s.d[1] = (int64_t)((v1 >> 40) & 0xFFFFFull);
    s.d[2] = (int64_t)(v2 & 0xFFFFFull); s.d[3] = (int64_t)((v2 >> 20) & 0xFFFFFull);
    s.a[3] = bytes - s.c - s.a[0] - s.a[1] - s.a[2];
    return s;
}

// Sum over the wave of class-form summaries held one per lane in lane order (lanes >= count hold zeros): each lane
// rotates its classes by the line count of the lanes before it, then the fields are added up.
__device__ __forceinline__ ClassSum wave_merge(ClassSum mine) {
    const u64 incl = dpp_scan_u64((u64)mine.c);
    const int rot = (int)((incl - (u64)mine.c) & 3);
    cs_rotate(mine.a, rot); cs_rotate(mine.d, rot);
    ClassSum r;
    r.c = (int64_t)wave_sum_u64((u64)mine.c);
    // fields are < 2^22 each and there are 64 lanes: sums < 2^28, two per u64
    const u64 s0 = wave_sum_u64((u64)mine.a[0] | ((u64)mine.a[1] << 32)), s1 = wave_sum_u64((u64)mine.a[2] | ((u64)mine.a[3] << 32));
    const u64 s2 = wave_sum_u64((u64)mine.d[0] | ((u64)mine.d[1] << 32)), s3 = wave_sum_u64((u64)mine.d[2] | ((u64)mine.d[3] << 32));
    r.a[0] = (int64_t)(s0 & 0xFFFFFFFFull); r.a[1] = (int64_t)(s0 >> 32); r.a[2] = (int64_t)(s1 & 0xFFFFFFFFull); r.a[3] = (int64_t)(s1 >> 32);
    r.d[0] = (int64_t)(s2 & 0xFFFFFFFFull); r.d[1] = (int64_t)(s2 >> 32); r.d[2] = (int64_t)(s3 & 0xFFFFFFFFull); r.d[3] = (int64_t)(s3 >> 32);
    return r;
}

struct Prefix { int64_t P, S, Q, I; };
// what a run of lines with summary s adds when it starts at line index P0
__device__ __forceinline__ Prefix prefix_advance(Prefix p, const ClassSum& s) {
    const int ph = (int)(p.P & 3);
    Prefix r;
    r.P = p.P + s.c;
    r.S = p.S + pick4(s.a, 1 - ph);
    r.Q = p.Q + pick4(s.a, 3 - ph);
    r.I = p.I + pick4(s.d, 0 - ph);
    return r;
}

template <bool CA, bool CQ, bool OFFS>
static __global__ __launch_bounds__(BLOCK) void k_stream(StreamArgs sa) {
    const FusedArgs& a = sa.f;
    __shared__ EmitShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Which super-tile?  NOT blockIdx: workgroups are handed to the 8 XCDs round-robin (block b runs on XCD b % 8) and every
    // XCD works through its own blocks in order, but the XCDs drift apart by 10-40 us (scripts/probes/order_probe.hip), so
    // a workgroup that waits for its blockIdx predecessor waits for another XCD to catch up (measured: 60 us per
    // workgroup).  A ticket makes the index order the START order on the whole chip: a workgroup only ever waits for
    // workgroups that started before it.  One returning atomic per super-tile (~40 per us at ST = 4) is below what one
    // counter sustains (~85 per us; with one 16 KiB tile per workgroup it was the bottleneck).
    if (tid == 0) sh.bcast[0] = (int64_t)atomicAdd(sa.ticket, 1ull);
    __syncthreads();
    const int64_t w = sh.bcast[0];
    __syncthreads();
    if (w >= sa.n_wg) return;
    const int64_t tb = w * ST;                      // first tile of this workgroup
    // ---- fetch: all ST tiles into registers
    uint4 r[ST][4];
    uint32_t prevb[ST];
    int valid[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        const int64_t t0 = (tb + s) * TILE;
        valid[s] = t0 < a.n ? (int)((a.n - t0) < TILE ? (a.n - t0) : TILE) : 0;
        prevb[s] = 10u;
        if (valid[s] > 0) {
            prevb[s] = t0 > 0 ? (uint32_t)a.g[t0 - 1] : a.prev_byte;
            tile_fetch(a.g, a.n, t0, valid[s], r[s]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) r[s][q] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // ---- phase A: summaries of the tiles, merged into the super-tile's
    uint32_t tc[ST];          // per tile, packed (wave-uniform values: kept in scalar registers)
    u64 tpa[ST], tpd[ST];
    ClassSum mine = cs_zero();
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        uint32_t c = 0; u64 pa = 0, pd = 0;
        if (valid[s] > 0) tile_summary(r[s], valid[s], prevb[s] == 10u, sh, c, pa, pd);
        tc[s] = __builtin_amdgcn_readfirstlane(c); tpa[s] = uniform64(pa); tpd[s] = uniform64(pd);
        mine = cs_merge(mine, cs_from_packed(tc[s], tpa[s], tpd[s]));
    }
    // ---- publish + look-back (wave 0)
    const int64_t grp = w / SGRP;
    const int gi = (int)(w - grp * SGRP);
    // EXPERIMENTS build, option ablate bit 64: where does a workgroup's time go (10 ns ticks, one workgroup in 64)?
    const bool probe = BZQ_ABLATE(64) && tid == 0 && gi == 37;
    u64 tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, pl1 = 0, pl2 = 0;
    (void)tk3;
    if (BZQ_ABLATE(64)) tk0 = wall_clock64();
    const bool last_in_group = gi == SGRP - 1 || w == sa.n_wg - 1;
    if (wave == 0) {
        if (lane == 0) {
            st_agent(&sa.wd[w * WD_WORDS + 0], F_SET | (u64)mine.c | ((u64)mine.a[0] << 20) | ((u64)mine.a[1] << 40));
            st_agent(&sa.wd[w * WD_WORDS + 1], F_SET | (u64)mine.a[2] | ((u64)mine.d[0] << 20) | ((u64)mine.d[1] << 40));
            st_agent(&sa.wd[w * WD_WORDS + 2], F_SET | (u64)mine.d[2] | ((u64)mine.d[3] << 20));
        }
        bool timeout = false;
        // level 1: the workgroups of this group before this one (lane L <-> workgroup grp*SGRP + L)
        ClassSum before = cs_zero();
        if (gi > 0) {
            const int64_t q = grp * SGRP + lane;
            const bool want = lane < gi;
            u64 g0 = 0, g1 = 0, g2 = 0;
            bool ok = !want;
            int spins = 0;
            for (;;) {
                if (!ok) {   // only the lanes still waiting load again: polls must not eat the bandwidth the tiles need
                    g0 = ld_agent(&sa.wd[q * WD_WORDS + 0]); g1 = ld_agent(&sa.wd[q * WD_WORDS + 1]); g2 = ld_agent(&sa.wd[q * WD_WORDS + 2]);
                    ok = g_set(g0) && g_set(g1) && g_set(g2);
                }
                ++pl1;
                if (__ballot(!ok) == 0) break;
                __builtin_amdgcn_s_sleep(32);
                if (++spins > SPIN_LIMIT) { timeout = true; break; }
            }
            // bytes of workgroup q: full super-tiles except possibly the chunk's last one (which is never a predecessor)
            ClassSum v = want && !timeout ? wd_unpack(g0, g1, g2, (int64_t)ST * TILE) : cs_zero();
            before = wave_merge(v);
        }
        if (BZQ_ABLATE(64)) tk1 = wall_clock64();
        const ClassSum through = cs_merge(before, mine);   // group start .. end of this workgroup
        if (last_in_group && lane == 0) {
            u64* gd = &sa.gd[grp * GD_WORDS];
            st_agent(&gd[0], F_SET | (u64)through.c | ((u64)through.a[0] << 31));
            st_agent(&gd[1], F_SET | (u64)through.a[1] | ((u64)through.a[2] << 31));
            st_agent(&gd[2], F_SET | (u64)through.d[0] | ((u64)through.d[1] << 31));
            st_agent(&gd[3], F_SET | (u64)through.d[2] | ((u64)through.d[3] << 31));
        }
        // level 2: the prefix at the start of this group.  The group's FIRST workgroup resolves it -- it walks the groups
        // before, nearest first (lane L <-> group base - L), up to the nearest one whose inclusive prefix is published, adding
        // the aggregates in between -- and publishes it as the inclusive prefix of the previous group; the other 63 workgroups
        // of the group only poll that one line (a walk by every workgroup cost more fabric requests than the tiles themselves).
        Prefix x{a.st->P0, a.st->S0, a.st->Q0, a.st->I0};   // prefix at the start of this group
        if (grp > 0 && gi > 0 && !timeout) {
            const u64* gp = &sa.gd[(grp - 1) * GD_WORDS + 4];
            u64 v = 0;
            int spins = 0;
            for (;;) {
                if (lane < 4) v = ld_agent(&gp[lane]);
                ++pl2;
                if ((__ballot(lane < 4 && g_set(v)) & 0xFull) == 0xFull) break;
                __builtin_amdgcn_s_sleep(32);
                if (++spins > SPIN_LIMIT) { timeout = true; break; }
            }
            auto lane_val = [&](int l) { return (int64_t)(((u64)(uint32_t)__shfl((int)(uint32_t)v, l, 64)) | ((u64)(uint32_t)__shfl((int)(uint32_t)(v >> 32), l, 64) << 32)); };
            const u64 vp = (u64)lane_val(0), vs = (u64)lane_val(1), vq = (u64)lane_val(2), vi = (u64)lane_val(3);
            x.P = (int64_t)g_val(vp) - DESC_BIAS; x.S = (int64_t)g_val(vs) - DESC_BIAS; x.Q = (int64_t)g_val(vq) - DESC_BIAS; x.I = (int64_t)g_val(vi) - DESC_BIAS;
        }
        if (grp > 0 && gi == 0 && !timeout) {
            ClassSum run = cs_zero();                      // groups (base, grp) not yet covered by a prefix, merged oldest first
            int64_t base = grp - 1;
            int spins = 0;
            for (;;) {
                const int64_t q = base - lane;
                u64 v[GD_WORDS];
                bool have_a = false, have_p = false;
                if (q >= 0) {
                    const u64* gd = &sa.gd[q * GD_WORDS];
#pragma unroll
                    for (int i = 0; i < GD_WORDS; ++i) v[i] = ld_agent(&gd[i]);
                    have_a = g_set(v[0]) && g_set(v[1]) && g_set(v[2]) && g_set(v[3]);
                    have_p = g_set(v[4]) && g_set(v[5]) && g_set(v[6]) && g_set(v[7]);
                } else {
#pragma unroll
                    for (int i = 0; i < GD_WORDS; ++i) v[i] = 0;
                    have_p = true;   // before the first group: the chunk's initial prefix
                }
                const u64 pm = __ballot(have_p), am = __ballot(have_a || have_p);
                const int f = pm ? __builtin_ctzll(pm) : 64;                 // nearest lane with a prefix
                const u64 need = f >= 64 ? ~0ull : ((1ull << f) - 1ull);     // nearer lanes must at least have their aggregate
                if ((am & need) != need) {
                    __builtin_amdgcn_s_sleep(32);
                    if (++spins > SPIN_LIMIT) { timeout = true; break; }
                    continue;
                }
                // aggregates of the lanes nearer than f, merged oldest (highest lane) first
                ClassSum g = cs_zero();
                if (lane < f && q >= 0) {
                    const u64 v0 = g_val(v[0]), v1 = g_val(v[1]), v2 = g_val(v[2]), v3 = g_val(v[3]);
                    g.c = (int64_t)(v0 & 0x7FFFFFFFull); g.a[0] = (int64_t)(v0 >> 31);
                    g.a[1] = (int64_t)(v1 & 0x7FFFFFFFull); g.a[2] = (int64_t)(v1 >> 31);
                    g.d[0] = (int64_t)(v2 & 0x7FFFFFFFull); g.d[1] = (int64_t)(v2 >> 31);
                    g.d[2] = (int64_t)(v3 & 0x7FFFFFFFull); g.d[3] = (int64_t)(v3 >> 31);
                    g.a[3] = (int64_t)SGRP * ST * TILE - g.c - g.a[0] - g.a[1] - g.a[2];
                }
                // wave_merge wants stream order in lane order: reverse the lanes (lane L <- lane 63-L)
                ClassSum rev;
                auto flip = [&](int64_t x) { return (int64_t)(((u64)(uint32_t)__shfl((int)(uint32_t)x, 63 - lane, 64)) | ((u64)(uint32_t)__shfl((int)(uint32_t)((u64)x >> 32), 63 - lane, 64) << 32)); };
                rev.c = flip(g.c);
#pragma unroll
                for (int k = 0; k < 4; ++k) { rev.a[k] = flip(g.a[k]); rev.d[k] = flip(g.d[k]); }
                const ClassSum part = wave_merge(rev);      // groups base-f+1 .. base in stream order
                run = cs_merge(part, run);
                if (f < 64) {
                    // lane f holds the prefix (or stands for the chunk start)
                    const int64_t qf = base - f;
                    Prefix pf{a.st->P0, a.st->S0, a.st->Q0, a.st->I0};
                    if (qf >= 0) {
                        auto pick = [&](u64 x) { return (int64_t)(((u64)(uint32_t)__shfl((int)(uint32_t)x, f, 64)) | ((u64)(uint32_t)__shfl((int)(uint32_t)(x >> 32), f, 64) << 32)); };
                        pf.P = (int64_t)g_val((u64)pick(v[4])) - DESC_BIAS; pf.S = (int64_t)g_val((u64)pick(v[5])) - DESC_BIAS;
                        pf.Q = (int64_t)g_val((u64)pick(v[6])) - DESC_BIAS; pf.I = (int64_t)g_val((u64)pick(v[7])) - DESC_BIAS;
                    }
                    x = prefix_advance(pf, run);
                    break;
                }
                base -= 64;
            }
            if (lane == 0 && !timeout) {   // = the inclusive prefix of the previous group (its last workgroup writes the same values)
                u64* gd = &sa.gd[(grp - 1) * GD_WORDS];
                st_agent(&gd[4], F_SET | (u64)(x.P + DESC_BIAS)); st_agent(&gd[5], F_SET | (u64)(x.S + DESC_BIAS));
                st_agent(&gd[6], F_SET | (u64)(x.Q + DESC_BIAS)); st_agent(&gd[7], F_SET | (u64)(x.I + DESC_BIAS));
            }
        }
        if (last_in_group && lane == 0 && !timeout) {
            const Prefix e = prefix_advance(x, through);
            u64* gd = &sa.gd[grp * GD_WORDS];
            st_agent(&gd[4], F_SET | (u64)(e.P + DESC_BIAS)); st_agent(&gd[5], F_SET | (u64)(e.S + DESC_BIAS));
            st_agent(&gd[6], F_SET | (u64)(e.Q + DESC_BIAS)); st_agent(&gd[7], F_SET | (u64)(e.I + DESC_BIAS));
        }
        const Prefix p = prefix_advance(x, before);          // at this workgroup's first byte
        if (BZQ_ABLATE(64)) tk2 = wall_clock64();
        if (lane == 0) {
            if (probe) {
                atomicAdd(&a.st->phase_cycles[0], tk1 - tk0); atomicAdd(&a.st->phase_cycles[1], tk2 - tk1);
                atomicAdd(&a.st->phase_cycles[4], pl1); atomicAdd(&a.st->phase_cycles[5], pl2); atomicAdd(&a.st->phase_cycles[6], 1ull);
            }
            sh.bcast[0] = p.P; sh.bcast[1] = p.S; sh.bcast[2] = p.Q; sh.bcast[3] = p.I; sh.bcast[4] = timeout ? 1 : 0;
            if (timeout) a.st->lookback_timeout = 1;
        }
    }
    __syncthreads();
    Prefix p{sh.bcast[0], sh.bcast[1], sh.bcast[2], sh.bcast[3]};
    const bool dead = sh.bcast[4] != 0;
    __syncthreads();
    // ---- phase B: the tiles one after the other through the LDS buffer
    ErrAcc err{~0ull, ~0ull};
    bool overflow = false, h_bad = false;
#pragma unroll
    for (int s = 0; s < ST; ++s) {
        if (valid[s] > 0 && !dead) {
            emit_tile<CA, CQ, OFFS, true>(a, sh, tb + s, r[s], p.P, p.S, p.Q, p.I, prevb[s], err, overflow, h_bad);
            if (tc[s] > 0 && tid == 0) atomicMax((long long*)&a.st->last_nl_tile, (long long)(tb + s));
        }
        p = prefix_advance(p, cs_from_packed(tc[s], tpa[s], tpd[s]));
        __syncthreads();
    }
    if (probe) { tk3 = wall_clock64(); atomicAdd(&a.st->phase_cycles[2], tk3 - tk0); }
    if (w == sa.n_wg - 1 && tid == 0 && !dead) { a.st->P = p.P; a.st->S = p.S; a.st->Q = p.Q; a.st->I = p.I; }
    if (err.e_struct != ~0ull) atomicMin(&a.st->err_struct, err.e_struct);
    if (err.e_valid != ~0ull) atomicMin(&a.st->err_valid, err.e_valid);
    if (overflow) atomicOr(&a.st->rec_overflow, 1);
    if (h_bad) a.st->lookback_timeout = 2;   // hypothesis H failed somewhere: the host repeats the chunk on the two-pass kernels
}

} // namespace bzq
