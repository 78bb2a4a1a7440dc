// bzq_fused.hpp -- the emit kernel of the FASTQ batch-parse path: k_fused<..., LB>.
//
// The PRODUCT instantiates only LB = false: pass B of the two-pass path.  Pass A (k_tile_aggregate_h, bzq_device.hpp) and the
// tile scan have already resolved the two cross-tile dependencies
//   (1) the line index of the tile's first byte        -> which lines are header / sequence / '+' / quality
//   (2) the column offsets of the tile's three streams -> where the bytes go
// so every 16 KiB tile is loaded (coalesced 16 B/lane into LDS and registers), analysed, and its sequence / quality / id
// streams are scattered straight into the packed FastqBatch columns -- no waiting between workgroups.
//
// LB = true is the single-read look-back variant of round 1 (both dependencies resolved in-kernel by decoupled look-backs over
// 8-byte {flag, value} granules, tiles numbered by an atomic ticket, bounded spins with a host fallback).  It is correct and
// slower than two passes on this part (profiles/r2_single_read.md) and is compiled only into the EXPERIMENTS library
// (experiments/csrc, `make exp`); the look-back helpers below exist for it.
//
// Reference semantics implemented here: see the header of bzq_device.hpp (same citations).
#pragma once
#include "bzq_device.hpp"

namespace bzq {

__device__ __forceinline__ int64_t wave_sum(int64_t v) { return (int64_t)wave_sum_u64((u64)v); }

// Timing experiments (skip / stop-after-phase switches and the phase clock) cost SGPRs and prologue instructions in
// every workgroup, so they exist only in an EXPERIMENTS build; the shipped kernels ignore the "ablate" option.
#ifndef BZQ_EXPERIMENTS
#define BZQ_EXPERIMENTS 0
#endif
#define BZQ_ABLATE(bit) (BZQ_EXPERIMENTS && (a.ablate & (bit)))

struct FusedArgs {
    const uint8_t* g;
    int64_t n;
    uint32_t prev_byte;
    int64_t n_tiles;
    u64* ticket;     // 1 word, zeroed before the launch
    u64* desc_c;     // [n_tiles]   newline count aggregate / inclusive line prefix
    u64* desc_agg;   // [n_tiles]   packed (seq, qual, id) byte counts of the tile
    u64* desc_pre;   // [3*n_tiles] inclusive column prefixes S, Q, I
    // LB == false: prefixes come from the tile scan (k_tile_aggregate2 + k_scan_*)
    int64_t tile_begin, tile_end;
    const int64_t* tileP;
    const int64_t* tileS;
    const int64_t* tileQ;
    const int64_t* tileI;
    uint8_t* col_seq;
    uint8_t* col_qual;
    uint8_t* col_id;
    int64_t* ends;
    int64_t* id_ends;
    int64_t* rec_end;
    int64_t rec_cap;
    int64_t* o_hdr;
    int64_t* o_seq;
    int64_t* o_sep;
    int64_t* o_qual;
    ChunkState* st;
    uint32_t q_lower, q_upper;
    int32_t force_dense;
    int64_t walk_limit;   // ByteSrc::walk_limit (0 = none)
    int32_t check_h;      // pass A was k_tile_aggregate_h: flag every header line whose kept range is not "all but the '@'"
    int32_t ablate;   // timing experiments, compiled in only with -DBZQ_EXPERIMENTS=1 (make EXPERIMENTS=1): see BZQ_ABLATE uses
    // LB only, round-3 experiment (profiles/r3_single_read_xcd.md): > 0 = every XCD walks its own contiguous run of this many
    // tiles with look-backs that never leave the XCD; xcd_base[4 e + {0,1,2,3}] = line index and the three column offsets at the
    // start of run e -- handed in by the host (taken from a two-pass run of the same input: what a per-run column layout would make
    // unnecessary is simply given here, so that the kernel's time can be measured and its output compared)
    int64_t xcd_tiles;
    const int64_t* xcd_base;
    // fold != 0 (two-pass path, one pass, batch_size >= FOLD_MIN_BATCH): the emit writes the per-batch ends DIRECTLY (record_batch.mojo:
    // 77-87: _ends / _id_ends restart at every batch) from the batch bases k_batch_bases left in bb -- and NOT the chunk-cumulative
    // `ends` / `id_ends`, which nothing on the batches() path reads (bzq_chunk_cumulative_ends derives them on demand) -- and does the
    // reference's buffer-capacity refusal (parser.mojo:484-492) from the record's own length, with the end of the record before the
    // tile's first one located through tileP / tile_last.  The pass over the per-record arrays (k_rebase) is gone, and the emit
    // writes the same 24 B per record as before (measured: writing both kinds of ends from the emit costs what k_rebase cost)
    int32_t fold;
    int64_t* b_ends;
    int64_t* b_id_ends;
    const int64_t* bb;         // bb[2k], bb[2k+1]: ends / id_ends at the last record of batch k
    int64_t bb_cap;            // batches the table holds (a tile beyond it only ends records beyond rec_cap: nothing is written there)
    const int32_t* tileB;      // batch index of the record the tile's first line belongs to
    const u64* tile_last;      // AggArgs::tile_last
    int64_t batch, len_limit, first_header;
};
// A pointer that went through LDS as an integer has lost its address space: dereferenced as it is, it becomes a FLAT access, which
// counts against the LDS counter too and stalls the LDS reads of the scatter behind it (measured: emit + 3 %).  Back to global.
typedef __attribute__((address_space(1))) int64_t g_i64;
__device__ __forceinline__ g_i64* as_global(int64_t p) { return (g_i64*)(unsigned long long)p; }
constexpr int64_t FOLD_MIN_BATCH = 256;   // a fast-path tile ends at most 255 records: at most ONE batch boundary per tile

// End offset (its '\n') of the record whose last line has index lprev, a line that ends before tile t: the tile that holds it is
// the last one whose line prefix is <= lprev, and there it is the last newline of its class (fewer than four newlines follow it
// before tile t's first record end).  Gives up (too_long) once the walk has covered more bytes than the longest record the
// reference's buffer holds: the record that straddles into tile t is refused whatever its exact length.
__device__ __forceinline__ int64_t prev_record_end(const int64_t* tileP, int64_t tile_last /* address */, int64_t t, int64_t lprev, int64_t len_limit,
                                                   int64_t p_m1, u64 last_m1, bool& too_long) {
    // tile t-1 first, from the words the kernel loaded with its own prefixes (no load on this path: nearly every record)
    int64_t tt = t - 1;
    int64_t pt = p_m1;
    u64 lw = last_m1;
    if (pt > lprev) {
        const int64_t max_steps = len_limit / TILE + 2;
        int64_t steps = 0;
        while (tt > 0 && pt > lprev) {
            --tt;
            pt = tileP[tt];
            if (++steps > max_steps) { too_long = true; return 0; }
        }
        lw = (u64)as_global(tile_last)[tt];
    }
    const int jl = (int)(lprev - pt);
    return tt * (int64_t)TILE + (int64_t)((lw >> (16 * (jl & 3))) & 0xFFFFull);
}

#if BZQ_EXPERIMENTS
#include "bzq_lookback.hpp"   // experiments/csrc: the decoupled look-backs of the single-read variant (LB = true)
#endif

template <int ROLE, bool CA, bool CQ>
__device__ __forceinline__ void validate_window(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int a, int b,
                                                int64_t rec, uint32_t qlo, uint32_t qhi, ErrAcc& err) {
    if (!(CA || (CQ && ROLE == 3))) return;
    if (a > 0 || b < 16) {
        const uint32_t m0 = byte_range_mask(0, a, b), m1 = byte_range_mask(1, a, b), m2 = byte_range_mask(2, a, b),
                       m3 = byte_range_mask(3, a, b);
        const uint32_t fill = (CQ && ROLE == 3) ? 0x01010101u * qlo : 0u;
        w0 = (w0 & m0) | (fill & ~m0); w1 = (w1 & m1) | (fill & ~m1);
        w2 = (w2 & m2) | (fill & ~m2); w3 = (w3 & m3) | (fill & ~m3);
    }
    if (CA && any_non_ascii(w0 | w1 | w2 | w3)) err.valid(rec, 4);
    if (CQ && ROLE == 3) {
        if (any_out_of_range(w0, qlo, qhi) | any_out_of_range(w1, qlo, qhi) | any_out_of_range(w2, qlo, qhi) |
            any_out_of_range(w3, qlo, qhi))
            err.valid(rec, 5);
    }
}

// ---- source-driven scatter --------------------------------------------------------------------
// gfx950 handles byte-unaligned vector accesses in hardware (one global_store_dwordx4 for an
// align-1 16-byte store), so an aligned 16-byte source piece that lies wholly inside one line goes
// from the registers it was loaded into straight to its column position; only the <= 15-byte head
// and tail of every line go through the LDS window.
struct __attribute__((packed, aligned(1))) U16B { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U8B { uint32_t x, y; };
struct __attribute__((packed, aligned(1))) U4B { uint32_t x; };
struct __attribute__((packed, aligned(1))) U2B { uint16_t x; };

__device__ __forceinline__ void store_bytes(uint8_t* p, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int n) {
    // first n (< 16) bytes of the little-endian 16-byte value {w0..w3}
    if (n & 8) { U8B v{w0, w1}; *reinterpret_cast<U8B*>(p) = v; p += 8; w0 = w2; w1 = w3; }
    if (n & 4) { U4B v{w0}; *reinterpret_cast<U4B*>(p) = v; p += 4; w0 = w1; }
    if (n & 2) { U2B v{(uint16_t)w0}; *reinterpret_cast<U2B*>(p) = v; p += 2; w0 >>= 16; }
    if (n & 1) *p = (uint8_t)w0;
}

// n (1..15) bytes of the LDS tile starting at tile offset s0 -> col[gaddr ...]
template <int ROLE, bool CA, bool CQ>
__device__ __forceinline__ void emit_part(uint8_t* __restrict__ col, int64_t gaddr, int s0, int n, const uint8_t* s_tile,
                                          int64_t rec, uint32_t qlo, uint32_t qhi, ErrAcc& err) {
    if (n <= 0) return;
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(s_tile - 16);
    const int ws = s0 + 16, wd = ws >> 2, sh = ws & 3;
    const uint32_t d0 = tw[wd], d1 = tw[wd + 1], d2 = tw[wd + 2], d3 = tw[wd + 3], d4 = tw[wd + 4];
    const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                   w2 = __builtin_amdgcn_alignbyte(d3, d2, sh), w3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
    validate_window<ROLE, CA, CQ>(w0, w1, w2, w3, 0, n, rec, qlo, qhi, err);
    store_bytes(col + gaddr, w0, w1, w2, w3, n);
}

// 16 bytes of the LDS tile at any byte offset (gfx950 reads unaligned LDS addresses in one ds_read_b128)
__device__ __forceinline__ U16B lds_window16(const uint8_t* s_tile, int off) {
    return *reinterpret_cast<const U16B*>(s_tile + off);
}

// emit_part with the role decided at run time (all threads of the block share the head/tail work)
template <bool CA, bool CQ>
__device__ __forceinline__ void emit_part_rt(uint8_t* __restrict__ col, int64_t gaddr, int s0, int n, const uint8_t* s_tile,
                                             int64_t rec, bool is_qual, uint32_t qlo, uint32_t qhi, ErrAcc& err) {
    if (n <= 0) return;
    const U16B w = lds_window16(s_tile, s0);
    const uint32_t w0 = w.x, w1 = w.y, w2 = w.z, w3 = w.w;
    if (CA || CQ) {
        const uint32_t m0 = byte_range_mask(0, 0, n), m1 = byte_range_mask(1, 0, n), m2 = byte_range_mask(2, 0, n),
                       m3 = byte_range_mask(3, 0, n);
        if (CA && any_non_ascii((w0 & m0) | (w1 & m1) | (w2 & m2) | (w3 & m3))) err.valid(rec, 4);
        if (CQ && is_qual) {
            const uint32_t fill = 0x01010101u * qlo;
            if (any_out_of_range((w0 & m0) | (fill & ~m0), qlo, qhi) | any_out_of_range((w1 & m1) | (fill & ~m1), qlo, qhi) |
                any_out_of_range((w2 & m2) | (fill & ~m2), qlo, qhi) | any_out_of_range((w3 & m3) | (fill & ~m3), qlo, qhi))
                err.valid(rec, 5);
        }
    }
    store_bytes(col + gaddr, w0, w1, w2, w3, n);
}

// 16 whole bytes of one line: LDS tile offset off -> col[gaddr, gaddr+16), both unaligned
template <bool CA, bool CQ>
__device__ __forceinline__ void copy16(uint8_t* __restrict__ col, int64_t gaddr, int off, const uint8_t* s_tile, int64_t rec,
                                       bool is_qual, uint32_t qlo, uint32_t qhi, ErrAcc& err) {
    const U16B w = lds_window16(s_tile, off);
    if (CA && any_non_ascii(w.x | w.y | w.z | w.w)) err.valid(rec, 4);
    if (CQ && is_qual &&
        (any_out_of_range(w.x, qlo, qhi) | any_out_of_range(w.y, qlo, qhi) | any_out_of_range(w.z, qlo, qhi) | any_out_of_range(w.w, qlo, qhi)))
        err.valid(rec, 5);
    *reinterpret_cast<U16B*>(col + gaddr) = w;
}

// One workgroup per tile, straight-line (no persistent loop: a loop lets the compiler hoist ~35 VGPRs of invariants,
// and a register-prefetching persistent variant measured 10% SLOWER -- this kernel is bound by HBM traffic, not by
// reads in flight: capping it at 4 or 5 workgroups per CU instead of 6 does not change its time).
template <bool CA, bool CQ, bool OFFS, bool LB, bool FOLD = false>
static __global__ __launch_bounds__(BLOCK) void k_fused(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile_raw[16 + TILE + 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];   // newline bitmap, 16 bits per 16-byte piece
    __shared__ uint16_t s_nl[MAXL + 4];
    // segment k of role slot (0 id, 1 sequence, 2 quality): tile offset of its kept bytes | length << 16 |
    // (offset in the tile's part of the column - tile offset) << 32; one ds_read_b64 per lookup
    __shared__ u64 s_seg[3][SEGS];
    __shared__ u64 s_colbase[4];   // by line role: column pointer + this tile's column offset (role 2: unused)
    // tile-local line index at each 16-byte piece's first byte.  Lives in s_mask: a thread overwrites exactly the
    // 8 bytes of the bitmap it has just read (the serial dense path keeps the bitmap instead).
    uint16_t* s_pline = s_mask;
    __shared__ u64 s_w64[4];
    __shared__ uint32_t s_w[4];
    __shared__ int64_t s_bcast[4];   // tile, P / S, Q, I
    __shared__ int s_cnt[3];
    // FusedArgs::fold, one word per entry: base ends / base id_ends of the tile's first batch [0, 1] and of the next one [2, 3];
    // tileP[t-1], tile_last[t-1]; first record of the next batch, batch index of the tile's first record; then arguments
    __shared__ int64_t s_fold[16];
    enum { F_PM1 = 4, F_LASTM1 = 5, F_NEXTB = 6, F_KB0 = 7, F_BENDS = 8, F_BIDENDS = 9, F_LENLIM = 10, F_FIRSTHDR = 11, F_TILELAST = 12, F_BATCH = 13, F_BB = 14, F_BBCAP = 15 };
#if BZQ_EXPERIMENTS && defined(BZQ_PAD_LDS)
    __shared__ uint8_t s_pad[BZQ_PAD_LDS];   // experiment: cap workgroups per CU through LDS
    if (a.n < 0) s_pad[a.n & 1023] = 1;
#endif
    uint8_t* s_tile = s_tile_raw + 16;
    const int tid0 = threadIdx.x;

    if (LB) {
        if (tid0 == 0) {
            if (a.xcd_tiles > 0) {   // a ticket of THIS XCD's run (HW_REG_XCC_ID, bits 3:0); a run that is used up: help the next one
                uint32_t x = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;
                int64_t tt = -1;
                for (int tries = 0; tries < 8 && tt < 0; ++tries, x = (x + 1) & 7u) {
                    const int64_t lo = (int64_t)x * a.xcd_tiles, hi = lo + a.xcd_tiles < a.n_tiles ? lo + a.xcd_tiles : a.n_tiles;
                    if (lo >= hi) continue;
                    const int64_t k = (int64_t)atomicAdd(&a.ticket[x], 1ull);
                    if (lo + k < hi) tt = lo + k;
                }
                s_bcast[0] = tt;
            } else s_bcast[0] = (int64_t)atomicAdd(a.ticket, 1ull);
        }
        __syncthreads();
    }
    int64_t t;
    if (LB) { t = s_bcast[0]; if (t < 0) return; }
    else {
        t = a.tile_begin + xcd_tile();   // (neighbouring tiles on one XCD: bzq_device.hpp)
        if (t >= a.tile_end) return;
    }
    // every global LOAD of a tile is issued by fetch(), before any store of the same loop trip: vmcnt retires in
    // order, so a load waited for after stores would also wait for those stores' round trip
    uint4 r[4];   // four source pieces per thread (q = tid + 256 s)
    int64_t tP = 0, tS = 0, tQ = 0, tI = 0;
    int64_t fold_v = 0;   // fold: lanes 0..7 of wave 0 fetch one word each (FOLD_* below) and hand them over through s_fold
    uint32_t prevb = 10u;
    auto fetch = [&](int64_t tt) {
        const int64_t f0 = tt * TILE;
        if (!LB) { tP = a.tileP[tt]; tS = a.tileS[tt]; tQ = a.tileQ[tt]; tI = a.tileI[tt]; }
        prevb = f0 > 0 ? (uint32_t)a.g[f0 - 1] : a.prev_byte;
        tile_fetch(a.g, a.n, f0, (int)((a.n - f0) < TILE ? (a.n - f0) : TILE), r);
        if (FOLD && tid0 < 16) {
            // What the fold needs, one word per lane (F_* below), handed over through s_fold: the two batch bases the tile can need
            // (bb[2 tb - 2 .. 2 tb + 1]), the line prefix and the last-newline word of tile t - 1 -- VECTOR loads behind the tile's
            // own (they retire in order: nothing waits for them that does not wait for the tile anyway) -- and the kernel
            // arguments only the per-record outputs use.  Kept as uniform values all of this cost 100 spilled SGPRs in a kernel
            // that already uses every one of them (measured: emit + 2.5 %).
            const int64_t tb = a.tileB[tt];
            const int i = tid0;
            const int64_t* src = nullptr;
            if (i < 4) { const int64_t o = 2 * tb - 2 + i; if (o >= 0 && (o >> 1) < a.bb_cap) src = a.bb + o; }
            else if (i == 4) { if (tt > 0) src = a.tileP + (tt - 1); }
            else if (i == 5) { if (tt > 0) src = reinterpret_cast<const int64_t*>(a.tile_last) + (tt - 1); }
            const int64_t args[10] = {(tb + 1) * a.batch, tb, (int64_t)a.b_ends, (int64_t)a.b_id_ends, a.len_limit, a.first_header,
                                      (int64_t)a.tile_last, a.batch, (int64_t)a.bb, a.bb_cap};
            int64_t v = 0;
#pragma unroll
            for (int k = 0; k < 10; ++k) v = i == 6 + k ? args[k] : v;
            fold_v = i < 6 ? (src ? *src : 0) : v;
        }
    };
    fetch(t);
    const int tid = tid0, lane = tid & 63, wave = tid >> 6;
    (void)lane; (void)wave;   // (the look-back blocks of the EXPERIMENTS build)
    u64 tprev = 0;
    auto phase_mark = [&](int i) {
        if (BZQ_ABLATE(64) && tid == 0 && (t & 63) == 0) {   // 1 workgroup in 64 (all-workgroup atomics would dominate)
            const u64 now = __builtin_readcyclecounter();
            if (i >= 0) atomicAdd(&a.st->phase_cycles[i], now - tprev);
            tprev = now;
        }
    };
    phase_mark(-1);
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    const int64_t cP = tP, cS = tS, cQ = tQ, cI = tI;
    const bool first_starts = prevb == 10u;
    tile_stage<true>(r, valid, s_mask, s_tile);   // r[] also stays in registers for the scatter
    ByteSrc bs{a.g, a.n, a.prev_byte, s_tile, t0, valid};
    if (a.walk_limit > 0) bs.walk_limit = a.walk_limit;
    __syncthreads();
    phase_mark(0);   // tile loaded, masks built, staged
    if (!LB && BZQ_ABLATE(128)) return;   // experiment: stop here
    const u64* s_mask64 = reinterpret_cast<const u64*>(s_mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), s_w, c);
    const bool dense = ((int)c > MAXL) || a.force_dense;
    reinterpret_cast<uint32_t*>(&s_seg[0][tid])[0] = 0u; reinterpret_cast<uint32_t*>(&s_seg[1][tid])[0] = 0u;
    reinterpret_cast<uint32_t*>(&s_seg[2][tid])[0] = 0u;   // length 0 until the line pass fills it
    if (!dense) {   // line index at the first byte of each of this thread's four analysis pieces (bytes 64*tid + 16*i)
        const uint32_t l0 = excl, l1 = l0 + (uint32_t)__popc((uint32_t)m64 & 0xFFFFu),
                       l2 = l0 + (uint32_t)__popc((uint32_t)m64), l3 = l0 + (uint32_t)__popcll(m64 & 0xFFFFFFFFFFFFull);
        *reinterpret_cast<u64*>(&s_pline[4 * tid]) = (u64)l0 | ((u64)l1 << 16) | ((u64)l2 << 32) | ((u64)l3 << 48);
    }

#if BZQ_EXPERIMENTS
    // ---- look-back 1: line index of the tile's first line --------------------------------------
    if (LB && wave == 0) {
        if (lane == 0 && t > 0) st_agent(&a.desc_c[t], DESC_A | (u64)c);   // (the first tile of a chain publishes its prefix directly below)
        const int64_t run0 = a.xcd_tiles > 0 ? (t / a.xcd_tiles) * a.xcd_tiles : 0;   // first tile of the look-back chain
        const int64_t Pex = lookback_lines(a.desc_c + run0, t - run0, a.xcd_tiles > 0 ? a.xcd_base[4 * (run0 / a.xcd_tiles)] : a.st->P0, lane, a.st);
        if (lane == 0) {
            st_agent(&a.desc_c[t], DESC_P | (u64)(Pex + (int64_t)c + DESC_BIAS));
            s_bcast[1] = Pex;
        }
    }
#endif
    if (!dense) { // newline position table (other waves overlap this with wave 0's look-back)
        u64 m = m64;
        int idx = 0;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            s_nl[excl + idx] = (uint16_t)(tid * 64 + bit);
            ++idx;
        }
    }
    __syncthreads();
    phase_mark(1);   // newline count scan + position table
    if (!LB && BZQ_ABLATE(256)) return;   // experiment: stop here
    const int64_t P = LB ? s_bcast[1] : cP;
    ErrAcc err{~0ull, ~0ull};
    bool overflow = false, h_bad = false;
    const int ph = (int)(P & 3);
    // fold: the two batches a fast-path tile can touch (uniform loads), and who the tile's first record end is measured from
    constexpr bool fold = FOLD && !LB;
    u64 e_buf = ~0ull;
    // end of the record before the first one that ends in this tile (its quality line is local line (3 - ph) & 3)
    auto first_prev_end = [&](bool& too_long) -> int64_t {
        const int64_t lprev = P + (int64_t)((3 - ph) & 3) - 4;
        if (lprev < 0) return s_fold[F_FIRSTHDR] - 1;   // record 0: measured from its header (a shard's head lines end right before it)
        return prev_record_end(a.tileP, s_fold[F_TILELAST], t, lprev, s_fold[F_LENLIM], s_fold[F_PM1], (u64)s_fold[F_LASTM1], too_long);
    };

    // walks every line of the tile serially (any input): used by the dense path, count then emit
    auto dense_walk = [&](bool emit, int64_t S, int64_t Q, int64_t I, int64_t& ns, int64_t& nq, int64_t& ni) {
        int64_t rs = S, rq = Q, ri = I;
        int j = 0, line_start = 0;
        bool start_in = first_starts;
        // fold: batch of the record at hand (records come in order), end of the record before it
        int64_t dk = fold ? s_fold[F_KB0] : 0, dnext = fold ? s_fold[F_NEXTB] : 0, dbe = fold ? s_fold[0] : 0, dbi = fold ? s_fold[1] : 0, dprev = 0;
        g_i64* const dbb = as_global(fold ? s_fold[F_BB] : 0);
        g_i64* const d_be = as_global(fold ? s_fold[F_BENDS] : 0);
        g_i64* const d_bi = as_global(fold ? s_fold[F_BIDENDS] : 0);
        bool dlong = false, dfirst = true;
        auto batch_of = [&](int64_t rec) {
            while (rec >= dnext) { ++dk; dnext += s_fold[F_BATCH]; if (dk <= s_fold[F_BBCAP]) { dbe = dbb[2 * (dk - 1)]; dbi = dbb[2 * (dk - 1) + 1]; } }
        };
        auto handle = [&](int start, int end, bool end_in) {
            const int64_t L = P + j;
            const int role = (int)(L & 3);
            const int64_t rec = L >> 2;
            const int64_t ls = t0 + start, le = t0 + end;
            const bool sin = start_in && start < valid;
            if (role == 0) {
                if (emit && sin) {
                    if (s_tile[start] != 64) err.structure(rec, 1);
                    if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_hdr[rec] = ls;
                }
                int64_t lo = ls, hi = ls;
                if (end > start) header_kept(bs, ls, le, start_in, end_in, t0 + valid, lo, hi);
                if (emit && a.check_h && (hi - lo) != (int64_t)(end - start) - ((start_in && end > start) ? 1 : 0)) h_bad = true;
                if (emit)
                    for (int64_t p = lo; p < hi; ++p) {
                        const uint8_t ch = s_tile[p - t0];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (ri + (p - lo) >= 0) a.col_id[ri + (p - lo)] = ch;
                    }
                ri += hi - lo;
                if (emit && end_in && rec >= 0) {
                    if (rec < a.rec_cap) {
                        if (fold) { batch_of(rec); d_bi[rec] = ri - dbi; }
                        else a.id_ends[rec] = ri;
                    } else overflow = true;
                }
            } else if (role == 1) {
                if (emit && sin && OFFS && rec >= 0 && rec < a.rec_cap) a.o_seq[rec] = ls;
                if (emit)
                    for (int p = start; p < end; ++p) {
                        const uint8_t ch = s_tile[p];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (rs + (p - start) >= 0) a.col_seq[rs + (p - start)] = ch;
                    }
                rs += end - start;
            } else if (role == 2) {
                if (emit && sin) {
                    if (s_tile[start] != 43) err.structure(rec, 2);
                    if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_sep[rec] = ls;
                }
            } else {
                if (emit && sin && OFFS && rec >= 0 && rec < a.rec_cap) a.o_qual[rec] = ls;
                if (emit)
                    for (int p = start; p < end; ++p) {
                        const uint8_t ch = s_tile[p];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (CQ && (uint32_t)((ch - a.q_lower) & 0xFFu) > (a.q_upper - a.q_lower)) err.valid(rec, 5);
                        if (rq + (p - start) >= 0) a.col_qual[rq + (p - start)] = ch;
                    }
                rq += end - start;
                if (emit && end_in && fold) {   // (head lines of a shard too: the next record is measured from their end)
                    if (dfirst) { dprev = first_prev_end(dlong); dfirst = false; }
                    if (rec >= 0 && (dlong || le - dprev > s_fold[F_LENLIM]) && ((u64)rec << 3) < e_buf) e_buf = (u64)rec << 3;
                    dprev = le; dlong = false;
                }
                if (emit && end_in && rec >= 0) {
                    if (rec < a.rec_cap) {
                        a.rec_end[rec] = le;
                        if (fold) { batch_of(rec); d_be[rec] = rq - dbe; }
                        else a.ends[rec] = rq;
                    } else overflow = true;
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
        ns = rs - S; nq = rq - Q; ni = ri - I;
    };

    int n_id = 0, n_seq = 0, n_qual = 0;
    (void)n_id; (void)n_seq; (void)n_qual;
    uint32_t lh = 0, lsq = 0, lq = 0;   // this thread's segment (k = tid) of each role: kept length ...
    int dh = 0, ds = 0, dq = 0;         // ... and offset within the tile's part of the column
    if (dense) {
        if (LB) {
            if (tid == 0) {
                int64_t ns, nq, ni;
                dense_walk(false, 0, 0, 0, ns, nq, ni);
                s_cnt[0] = (int)ni; s_cnt[1] = (int)ns; s_cnt[2] = (int)nq;
            }
            __syncthreads();
            n_id = s_cnt[0]; n_seq = s_cnt[1]; n_qual = s_cnt[2];
        }
    } else {
        // ---- line pass: one line per thread (j = tid, tid+256, ...), so a 150 bp tile (~207 lines)
        // keeps all four waves busy; line j has role (ph+j)&3 and is segment k = j>>2 of that role
        for (int j = tid; j <= (int)c; j += BLOCK) {
            const int role = (ph + j) & 3;
            const int k = j >> 2;
            const int start = j ? (int)s_nl[j - 1] + 1 : 0;
            const bool end_in = j < (int)c;
            const int end = end_in ? (int)s_nl[j] : valid;
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
                if (a.check_h && (hi - lo) != (int64_t)(end - start) - ((sknown && end > start) ? 1 : 0)) h_bad = true;
                reinterpret_cast<uint32_t*>(&s_seg[0][k])[0] = (uint32_t)(lo - t0) | ((uint32_t)(hi - lo) << 16);
            } else if (role == 2) {
                if (sin) {
                    if (s_tile[start] != 43) err.structure(rec, 2);   // '+', utils.mojo:456
                    if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_sep[rec] = ls;
                }
            } else {
                const int slot = role == 1 ? 1 : 2;
                if (sin && OFFS && rec >= 0 && rec < a.rec_cap) (role == 1 ? a.o_seq : a.o_qual)[rec] = ls;
                reinterpret_cast<uint32_t*>(&s_seg[slot][k])[0] = (uint32_t)start | ((uint32_t)(end - start) << 16);
            }
        }
        __syncthreads();
        phase_mark(2);   // line pass
    if (!LB && BZQ_ABLATE(512)) return;   // experiment: stop here
        const uint32_t g0 = reinterpret_cast<const uint32_t*>(&s_seg[0][tid])[0], g1 = reinterpret_cast<const uint32_t*>(&s_seg[1][tid])[0],
                       g2 = reinterpret_cast<const uint32_t*>(&s_seg[2][tid])[0];
        lh = g0 >> 16; lsq = g1 >> 16; lq = g2 >> 16;
        const u64 packed = (u64)lh | ((u64)lsq << 21) | ((u64)lq << 42);
        u64 tot = 0;
        const u64 ex = block_exclusive_scan<u64, 4>(packed, s_w64, tot);
        dh = (int)(ex & 0x1FFFFFull); ds = (int)((ex >> 21) & 0x1FFFFFull); dq = (int)((ex >> 42) & 0x1FFFFFull);
        reinterpret_cast<int32_t*>(&s_seg[0][tid])[1] = dh - (int)(g0 & 0xFFFFu);
        reinterpret_cast<int32_t*>(&s_seg[1][tid])[1] = ds - (int)(g1 & 0xFFFFu);
        reinterpret_cast<int32_t*>(&s_seg[2][tid])[1] = dq - (int)(g2 & 0xFFFFu);
        if (!LB && tid < 4)
            s_colbase[tid] = (u64)(tid == 0 ? a.col_id + cI : (tid == 1 ? a.col_seq + cS : a.col_qual + cQ));
        n_id = (int)(tot & 0x1FFFFFull); n_seq = (int)((tot >> 21) & 0x1FFFFFull); n_qual = (int)((tot >> 42) & 0x1FFFFFull);
    }

#if BZQ_EXPERIMENTS
    // ---- look-back 2: column offsets --------------------------------------------------------------
    if (LB && wave == 0) {
        if (lane == 0 && t > 0)
            st_agent(&a.desc_agg[t], DESC_A | (u64)n_seq | ((u64)n_qual << 20) | ((u64)n_id << 40));
        const int64_t run0 = a.xcd_tiles > 0 ? (t / a.xcd_tiles) * a.xcd_tiles : 0;
        const int64_t* xb = a.xcd_tiles > 0 ? a.xcd_base + 4 * (run0 / a.xcd_tiles) : nullptr;
        const Cols ex = lookback_cols(a.desc_agg + run0, a.desc_pre + 3 * run0, t - run0, xb ? xb[1] : a.st->S0, xb ? xb[2] : a.st->Q0, xb ? xb[3] : a.st->I0, lane, a.st);
        if (lane == 0) {
            st_agent(&a.desc_pre[3 * t], DESC_P | (u64)(ex.s + n_seq + DESC_BIAS));
            st_agent(&a.desc_pre[3 * t + 1], DESC_P | (u64)(ex.q + n_qual + DESC_BIAS));
            st_agent(&a.desc_pre[3 * t + 2], DESC_P | (u64)(ex.d + n_id + DESC_BIAS));
            s_bcast[1] = ex.s; s_bcast[2] = ex.q; s_bcast[3] = ex.d;
            if (t == a.n_tiles - 1) { // totals for the host
                a.st->P = P + (int64_t)c; a.st->S = ex.s + n_seq; a.st->Q = ex.q + n_qual; a.st->I = ex.d + n_id;
            }
            if (c > 0) atomicMax((long long*)&a.st->last_nl_tile, (long long)t);
        }
    }
#endif
    // (as late as the barrier in front of the first reader allows: the loads behind fold_v were issued one scalar-load latency
    // after the tile's, and wave 0 must not wait for them while the other waves wait for wave 0)
    if (FOLD && tid0 < 16) s_fold[tid0] = fold_v;
    __syncthreads();
    const int64_t S = LB ? s_bcast[1] : cS, Q = LB ? s_bcast[2] : cQ, I = LB ? s_bcast[3] : cI;
    phase_mark(3);   // segment scan
    if (!LB && BZQ_ABLATE(1024)) return;   // experiment: stop here

    if (dense) {
        if (tid == 0) {
            int64_t ns, nq, ni;
            dense_walk(true, S, Q, I, ns, nq, ni);
            atomicAdd((u64*)&a.st->dense_tiles, 1ull);
        }
    } else {
        // ---- per-record outputs of lines that END in this tile --------------------------------------
        const int jh = (0 - ph) & 3, jq = (3 - ph) & 3;
        if (LB) {   // column offsets only known now
            if (tid < 4) s_colbase[tid] = (u64)(tid == 0 ? a.col_id + I : (tid == 1 ? a.col_seq + S : a.col_qual + Q));
            __syncthreads();
        }
        {
            const int j = 4 * tid + jh;           // this thread's header line
            const int64_t rec = (P + j) >> 2;
            if (j < (int)c && rec >= 0) {
                if (rec < a.rec_cap) {
                    const int64_t ie = I + (int64_t)(dh + (int)lh);
                    if (fold) as_global(s_fold[F_BIDENDS])[rec] = ie - s_fold[rec >= s_fold[F_NEXTB] ? 3 : 1];
                    else a.id_ends[rec] = ie;
                } else overflow = true;
            }
        }
        {
            const int j = 4 * tid + jq;           // this thread's quality line
            const int64_t rec = (P + j) >> 2;
            if (j < (int)c && rec >= 0) {
                const int64_t qe = Q + (int64_t)(dq + (int)lq);
                // sequence bytes up to and including this record's sequence line (line j-2): with this thread's
                // sequence segment when that line is in the same group of four, else everything before it
                const int64_t se = S + (int64_t)(jq >= 2 ? ds + (int)lsq : ds);
                if (rec < a.rec_cap) {
                    a.rec_end[rec] = t0 + (int64_t)s_nl[j];
                    if (fold) as_global(s_fold[F_BENDS])[rec] = qe - s_fold[rec >= s_fold[F_NEXTB] ? 2 : 0];
                    else a.ends[rec] = qe;
                } else overflow = true;
                if (se != qe) err.structure(rec, 3); // utils.mojo:458-461 as a cumulative test
                if (fold) {   // header_start .. '\n' inclusive against the reference's buffer limit (parser.mojo:484-492)
                    bool too_long = false;
                    const int64_t prev = j >= 4 ? t0 + (int64_t)s_nl[j - 4] : first_prev_end(too_long);   // (j < 4: thread 0 only)
                    if (too_long || t0 + (int64_t)s_nl[j] - prev > s_fold[F_LENLIM]) e_buf = (u64)rec << 3;
                }
            }
        }
        phase_mark(4);   // record outputs
        // ---- scatter --------------------------------------------------------------------------------------
        // lines with a negative index belong to the previous shard (their column offsets are below 0): not written
        const int jmin = P < 0 ? (int)(-P) : 0;
        if (!BZQ_ABLATE(1)) {
            // whole 16-byte source pieces that lie inside one kept line, straight from the registers they
            // were loaded into (consecutive lanes -> consecutive destination bytes within a line)
            if (!BZQ_ABLATE(4)) {
                uint32_t jj[4];
                u64 sg[4], cb[4];
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) jj[sidx] = s_pline[tid + BLOCK * sidx];
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const uint32_t role = ((uint32_t)ph + jj[sidx]) & 3u;
                    sg[sidx] = s_seg[role < 2u ? role : 2u][jj[sidx] >> 2];
                    cb[sidx] = s_colbase[role];
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
                        int64_t off = (int64_t)(delta + pos);
                        if (BZQ_ABLATE(8)) off &= 0xFFFFF;
                        const U16B v{pv.x, pv.y, pv.z, pv.w};
                        if (!BZQ_ABLATE(16)) *reinterpret_cast<U16B*>(reinterpret_cast<uint8_t*>(cb[sidx]) + off) = v;
                    }
                }
            }
            phase_mark(5);   // whole-piece stores
            // line heads and tails: a line of >= 16 bytes gets its FIRST 16 and its LAST 16 bytes copied whole
            // (one unaligned LDS read + one unaligned 16-byte store each); with the whole pieces above that
            // covers every byte, and bytes written twice carry the same value.  Shorter lines go byte-exact.
            if (!BZQ_ABLATE(2)) {
                const int nseg = ((int)c + 4) >> 2; // segment indices in use
                for (int pidx = tid; pidx < 6 * nseg; pidx += BLOCK) {
                    const int side = pidx & 1, sk = pidx >> 1;
                    const int k = sk / 3, slot = sk - 3 * k;
                    const u64 sg = s_seg[slot][k];
                    const int len = (int)((uint32_t)sg >> 16);
                    const int role = slot == 2 ? 3 : slot;
                    const int j = 4 * k + ((role - ph) & 3);
                    if (len == 0 || j < jmin) continue;
                    const int src = (int)((uint32_t)sg & 0xFFFFu), delta = (int)(sg >> 32);
                    uint8_t* col = reinterpret_cast<uint8_t*>(s_colbase[role]);
                    const int64_t rec = (P + j) >> 2;
                    if (len >= 16) {
                        const int off = side ? src + len - 16 : src;
                        if (((side ? src + len : src) & 15) == 0) continue;   // that end is a whole piece already
                        int64_t g = (int64_t)(delta + off);
                        if (BZQ_ABLATE(8)) g &= 0xFFFFF;
                        copy16<CA, CQ>(col, g, off, s_tile, rec, slot == 2, a.q_lower, a.q_upper, err);
                    } else if (side == 0) {
                        emit_part_rt<CA, CQ>(col, (int64_t)(delta + src), src, len, s_tile, rec, slot == 2, a.q_lower, a.q_upper, err);
                    }
                }
            }
        }
    }
    phase_mark(6);   // junctions
    if (err.e_struct != ~0ull) atomicMin(&a.st->err_struct, err.e_struct);
    if (err.e_valid != ~0ull) atomicMin(&a.st->err_valid, err.e_valid);
    if (e_buf != ~0ull) atomicMin(&a.st->err_buf, e_buf);
    if (overflow) atomicOr(&a.st->rec_overflow, 1);
    if (h_bad) a.st->lookback_timeout = 2;   // an id lost bytes to the strip: pass A's hypothesis was wrong, the host repeats the chunk with the exact pass A
}


// Pass A of the two-pass mode.  The phase (which line of a record a tile starts on) is unknown here, so every line
// is measured as if it were a header and the byte counts are kept per CLASS (line index mod 4); the scan kernels
// pick the class that turns out to be the header / sequence / quality one.  One workgroup per tile, one line per
// thread; 20 KiB of LDS and < 64 VGPRs so that eight workgroups fit a CU (this pass lives on occupancy: its time
// falls 1.09 / 0.89 / 0.78 / 0.70 ms at 3 / 4 / 5 / 6 workgroups per CU).  Same output as k_tile_aggregate.
constexpr int MAXL_A = 1012;   // s_nl sized so the kernel's LDS is 8 x 20480 B per CU; more newlines -> serial path
static __global__ __launch_bounds__(BLOCK) void k_tile_aggregate2(AggArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[TILE];
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];
    __shared__ uint16_t s_nl[MAXL_A];
    __shared__ __attribute__((aligned(8))) uint32_t s_w[4];   // scan scratch, then the two block sums (2 x u64)
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t t = a.tile_begin + xcd_tile();
    if (t >= a.tile_end) return;
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    const uint32_t prev_b = t0 > 0 ? (uint32_t)a.g[t0 - 1] : a.prev_byte;
    uint4 r[4];
    tile_fetch<true>(a.g, a.n, t0, valid, r);
    tile_stage<true>(r, valid, s_mask, s_tile);
    ByteSrc bs{a.g, a.n, a.prev_byte, s_tile, t0, valid};
    if (a.walk_limit > 0) bs.walk_limit = a.walk_limit;
    const bool first_starts = (prev_b == 10u);
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(s_mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), s_w, c);
    u64* s_sum = reinterpret_cast<u64*>(s_w);
    if (tid < 2) s_sum[tid] = 0;
    u64 pa = 0, pi = 0; // 4 x 16-bit fields: bytes / id bytes per class
    if ((int)c <= MAXL_A) {
        u64 m = m64;
        int idx = 0;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            s_nl[excl + idx] = (uint16_t)(tid * 64 + bit);
            ++idx;
        }
        __syncthreads();
        if (tid == 0 && a.tile_last) a.tile_last[t] = last_by_class(s_nl, (int)c);
        for (int j = tid; j <= (int)c; j += BLOCK) {
            const int start = j ? (int)s_nl[j - 1] + 1 : 0;
            const bool end_in = j < (int)c;
            const int end = end_in ? (int)s_nl[j] : valid;
            if (end > start) {
                int64_t lo, hi;
                header_kept(bs, t0 + start, t0 + end, j > 0 ? true : first_starts, end_in, t0 + valid, lo, hi);
                pa += (u64)(end - start) << (16 * (j & 3));
                pi += (u64)(hi - lo) << (16 * (j & 3));
            }
        }
    } else {
        __syncthreads();
        if (tid == 0) { // serial path for tiles with > MAXL_A newlines
            int j = 0, line_start = 0;
            bool start_in = first_starts;
            uint32_t la[4] = {0, 0, 0, 0}, li[4] = {0, 0, 0, 0};
            u64 lastw = ~0ull;
            auto handle = [&](int start, int end, bool end_in) {
                if (end_in) lastw = (lastw & ~(0xFFFFull << (16 * (j & 3)))) | ((u64)end << (16 * (j & 3)));
                if (end > start) {
                    int64_t lo, hi;
                    header_kept(bs, t0 + start, t0 + end, start_in, end_in, t0 + valid, lo, hi);
                    la[j & 3] += (uint32_t)(end - start);
                    li[j & 3] += (uint32_t)(hi - lo);
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
            if (a.tile_last) a.tile_last[t] = lastw;
            for (int k = 0; k < 4; ++k) { pa |= (u64)la[k] << (16 * k); pi |= (u64)li[k] << (16 * k); }
        }
    }
    // block sum of the packed fields (every field total <= 16384, no carry between fields)
    pa = wave_sum_u64(pa); pi = wave_sum_u64(pi);
    if (lane == 0) { atomicAdd(&s_sum[0], pa); atomicAdd(&s_sum[1], pi); }
    __syncthreads();
    if (tid == 0) {
        a.tile_c[t] = c;
        a.tile_a[t] = s_sum[0];
        a.tile_idc[t] = s_sum[1];
    }
}

// Pass A from the newline bitmap ALONE (the default): no LDS copy of the bytes, no byte lookups -- line lengths per class come
// out of the newline positions, and the id bytes per class under the hypothesis that no header line loses bytes to
// _strip_spaces (utils.mojo:221-242): kept = length - 1 for a line that starts in the tile ('@' dropped), the whole piece for
// the continuation of a line that started earlier.  Real FASTQ ids have no leading / trailing posix spaces, so this is exact;
// the emit kernel measures every header line exactly (header_kept) anyway and flags the chunk when a line contradicts the
// hypothesis (FusedArgs::check_h) -- the host then repeats it with k_tile_aggregate2.  4 KiB of LDS instead of 20, half the
// instructions: the pass runs at the rate of a bare read of the input.  Same output format as k_tile_aggregate2.
static __global__ __launch_bounds__(BLOCK) void k_tile_aggregate_h(AggArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];
    __shared__ uint16_t s_nl[MAXL_A];
    __shared__ __attribute__((aligned(8))) uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t t = a.tile_begin + xcd_tile();
    if (t >= a.tile_end) return;
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    const uint32_t prev_b = t0 > 0 ? (uint32_t)a.g[t0 - 1] : a.prev_byte;
    uint4 r[4];
    tile_fetch<true>(a.g, a.n, t0, valid, r);
    tile_stage<false>(r, valid, s_mask, nullptr);
    const bool first_starts = (prev_b == 10u);
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(s_mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), s_w, c);
    u64* s_sum = reinterpret_cast<u64*>(s_w);
    if (tid < 2) s_sum[tid] = 0;
    u64 pa = 0, pi = 0; // 4 x 16-bit fields: bytes / id bytes per class
    auto add_line = [&](int j, int start, int end) {
        const int len = end - start;
        if (len <= 0) return;
        pa += (u64)len << (16 * (j & 3));
        pi += (u64)(len - ((j > 0 || first_starts) ? 1 : 0)) << (16 * (j & 3));
    };
    if ((int)c <= MAXL_A) {
        u64 m = m64;
        int idx = 0;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            s_nl[excl + idx] = (uint16_t)(tid * 64 + bit);
            ++idx;
        }
        __syncthreads();
        if (tid == 0 && a.tile_last) a.tile_last[t] = last_by_class(s_nl, (int)c);
        for (int j = tid; j <= (int)c; j += BLOCK) add_line(j, j ? (int)s_nl[j - 1] + 1 : 0, j < (int)c ? (int)s_nl[j] : valid);
    } else {
        __syncthreads();
        if (tid == 0) {   // more newlines than the table holds: one thread walks the bitmap
            int j = 0, line_start = 0;
            u64 lastw = ~0ull;
            for (int w = 0; w < BLOCK; ++w) {
                u64 m = s_mask64[w];
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    add_line(j, line_start, w * 64 + bit);
                    lastw = (lastw & ~(0xFFFFull << (16 * (j & 3)))) | ((u64)(w * 64 + bit) << (16 * (j & 3)));
                    line_start = w * 64 + bit + 1;
                    ++j;
                }
            }
            add_line(j, line_start, valid);
            if (a.tile_last) a.tile_last[t] = lastw;
        }
    }
    pa = wave_sum_u64(pa); pi = wave_sum_u64(pi);
    if (lane == 0) { atomicAdd(&s_sum[0], pa); atomicAdd(&s_sum[1], pi); }
    __syncthreads();
    if (tid == 0) {
        a.tile_c[t] = c;
        a.tile_a[t] = s_sum[0];
        a.tile_idc[t] = s_sum[1];
    }
}


// ---- batch bases (FusedArgs::fold) ---------------------------------------------------------------------------------------------
// FastqBatch._ends / _id_ends restart at every batch (record_batch.mojo:77-87 via parser.mojo:243): batch k's arrays are the
// chunk's running sums minus their values at the last record of batch k - 1.  Those <= n_records / batch_size values are known
// BEFORE the emit: record R = k * batch - 1 ends at the newline of line 4 R + 3, the tile prefixes of the scan say which tile
// holds that newline and what the columns hold before the tile, and one look at that tile gives the rest -- the quality bytes and
// the kept id bytes of the tile in front of the newline, measured exactly as the emit measures them (header_kept).  One workgroup
// per boundary (2441 of them for 10 M records in batches of 4096: 40 MB read), so that the emit kernel can write the per-batch
// arrays itself and the separate pass over the per-record arrays (k_rebase: 0.10 ms of a 1.77 ms step) is gone.
struct BasesArgs {
    const uint8_t* g;
    int64_t n;
    uint32_t prev_byte;
    int64_t n_tiles;
    const int64_t* tileP;
    const int64_t* tileQ;
    const int64_t* tileI;
    int64_t batch;
    int64_t* bb;
    int64_t bb_cap;
    const ChunkState* st;
    int64_t walk_limit;
    const int64_t* btile;   // ScanArgs::btile
};

static __global__ __launch_bounds__(BLOCK) void k_batch_bases(BasesArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[TILE];
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];
    __shared__ uint16_t s_nl[MAXL_A];
    __shared__ __attribute__((aligned(8))) uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t k = (int64_t)blockIdx.x + 1;
    const int64_t lines = a.st->P;
    const int64_t n_rec = lines > 0 ? (lines >> 2) : 0;
    const int64_t R = k * a.batch - 1;
    if (R >= n_rec || k - 1 >= a.bb_cap) return;
    const int64_t lt = 4 * R + 3;   // line whose newline ends record R
    const int64_t t = a.btile[k - 1];   // (the scan noted which tile holds that newline: ScanArgs::btile)
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    const uint32_t prev_b = t0 > 0 ? (uint32_t)a.g[t0 - 1] : a.prev_byte;
    uint4 r[4];
    tile_fetch(a.g, a.n, t0, valid, r);
    tile_stage<true>(r, valid, s_mask, s_tile);
    ByteSrc bs{a.g, a.n, a.prev_byte, s_tile, t0, valid};
    if (a.walk_limit > 0) bs.walk_limit = a.walk_limit;
    const bool first_starts = (prev_b == 10u);
    const int64_t P = a.tileP[t];
    const int ph = (int)(P & 3);
    const int jt = (int)(lt - P);   // the tile's local line that ends record R
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(s_mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), s_w, c);
    u64* s_sum = reinterpret_cast<u64*>(s_w);
    if (tid < 2) s_sum[tid] = 0;
    u64 q = 0, d = 0;   // quality bytes / kept id bytes of the tile in front of that newline
    if ((int)c <= MAXL_A) {
        u64 m = m64;
        int idx = 0;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            s_nl[excl + idx] = (uint16_t)(tid * 64 + bit);
            ++idx;
        }
        __syncthreads();
        for (int j = tid; j <= jt && j < (int)c; j += BLOCK) {
            const int role = (ph + j) & 3;
            const int start = j ? (int)s_nl[j - 1] + 1 : 0, end = (int)s_nl[j];
            if (role == 3) q += (u64)(end - start);
            else if (role == 0 && end > start) {
                int64_t lo, hi;
                header_kept(bs, t0 + start, t0 + end, j > 0 ? true : first_starts, true, t0 + valid, lo, hi);
                d += (u64)(hi - lo);
            }
        }
    } else {
        __syncthreads();
        if (tid == 0) {   // more newlines than the table holds: one thread walks the bitmap
            int j = 0, line_start = 0;
            bool start_in = first_starts;
            for (int w = 0; w < BLOCK && j <= jt; ++w) {
                u64 m = s_mask64[w];
                while (m && j <= jt) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const int nl = w * 64 + bit, role = (ph + j) & 3;
                    if (role == 3) q += (u64)(nl - line_start);
                    else if (role == 0 && nl > line_start) {
                        int64_t lo, hi;
                        header_kept(bs, t0 + line_start, t0 + nl, start_in, true, t0 + valid, lo, hi);
                        d += (u64)(hi - lo);
                    }
                    line_start = nl + 1;
                    start_in = true;
                    ++j;
                }
            }
        }
    }
    q = wave_sum_u64(q); d = wave_sum_u64(d);
    if (lane == 0) { atomicAdd(&s_sum[0], q); atomicAdd(&s_sum[1], d); }
    __syncthreads();
    if (tid == 0) {
        a.bb[2 * (k - 1)] = a.tileQ[t] + (int64_t)s_sum[0];
        a.bb[2 * (k - 1) + 1] = a.tileI[t] + (int64_t)s_sum[1];
    }
}

// What is left of k_rebase with FusedArgs::fold, for the re-run paths (the first run does it inside k_tail): one thread.
static __global__ void k_finish(ChunkFinishArgs f, ChunkState* st) {
    if (threadIdx.x || blockIdx.x) return;
    chunk_finish(f, st);
}

} // namespace bzq
