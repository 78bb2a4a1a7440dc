// bzq_device.hpp -- gfx950 (CDNA4, wave64) kernels of the FASTQ batch-parse path.
//
// What this replaces in the reference (paths relative to the BlazeSeq tree):
//   _scan_record + _validate_fastq_structure   blazeseq/utils.mojo:448-551
//   _strip_spaces / is_posix_space             blazeseq/utils.mojo:221-242, 267-289
//   Validator._validate (ascii, quality)       blazeseq/fastq/record.mojo:76-116, 162-172
//   FastqBatch.add (SoA columns + ends)        blazeseq/fastq/record_batch.mojo:77-87
//   stage_batch_to_host/move_staged_to_device  blazeseq/fastq/record_batch.mojo:308-411
//
// Formulation (not a translation of the record-at-a-time CPU loop): the chunk starts at a record
// start, so the k-th '\n' ends line k, line k has role k&3 (0 header, 1 sequence, 2 '+', 3 quality)
// and belongs to record k>>2 -- strict 4-line framing, exactly what the reference does (it never
// re-synchronises on '@').  The chunk is cut into 16 KiB tiles, one 256-thread workgroup each:
//
//   k_tile_aggregate  pass A: per tile, phase-agnostic summary {newlines, bytes per line class
//                     (line index mod 4), id bytes per class after strip}.  Pure streaming read.
//   k_scan_tiles      exclusive scan of the summaries -> per tile {line index, seq/qual/id column
//                     offsets}.  One workgroup, tiny data.
//   k_tile_emit       pass B: re-reads the tile (coalesced 16 B/lane) into LDS, rebuilds the line
//                     table, and gathers the sequence / quality / id streams out of LDS into the
//                     packed columns with aligned, coalesced 16 B stores; writes ends / id_ends /
//                     record_end per record; structure + ascii + quality checks feed a
//                     min-reduction on (record << 3 | code) = "first failing record".
//   k_rebase          per-batch `ends` restart + longest-record check (BUFFER_EXCEEDED parity).
//
// No MFMA: byte/integer work, HBM-bound.  Algorithmic traffic per record = B + 2L + D + 16
// (DESIGN.md); this two-read design moves 2B + 2L + D + ~40.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The first-generation two-pass kernels (k_tile_aggregate / k_tile_emit) and the single-launch variants are kept as
// independent implementations for cross-checks; they are compiled only into an EXPERIMENTS build (make EXPERIMENTS=1 ->
// libblazeseq_hip_exp.so), never into the product library.
#ifndef BZQ_EXPERIMENTS
#define BZQ_EXPERIMENTS 0
#endif

namespace bzq {

constexpr int TILE = 16384;         // bytes per workgroup tile
constexpr int BLOCK = 256;          // 4 waves of 64
constexpr int PIECES = TILE / 16;   // 16-byte pieces per tile
constexpr int MAXL = 1020;          // fast path handles tiles with <= MAXL newlines (<= 256 lines per role)
constexpr int SEGS = 256;

typedef unsigned long long u64;

// Device-resident per-chunk state (host reads it back once per chunk).
struct ChunkState {
    // initial prefix (host)
    int64_t P0, S0, Q0, I0;
    // running carry of the tile scan (scan kernel), = totals after the last pass
    int64_t P, S, Q, I;
    int64_t last_nl_tile;     // last tile that holds a newline, -1 if none
    int64_t tail_start;       // offset after the last newline (k_tail)
    int32_t tail_nonblank;    // tail holds a byte outside {\n,\r,space,tab} (utils.mojo:311-322)
    int32_t rec_overflow;     // a record index exceeded the per-record array capacity
    u64 err_struct;           // min (record<<3 | code) over codes 1..3
    u64 err_valid;            // min over codes 4..5
    u64 err_buf;              // min (record<<3) over records longer than the reference buffer limit
    int64_t dense_tiles;      // tiles that took the serial path (diagnostics)
    int64_t max_record_len;
    int64_t first_nl[4];      // offsets of the first four newlines (shard stitch), -1 if absent
    int32_t lookback_timeout; // single-pass kernel gave up waiting for a predecessor tile (never expected)
    int32_t _pad;
    // written by k_rebase (one D2H of this struct is all the host needs in the common case)
    int64_t n_complete;       // records with all four newlines
    int64_t last_record_end;  // record_end[n_complete-1], or first_header-1
    int64_t last_ends, last_id_ends;
    // debug only (option ablate bit 64): shader-clock cycles workgroup thread 0 spent in each phase of the emit kernel
    unsigned long long phase_cycles[12];
    // running totals between the passes of one chunk (pass_bytes): P, S, Q, I, last tile with a newline + 1;
    // pass k reads slot k & 1 and leaves slot (k + 1) & 1
    int64_t pass_carry[2][5];
    // views mode: pool slots handed to tiles with more newlines than an ordinary entry slot holds; views_fallback: the
    // pool ran out (a chunk of records of a few bytes), the host repeats the chunk on the byte-level kernels
    unsigned long long listed_tiles;
    int32_t views_fallback;
    // shard scan: the shard's first and last byte (k_first_newlines), so that the summary is one copy back
    uint8_t edge_first, edge_last, _pad2[2];
};

__device__ __forceinline__ bool is_posix_space(uint32_t c) {
    // blazeseq/utils.mojo:267-289: {9,10,11,12,13,28,29,30,32}
    return c <= 32u && ((0x170003E00ull >> c) & 1ull);
}

// 16-bit newline mask of 16 bytes.  v_perm_b32 with the DATA as the selector: selector byte 12 yields 0x00 and,
// with both sources all-ones, every other selector value yields 0xFF (0-7 pick source bytes, 8-11 their sign
// bits, >= 13 the constant 0xFF) -- so (x ^ 0x06) used as the selector is 0x00 exactly where x == '\n' and -1
// elsewhere.  A signed dot4 with weights 1,2,4,8 | 16,32,64,-128 and a start value of 127 then accumulates the
// flags of two dwords straight into mask bits (bit 7 comes out inverted and is flipped at the end).
__device__ __forceinline__ uint32_t nl_flags(uint32_t x) {
    return __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, x ^ 0x06060606u);
}
__device__ __forceinline__ uint32_t nl_mask16(uint4 v) {
    int lo = __builtin_amdgcn_sdot4((int)nl_flags(v.x), 0x08040201, 127, false);
    lo = __builtin_amdgcn_sdot4((int)nl_flags(v.y), (int)0x80402010, lo, false);
    int hi = __builtin_amdgcn_sdot4((int)nl_flags(v.z), 0x08040201, 127, false);
    hi = __builtin_amdgcn_sdot4((int)nl_flags(v.w), (int)0x80402010, hi, false);
    return (((uint32_t)hi << 8) | (uint32_t)lo) ^ 0x8080u;
}

// 16 bytes from global memory at ANY byte address, in one global_load_dwordx4 (gfx950 loads unaligned).  Typed as
// unaligned on purpose: a chunk handed over by the ingest pipeline starts wherever its carry starts, and behind an
// aligned vector type the compiler would be entitled to assume otherwise.
// NT = non-temporal.  Worth it in the kernels that ONLY read the input (pass A, the views line pass, FASTA pass 1: 0.527 ->
// 0.47 ms for the FASTQ pass A, same-box A/B; a tile-shaped copy probe, scripts/probes/copy_probe.hip, shows the same: 6.23 TB/s
// against 5.89).  NOT in the emit kernels: there nt loads cost 5 % (1.23 -> 1.30 ms) and slow the following k_rebase.
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef v4u32 v4u32_any __attribute__((aligned(1)));
template <bool NT = false>
__device__ __forceinline__ uint4 load16_any(const uint8_t* __restrict__ p) {
    v4u32 v;
    if (NT) v = __builtin_nontemporal_load(reinterpret_cast<const v4u32_any*>(p));
    else v = *reinterpret_cast<const v4u32_any*>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Guarded 16-byte load of chunk bytes [pos, pos+16): bytes at or beyond n read as 0.  The one piece that straddles
// n is taken from the 16 bytes that END at n and shifted down (no byte-by-byte path in the hot kernels).
__device__ __forceinline__ uint4 load16(const uint8_t* __restrict__ g, int64_t pos, int64_t n) {
    if (pos + 16 <= n) return load16_any(g + pos);
    u64 lo = 0, hi = 0;
    if (pos < n) {
        if (n >= 16) {
            // an UNALIGNED load and typed as one: behind a uint4* the compiler may assume n % 16 == 0
            struct __attribute__((packed, aligned(1))) Tail16 { u64 lo, hi; };
            const Tail16 v = *reinterpret_cast<const Tail16*>(g + (n - 16));
            const u64 vlo = v.lo, vhi = v.hi;
            const int sh = (int)(pos + 16 - n) * 8;   // 8..120: drop the bytes before pos
            if (sh < 64) { lo = (vlo >> sh) | (vhi << (64 - sh)); hi = vhi >> sh; }
            else lo = vhi >> (sh - 64);
        } else {   // a chunk shorter than 16 bytes
#pragma unroll 1
            for (int i = 0; pos + i < n; ++i) {
                const u64 b = (u64)g[pos + i];
                if (i < 8) lo |= b << (8 * i); else hi |= b << (8 * (i - 8));
            }
        }
    }
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

// views mode (bzq_views.hpp): the ticket of the entry pool.  Pass A draws from it, the scan behind it moves the count into the chunk
// state and leaves it zero -- so pass A needs nothing initialised by the submit and can start before the state's initial values arrive
struct ViewsPool {
    unsigned long long listed;
    int32_t fallback, _pad;
};

struct ByteSrc {
    const uint8_t* __restrict__ g;
    int64_t n;
    uint32_t prev_byte; // the byte before offset 0 ('\n' for a chunk that starts at a record start)
    const uint8_t* lds; // this tile staged in LDS (bytes [t0, t0+valid)), or nullptr
    int64_t t0;
    int valid;
    // header_kept never walks further than this from the tile: a space run longer than the reference's buffer limit lies in
    // a record the reference refuses (BUFFER_EXCEEDED / AT_MAX, caught by the length check in k_rebase) -- none of its bytes
    // is delivered, so any answer will do as long as both passes give the same one
    int64_t walk_limit = (int64_t)1 << 62;
    __device__ __forceinline__ uint32_t at(int64_t pos) const {
        const int64_t d = pos - t0;
        if (lds && d >= 0 && d < valid) return lds[d];
        if (pos >= 0) return g[pos];
        return pos == -1 ? prev_byte : 10u;
    }
};

// Kept (post-strip, '@' excluded) part of the header-line bytes [ls, le) that lie in this tile.
// The line really spans [h, e) with h <= ls and e >= le; start_known: ls == h (the '@' position is
// ls), end_known: le == e (the newline is at le).  Bytes are classified by where they ARE, so a
// leading/trailing space run that crosses a tile edge is looked at from both sides and every byte
// is dropped exactly once.  Follows _strip_spaces, blazeseq/utils.mojo:221-242 (all-space ids
// collapse to empty).
__device__ inline void header_kept(const ByteSrc& b, int64_t ls, int64_t le, bool start_known,
                                   bool end_known, int64_t tile_end, int64_t& lo_out, int64_t& hi_out) {
    int64_t lo = ls, hi = le;
    if (hi <= lo) { lo_out = lo; hi_out = lo; return; }
    bool in_lead;
    if (start_known) {
        lo = ls + 1; // '@' (or whatever sits at header_start) is never part of the id
        in_lead = true;
    } else {
        // does the leading-space run that began right after header_start reach this tile?
        int64_t p = ls - 1;
        in_lead = false;
        for (;;) {
            if (ls - p > b.walk_limit) break;                    // see ByteSrc::walk_limit
            uint32_t c = b.at(p);
            if (c == 10u) { in_lead = true; break; }           // p+1 was header_start and a space
            if (!is_posix_space(c)) {                            // non-space: only fine at header_start
                in_lead = (b.at(p - 1) == 10u);
                break;
            }
            --p;
        }
    }
    if (in_lead)
        while (lo < hi && is_posix_space(b.at(lo))) ++lo;
    if (hi > lo && is_posix_space(b.at(hi - 1))) {
        bool trailing = end_known;
        if (!end_known) {
            // the line continues past this tile: trailing run only if everything up to '\n' is space
            int64_t p = tile_end;
            trailing = false;
            while (p < b.n && p - tile_end <= b.walk_limit) {
                uint32_t c = b.at(p);
                if (c == 10u) { trailing = true; break; }
                if (!is_posix_space(c)) break;
                ++p;
            }
        }
        if (trailing)
            while (hi > lo && is_posix_space(b.at(hi - 1))) --hi;
    }
    lo_out = lo;
    hi_out = hi;
}

// ---- block-wide scans ---------------------------------------------------------------------------
// Wave64 inclusive scan on DPP (row_shr within rows of 16, then row_bcast:15 / :31 across rows):
// six v_add_*_dpp, no LDS round trips.
__device__ __forceinline__ uint32_t dpp_scan_u32(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);
    return v;
}
#define BZQ_DPP64(ctrl, rm, bc)                                                                    \
    do {                                                                                           \
        const uint32_t lo_ = __builtin_amdgcn_update_dpp(0u, (uint32_t)v, ctrl, rm, 0xf, bc);      \
        const uint32_t hi_ = __builtin_amdgcn_update_dpp(0u, (uint32_t)(v >> 32), ctrl, rm, 0xf, bc); \
        v += ((u64)hi_ << 32) | lo_;                                                               \
    } while (0)
__device__ __forceinline__ u64 dpp_scan_u64(u64 v) {
    BZQ_DPP64(0x111, 0xf, true); BZQ_DPP64(0x112, 0xf, true); BZQ_DPP64(0x114, 0xf, true); BZQ_DPP64(0x118, 0xf, true);
    BZQ_DPP64(0x142, 0xa, false); BZQ_DPP64(0x143, 0xc, false);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v, int lane) {
    (void)lane;
    if constexpr (sizeof(T) == 4) return (T)dpp_scan_u32((uint32_t)v);
    else return (T)dpp_scan_u64((u64)v);
}
// sum over the wave, result in every lane
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
    v = dpp_scan_u64(v);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, 63), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), 63);
    return ((u64)hi << 32) | lo;
}

// Exclusive scan over the block; s_w needs NW entries.  Two barriers.
template <typename T, int NW>
__device__ __forceinline__ T block_exclusive_scan(T v, T* s_w, T& total) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    T incl = wave_inclusive_scan<T>(v, lane);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        T x = s_w[i];
        if (i < w) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// Which tile a workgroup of a one-workgroup-per-tile kernel takes.  Workgroup b runs on XCD b % 8 and every XCD has an L2 of
// its own: handing out the tiles round robin puts two NEIGHBOURING tiles -- whose outputs meet inside a cache line of every
// column -- on two different L2s.  Each XCD gets one contiguous eighth of the tiles instead (any grid size: the first
// gridDim.x % 8 XCDs take one tile more).  Measured on the FASTQ emit: 1.22 -> 1.17 ms, on the FASTA kernels 1.96 -> 1.89 ms
// (same-box A/B, three runs each); groups of 32 tiles per XCD do the same, groups of 256 are slower (1.21).
__device__ __forceinline__ int64_t xcd_tile() {
    const uint32_t q = gridDim.x >> 3, r = gridDim.x & 7u, x = blockIdx.x & 7u;
    return (int64_t)x * q + (int64_t)(x < r ? x : r) + (int64_t)(blockIdx.x >> 3);
}

// ---- tile front end shared by both passes -----------------------------------------------------
// tile_fetch: coalesced 16 B per lane, 4 rounds (piece q = tid + 256*s), into registers.
// tile_stage: newline bitmap s_mask[q] (16-bit mask of piece q; read back as one u64 per thread = the
// 64 contiguous bytes [64*tid, 64*tid+64)) and, optionally, the bytes themselves into LDS.
template <bool NT = false>
__device__ __forceinline__ void tile_fetch(const uint8_t* __restrict__ g, int64_t n, int64_t t0, int valid, uint4 (&r)[4]) {
    const int tid = threadIdx.x;
    if (valid == TILE) {   // every tile but the last: no guards
        const uint8_t* __restrict__ p = g + t0 + tid * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) r[s] = load16_any<NT>(p + BLOCK * 16 * s);
        return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pos = (tid + BLOCK * s) * 16;
        r[s] = make_uint4(0u, 0u, 0u, 0u);
        if (pos < valid) r[s] = load16(g, t0 + pos, n);
    }
}
template <bool STAGE>
__device__ __forceinline__ void tile_stage(const uint4 (&r)[4], int valid, uint16_t* s_mask, uint8_t* s_tile) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int q = tid + BLOCK * s;
        const int pos = q * 16;
        uint32_t m = nl_mask16(r[s]);
        if (valid != TILE) {   // last tile: bytes at or beyond the end are not newlines
            const int rem = valid - pos;
            if (rem < 16) m &= rem > 0 ? ((1u << rem) - 1u) : 0u;
        }
        s_mask[q] = (uint16_t)m;
        if (STAGE) *reinterpret_cast<uint4*>(s_tile + pos) = r[s];
    }
}
template <bool STAGE>
__device__ __forceinline__ void tile_load(const uint8_t* __restrict__ g, int64_t n, int64_t t0, int valid,
                                          uint16_t* s_mask, uint8_t* s_tile /* +16 front pad applied */) {
    uint4 r[4];
    tile_fetch(g, n, t0, valid, r);
    tile_stage<STAGE>(r, valid, s_mask, s_tile);
}

__device__ __forceinline__ int prev_newline_before_word(const u64* s_mask64, int word) {
    for (int w = word - 1; w >= 0; --w) {
        u64 mm = s_mask64[w];
        if (mm) return w * 64 + 63 - __builtin_clzll(mm);
    }
    return -1;
}

// Calls f(j, start, end, end_in_tile) for every line this thread owns: a line is owned by the
// thread whose 64 bytes hold its terminating newline; the tile's unterminated last line (index c)
// by thread BLOCK-1.  start/end are tile-local, end is the newline position (or `valid`).
template <typename F>
__device__ __forceinline__ void for_each_owned_line(const u64* s_mask64, u64 m64, int excl, int c, int valid, F&& f) {
    const int tid = threadIdx.x;
    int prev = -2, idx = 0;
    u64 m = m64;
    while (m) {
        const int bit = __builtin_ctzll(m);
        m &= m - 1;
        const int nl = tid * 64 + bit;
        const int start = (idx == 0 ? prev_newline_before_word(s_mask64, tid) : prev) + 1;
        f(excl + idx, start, nl, true);
        prev = nl;
        ++idx;
    }
    if (tid == BLOCK - 1) {
        int last = m64 ? (tid * 64 + 63 - __builtin_clzll(m64)) : prev_newline_before_word(s_mask64, tid);
        f(c, last + 1, valid, false);
    }
}

// =================================================================================== pass A
struct AggArgs {
    const uint8_t* g;
    int64_t n;
    uint32_t prev_byte;
    int64_t tile_begin;
    int64_t tile_end;   // persistent kernels walk [tile_begin + blockIdx.x, tile_end) with stride gridDim.x
    uint32_t* tile_c;   // newlines per tile
    u64* tile_a;        // 4 x u16: non-newline bytes per line class (local line index & 3)
    u64* tile_idc;      // 4 x u16: id bytes per class if that class were the header role
    int64_t walk_limit; // ByteSrc::walk_limit (0 = none)
    // 4 x u16: tile offset of the LAST newline of each line class (0xFFFF: the tile has none of that class).  With the line
    // prefixes of the scan this locates the end of the record before the one that straddles a tile's start without reading
    // that tile again (the record-length check of the emit kernel, FusedArgs::fold); nullptr = not wanted
    u64* tile_last;
};

// tile_last word from the newline table: the last newline of class k is local newline c-1-((c-1-k) & 3)
__device__ __forceinline__ u64 last_by_class(const uint16_t* s_nl, int c) {
    u64 w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) w |= (u64)(c > k ? s_nl[c - 1 - ((c - 1 - k) & 3)] : (uint16_t)0xFFFFu) << (16 * k);
    return w;
}

#if BZQ_EXPERIMENTS
static __global__ __launch_bounds__(BLOCK) void k_tile_aggregate(AggArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[TILE];
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_a[4], s_idc[4];
    const int tid = threadIdx.x;
    const int64_t t = a.tile_begin + blockIdx.x;
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    if (tid < 4) { s_a[tid] = 0; s_idc[tid] = 0; }
    tile_load<true>(a.g, a.n, t0, valid, s_mask, s_tile);
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(s_mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), s_w, c);
    ByteSrc bs{a.g, a.n, a.prev_byte, s_tile, t0, valid};
    const bool first_starts = (bs.at(t0 - 1) == 10u);
    uint32_t la[4] = {0, 0, 0, 0}, li[4] = {0, 0, 0, 0};
    for_each_owned_line(s_mask64, m64, (int)excl, (int)c, valid, [&](int j, int start, int end, bool end_in) {
        const int cls = j & 3;
        const int len = end - start;
        if (len <= 0) return;
        la[cls] += (uint32_t)len;
        const bool start_in = j > 0 ? true : first_starts;
        int64_t lo, hi;
        header_kept(bs, t0 + start, t0 + end, start_in, end_in, t0 + valid, lo, hi);
        li[cls] += (uint32_t)(hi - lo);
    });
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (la[k]) atomicAdd(&s_a[k], la[k]);
        if (li[k]) atomicAdd(&s_idc[k], li[k]);
    }
    __syncthreads();
    if (tid == 0) {
        a.tile_c[t] = c;
        a.tile_a[t] = (u64)s_a[0] | ((u64)s_a[1] << 16) | ((u64)s_a[2] << 32) | ((u64)s_a[3] << 48);
        a.tile_idc[t] = (u64)s_idc[0] | ((u64)s_idc[1] << 16) | ((u64)s_idc[2] << 32) | ((u64)s_idc[3] << 48);
    }
}

#endif   // BZQ_EXPERIMENTS

// =================================================================================== tile scan
// Two small kernels over the per-tile summaries (20 B per 16 KiB tile):
//   k_scan_reduce  one workgroup per 1024 tiles: the group's summary in the same phase-agnostic
//                  class form (a tile that starts L lines into the group contributes its class k to
//                  group class (k+L)&3)
//   k_scan_down    one workgroup per group.  First it derives the group's own carry (line index, seq, qual, id at
//                  its first tile) from ALL earlier groups' summaries -- a block scan of the line counts resolves
//                  every group's phase, which picks the class to sum for each column -- then the exclusive prefix
//                  per tile.  (A separate one-wave "spine" kernel walking the groups serially took 25 us.)
constexpr int SG_THREADS = 256;
constexpr int SG_ITEMS = 4;
constexpr int SG_TILES = SG_THREADS * SG_ITEMS;

struct ScanArgs {
    int64_t tile_begin, tile_end;
    const uint32_t* tile_c;
    const u64* tile_a;
    const u64* tile_idc;
    int64_t* tileP;
    int64_t* tileS;
    int64_t* tileQ;
    int64_t* tileI;
    // per group (index relative to the group of tile_begin): c, a[4], idc[4], last tile with a newline + 1
    int64_t* grp;        // 10 x int64 per group
    ChunkState* st;
    int32_t pass;        // 0: the carry starts from P0..; k > 0: from pass_carry[k & 1] left by the previous pass
    // batch index (record / batch) of the record the tile's first line belongs to, 0 for the lines of a shard's head (record < 0);
    // nullptr = not wanted.  One division per tile here instead of one per record in the emit kernel (FusedArgs::fold)
    int32_t* tileB;
    int64_t batch;
    // btile[k - 1] = the tile that holds the newline ending record k * batch - 1 (line 4 k batch - 1), for k_batch_bases; bb_cap entries
    int64_t* btile;
    int64_t bb_cap;
    ViewsPool* pool;     // views mode: see ViewsPool; nullptr otherwise
    // diagnostic (option state_init_in_kernel): the chunk state's initial values are written by workgroup 0 of
    // k_scan_reduce instead of being copied in.  0 = off (the product's way: a copy on the side stream)
    int32_t init_mode;
    int64_t init0[4];    // P0, S0, Q0, I0
};

__device__ __forceinline__ int64_t field16(u64 v, int k) { return (int64_t)((v >> (16 * (k & 3))) & 0xFFFFull); }

// block-wide sum of an int64, result in every thread (NW waves; s_r needs NW entries; two barriers)
template <int NW>
__device__ __forceinline__ int64_t block_sum_i64(int64_t v, int64_t* s_r) {
    const u64 w = wave_sum_u64((u64)v);
    if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = (int64_t)w;
    __syncthreads();
    int64_t t = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += s_r[i];
    __syncthreads();
    return t;
}

// diagnostic (state_init_in_kernel = 4): the state's initial values by a kernel of its own, in FRONT of pass A on the ctx stream
static __global__ __launch_bounds__(64) void k_state_init(ChunkState* st, int64_t P0, int64_t S0, int64_t Q0, int64_t I0) {
    if (threadIdx.x == 0) {
        ChunkState z{};
        z.P0 = P0; z.S0 = S0; z.Q0 = Q0; z.I0 = I0; z.P = P0; z.S = S0; z.Q = Q0; z.I = I0;
        z.last_nl_tile = -1; z.err_struct = ~0ull; z.err_valid = ~0ull; z.err_buf = ~0ull;
        for (int i = 0; i < 4; ++i) z.first_nl[i] = -1;
        *st = z;
    }
}

static __global__ __launch_bounds__(SG_THREADS) void k_scan_reduce(ScanArgs a) {
    constexpr int NW = SG_THREADS / 64;
    __shared__ int64_t s_w[NW];
    __shared__ int64_t s_acc[NW][9];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i0 = a.tile_begin + (int64_t)blockIdx.x * SG_TILES + (int64_t)tid * SG_ITEMS;
    if (a.init_mode && blockIdx.x == 0) {   // diagnostic only (see ScanArgs::init_mode)
        u64* w = reinterpret_cast<u64*>(a.st);
        constexpr int NWORDS = (int)(sizeof(ChunkState) / 8);
        static_assert(sizeof(ChunkState) % 8 == 0, "ChunkState is a whole number of 8-byte words");
        if (a.init_mode == 1 || a.init_mode == 3) {   // every word by one thread, in order: no two threads store to the same word
            if (tid == 0) {
                ChunkState z{};
                z.P0 = a.init0[0]; z.S0 = a.init0[1]; z.Q0 = a.init0[2]; z.I0 = a.init0[3];
                z.P = z.P0; z.S = z.S0; z.Q = z.Q0; z.I = z.I0;
                z.last_nl_tile = -1; z.err_struct = ~0ull; z.err_valid = ~0ull; z.err_buf = ~0ull;
                for (int i = 0; i < 4; ++i) z.first_nl[i] = -1;
                *a.st = z;
                if (a.init_mode == 3) __threadfence();   // (3: an agent-scope release by the writer itself, before the kernel's own end-of-kernel release)
            }
        } else {                  // init_mode 2: the words zeroed by all threads, the non-zero fields by thread 0 BEHIND a barrier
            for (int i = tid; i < NWORDS; i += SG_THREADS) w[i] = 0ull;
            __syncthreads();
            if (tid == 0) {
                a.st->P0 = a.init0[0]; a.st->S0 = a.init0[1]; a.st->Q0 = a.init0[2]; a.st->I0 = a.init0[3];
                a.st->P = a.init0[0]; a.st->S = a.init0[1]; a.st->Q = a.init0[2]; a.st->I = a.init0[3];
                a.st->last_nl_tile = -1; a.st->err_struct = ~0ull; a.st->err_valid = ~0ull; a.st->err_buf = ~0ull;
                for (int i = 0; i < 4; ++i) a.st->first_nl[i] = -1;
            }
        }
    }
    int64_t c[SG_ITEMS];
    u64 av[SG_ITEMS], iv[SG_ITEMS];
    int64_t sum = 0, my_last = 0;   // last tile with a newline, + 1 (0 = none)
#pragma unroll
    for (int k = 0; k < SG_ITEMS; ++k) {
        const int64_t i = i0 + k;
        const bool ok = i < a.tile_end;
        c[k] = ok ? (int64_t)a.tile_c[i] : 0;
        av[k] = ok ? a.tile_a[i] : 0ull;
        iv[k] = ok ? a.tile_idc[i] : 0ull;
        if (ok && c[k] > 0) my_last = i + 1;
        sum += c[k];
    }
    int64_t tot;
    int64_t ell = block_exclusive_scan<int64_t, NW>(sum, s_w, tot);
    int64_t A[4] = {0, 0, 0, 0}, D[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < SG_ITEMS; ++k) {
        const int rot = (int)(ell & 3);
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            A[(cls + rot) & 3] += field16(av[k], cls);
            D[(cls + rot) & 3] += field16(iv[k], cls);
        }
        ell += c[k];
    }
    // my_last grows with the thread index, so the wave/block maximum is the last non-zero one: sum of a one-hot is
    // not available, take the max through a u64 add-free path: tiles are < 2^40, pack (my_last) as is and use max
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u64 sa = wave_sum_u64((u64)A[k]), sd = wave_sum_u64((u64)D[k]);
        if (lane == 0) { s_acc[wave][k] = (int64_t)sa; s_acc[wave][4 + k] = (int64_t)sd; }
    }
    {
        int64_t m = my_last;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int64_t o = __shfl_xor(m, off, 64);
            m = o > m ? o : m;
        }
        if (lane == 0) s_acc[wave][8] = m;
    }
    __syncthreads();
    if (tid < 10) {
        int64_t v;
        if (tid == 0) v = tot;
        else if (tid == 9) { v = 0; for (int w = 0; w < NW; ++w) v = s_acc[w][8] > v ? s_acc[w][8] : v; }
        else { v = 0; for (int w = 0; w < NW; ++w) v += s_acc[w][tid - 1]; }
        a.grp[(int64_t)blockIdx.x * 10 + tid] = v;
    }
}

static __global__ __launch_bounds__(SG_THREADS) void k_scan_down(ScanArgs a) {
    constexpr int NW = SG_THREADS / 64;
    __shared__ int64_t s_w[NW];
    const int tid = threadIdx.x;
    const int64_t g = blockIdx.x, ng = gridDim.x;
    // ---- this group's carry from the groups before it ---------------------------------------------------
    int64_t cP, cS, cQ, cI, last;
    if (a.pass == 0) { cP = a.st->P0; cS = a.st->S0; cQ = a.st->Q0; cI = a.st->I0; last = 0; }
    else {
        const int64_t* pc = a.st->pass_carry[a.pass & 1];
        cP = pc[0]; cS = pc[1]; cQ = pc[2]; cI = pc[3]; last = pc[4];
    }
    for (int64_t base = 0; base < g; base += SG_THREADS) {
        const int64_t h = base + tid;
        const bool ok = h < g;
        const int64_t* gs = &a.grp[h * 10];
        const int64_t ch = ok ? gs[0] : 0;
        int64_t tot;
        const int64_t ph64 = cP + block_exclusive_scan<int64_t, NW>(ch, s_w, tot);
        const int ph = (int)(ph64 & 3);                       // role of group h's first line
        const int64_t sv = ok ? gs[1 + ((1 - ph) & 3)] : 0;   // its class whose role is 1 (sequence)
        const int64_t qv = ok ? gs[1 + ((3 - ph) & 3)] : 0;   // role 3 (quality)
        const int64_t dv = ok ? gs[5 + ((0 - ph) & 3)] : 0;   // role 0 (header) after strip
        int64_t lv = ok ? gs[9] : 0;
        cS += block_sum_i64<NW>(sv, s_w);
        cQ += block_sum_i64<NW>(qv, s_w);
        cI += block_sum_i64<NW>(dv, s_w);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const int64_t o = __shfl_xor(lv, off, 64); lv = o > lv ? o : lv; }
        if ((tid & 63) == 0) s_w[tid >> 6] = lv;
        __syncthreads();
        for (int w = 0; w < NW; ++w) last = s_w[w] > last ? s_w[w] : last;
        __syncthreads();
        cP += tot;
    }
    // ---- exclusive prefix per tile -------------------------------------------------------------------------
    const int64_t i0 = a.tile_begin + g * SG_TILES + (int64_t)tid * SG_ITEMS;
    int64_t c[SG_ITEMS];
    u64 av[SG_ITEMS], iv[SG_ITEMS];
    int64_t sum = 0;
#pragma unroll
    for (int k = 0; k < SG_ITEMS; ++k) {
        const int64_t i = i0 + k;
        const bool ok = i < a.tile_end;
        c[k] = ok ? (int64_t)a.tile_c[i] : 0;
        av[k] = ok ? a.tile_a[i] : 0ull;
        iv[k] = ok ? a.tile_idc[i] : 0ull;
        sum += c[k];
    }
    int64_t tot;
    int64_t p = cP + block_exclusive_scan<int64_t, NW>(sum, s_w, tot);
    int64_t ps[SG_ITEMS], sv[SG_ITEMS], qv[SG_ITEMS], dv[SG_ITEMS];
    int64_t ss = 0, sq = 0, si = 0;
#pragma unroll
    for (int k = 0; k < SG_ITEMS; ++k) {
        ps[k] = p;
        const int ph = (int)(p & 3);            // role of the tile's first line
        sv[k] = field16(av[k], 1 - ph);         // class whose role is 1 (sequence)
        qv[k] = field16(av[k], 3 - ph);         // role 3 (quality)
        dv[k] = field16(iv[k], 0 - ph);         // role 0 (header) after strip
        ss += sv[k]; sq += qv[k]; si += dv[k];
        p += c[k];
    }
    int64_t tS, tQ, tI;
    int64_t eS = cS + block_exclusive_scan<int64_t, NW>(ss, s_w, tS);
    int64_t eQ = cQ + block_exclusive_scan<int64_t, NW>(sq, s_w, tQ);
    int64_t eI = cI + block_exclusive_scan<int64_t, NW>(si, s_w, tI);
#pragma unroll
    for (int k = 0; k < SG_ITEMS; ++k) {
        const int64_t i = i0 + k;
        if (i < a.tile_end) {
            a.tileP[i] = ps[k]; a.tileS[i] = eS; a.tileQ[i] = eQ; a.tileI[i] = eI;
            if (a.tileB) {
                a.tileB[i] = ps[k] >= 4 ? (int32_t)((ps[k] >> 2) / a.batch) : 0;
                // batch boundaries among this tile's newlines: lines [ps, ps + c) against the lines u k - 1, u = 4 batch
                const int64_t u = 4 * a.batch, hi_line = ps[k] + c[k];
                if (c[k] > 0 && hi_line > 0) {
                    int64_t kb = ps[k] + 1 <= 0 ? 1 : (ps[k] + u) / u;
                    const int64_t ke = hi_line / u;
                    for (; kb <= ke; ++kb)
                        if (kb - 1 < a.bb_cap) a.btile[kb - 1] = i;
                }
            }
        }
        eS += sv[k]; eQ += qv[k]; eI += dv[k];
    }
    if (g == ng - 1 && tid == 0) {   // totals of this pass: the next pass's carry and the host's counts
        const int64_t own = a.grp[g * 10 + 9];
        if (own > last) last = own;
        int64_t* pc = a.st->pass_carry[(a.pass + 1) & 1];
        pc[0] = cP + tot; pc[1] = cS + tS; pc[2] = cQ + tQ; pc[3] = cI + tI; pc[4] = last;
        a.st->P = cP + tot; a.st->S = cS + tS; a.st->Q = cQ + tQ; a.st->I = cI + tI; a.st->last_nl_tile = last - 1;
        if (a.pool) {
            a.st->listed_tiles += a.pool->listed;
            if (a.pool->fallback) a.st->views_fallback = 1;
            a.pool->listed = 0ull; a.pool->fallback = 0;
        }
    }
}

// Tail after the last newline: where it starts and whether it is more than blanks
// (_check_end_qual, blazeseq/utils.mojo:292-329).  One workgroup; the last tile with a newline is read as 16-byte
// pieces like every other tile.
// With FusedArgs::fold (no k_rebase behind the emit) the kernel also leaves the chunk totals the host reads with the state and the
// batch-table entry of the last, partial batch (fin.on).
struct ChunkFinishArgs {
    int32_t on;
    const int64_t* b_ends;      // per-batch ends as the emit wrote them (the chunk-cumulative arrays do not exist in this mode)
    const int64_t* b_id_ends;
    const int64_t* rec_end;
    int64_t batch, first_header, rec_cap;
    int64_t* bb;
    int64_t bb_cap;
    // k_tail, the submit's last kernel on this path, also PUBLISHES what the host reads: the chunk state and the batch-boundary table
    // go straight into the host's pinned copies (zero-copy stores), so that no copy packet -- two idle gaps of the queue -- follows the
    // kernels.  nullptr = the host copies as before.
    ChunkState* h_state;
    int64_t* h_bb;
    int64_t h_bb_cap;   // batches the pinned table holds
};
__device__ __forceinline__ void chunk_finish(const ChunkFinishArgs& f, ChunkState* st) {
    const int64_t lines = st->P;
    int64_t n_rec = lines > 0 ? (lines >> 2) : 0;
    if (n_rec > f.rec_cap) n_rec = f.rec_cap;   // overflow: the host re-sizes and re-runs
    st->n_complete = n_rec;
    st->last_record_end = n_rec ? f.rec_end[n_rec - 1] : f.first_header - 1;
    int64_t le = 0, li = 0;
    if (n_rec) {   // chunk-cumulative = per-batch value + the batch's base (the table holds every full batch: FusedArgs::fold requires it)
        const int64_t k = (n_rec - 1) / f.batch;
        le = f.b_ends[n_rec - 1] + (k ? f.bb[2 * (k - 1)] : 0);
        li = f.b_id_ends[n_rec - 1] + (k ? f.bb[2 * (k - 1) + 1] : 0);
        if (k < f.bb_cap) { f.bb[2 * k] = le; f.bb[2 * k + 1] = li; }
    }
    st->last_ends = le; st->last_id_ends = li;
}

// The chunk state into the host's pinned copy by zero-copy stores (views mode, whose last kernel -- the join -- has many workgroups and
// cannot publish itself): one small kernel instead of a copy packet behind the kernels (a 664-byte device-to-host copy is a blit
// kernel of its own, ~10-14 us in the queue).
static __global__ __launch_bounds__(128) void k_publish_state(const ChunkState* st, ChunkState* h_state) {
    static_assert(sizeof(ChunkState) % 8 == 0, "the state is copied in 8-byte words");
    const u64* src = reinterpret_cast<const u64*>(st);
    u64* dst = reinterpret_cast<u64*>(h_state);
    for (int i = threadIdx.x; i < (int)(sizeof(ChunkState) / 8); i += 128) dst[i] = src[i];
}

// Chunk-cumulative ends from the per-batch ones (bzq_chunk_cumulative_ends; cold path): ends[r] = b_ends[r] + bb[2 (r / batch - 1)]
static __global__ __launch_bounds__(BLOCK) void k_cumulate(const int64_t* b_ends, const int64_t* b_id_ends, const int64_t* bb, int64_t batch, int64_t n,
                                                           int64_t* ends, int64_t* id_ends) {
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * BLOCK) {
        const int64_t k = r / batch;
        ends[r] = b_ends[r] + (k ? bb[2 * (k - 1)] : 0);
        id_ends[r] = b_id_ends[r] + (k ? bb[2 * (k - 1) + 1] : 0);
    }
}

// k_rebase_range / k_fix_last for a chunk whose per-batch ends were written by the emit (no chunk-cumulative arrays)
static __global__ __launch_bounds__(BLOCK) void k_rebase_range_b(const int64_t* b_ends, const int64_t* b_id_ends, const int64_t* bb, int64_t batch,
                                                                 int64_t first, int64_t count, int64_t* out_e, int64_t* out_i) {
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= count) return;
    auto cum = [&](int64_t x, int64_t& e, int64_t& i) {
        const int64_t k = x / batch;
        e = b_ends[x] + (k ? bb[2 * (k - 1)] : 0);
        i = b_id_ends[x] + (k ? bb[2 * (k - 1) + 1] : 0);
    };
    int64_t e0 = 0, i0 = 0, e1, i1;
    if (first) cum(first - 1, e0, i0);
    cum(first + r, e1, i1);
    out_e[r] = e1 - e0;
    out_i[r] = i1 - i0;
}
static __global__ void k_fix_last_b(int64_t rec, int64_t n, int64_t batch, int64_t* rec_end, int64_t* b_ends, int64_t* b_id_ends, int64_t* bb,
                                    int64_t bb_cap, const ChunkState* st) {
    if (threadIdx.x || blockIdx.x) return;
    const int64_t k = rec / batch;
    rec_end[rec] = n;
    b_ends[rec] = st->Q - (k ? bb[2 * (k - 1)] : 0);
    b_id_ends[rec] = st->I - (k ? bb[2 * (k - 1) + 1] : 0);
    if (k < bb_cap) { bb[2 * k] = st->Q; bb[2 * k + 1] = st->I; }   // the table's last entry now ends at this record
}

static __global__ __launch_bounds__(BLOCK) void k_tail(const uint8_t* __restrict__ g, int64_t n, ChunkState* st, ChunkFinishArgs fin) {
    __shared__ int s_pos;
    __shared__ int s_nb;
    const int tid = threadIdx.x;
    if (tid == 0) { s_pos = -1; s_nb = 0; }
    __syncthreads();
    const int64_t lt = st->last_nl_tile;
    int64_t tail = 0;
    if (lt >= 0) {
        const int64_t t0 = lt * TILE;
        const int valid = (int)((n - t0) < TILE ? (n - t0) : TILE);
        uint4 r[4];
        tile_fetch(g, n, t0, valid, r);
        int best = -1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = (tid + BLOCK * s) * 16;
            uint32_t m = nl_mask16(r[s]);
            const int rem = valid - pos;
            if (rem < 16) m &= rem > 0 ? ((1u << rem) - 1u) : 0u;
            if (m) best = pos + 31 - __builtin_clz(m);
        }
        if (best >= 0) atomicMax(&s_pos, best);
        __syncthreads();
        tail = t0 + s_pos + 1;
    }
    int nb = 0;
    for (int64_t p = tail + tid; p < n; p += BLOCK) {
        const uint32_t c = g[p];
        if (c != 10u && c != 13u && c != 32u && c != 9u) { nb = 1; break; }
    }
    if (nb) atomicOr(&s_nb, 1);
    __syncthreads();
    if (tid == 0) {
        st->tail_start = tail; st->tail_nonblank = s_nb;
        if (fin.on) chunk_finish(fin, st);
    }
    if (fin.on && fin.h_state) {   // (fin is a kernel argument: the same for every thread)
        __threadfence_block();
        __syncthreads();
        const int64_t batch = fin.batch > 0 ? fin.batch : 1;
        int64_t nb = (st->n_complete + batch - 1) / batch;
        if (nb > fin.bb_cap) nb = fin.bb_cap;
        if (nb > fin.h_bb_cap) nb = fin.h_bb_cap;
        // (16 bytes per lane, a batch's pair: consecutive lanes write consecutive addresses -- the stores cross PCIe, and 8-byte stores
        // of a strided loop were 8 us of this kernel)
        if (fin.h_bb && fin.bb) {
            const uint4* s4 = reinterpret_cast<const uint4*>(fin.bb);
            uint4* d4 = reinterpret_cast<uint4*>(fin.h_bb);
            for (int64_t i = tid; i < nb; i += BLOCK) d4[i] = s4[i];
        }
        static_assert(sizeof(ChunkState) % 8 == 0, "the state is copied in 8-byte words");
        const u64* src = reinterpret_cast<const u64*>(st);
        u64* dst = reinterpret_cast<u64*>(fin.h_state);
        for (int i = tid; i < (int)(sizeof(ChunkState) / 8); i += BLOCK) dst[i] = src[i];
    }
}

#if BZQ_EXPERIMENTS
// =================================================================================== pass B
struct EmitArgs {
    const uint8_t* g;
    int64_t n;
    uint32_t prev_byte;
    int64_t tile_begin;
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
};

#endif   // BZQ_EXPERIMENTS

struct ErrAcc {
    u64 e_struct, e_valid;
    __device__ __forceinline__ void structure(int64_t rec, int code) {
        const u64 k = ((u64)rec << 3) | (u64)code;
        if (rec >= 0 && k < e_struct) e_struct = k;
    }
    __device__ __forceinline__ void valid(int64_t rec, int code) {
        const u64 k = ((u64)rec << 3) | (u64)code;
        if (rec >= 0 && k < e_valid) e_valid = k;
    }
};

__device__ __forceinline__ uint32_t byte_range_mask(int i, int a, int b) {
    // mask of the bytes of dword i (bytes 4i..4i+3 of a 16-byte piece) that fall in [a, b)
    int lo = a - 4 * i, hi = b - 4 * i;
    lo = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
    hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
    if (hi <= lo) return 0u;
    const uint32_t mh = (uint32_t)((1ull << (8 * hi)) - 1ull);
    const uint32_t ml = (uint32_t)((1ull << (8 * lo)) - 1ull);
    return mh & ~ml;
}

// ascii: any byte with the high bit (record.mojo:106-116, utils.mojo:245-263)
__device__ __forceinline__ bool any_non_ascii(uint32_t x) { return (x & 0x80808080u) != 0u; }
// quality: any byte outside [lower, upper], i.e. (q - lower) > (upper - lower) unsigned (record.mojo:99-102)
__device__ __forceinline__ bool any_out_of_range(uint32_t x, uint32_t lower, uint32_t upper) {
    const uint32_t less = (x - 0x01010101u * lower) & ~x & 0x80808080u;            // some byte < lower (lower <= 128)
    const uint32_t more = ((x + 0x01010101u * (127u - upper)) | x) & 0x80808080u;  // some byte > upper (upper <= 127)
    return (less | more) != 0u;
}

#if BZQ_EXPERIMENTS
// Gather one role's byte stream of this tile into its packed column.
//   stream coordinate o in [0, n_role): the o-th byte of this role inside the tile;
//   segment k: bytes [seg_dst[k], seg_dst[k]+seg_len[k]) of the stream come from tile offset seg_src[k].
// Each lane produces one aligned 16-byte piece of the destination column from <= a few unaligned
// LDS windows; interior pieces are single coalesced dwordx4 stores, the (at most two) partial edge
// pieces of the tile's range are stored bytewise because the neighbouring tiles own the rest.
template <int ROLE, bool CA, bool CQ>
__device__ __forceinline__ void gather_role(uint8_t* __restrict__ col, int64_t D, int n_role,
                                            const uint16_t* seg_src, const uint16_t* seg_len,
                                            const uint16_t* seg_dst, int nk, const uint8_t* s_tile,
                                            int64_t line0 /* P + j0(role) */, uint32_t qlo, uint32_t qhi,
                                            ErrAcc& err) {
    if (n_role <= 0) return;
    const int tid = threadIdx.x;
    const int64_t hi_abs = D + n_role;
    const int64_t pa = D >> 4, pb = (hi_abs - 1) >> 4;
    const uint32_t* tw = reinterpret_cast<const uint32_t*>(s_tile - 16); // dword view incl. the front pad
    for (int64_t pi = pa + tid; pi <= pb; pi += BLOCK) {
        const int64_t p0 = pi << 4;
        if (p0 + 16 <= 0) continue; // bytes of the straddling head record (owned by the previous shard)
        int xl = (int)(D > p0 ? D - p0 : 0);
        int xh = (int)(hi_abs - p0 < 16 ? hi_abs - p0 : 16);
        if (p0 < 0 && xl < (int)(-p0)) xl = (int)(-p0);
        int o = (int)(p0 - D) + xl;
        // segment holding stream byte o: last k with seg_dst[k] <= o
        int lo = 0, hi = nk;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int)seg_dst[mid] <= o) lo = mid + 1; else hi = mid;
        }
        int k = lo - 1;
        uint32_t acc[4] = {0u, 0u, 0u, 0u};
        int x = xl;
        while (x < xh) {
            int send = (int)seg_dst[k] + (int)seg_len[k];
            while (o >= send) { ++k; send = (int)seg_dst[k] + (int)seg_len[k]; }
            int take = xh - x;
            if (send - o < take) take = send - o;
            const int A = (int)seg_src[k] + (o - (int)seg_dst[k]); // tile offset of piece byte x
            const int ws = A - x + 16;                               // window start incl. front pad (>= 1)
            const int wd = ws >> 2, sh = ws & 3;
            const uint32_t d0 = tw[wd], d1 = tw[wd + 1], d2 = tw[wd + 2], d3 = tw[wd + 3], d4 = tw[wd + 4];
            uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
            uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            uint32_t w2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
            uint32_t w3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
            if (take < 16) {
                const uint32_t m0 = byte_range_mask(0, x, x + take), m1 = byte_range_mask(1, x, x + take),
                               m2 = byte_range_mask(2, x, x + take), m3 = byte_range_mask(3, x, x + take);
                w0 &= m0; w1 &= m1; w2 &= m2; w3 &= m3;
                if (CA || CQ) {
                    if (CA && any_non_ascii(w0 | w1 | w2 | w3)) err.valid((line0 + 4 * (int64_t)k) >> 2, 4);
                    if (CQ && ROLE == 3) {
                        const uint32_t fill = 0x01010101u * qlo;
                        if (any_out_of_range(w0 | (fill & ~m0), qlo, qhi) | any_out_of_range(w1 | (fill & ~m1), qlo, qhi) |
                            any_out_of_range(w2 | (fill & ~m2), qlo, qhi) | any_out_of_range(w3 | (fill & ~m3), qlo, qhi))
                            err.valid((line0 + 4 * (int64_t)k) >> 2, 5);
                    }
                }
                acc[0] |= w0; acc[1] |= w1; acc[2] |= w2; acc[3] |= w3;
            } else {
                if (CA && any_non_ascii(w0 | w1 | w2 | w3)) err.valid((line0 + 4 * (int64_t)k) >> 2, 4);
                if (CQ && ROLE == 3) {
                    if (any_out_of_range(w0, qlo, qhi) | any_out_of_range(w1, qlo, qhi) |
                        any_out_of_range(w2, qlo, qhi) | any_out_of_range(w3, qlo, qhi))
                        err.valid((line0 + 4 * (int64_t)k) >> 2, 5);
                }
                acc[0] = w0; acc[1] = w1; acc[2] = w2; acc[3] = w3;
            }
            x += take;
            o += take;
        }
        if (xl == 0 && xh == 16) {
            *reinterpret_cast<uint4*>(col + p0) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        } else {
            for (int i = xl; i < xh; ++i) col[p0 + i] = (uint8_t)(acc[i >> 2] >> (8 * (i & 3)));
        }
    }
}

template <bool CA, bool CQ, bool OFFS>
static __global__ __launch_bounds__(BLOCK) void k_tile_emit(EmitArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile_raw[16 + TILE + 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];
    __shared__ uint16_t s_src[3][SEGS], s_len[3][SEGS], s_dst[3][SEGS];
    __shared__ u64 s_w64[4];
    __shared__ uint32_t s_w[4];
    uint8_t* s_tile = s_tile_raw + 16;
    const int tid = threadIdx.x;
    const int64_t t = a.tile_begin + blockIdx.x;
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    tile_load<true>(a.g, a.n, t0, valid, s_mask, s_tile);
    // role slot in the segment tables: 0 header, 1 sequence, 2 quality
    s_len[0][tid] = 0; s_len[1][tid] = 0; s_len[2][tid] = 0;
    s_src[0][tid] = 0; s_src[1][tid] = 0; s_src[2][tid] = 0;
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(s_mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), s_w, c);
    const int64_t P = a.tileP[t];
    const int64_t S = a.tileS[t], Q = a.tileQ[t], I = a.tileI[t];
    ByteSrc bs{a.g, a.n, a.prev_byte, s_tile, t0, valid};
    const bool first_starts = (bs.at(t0 - 1) == 10u);
    ErrAcc err{~0ull, ~0ull};
    bool overflow = false;

    if ((int)c > MAXL || a.force_dense) {
        // ---------------------------------------------------------------- serial path (any input)
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
                    for (int64_t p = lo; p < hi; ++p) {
                        const uint8_t ch = s_tile[p - t0];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (ri >= 0) a.col_id[ri] = ch;
                        ++ri;
                    }
                    if (end_in && rec >= 0) { if (rec < a.rec_cap) a.id_ends[rec] = ri; else overflow = true; }
                } else if (role == 1) {
                    if (sin && OFFS && rec >= 0 && rec < a.rec_cap) a.o_seq[rec] = ls;
                    for (int p = start; p < end; ++p) {
                        const uint8_t ch = s_tile[p];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (rs >= 0) a.col_seq[rs] = ch;
                        ++rs;
                    }
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
                        if (rq >= 0) a.col_qual[rq] = ch;
                        ++rq;
                    }
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
    } else {
        // ---------------------------------------------------------------- fast path
        // pass 1 over owned lines: segment tables + line-start checks
        for_each_owned_line(s_mask64, m64, (int)excl, (int)c, valid, [&](int j, int start, int end, bool end_in) {
            const int64_t L = P + j;
            const int role = (int)(L & 3);
            const int64_t rec = L >> 2;
            const int k = j >> 2;
            const bool sin = (j > 0 ? true : first_starts) && start < valid;
            const int64_t ls = t0 + start;
            if (role == 0) {
                if (sin) {
                    if (s_tile[start] != 64) err.structure(rec, 1);   // '@', utils.mojo:454
                    if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_hdr[rec] = ls;
                }
                int64_t lo = ls, hi = ls;
                if (end > start) header_kept(bs, ls, t0 + end, j > 0 ? true : first_starts, end_in, t0 + valid, lo, hi);
                s_src[0][k] = (uint16_t)(lo - t0);
                s_len[0][k] = (uint16_t)(hi - lo);
            } else if (role == 1) {
                if (sin && OFFS && rec >= 0 && rec < a.rec_cap) a.o_seq[rec] = ls;
                s_src[1][k] = (uint16_t)start;
                s_len[1][k] = (uint16_t)(end - start);
            } else if (role == 2) {
                if (sin) {
                    if (s_tile[start] != 43) err.structure(rec, 2);   // '+', utils.mojo:456
                    if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_sep[rec] = ls;
                }
            } else {
                if (sin && OFFS && rec >= 0 && rec < a.rec_cap) a.o_qual[rec] = ls;
                s_src[2][k] = (uint16_t)start;
                s_len[2][k] = (uint16_t)(end - start);
            }
        });
        __syncthreads();
        // exclusive scan of the three segment-length tables at once (21-bit fields)
        const u64 packed = (u64)s_len[0][tid] | ((u64)s_len[1][tid] << 21) | ((u64)s_len[2][tid] << 42);
        u64 tot = 0;
        const u64 ex = block_exclusive_scan<u64, 4>(packed, s_w64, tot);
        s_dst[0][tid] = (uint16_t)(ex & 0x1FFFFFull);
        s_dst[1][tid] = (uint16_t)((ex >> 21) & 0x1FFFFFull);
        s_dst[2][tid] = (uint16_t)((ex >> 42) & 0x1FFFFFull);
        const int n_id = (int)(tot & 0x1FFFFFull), n_seq = (int)((tot >> 21) & 0x1FFFFFull),
                  n_qual = (int)((tot >> 42) & 0x1FFFFFull);
        __syncthreads();
        // pass 2 over owned lines that END here: per-record outputs
        {
            u64 m = m64;
            int idx = 0;
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                const int j = (int)excl + idx;
                ++idx;
                const int64_t L = P + j;
                const int role = (int)(L & 3);
                const int64_t rec = L >> 2;
                if (rec < 0) continue;
                const int k = j >> 2;
                if (role == 0) {
                    if (rec < a.rec_cap) a.id_ends[rec] = I + (int64_t)s_dst[0][k] + (int64_t)s_len[0][k];
                    else overflow = true;
                } else if (role == 3) {
                    const int64_t qe = Q + (int64_t)s_dst[2][k] + (int64_t)s_len[2][k];
                    // sequence bytes seen so far = column offset after this record's sequence line (line j-2)
                    int64_t se = S;
                    if (j >= 2) { const int k1 = (j - 2) >> 2; se = S + (int64_t)s_dst[1][k1] + (int64_t)s_len[1][k1]; }
                    if (rec < a.rec_cap) { a.ends[rec] = qe; a.rec_end[rec] = t0 + tid * 64 + bit; }
                    else overflow = true;
                    // seq_len != qual_len for some record <= rec  <=>  cumulative sums differ (utils.mojo:458-461)
                    if (se != qe) err.structure(rec, 3);
                }
            }
        }
        // gather the three streams
        const int jh = (int)((0 - P) & 3), js = (int)((1 - P) & 3), jq = (int)((3 - P) & 3);
        const int nl_lines = (int)c + 1;
        const int nk_h = jh < nl_lines ? ((nl_lines - 1 - jh) >> 2) + 1 : 0;
        const int nk_s = js < nl_lines ? ((nl_lines - 1 - js) >> 2) + 1 : 0;
        const int nk_q = jq < nl_lines ? ((nl_lines - 1 - jq) >> 2) + 1 : 0;
        gather_role<1, CA, CQ>(a.col_seq, S, n_seq, s_src[1], s_len[1], s_dst[1], nk_s, s_tile, P + js, a.q_lower, a.q_upper, err);
        gather_role<3, CA, CQ>(a.col_qual, Q, n_qual, s_src[2], s_len[2], s_dst[2], nk_q, s_tile, P + jq, a.q_lower, a.q_upper, err);
        gather_role<0, CA, CQ>(a.col_id, I, n_id, s_src[0], s_len[0], s_dst[0], nk_h, s_tile, P + jh, a.q_lower, a.q_upper, err);
    }
    if (err.e_struct != ~0ull) atomicMin(&a.st->err_struct, err.e_struct);
    if (err.e_valid != ~0ull) atomicMin(&a.st->err_valid, err.e_valid);
    if (overflow) atomicOr(&a.st->rec_overflow, 1);
}

#endif   // BZQ_EXPERIMENTS

// =================================================================================== per record
struct RebaseArgs {
    const int64_t* ends;
    const int64_t* id_ends;
    const int64_t* rec_end;
    int64_t* b_ends;
    int64_t* b_id_ends;
    int64_t batch;
    int64_t first_header;  // offset of record 0's header
    int64_t len_limit;     // records longer than this are refused by the reference's buffer
    int64_t rec_cap;
    ChunkState* st;
    // optional host-SIMD-width emulation of the quality check (record.mojo:90-97), 0 = off
    const uint8_t* g;
    int32_t compat_w;
    uint32_t q_upper;
    // ends / id_ends at the last record of every batch (and of the chunk): bb[2k], bb[2k+1] -- one small copy to the host per
    // chunk instead of four 8-byte copies per bzq_batch_view; nullptr or bb_cap batches exceeded: not written
    int64_t* bb;
    int64_t bb_cap;
};

// FastqBatch._ends / _id_ends restart at every batch (record_batch.mojo:77-87 via parser.mojo:243):
// b_ends[r] = ends[r] - ends[first record of r's batch - 1].  Also finds the first record the
// reference could not hold in its buffer (parser.mojo:484-492).  Grid-stride over the records; the
// record count is read from the device state, so the host can enqueue this without a sync.
static __global__ __launch_bounds__(BLOCK) void k_rebase(RebaseArgs a) {
    const int64_t lines = a.st->P;
    int64_t n_rec = lines > 0 ? (lines >> 2) : 0;
    if (n_rec > a.rec_cap) n_rec = a.rec_cap; // overflow: the host re-sizes and re-runs
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n_rec; r += (int64_t)gridDim.x * BLOCK) {
        const int64_t b0 = (r / a.batch) * a.batch;
        const int64_t e0 = b0 ? a.ends[b0 - 1] : 0, i0 = b0 ? a.id_ends[b0 - 1] : 0;
        const int64_t e = a.ends[r], ie = a.id_ends[r];
        a.b_ends[r] = e - e0;
        a.b_id_ends[r] = ie - i0;
        if (a.bb && ((r + 1) % a.batch == 0 || r == n_rec - 1)) {
            const int64_t k = r / a.batch;
            if (k < a.bb_cap) { a.bb[2 * k] = e; a.bb[2 * k + 1] = ie; }
        }
        const int64_t re = a.rec_end[r];
        const int64_t prev = r ? a.rec_end[r - 1] : a.first_header - 1;
        if (re - prev > a.len_limit) atomicMin(&a.st->err_buf, (u64)r << 3); // header_start .. '\n' inclusive
        if (a.compat_w > 0) {
            // inside the first floor(n/W)*W quality bytes a byte equal to UPPER is rejected too
            const int64_t qlen = e - (r ? a.ends[r - 1] : 0);
            const int64_t qs = re - qlen, body = (qlen / a.compat_w) * a.compat_w;
            for (int64_t i = 0; i < body; ++i)
                if (a.g[qs + i] == a.q_upper) { atomicMin(&a.st->err_valid, ((u64)r << 3) | 5ull); break; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.st->n_complete = n_rec;
        a.st->last_record_end = n_rec ? a.rec_end[n_rec - 1] : a.first_header - 1;
        a.st->last_ends = n_rec ? a.ends[n_rec - 1] : 0;
        a.st->last_id_ends = n_rec ? a.id_ends[n_rec - 1] : 0;
    }
}

// Rebased ends for an arbitrary record range (a next_batch call that is not batch aligned).
static __global__ __launch_bounds__(BLOCK) void k_rebase_range(const int64_t* ends, const int64_t* id_ends, int64_t first,
                                                        int64_t count, int64_t* out_e, int64_t* out_i) {
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= count) return;
    const int64_t e0 = first ? ends[first - 1] : 0, i0 = first ? id_ends[first - 1] : 0;
    out_e[r] = ends[first + r] - e0;
    out_i[r] = id_ends[first + r] - i0;
}

// Writes the per-record outputs of a last record that has no trailing newline (parser.mojo:464-475,
// utils.mojo:327-329): the columns already hold its bytes.
static __global__ void k_fix_last(int64_t rec, int64_t n, int64_t batch, int64_t* ends, int64_t* id_ends,
                           int64_t* rec_end, int64_t* b_ends, int64_t* b_id_ends, const ChunkState* st) {
    if (threadIdx.x || blockIdx.x) return;
    ends[rec] = st->Q; id_ends[rec] = st->I; rec_end[rec] = n;
    const int64_t b0 = (rec / batch) * batch;
    b_ends[rec] = st->Q - (b0 ? ends[b0 - 1] : 0);
    b_id_ends[rec] = st->I - (b0 ? id_ends[b0 - 1] : 0);
}

// =================================================================================== shard stitch
// Offsets of the first four newlines of a shard (one workgroup; a record is a few hundred bytes to
// a few tens of kB, so this touches a handful of 4 KiB steps).
static __global__ __launch_bounds__(BLOCK) void k_first_newlines(const uint8_t* __restrict__ g, int64_t n, ChunkState* st) {
    __shared__ uint32_t s_w[4];
    __shared__ int s_found;
    const int tid = threadIdx.x;
    if (tid < 4) st->first_nl[tid] = -1;
    if (tid == 0) { s_found = 0; st->edge_first = n > 0 ? g[0] : 10; st->edge_last = n > 0 ? g[n - 1] : 10; }
    __syncthreads();
    for (int64_t base = 0; base < n; base += BLOCK * 16) {
        const int64_t pos = base + (int64_t)tid * 16;
        uint32_t m = 0;
        if (pos < n) {
            m = nl_mask16(load16(g, pos, n));
            const int64_t rem = n - pos;
            if (rem < 16) m &= (1u << rem) - 1u;
        }
        uint32_t tot = 0;
        const uint32_t ex = block_exclusive_scan<uint32_t, 4>((uint32_t)__popc(m), s_w, tot);
        const int have = s_found;
        uint32_t mm = m;
        int idx = 0;
        while (mm) {
            const int bit = __builtin_ctz(mm);
            mm &= mm - 1;
            const int rank = have + (int)ex + idx;
            if (rank < 4) st->first_nl[rank] = pos + bit;
            ++idx;
        }
        __syncthreads();
        if (tid == 0) s_found = have + (int)tot;
        __syncthreads();
        if (s_found >= 4) break;
    }
}

// Column offsets that make the first OWNED record of a shard start at 0: the head lines (the
// straddling record's remainder, owned by the previous shard) get negative offsets.
static __global__ void k_head(const uint8_t* __restrict__ g, int64_t n, uint32_t prev_byte, int head_lines, ChunkState* st) {
    if (threadIdx.x || blockIdx.x) return;
    ByteSrc bs{g, n, prev_byte, nullptr, 0, 0};
    int64_t s0 = 0, q0 = 0, i0 = 0;
    int64_t start = 0;
    const int64_t P0 = st->P0;
    for (int j = 0; j < head_lines; ++j) {
        const int64_t end = st->first_nl[j];
        if (end < 0) break;
        const int role = (int)((P0 + j) & 3);
        if (role == 1) s0 -= end - start;
        else if (role == 3) q0 -= end - start;
        else if (role == 0 && end > start) {
            int64_t lo, hi;
            header_kept(bs, start, end, j > 0 ? true : (prev_byte == 10u), true, end, lo, hi);
            i0 -= hi - lo;
        }
        start = end + 1;
    }
    st->S0 = s0; st->Q0 = q0; st->I0 = i0;
}

// =================================================================================== generator
// generate_synthetic_fastq_buffer for fixed-length reads (blazeseq/utils.mojo:736-917): record i
// depends only on i, one thread per record.
struct GenArgs {
    uint8_t* out;
    int64_t first, count, num_reads;
    int32_t read_len, num_digits, min_phred, max_phred;   // read_len = min_len
    uint32_t q_offset, q_lower, q_upper;
    int64_t len_range;           // max_len - min_len + 1 (1 = fixed length)
    int64_t period;              // lengths repeat every `period` records
    const int64_t* len_prefix;   // [period + 1] prefix sums of one period's lengths (device)
};

static __global__ __launch_bounds__(BLOCK) void k_generate(GenArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= a.count) return;
    const int64_t i = a.first + idx;
    // read length and record offset (utils.mojo:753-757): lengths repeat with period `period`, len_prefix[r] is the
    // sum of the first r lengths of one period
    int64_t L = a.read_len, off;
    const int64_t fixed = 6 + a.num_digits + 1 + 2 + 2;    // "@read_" digits '\n' ... '\n' "+\n" ... '\n'
    if (a.len_range > 1) {
        L = a.read_len + (int64_t)(((u64)i * 31ull + 7ull) % (u64)a.len_range);
        const int64_t q = i / a.period, r = i - q * a.period;
        const int64_t q0 = a.first / a.period, r0 = a.first - q0 * a.period;
        const int64_t sum_i = q * a.len_prefix[a.period] + a.len_prefix[r];
        const int64_t sum_0 = q0 * a.len_prefix[a.period] + a.len_prefix[r0];
        off = idx * fixed + 2 * (sum_i - sum_0);
    } else {
        off = idx * (fixed + 2 * L);
    }
    uint8_t* o = a.out + off;
    const uint8_t lut[8] = {'G', 'C', 'G', 'C', 'A', 'T', 'A', 'T'}; // gc_bias = 0.5, utils.mojo:707-733
    *o++ = '@'; *o++ = 'r'; *o++ = 'e'; *o++ = 'a'; *o++ = 'd'; *o++ = '_';
    {
        int64_t v = i;
        for (int d = a.num_digits - 1; d >= 0; --d) { o[d] = (uint8_t)('0' + (int)(v % 10)); v /= 10; }
        o += a.num_digits;
    }
    *o++ = '\n';
    const u64 MASK = 0x7FFFFFFFFFFFFFFFull;
    u64 st = ((u64)i * 6364136223846793005ull + 1442695040888963407ull) & MASK;
    for (int64_t b = 0; b < L; ++b) {
        st = (st * 6364136223846793005ull + 1442695040888963407ull) & MASK;
        *o++ = lut[(st >> 33) & 7];
    }
    *o++ = '\n'; *o++ = '+'; *o++ = '\n';
    const int64_t q_start = a.max_phred, q_range = a.max_phred - a.min_phred, noise_amp = q_range / 6 + 1;
    u64 qr = ((u64)i * 2654435761ull + 1013904223ull) & MASK;
    const int64_t lm1 = L - 1;
    for (int64_t p = 0; p < L; ++p) {
        const int64_t mean = lm1 == 0 ? q_start : q_start - (q_range * p + lm1 / 2) / lm1;
        qr = (qr * 1664525ull + 1013904223ull) & MASK;
        const int64_t noise = (int64_t)((qr >> 17) % (u64)(2 * noise_amp + 1));
        int64_t ph = mean + noise - noise_amp;
        ph = ph < a.min_phred ? a.min_phred : (ph > a.max_phred ? a.max_phred : ph);
        int64_t c = (int64_t)a.q_offset + ph;
        c = c < (int64_t)a.q_lower ? (int64_t)a.q_lower : (c > (int64_t)a.q_upper ? (int64_t)a.q_upper : c);
        *o++ = (uint8_t)c;
    }
    *o++ = '\n';
}

} // namespace bzq
