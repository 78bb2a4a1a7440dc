// bzq_gzip.hpp -- ANY gzip file inflated on the GPU in parallel (C ABI bzq_gzip_*; the ingest's path for plain .gz files).
//
// Replaces the reference's RapidgzipReader in front of the parser (blazeseq/io/readers.mojo:380-443: `RapidgzipFile.open(path,
// parallelism)` + read_to_buffer; the un-vendored mosafi2/rapidgzip_mojo binding of rapidgzip) -- and GZFile (readers.mojo:283-377,
// libz gzread) for files that are not BGZF.  A gzip member is ONE DEFLATE stream (RFC 1951/1952): a block can only be decoded
// once the Huffman tables in its header are known, and its matches reach up to 32 KiB back into whatever the stream produced
// before it.  rapidgzip's published answer (Knespel & Brunst, HPDC '23) is speculation in two stages, and that is what runs
// here, stage one and two on the device:
//
//   0. the compressed piece is cut into chunks of CH bytes.  FIND (k_gz_find, one wave per chunk): the first position in the
//      chunk at which a DEFLATE block can start -- every lane tests 16 consecutive BIT offsets at once for a non-final dynamic
//      block header (13 bits that need no arithmetic, by word operations); the survivors, compacted onto the lanes, for a
//      complete code length code (up to 57 more bits); what passes is queued and looked at 64 at a time, one candidate per lane
//      (the code lengths and the two codes they describe, by zlib's rules); what is left -- on FASTQ exactly the true starts --
//      is judged by the whole wave.  Byte positions are also tested for a gzip member header followed by a valid block header.
//      A false positive costs time, never correctness (below).  ORDER (k_gz_span, k_gz_order): the longest jobs first.
//   1. DECODE (k_gz_decode, one wave per chunk that has a start): blocks are decoded from the chunk's start until the decoder
//      arrives EXACTLY at a later chunk's start (a candidate it passes over without meeting it was a false positive and is
//      dropped), through member trailers and member headers where they come.  What lies more than `pos` bytes back is not
//      known yet, so the output is 16-bit symbols: 0..255 a byte, 0x8000 | w "the byte at index w of the 32 KiB window in
//      front of this chunk's output" (rapidgzip's markers); markers are copied by later matches like any symbol.  Output goes
//      to 128 KiB pages taken from a pool with one atomic (its size is unknown beforehand).
//   2. the host walks the chain (chunk 0 starts at a known position; every chunk names the chunk it ended on), and the device
//      finishes: CHAIN (k_gz_chain / k_gz_chain_groups): the 32 KiB window behind every chunk from the window in front of it
//      -- window maps compose, so groups of ~sqrt(n) chunks build their composed map in parallel and only the groups are
//      walked serially; RESOLVE (k_gz_resolve, one workgroup per page): every symbol to its byte, contiguous in the caller's
//      buffer; CRC (k_gz_crc): CRC-32 of every MiB of every member, combined on the host and checked against the member
//      trailers together with ISIZE.
//   The NEXT piece's copy to the device runs under all of this (gz_stage), and its FIND under the chain / resolve / CRC of this
//   one: the finder needs the piece's bytes and its carry, and the carry is known as soon as this piece's chain is.
//
// Correctness does not rest on the speculation: the chain starts at an exact position and only ever follows exact ends, so
// what is delivered is the sequential decode; a chunk the chain never lands on is ignored.  Streams that defeat the
// speculation (only fixed-Huffman or stored blocks: nothing to find) decode on fewer waves, in the limit on one.
// Input that is not a valid gzip stream fails the call (BZQ_ERR_IO) -- never a parse result.
#pragma once
#include "bzq_inflate.hpp"

namespace bzq {
namespace gz {

using inf::Code;
using inf::U32U;
using inf::rdlane;
using inf::uni;

constexpr int WAVES = BLOCK / 64;
constexpr int PAGE_SHIFT = 16;
constexpr int PAGE = 1 << PAGE_SHIFT;      // symbols per page (128 KiB of pool).  (Every decoder wave leaves its last page partly used: ~2.3 pages of output round up to 3, so the pool is ~24 x the compressed bytes at 16 KiB per wave.  Pages of 32768 symbols -- the window's size, the smallest that keeps a back reference in the current page or the one before it -- were tried in round 5 and are NOT a drop-in: members then fail their CRC)
constexpr u64 POS_NONE = ~0ull;
constexpr uint32_t NO_PAGE = 0xFFFFFFFFu;
constexpr int MAX_PASSED = 4;              // candidates a decoder may pass over before it gives up (a decoder fed garbage passes them all)

// A position in the compressed piece: (bit offset << 1) | kind, kind 0 = a DEFLATE block header starts at that bit, kind 1 = a
// gzip member header starts there (bit offset a multiple of 8).
__host__ __device__ inline u64 pos_deflate(u64 bit) { return bit << 1; }
__host__ __device__ inline u64 pos_header(u64 byte) { return (byte << 4) | 1ull; }

enum : int32_t {
    ST_EMPTY = 0,        // no start in this chunk
    ST_TARGET = 1,       // ended exactly on the start of job `next_job`
    ST_NEED_MORE = 2,    // the input ends inside the block / header / trailer behind `end`
    ST_END_INPUT = 3,    // a member's trailer ended exactly at the end of the input
    ST_BAD_HEADER = 4,   // no gzip member header at `end`
    ST_ERROR = 5,        // invalid DEFLATE data behind `end`
    ST_POOL_FULL = 6,
    ST_LOST = 7,         // passed over more than MAX_PASSED candidates: stopped at the block boundary `end`
    ST_EVENTS_FULL = 8,
    ST_TOO_BIG = 10,     // one block's output went beyond what the caller's buffer can take at all (a decoder fed garbage, or a block larger than out_capacity)
    ST_SPLIT = 9,        // its output reached max_job_syms: stopped at the block boundary `end` (bounds what one job can ask of the caller's buffer)
    ST_FAR = 11          // went far_bytes of compressed input beyond its start without meeting a start the finder found (fixed-Huffman / stored blocks
                         // only, or a block of that size): stopped at the block boundary `end` -- such a stretch is ONE wave's work here (~10 MB/s), the host takes it
};

struct Job { u64 start; int32_t cand_from; int32_t pad; };
struct JobOut {
    u64 start, end;            // `end`: the last boundary reached (everything in front of it is decoded and stored)
    u64 out_syms;              // symbols stored up to `end`
    u64 err_bit;               // ST_ERROR: where the decoder stood
    uint32_t first_page, n_pages;
    int32_t status, next_job;
    uint32_t n_events, passed;
};
struct Event { uint32_t job, seq; u64 out_syms; u64 trailer_byte; };   // a member ended: its trailer's offset in the piece
struct Args {
    const uint8_t* comp; int64_t n;          // the piece
    Job* jobs; JobOut* outs;
    int32_t job_base, n_jobs;                // this launch runs jobs [job_base, job_base + n_jobs)
    int32_t n_cand;                          // jobs [0, n_cand) are the chunk starts every decoder compares itself with
    int32_t chunk_bytes;
    uint16_t* pool; uint32_t pool_pages; uint32_t* page_next;
    uint32_t* counters;                      // [0] pages taken, [1] events taken
    Event* events; uint32_t max_events;
    int64_t max_job_syms;                    // a job stops at the first boundary at which it has stored this much
    int64_t hard_cap_syms;                   // ... and gives up in the middle of a block beyond this much (bounds what a wrong guess can take from the pool)
    const uint32_t* order;                   // workgroup k of the decode launch runs job order[k] (nullptr: job_base + k)
    int64_t far_bytes;                       // 0 = no limit; else a job stops (ST_FAR) at the first boundary this far behind its start
};

// ---- the bit reader of one wave (cf. inf::Bits), addressed by absolute bit offset in the piece -------------------------------
struct GBits {
    const uint8_t* base; int64_t limit;
    u64 buf; int cnt;
    int next;               // next dword of the piece to enter buf
    int win_base;
    uint32_t win;           // lane i = dword win_base + i
    int ran_out;            // the reader is well past the end of the input (everything it hands out now is zero bits)
    __device__ __forceinline__ void start(const uint8_t* p, int64_t lim, int64_t bit) {
        base = p; limit = lim; buf = 0; cnt = 0; next = (int)(bit >> 5); win_base = next - 64; win = 0; ran_out = 0;
        refill();
        take((int)(bit & 31));
    }
    __device__ __forceinline__ void refill() {
        while (cnt <= 32) {
            if (next - win_base >= 64) {
                win_base = next;
                const int64_t o = 4ll * (win_base + (int)(threadIdx.x & 63));
                win = o + 4 <= limit ? reinterpret_cast<const U32U*>(base + o)->v
                                     : (o < limit ? (uint32_t)base[o] | (o + 1 < limit ? (uint32_t)base[o + 1] << 8 : 0u) | (o + 2 < limit ? (uint32_t)base[o + 2] << 16 : 0u) : 0u);
                if (4ll * win_base >= limit + 16) ran_out = 1;
            }
            buf |= (u64)rdlane(win, next - win_base) << cnt;
            cnt += 32; ++next;
        }
        pin();
    }
    __device__ __forceinline__ void pin() {
        cnt = (int)uni((uint32_t)cnt); next = (int)uni((uint32_t)next); win_base = (int)uni((uint32_t)win_base); ran_out = (int)uni((uint32_t)ran_out);
        buf = ((u64)uni((uint32_t)(buf >> 32)) << 32) | uni((uint32_t)buf);
    }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = (uint32_t)buf & ((1u << n) - 1u); buf >>= n; cnt -= n; return v; }
    __device__ __forceinline__ int64_t bitpos() const { return 32ll * next - cnt; }
    __device__ __forceinline__ bool beyond() const { return ran_out || bitpos() > 8 * limit; }
};

// ---- the dynamic block header (RFC 1951 3.2.7), with zlib's validity rules ----------------------------------------------------
// On entry the 3 header bits are consumed.  Builds both codes (and the literal/length direct table when `tables`).
__device__ __forceinline__ bool read_dynamic(GBits& b, uint8_t* lens, uint16_t* sym_ll, uint16_t* sym_d, uint32_t* lut2, uint32_t* dlut, Code& ll, Code& dd, bool tables) {
    const int lane = threadIdx.x & 63;
    b.refill();
    const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    if (lane < 19) lens[lane] = 0;
    for (int i = 0; i < hclen; ++i) {
        b.refill();
        const uint32_t v = b.take(3);
        if (lane == 0) lens[inf::CL_ORDER[i]] = (uint8_t)v;
    }
    __builtin_amdgcn_wave_barrier();
    Code cl;
    if (!inf::build_code(lens, 19, sym_d, cl) || !inf::code_valid(cl, false, true)) return false;
    int n = 0;
    uint32_t prev = 0;
    const int total = hlit + hdist;
    while (n < total) {
        b.refill();
        const int s = inf::decode_sym(b, cl, sym_d);
        if (s < 0) return false;
        uint32_t val = 0; int rep = 1;
        if (s < 16) { val = (uint32_t)s; prev = val; }
        else if (s == 16) { if (n == 0) return false; val = prev; rep = 3 + (int)b.take(2); }
        else if (s == 17) { rep = 3 + (int)b.take(3); prev = 0; }
        else { rep = 11 + (int)b.take(7); prev = 0; }
        if (n + rep > total) return false;
        for (int i = lane; i < rep; i += 64) lens[32 + n + i] = (uint8_t)val;
        n += rep;
        if (b.ran_out) return false;
    }
    __builtin_amdgcn_wave_barrier();
    if (lens[32 + 256] == 0) return false;   // no end-of-block code
    if (!inf::build_code(lens + 32, hlit, sym_ll, ll) || !inf::code_valid(ll, false, false)) return false;
    if (!inf::build_code(lens + 32 + hlit, hdist, sym_d, dd) || !inf::code_valid(dd, true, false)) return false;
    if (tables) { inf::build_lut2(ll, sym_ll, lens + 32, lut2); inf::build_dlut(dd, sym_d, lens + 32 + hlit, dlut); }
    return true;
}

// ---- gzip member header (RFC 1952 2.3) at byte B: > 0 = offset of the DEFLATE data, 0 = the input ends inside it, -1 = none ----
__device__ __forceinline__ int64_t parse_member_header(const uint8_t* comp, int64_t n, int64_t B) {
    const int lane = threadIdx.x & 63;
    auto ub = [&](int64_t i) -> uint32_t { return i < n ? uni((uint32_t)comp[i]) : 0u; };
    if (B + 2 <= n && (ub(B) != 0x1fu || ub(B + 1) != 0x8bu)) return -1;
    if (B + 10 > n) return 0;
    if (ub(B + 2) != 8u) return -1;
    const uint32_t flg = ub(B + 3);
    if (flg & 0xE0u) return -1;
    int64_t p = B + 10;
    if (flg & 4u) { if (p + 2 > n) return 0; p += 2 + (int64_t)(ub(p) | (ub(p + 1) << 8)); }
    for (int f = 8; f <= 16; f <<= 1) {   // FNAME, FCOMMENT: zero terminated
        if (!(flg & (uint32_t)f)) continue;
        for (;;) {
            if (p >= n) return 0;
            const u64 m = __ballot(p + lane < n && comp[p + lane] == 0);
            if (m) { p += __builtin_ctzll(m) + 1; break; }
            p += 64;
        }
    }
    if (flg & 2u) p += 2;
    return p < n ? p : 0;   // (at least the first byte of the DEFLATE data must be there)
}

// ---- the symbol loop, by hand (cf. inf::sym_run) ----------------------------------------------------------------------------------
// The same loop for this decoder's output: 16-bit symbols into the CURRENT page (`page` = its first symbol, st.pos = the offset
// of the next symbol in it, 0..65536), a match source in front of the job's output becomes a marker -- which is simply the low
// 16 bits of its (negative) position: 0x8000 | (32768 + q) == q & 0xFFFF for -32768 <= q < 0.  A distance never needs a check
// here (the format ends at 32768, and 32 KiB in front of the job are addressable as markers).  What the asm block does not take
// it hands back untouched: a match that would cross the page's end or whose source lies in the page before, or straddles
// buffer and page (4), long or missing codes (0, 3), the window register running out (2), a page with fewer than two free
// slots (6: the caller decodes one symbol itself).  Matches longer than 63 symbols and matches that overlap themselves (runs:
// common in a sequencer's quality lines) are taken in pieces of up to 63, lane i of a piece reading source symbol i mod dist.
// `first` = the page is the job's first (sources may be markers).
//
// What bounds this kernel is the CU's ONE scalar unit (the microbenchmark experiments/micro/salu_loop.hip and PMC: 24 waves
// of a CU want ~6 scalar instructions per cycle, it issues 1), so the loop is written to need few of them and to put what it
// can on the vector unit: the table index is computed there (v_bfe, v_lshl_add), literals go from the entry register to the
// buffer without passing a scalar register (lanes 0 and 1 are the only active ones between matches: lane 0 writes the
// first literal of the entry, lane 1 the second), a length entry is its own s_bfe operand.  11 scalar instructions per
// lookup of one or two literals, 51 per match (the version before: 15 and 71).
//
// The symbols do not go to the page one match at a time: they collect in an LDS buffer (`obuf`, OB_SLOTS dwords, one symbol
// each) that is written out 64 lanes wide when it holds more than OB_FLUSH.  That is what lets a match NOT wait for its
// source: global_load_lds_ushort puts lane i's symbol straight into buffer slot i of the match (the instruction writes a
// zero-extended DWORD per lane at M0 + 4 * lane, M0 rounded down to a dword: experiments/micro/glds_u16.hip), and the loop
// goes on decoding while it travels.  A source that is still in the buffer is copied inside it (after a wait, if a load into
// those very slots may be in flight: s57 = the lowest slot written by a load since the last vmcnt(0)); one that straddles
// buffer and page is handed back (4).  Every way out writes the buffer out: outside this block the page holds everything in
// front of st.pos.
using inf::OB_SLOTS;
using inf::OB_FLUSH;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ int sym_run_gz(GBits& b, const uint32_t* lut2, const uint32_t* dlut, uint32_t* obuf, uint16_t* page, bool first, inf::SymState& st) {
    uint32_t reason, vt, vt2, vq, ve, vslot, vsrc, ee;
    u64 buf = ((u64)uni((uint32_t)(b.buf >> 32)) << 32) | uni((uint32_t)b.buf);
    int cnt = (int)uni((uint32_t)b.cnt), next = (int)uni((uint32_t)b.next), pos = (int)uni((uint32_t)st.pos);
    int len = (int)uni((uint32_t)st.len), dist = 0;
    const int din = (int)uni((uint32_t)st.dist);   // != 0: the match's distance is known (the caller decoded a long distance code): straight to the copy
    const int wb = (int)uni((uint32_t)b.win_base), fp = (int)uni(first ? 0x40000000u : 0u);   // how far in front of the page a source may lie
    const uint32_t lds = uni((uint32_t)(uintptr_t)lut2), ldd = uni((uint32_t)(uintptr_t)dlut), ldo = uni((uint32_t)(uintptr_t)obuf);
    const u64 ob = ((u64)uni((uint32_t)((uintptr_t)page >> 32)) << 32) | uni((uint32_t)(uintptr_t)page);
    const uint32_t lane = threadIdx.x & 63, lane4 = lane << 2, sh8 = lane << 3;
    const float lh = (float)lane + 0.5f;
    asm volatile(
        "\ts_mov_b64 s[68:69], exec\n"
        "\ts_mov_b64 s[40:41], %[buf]\n"
        "\ts_sub_i32 s42, %[cnt], 33\n"
        "\ts_mov_b32 s43, %[next]\n"
        "\ts_mov_b32 s44, %[pos]\n"
        "\ts_mov_b32 s51, %[len]\n"
        "\ts_mov_b32 s53, %[fp]\n"
        "\ts_mov_b32 s54, %[wb]\n"
        "\ts_mov_b32 s55, %[lds]\n"
        "\ts_mov_b32 s56, %[ldd]\n"
        "\ts_mov_b64 s[60:61], %[ob]\n"
        "\ts_mov_b32 s65, %[obuf]\n"
        "\ts_mov_b32 s52, %[din]\n"
        "\ts_mov_b32 s64, 0\n"
        "\ts_mov_b32 s57, 0x7fffffff\n"
        "\ts_sub_i32 s67, 0xfffe, s44\n"
        "\ts_min_i32 s67, s67, 128\n"
        "\tv_add_u32 %[vslot], s65, %[lane4]\n"
        "\ts_mov_b64 exec, 3\n"
        "\ts_cmp_lg_u32 s52, 0\n"
        "\ts_cbranch_scc1 47f\n"
        "\ts_cmp_lg_u32 s51, 0\n"
        "\ts_cbranch_scc1 4f\n"
        // ---- between symbols: room for two more literals in buffer and page?
        "8:\n"
        "\ts_cmp_le_i32 s64, s67\n"
        "\ts_cbranch_scc1 1f\n"
        "\ts_cmp_eq_u32 s64, 0\n"
        "\ts_cbranch_scc1 12f\n"
        "\ts_mov_b32 s66, 0\n"
        "\ts_branch 30f\n"
        // ---- a symbol: look up (lanes 0 and 1 are the active ones here)
        "1:\n"
        "\ts_cmp_ge_i32 s42, 0\n"
        "\ts_cbranch_scc0 10f\n"
        "2:\n"
        "\tv_bfe_u32 %[vt], s40, 0, 10\n"
        "\tv_lshl_add_u32 %[vt], %[vt], 2, s55\n"
        "\tds_read_b32 %[ve], %[vt]\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tv_readfirstlane_b32 s46, %[ve]\n"
        "\ts_cmp_lt_i32 s46, 0\n"
        "\ts_cbranch_scc1 3f\n"
        // one or two literals: lane 0 writes the first into the next buffer slot, lane 1 the second into the one after (junk, and
        // overwritten by the next symbol, when the entry holds one)
        "\tv_bfe_u32 %[vt2], %[ve], %[sh8], 8\n"
        "\tds_write_b32 %[vslot], %[vt2]\n"
        "\ts_bfe_u32 s47, s46, 0x20018\n"
        "\ts_add_i32 s64, s64, s47\n"
        "\tv_lshl_add_u32 %[vslot], s47, 2, %[vslot]\n"
        "\ts_bfe_u32 s47, s46, 0x50010\n"
        "\ts_lshr_b64 s[40:41], s[40:41], s47\n"
        // (s42 = bits in the buffer - 33, >= 0 here: the subtraction's borrow IS "fewer than 33 left"; then the room test of label 8 in
        // place -- two scalar instructions fewer per lookup than the branch to 8 and its two tests; inf::sym_run_ob has the same)
        "\ts_sub_u32 s42, s42, s47\n"
        "\ts_cbranch_scc1 13f\n"
        "\ts_cmp_le_i32 s64, s67\n"
        "\ts_cbranch_scc1 2b\n"
        "\ts_branch 8b\n"
        // ---- not a literal: a length symbol with its base and extra bits folded into the entry (the entry is the s_bfe operand)
        "3:\n"
        "\ts_bitcmp1_b32 s46, 30\n"
        "\ts_cbranch_scc0 70f\n"
        "\ts_bfe_u32 s48, s40, s46\n"
        "\ts_bfe_u32 s51, s46, 0x90005\n"
        "\ts_add_i32 s51, s51, s48\n"
        "\ts_bfe_u32 s47, s46, 0x50017\n"
        "\ts_lshr_b64 s[40:41], s[40:41], s47\n"
        "\ts_sub_i32 s42, s42, s47\n"
        // ---- the distance
        "4:\n"
        "\ts_cmp_ge_i32 s42, 0\n"
        "\ts_cbranch_scc0 11f\n"
        "\tv_bfe_u32 %[vt], s40, 0, 8\n"
        "\tv_lshl_add_u32 %[vt], %[vt], 2, s56\n"
        "\tds_read_b32 %[ve], %[vt]\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tv_readfirstlane_b32 s46, %[ve]\n"
        "\ts_cmp_lt_i32 s46, 0\n"
        "\ts_cbranch_scc1 83f\n"
        "\ts_bfe_u32 s47, s46, 0x50014\n"
        "\ts_lshr_b64 s[40:41], s[40:41], s47\n"
        "\ts_sub_i32 s42, s42, s47\n"
        "\ts_bfe_u32 s47, s46, 0x40010\n"
        "\ts_bfm_b32 s48, s47, 0\n"
        "\ts_and_b32 s48, s48, s40\n"
        "\ts_and_b32 s52, s46, 0x7fff\n"
        "\ts_add_i32 s52, s52, s48\n"
        "\ts_lshr_b64 s[40:41], s[40:41], s47\n"
        "\ts_sub_i32 s42, s42, s47\n"
        // ---- what stays in here: a match inside this page whose source is in this page (or markers)
        "47:\n"
        "\ts_add_i32 s47, s44, s64\n"
        "\ts_add_i32 s48, s47, s51\n"
        "\ts_cmp_gt_u32 s48, 0x10000\n"
        "\ts_cbranch_scc1 84f\n"
        "\ts_add_i32 s48, s47, s53\n"
        "\ts_cmp_gt_u32 s52, s48\n"
        "\ts_cbranch_scc1 84f\n"
        // ---- the common kind: up to 63 symbols, not overlapping itself (the others: 70, in pieces)
        "\ts_min_u32 s48, s52, 63\n"
        "\ts_cmp_gt_u32 s51, s48\n"
        "\ts_cbranch_scc1 40f\n"
        // where the source lies: dist - len >= what the buffer holds -> all of it is in memory (or in front of the page: markers)
        "\ts_sub_i32 s48, s52, s51\n"
        "\ts_cmp_ge_u32 s48, s64\n"
        "\ts_cbranch_scc0 65f\n"
        // far: lane i < len fetches the symbol at q = pos - dist + i STRAIGHT INTO its buffer slot (LDS-direct load: nothing to wait
        // for, nothing held in registers; the decoding goes on while the symbols travel)
        "\ts_bfm_b64 exec, s51, 0\n"
        "\ts_sub_i32 s47, s47, s52\n"
        "\tv_add_u32 %[vq], s47, %[lane]\n"
        "\ts_lshl2_add_u32 m0, s64, s65\n"
        "\ts_cmp_ge_i32 s47, 0\n"
        "\ts_cbranch_scc0 64f\n"
        "62:\n"
        "\tv_lshlrev_b32 %[vt], 1, %[vq]\n"
        "\ts_min_u32 s57, s57, s64\n"
        "\tglobal_load_lds_ushort %[vt], s[60:61]\n"
        "63:\n"
        "\ts_mov_b64 exec, 3\n"
        "\ts_add_i32 s64, s64, s51\n"
        "\tv_lshl_add_u32 %[vslot], s51, 2, %[vslot]\n"
        "\ts_mov_b32 s51, 0\n"
        "\ts_cmp_le_i32 s64, s67\n"
        "\ts_cbranch_scc1 1b\n"
        "\ts_branch 8b\n"
        // (first page) lanes with q < 0 write the marker q & 0xFFFF themselves
        "64:\n"
        "\tv_cmp_gt_i32 vcc, 0, %[vq]\n"
        "\ts_and_saveexec_b64 s[62:63], vcc\n"
        "\tv_and_b32 %[vt2], 0xffff, %[vq]\n"
        "\ts_lshl2_add_u32 s48, s64, s65\n"
        "\tv_add_u32 %[vt], s48, %[lane4]\n"
        "\tds_write_b32 %[vt], %[vt2]\n"
        "\ts_andn2_b64 exec, s[62:63], vcc\n"
        "\ts_cbranch_execz 63b\n"
        "\ts_branch 62b\n"
        // near: dist <= what the buffer holds -> all of it is in the buffer (slots ob_n - dist ..); anything else straddles: handed back
        "65:\n"
        "\ts_cmp_le_u32 s52, s64\n"
        "\ts_cbranch_scc0 84f\n"
        "\ts_sub_i32 s47, s64, s52\n"
        "\ts_add_i32 s48, s47, s51\n"
        "\ts_cmp_le_u32 s48, s57\n"
        "\ts_cbranch_scc1 66f\n"
        "\ts_waitcnt vmcnt(0)\n"
        "\ts_mov_b32 s57, 0x7fffffff\n"
        "66:\n"
        "\ts_bfm_b64 exec, s51, 0\n"
        "\ts_lshl2_add_u32 s47, s47, s65\n"
        "\tv_add_u32 %[vt], s47, %[lane4]\n"
        "\tds_read_b32 %[vt2], %[vt]\n"
        "\ts_lshl2_add_u32 s48, s64, s65\n"
        "\tv_add_u32 %[vt], s48, %[lane4]\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tds_write_b32 %[vt], %[vt2]\n"
        "\ts_branch 63b\n"
        // ---- a long match, or one that overlaps itself (dist < len: the source repeats with period dist -- runs of one quality value,
        // poly-G tails, duplicate reads: rare in the benchmark's synthetic FASTQ, common in real files): in pieces of up to 63, lane i
        // of a piece taking source symbol i mod dist.  Pieces go through the buffer like any match; a later piece may read what an
        // earlier one wrote (the in-flight test covers that); what straddles buffer and memory is handed back with the rest (4).
        "40:\n"
        "\ts_mov_b64 exec, s[68:69]\n"
        "\ts_min_u32 s70, s51, 63\n"
        "\ts_min_u32 s71, s70, s52\n"
        "\tv_mov_b32 %[vsrc], %[lane]\n"
        "\ts_cmp_le_u32 s70, s52\n"
        "\ts_cbranch_scc1 41f\n"

        // i mod dist = i - dist * floor((i + 0.5) / dist): exact in fp32 for i < 64 (the quotient stays > 0.007 away from an integer)
        "\tv_cvt_f32_u32 %[vt], s52\n"
        "\tv_rcp_f32 %[vt], %[vt]\n"
        "\ts_nop 1\n"   // (a transcendental's result is not interlocked against the next VALU read: experiments/micro/lane_mod.hip)
        "\tv_mul_f32 %[vt], %[lh], %[vt]\n"
        "\tv_cvt_u32_f32 %[vt], %[vt]\n"
        "\tv_mul_lo_u32 %[vt], %[vt], s52\n"
        "\tv_sub_u32 %[vsrc], %[lane], %[vt]\n"
        "41:\n"
        "\ts_sub_i32 s48, s52, s71\n"
        "\ts_cmp_ge_u32 s48, s64\n"
        "\ts_cbranch_scc0 45f\n"
        "\ts_bfm_b64 exec, s70, 0\n"
        "\ts_add_i32 s47, s44, s64\n"
        "\ts_sub_i32 s47, s47, s52\n"
        "\tv_add_u32 %[vq], s47, %[vsrc]\n"
        "\ts_lshl2_add_u32 m0, s64, s65\n"
        "\ts_cmp_ge_i32 s47, 0\n"
        "\ts_cbranch_scc0 44f\n"
        "42:\n"
        "\tv_lshlrev_b32 %[vt], 1, %[vq]\n"
        "\ts_min_u32 s57, s57, s64\n"
        "\tglobal_load_lds_ushort %[vt], s[60:61]\n"
        "43:\n"
        "\ts_mov_b64 exec, 3\n"
        "\ts_add_i32 s64, s64, s70\n"
        "\tv_lshl_add_u32 %[vslot], s70, 2, %[vslot]\n"
        "\ts_sub_i32 s51, s51, s70\n"
        "\ts_cmp_eq_u32 s51, 0\n"
        "\ts_cbranch_scc1 8b\n"
        "\ts_cmp_le_u32 s64, 128\n"
        "\ts_cbranch_scc1 40b\n"
        "\ts_mov_b32 s66, 2\n"
        "\ts_branch 30f\n"
        // (first page) lanes with q < 0 write the marker q & 0xFFFF themselves
        "44:\n"
        "\tv_cmp_gt_i32 vcc, 0, %[vq]\n"
        "\ts_and_saveexec_b64 s[62:63], vcc\n"
        "\tv_and_b32 %[vt2], 0xffff, %[vq]\n"
        "\ts_lshl2_add_u32 s48, s64, s65\n"
        "\tv_add_u32 %[vt], s48, %[lane4]\n"
        "\tds_write_b32 %[vt], %[vt2]\n"
        "\ts_andn2_b64 exec, s[62:63], vcc\n"
        "\ts_cbranch_execz 43b\n"
        "\ts_branch 42b\n"
        "45:\n"
        "\ts_cmp_le_u32 s52, s64\n"
        "\ts_cbranch_scc0 84f\n"
        "\ts_sub_i32 s47, s64, s52\n"
        "\ts_add_i32 s48, s47, s71\n"
        "\ts_cmp_le_u32 s48, s57\n"
        "\ts_cbranch_scc1 46f\n"
        "\ts_waitcnt vmcnt(0)\n"
        "\ts_mov_b32 s57, 0x7fffffff\n"
        "46:\n"
        "\ts_bfm_b64 exec, s70, 0\n"
        "\ts_lshl2_add_u32 s47, s47, s65\n"
        "\tv_lshl_add_u32 %[vt], %[vsrc], 2, s47\n"
        "\tds_read_b32 %[vt2], %[vt]\n"
        "\ts_lshl2_add_u32 s48, s64, s65\n"
        "\tv_add_u32 %[vt], s48, %[lane4]\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tds_write_b32 %[vt], %[vt2]\n"
        "\ts_branch 43b\n"
        // ---- the bit buffer's refills (every ~5 symbols: out of the way)
        "10:\n"
        "\ts_sub_i32 s47, s43, s54\n"
        "\ts_cmp_gt_i32 s47, 63\n"
        "\ts_cbranch_scc1 80f\n"
        "\tv_readlane_b32 s48, %[win], s47\n"
        "\ts_mov_b32 s49, 0\n"
        "\ts_add_i32 s47, s42, 33\n"
        "\ts_lshl_b64 s[48:49], s[48:49], s47\n"
        "\ts_or_b64 s[40:41], s[40:41], s[48:49]\n"
        "\ts_add_i32 s42, s42, 32\n"
        "\ts_add_i32 s43, s43, 1\n"
        "\ts_branch 1b\n"
        "13:\n"
        "\ts_sub_i32 s47, s43, s54\n"
        "\ts_cmp_gt_i32 s47, 63\n"
        "\ts_cbranch_scc1 80f\n"
        "\tv_readlane_b32 s48, %[win], s47\n"
        "\ts_mov_b32 s49, 0\n"
        "\ts_add_i32 s47, s42, 33\n"
        "\ts_lshl_b64 s[48:49], s[48:49], s47\n"
        "\ts_or_b64 s[40:41], s[40:41], s[48:49]\n"
        "\ts_add_i32 s42, s42, 32\n"
        "\ts_add_i32 s43, s43, 1\n"
        "\ts_branch 8b\n"
        "11:\n"
        "\ts_sub_i32 s47, s43, s54\n"
        "\ts_cmp_gt_i32 s47, 63\n"
        "\ts_cbranch_scc1 80f\n"
        "\tv_readlane_b32 s48, %[win], s47\n"
        "\ts_mov_b32 s49, 0\n"
        "\ts_add_i32 s47, s42, 33\n"
        "\ts_lshl_b64 s[48:49], s[48:49], s47\n"
        "\ts_or_b64 s[40:41], s[40:41], s[48:49]\n"
        "\ts_add_i32 s42, s42, 32\n"
        "\ts_add_i32 s43, s43, 1\n"
        "\ts_branch 4b\n"
        // (on the way out through 6 the bit buffer is as full as on every other way out: the caller decodes a symbol from it)
        "12:\n"
        "\ts_cmp_ge_i32 s42, 0\n"
        "\ts_cbranch_scc1 86f\n"
        "\ts_sub_i32 s47, s43, s54\n"
        "\ts_cmp_gt_i32 s47, 63\n"
        "\ts_cbranch_scc1 80f\n"
        "\tv_readlane_b32 s48, %[win], s47\n"
        "\ts_mov_b32 s49, 0\n"
        "\ts_add_i32 s47, s42, 33\n"
        "\ts_lshl_b64 s[48:49], s[48:49], s47\n"
        "\ts_or_b64 s[40:41], s[40:41], s[48:49]\n"
        "\ts_add_i32 s42, s42, 32\n"
        "\ts_add_i32 s43, s43, 1\n"
        "\ts_branch 12b\n"
        // ---- the buffer (s64 symbols, a dword each) to the page: everything in flight has landed first
        "30:\n"
        "\ts_mov_b64 exec, s[68:69]\n"
        "\ts_waitcnt vmcnt(0)\n"
        "\tv_add_u32 %[vq], s44, %[lane]\n"
        "\tv_lshlrev_b32 %[vq], 1, %[vq]\n"
        "\tv_add_u32 %[vt], s65, %[lane4]\n"
        "\tv_cmp_gt_i32 vcc, s64, %[lane]\n"
        "\ts_and_saveexec_b64 s[58:59], vcc\n"
        "\tds_read_b32 %[vt2], %[vt]\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tglobal_store_short %[vq], %[vt2], s[60:61]\n"
        "\ts_mov_b64 exec, s[58:59]\n"
        "\ts_sub_i32 s48, s64, 64\n"
        "\tv_cmp_gt_i32 vcc, s48, %[lane]\n"
        "\ts_and_saveexec_b64 s[58:59], vcc\n"
        "\tds_read_b32 %[vt2], %[vt] offset:256\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tglobal_store_short %[vq], %[vt2], s[60:61] offset:128\n"
        "\ts_mov_b64 exec, s[58:59]\n"
        "\ts_sub_i32 s48, s64, 128\n"
        "\tv_cmp_gt_i32 vcc, s48, %[lane]\n"
        "\ts_and_saveexec_b64 s[58:59], vcc\n"
        "\tds_read_b32 %[vt2], %[vt] offset:512\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tglobal_store_short %[vq], %[vt2], s[60:61] offset:256\n"
        "\ts_mov_b64 exec, s[58:59]\n"
        "\ts_add_i32 s44, s44, s64\n"
        "\ts_mov_b32 s64, 0\n"
        "\ts_mov_b32 s57, 0x7fffffff\n"
        "\ts_sub_i32 s67, 0xfffe, s44\n"
        "\ts_min_i32 s67, s67, 128\n"
        "\tv_add_u32 %[vslot], s65, %[lane4]\n"
        "\ts_cmp_eq_u32 s66, 1\n"
        "\ts_cbranch_scc1 91f\n"
        "\ts_mov_b64 exec, 3\n"
        "\ts_cmp_eq_u32 s66, 2\n"
        "\ts_cbranch_scc1 40b\n"
        "\ts_branch 8b\n"
        // ---- ways out
        "70:\n"
        "\ts_mov_b32 s50, 0\n"
        "\ts_branch 9f\n"
        "80:\n"
        "\ts_mov_b32 s50, 2\n"
        "\ts_branch 9f\n"
        "83:\n"
        "\ts_mov_b32 s50, 3\n"
        "\ts_branch 9f\n"
        "84:\n"
        "\ts_mov_b32 s50, 4\n"
        "\ts_branch 9f\n"
        "86:\n"
        "\ts_mov_b32 s50, 6\n"
        "9:\n"
        "\ts_mov_b32 s66, 1\n"
        "\ts_branch 30b\n"
        "91:\n"
        "\ts_mov_b64 %[buf], s[40:41]\n"
        "\ts_add_i32 %[cnt], s42, 33\n"
        "\ts_mov_b32 %[next], s43\n"
        "\ts_mov_b32 %[pos], s44\n"
        "\ts_mov_b32 %[e], s46\n"
        "\ts_mov_b32 %[len], s51\n"
        "\ts_mov_b32 %[dist], s52\n"
        "\ts_mov_b32 %[reason], s50"
        : [buf] "+s"(buf), [cnt] "+s"(cnt), [next] "+s"(next), [pos] "+s"(pos), [len] "+s"(len), [dist] "+s"(dist),
          [vt] "=&v"(vt), [vt2] "=&v"(vt2), [vq] "=&v"(vq), [ve] "=&v"(ve), [vslot] "=&v"(vslot), [vsrc] "=&v"(vsrc), [e] "=s"(ee), [reason] "=s"(reason)
        : [wb] "s"(wb), [fp] "s"(fp), [din] "s"(din), [win] "v"(b.win), [lane] "v"(lane), [lane4] "v"(lane4), [sh8] "v"(sh8), [lh] "v"(lh), [lds] "s"(lds), [ldd] "s"(ldd), [obuf] "s"(ldo), [ob] "s"(ob)
        : "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",
          "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "m0", "scc", "vcc", "memory");
    b.buf = buf; b.cnt = cnt; b.next = next; st.pos = pos; st.len = len; st.dist = dist; st.e = ee;
    return (int)reason;
}
#pragma clang diagnostic pop

// ---- the decoder of one job ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void run_job(const Args& a, int ji, uint16_t* sym_ll, uint16_t* sym_d, uint8_t* lens, uint32_t* lut2, uint32_t* dlut, uint32_t* obuf) {
    const int lane = threadIdx.x & 63;
    const Job job = a.jobs[ji];
    JobOut o;
    o.start = job.start; o.end = job.start; o.out_syms = 0; o.err_bit = 0; o.first_page = NO_PAGE; o.n_pages = 0; o.status = ST_EMPTY; o.next_job = -1;
    o.n_events = 0; o.passed = 0;
    if (job.start == POS_NONE) { if (lane == 0) a.outs[ji] = o; return; }

    const bool count_them = (a.counters[7] & 1u) != 0;   // (BZQ_GZ_COUNT=1: why sym_run_gz hands back, counted)
    uint32_t lbase, lext, dbase, dext;
    inf::length_dist_tables(lbase, lext, dbase, dext);
    // output: stored symbols [0, opos) + ns pending literals (lane k holds the k-th)
    int64_t opos = 0;
    int ns = 0;
    uint32_t mylit = 0;
    uint16_t *cur = nullptr, *prev = nullptr;
    int64_t cur_idx = -1;
    uint32_t cur_page = NO_PAGE;
    bool pool_full = false, too_big = false;
    auto ensure = [&](int64_t p_end) -> bool {   // pages for positions < p_end
        if (p_end > a.hard_cap_syms) { too_big = true; return false; }
        while (((p_end - 1) >> PAGE_SHIFT) > cur_idx) {
            uint32_t np = 0;
            if (lane == 0) np = atomicAdd(&a.counters[0], 1u);
            np = uni(np);
            if (np >= a.pool_pages) { pool_full = true; return false; }
            if (lane == 0) { a.page_next[np] = NO_PAGE; if (cur_idx >= 0) a.page_next[cur_page] = np; }
            if (cur_idx < 0) o.first_page = np;
            prev = cur; cur = a.pool + ((size_t)np << PAGE_SHIFT); cur_page = np; ++cur_idx; ++o.n_pages;
        }
        return true;
    };
    auto at = [&](int64_t p) -> uint16_t* { return ((p >> PAGE_SHIFT) == cur_idx ? cur : prev) + (p & (PAGE - 1)); };
    auto flush = [&]() -> bool {
        if (ns == 0) return true;
        if (!ensure(opos + ns)) return false;
        if (lane < ns) *at(opos + lane) = (uint16_t)(mylit & 0xFFu);   // (the literal loop leaves whatever sat above bit 7 of its table entry)
        opos += ns; ns = 0;
        return true;
    };

    u64 pos = job.start;
    int j = job.cand_from;
    GBits b;
    bool reader_on = false;
    int32_t status = ST_ERROR;
    Code ll, dd;
    for (;;) {
        // ---- a boundary: everything in front of `pos` is decoded
        if (!flush()) { status = too_big ? ST_TOO_BIG : ST_POOL_FULL; break; }
        o.end = pos; o.out_syms = (u64)opos;
        if (pos != job.start) {
            bool hit = false;
            while (j < a.n_cand) {
                const u64 s = a.jobs[j].start;
                if (s == POS_NONE || s < pos) { if (s != POS_NONE) ++o.passed; ++j; continue; }
                hit = s == pos;
                break;
            }
            if (hit) { status = ST_TARGET; o.next_job = j; break; }
            if (o.passed > (uint32_t)MAX_PASSED) { status = ST_LOST; break; }
            if (opos >= a.max_job_syms) { status = ST_SPLIT; break; }
            if (a.far_bytes && (int64_t)(pos >> 4) - (int64_t)(job.start >> 4) > a.far_bytes) { status = ST_FAR; break; }
        }
        if (pos & 1ull) {   // a member header
            const int64_t B = (int64_t)(pos >> 4);
            if (B >= a.n) { status = ST_END_INPUT; break; }
            const int64_t d = parse_member_header(a.comp, a.n, B);
            if (d < 0) { status = ST_BAD_HEADER; break; }
            if (d == 0) { status = ST_NEED_MORE; break; }
            pos = pos_deflate(8ull * (u64)d);
            reader_on = false;
            continue;
        }
        if (!reader_on) { b.start(a.comp, a.n, (int64_t)(pos >> 1)); reader_on = true; }

        // ---- one block
        bool ok = true;        // false: invalid data (or the input ran out: told apart below)
        b.refill();
        const bool last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 3) ok = false;
        else if (type == 0) {   // stored: to the next byte edge, LEN, ~LEN, LEN bytes
            b.take(b.cnt & 7);
            b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            const int64_t src = b.bitpos() >> 3;
            if ((len ^ nlen) != 0xFFFFu) ok = false;
            else if (src + (int64_t)len > a.n) { ok = false; b.ran_out = 1; }
            else {
                if (!flush()) { status = too_big ? ST_TOO_BIG : ST_POOL_FULL; break; }
                for (int64_t done = 0; done < (int64_t)len && !pool_full && !too_big;) {
                    const int64_t p = opos + done;
                    const int room = PAGE - (int)(p & (PAGE - 1));
                    const int m = (int)((int64_t)len - done < room ? (int64_t)len - done : room);
                    if (!ensure(p + m)) break;
                    uint16_t* dst = cur + (p & (PAGE - 1));
                    for (int i = lane; i < m; i += 64) dst[i] = a.comp[src + done + i];
                    done += m;
                }
                if (pool_full || too_big) { status = too_big ? ST_TOO_BIG : ST_POOL_FULL; break; }
                opos += len;
                b.start(a.comp, a.n, 8 * (src + (int64_t)len));
            }
        } else {
            if (type == 1) {   // fixed code (RFC 1951 3.2.6)
                for (int s = lane; s < 288; s += 64) lens[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
                if (lane < 32) lens[288 + lane] = 5;
                __builtin_amdgcn_wave_barrier();
                if (!inf::build_code(lens, 288, sym_ll, ll) || !inf::build_code(lens + 288, 30, sym_d, dd)) ok = false;
                else { inf::build_lut2(ll, sym_ll, lens, lut2); inf::build_dlut(dd, sym_d, lens + 288, dlut); }
            } else if (!read_dynamic(b, lens, sym_ll, sym_d, lut2, dlut, ll, dd, true)) ok = false;
            // the symbols: sym_run_gz does the common cases inside one asm block; what it hands back is rare
            inf::SymState st{0, 0, 0, 0, 0u};
            auto general_copy = [&](int len, int dist) -> bool {
                // out[opos + i] = out[opos - dist + i]; with dist < len the source repeats with period dist.  A source in front of
                // this job's output is not known yet: a marker, the low 16 bits of its (negative) position.
                if (!flush() || !ensure(opos + len)) return false;
                for (int i = lane; i < len; i += 64) {
                    const int64_t q = opos - dist + (dist >= len ? i : i % dist);
                    *at(opos + i) = q >= 0 ? *at(q) : (uint16_t)(0x8000u | (uint32_t)(32768 + q));
                }
                opos += len;
                return true;
            };
            while (ok) {
                if (!flush()) { ok = false; break; }            // (the block below knows no pending literals)
                if (!ensure(opos + 1)) { ok = false; break; }   // `cur` is the page of position opos
                st.pos = (int)(opos & (PAGE - 1));
                const int why = sym_run_gz(b, lut2, dlut, obuf, cur, cur_idx == 0, st);
                if (count_them && lane == 0) atomicAdd(&a.counters[16 + why], 1u);
                opos = (opos & ~(int64_t)(PAGE - 1)) + st.pos;
                if (why != 4) st.dist = 0;   // (a distance handed in was used; 4 hands the match back: len, dist)
                if (why == 2) { b.refill(); if (b.ran_out) { ok = false; break; } continue; }   // (with st.len set it resumes in the distance half)
                if (why == 4) { if (!general_copy(st.len, st.dist)) { ok = false; break; } st.len = 0; st.dist = 0; continue; }
                if (why == 3) {   // a distance code the direct table does not hold
                    const int ds = inf::decode_sym(b, dd, sym_d);
                    if (ds < 0 || ds > 29) { ok = false; break; }
                    st.dist = (int)(rdlane(dbase, ds) + b.take((int)rdlane(dext, ds)));   // sym_run_gz goes on with the copy
                    continue;
                }
                // why == 0: end of block, a literal / length code longer than the table's index, or no code at all;
                // why == 6: the page has fewer than two slots left -- one symbol the long way (a literal then crosses into the next page here)
                const uint32_t e = why == 6 ? inf::LUT_LONG : st.e & 0xFFFFu;
                int s;
                if (e != inf::LUT_LONG) { s = (int)(e & 0xFFFu); const int l = (int)(e >> 12); b.buf >>= l; b.cnt -= l; }
                else { s = inf::decode_sym(b, ll, sym_ll); if (s < 0) { ok = false; break; } }
                if (s < 256) {   // a literal with a long code
                    mylit = inf::wrlane((uint32_t)s, ns, mylit);
                    if (++ns >= 63 && (!flush() || b.ran_out)) { ok = false; break; }
                    continue;
                }
                if (s == 256) break;
                if (s > 285 || b.ran_out) { ok = false; break; }
                st.len = (int)(rdlane(lbase, s - 257) + b.take((int)rdlane(lext, s - 257)));   // sym_run_gz goes on with its distance
            }
        }
        if (too_big) { status = ST_TOO_BIG; break; }
        if (pool_full) { status = ST_POOL_FULL; break; }
        if (!ok || b.beyond()) {   // invalid data -- unless the reader had run out of input: then the block is just not whole yet
            status = b.beyond() ? ST_NEED_MORE : ST_ERROR;
            o.err_bit = (u64)b.bitpos();
            break;
        }
        if (!flush()) { status = too_big ? ST_TOO_BIG : ST_POOL_FULL; break; }
        const int64_t bit = b.bitpos();
        if (last) {   // member trailer: CRC-32 and ISIZE behind the next byte edge (checked by the host from the event)
            const int64_t T = (bit + 7) >> 3;
            if (T + 8 > a.n) { status = ST_NEED_MORE; break; }
            uint32_t e = 0;
            if (lane == 0) e = atomicAdd(&a.counters[1], 1u);
            e = uni(e);
            if (e >= a.max_events) { status = ST_EVENTS_FULL; break; }
            if (lane == 0) a.events[e] = Event{(uint32_t)ji, o.n_events, (u64)opos, (u64)T};
            ++o.n_events;
            pos = pos_header((u64)(T + 8));
            reader_on = false;
        } else {
            pos = pos_deflate((u64)bit);
        }
    }
    // (on anything but a boundary exit o.end / o.out_syms still name the last boundary: what lies behind it is discarded)
    o.status = status;
    if (lane == 0) a.outs[ji] = o;
}

// One wave per workgroup: a workgroup's LDS and wave slots are held until its LAST wave is done, and the jobs' lengths differ by
// a factor of three and more -- with four jobs to a workgroup the slots sat behind the longest of four.
constexpr int DEC_BLOCK = 64;
static __global__ __launch_bounds__(DEC_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_gz_decode(Args a) {
    __shared__ uint16_t s_ll[288 + 32];
    __shared__ uint32_t s_lut[1 << inf::LUT_BITS];
    __shared__ uint32_t s_dlut[1 << inf::DLUT_BITS];
    __shared__ uint32_t s_ob[OB_SLOTS];   // sym_run_gz's output buffer; between its calls (it leaves it empty) the code lengths of a block header
    static_assert(OB_FLUSH == 128 && OB_SLOTS >= OB_FLUSH + 64 && OB_SLOTS * 4 >= 320 + 64, "the code lengths share the output buffer");
    if ((int)blockIdx.x >= a.n_jobs) return;
    const int k = a.order ? (int)a.order[blockIdx.x] : (int)blockIdx.x;
    run_job(a, a.job_base + k, s_ll, s_ll + 288, reinterpret_cast<uint8_t*>(s_ob), s_lut, s_dlut, s_ob);
}

// ---- one lane's look at a dynamic block header: a NECESSARY condition, cheap ---------------------------------------------------
// The whole-wave judgement (read_dynamic) costs ~50 us per candidate (up to 316 code lengths decoded one after the other), and
// about one bit position in 2 000 passes the code length code test: ~60 candidates per 16 KiB chunk, which was the finder's
// time.  Here every lane decodes the code lengths of ITS candidate by itself -- a 7-bit direct table of the code length code in
// LDS (128 bytes per lane, lanes 33 dwords apart: no bank conflicts), the repeat codes, the Kraft sums of the two codes it
// describes, the end-of-block code -- with zlib's rules (inflate_table).  ~5 000 instructions, so it only pays with the lanes
// FULL: the finder queues its candidates and looks at 64 at once (run by the lane that met a candidate, with 63 idle, it was
// slower than the whole-wave judgement: measured).  It must never refuse what read_dynamic accepts (a wrong refusal would cost
// a chunk its start, i.e. parallelism -- tests/test_gpu_gzip.py::test_the_speculation_is_what_runs watches that); what it
// lets through is still judged by the wave.
constexpr int LANE_LUT_STRIDE = 132;

constexpr int FIND_STAGE = 2048;   // bytes of a chunk in LDS at a time (32 rounds)
__device__ __forceinline__ bool lane_check_dynamic(const uint8_t* comp, int64_t n, int64_t P, u64 w, uint32_t hclen, uint32_t hlit, uint32_t hdist, uint8_t* lut) {
    // code length code: lengths by symbol (RFC 1951 3.2.7 stores them in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15)
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto len_at = [&](uint32_t i) -> uint32_t { return i < hclen ? (uint32_t)(w >> (3 * i)) & 7u : 0u; };
#pragma unroll
    for (int i = 0; i < 19; ++i) {
        const uint32_t l = len_at((uint32_t)i);
#pragma unroll
        for (int q = 1; q < 8; ++q) cnt[q] += l == (uint32_t)q;
    }
    uint32_t next[8];
    next[0] = 0; next[1] = 0;
#pragma unroll
    for (int q = 2; q < 8; ++q) next[q] = (next[q - 1] + cnt[q - 1]) << 1;
    // the direct table: entry = symbol << 3 | length, for all 7-bit patterns (the code is complete: every pattern is covered)
    constexpr uint8_t INV[19] = {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2};   // position of symbol s in that order
#pragma unroll
    for (int sym = 0; sym < 19; ++sym) {
        const uint32_t l = len_at(INV[sym]);
        if (l) {
            uint32_t code = 0;
#pragma unroll
            for (int q = 1; q < 8; ++q) if (l == (uint32_t)q) { code = next[q]; next[q] += 1; }
            const uint32_t r = __builtin_bitreverse32(code) >> (32 - l);
            for (uint32_t k = r; k < 128u; k += 1u << l) lut[k] = (uint8_t)((sym << 3) | l);
        }
    }
    // the code lengths
    const int64_t start = P + 17 + 3 * (int64_t)hclen;
    int64_t nb = start >> 3;
    u64 buf = 0;
    int bc = 0;
    // (two dwords ahead of the one being shifted in; 16 bytes per load and one load ahead was tried: slower)
    auto load = [&](int64_t at) -> uint32_t { return at + 4 <= n + 60 ? reinterpret_cast<const U32U*>(comp + at)->v : 0u; };   // (64 bytes of slack behind the piece)
    uint32_t n0 = load(nb), n1 = load(nb + 4);
    auto refill = [&]() {
        if (bc <= 32) {
            buf |= (u64)n0 << bc; bc += 32; nb += 4;
            n0 = n1; n1 = load(nb + 4);
        }
    };
    refill();
    { const int sk = (int)(start & 7); buf >>= sk; bc -= sk; }
    const int total = (int)(hlit + 257 + hdist + 1), n_ll = (int)hlit + 257;
    int i = 0;
    uint32_t prev = 0, kr_ll = 0, kr_d = 0, c_ll = 0, c_d = 0, eob = 0;
    bool ok = true;
    while (i < total && ok) {   // (a look that stopped after 64 code lengths and left the rest to the wave was tried: 19 times as many candidates reach the wave -- what random bits describe is rarely refused early -- and the kernel took twice the time)
        refill();
        const uint32_t e = lut[(uint32_t)buf & 127u];
        const uint32_t l = e & 7u, sym = e >> 3;
        buf >>= l; bc -= (int)l;
        uint32_t val = 0;
        int rep = 1;
        if (sym < 16u) { val = sym; prev = sym; }
        else if (sym == 16u) { if (i == 0) { ok = false; break; } val = prev; rep = 3 + (int)((uint32_t)buf & 3u); buf >>= 2; bc -= 2; }
        else if (sym == 17u) { rep = 3 + (int)((uint32_t)buf & 7u); buf >>= 3; bc -= 3; prev = 0; }
        else { rep = 11 + (int)((uint32_t)buf & 127u); buf >>= 7; bc -= 7; prev = 0; }
        if (i + rep > total) { ok = false; break; }
        if (val) {
            const int a = i >= n_ll ? 0 : (i + rep <= n_ll ? rep : n_ll - i);
            const uint32_t unit = 32768u >> val;
            kr_ll += (uint32_t)a * unit; c_ll += (uint32_t)a;
            kr_d += (uint32_t)(rep - a) * unit; c_d += (uint32_t)(rep - a);
        }
        if (i <= 256 && 256 < i + rep) eob = val;
        i += rep;
        // an over-subscribed code is refused the moment it is one (zlib: inflate_table returns -1), not after the last length:
        // what random bits describe is over-subscribed within ~20 lengths, and without this every lane of a look walked all
        // ~300 -- the wave was done when its slowest lane was, ~100 us a look, a quarter of the kernel's time
        if (kr_ll > 32768u || kr_d > 32768u) { ok = false; break; }
    }
    if (!ok || eob == 0u) return false;
    if (8 * (nb - 4) + (32 - bc) > 8 * n + 64) return false;   // (ran past the input: generous, the wave's judgement is exact)
    const bool ll_ok = kr_ll == 32768u || (c_ll == 1u && kr_ll == 16384u);
    const bool d_ok = kr_d == 32768u || (c_d == 1u && kr_d == 16384u) || c_d == 0u;
    return ll_ok && d_ok;
}

// ---- the finder's rare, long paths, OUT OF LINE ----------------------------------------------------------------------------------
// Inlined, the judgement's registers pushed the scan loop's scalars into spill lanes (a third of the loop's instructions were
// v_readlane / v_writelane).  A call keeps the two allocations apart; LDS addresses travel as 32-bit offsets so that the
// callee's accesses stay ds_ instructions (a generic pointer parameter would make them flat_).
template <class T> __device__ __forceinline__ uint32_t lds_off(T* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)p; }
template <class T> __device__ __forceinline__ T* lds_ptr(uint32_t off) { return (T*)(__attribute__((address_space(3))) T*)(uintptr_t)off; }

// a dynamic block header at bit Q, judged by the whole wave: all three Huffman codes must be ones zlib accepts
static __device__ __forceinline__ bool judge_dynamic_at(const uint8_t* comp, int64_t n, int64_t Q, uint8_t* lens, uint16_t* sym_ll, uint16_t* sym_d) {
    GBits b;
    Code ll, dd;
    b.start(comp, n, Q + 3);
    return read_dynamic(b, lens, sym_ll, sym_d, nullptr, nullptr, ll, dd, false) && !b.beyond();
}

// the first QN queued candidates (QN <= 64), one per lane: the per-lane look, then the whole wave's judgement of whoever is
// left, in stream order.  Returns the first position that holds, or POS_NONE.
static __device__ __attribute__((noinline)) u64 find_flush(const uint8_t* comp, int64_t n, int64_t lo, uint32_t queue_o, int QN, uint32_t lane_lut_o, uint32_t lock_o,
                                                           uint32_t lens_o, uint32_t sym_ll_o, uint32_t sym_d_o, uint32_t* counter, uint32_t* clocks) {
    const int lane = threadIdx.x & 63;
    const uint32_t* queue = lds_ptr<uint32_t>(queue_o);
    const bool have = lane < QN;
    const int64_t P = 8 * lo + (have ? (int64_t)queue[lane] : 0), byte = P >> 3;
    const int sh = (int)(P & 7);
    const uint32_t d0 = reinterpret_cast<const U32U*>(comp + byte)->v, d1 = reinterpret_cast<const U32U*>(comp + byte + 4)->v;
    const uint32_t d2 = reinterpret_cast<const U32U*>(comp + byte + 8)->v, d3 = reinterpret_cast<const U32U*>(comp + byte + 12)->v;
    const u64 lo64 = (u64)d0 | ((u64)d1 << 32), hi64 = (u64)d2 | ((u64)d3 << 32);
    const u64 v = sh ? (lo64 >> sh) | (hi64 << (64 - sh)) : lo64, vh = hi64 >> sh;
    const u64 w = (v >> 17) | (vh << 47);
    bool ok = have;
    // (the per-lane tables are the workgroup's, not the wave's: 8 KiB each, and with one set per wave the LDS allowed two
    // workgroups per CU -- the scan loop, which is most of the kernel, ran at half the vector unit's rate for want of waves.
    // A look takes ~20 us once or twice per chunk: the waves take turns.)
    uint32_t* lock = lds_ptr<uint32_t>(lock_o);
    const long long t0 = clocks ? clock64() : 0;
    if (lane == 0) while (atomicCAS(lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(8);
    const long long t0b = clocks ? clock64() : 0;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (ok) ok = lane_check_dynamic(comp, n, P, w, ((uint32_t)(v >> 13) & 15u) + 4u, (uint32_t)(v >> 3) & 31u, (uint32_t)(v >> 8) & 31u, lds_ptr<uint8_t>(lane_lut_o) + lane * LANE_LUT_STRIDE);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) atomicExch(lock, 0u);
    const long long t1 = clocks ? clock64() : 0;
    u64 m = __ballot(ok);
    if (counter && lane == 0) atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    u64 got = POS_NONE;
    while (m && got == POS_NONE) {
        const int L = __builtin_ctzll(m);
        m &= m - 1;
        const int64_t Q = 8 * lo + (int64_t)queue[L];
        if (judge_dynamic_at(comp, n, Q, lds_ptr<uint8_t>(lens_o), lds_ptr<uint16_t>(sym_ll_o), lds_ptr<uint16_t>(sym_d_o))) got = pos_deflate((u64)Q);
    }
    if (clocks && lane == 0) {   // (debug: where a flush's time goes, in units of 256 clocks: lock wait, the lanes' look, the wave's judgement)
        atomicAdd(clocks, (uint32_t)((t0b - t0) >> 8)); atomicAdd(clocks + 1, (uint32_t)((t1 - t0b) >> 8)); atomicAdd(clocks + 2, (uint32_t)((clock64() - t1) >> 8));
    }
    return got;
}

// a member header at byte B ... followed by a block header that can be one
static __device__ __attribute__((noinline)) bool find_header(const uint8_t* comp, int64_t n, int64_t B, uint32_t lens_o, uint32_t sym_ll_o, uint32_t sym_d_o) {
    const int64_t dpos = parse_member_header(comp, n, B);
    if (dpos <= 0) return false;
    GBits b;
    Code ll, dd;
    b.start(comp, n, 8 * dpos);
    b.refill();
    (void)b.take(1);
    const uint32_t type = b.take(2);
    if (type == 3) return false;
    if (type == 0) { b.take(b.cnt & 7); b.refill(); const uint32_t len = b.take(16), nlen = b.take(16); return (len ^ nlen) == 0xFFFFu; }
    if (type == 2) return read_dynamic(b, lds_ptr<uint8_t>(lens_o), lds_ptr<uint16_t>(sym_ll_o), lds_ptr<uint16_t>(sym_d_o), nullptr, nullptr, ll, dd, false) && !b.beyond();
    return true;
}

// ---- FIND: the first position in chunk c at which a block (or a member) can start ----------------------------------------------
// 512 bit positions per round, in two stages.  Stage 1, every lane 8 positions: the 13 header bits that need no arithmetic
// (BFINAL = 0, BTYPE = 10, HLIT <= 29, HDIST <= 29: 11 % of random positions pass) and, at byte positions, the member magic.
// Stage 2, the survivors COMPACTED onto the lanes in stream order: 128 bits of the stream each, the code length code must be
// complete (a 19-step Kraft sum) -- run once per 64 survivors instead of once per 64 positions.  Whoever passes that is judged
// by the whole wave, in stream order: all three Huffman codes of the header must be ones zlib accepts.
static __global__ __launch_bounds__(BLOCK) void k_gz_find(Args a) {
    __shared__ uint16_t s_ll[WAVES][288 + 32];
    __shared__ uint8_t s_len[WAVES][320 + 64];
    __shared__ uint16_t s_surv[WAVES][1024 + 64];
    __shared__ uint8_t kraft4[4096];   // four 3-bit code lengths -> their Kraft sum in 1/128 (saturated: anything > 128 is refused anyway)
    __shared__ uint32_t s_queue[WAVES][64 + 64 + 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[WAVES][(FIND_STAGE + 64) / 4];
    __shared__ __attribute__((aligned(4))) uint8_t s_lane_lut[64 * LANE_LUT_STRIDE];   // one set for the workgroup's waves (find_flush)
    __shared__ uint32_t s_lane_lock;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int c = (int)blockIdx.x * WAVES + wave;
    if (threadIdx.x == 0) s_lane_lock = 0;
    for (uint32_t e = threadIdx.x; e < 4096u; e += BLOCK) {
        uint32_t sum = 0;
#pragma unroll
        for (int f = 0; f < 4; ++f) { const uint32_t l = (e >> (3 * f)) & 7u; sum += l ? 128u >> l : 0u; }
        kraft4[e] = (uint8_t)(sum > 200u ? 200u : sum);
    }
    __syncthreads();
    if (c >= a.n_cand || c == 0) return;   // (job 0 is the piece's exact start, written by the host)
    uint16_t* surv = s_surv[wave];
    uint32_t* queue = s_queue[wave];
    uint32_t* stg = s_stage[wave];
    uint8_t* lane_lut = s_lane_lut;
    uint16_t* sym_ll = s_ll[wave];
    uint16_t* sym_d = s_ll[wave] + 288;
    uint8_t* lens = s_len[wave];
    const int64_t lo = (int64_t)c * a.chunk_bytes, hi = lo + a.chunk_bytes < a.n ? lo + a.chunk_bytes : a.n;
    const bool count_them = (a.counters[7] & 1u) != 0, time_them = (a.counters[7] & 2u) != 0;   // (BZQ_GZ_COUNT=1: how many positions survive each stage; =2: where the time goes)
    const long long t_begin = time_them ? clock64() : 0;
    u64 found = POS_NONE;
    // candidates that passed the code length code test, waiting for the per-lane look: positions relative to the chunk's first bit
    int qn = 0, qh = 0;   // (entries queue[qh .. qh + qn))
    const uint32_t judge_lds[3] = {lds_off(lens), lds_off(sym_ll), lds_off(sym_d)};
    auto flush_queue = [&](int QN) -> u64 {
        const u64 got = find_flush(a.comp, a.n, lo, lds_off(queue + qh), QN, lds_off(lane_lut), lds_off(&s_lane_lock), judge_lds[0], judge_lds[1], judge_lds[2], count_them ? a.counters + 5 : nullptr, time_them ? a.counters + 9 : nullptr);
        qh += QN; qn -= QN;
        return got;
    };
    auto judge_header = [&](int64_t B) -> bool { return find_header(a.comp, a.n, B, judge_lds[0], judge_lds[1], judge_lds[2]); };
    // The chunk goes through LDS FIND_STAGE bytes at a time, the next stage's loads in flight while this one is looked at (read
    // from global memory round by round, every round waited ~2 us for 64 bytes: that, not arithmetic, was the finder's time).
    struct Q4 { uint32_t x, y, z, w; };
    struct __attribute__((packed, aligned(1))) Q4U { Q4 v; };
    Q4 r0, r1, r2;
    auto fetch = [&](int64_t s0) {
        const int64_t lim = a.n + 64;   // (64 bytes of slack behind the piece; every byte of the piece lies >= 64 in front of lim)
        const int64_t p0 = s0 + 16 * lane, p1 = p0 + FIND_STAGE / 2, p2 = s0 + FIND_STAGE + 16 * lane;
        r0 = p0 + 16 <= lim ? reinterpret_cast<const Q4U*>(a.comp + p0)->v : Q4{0, 0, 0, 0};
        r1 = p1 + 16 <= lim ? reinterpret_cast<const Q4U*>(a.comp + p1)->v : Q4{0, 0, 0, 0};
        r2 = (lane < 4 && p2 + 16 <= lim) ? reinterpret_cast<const Q4U*>(a.comp + p2)->v : Q4{0, 0, 0, 0};
    };
    fetch(lo);
    for (int64_t s0 = lo; s0 < hi && found == POS_NONE; s0 += FIND_STAGE) {
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<Q4*>(stg + 4 * lane) = r0;
    *reinterpret_cast<Q4*>(stg + FIND_STAGE / 8 + 4 * lane) = r1;
    if (lane < 4) *reinterpret_cast<Q4*>(stg + FIND_STAGE / 4 + 4 * lane) = r2;
    __builtin_amdgcn_wave_barrier();
    if (s0 + FIND_STAGE < hi) fetch(s0 + FIND_STAGE);
    const int64_t s1 = s0 + FIND_STAGE < hi ? s0 + FIND_STAGE : hi;
    // limits relative to the stage's first bit (32-bit arithmetic in the loops)
    const int lim_in = (int)(8 * (s1 - s0));                                              // positions of this chunk
    const int lim_n = (int)(8 * a.n - 8 * s0 < (1 << 20) ? 8 * a.n - 8 * s0 : (1 << 20));   // bits of the piece
    for (int gb = 0; 8 * gb < lim_in && found == POS_NONE; gb += 128) {
        // stage 1, sixteen positions per lane AT ONCE: bit p of x is the first bit of candidate p, and the thirteen header bits
        // that need no arithmetic are tested for all sixteen with word operations --
        //   BFINAL = 0, BTYPE = 10:  x[p] = 0, x[p+1] = 0, x[p+2] = 1
        //   HLIT <= 29:  not all of x[p+4 .. p+7];   HDIST <= 29:  not all of x[p+9 .. p+12]
        const int bo = gb + 2 * lane;   // (the lane's 32-bit window: bytes bo .. bo + 3 of the stage)
        const uint32_t x = __builtin_amdgcn_alignbyte(stg[(bo >> 2) + 1], stg[bo >> 2], (uint32_t)bo & 3u);
        const uint32_t p2 = x & (x >> 1), ones4 = p2 & (p2 >> 2);   // ones4[i]: x[i .. i+3] all set
        const int rem = lim_in - 8 * bo;   // how many of the lane's positions belong to the chunk
        uint32_t m = (x >> 2) & ~(x | (x >> 1)) & ~(ones4 >> 4) & ~(ones4 >> 9) & (rem >= 16 ? 0xFFFFu : rem <= 0 ? 0u : (1u << rem) - 1u);
        // member magic (1F 8B 08) at the lane's two bytes; the flag byte is judge_header's business
        const bool hdr0 = (x & 0xFFFFFFu) == 0x088B1Fu && rem > 0, hdr1 = (x >> 8) == 0x088B1Fu && rem > 8;
        const u64 h0 = __ballot(hdr0), h1 = __ballot(hdr1);
        // every lane's survivors go to the wave's list in stream order: exclusive prefix of the counts (<= 16: five ballots)
        const uint32_t cnt = (uint32_t)__builtin_popcount(m);
        uint32_t off = 0;
        int total = 0;
#pragma unroll
        for (int bit = 0; bit < 5; ++bit) {
            const u64 bm = __ballot((cnt >> bit) & 1u);
            off += __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u)) << bit;
            total += __builtin_popcountll(bm) << bit;
        }
        while (m) {   // (as many trips as the fullest lane has survivors: ~5 of 16)
            surv[off++] = (uint16_t)(16 * lane + __builtin_ctz(m));
            m &= m - 1;
        }
        __builtin_amdgcn_wave_barrier();
        if (count_them && lane == 0) atomicAdd(&a.counters[3], (uint32_t)total);
        const bool any_hdr = (h0 | h1) != 0, last_group = 8 * (gb + 128) >= lim_in && s1 == hi;
        // stage 2: survivor k of the group (stream order) goes to lane k & 63; who passes is queued
        for (int k0 = 0;; k0 += 64) {
            if (k0 < total) {
                const int k = k0 + lane;
                const bool have = k < total;
                const int sp = have ? (int)surv[k] : 0;
                const int sh = sp & 7, sb = gb + (sp >> 3);
                const uint32_t* e = stg + (sb >> 2);
                const uint32_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4], by = (uint32_t)sb & 3u;
                const uint32_t d0 = __builtin_amdgcn_alignbyte(e1, e0, by), d1 = __builtin_amdgcn_alignbyte(e2, e1, by);
                const uint32_t d2 = __builtin_amdgcn_alignbyte(e3, e2, by), d3 = __builtin_amdgcn_alignbyte(e4, e3, by);
                const u64 lo64 = (u64)d0 | ((u64)d1 << 32), hi64 = (u64)d2 | ((u64)d3 << 32);
                const u64 v = sh ? (lo64 >> sh) | (hi64 << (64 - sh)) : lo64, vh = hi64 >> sh;
                const uint32_t hclen = ((uint32_t)(v >> 13) & 15u) + 4u;
                // the code length code must be complete: its 19 (hclen stored) 3-bit lengths, four at a time through the table
                const u64 w = ((v >> 17) | (vh << 47)) & ((1ull << (3 * hclen)) - 1ull);
                const uint32_t kraft = (uint32_t)kraft4[(uint32_t)w & 4095u] + kraft4[(uint32_t)(w >> 12) & 4095u] + kraft4[(uint32_t)(w >> 24) & 4095u] +
                                       kraft4[(uint32_t)(w >> 36) & 4095u] + kraft4[(uint32_t)(w >> 48) & 4095u];
                const int rel = 8 * gb + sp;   // (relative to the stage)
                const bool pass2 = have && kraft == 128u && rel + 17 + 3 * (int)hclen <= lim_n;
                const u64 m2 = __ballot(pass2);
                if (count_them && lane == 0) atomicAdd(&a.counters[4], (uint32_t)__builtin_popcountll(m2));
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
                if (pass2) queue[qh + qn + (int)rank] = (uint32_t)(8 * (s0 - lo) + rel);
                qn += __builtin_popcountll(m2);
                __builtin_amdgcn_wave_barrier();
            }
            // the ONE place the queue is looked at (the judgement is long code: one copy of it): full lanes; or, behind the
            // group's last survivors, a member header in the group or the end of the chunk -- then whatever there is
            const bool last = k0 + 64 >= total;
            const bool drain = last && (any_hdr || last_group);
            while (found == POS_NONE && (qn >= 64 || (drain && qn > 0))) found = flush_queue(qn < 64 ? qn : 64);
            if (qh) {   // what is left (< 64) moves to the front
                const uint32_t t = lane < qn ? queue[qh + lane] : 0u;
                __builtin_amdgcn_wave_barrier();
                if (lane < qn) queue[lane] = t;
                qh = 0;
                __builtin_amdgcn_wave_barrier();
            }
            if (last || found != POS_NONE) break;
        }
        // member headers of the group (rare), in stream order; everything queued in front of them has been looked at
        u64 ha = h0, hb = h1;
        while ((ha | hb) && found == POS_NONE) {
            const int ia = ha ? 2 * __builtin_ctzll(ha) : 1 << 20, ib = hb ? 2 * __builtin_ctzll(hb) + 1 : 1 << 20;
            const int ib_ = ia < ib ? ia : ib;
            if (ia < ib) ha &= ha - 1; else hb &= hb - 1;
            const int64_t B = s0 + gb + ib_;
            if (B + 18 <= a.n && judge_header(B)) found = pos_header((u64)B);
        }
        __builtin_amdgcn_wave_barrier();   // (the next group rewrites the list)
    }
    }
    if (time_them && lane == 0) atomicAdd(&a.counters[12], (uint32_t)((clock64() - t_begin) >> 8));
    if (lane == 0) a.jobs[c] = Job{found, c + 1, 0};
}

// ---- ORDER: the longest jobs first ---------------------------------------------------------------------------------------------
// A job runs from its chunk's start to the next chunk that has one: mostly one or two chunks, sometimes many (a chunk inside a
// long block has no start).  Workgroups are dispatched in index order, so with the jobs in chunk order the launch ended with
// a few long jobs running alone (the decode kernel took twice the time its work divided by the machine's wave slots).  Two tiny
// launches sort the chunks by the distance to the next start, descending (a counting sort over 64 bins; chunks without a start
// last): longest processing time first.
constexpr int ORDER_BINS = 64;
static __global__ __launch_bounds__(BLOCK) void k_gz_span(const Job* jobs, int n, uint8_t* keys, uint32_t* bins) {
    __shared__ uint32_t s_bins[ORDER_BINS];   // (one global atomic per bin and workgroup: 16 384 threads adding to the same few words took 0.15 ms)
    if (threadIdx.x < ORDER_BINS) s_bins[threadIdx.x] = 0;
    __syncthreads();
    const int c = (int)(blockIdx.x * BLOCK + threadIdx.x);
    if (c < n) {
        int key = 0;
        if (jobs[c].start != POS_NONE) {
            key = 1;
            while (key < ORDER_BINS - 1 && c + key < n && jobs[c + key].start == POS_NONE) ++key;
        }
        keys[c] = (uint8_t)key;
        atomicAdd(&s_bins[key], 1u);
    }
    __syncthreads();
    if (threadIdx.x < ORDER_BINS && s_bins[threadIdx.x]) atomicAdd(&bins[threadIdx.x], s_bins[threadIdx.x]);
}
static __global__ __launch_bounds__(BLOCK) void k_gz_order(int n, const uint8_t* keys, const uint32_t* bins, uint32_t* cursor, uint32_t* order) {
    __shared__ uint32_t base[ORDER_BINS], s_cnt[ORDER_BINS], s_at[ORDER_BINS];
    if (threadIdx.x < ORDER_BINS) {
        uint32_t b = 0;
        for (int k = ORDER_BINS - 1; k > (int)threadIdx.x; --k) b += bins[k];
        base[threadIdx.x] = b;
        s_cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    const int c = (int)(blockIdx.x * BLOCK + threadIdx.x);
    const int key = c < n ? keys[c] : 0;
    uint32_t mine = 0;
    if (c < n) mine = atomicAdd(&s_cnt[key], 1u);                    // rank among the workgroup's chunks of this key
    __syncthreads();
    if (threadIdx.x < ORDER_BINS && s_cnt[threadIdx.x]) s_at[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_cnt[threadIdx.x]);   // the workgroup's range in the bin
    __syncthreads();
    if (c < n) order[base[key] + s_at[key] + mine] = (uint32_t)c;
}

static __global__ void k_gz_job0(Job* jobs, u64 start) { jobs[0] = Job{start, 1, 0}; }   // (job 0 = the piece's exact start)

// starts found on a grid laid over the piece's NEW bytes before its carry was known (gz_stage): jobs[1 ..] += shift, in position units
static __global__ __launch_bounds__(BLOCK) void k_gz_shift(Job* jobs, int n, long long shift) {
    const int i = (int)(blockIdx.x * BLOCK + threadIdx.x);
    if (i < n && jobs[i].start != POS_NONE) jobs[i].start = (u64)((long long)jobs[i].start + shift);
}

// ORDER of a piece's jobs on `st` (jobs[0] set, the others found); order_buf = [bins][cursor][order n_chunks][keys n_chunks].  Returns the order array.
static inline const uint32_t* launch_order(hipStream_t st, const Job* jobs, void* order_buf, int n_chunks) {
    uint32_t* bins = (uint32_t*)order_buf;
    uint32_t* order = bins + 2 * ORDER_BINS;
    uint8_t* keys = (uint8_t*)(order + n_chunks);
    (void)hipMemsetAsync(bins, 0, 2 * ORDER_BINS * 4, st);
    const unsigned g1 = (unsigned)((n_chunks + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_gz_span, dim3(g1), dim3(BLOCK), 0, st, jobs, n_chunks, keys, bins);
    hipLaunchKernelGGL(k_gz_order, dim3(g1), dim3(BLOCK), 0, st, n_chunks, (const uint8_t*)keys, (const uint32_t*)bins, bins + ORDER_BINS, order);
    return order;
}
// FIND + ORDER of a piece on `st`: jobs[0] must be set.
static inline const uint32_t* launch_find(hipStream_t st, const Args& a, void* order_buf, int n_chunks) {
    hipLaunchKernelGGL(k_gz_find, dim3((unsigned)((n_chunks + WAVES - 1) / WAVES)), dim3(BLOCK), 0, st, a);
    return launch_order(st, (const Job*)a.jobs, order_buf, n_chunks);
}

// ---- CHAIN: the window behind every chain chunk; its tail (the last <= 32 KiB) goes out final ---------------------------------------
// The window behind chunk i is a function of the window in front of it: new[k] = a byte the chunk wrote, or old[index] where it
// wrote a marker.  Such functions compose (a table of 32768 symbols whose markers point into the OLDER window), so the chain
// is not walked serially over all chunks: the chain is cut into groups of ~sqrt(n) chunks;
//   pass 1 (k_gz_chain<true>, one workgroup per group, all groups at once): the composed table of the group, in LDS as 16-bit
//          symbols, starting from the identity;
//   pass 2 (k_gz_chain_groups, one workgroup): the window in front of every group, group table by group table;
//   pass 3 (k_gz_chain<false>, one workgroup per group): the walk proper, from the group's true entry window, tails to the output.
struct ChainItem { int64_t out_base, len; uint32_t page_a, page_b; };   // pages of positions len - t and len - 1, t = min(len, 32768)
constexpr int CHAIN_THREADS = 1024;
struct __attribute__((packed, aligned(2))) V16A2 { uint32_t w[4]; };   // 16 bytes at a 2-byte aligned address
struct __attribute__((packed, aligned(1))) V16A1 { uint32_t w[4]; };   // 16 bytes anywhere
template <bool SYM> struct ChainWin { typedef uint8_t T; };
template <> struct ChainWin<true> { typedef uint16_t T; };

// One step per chain chunk: thread i makes entries [32 i, 32 i + 32) of the new window -- the old window shifted by the chunk's
// output, or (the usual case: the chunk wrote 32 KiB or more) 32 symbols of its tail, loaded as four 16-byte pieces, markers
// looked up in the old window (LDS).  What a thread reads of an item depends on the item alone, not on the window: the symbols
// of item i + 1 are fetched while item i is worked on.
template <bool SYM>
static __global__ __launch_bounds__(CHAIN_THREADS) void k_gz_chain(const ChainItem* items, int n_items, int per_group, const uint16_t* pool,
                                                                   const uint8_t* entry_win, uint8_t* out, uint16_t* group_maps, uint8_t* w_next) {
    typedef typename ChainWin<SYM>::T W;
    __shared__ __attribute__((aligned(16))) W win[2][32768];
    const int tid = threadIdx.x;
    const int k0 = tid * 32;
    const int g = (int)blockIdx.x;
    const int i_lo = g * per_group, i_hi = i_lo + per_group < n_items ? i_lo + per_group : n_items;
    if (SYM) {
        for (int k = tid; k < 32768; k += CHAIN_THREADS) win[0][k] = (W)(0x8000u | (uint32_t)k);
    } else {
        const uint8_t* w0 = entry_win + (size_t)g * 32768;
        for (int k = tid * 16; k < 32768; k += CHAIN_THREADS * 16) *reinterpret_cast<uint4*>(&win[0][k]) = *reinterpret_cast<const uint4*>(w0 + k);
    }
    __syncthreads();
    int cur = 0;
    struct Fetch { ChainItem it; int64_t p0; bool fast; uint32_t r[16]; };
    auto fetch = [&](int i, Fetch& f) {
        f.it = items[i];
        const int t = (int)(f.it.len < 32768 ? f.it.len : 32768);
        const int64_t first = f.it.len - t;
        f.p0 = first + (k0 - (32768 - t));
        f.fast = k0 >= 32768 - t && (f.p0 >> PAGE_SHIFT) == ((f.p0 + 31) >> PAGE_SHIFT);
        if (f.fast) {
            const uint32_t pg = (f.p0 >> PAGE_SHIFT) == (first >> PAGE_SHIFT) ? f.it.page_a : f.it.page_b;
            const uint16_t* src = pool + ((size_t)pg << PAGE_SHIFT) + (f.p0 & (PAGE - 1));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const V16A2 v = *reinterpret_cast<const V16A2*>(src + 8 * q);
                f.r[4 * q] = v.w[0]; f.r[4 * q + 1] = v.w[1]; f.r[4 * q + 2] = v.w[2]; f.r[4 * q + 3] = v.w[3];
            }
        }
    };
    Fetch fa, fb;
    if (i_lo < i_hi) fetch(i_lo, fa);
    for (int i = i_lo; i < i_hi; ++i) {
        if (i + 1 < i_hi) fetch(i + 1, fb);
        const ChainItem it = fa.it;
        const int t = (int)(it.len < 32768 ? it.len : 32768);
        const int64_t first = it.len - t;
        const int shift = 32768 - t;      // new[k] = old[k + t] for k < shift, the tail's symbol k - shift behind it
        const W* ow = win[cur];
        W* nw = win[cur ^ 1];
        if (fa.fast) {
            uint32_t a[32];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                uint32_t a0 = fa.r[q] & 0xFFFFu, a1 = fa.r[q] >> 16;
                if (fa.r[q] & 0x80008000u) {
                    if (a0 & 0x8000u) a0 = ow[a0 & 0x7FFFu];
                    if (a1 & 0x8000u) a1 = ow[a1 & 0x7FFFu];
                }
                a[2 * q] = a0; a[2 * q + 1] = a1;
            }
            if (SYM) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<uint4*>(nw + k0 + 8 * q) = uint4{a[8 * q] | (a[8 * q + 1] << 16), a[8 * q + 2] | (a[8 * q + 3] << 16),
                                                                       a[8 * q + 4] | (a[8 * q + 5] << 16), a[8 * q + 6] | (a[8 * q + 7] << 16)};
            } else {
                uint32_t o8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) o8[q] = a[4 * q] | (a[4 * q + 1] << 8) | (a[4 * q + 2] << 16) | (a[4 * q + 3] << 24);
                *reinterpret_cast<uint4*>(nw + k0) = uint4{o8[0], o8[1], o8[2], o8[3]};
                *reinterpret_cast<uint4*>(nw + k0 + 16) = uint4{o8[4], o8[5], o8[6], o8[7]};
                uint8_t* dst = out + it.out_base + fa.p0;
                *reinterpret_cast<V16A1*>(dst) = V16A1{{o8[0], o8[1], o8[2], o8[3]}};
                *reinterpret_cast<V16A1*>(dst + 16) = V16A1{{o8[4], o8[5], o8[6], o8[7]}};
            }
        } else {
            for (int k = k0; k < k0 + 32; ++k) {
                if (k < shift) { nw[k] = ow[k + t]; continue; }
                const int64_t p = first + (k - shift);
                const uint32_t pg = (p >> PAGE_SHIFT) == (first >> PAGE_SHIFT) ? it.page_a : it.page_b;
                const uint32_t sym = pool[((size_t)pg << PAGE_SHIFT) + (p & (PAGE - 1))];
                const W v = (sym & 0x8000u) ? ow[sym & 0x7FFFu] : (W)sym;
                nw[k] = v;
                if (!SYM) out[it.out_base + p] = (uint8_t)v;
            }
        }
        __syncthreads();
        cur ^= 1;
        fa = fb;
    }
    if (SYM) {
        uint16_t* dst = group_maps + (size_t)g * 32768;
        for (int k = tid * 8; k < 32768; k += CHAIN_THREADS * 8) *reinterpret_cast<uint4*>(dst + k) = *reinterpret_cast<const uint4*>(&win[cur][k]);
    } else if (i_hi == n_items) {   // the last group (or the only one): the window in front of the next piece
        for (int k = tid * 16; k < 32768; k += CHAIN_THREADS * 16) *reinterpret_cast<uint4*>(w_next + k) = *reinterpret_cast<const uint4*>(&win[cur][k]);
    }
}

// pass 2: entry_win[g] = the window in front of group g, from w0 through the groups' composed tables
static __global__ __launch_bounds__(CHAIN_THREADS) void k_gz_chain_groups(const uint16_t* group_maps, int n_groups, const uint8_t* w0, uint8_t* entry_win) {
    __shared__ __attribute__((aligned(16))) uint8_t win[2][32768];
    const int tid = threadIdx.x;
    const int k0 = tid * 32;
    for (int k = tid * 16; k < 32768; k += CHAIN_THREADS * 16) *reinterpret_cast<uint4*>(&win[0][k]) = *reinterpret_cast<const uint4*>(w0 + k);
    __syncthreads();
    int cur = 0;
    uint4 ra[4], rb[4];
    auto fetch = [&](int g, uint4* r) {
        const uint4* src = reinterpret_cast<const uint4*>(group_maps + (size_t)g * 32768 + k0);
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = src[q];
    };
    if (n_groups > 0) fetch(0, ra);
    for (int g = 0; g < n_groups; ++g) {
        if (g + 1 < n_groups) fetch(g + 1, rb);
        const uint8_t* ow = win[cur];
        uint8_t* nw = win[cur ^ 1];
        uint8_t* ew = entry_win + (size_t)g * 32768;
        *reinterpret_cast<uint4*>(ew + k0) = *reinterpret_cast<const uint4*>(ow + k0);
        *reinterpret_cast<uint4*>(ew + k0 + 16) = *reinterpret_cast<const uint4*>(ow + k0 + 16);
        const uint32_t r[16] = {ra[0].x, ra[0].y, ra[0].z, ra[0].w, ra[1].x, ra[1].y, ra[1].z, ra[1].w, ra[2].x, ra[2].y, ra[2].z, ra[2].w, ra[3].x, ra[3].y, ra[3].z, ra[3].w};
        uint32_t o8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uint32_t a0 = r[2 * q] & 0xFFFFu, a1 = r[2 * q] >> 16, a2 = r[2 * q + 1] & 0xFFFFu, a3 = r[2 * q + 1] >> 16;
            if ((r[2 * q] | r[2 * q + 1]) & 0x80008000u) {
                if (a0 & 0x8000u) a0 = ow[a0 & 0x7FFFu];
                if (a1 & 0x8000u) a1 = ow[a1 & 0x7FFFu];
                if (a2 & 0x8000u) a2 = ow[a2 & 0x7FFFu];
                if (a3 & 0x8000u) a3 = ow[a3 & 0x7FFFu];
            }
            o8[q] = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
        }
        *reinterpret_cast<uint4*>(nw + k0) = uint4{o8[0], o8[1], o8[2], o8[3]};
        *reinterpret_cast<uint4*>(nw + k0 + 16) = uint4{o8[4], o8[5], o8[6], o8[7]};
        __syncthreads();
        cur ^= 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) ra[q] = rb[q];
    }
}

// ---- CHAIN, the form that runs BESIDE the decoders (round 5) ------------------------------------------------------------------------
// The three kernels above keep the window in LDS (64 / 128 KiB, 1024 threads a workgroup): beside the NEXT piece's decoders -- 24
// one-wave workgroups of 80 VGPRs and 6.5 KiB of LDS a CU, i.e. 480 of a SIMD's 512 VGPRs and 157 of 160 KiB -- such a workgroup
// gets a CU only when the decoders' queue has drained (k_gz_chain_groups: 0.28 ms alone, 8.9 ms there), so a piece's call returned
// when the next piece's decoders were through.  What a CU always has left beside them is one wave of <= 32 VGPRs per SIMD and
// 4 KiB of LDS: these kernels are 256 threads, no LDS, <= 32 VGPRs, and the window lives where the L2 holds it:
//   pass 1 (k_gz_chainl_tab, a workgroup per group): the composed table of every job goes to one of two 64 KiB tables of the
//          group in global memory, the lookups of the next job read it from there; the last job's table is the group's;
//   pass 2 (k_gz_chainl_groups, one workgroup): entry_win[g + 1] = the group's table looked up in entry_win[g] -- the window IS
//          the array the previous step wrote;
//   pass 3 (k_gz_chainl_out, a workgroup per group): a job's tail goes out final, markers read the 32 KiB in front of the job
//          from the OUTPUT itself (this workgroup wrote them one step earlier) or, in front of the group's first job, from
//          entry_win[g].  No table at all.
// A step's stores are visible to the step behind it through the barrier (one CU, one vector L1: workgroup scope).
constexpr int CHL_THREADS = 256;
constexpr int CHL_VEC = 32768 / 8 / CHL_THREADS;   // 16-byte vectors of a 32768-symbol table per thread: 16
struct __attribute__((packed, aligned(1))) U64A1 { u64 v; };

__device__ __forceinline__ const uint16_t* chl_sym(const ChainItem& it, const uint16_t* pool, int64_t first, int64_t p) {
    const uint32_t pg = (p >> PAGE_SHIFT) == (first >> PAGE_SHIFT) ? it.page_a : it.page_b;
    return pool + ((size_t)pg << PAGE_SHIFT) + (p & (PAGE - 1));
}

static __global__ __launch_bounds__(CHL_THREADS) void k_gz_chainl_tab(const ChainItem* items, int n_items, int per_group, const uint16_t* pool,
                                                                      uint16_t* tabs, uint16_t* group_maps) {
    const int tid = threadIdx.x, g = (int)blockIdx.x;
    const int i_lo = g * per_group, i_hi = i_lo + per_group < n_items ? i_lo + per_group : n_items;
    const uint16_t* ow = nullptr;   // the table in front of the job at hand; nullptr = the identity (the group's first job)
    for (int i = i_lo; i < i_hi; ++i) {
        const ChainItem it = items[i];
        const int t = (int)(it.len < 32768 ? it.len : 32768);
        const int64_t first = it.len - t;
        const int shift = 32768 - t;      // new[k] = old[k + t] for k < shift, the tail's symbol k - shift behind it
        uint16_t* nw = i == i_hi - 1 ? group_maps + (size_t)g * 32768 : tabs + ((size_t)g * 2 + (size_t)((i - i_lo) & 1)) * 32768;
        auto look = [&](uint32_t sym) -> uint32_t { return (sym & 0x8000u) && ow ? (uint32_t)ow[sym & 0x7FFFu] : sym; };
#pragma unroll 1
        for (int r = 0; r < CHL_VEC; ++r) {
            const int k0 = (r * CHL_THREADS + tid) * 8;
            uint32_t a[8];
            const int64_t p0 = first + (k0 - shift);
            if (k0 + 8 <= shift) {   // the old table, shifted (a job that wrote less than 32 KiB)
                if (ow) {
                    const V16A2 v = *reinterpret_cast<const V16A2*>(ow + k0 + t);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a[2 * q] = v.w[q] & 0xFFFFu; a[2 * q + 1] = v.w[q] >> 16; }
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = 0x8000u | (uint32_t)(k0 + t + q);
                }
            } else if (k0 >= shift && (p0 >> PAGE_SHIFT) == ((p0 + 7) >> PAGE_SHIFT)) {
                const V16A2 v = *reinterpret_cast<const V16A2*>(chl_sym(it, pool, first, p0));
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[2 * q] = v.w[q] & 0xFFFFu; a[2 * q + 1] = v.w[q] >> 16; }
                if (ow && ((v.w[0] | v.w[1] | v.w[2] | v.w[3]) & 0x80008000u)) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (a[q] & 0x8000u) a[q] = ow[a[q] & 0x7FFFu];
                }
            } else {   // the vector straddles the shift or a page: symbol by symbol, stored at once (rare; kept out of the registers of the paths above)
#pragma unroll 1
                for (int q = 0; q < 8; ++q) {
                    const int k = k0 + q;
                    uint32_t e;
                    if (k < shift) e = ow ? (uint32_t)ow[k + t] : (0x8000u | (uint32_t)(k + t));
                    else e = look(*chl_sym(it, pool, first, first + (k - shift)));
                    nw[k] = (uint16_t)e;
                }
                continue;
            }
            *reinterpret_cast<uint4*>(nw + k0) = uint4{a[0] | (a[1] << 16), a[2] | (a[3] << 16), a[4] | (a[5] << 16), a[6] | (a[7] << 16)};
        }
        __syncthreads();
        ow = nw;
    }
}

// pass 2: entry_win[0] = w0, entry_win[g + 1][k] = group g's table entry k, markers looked up in entry_win[g]
static __global__ __launch_bounds__(CHL_THREADS) void k_gz_chainl_groups(const uint16_t* group_maps, int n_groups, const uint8_t* w0, uint8_t* entry_win) {
    const int tid = threadIdx.x;
    for (int k = tid * 16; k < 32768; k += CHL_THREADS * 16) *reinterpret_cast<uint4*>(entry_win + k) = *reinterpret_cast<const uint4*>(w0 + k);
    __syncthreads();
    for (int g = 0; g + 1 < n_groups; ++g) {
        const uint8_t* ow = entry_win + (size_t)g * 32768;
        uint8_t* nw = entry_win + (size_t)(g + 1) * 32768;
        const uint16_t* map = group_maps + (size_t)g * 32768;
#pragma unroll 2
        for (int r = 0; r < CHL_VEC; ++r) {
            const int k0 = (r * CHL_THREADS + tid) * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(map + k0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t a[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[2 * q] = w[q] & 0xFFFFu; a[2 * q + 1] = w[q] >> 16; }
            if ((v.x | v.y | v.z | v.w) & 0x80008000u) {
#pragma unroll
                for (int q = 0; q < 8; ++q) if (a[q] & 0x8000u) a[q] = ow[a[q] & 0x7FFFu];
            }
            *reinterpret_cast<uint2*>(nw + k0) = uint2{a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24), a[4] | (a[5] << 8) | (a[6] << 16) | (a[7] << 24)};
        }
        __syncthreads();
    }
}

// pass 3: the tails go out final.  `entry_win` holds the window in front of every group (one group: the piece's entry window itself).
static __global__ __launch_bounds__(CHL_THREADS) void k_gz_chainl_out(const ChainItem* items, int n_items, int per_group, const uint16_t* pool,
                                                                      const uint8_t* entry_win, uint8_t* out, uint8_t* w_next) {
    const int tid = threadIdx.x, g = (int)blockIdx.x;
    const int i_lo = g * per_group, i_hi = i_lo + per_group < n_items ? i_lo + per_group : n_items;
    const uint8_t* ew = entry_win + (size_t)g * 32768;
    const int64_t B0 = i_lo < i_hi ? items[i_lo].out_base : 0;   // the group's output starts here: what lies in front of it is ew
    // byte at output position gp (inside the 32 KiB in front of a job of this group, or of the piece's end)
    auto window = [&](int64_t gp) -> uint32_t { return gp >= B0 ? (uint32_t)out[gp] : (uint32_t)ew[gp - (B0 - 32768)]; };
    int64_t end = B0;
    for (int i = i_lo; i < i_hi; ++i) {
        const ChainItem it = items[i];
        const int t = (int)(it.len < 32768 ? it.len : 32768);
        const int64_t first = it.len - t, wbase = it.out_base - 32768;
        const int nv = (t + 7) >> 3;
        for (int v = tid; v < nv; v += CHL_THREADS) {
            const int64_t p0 = first + 8 * (int64_t)v;
            if (p0 + 8 <= it.len && (p0 >> PAGE_SHIFT) == ((p0 + 7) >> PAGE_SHIFT)) {
                const V16A2 s4 = *reinterpret_cast<const V16A2*>(chl_sym(it, pool, first, p0));
                uint32_t a[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[2 * q] = s4.w[q] & 0xFFFFu; a[2 * q + 1] = s4.w[q] >> 16; }
                if ((s4.w[0] | s4.w[1] | s4.w[2] | s4.w[3]) & 0x80008000u) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (a[q] & 0x8000u) a[q] = window(wbase + (int64_t)(a[q] & 0x7FFFu));
                }
                reinterpret_cast<U64A1*>(out + it.out_base + p0)->v =
                    (u64)(a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24)) | ((u64)(a[4] | (a[5] << 8) | (a[6] << 16) | (a[7] << 24)) << 32);
            } else {
                for (int64_t p = p0; p < p0 + 8 && p < it.len; ++p) {
                    uint32_t sym = *chl_sym(it, pool, first, p);
                    if (sym & 0x8000u) sym = window(wbase + (int64_t)(sym & 0x7FFFu));
                    out[it.out_base + p] = (uint8_t)sym;
                }
            }
        }
        __syncthreads();
        end = it.out_base + it.len;
    }
    if (i_hi == n_items && w_next) {   // the last group (or the only one): the window in front of the next piece
        for (int k = tid; k < 32768; k += CHL_THREADS) w_next[k] = (uint8_t)window(end - 32768 + k);
    }
}

// ---- RESOLVE: one page of symbols -> bytes at their final place -------------------------------------------------------------------
struct ResItem { uint32_t page, n; int64_t dst, chunk_base; };   // symbols [0, n) of the page -> out[dst ..); the chunk's output starts at chunk_base
struct __attribute__((packed, aligned(1))) U64U { u64 v; };
static __global__ __launch_bounds__(BLOCK) void k_gz_resolve(const ResItem* items, const uint16_t* pool, const uint8_t* w0, uint8_t* out) {
    const ResItem it = items[blockIdx.x];
    const uint16_t* src = pool + ((size_t)it.page << PAGE_SHIFT);
    auto window = [&](uint32_t w) -> uint32_t {   // byte w of the 32 KiB in front of the chunk: final output (the chain kernel wrote it) or the piece's entry window
        const int64_t g = it.chunk_base - 32768 + (int64_t)w;
        return g >= 0 ? out[g] : w0[32768 + g];
    };
    for (uint32_t k = threadIdx.x * 8u; k < it.n; k += BLOCK * 8u) {
        if (k + 8u <= it.n) {
            const uint4 s = *reinterpret_cast<const uint4*>(src + k);
            const uint32_t wv[4] = {s.x, s.y, s.z, s.w};
            u64 v = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                uint32_t a0 = wv[h] & 0xFFFFu, a1 = wv[h] >> 16;
                if (a0 & 0x8000u) a0 = window(a0 & 0x7FFFu);
                if (a1 & 0x8000u) a1 = window(a1 & 0x7FFFu);
                v |= (u64)(a0 | (a1 << 8)) << (16 * h);
            }
            reinterpret_cast<U64U*>(out + it.dst + k)->v = v;
        } else {
            for (uint32_t q = k; q < it.n; ++q) {
                uint32_t a0 = src[q];
                if (a0 & 0x8000u) a0 = window(a0 & 0x7FFFu);
                out[it.dst + q] = (uint8_t)a0;
            }
        }
    }
}

// ---- CRC-32 of out[off, off + n): one wave per segment (inf::block_crc32) -----------------------------------------------------------
struct CrcSeg { int64_t off; int32_t n; uint32_t pad; };
// CRC_SUB waves per segment, each a sixteenth of it (one wave per MiB walked 16 KiB per lane through a byte table, lookup
// after dependent lookup: 2.1 ms for a piece's 495 segments on an otherwise empty GPU).  The remainder of a concatenation is
// linear in its parts (inf::block_crc32), so every wave XORs its part's term into crcs[k] (zeroed by the host) and is done.
constexpr int CRC_SUB = 16;
static __global__ __launch_bounds__(BLOCK) void k_gz_crc(const CrcSeg* segs, int n_segs, const uint8_t* out, uint32_t* crcs) {
    __shared__ uint32_t s_crc_tab[256], s_x2n[32];
    inf::crc_tables(s_crc_tab, s_x2n);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int g = (int)blockIdx.x * WAVES + wave, k = g / CRC_SUB, sub = g % CRC_SUB;
    if (k >= n_segs) return;
    const CrcSeg sg = segs[k];
    const int n = sg.n, part = (n + CRC_SUB - 1) / CRC_SUB;
    const int p_lo = sub * part < n ? sub * part : n, p_hi = p_lo + part < n ? p_lo + part : n;
    const int per = (p_hi - p_lo + 63) >> 6;
    const int lo = p_lo + lane * per < p_hi ? p_lo + lane * per : p_hi, hi = lo + per < p_hi ? lo + per : p_hi;
    const uint8_t* src = out + sg.off;
    uint32_t r = 0;
    int i = lo;
    // (a lane's bytes are 1 KiB from its neighbours': every load touches 64 cache lines, and with a dword per load every line
    // came from L2 sixteen times -- 8 GB of traffic for a piece's 519 MB, which was the kernel's time.  64 bytes per trip,
    // the four loads of a line back to back.)
    struct W4 { uint32_t w[4]; };
    struct __attribute__((packed, aligned(1))) W4U { W4 v; };
    for (; i + 64 <= hi; i += 64) {
        W4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = reinterpret_cast<const W4U*>(src + i + 16 * j)->v;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t w = q[j >> 2].w[j & 3];
#pragma unroll
            for (int t = 0; t < 4; ++t) { r = s_crc_tab[(r ^ w) & 0xFFu] ^ (r >> 8); w >>= 8; }
        }
    }
    for (; i + 4 <= hi; i += 4) {
        uint32_t w = reinterpret_cast<const U32U*>(src + i)->v;
#pragma unroll
        for (int t = 0; t < 4; ++t) { r = s_crc_tab[(r ^ w) & 0xFFu] ^ (r >> 8); w >>= 8; }
    }
    for (; i < hi; ++i) r = s_crc_tab[(r ^ src[i]) & 0xFFu] ^ (r >> 8);
    r = hi > lo ? inf::crc_mul(r, inf::crc_x8n((uint32_t)(n - hi), s_x2n)) : 0u;
    if (sub == 0 && lane == 0) r ^= inf::crc_mul(0xFFFFFFFFu, inf::crc_x8n((uint32_t)n, s_x2n)) ^ 0xFFFFFFFFu;   // the all-ones register, the final inversion
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) r ^= (uint32_t)__shfl_xor((int)r, d, 64);
    if (lane == 0 && r) atomicXor(&crcs[k], r);
}

} // namespace gz
} // namespace bzq

// ================================================================================ host side
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <vector>
#include <zlib.h>   // crc32_combine only: the CRCs themselves are computed on the device

/* (declared in include/blazeseq_hip.h) */
struct bzq_gzip {
    int device = 0;
    hipStream_t stream = nullptr;       // the stream the kernels and copies of a call run on
    hipStream_t own_stream = nullptr;
    std::string err;
    int32_t chunk_bytes = 16384;        // CH: one decoder wave per this many compressed bytes (zlib closes a block every ~20 KiB of FASTQ output stream)
    // device
    struct Buf { void* p = nullptr; size_t cap = 0; };
    // what a piece's decoders write (symbol pool, page links, counters, job results, member events) exists TWICE: piece k + 1 is decoded
    // (on find_stream, behind its finder) while the chain / resolve / CRC kernels of piece k still read piece k's -- round 4
    Buf comp[3], jobs[2], jobs_e[3], order_e[3], counters_e[3], outs_[2], order[2], counters2, pool_[2], page_next_[2], counters_[2], events_[2], items, crcs, win[2], chain_maps, chain_wins, chain_tabs;
    Buf h_outs_[2], h_events, h_pages, h_items, h_crcs;   // pinned host staging
    int pcur = 0;                       // which of the two sets the piece being (last) decoded uses
    bool predecode = true;              // option "predecode": 0 = a piece's decoders start behind the kernels of the piece in front
    uint32_t pool_pages = 0;
    int wcur = 0;                       // win[wcur]: the 32 KiB of output in front of the next piece
    // the NEXT piece's block finder, launched (on find_stream) the moment this piece's chain is known -- its carry is then
    // known too -- so that it runs under this piece's chain / resolve / CRC kernels and the host work between them
    struct Pre { bool valid = false; const uint8_t* src = nullptr; uint64_t n_new = 0, nc = 0; unsigned long long start_pos = 0; int jb = 0, cb = 0;
                 bool shifted = false; int nch = 0;   // shifted: jobs_e / order_e[cb] hold it, on the grid over the new bytes (job j >= 1 = chunk j - 1 of them)
                 bool decoded = false; int pb = 0; uint32_t pool_pages = 0; int64_t max_job = 0; } pre;   // decoded: its decoders have run too (set pb), with that pool and job bound
    int jlast = 1;                      // which of jobs[] / order[] the last decode used
    hipStream_t find_stream = nullptr;
    hipStream_t early_stream = nullptr;   // option early_find: a staged piece's finder, behind its copy (not ON the copy stream: beside the decoders it takes 9 ms, and the next piece's copy would wait behind it)
    hipEvent_t pre_ev = nullptr, pre_copy_ev = nullptr, pre_dec_ev = nullptr;
    // pieces on their way to the device while the one in front of them is decoded (bzq_gzip_stage): comp[b] holds one
    // STAGE_RESERVE bytes in, so that the bytes carried over from the piece in front can be put before it
    struct StagedPiece { const uint8_t* src; uint64_t n; int buf; bool found; int nch; };   // found: its finder has been launched too (early_ev[buf]), on a grid of nch - 1 chunks over its own bytes
    std::mutex stage_mu;                // (bzq_gzip_stage may come from a second thread)
    std::deque<StagedPiece> staged;     // oldest first, at most two
    bool comp_busy[3] = {false, false, false}; // holds a staged piece, or the piece being decoded (three: the piece whose chain / resolve / CRC run, the one being
                                        // decoded under them, and the one behind it on its way to the device)
    hipStream_t copy_stream = nullptr;
    hipEvent_t staged_ev[3] = {nullptr, nullptr, nullptr}, early_ev[3] = {nullptr, nullptr, nullptr};
    // option "early_find" (default 0; the ingest sets it): a staged piece's finder runs behind its COPY (on a stream of its own) instead of
    // behind the decoding of the piece in front.  Alone it was no gain in round 4 (27.3 against 28.2 GB/s file -> records: with the finder
    // off the critical path the chain / resolve / CRC kernels of the piece in front were on it -- beside 16 000 decoder waves they took
    // 12 ms instead of 3); with chain_l2 and defer_verify (round 5) the three are the file pipeline's steady state, 15.8 -> 14.0 ms a piece
    bool early_find = false;
    // option "chain_l2": the chain kernels in the form that runs BESIDE the next piece's decoders (k_gz_chainl_*: 256 threads, no LDS, <= 32 VGPRs, windows through the L2)
    bool chain_l2 = false;
    // option "defer_verify" (the ingest sets it; default 0): a call that launched the NEXT piece's decoders returns without waiting for
    // its own chain / resolve / CRC kernels -- its output is complete in STREAM ORDER (on `stream`), not at return -- and the member
    // checks of the piece (CRC-32, ISIZE) are made at the start of the next call, which fails if they fail.  Only for a caller whose
    // consumers are on `stream` (the ingest's FIFO copies are): the piece's pinned buffer then goes back to the reader a chain's
    // length earlier, and the piece after next is on the device -- and its finder through -- when the next chain has been walked.
    bool defer_verify = false;
    struct PendingVerify { bool on = false; std::vector<int32_t> seg_n; std::vector<uint8_t> closes; std::vector<uint32_t> crc_t, isize_t; } pend;
    hipEvent_t ver_ev = nullptr;
    uint64_t deferred_calls = 0;        // query "deferred_calls"
    // the stream
    std::vector<uint8_t> carry;         // compressed bytes not consumed yet
    unsigned long long start_pos = 1;   // where decoding resumes inside `carry`: pos_header(0), or pos_deflate(bit 0..7)
    std::atomic<bool> finished{false};   // (read by bzq_gzip_stage, which may run on a second thread)
    uint64_t members_done = 0;
    uint32_t crc_run = 0;               // CRC-32 and length of the open member's output so far
    uint64_t len_run = 0;
    // a stretch without block starts the finder can find (ST_FAR) goes to the host (gz_decode_host): zlib, one core, ~30 x one wave
    bool host_cont = true;              // option "host_continuation"
    int64_t far_bytes = 256 << 10;      // option "far_kib": how far a decoder may go without meeting a found start
    bool host_next = false;             // the next call continues on the host
    uint64_t host_budget = 32ull << 20, host_budget_min = 32ull << 20;   // ... for about this much output; doubles while the device keeps handing over, back to the minimum (option "host_budget_kib") when a piece went through
    uint64_t host_calls = 0, host_bytes_out = 0;
    Buf h_host_out;                     // pinned staging of the host's output
    bzq_gzip_stats stats{};
};

namespace bzq {
namespace gz {

constexpr int MAX_FALLBACK = 64;   // explicit restarts per piece before the call gives up
constexpr int32_t CRC_SEG = 1 << 20;

inline int gz_fail(bzq_gzip* h, int code, const std::string& msg) { h->err = msg; return code; }

// CRC-32 of a concatenation: crc(A B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] mod the CRC polynomial (reflected bit order).
// zlib 1.2.11's crc32_combine squares a 32x32 matrix ~20 times per call (~10 us): with a segment per MiB that was 3..6 ms of
// host time per piece.  Here the power of x is computed once per distinct length (all full segments share one) and applied
// with 32 shift-and-xor steps.
inline uint32_t crc_mul(uint32_t a, uint32_t b) {   // a * b mod P
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
inline uint32_t crc_xpow8(uint64_t n) {   // x^(8 n) mod P
    static const std::array<uint32_t, 64> sq = [] { std::array<uint32_t, 64> t{}; t[0] = 1u << 30; for (int i = 1; i < 64; ++i) t[(size_t)i] = crc_mul(t[(size_t)i - 1], t[(size_t)i - 1]); return t; }();   // x^(2^i)
    uint32_t p = 1u << 31;
    for (int k = 3; n; n >>= 1, ++k) if (n & 1u) p = crc_mul(sq[(size_t)(k & 63)], p);
    return p;
}
#define GZCHK(h, call)                                                                                          \
    do {                                                                                                        \
        const hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return bzq::gz::gz_fail((h), BZQ_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Pages of the symbol pool a piece of n compressed bytes in n_chunks decoder jobs can need.  Every job rounds its output up to whole
// pages, so pages <= output symbols / PAGE + jobs (+ restarts); the unknown is the output: n x the stream's ratio so far (and a quarter
// on top), or 3 x for a stream's first piece (FASTQ under gzip: 2 - 4).  Rounds 3-5 reserved 5 x AND two pages per job: 3.3 GiB per
// pool for a 128 MiB piece that uses 1.1 -- twice (two pools), 30-80 ms of hipMalloc per GiB in a fresh process.  A piece that needs more
// says so (counters[0]: the decoders count what they ask for) and is decoded again with that (gz_decode's attempt loop).
inline uint32_t pool_want(const bzq_gzip* h, uint64_t n, int n_chunks) {
    double ratio = 3.0;
    if (h->stats.bytes_consumed >= (4ull << 20)) ratio = std::max(1.5, 1.25 * (double)h->stats.bytes_out / (double)h->stats.bytes_consumed);
    const double pages = std::min((double)n * ratio / (double)PAGE, (double)(1u << 24));
    return (uint32_t)pages + (uint32_t)n_chunks + (uint32_t)MAX_FALLBACK + 64u;
}

// every stream this decoder enqueues work on: the call's stream (the caller's, or its own) and its side streams
inline hipError_t gz_quiesce(bzq_gzip* h) {
    hipError_t e = hipSuccess, r;
    for (hipStream_t s : {h->stream, h->own_stream, h->copy_stream, h->find_stream, h->early_stream})
        if (s && (r = hipStreamSynchronize(s)) != hipSuccess) e = r;
    return e;
}

inline int gz_ensure(bzq_gzip* h, bzq_gzip::Buf& b, size_t bytes, bool pinned = false, int slack_shift = 2) {
    if (bytes <= b.cap) return 0;
    bzq::cache::Pool& pool = pinned ? bzq::cache::pinned_pool() : bzq::cache::device_pool();   // (bzq_bufcache.hpp: the buffers of a decoder outlive it)
    // (what goes back to the cache skips hipFree's wait: whoever may still touch the old buffer is waited for here -- the decoder's own
    // streams, not the device: a caller's kernels on other streams are not this decoder's business)
    if (b.p) { GZCHK(h, gz_quiesce(h)); pool.put(b.p); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + (bytes >> slack_shift) + 256;   // (room to grow without a new allocation: a quarter; the symbol pool, gigabytes, a sixteenth)
    const hipError_t e = pool.get(h->device, want, &b.p);
    if (e != hipSuccess) { b.p = nullptr; (void)hipGetLastError(); return gz_fail(h, BZQ_ERR_NOMEM, "bzq_gzip: cannot allocate " + std::to_string(want) + " bytes"); }
    b.cap = want;
    if (pinned) pool.pin(b.p, 0, want);   // (small host buffers the device writes results into: registered whole, at once)
    return 0;
}

inline void gz_free(bzq_gzip* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    // (helper streams go back to the process-wide pool instead of hipStreamDestroy: bzq_bufcache.hpp)
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); bzq::cache::stream_pool().put(h->device, h->copy_stream); }
    if (h->find_stream) { (void)hipStreamSynchronize(h->find_stream); bzq::cache::stream_pool().put(h->device, h->find_stream); }
    if (h->early_stream) { (void)hipStreamSynchronize(h->early_stream); bzq::cache::stream_pool().put(h->device, h->early_stream); }
    for (hipEvent_t e : {h->pre_ev, h->pre_copy_ev, h->pre_dec_ev, h->ver_ev}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->staged_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->early_ev) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamSynchronize(h->stream);   // (a caller's stream may still run the last decode: what goes back to the cache skips hipFree's wait)
    for (bzq_gzip::Buf* b : {&h->comp[0], &h->comp[1], &h->comp[2], &h->jobs_e[0], &h->jobs_e[1], &h->jobs_e[2], &h->order_e[0], &h->order_e[1], &h->order_e[2], &h->counters_e[0], &h->counters_e[1], &h->counters_e[2], &h->order[0], &h->order[1], &h->jobs[0], &h->jobs[1], &h->counters2, &h->outs_[0], &h->outs_[1], &h->pool_[0], &h->pool_[1], &h->page_next_[0], &h->page_next_[1], &h->counters_[0], &h->counters_[1], &h->events_[0], &h->events_[1], &h->items, &h->crcs, &h->win[0], &h->win[1], &h->chain_maps, &h->chain_wins, &h->chain_tabs})
        bzq::cache::device_pool().put(b->p);
    for (bzq_gzip::Buf* b : {&h->h_outs_[0], &h->h_outs_[1], &h->h_events, &h->h_pages, &h->h_items, &h->h_crcs, &h->h_host_out})
        bzq::cache::pinned_pool().put(b->p);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

inline int gz_open(int device, bzq_gzip** out, std::string& err) {
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) { err = "bzq_gzip_open: hipSetDevice failed"; return BZQ_ERR_HIP; }
    bzq_gzip* h = new bzq_gzip();
    if (const char* e = getenv("BZQ_GZ_CHAIN_L2")) h->chain_l2 = atoi(e) != 0;   // (A/B runs through the ingest, which has no handle to set options on)
    if (const char* e = getenv("BZQ_GZ_EARLY_FIND")) h->early_find = atoi(e) != 0;
    h->device = device;
    // the stream of a piece's chain / resolve / CRC kernels: highest priority -- they run beside the NEXT piece's decoders (find_stream),
    // which fill every CU, and the caller waits for them (BZQ_GZ_PRIORITY=0: default class, for measurements)
    int plo = 0, phi = 0;
    const char* pe = getenv("BZQ_GZ_PRIORITY");
    const bool prio = (!pe || pe[0] != '0') && hipDeviceGetStreamPriorityRange(&plo, &phi) == hipSuccess && phi < plo;
    if ((prio ? hipStreamCreateWithPriority(&h->own_stream, hipStreamNonBlocking, phi) : hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) { err = "bzq_gzip_open: hipStreamCreate failed"; delete h; return BZQ_ERR_HIP; }
    h->stream = h->own_stream;
    if (bzq::cache::stream_pool().get(device, &h->find_stream) != hipSuccess || hipEventCreateWithFlags(&h->pre_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->pre_copy_ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ver_ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->pre_dec_ev, hipEventDisableTiming) != hipSuccess || bzq::cache::stream_pool().get(device, &h->copy_stream) != hipSuccess || hipEventCreateWithFlags(&h->staged_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->staged_ev[1], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->staged_ev[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->early_ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->early_ev[1], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->early_ev[2], hipEventDisableTiming) != hipSuccess) {
        err = "bzq_gzip_open: hipStreamCreate failed"; gz_free(h); return BZQ_ERR_HIP;
    }
    int rc;
    if ((rc = gz_ensure(h, h->win[0], 32768)) || (rc = gz_ensure(h, h->win[1], 32768)) || (rc = gz_ensure(h, h->counters_e[0], 128)) || (rc = gz_ensure(h, h->counters_e[1], 128)) || (rc = gz_ensure(h, h->counters_e[2], 128)) || (rc = gz_ensure(h, h->counters_[0], 128)) || (rc = gz_ensure(h, h->counters_[1], 128)) || (rc = gz_ensure(h, h->counters2, 128))) { err = h->err; gz_free(h); return rc; }
    if (hipMemsetAsync(h->counters2.p, 0, 128, h->stream) != hipSuccess || hipMemsetAsync(h->win[0].p, 0, 32768, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { err = "bzq_gzip_open: hipMemset failed"; gz_free(h); return BZQ_ERR_HIP; }
    h->start_pos = pos_header(0);
    for (uint64_t len : {1ull, 4097ull, (unsigned long long)CRC_SEG})   // (the combine above against zlib's, once)
        if ((crc_mul(crc_xpow8(len), 0x12345678u) ^ 0x9ABCDEF0u) != (uint32_t)crc32_combine(0x12345678u, 0x9ABCDEF0u, (z_off_t)len)) { err = "bzq_gzip_open: internal: CRC combine self-check failed"; gz_free(h); return BZQ_ERR_HIP; }
    *out = h;
    return 0;
}

// A piece that a LATER gz_decode will be given starts its way to the device now (pinned host memory, or the call is
// pointless): the gz_decode that gets the same (src, n) finds it there instead of copying -- 5 ms of a 256 MiB piece's ~35,
// hidden behind the decoding of the piece in front.  Three pieces can be outstanding (three buffers: one may be the piece being
// decoded); with all taken the call does nothing, and that piece is copied by its gz_decode as if never staged.  Pieces are
// taken in the order staged.  May be called from another thread than gz_decode's while that runs (the ingest's read-ahead
// thread does), not concurrently with itself; src must stay untouched until its gz_decode has returned.
// ---- CRC-32 and ISIZE of every member that ended in the last piece (RFC 1952 2.3.1): the device's per-segment remainders (h_crcs, behind
// ver_ev when the call that launched them did not wait) combined in stream order and compared with the trailers noted in h->pend
inline int gz_verify_pending(bzq_gzip* h) {
    bzq_gzip::PendingVerify& p = h->pend;
    if (p.seg_n.empty() && !p.on) return 0;
    if (p.on) { p.on = false; GZCHK(h, hipEventSynchronize(h->ver_ev)); }
    const uint32_t* crcs = (const uint32_t*)h->h_crcs.p;
    const uint32_t xp_full = crc_xpow8((uint64_t)CRC_SEG);
    int rc = 0;
    for (size_t k = 0; k < p.seg_n.size() && !rc; ++k) {
        if (p.seg_n[k]) {
            h->crc_run = h->len_run ? crc_mul(p.seg_n[k] == CRC_SEG ? xp_full : crc_xpow8((uint64_t)p.seg_n[k]), h->crc_run) ^ crcs[k] : crcs[k];
            h->len_run += (uint64_t)p.seg_n[k];
        }
        if (p.closes[k]) {
            if ((h->len_run ? h->crc_run : 0u) != p.crc_t[k] || (uint32_t)h->len_run != p.isize_t[k])
                rc = gz_fail(h, BZQ_ERR_IO, "bzq_gzip: member " + std::to_string(h->members_done) + " fails its " + ((uint32_t)h->len_run != p.isize_t[k] ? "length" : "CRC-32") +
                                                " check (corrupt file)");
            else { h->members_done += 1; h->crc_run = 0; h->len_run = 0; }
        }
    }
    p.seg_n.clear(); p.closes.clear(); p.crc_t.clear(); p.isize_t.clear();
    h->stats.members = h->members_done;
    return rc;
}

constexpr uint64_t STAGE_RESERVE = 4ull << 20;
inline int gz_stage(bzq_gzip* h, const uint8_t* src, uint64_t n_new) {
    // (no h->err in here: that string is the decode thread's.  A piece that could not be staged is copied by its gz_decode, which
    // then meets whatever was wrong on its own thread; the return value only says that nothing was staged.)
    if (h->finished || !n_new) return 0;
    if (hipSetDevice(h->device) != hipSuccess) return BZQ_ERR_HIP;
    int bi;
    {
        std::lock_guard<std::mutex> lk(h->stage_mu);
        bi = !h->comp_busy[0] ? 0 : !h->comp_busy[1] ? 1 : !h->comp_busy[2] ? 2 : -1;
        if (bi < 0) return 0;
        h->comp_busy[bi] = true;
    }
    auto give_back = [&](int code) { (void)hipGetLastError(); std::lock_guard<std::mutex> lk(h->stage_mu); h->comp_busy[bi] = false; return code; };
    bzq_gzip::Buf& b = h->comp[bi];
    if (b.cap < STAGE_RESERVE + n_new + 64) {   // (not gz_ensure: that waits for the decode stream, which the other thread may be feeding)
        if (b.p) { const hipError_t e = hipDeviceSynchronize(); bzq::cache::device_pool().put(b.p); b.p = nullptr; b.cap = 0; if (e != hipSuccess) return give_back(BZQ_ERR_HIP); }
        const size_t want = (size_t)(STAGE_RESERVE + n_new + n_new / 4 + 256);
        if (bzq::cache::device_pool().get(h->device, want, &b.p) != hipSuccess) { b.p = nullptr; return give_back(BZQ_ERR_NOMEM); }
        b.cap = want;
    }
    uint8_t* d = (uint8_t*)b.p + STAGE_RESERVE;
    hipError_t e = h->early_stream ? hipStreamWaitEvent(h->copy_stream, h->early_ev[bi], 0) : hipSuccess;   // (a finder of this buffer's last piece that nobody waited for: a piece that was dropped)
    if (e == hipSuccess) e = bzq::cache::pinned_pool().h2d(d, src, n_new, h->copy_stream);   // (the ingest's slots are pinned lazily, block by block: bzq_bufcache.hpp)
    if (e == hipSuccess) e = hipMemsetAsync(d + n_new, 0, 64, h->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(h->staged_ev[bi], h->copy_stream);
    if (e != hipSuccess) { (void)hipStreamSynchronize(h->copy_stream); return give_back(BZQ_ERR_HIP); }
    // The piece's FINDER behind its copy, on the same stream (round 4): it needs the piece's bytes and nothing else -- not where the
    // piece in front will end -- if its chunk grid is laid over the piece's own bytes: job j >= 1 = chunk j - 1 of them.  Positions
    // come out relative to CH bytes in front of the piece (the kernel's chunk 0 is job 0's, never looked at); once the carry is
    // known they are shifted to where the piece then starts (k_gz_shift, gz_decode).  3 of a piece's 15.5 ms off its critical path.
    bool found = false;
    int nch = 0;
    static const bool no_early = getenv("BZQ_GZ_NO_EARLY_FIND") != nullptr;
    if (h->early_find && !no_early) {
        const int CH = h->chunk_bytes;
        nch = (int)((n_new + (uint64_t)CH - 1) / (uint64_t)CH) + 1;
        auto fit = [&](bzq_gzip::Buf& q, size_t bytes) {   // (the buffers of comp[bi]'s last piece: nothing in flight touches them -- its call has returned)
            if (bytes <= q.cap) return true;
            if (q.p) bzq::cache::device_pool().put(q.p);
            q.p = nullptr; q.cap = 0;
            const size_t want = bytes + bytes / 4 + 256;
            if (bzq::cache::device_pool().get(h->device, want, &q.p) != hipSuccess) { q.p = nullptr; return false; }
            q.cap = want;
            return true;
        };
        if (fit(h->jobs_e[bi], (size_t)(nch + MAX_FALLBACK) * sizeof(Job)) && fit(h->order_e[bi], (size_t)2 * ORDER_BINS * 4 + (size_t)nch * 5 + 64)) {
            Args a2{};
            a2.comp = d - CH; a2.n = (int64_t)n_new + CH; a2.jobs = (Job*)h->jobs_e[bi].p; a2.n_jobs = nch; a2.n_cand = nch; a2.chunk_bytes = CH; a2.counters = (uint32_t*)h->counters_e[bi].p;
            if (!h->early_stream) e = bzq::cache::stream_pool().get(h->device, &h->early_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(h->early_stream, h->staged_ev[bi], 0);
            if (e == hipSuccess) e = hipMemsetAsync(h->counters_e[bi].p, 0, 128, h->early_stream);
            if (e == hipSuccess) { hipLaunchKernelGGL(k_gz_find, dim3((unsigned)((nch + WAVES - 1) / WAVES)), dim3(BLOCK), 0, h->early_stream, a2); e = hipGetLastError(); }
            if (e == hipSuccess) e = hipEventRecord(h->early_ev[bi], h->early_stream);
            found = e == hipSuccess;
            if (!found) (void)hipGetLastError();
        }
    }
    std::lock_guard<std::mutex> lk(h->stage_mu);
    h->staged.push_back(bzq_gzip::StagedPiece{src, n_new, bi, found, nch});
    return 0;
}

// ---- gzip member header on the host (the rules of parse_member_header above): > 0 = index of the DEFLATE data, 0 = the input ends inside it, -1 = none
inline int64_t host_member_header(const uint8_t* in, uint64_t n, uint64_t B) {
    if (B + 2 <= n && (in[B] != 0x1f || in[B + 1] != 0x8b)) return -1;
    if (B + 10 > n) return 0;
    if (in[B + 2] != 8) return -1;
    const uint32_t flg = in[B + 3];
    if (flg & 0xE0u) return -1;
    uint64_t p = B + 10;
    if (flg & 4u) { if (p + 2 > n) return 0; p += 2 + (uint64_t)(in[p] | (in[p + 1] << 8)); }
    for (int f = 8; f <= 16; f <<= 1) {
        if (!(flg & (uint32_t)f)) continue;
        for (;;) { if (p >= n) return 0; if (in[p++] == 0) break; }
    }
    if (flg & 2u) p += 2;
    return p < n ? (int64_t)p : 0;
}

// The continuation of the stream ON THE HOST (zlib, raw inflate block by block) from where the device stopped with ST_FAR: a stretch
// in which the finder finds no block start -- fixed-Huffman or stored blocks only, or one block of hundreds of KiB -- is decoded by
// ONE wave on the device, at ~10 MB/s; one host core does 300+.  Same contract as gz_decode (whole blocks only, members verified
// by CRC-32 and ISIZE, the rest stays in the handle); about `budget` bytes of output, then the device is asked again.
// The reference's GZFile is this loop for the whole file (io/readers.mojo:283-377: gzread).
inline int gz_decode_host(bzq_gzip* h, const uint8_t* src, uint64_t n_new, bool is_last, uint8_t* d_out, uint64_t out_cap, uint64_t budget, uint64_t* out_bytes, int32_t* more) {
    const hipStream_t s = h->stream;
    if (h->pre.valid) { GZCHK(h, hipStreamSynchronize(h->find_stream)); h->pre.valid = false; h->pre.decoded = false; }   // (finder and decoders launched for the device's next piece: dropped)
    if (n_new) {   // a piece staged for the device is this call's: its buffer is free again
        int drop = -1;
        { std::lock_guard<std::mutex> lk(h->stage_mu); if (!h->staged.empty() && h->staged.front().src == src && h->staged.front().n == n_new) { drop = h->staged.front().buf; h->staged.pop_front(); } }
        if (drop >= 0) { GZCHK(h, hipEventSynchronize(h->staged_ev[drop])); std::lock_guard<std::mutex> lk(h->stage_mu); h->comp_busy[drop] = false; }
    }
    const uint64_t nc = h->carry.size(), n = nc + n_new;
    std::vector<uint8_t> joined;
    const uint8_t* in = h->carry.data();
    if (nc == 0) in = src;
    else if (n_new) { joined.resize((size_t)n); memcpy(joined.data(), h->carry.data(), (size_t)nc); memcpy(joined.data() + nc, src, (size_t)n_new); in = joined.data(); }
    // the 32 KiB in front: the dictionary of a stream entered in the middle of a member
    std::vector<uint8_t> win(32768);
    GZCHK(h, hipMemcpyAsync(win.data(), h->win[h->wcur].p, 32768, hipMemcpyDeviceToHost, s));
    GZCHK(h, hipStreamSynchronize(s));
    const uint64_t lim = std::min<uint64_t>(out_cap, std::max<uint64_t>(budget, 1ull << 20) + (4ull << 20));   // (a block is accepted whole: room beyond the budget for the one that crosses it)
    int rc;
    if ((rc = gz_ensure(h, h->h_host_out, (size_t)lim + 64, true))) return rc;
    uint8_t* hb = (uint8_t*)h->h_host_out.p;

    bool at_header = (h->start_pos & 1ull) != 0;
    uint64_t ip = 0; int bit = at_header ? 0 : (int)((h->start_pos >> 1) & 7ull);   // next input position
    uint64_t op = 0;                                                                // output accepted so far
    uint32_t crc = h->crc_run; uint64_t len = h->len_run;
    uint64_t members = h->members_done;
    int32_t status = ST_NEED_MORE;
    bool stop = false, first_of_call = true;
    while (!stop) {
        if (at_header) {
            if (ip >= n) { status = ST_END_INPUT; break; }
            const int64_t d = host_member_header(in, n, ip);
            if (d < 0) { status = ST_BAD_HEADER; break; }
            if (d == 0) { status = ST_NEED_MORE; break; }
            ip = (uint64_t)d; bit = 0; at_header = false; crc = 0; len = 0;
        }
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return gz_fail(h, BZQ_ERR_NOMEM, "bzq_gzip: inflateInit2 failed");
        struct End { z_stream* z; ~End() { inflateEnd(z); } } end_guard{&zs};
        if (first_of_call && len > 0) {   // entered inside a member: what it may refer back to
            const uint32_t k = (uint32_t)std::min<uint64_t>(len, 32768);
            if (inflateSetDictionary(&zs, win.data() + 32768 - k, k) != Z_OK) return gz_fail(h, BZQ_ERR_HIP, "bzq_gzip: internal: inflateSetDictionary refused the window");
        }
        first_of_call = false;
        uint64_t fed = ip;   // next input byte not yet given to zlib
        if (bit) { if (inflatePrime(&zs, 8 - bit, in[ip] >> bit) != Z_OK) return gz_fail(h, BZQ_ERR_HIP, "bzq_gzip: internal: inflatePrime"); fed = ip + 1; }
        zs.next_in = const_cast<Bytef*>(in + fed); zs.avail_in = 0;
        zs.next_out = hb + op; zs.avail_out = 0;
        for (;;) {   // block by block
            if (zs.avail_in == 0) { const uint64_t m = std::min<uint64_t>(n - fed, 1ull << 30); zs.next_in = const_cast<Bytef*>(in + fed); zs.avail_in = (uInt)m; fed += m; }
            if (zs.avail_out == 0) { const uint64_t produced = (uint64_t)(zs.next_out - hb); zs.avail_out = (uInt)std::min<uint64_t>(lim - produced, 1ull << 30); }
            const int zr = inflate(&zs, Z_BLOCK);
            if (zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR)
                return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: invalid DEFLATE data near byte " + std::to_string(h->stats.bytes_consumed + (uint64_t)(zs.next_in - in)) + " of the compressed stream");
            const uint64_t in_pos = (uint64_t)(zs.next_in - in), produced = (uint64_t)(zs.next_out - hb);
            if (zr == Z_STREAM_END || (zs.data_type & 128)) {   // a block boundary (bit 7: inflate returned right behind an end-of-block code)
                if (zr == Z_STREAM_END || (zs.data_type & 64)) {   // ... of the member's LAST block: CRC-32 and ISIZE behind the next byte edge (RFC 1952 2.3.1)
                    const uint64_t T = in_pos;   // (the unused bits of the last byte taken are padding)
                    if (T + 8 > n) { status = ST_NEED_MORE; stop = true; break; }   // (not accepted: the block is decoded again with the trailer in sight, as on the device)
                    const uint32_t c2 = (uint32_t)crc32(crc, hb + op, (uInt)(produced - op));
                    const uint64_t l2 = len + (produced - op);
                    uint32_t crc_t = 0, isize_t = 0;
                    for (int q = 0; q < 4; ++q) { crc_t |= (uint32_t)in[T + q] << (8 * q); isize_t |= (uint32_t)in[T + 4 + q] << (8 * q); }
                    if ((l2 ? c2 : 0u) != crc_t || (uint32_t)l2 != isize_t)
                        return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: member " + std::to_string(members) + " fails its " + ((uint32_t)l2 != isize_t ? "length" : "CRC-32") + " check (corrupt file)");
                    members += 1; crc = 0; len = 0; op = produced;
                    ip = T + 8; bit = 0; at_header = true;
                    if (op >= budget) { status = ST_SPLIT; stop = true; }
                    break;   // (the next member, if any, is a new zlib stream)
                }
                crc = (uint32_t)crc32(crc, hb + op, (uInt)(produced - op));
                len += produced - op; op = produced;
                const uint64_t bits = in_pos * 8 - (uint64_t)(zs.data_type & 7);
                ip = bits >> 3; bit = (int)(bits & 7);
                if (op >= budget) { status = ST_SPLIT; stop = true; break; }
                continue;
            }
            if (produced >= lim) {   // the room is used up inside a block
                if (op == 0 && lim >= out_cap)
                    return gz_fail(h, BZQ_ERR_NOMEM, "bzq_gzip: out_capacity (" + std::to_string(out_cap) + ") is below the output of one DEFLATE block near byte " +
                                                         std::to_string(h->stats.bytes_consumed + ip) + " of the compressed stream");
                status = ST_SPLIT; stop = true; break;
            }
            if (zs.avail_in == 0 && fed >= n) { status = ST_NEED_MORE; stop = true; break; }   // the input ends inside the block
        }
    }
    // ---- the output, the window behind it, what stays
    if (op) {
        // hb is a buffer of the pinned pool: anonymous memory registered in 32 MiB blocks, and one copy must not span two registrations
        // (bzq_bufcache.hpp) -- the host continuation can emit more than a block in one call (its budget is 32 MiB + 4, and doubles)
        GZCHK(h, bzq::cache::pinned_pool().h2d(d_out, hb, op, s));
        std::vector<uint8_t> w2(32768);
        if (op >= 32768) memcpy(w2.data(), hb + op - 32768, 32768);
        else { memcpy(w2.data(), win.data() + op, (size_t)(32768 - op)); memcpy(w2.data() + (32768 - op), hb, (size_t)op); }
        GZCHK(h, hipMemcpyAsync(h->win[h->wcur ^ 1].p, w2.data(), 32768, hipMemcpyHostToDevice, s));
        GZCHK(h, hipStreamSynchronize(s));
        h->wcur ^= 1;
    }
    h->crc_run = crc; h->len_run = len; h->members_done = members;
    const bool progressed = ip > 0 || op > 0;
    {
        std::vector<uint8_t> nk(in + ip, in + n);
        h->carry.swap(nk);
        h->start_pos = at_header ? pos_header(0) : pos_deflate((u64)bit);
    }
    h->stats.pieces += 1; h->stats.bytes_in += n_new; h->stats.bytes_consumed += ip; h->stats.bytes_out += op; h->stats.members = h->members_done;
    h->stats.fallback_jobs += 1; h->host_calls += 1; h->host_bytes_out += op;
    *out_bytes = op;
    if (status == ST_SPLIT) { *more = 1; return 0; }
    const bool cut_member = at_header && is_last && status == ST_NEED_MORE && h->carry.size() >= 2 && h->carry[0] == 0x1f && h->carry[1] == 0x8b;
    const bool garbage = at_header && !cut_member && (status == ST_BAD_HEADER || (is_last && status == ST_NEED_MORE));
    if (status == ST_BAD_HEADER && h->members_done == 0) return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: not a gzip stream (no member header at its start)");
    if (garbage && h->members_done > 0) { h->finished = true; h->carry.clear(); return 0; }
    if (is_last) {
        if (status == ST_END_INPUT) { h->finished = true; return 0; }
        return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: unexpected end of the gzip stream (truncated file)");
    }
    if (!progressed && h->carry.size() > (1ull << 31)) return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: 2 GiB of compressed bytes hold no whole DEFLATE block");
    return 0;
}

// The next piece of the compressed stream (host memory; pinned memory makes the copy a DMA) -> its bytes at d_out (device).
// As many whole DEFLATE blocks as the piece holds and out_capacity takes are decoded; what is left of the piece is kept
// inside the handle and decoded in front of the next piece.  *more = 1: the output was cut by out_capacity -- call again (n = 0
// is fine) to get the rest.  Synchronous: on return the bytes are in d_out.
inline int gz_decode(bzq_gzip* h, const uint8_t* src, uint64_t n_new, bool is_last, uint8_t* d_out, uint64_t out_cap, uint64_t* out_bytes, int32_t* more) {
    *out_bytes = 0; *more = 0;
    if (h->finished) return 0;   // (whatever follows the last member is ignored, like gzread does)
    GZCHK(h, hipSetDevice(h->device));
    { const int vrc = gz_verify_pending(h); if (vrc) return vrc; }   // (option defer_verify: the piece in front is judged now)
    const hipStream_t s = h->stream;
    const uint64_t nc = h->carry.size(), n = nc + n_new;
    auto byte_at = [&](uint64_t i) -> uint32_t { return i < nc ? h->carry[(size_t)i] : src[i - nc]; };
    if (n == 0) {
        if (is_last) {
            if (h->members_done == 0) return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: empty input is not a gzip stream");
            h->finished = true;
        }
        return 0;
    }
    if (n > (1ull << 33)) return gz_fail(h, BZQ_ERR_ARG, "bzq_gzip: more than 8 GiB of compressed bytes in one piece (a DEFLATE block that never ends?)");
    if (h->host_next) {   // the device stopped at a stretch it would decode with one wave (ST_FAR): zlib on the host for a while, then the device again
        h->host_next = false;
        return gz_decode_host(h, src, n_new, is_last, d_out, out_cap, h->host_budget, out_bytes, more);
    }
    int rc;
    static const bool timing = getenv("BZQ_GZ_TIMING") != nullptr;   // debug: phase times of every call on stderr
    auto t_prev = std::chrono::steady_clock::now();
    double t_ph[8] = {0};
    auto lap = [&](int k) { if (timing) { const auto t = std::chrono::steady_clock::now(); t_ph[k] += std::chrono::duration<double, std::milli>(t - t_prev).count(); t_prev = t; } };
    const int CH = h->chunk_bytes;
    // the buffer of the compressed bytes: the one this piece was staged into, or a free one (a staged piece that is not
    // this one stays where it is, unless both buffers hold such: then the younger is dropped)
    bool use_staged = false;
    int cb = -1;
    {
        std::unique_lock<std::mutex> lk(h->stage_mu);
        if (n_new && !h->staged.empty() && h->staged.front().src == src && h->staged.front().n == n_new) {
            cb = h->staged.front().buf;
            h->staged.pop_front();
            use_staged = nc <= STAGE_RESERVE;   // (else the copy is wasted: the carry does not fit in front of it)
        } else {
            cb = !h->comp_busy[0] ? 0 : !h->comp_busy[1] ? 1 : !h->comp_busy[2] ? 2 : -1;
            if (cb < 0) { cb = h->staged.back().buf; h->staged.pop_back(); }
            else h->comp_busy[cb] = true;
        }
    }
    struct Release { bzq_gzip* h; int b; ~Release() { std::lock_guard<std::mutex> lk(h->stage_mu); h->comp_busy[b] = false; } } release{h, cb};
    if (!use_staged) GZCHK(h, hipEventSynchronize(h->staged_ev[cb]));   // (whatever was last staged into it has arrived)
    // was this piece's finder launched under the piece in front (same bytes, same carry, same start)?
    bool prefound = false;
    if (h->pre.valid) {
        prefound = use_staged && h->pre.src == src && h->pre.n_new == n_new && h->pre.nc == nc && h->pre.start_pos == h->start_pos && h->pre.cb == cb;
        if (!prefound) GZCHK(h, hipStreamSynchronize(h->find_stream));   // (not this piece after all: its buffers are free again once it is done)
        h->pre.valid = false;
    }
    // the chunk grid: uniform over the piece (job c = chunk c), or -- found at staging time -- over the piece's new bytes (job j >= 1 = chunk j - 1 of them)
    const bool shifted = prefound && h->pre.shifted;
    const int n_chunks = shifted ? h->pre.nch : (int)((n + (uint64_t)CH - 1) / (uint64_t)CH);
    const int n_jobs_cap = n_chunks + MAX_FALLBACK;
    const uint32_t max_events = (uint32_t)std::min<uint64_t>(n / 18 + (uint64_t)n_chunks + 64, 1u << 28);
    const int jb = prefound ? h->pre.jb : (h->jlast ^ 1);
    if (!shifted) h->jlast = jb;
    bzq_gzip::Buf &jobsB = shifted ? h->jobs_e[cb] : h->jobs[jb], &orderB = shifted ? h->order_e[cb] : h->order[jb];
    // one job's output is bounded (it must fit the caller's buffer whole): a quarter of the buffer, 64 KiB .. 16 MiB
    const int64_t max_job = (int64_t)std::min<uint64_t>(16ull << 20, std::max<uint64_t>(64ull << 10, out_cap / 4));
    // ... and have its decoders run as well, into the other set of pool / results (with a job bound this call's buffer takes)?
    bool predecoded = prefound && h->pre.decoded && h->pre.max_job <= max_job;
    if (h->pre.decoded && !predecoded && prefound) GZCHK(h, hipStreamSynchronize(h->find_stream));   // (the decoders are not used: their set is free once they are done)
    h->pre.decoded = false;
    const int pb = predecoded ? h->pre.pb : h->pcur;
    h->pcur = pb;
    bzq_gzip::Buf &pool = h->pool_[pb], &d_page_next = h->page_next_[pb], &counters = h->counters_[pb], &d_outs = h->outs_[pb], &events = h->events_[pb], &h_outs = h->h_outs_[pb];
    if ((rc = gz_ensure(h, orderB, (size_t)2 * ORDER_BINS * 4 + (size_t)n_chunks * 5 + 64)) || (!use_staged && (rc = gz_ensure(h, h->comp[cb], n + 64))) || (rc = gz_ensure(h, jobsB, (size_t)n_jobs_cap * sizeof(Job))) ||
        (rc = gz_ensure(h, d_outs, (size_t)n_jobs_cap * sizeof(JobOut))) || (rc = gz_ensure(h, events, (size_t)max_events * sizeof(Event))) ||
        (rc = gz_ensure(h, h_outs, (size_t)n_jobs_cap * sizeof(JobOut) + 128, true)))
        return rc;
    uint8_t* d_comp = (uint8_t*)h->comp[cb].p + (use_staged ? STAGE_RESERVE - nc : 0);
    // (the carry is pageable memory; the vector is not touched before the stream has been waited for, further down)
    if (prefound) GZCHK(h, hipStreamWaitEvent(s, h->pre_ev, 0));   // (the carry is in place, device to device, and the finder has run)
    else if (nc) GZCHK(h, hipMemcpyAsync(d_comp, h->carry.data(), nc, hipMemcpyHostToDevice, s));
    if (use_staged) GZCHK(h, hipStreamWaitEvent(s, h->staged_ev[cb], 0));
    else {
        if (n_new) GZCHK(h, bzq::cache::pinned_pool().h2d(d_comp + nc, src, n_new, s));
        GZCHK(h, hipMemsetAsync(d_comp + n, 0, 64, s));
    }
    if (timing) GZCHK(h, hipStreamSynchronize(s));
    lap(0);

    JobOut* outs = (JobOut*)h_outs.p;
    uint32_t* h_counters = (uint32_t*)((uint8_t*)h_outs.p + (size_t)n_jobs_cap * sizeof(JobOut));
    {
        const uint32_t want = pool_want(h, n, n_chunks);
        if (h->pool_pages < want) h->pool_pages = want;
    }
    Args a{};
    for (int attempt = 0;; ++attempt) {
        const uint32_t pages_now = predecoded && attempt == 0 ? h->pre.pool_pages : h->pool_pages;
        if (!(predecoded && attempt == 0) && ((rc = gz_ensure(h, pool, ((size_t)h->pool_pages << PAGE_SHIFT) * 2, false, 4)) || (rc = gz_ensure(h, d_page_next, (size_t)h->pool_pages * 4)))) return rc;
        const Job j0{h->start_pos, 1, 0};
        if (!(predecoded && attempt == 0)) GZCHK(h, hipMemsetAsync(counters.p, 0, 128, s));
        static const uint32_t counting = getenv("BZQ_GZ_COUNT") ? (uint32_t)atoi(getenv("BZQ_GZ_COUNT")) : 0u;   // debug: 1 = survivor / hand-back counts (atomics in the loops: not for timing), 2 = the finder's clocks
        if (counting && !(predecoded && attempt == 0)) GZCHK(h, hipMemcpyAsync((uint32_t*)counters.p + 7, &counting, 4, hipMemcpyHostToDevice, s));
        a = Args{d_comp, (int64_t)n, (Job*)jobsB.p, (JobOut*)d_outs.p, 0, n_chunks, n_chunks, CH, (uint16_t*)pool.p, pages_now,
                 (uint32_t*)d_page_next.p, (uint32_t*)counters.p, (Event*)events.p, max_events, max_job, (int64_t)std::max<uint64_t>(out_cap, 1ull << 20)};
        a.far_bytes = h->host_cont ? std::min<int64_t>(h->far_bytes, std::max<int64_t>(64 << 10, (int64_t)(n / 4))) : 0;   // (a small piece: a quarter of it)
        if (predecoded && attempt == 0) {   // found, ordered AND decoded under the piece in front: the results are on their way to h_outs
            a.order = (const uint32_t*)orderB.p + 2 * ORDER_BINS;
            GZCHK(h, hipEventSynchronize(h->pre_dec_ev));
        } else {
            if (prefound) a.order = (const uint32_t*)orderB.p + 2 * ORDER_BINS;   // (found and ordered under the piece in front; a repeat after a pool overflow keeps that)
            else {
                GZCHK(h, hipMemcpyAsync(jobsB.p, &j0, sizeof j0, hipMemcpyHostToDevice, s));
                a.order = launch_find(s, a, orderB.p, n_chunks);
            }
            if (timing) { GZCHK(h, hipStreamSynchronize(s)); lap(1); }
            hipLaunchKernelGGL(k_gz_decode, dim3((unsigned)n_chunks), dim3(DEC_BLOCK), 0, s, a);
            GZCHK(h, hipGetLastError());
            GZCHK(h, hipMemcpyAsync(outs, d_outs.p, (size_t)n_chunks * sizeof(JobOut), hipMemcpyDeviceToHost, s));
            GZCHK(h, hipMemcpyAsync(h_counters, counters.p, 128, hipMemcpyDeviceToHost, s));
            GZCHK(h, hipStreamSynchronize(s));
        }
        lap(2);
        if (counting) fprintf(stderr, "bzq_gzip decoder: the symbol loop handed back %u times for a code it does not take / the end of a block, %u for its window, %u for a long distance code, %u for a copy it does not take, %u at a page's end\n",
                              h_counters[16], h_counters[18], h_counters[19], h_counters[20], h_counters[22]);
        if (counting) fprintf(stderr, "bzq_gzip finder clocks (x256, summed over waves): total %u, in flushes: waiting for the tables %u, the lanes' look %u, the wave's judgement %u\n", h_counters[12], h_counters[9], h_counters[10], h_counters[11]);
        if (counting) fprintf(stderr, "bzq_gzip finder: %u positions passed the 13-bit filter, %u of them the code length code test, %u of those one lane's look at the code lengths (judged by the whole wave)\n", h_counters[3], h_counters[4], h_counters[5]);
        if (h_counters[0] <= pages_now) break;
        if (attempt == 8) return gz_fail(h, BZQ_ERR_NOMEM, "bzq_gzip: the symbol pool keeps overflowing (" + std::to_string(h->pool_pages) + " pages)");
        // (h_counters[0] counts the refused requests too, but a job that was refused a page stopped there: what it would still have asked
        // for is unknown -- a quarter on top, and a page per job; doubling, as rounds 3-4 did, made a 4 GiB pool an 8 GiB one for good)
        h->pool_pages = std::max<uint32_t>(std::max(h->pool_pages, pages_now) + std::max(h->pool_pages, pages_now) / 4u, h_counters[0] + (uint32_t)n_chunks);
        h->stats.pool_retries += 1;
    }

    // ---- the chain: from the piece's exact start, from every job to the job it ended on
    std::vector<int> chain;
    int fallbacks = 0;
    unsigned long long final_pos = h->start_pos;
    int32_t final_status = ST_ERROR;
    for (int ji = 0;;) {
        const JobOut& o = outs[ji];
        chain.push_back(ji);
        if (o.status == ST_TARGET) {   // (positions only grow along the chain: the index does too)
            if (o.next_job <= ji && ji < n_chunks) return gz_fail(h, BZQ_ERR_HIP, "bzq_gzip: internal: the chain does not advance");
            if (o.next_job < 0 || o.next_job >= n_chunks) return gz_fail(h, BZQ_ERR_HIP, "bzq_gzip: internal: the chain leaves the job table");
            ji = o.next_job;
            continue;
        }
        if (o.status == ST_LOST || o.status == ST_SPLIT) {
            // it passed over too many (false) candidates, or its output reached the bound: go on from where it stopped, with an
            // explicit start.  A stream without findable block starts (fixed-Huffman or stored blocks only) proceeds like this,
            // serially; after MAX_FALLBACK restarts the call hands over what it has (*more).
            if (fallbacks == MAX_FALLBACK) { final_status = ST_SPLIT; final_pos = o.end; break; }
            const int k = n_chunks + fallbacks++;
            const uint64_t eb = (uint64_t)(o.end >> 4);   // (the first job whose chunk can hold a start at or behind it)
            const Job jfb{o.end, (int32_t)std::min<uint64_t>((uint64_t)n_chunks, shifted ? (eb < nc ? 1 : (eb - nc) / (uint64_t)CH + 1) : eb / (uint64_t)CH), 0};
            GZCHK(h, hipMemcpyAsync((Job*)jobsB.p + k, &jfb, sizeof jfb, hipMemcpyHostToDevice, s));
            a.job_base = k; a.n_jobs = 1; a.order = nullptr;
            hipLaunchKernelGGL(k_gz_decode, dim3(1), dim3(DEC_BLOCK), 0, s, a);
            GZCHK(h, hipMemcpyAsync(outs + k, (JobOut*)d_outs.p + k, sizeof(JobOut), hipMemcpyDeviceToHost, s));
            GZCHK(h, hipMemcpyAsync(h_counters, counters.p, 16, hipMemcpyDeviceToHost, s));
            GZCHK(h, hipStreamSynchronize(s));
            if (outs[k].status == ST_POOL_FULL) return gz_fail(h, BZQ_ERR_NOMEM, "bzq_gzip: symbol pool exhausted in a restart");
            h->stats.fallback_jobs += 1;
            ji = k;
            continue;
        }
        if (o.status == ST_ERROR)
            return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: invalid DEFLATE data near byte " + std::to_string(h->stats.bytes_consumed + (o.err_bit >> 3)) + " of the compressed stream");
        if (o.status == ST_TOO_BIG)
            return gz_fail(h, BZQ_ERR_NOMEM, "bzq_gzip: out_capacity (" + std::to_string(out_cap) + ") is below the output of one DEFLATE block near byte " +
                                                 std::to_string(h->stats.bytes_consumed + (o.end >> 4)) + " of the compressed stream");
        if (o.status == ST_POOL_FULL || o.status == ST_EMPTY) return gz_fail(h, BZQ_ERR_HIP, "bzq_gzip: internal: chain reached a job in state " + std::to_string(o.status));
        final_status = o.status; final_pos = o.end;
        break;
    }
    const int n_jobs_total = n_chunks + fallbacks;

    // ---- what fits the caller's buffer
    std::vector<int64_t> base(chain.size(), 0);
    uint64_t total = 0;
    size_t accepted = chain.size();
    for (size_t i = 0; i < chain.size(); ++i) {
        const uint64_t len = outs[chain[i]].out_syms;
        if (total + len > out_cap) { accepted = i; break; }
        base[i] = (int64_t)total; total += len;
    }
    if (accepted < chain.size()) {
        if (accepted == 0)
            return gz_fail(h, BZQ_ERR_NOMEM, "bzq_gzip: out_capacity (" + std::to_string(out_cap) + ") is below the output of one run of blocks (" + std::to_string(outs[chain[0]].out_syms) + ")");
        final_pos = outs[chain[accepted]].start; final_status = ST_TARGET;
        *more = 1;
    }

    std::function<int()> deferred_predecode;   // the next piece's decoders: launched behind this piece's chain / resolve / CRC kernels (below)
    // ---- the NEXT piece's finder, now: its carry is what this piece leaves behind final_pos, and that is known.  It runs on its
    // own stream under this piece's chain / resolve / CRC kernels (3 of a piece's 17 ms of kernels were the finder's).
    if (!*more && !is_last && (final_status == ST_NEED_MORE || final_status == ST_END_INPUT)) {
        const uint8_t* src2 = nullptr; uint64_t n2 = 0; int cb2 = -1; bool early = false; int nch_e = 0;
        { std::lock_guard<std::mutex> lk(h->stage_mu); if (!h->staged.empty()) { src2 = h->staged.front().src; n2 = h->staged.front().n; cb2 = h->staged.front().buf; early = h->staged.front().found; nch_e = h->staged.front().nch; } }
        const uint64_t keep2 = (uint64_t)(final_pos >> 4);
        if (cb2 >= 0 && keep2 <= n && n - keep2 <= STAGE_RESERVE && n - keep2 + n2 <= (1ull << 33)) {
            const uint64_t nc2 = n - keep2, nn = nc2 + n2;
            const int nch2 = early ? nch_e : (int)((nn + (uint64_t)CH - 1) / (uint64_t)CH), jn = jb ^ 1;
            bzq_gzip::Buf &jobsN = early ? h->jobs_e[cb2] : h->jobs[jn], &orderN = early ? h->order_e[cb2] : h->order[jn];
            if (!early && ((rc = gz_ensure(h, orderN, (size_t)2 * ORDER_BINS * 4 + (size_t)nch2 * 5 + 64)) || (rc = gz_ensure(h, jobsN, (size_t)(nch2 + MAX_FALLBACK) * sizeof(Job))))) return rc;
            uint8_t* d2 = (uint8_t*)h->comp[cb2].p + STAGE_RESERVE - nc2;
            const hipStream_t fs = h->find_stream;
            GZCHK(h, hipStreamWaitEvent(fs, early ? h->early_ev[cb2] : h->staged_ev[cb2], 0));
            if (nc2) GZCHK(h, hipMemcpyAsync(d2, d_comp + keep2, nc2, hipMemcpyDeviceToDevice, fs));
            GZCHK(h, hipEventRecord(h->pre_copy_ev, fs));
            const unsigned long long start2 = (final_pos & 1ull) ? pos_header(0) : pos_deflate((final_pos >> 1) & 7ull);
            hipLaunchKernelGGL(k_gz_job0, dim3(1), dim3(1), 0, fs, (Job*)jobsN.p, (u64)start2);
            if (early) {   // found behind its copy (gz_stage), relative to CH bytes in front of its own bytes: to where the piece starts now
                hipLaunchKernelGGL(k_gz_shift, dim3((unsigned)((nch2 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, fs, (Job*)jobsN.p + 1, nch2 - 1, ((long long)nc2 - (long long)CH) * 16);
                (void)launch_order(fs, (const Job*)jobsN.p, orderN.p, nch2);
            } else
            {
                Args a2{};
                a2.comp = d2; a2.n = (int64_t)nn; a2.jobs = (Job*)jobsN.p; a2.n_jobs = nch2; a2.n_cand = nch2; a2.chunk_bytes = CH; a2.counters = (uint32_t*)h->counters2.p;
                (void)launch_find(fs, a2, orderN.p, nch2);
            }
            GZCHK(h, hipGetLastError());
            GZCHK(h, hipEventRecord(h->pre_ev, fs));
            GZCHK(h, hipEventSynchronize(h->pre_copy_ev));   // (this piece's buffer is given back when the call returns: the carry has left it)
            h->pre.shifted = early; h->pre.nch = nch2;
            h->pre.valid = true; h->pre.src = src2; h->pre.n_new = n2; h->pre.nc = nc2; h->pre.start_pos = start2; h->pre.jb = jn; h->pre.cb = cb2;
            // ... and its DECODERS behind the finder, into the other set: they need nothing of this piece but where it ended, and this
            // piece's chain / resolve / CRC kernels (5.7 of a 17.6 ms cycle, the scalar-bound decoders idle meanwhile) run beside them
            h->pre.decoded = false;
            static const bool no_predecode = getenv("BZQ_GZ_NO_PREDECODE") != nullptr;
            if (h->predecode && !no_predecode) deferred_predecode = [=, &jobsN, &orderN]() -> int {
                int rc;
                const int pn = pb ^ 1;
                const int ncap2 = nch2 + MAX_FALLBACK;
                const uint32_t maxev2 = (uint32_t)std::min<uint64_t>(nn / 18 + (uint64_t)nch2 + 64, 1u << 28);
                const uint32_t want2 = pool_want(h, nn, nch2);
                if (h->pool_pages < want2) h->pool_pages = want2;
                if ((rc = gz_ensure(h, h->pool_[pn], ((size_t)h->pool_pages << PAGE_SHIFT) * 2, false, 4)) || (rc = gz_ensure(h, h->page_next_[pn], (size_t)h->pool_pages * 4)) ||
                    (rc = gz_ensure(h, h->outs_[pn], (size_t)ncap2 * sizeof(JobOut))) || (rc = gz_ensure(h, h->events_[pn], (size_t)maxev2 * sizeof(Event))) ||
                    (rc = gz_ensure(h, h->h_outs_[pn], (size_t)ncap2 * sizeof(JobOut) + 128, true)))
                    return rc;
                GZCHK(h, hipMemsetAsync(h->counters_[pn].p, 0, 128, fs));
                Args a3{d2, (int64_t)nn, (Job*)jobsN.p, (JobOut*)h->outs_[pn].p, 0, nch2, nch2, CH, (uint16_t*)h->pool_[pn].p, h->pool_pages,
                        (uint32_t*)h->page_next_[pn].p, (uint32_t*)h->counters_[pn].p, (Event*)h->events_[pn].p, maxev2, max_job, (int64_t)std::max<uint64_t>(out_cap, 1ull << 20)};
                a3.order = (const uint32_t*)orderN.p + 2 * ORDER_BINS;
                a3.far_bytes = h->host_cont ? std::min<int64_t>(h->far_bytes, std::max<int64_t>(64 << 10, (int64_t)(nn / 4))) : 0;
                hipLaunchKernelGGL(k_gz_decode, dim3((unsigned)nch2), dim3(DEC_BLOCK), 0, fs, a3);
                GZCHK(h, hipGetLastError());
                JobOut* ho2 = (JobOut*)h->h_outs_[pn].p;
                GZCHK(h, hipMemcpyAsync(ho2, h->outs_[pn].p, (size_t)nch2 * sizeof(JobOut), hipMemcpyDeviceToHost, fs));
                GZCHK(h, hipMemcpyAsync((uint8_t*)ho2 + (size_t)ncap2 * sizeof(JobOut), h->counters_[pn].p, 128, hipMemcpyDeviceToHost, fs));
                GZCHK(h, hipEventRecord(h->pre_dec_ev, fs));
                h->pre.decoded = true; h->pre.pb = pn; h->pre.pool_pages = h->pool_pages; h->pre.max_job = max_job;
                return 0;
            };
        }
    }

    // ---- members that ended inside the accepted part: (position in the output, trailer in the piece)
    struct MemberEnd { int64_t out_pos; uint64_t trailer; size_t ci; uint32_t seq; };
    std::vector<MemberEnd> ends;
    const uint32_t n_events = std::min(h_counters[1], max_events);
    if (n_events) {
        if ((rc = gz_ensure(h, h->h_events, (size_t)n_events * sizeof(Event), true))) return rc;
        GZCHK(h, hipMemcpyAsync(h->h_events.p, events.p, (size_t)n_events * sizeof(Event), hipMemcpyDeviceToHost, s));
        GZCHK(h, hipStreamSynchronize(s));
        std::vector<int> chain_idx((size_t)n_jobs_total, -1);
        for (size_t i = 0; i < accepted; ++i) chain_idx[(size_t)chain[i]] = (int)i;
        const Event* ev = (const Event*)h->h_events.p;
        for (uint32_t e = 0; e < n_events; ++e) {
            if ((int)ev[e].job >= n_jobs_total) continue;
            const int ci = chain_idx[ev[e].job];
            if (ci < 0 || ev[e].seq >= outs[ev[e].job].n_events || ev[e].out_syms > outs[ev[e].job].out_syms) continue;
            ends.push_back(MemberEnd{base[(size_t)ci] + (int64_t)ev[e].out_syms, ev[e].trailer_byte, (size_t)ci, ev[e].seq});
        }
        std::sort(ends.begin(), ends.end(), [](const MemberEnd& x, const MemberEnd& y) { return x.ci != y.ci ? x.ci < y.ci : x.seq < y.seq; });
    }

    // ---- work lists: chain items, resolve items, CRC segments
    const uint32_t pages_used = std::min(h_counters[0], a.pool_pages);
    if ((rc = gz_ensure(h, h->h_pages, (size_t)pages_used * 4 + 16, true))) return rc;
    if (pages_used) GZCHK(h, hipMemcpyAsync(h->h_pages.p, d_page_next.p, (size_t)pages_used * 4, hipMemcpyDeviceToHost, s));
    GZCHK(h, hipStreamSynchronize(s));
    const uint32_t* page_next = (const uint32_t*)h->h_pages.p;
    std::vector<ChainItem> citems;
    std::vector<ResItem> ritems;
    std::vector<uint32_t> pages;
    for (size_t i = 0; i < accepted; ++i) {
        const JobOut& o = outs[chain[i]];
        const int64_t len = (int64_t)o.out_syms;
        if (len == 0) continue;
        const int64_t np = (len + PAGE - 1) >> PAGE_SHIFT;
        pages.clear();
        for (uint32_t p = o.first_page; (int64_t)pages.size() < np; p = page_next[p]) {
            if (p == NO_PAGE || p >= pages_used) return gz_fail(h, BZQ_ERR_HIP, "bzq_gzip: internal: broken page list");
            pages.push_back(p);
        }
        const int64_t t = std::min<int64_t>(len, 32768);
        citems.push_back(ChainItem{base[i], len, pages[(size_t)((len - t) >> PAGE_SHIFT)], pages[(size_t)((len - 1) >> PAGE_SHIFT)]});
        for (int64_t pi = 0; pi * PAGE < len - t; ++pi)
            ritems.push_back(ResItem{pages[(size_t)pi], (uint32_t)std::min<int64_t>(PAGE, len - t - pi * PAGE), base[i] + pi * PAGE, base[i]});
    }
    std::vector<CrcSeg> segs;
    std::vector<int> seg_closes;   // per segment: index into `ends` of the member it closes, -1 = none
    {
        int64_t cur = 0;
        auto cover = [&](int64_t to, int close) {   // [cur, to) in pieces; the last one (or an empty one) carries `close`
            if (to == cur && close >= 0) { segs.push_back(CrcSeg{cur, 0, 0}); seg_closes.push_back(close); return; }
            while (cur < to) {
                const int32_t m = (int32_t)std::min<int64_t>(CRC_SEG, to - cur);
                segs.push_back(CrcSeg{cur, m, 0});
                cur += m;
                seg_closes.push_back(cur == to ? close : -1);
            }
        };
        for (size_t k = 0; k < ends.size(); ++k) cover(ends[k].out_pos, (int)k);
        cover((int64_t)total, -1);
    }
    const size_t b_chain = citems.size() * sizeof(ChainItem), b_res = ritems.size() * sizeof(ResItem), b_seg = segs.size() * sizeof(CrcSeg);
    const size_t o_res = (b_chain + 15) & ~(size_t)15, o_seg = (o_res + b_res + 15) & ~(size_t)15, b_all = o_seg + b_seg + 16;
    if ((rc = gz_ensure(h, h->items, b_all)) || (rc = gz_ensure(h, h->h_items, b_all, true)) || (rc = gz_ensure(h, h->crcs, segs.size() * 4 + 16)) ||
        (rc = gz_ensure(h, h->h_crcs, segs.size() * 4 + 16, true)))
        return rc;
    uint8_t* hi = (uint8_t*)h->h_items.p;
    if (b_chain) memcpy(hi, citems.data(), b_chain);
    if (b_res) memcpy(hi + o_res, ritems.data(), b_res);
    if (b_seg) memcpy(hi + o_seg, segs.data(), b_seg);
    lap(3);
    GZCHK(h, hipMemcpyAsync(h->items.p, hi, b_all, hipMemcpyHostToDevice, s));
    const uint8_t* w0 = (const uint8_t*)h->win[h->wcur].p;
    uint8_t* w_next = (uint8_t*)h->win[h->wcur ^ 1].p;
    {
        const int n_it = (int)citems.size();
        int per_group = 16;
        while (per_group * per_group < n_it) per_group += 8;
        const int n_groups = n_it ? (n_it + per_group - 1) / per_group : 1;
        const ChainItem* d_items = (const ChainItem*)h->items.p;
        if (h->chain_l2 && deferred_predecode) {   // the form that runs beside the next piece's decoders (windows through the L2, no LDS, <= 32 VGPRs); alone, the LDS form is 6 x faster
            if (n_groups <= 1) {
                hipLaunchKernelGGL(k_gz_chainl_out, dim3(1), dim3(CHL_THREADS), 0, s, d_items, n_it, std::max(1, n_it), (const uint16_t*)pool.p, w0, d_out, w_next);
            } else {
                if ((rc = gz_ensure(h, h->chain_maps, (size_t)n_groups * 65536)) || (rc = gz_ensure(h, h->chain_wins, (size_t)n_groups * 32768)) ||
                    (rc = gz_ensure(h, h->chain_tabs, (size_t)n_groups * 131072)))
                    return rc;
                hipLaunchKernelGGL(k_gz_chainl_tab, dim3((unsigned)n_groups), dim3(CHL_THREADS), 0, s, d_items, n_it, per_group, (const uint16_t*)pool.p, (uint16_t*)h->chain_tabs.p,
                                   (uint16_t*)h->chain_maps.p);
                hipLaunchKernelGGL(k_gz_chainl_groups, dim3(1), dim3(CHL_THREADS), 0, s, (const uint16_t*)h->chain_maps.p, n_groups, w0, (uint8_t*)h->chain_wins.p);
                hipLaunchKernelGGL(k_gz_chainl_out, dim3((unsigned)n_groups), dim3(CHL_THREADS), 0, s, d_items, n_it, per_group, (const uint16_t*)pool.p, (const uint8_t*)h->chain_wins.p, d_out,
                                   w_next);
            }
        } else if (n_groups <= 1) {
            hipLaunchKernelGGL(k_gz_chain<false>, dim3(1), dim3(CHAIN_THREADS), 0, s, d_items, n_it, std::max(1, n_it), (const uint16_t*)pool.p, w0, d_out, (uint16_t*)nullptr, w_next);
        } else {
            if ((rc = gz_ensure(h, h->chain_maps, (size_t)n_groups * 65536)) || (rc = gz_ensure(h, h->chain_wins, (size_t)n_groups * 32768))) return rc;
            hipLaunchKernelGGL(k_gz_chain<true>, dim3((unsigned)n_groups), dim3(CHAIN_THREADS), 0, s, d_items, n_it, per_group, (const uint16_t*)pool.p, (const uint8_t*)nullptr, (uint8_t*)nullptr,
                               (uint16_t*)h->chain_maps.p, (uint8_t*)nullptr);
            hipLaunchKernelGGL(k_gz_chain_groups, dim3(1), dim3(CHAIN_THREADS), 0, s, (const uint16_t*)h->chain_maps.p, n_groups, w0, (uint8_t*)h->chain_wins.p);
            hipLaunchKernelGGL(k_gz_chain<false>, dim3((unsigned)n_groups), dim3(CHAIN_THREADS), 0, s, d_items, n_it, per_group, (const uint16_t*)pool.p, (const uint8_t*)h->chain_wins.p, d_out,
                               (uint16_t*)nullptr, w_next);
        }
    }
    if (timing) { GZCHK(h, hipStreamSynchronize(s)); lap(4); }
    if (!ritems.empty())
        hipLaunchKernelGGL(k_gz_resolve, dim3((unsigned)ritems.size()), dim3(BLOCK), 0, s, (const ResItem*)((uint8_t*)h->items.p + o_res), (const uint16_t*)pool.p, w0, d_out);
    if (timing) { GZCHK(h, hipStreamSynchronize(s)); lap(5); }
    if (!segs.empty()) {
        GZCHK(h, hipMemsetAsync(h->crcs.p, 0, segs.size() * 4, s));
        hipLaunchKernelGGL(k_gz_crc, dim3((unsigned)((segs.size() * CRC_SUB + WAVES - 1) / WAVES)), dim3(BLOCK), 0, s, (const CrcSeg*)((uint8_t*)h->items.p + o_seg), (int)segs.size(),
                           (const uint8_t*)d_out, (uint32_t*)h->crcs.p);
        GZCHK(h, hipMemcpyAsync(h->h_crcs.p, h->crcs.p, segs.size() * 4, hipMemcpyDeviceToHost, s));
    }
    GZCHK(h, hipGetLastError());
    // (enqueued BEHIND this piece's last kernels: launched in front of them, the 16 000 decoder waves took every CU and the chain
    // kernels -- 1024 threads and 64 KiB of LDS a workgroup -- waited for the decoders' end: 8 ms instead of 0.9)
    const bool next_launched = (bool)deferred_predecode;
    if (deferred_predecode && (rc = deferred_predecode())) return rc;
    // ---- CRC-32 and ISIZE of every member that ended (RFC 1952 2.3.1): what the check needs from the piece's bytes is noted now (the
    // piece's buffer goes back to the reader when the call returns); the check itself waits for the CRC kernel -- here, or, with
    // option defer_verify and the next piece's decoders launched, at the start of the next call (gz_verify_pending)
    {
        bzq_gzip::PendingVerify& p = h->pend;
        p.seg_n.resize(segs.size()); p.closes.assign(segs.size(), 0); p.crc_t.assign(segs.size(), 0u); p.isize_t.assign(segs.size(), 0u);
        for (size_t k = 0; k < segs.size(); ++k) {
            p.seg_n[k] = segs[k].n;
            if (seg_closes[k] >= 0) {
                const uint64_t T = ends[(size_t)seg_closes[k]].trailer;
                uint32_t crc_t = 0, isize_t = 0;
                for (int q = 0; q < 4; ++q) { crc_t |= byte_at(T + q) << (8 * q); isize_t |= byte_at(T + 4 + q) << (8 * q); }
                p.closes[k] = 1; p.crc_t[k] = crc_t; p.isize_t[k] = isize_t;
            }
        }
        static const int defer_env = getenv("BZQ_GZ_DEFER") ? atoi(getenv("BZQ_GZ_DEFER")) : -1;   // A/B switch
        const bool defer = (defer_env >= 0 ? defer_env != 0 : h->defer_verify) && next_launched && !timing && !is_last && !*more;
        if (defer) {
            GZCHK(h, hipEventRecord(h->ver_ev, s));
            p.on = true;
            h->deferred_calls += 1;
        } else {
            GZCHK(h, hipStreamSynchronize(s));
            if ((rc = gz_verify_pending(h))) return rc;
        }
    }
    lap(6);
    h->wcur ^= 1;

    // ---- what stays for the next call
    const bool at_header = (final_pos & 1ull) != 0;
    const uint64_t keep_from = at_header ? (uint64_t)(final_pos >> 4) : (uint64_t)(final_pos >> 4);   // byte of the position (bit >> 3)
    const bool progressed = final_pos != h->start_pos || total > 0;
    {
        std::vector<uint8_t> nk;
        if (keep_from < n) {
            nk.resize((size_t)(n - keep_from));
            size_t w = 0;
            if (keep_from < nc) { memcpy(nk.data(), h->carry.data() + keep_from, (size_t)(nc - keep_from)); w = (size_t)(nc - keep_from); }
            const uint64_t s0 = keep_from > nc ? keep_from - nc : 0;
            if (n_new > s0) memcpy(nk.data() + w, src + s0, (size_t)(n_new - s0));
        }
        h->carry.swap(nk);
        h->start_pos = at_header ? pos_header(0) : pos_deflate((final_pos >> 1) & 7ull);
    }
    h->stats.pieces += 1; h->stats.bytes_in += n_new; h->stats.bytes_consumed += keep_from; h->stats.bytes_out += total;
    h->stats.chunks += (uint64_t)n_chunks; h->stats.chain_jobs += accepted; h->stats.members = h->members_done;
    for (int c = 1; c < n_chunks; ++c) h->stats.chunks_with_start += outs[c].status != ST_EMPTY;
    *out_bytes = total;
    lap(7);
    if (timing)
        fprintf(stderr, "bzq_gzip piece: %.1f MiB in -> %.1f MiB out, %d chunks, %zu chain jobs | ms: h2d %.2f find %.2f decode %.2f host-plan %.2f chain %.2f resolve %.2f crc %.2f host-verify+carry %.2f\n",
                n / 1048576.0, total / 1048576.0, n_chunks, accepted, t_ph[0], t_ph[1], t_ph[2], t_ph[3], t_ph[4], t_ph[5], t_ph[6], t_ph[7]);

    if (final_status == ST_EVENTS_FULL || final_status == ST_SPLIT) *more = 1;   // (cannot happen with the event table sized as it is; call again)
    if (final_status == ST_FAR) {   // the next call continues on the host; the longer the device keeps handing over, the longer the host stays at it
        *more = 1; h->host_next = true;
        h->host_budget = total > (8ull << 20) ? h->host_budget_min : std::min<uint64_t>(h->host_budget * 2, 1ull << 30);
    } else if (total > (8ull << 20)) h->host_budget = h->host_budget_min;
    if (*more) return 0;
    // bytes behind the last member that are no member are ignored like gzread ignores them -- but a member whose magic (1f 8b) is
    // there and whose header or first block is cut off is a TRUNCATED file, never a clean end (gzread: "unexpected end of file"
    // once the magic has matched)
    const bool cut_member = at_header && is_last && final_status == ST_NEED_MORE && h->carry.size() >= 2 && h->carry[0] == 0x1f && h->carry[1] == 0x8b;
    const bool garbage = at_header && !cut_member && (final_status == ST_BAD_HEADER || (is_last && final_status == ST_NEED_MORE));
    if (final_status == ST_BAD_HEADER && h->members_done == 0) return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: not a gzip stream (no member header at its start)");
    if (garbage && h->members_done > 0) { h->finished = true; h->carry.clear(); return 0; }   // trailing bytes that are no member: ignored, like gzread
    if (is_last) {
        if (final_status == ST_END_INPUT) { h->finished = true; return 0; }
        return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: unexpected end of the gzip stream (truncated file)");
    }
    if (!progressed && h->carry.size() > (1ull << 31)) return gz_fail(h, BZQ_ERR_IO, "bzq_gzip: 2 GiB of compressed bytes hold no whole DEFLATE block");
    return 0;
}

} // namespace gz
} // namespace bzq
