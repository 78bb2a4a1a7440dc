// bzq_inflate.hpp -- BGZF blocks inflated ON the GPU (C ABI bzq_bgzf_inflate; the ingest's "ingest_gpu_inflate" option).
//
// Replaces, for blocked gzip, the reference's RapidgzipReader / GZFile in front of the parser (blazeseq/io/readers.mojo:283-443):
// there the host decompresses and the parser sees plain bytes; here the COMPRESSED bytes cross PCIe (3-4x fewer) and every
// BGZF block -- an independent raw-DEFLATE stream of at most 64 KiB of output (SAM spec 4.1; RFC 1951) -- is decoded by one
// wave64.  A 3 GB chunk is ~47 000 blocks: the parallelism is across blocks, the decode inside a block is serial, so the wave
// runs it as UNIFORM code (bit buffer, positions and symbols live in scalar registers) and uses its 64 lanes where DEFLATE
// offers width:
//   * the input window: one coalesced load puts 256 bytes of the stream into a VGPR (lane i = dword i); refills of the bit
//     buffer are v_readlane with a scalar index, no memory latency on the critical path;
//   * Huffman decode without big tables: canonical codes are ordered by length, so with the next 15 stream bits reversed into
//     a left-aligned code c, lane L holds the left-aligned upper bound of the codes of length L and ONE compare + ballot +
//     s_ff1 gives the length; first code and symbol offset of that length are readlanes, the symbol one LDS read
//     (per wave: 288 + 32 sorted symbols = 640 bytes of LDS); the literal/length code also has a 10-bit direct table (2 KiB)
//     for its short codes, and runs of literals stay in a loop of their own (build_lut, inflate_block);
//   * table construction: lengths histogram by ballots, the sort of the symbols by (length, symbol) by ballots and popcounts;
//   * literals collect in a VGPR (lane k = k-th pending byte) and leave 64 at a time; matches are copied by all lanes.
// Output goes straight to the chunk buffer in device memory; matches read it back (same wave, same L1: program order holds).
// ISIZE, every distance and every length are checked while decoding, and the CRC-32 of the block's output afterwards (the wave
// re-reads its 64 KiB: 64 lanes x one contiguous piece each, combined by multiplying every piece's remainder by x^(8 * bytes
// behind it) -- 0.1 % of the decode time).  A block that fails any of it fails the whole call.
#pragma once
#include "bzq_device.hpp"

namespace bzq {
namespace inf {

constexpr int WAVES = BLOCK / 64;
struct DevBlock { uint64_t coff, uoff; uint32_t csize, usize, crc, pad; };   // deflate payload [coff, coff + csize) -> out[uoff, uoff + usize), CRC-32 of the output
struct Args { const uint8_t* comp; uint64_t comp_bytes; const DevBlock* blocks; int64_t n_blocks; uint8_t* out; unsigned long long* first_bad; };

struct __attribute__((packed, aligned(1))) U32U { uint32_t v; };

__device__ __forceinline__ uint32_t rdlane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// the bit reader of one wave: everything uniform except `win` (lane i = dword win_base + i of the stream)
struct Bits {
    const uint8_t* base;    // first byte of the stream
    int64_t limit;          // bytes that may be read behind base (payload + padding of the buffer)
    u64 buf; int cnt;
    int next;               // next dword to enter buf
    int win_base;
    uint32_t win;
    __device__ __forceinline__ void start(const uint8_t* p, int64_t lim) { base = p; limit = lim; buf = 0; cnt = 0; next = 0; win_base = -64; win = 0; }
    __device__ __forceinline__ void refill() {
        while (cnt <= 32) {
            if (next - win_base >= 64) {
                win_base = next;
                const int64_t o = 4ll * (win_base + (int)(threadIdx.x & 63));
                win = o + 4 <= limit ? reinterpret_cast<const U32U*>(base + o)->v : 0u;
            }
            buf |= (u64)rdlane(win, next - win_base) << cnt;
            cnt += 32; ++next;
        }
        pin();
    }
    // the state is the same in every lane; say so (the compiler otherwise keeps it in vector registers and runs the whole
    // decoder under exec masks)
    __device__ __forceinline__ void pin() {
        cnt = (int)uni((uint32_t)cnt); next = (int)uni((uint32_t)next); win_base = (int)uni((uint32_t)win_base);
        buf = ((u64)uni((uint32_t)(buf >> 32)) << 32) | uni((uint32_t)buf);
    }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = (uint32_t)buf & ((1u << n) - 1u); buf >>= n; cnt -= n; return v; }
    __device__ __forceinline__ int64_t byte_pos() const { return 4ll * next - (cnt >> 3); }   // after discarding to a byte edge
};

// One canonical Huffman code, spread over the lanes: lane L (1..15) holds, for the codes of length L, the left-aligned (15-bit)
// upper bound `lim`, the first code `first` and the index of its first symbol in the sorted symbol table `offs`.
struct Code { uint32_t lim, first, offs; int n_coded; int kraft; };   // kraft: code space used, in units of 2^-15 (32768 = complete)

// zlib's verdict on a set of code lengths (inflate_table, inftrees.c): over-subscribed sets are refused; an incomplete set only
// passes as a single code of one bit (literal/length or distance code), or as no code at all (distance code of a block
// without matches); the code length code must be complete.
__device__ __forceinline__ bool code_valid(const Code& c, bool is_dist, bool is_precode) {
    if (c.kraft == 32768) return true;
    if (c.kraft > 32768 || is_precode) return false;
    return (c.n_coded == 1 && c.kraft == 16384) || (is_dist && c.n_coded == 0);
}

// lens[0..n) (LDS, one byte per symbol, 0 = unused) -> Code + symtab (LDS).  Returns false when the lengths over-subscribe the
// code space.  n <= 320.
__device__ __forceinline__ bool build_code(const uint8_t* lens, int n, uint16_t* symtab, Code& code) {
    const int lane = threadIdx.x & 63;
    uint32_t count = 0;      // lane L: number of symbols of length L
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int l = s0 + lane < n ? lens[s0 + lane] : 0;
#pragma unroll
        for (int L = 1; L <= 15; ++L) {
            const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(l == L));
            if (lane == L) count += c;
        }
    }
    // first code, symbol offset, left-aligned limit per length (a 15-step uniform recurrence)
    uint32_t first = 0, offs = 0, lim = 0, c = 0, o = 0;
    bool over = false;
#pragma unroll
    for (int L = 1; L <= 15; ++L) {
        const uint32_t n_l = rdlane(count, L);
        c <<= 1;
        if (lane == L) { first = c; offs = o; lim = (c + n_l) << (15 - L); }
        if (c + n_l > (1u << L)) over = true;
        c += n_l; o += n_l;
    }
    code.first = first; code.offs = offs; code.lim = (lane >= 1 && lane <= 15) ? lim : 0u; code.n_coded = (int)o; code.kraft = over ? 65536 : (int)c;
    // symbols sorted by (length, symbol): position = offs[length] + symbols of the same length before it
    uint32_t run = 0;        // lane L: symbols of length L placed so far
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int l = s0 + lane < n ? lens[s0 + lane] : 0;
        uint32_t pos = 0;
#pragma unroll
        for (int L = 1; L <= 15; ++L) {
            const u64 m = __ballot(l == L);
            const uint32_t basepos = rdlane(offs, L) + rdlane(run, L);
            if (l == L) pos = basepos + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (lane == L) run += (uint32_t)__builtin_popcountll(m);
        }
        if (l) symtab[pos] = (uint16_t)(s0 + lane);
    }
    __builtin_amdgcn_wave_barrier();
    return !over;
}

// next symbol of `code` (uniform); -1 when the bits are no code.  Consumes its bits.
template <class BitsT>   // (inf::Bits, or the absolute-position reader of bzq_gzip.hpp: only buf / cnt are touched)
__device__ __forceinline__ int decode_sym(BitsT& b, const Code& code, const uint16_t* symtab) {
    const uint32_t c = __builtin_bitreverse32((uint32_t)b.buf) >> 17;
    const u64 m = __ballot(c < code.lim);
    if (!m) return -1;
    const int L = __builtin_ctzll(m);
    const uint32_t idx = rdlane(code.offs, L) + (c >> (15 - L)) - rdlane(code.first, L);
    b.buf >>= L; b.cnt -= L;
    return (int)uni(symtab[idx]);
}

// The literal/length code also gets a direct table for its codes of up to LUT_BITS bits (in FASTQ: nearly all of them):
// lut[next LUT_BITS stream bits] = symbol | length << 12, LUT_LONG = a longer code (decode_sym).  One LDS read per symbol instead of
// compare + ballot + two readlanes + the read.  Filled from the sorted symbols: the code of the symbol at sorted position p is
// first[l] + (p - offs[l]), its bit-reversed value r selects the entries r + k * 2^l.
constexpr int LUT_BITS = 10;
constexpr uint32_t LUT_LONG = 0x100u;   // entry of a code longer than LUT_BITS (bit 8 set like every non-literal, length 0)
template <int BITS = LUT_BITS>
__device__ __forceinline__ void build_lut(const Code& code, const uint16_t* symtab, const uint8_t* lens, uint16_t* lut) {
    const int lane = threadIdx.x & 63;
    uint32_t* lut32 = reinterpret_cast<uint32_t*>(lut);
#pragma unroll
    for (int i = 0; i < (1 << BITS) / 2 / 64; ++i) lut32[i * 64 + lane] = LUT_LONG * 0x10001u;
    __builtin_amdgcn_wave_barrier();
    for (int p0 = 0; p0 < code.n_coded; p0 += 64) {
        const int p = p0 + lane;
        const uint32_t sym = p < code.n_coded ? symtab[p] : 0u;
        const int l = p < code.n_coded ? lens[sym] : 0;
        // (every lane takes part in the gathers: ds_bpermute reads nothing from a lane that is switched off)
        const uint32_t f = (uint32_t)__builtin_amdgcn_ds_bpermute(l * 4, (int)code.first), o = (uint32_t)__builtin_amdgcn_ds_bpermute(l * 4, (int)code.offs);
        if (l >= 1 && l <= BITS) {
            const uint32_t r = __builtin_bitreverse32(f + ((uint32_t)p - o)) >> (32 - l);
            const uint16_t e = (uint16_t)(sym | ((uint32_t)l << 12));
            for (uint32_t k = r; k < (1u << BITS); k += 1u << l) lut[k] = e;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// The table the literal loop reads: 32-bit entries, up to TWO literals per lookup.  The decoder is bound by instruction issue, not by
// latency, and in FASTQ most symbols are literals with short codes (bases: 2-3 bits, qualities: 4-6), so two of
// them usually fit the 10 index bits -- one trip through the loop then delivers two bytes.
//   bit 31      the first symbol is not a literal with a code of <= LUT_BITS bits.  Bit 30 set: a LENGTH symbol (257..285) with
//               such a code, base and extra bits folded in, laid out so that the entry itself is the operand of the s_bfe_u32
//               that pulls the extra bits out of the bit buffer (offset = S1[4:0], width = S1[22:16], the rest is ignored):
//               bits 0..4 = code bits, 16..18 = number of extra bits; bits 5..13 = base length, 23..27 = code + extra bits.
//               Bit 30 clear: bits 0..15 = the one-symbol entry of build_lut (end of block, a longer code, no code)
//   bits 0..7   first literal, bits 8..15 second literal, bits 24..25 = how many (1 or 2), bits 16..20 = code bits consumed by both
// lut2 = LUT_BITS-indexed, 4 KiB; its upper half serves as the one-symbol table while it is built.
__device__ __forceinline__ void build_lut2(const Code& code, const uint16_t* symtab, const uint8_t* lens, uint32_t* lut2) {
    const int lane = threadIdx.x & 63;
    uint16_t* lut1 = reinterpret_cast<uint16_t*>(lut2) + (1 << LUT_BITS);
    build_lut(code, symtab, lens, lut1);
    uint32_t ent[(1 << LUT_BITS) / 64];
#pragma unroll
    for (int k = 0; k < (1 << LUT_BITS) / 64; ++k) {
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
        const uint32_t e2 = lut1[i >> l1];   // (the bits above the 10 - l1 valid ones are zero: a code of <= 10 - l1 bits does not look at them)
        const uint32_t l2 = e2 >> 12;
        const bool two = !(e2 & 0x100u) && l1 + l2 <= (uint32_t)LUT_BITS;
        ent[k] = (e1 & 0xFFu) | (two ? ((e2 & 0xFFu) << 8) | (2u << 24) | ((l1 + l2) << 16) : (1u << 24) | (l1 << 16));
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < (1 << LUT_BITS) / 64; ++k) lut2[k * 64 + lane] = ent[k];
    __builtin_amdgcn_wave_barrier();
}
// lane `lane` of `old` := v (both uniform): v_writelane_b32 with the lane select in M0 (a VALU instruction takes one SGPR; M0 is
// extra) -- two instructions where "if (lane_id == lane) old = v" costs three.  This compiler has no builtin for it.  M0 is
// written in the same asm statement that reads it (the symbol loop's LDS-direct loads do the same with their base).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ uint32_t wrlane(uint32_t v, int lane, uint32_t old) {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(uni(v)), "s"((int)uni((uint32_t)lane)) : "m0");
    return old;
}

// The distance code's direct table: DLUT_BITS index bits, 32-bit entries with base and extra bits folded in (RFC 1951 3.2.5):
// bits 0..14 = base distance, 16..19 = number of extra bits, 20..24 = code bits; bit 31 = a longer code, no code, or one of
// the two symbols that must not occur (30, 31): the slow path judges those.  Its upper half serves as the 16-bit table while it
// is built.
constexpr int DLUT_BITS = 8;
__device__ __forceinline__ void build_dlut(const Code& code, const uint16_t* symtab, const uint8_t* lens, uint32_t* dlut) {
    const int lane = threadIdx.x & 63;
    uint16_t* t16 = reinterpret_cast<uint16_t*>(dlut) + (1 << DLUT_BITS);
    build_lut<DLUT_BITS>(code, symtab, lens, t16);
    uint32_t ent[(1 << DLUT_BITS) / 64];
#pragma unroll
    for (int k = 0; k < (1 << DLUT_BITS) / 64; ++k) {
        const uint32_t e1 = t16[k * 64 + lane], sym = e1 & 0xFFFu;
        if (e1 == LUT_LONG || sym > 29u) { ent[k] = 0x80000000u; continue; }
        const uint32_t ext = sym < 4u ? 0u : (sym - 2u) >> 1;
        const uint32_t base = sym < 4u ? 1u + sym : 1u + ((2u + (sym & 1u)) << ext);
        ent[k] = base | (ext << 16) | ((e1 >> 12) << 20);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < (1 << DLUT_BITS) / 64; ++k) dlut[k * 64 + lane] = ent[k];
    __builtin_amdgcn_wave_barrier();
}

// The same table with 64-bit entries, for sym_run_ob: the low word IS the operand of the s_bfe_u32 that pulls the extra bits out of the
// bit buffer BEFORE the code is shifted out (bits 0..4 = offset = code bits, 16..22 = width = number of extra bits), bits 23..27 = code
// bits + extra bits (one shift consumes both), bit 31 as above; the high word = the base distance.  Five scalar instructions per match
// where the 32-bit entry takes ten (two shifts of the buffer, a mask built and applied, the base masked out).  2 KiB; its last 512 bytes
// serve as the 16-bit table while it is built.
__device__ __forceinline__ void build_dlut64(const Code& code, const uint16_t* symtab, const uint8_t* lens, uint32_t* dlut) {
    const int lane = threadIdx.x & 63;
    uint16_t* t16 = reinterpret_cast<uint16_t*>(dlut) + 3 * (1 << DLUT_BITS);
    build_lut<DLUT_BITS>(code, symtab, lens, t16);
    uint32_t lo[(1 << DLUT_BITS) / 64], hi[(1 << DLUT_BITS) / 64];
#pragma unroll
    for (int k = 0; k < (1 << DLUT_BITS) / 64; ++k) {
        const uint32_t e1 = t16[k * 64 + lane], sym = e1 & 0xFFFu;
        if (e1 == LUT_LONG || sym > 29u) { lo[k] = 0x80000000u; hi[k] = 0u; continue; }
        const uint32_t ext = sym < 4u ? 0u : (sym - 2u) >> 1, bits = e1 >> 12;
        lo[k] = bits | (ext << 16) | ((bits + ext) << 23);
        hi[k] = sym < 4u ? 1u + sym : 1u + ((2u + (sym & 1u)) << ext);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < (1 << DLUT_BITS) / 64; ++k) { dlut[2 * (k * 64 + lane)] = lo[k]; dlut[2 * (k * 64 + lane) + 1] = hi[k]; }
    __builtin_amdgcn_wave_barrier();
}

// State handed in and out of the hand-written symbol loop.  (pos: bytes stored; ns: literals decoded and not yet stored, the
// C++ slow paths' business only; len / dist: a match decoded and not yet copied; e: the table entry the loop stopped at.)
struct SymState { int pos, ns, len, dist; uint32_t e; };
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// ---- the symbol loop of a block, by hand ------------------------------------------------------------------------------------------
// PMC on the C++ loop (round 2): 1.03 scalar instructions per CU cycle, 21.6 per output byte -- and in the benchmark's FASTQ 71 %
// of the bytes come out of matches of 5.5 bytes on average, so what the match path costs decides the rate.  What the compiler
// makes of the C++ loop is ~41 instructions per literal and ~120 per match (SGPR spills reloaded inside it, the reader's
// counters kept in vector registers, selects and copies around every exit).  A first hand-written generation (sym_run, round
// 3, in the history: refill from the window register by v_readlane, one ds_read_b32 per lookup, literals into a pending
// register by v_writelane, the whole match path inside the block) took that to ~21 scalar-heavy instructions per literal
// pair and ~70 per match: 26 -> 44 GB/s.  This is the second.
// What bounds these kernels is the CU's ONE scalar unit (experiments/micro/salu_loop.hip and PMC: the waves of a CU want
// several scalar instructions per cycle, it issues one), so this loop needs few of them and puts what it can on the vector
// unit: the table index is computed there (v_bfe, v_lshl_add), literals go from the entry register to the output buffer
// without passing a scalar register (lanes 0 and 1 are the only active ones between matches: lane 0 writes the first literal
// of the entry, lane 1 the second), a length entry is its own s_bfe operand.  11 scalar instructions per lookup of one or
// two literals, ~50 per match (sym_run: 15 and ~70).
// The bytes do not go to memory one match at a time: they collect in an LDS buffer (`obuf`, OB_SLOTS dwords, one byte each)
// that is written out 64 lanes wide when it holds more than OB_FLUSH.  That lets a match NOT wait for its source:
// global_load_lds_ubyte puts lane i's byte straight into buffer slot i of the match (a zero-extended DWORD per lane at
// M0 + 4 * lane: experiments/micro/glds_u16.hip), and the loop goes on decoding while it travels.  A source still in the buffer
// is copied inside it (after a wait, if a load into those very slots may be in flight: s57 = the lowest slot written by a
// load since the last vmcnt(0)); one that straddles buffer and memory is handed back (4).  Every way out writes the buffer
// out: outside this block `out` holds everything in front of st.pos.  No pending literals in here (st.ns must be 0).
// Fixed scalar registers (named in the clobber list) because inline asm cannot name the halves of a 64-bit operand.  It leaves with
//   0  an entry that is neither literal nor a folded length symbol (end of block, long code, no code): e, bits not consumed
//   2  the window register is used up (refill() and come back; with `len` > 0: come back into the distance half)
//   3  a distance code the table does not hold: len is decoded, the distance bits are not consumed
//   4  a match whose source straddles buffer and memory (what is left of it: len, dist).  Long matches and matches that overlap
//      themselves stay inside: in pieces of up to 63 bytes, lane i of a piece reading source byte i mod dist (label 40)
//   5  invalid: distance beyond the start of the output, or output beyond usize
//   6  fewer than two bytes of the block's size are left (the caller decodes one symbol itself; the bit buffer is refilled)
constexpr int OB_SLOTS = 192, OB_FLUSH = 128;   // (at most OB_FLUSH + 2 literals, or OB_FLUSH + a match of 63 bytes; the asm has the 128)
template <class BitsT>
__device__ __forceinline__ int sym_run_ob(BitsT& b, const uint32_t* lut2, const uint32_t* dlut, uint32_t* obuf, uint8_t* out, int usize, SymState& st) {
    uint32_t reason, vt, vt2, vq, ve, vh, vslot, vsrc, ee;
    u64 buf = ((u64)uni((uint32_t)(b.buf >> 32)) << 32) | uni((uint32_t)b.buf);
    int cnt = (int)uni((uint32_t)b.cnt), next = (int)uni((uint32_t)b.next), pos = (int)uni((uint32_t)st.pos);
    int len = (int)uni((uint32_t)st.len), dist = 0;
    const int din = (int)uni((uint32_t)st.dist);   // != 0: the match's distance is known (the caller decoded a long distance code): straight to the copy
    const int wb = (int)uni((uint32_t)b.win_base), us2 = (int)uni((uint32_t)(usize - 2)), us = (int)uni((uint32_t)usize);
    const uint32_t lds = uni((uint32_t)(uintptr_t)lut2), ldd = uni((uint32_t)(uintptr_t)dlut), ldo = uni((uint32_t)(uintptr_t)obuf);
    const u64 ob = ((u64)uni((uint32_t)((uintptr_t)out >> 32)) << 32) | uni((uint32_t)(uintptr_t)out);
    const uint32_t lane = threadIdx.x & 63, lane4 = lane << 2, sh8 = lane << 3;
    const float lh = (float)lane + 0.5f;
    asm volatile(
        "\ts_mov_b64 s[68:69], exec\n"
        "\ts_mov_b64 s[40:41], %[buf]\n"
        "\ts_sub_i32 s42, %[cnt], 33\n"
        "\ts_mov_b32 s43, %[next]\n"
        "\ts_mov_b32 s44, %[pos]\n"
        "\ts_mov_b32 s51, %[len]\n"
        "\ts_mov_b32 s53, %[us2]\n"
        "\ts_mov_b32 s63, %[us]\n"
        "\ts_mov_b32 s54, %[wb]\n"
        "\ts_mov_b32 s55, %[lds]\n"
        "\ts_mov_b32 s56, %[ldd]\n"
        "\ts_mov_b64 s[60:61], %[ob]\n"
        "\ts_mov_b32 s65, %[obuf]\n"
        "\ts_mov_b32 s52, %[din]\n"
        "\ts_mov_b32 s64, 0\n"
        "\ts_mov_b32 s57, 0x7fffffff\n"
        "\ts_sub_i32 s67, s53, s44\n"
        "\ts_min_i32 s67, s67, 128\n"
        "\tv_add_u32 %[vslot], s65, %[lane4]\n"
        "\ts_mov_b64 exec, 3\n"
        "\ts_cmp_lg_u32 s52, 0\n"
        "\ts_cbranch_scc1 47f\n"
        "\ts_cmp_lg_u32 s51, 0\n"
        "\ts_cbranch_scc1 4f\n"
        // ---- between symbols: room for two more literals in buffer and block?
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
        // (s42 >= 0 here, so the subtraction's borrow IS "fewer than 33 bits left"; then the room test of label 8 in place: two scalar
        // instructions fewer per lookup than a branch to 8 and its two tests)
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
        "\ts_sub_u32 s42, s42, s47\n"       // (s42 >= 0 here: the borrow is the refill test; 11 comes back through label 4)
        "\ts_cbranch_scc1 11f\n"
        // ---- the distance (label 4, out of line: the ways in with a length pending -- the call's start, a refill)
        "5:\n"
        "\tv_bfe_u32 %[vt], s40, 0, 8\n"
        "\tv_lshl_add_u32 %[vt], %[vt], 3, s56\n"
        "\tds_read_b32 %[ve], %[vt]\n"
        "\tds_read_b32 %[vh], %[vt] offset:4\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tv_readfirstlane_b32 s46, %[ve]\n"
        "\ts_cmp_lt_i32 s46, 0\n"
        "\ts_cbranch_scc1 83f\n"
        // (build_dlut64: the entry's low word is the operand that takes the extra bits from behind the code, its high word the base)
        "\tv_readfirstlane_b32 s52, %[vh]\n"
        "\ts_bfe_u32 s48, s40, s46\n"
        "\ts_add_i32 s52, s52, s48\n"
        "\ts_bfe_u32 s47, s46, 0x50017\n"
        "\ts_lshr_b64 s[40:41], s[40:41], s47\n"
        "\ts_sub_i32 s42, s42, s47\n"
        // ---- invalid: a source in front of the block's output, output beyond the block's size
        "47:\n"
        "\ts_add_i32 s47, s44, s64\n"
        "\ts_cmp_gt_u32 s52, s47\n"
        "\ts_cbranch_scc1 85f\n"
        "\ts_add_i32 s48, s47, s51\n"
        "\ts_cmp_gt_i32 s48, s63\n"
        "\ts_cbranch_scc1 85f\n"
        // ---- the common kind: up to 63 bytes, not overlapping itself (the others: 70, in pieces)
        "\ts_min_u32 s48, s52, 63\n"
        "\ts_cmp_gt_u32 s51, s48\n"
        "\ts_cbranch_scc1 40f\n"
        // where the source lies: dist - len >= what the buffer holds -> all of it is in memory
        "\ts_sub_i32 s48, s52, s51\n"
        "\ts_cmp_ge_u32 s48, s64\n"
        "\ts_cbranch_scc0 65f\n"
        // far: lane i < len fetches the byte at pos - dist + i STRAIGHT INTO its buffer slot (LDS-direct load: nothing to wait for)
        "\ts_bfm_b64 exec, s51, 0\n"
        "\ts_sub_i32 s47, s47, s52\n"
        "\tv_add_u32 %[vq], s47, %[lane]\n"
        "\ts_lshl2_add_u32 m0, s64, s65\n"
        "\ts_min_u32 s57, s57, s64\n"
        "\ts_nop 0\n"
        "\tglobal_load_lds_ubyte %[vq], s[60:61]\n"
        "63:\n"
        "\ts_mov_b64 exec, 3\n"
        "\ts_add_i32 s64, s64, s51\n"
        "\tv_lshl_add_u32 %[vslot], s51, 2, %[vslot]\n"
        "\ts_mov_b32 s51, 0\n"
        "\ts_cmp_le_i32 s64, s67\n"
        "\ts_cbranch_scc1 1b\n"
        "\ts_branch 8b\n"
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
        "42:\n"
        "\ts_min_u32 s57, s57, s64\n"
        "\ts_nop 0\n"
        "\tglobal_load_lds_ubyte %[vq], s[60:61]\n"
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
        "4:\n"
        "\ts_cmp_ge_i32 s42, 0\n"
        "\ts_cbranch_scc1 5b\n"
        "\ts_branch 11f\n"
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
        // ---- the buffer (s64 bytes, a dword each) to the output: everything in flight has landed first
        "30:\n"
        "\ts_mov_b64 exec, s[68:69]\n"
        "\ts_waitcnt vmcnt(0)\n"
        "\tv_add_u32 %[vq], s44, %[lane]\n"
        "\tv_add_u32 %[vt], s65, %[lane4]\n"
        "\tv_cmp_gt_i32 vcc, s64, %[lane]\n"
        "\ts_and_saveexec_b64 s[58:59], vcc\n"
        "\tds_read_b32 %[vt2], %[vt]\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tglobal_store_byte %[vq], %[vt2], s[60:61]\n"
        "\ts_mov_b64 exec, s[58:59]\n"
        "\ts_sub_i32 s48, s64, 64\n"
        "\tv_cmp_gt_i32 vcc, s48, %[lane]\n"
        "\ts_and_saveexec_b64 s[58:59], vcc\n"
        "\tds_read_b32 %[vt2], %[vt] offset:256\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tglobal_store_byte %[vq], %[vt2], s[60:61] offset:64\n"
        "\ts_mov_b64 exec, s[58:59]\n"
        "\ts_sub_i32 s48, s64, 128\n"
        "\tv_cmp_gt_i32 vcc, s48, %[lane]\n"
        "\ts_and_saveexec_b64 s[58:59], vcc\n"
        "\tds_read_b32 %[vt2], %[vt] offset:512\n"
        "\ts_waitcnt lgkmcnt(0)\n"
        "\tglobal_store_byte %[vq], %[vt2], s[60:61] offset:128\n"
        "\ts_mov_b64 exec, s[58:59]\n"
        "\ts_add_i32 s44, s44, s64\n"
        "\ts_mov_b32 s64, 0\n"
        "\ts_mov_b32 s57, 0x7fffffff\n"
        "\ts_sub_i32 s67, s53, s44\n"
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
        "85:\n"
        "\ts_mov_b32 s50, 5\n"
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
          [vt] "=&v"(vt), [vt2] "=&v"(vt2), [vq] "=&v"(vq), [ve] "=&v"(ve), [vh] "=&v"(vh), [vslot] "=&v"(vslot), [vsrc] "=&v"(vsrc), [e] "=s"(ee), [reason] "=s"(reason)
        : [wb] "s"(wb), [us2] "s"(us2), [us] "s"(us), [din] "s"(din), [win] "v"(b.win), [lane] "v"(lane), [lane4] "v"(lane4), [sh8] "v"(sh8), [lh] "v"(lh), [lds] "s"(lds), [ldd] "s"(ldd), [obuf] "s"(ldo), [ob] "s"(ob)
        : "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61",
          "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "m0", "scc", "vcc", "memory");
    b.buf = buf; b.cnt = cnt; b.next = next; st.pos = pos; st.len = len; st.dist = dist; st.e = ee;
    return (int)reason;
}
#pragma clang diagnostic pop

// RFC 1951 3.2.5: base and extra bits of the length codes 257..285 and the distance codes 0..29, lane i = code i
__device__ __forceinline__ void length_dist_tables(uint32_t& lbase, uint32_t& lext, uint32_t& dbase, uint32_t& dext) {
    const int i = threadIdx.x & 63;
    lext = (i < 8 || i >= 28) ? 0u : (uint32_t)((i - 4) >> 2);
    lbase = i < 8 ? 3u + i : (i >= 28 ? 258u : 3u + ((4u + (uint32_t)(i & 3)) << lext));
    dext = i < 4 ? 0u : (uint32_t)((i - 2) >> 1);
    dbase = i < 4 ? 1u + i : 1u + ((2u + (uint32_t)(i & 1)) << dext);
}

// order in which the lengths of the code length code are stored (RFC 1951 3.2.7)
static __device__ const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ---- CRC-32 (IEEE 802.3, reflected, polynomial 0xEDB88320: RFC 1952 8.) of a block's output, by the wave that wrote it ----
// With R(M) = M(x) * x^32 mod P (zero initial register, no final inversion) the remainder of a concatenation is
// R(A || B) = R(A) * x^(8 |B|) + R(B), so every lane takes one contiguous piece, multiplies its remainder by x^(8 * bytes
// behind the piece) and the pieces are XORed; the all-ones initial register and the final inversion of the standard CRC are
// the term 0xFFFFFFFF * x^(8 |M|) and a last XOR.  Polynomials in the reflected order: bit 31 = x^0.
constexpr uint32_t CRC_POLY = 0xEDB88320u;
__device__ __forceinline__ uint32_t crc_mul(uint32_t a, uint32_t b) {   // a * b mod P
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x^(8 n) mod P from x2n[k] = x^(2^k) mod P
__device__ __forceinline__ uint32_t crc_x8n(uint32_t n, const uint32_t* x2n) {
    uint32_t p = 0x80000000u;
    for (int k = 3; n; n >>= 1, ++k)
        if (n & 1u) p = crc_mul(x2n[k & 31], p);
    return p;
}
// s_tab[256]: the byte table; s_x2n[32]: x^(2^k).  Filled by the first wave of the workgroup (a barrier follows).
__device__ __forceinline__ void crc_tables(uint32_t* s_tab, uint32_t* s_x2n) {
    const int t = threadIdx.x;
    if (t < 256) {
        uint32_t c = (uint32_t)t;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        s_tab[t] = c;
    }
    if (t == 0) {
        uint32_t p = 0x40000000u;   // x^1
        for (int k = 0; k < 32; ++k) { s_x2n[k] = p; p = crc_mul(p, p); }
    }
}
__device__ __forceinline__ uint32_t block_crc32(const uint8_t* out, int n, const uint32_t* s_tab, const uint32_t* s_x2n) {
    const int lane = threadIdx.x & 63;
    const int seg = (n + 63) >> 6;
    const int lo = lane * seg < n ? lane * seg : n, hi = lo + seg < n ? lo + seg : n;
    uint32_t r = 0;
    int i = lo;
    // (64 bytes per trip, the four loads of a cache line back to back: the lanes' pieces lie seg bytes apart, and a dword per
    // load fetched every line from L2 sixteen times)
    struct W4 { uint32_t w[4]; };
    struct __attribute__((packed, aligned(1))) W4U { W4 v; };
    for (; i + 64 <= hi; i += 64) {
        W4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = reinterpret_cast<const W4U*>(out + i + 16 * j)->v;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t w = q[j >> 2].w[j & 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) { r = s_tab[(r ^ w) & 0xFFu] ^ (r >> 8); w >>= 8; }
        }
    }
    for (; i + 4 <= hi; i += 4) {
        uint32_t w = reinterpret_cast<const U32U*>(out + i)->v;
#pragma unroll
        for (int k = 0; k < 4; ++k) { r = s_tab[(r ^ w) & 0xFFu] ^ (r >> 8); w >>= 8; }
    }
    for (; i < hi; ++i) r = s_tab[(r ^ out[i]) & 0xFFu] ^ (r >> 8);
    r = hi > lo ? crc_mul(r, crc_x8n((uint32_t)(n - hi), s_x2n)) : 0u;
    if (lane == 0) r ^= crc_mul(0xFFFFFFFFu, crc_x8n((uint32_t)n, s_x2n));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) r ^= (uint32_t)__shfl_xor((int)r, d, 64);
    return uni(r) ^ 0xFFFFFFFFu;
}

// One BGZF block by one wave.  Every branch in here is uniform; false = the stream is not a valid DEFLATE stream of usize bytes.
__device__ __forceinline__ bool inflate_block(const uint8_t* comp, int64_t comp_left, int csize, uint8_t* out, int usize,
                                              uint16_t* sym_ll, uint16_t* sym_d, uint8_t* lens, uint32_t* lut2, uint32_t* dlut, uint32_t* obuf) {
    const int lane = threadIdx.x & 63;
    uint32_t lbase, lext, dbase, dext;
    length_dist_tables(lbase, lext, dbase, dext);
    Bits b;
    // The reader sees this block's payload and a few bytes of slack, nothing behind them: bits past it read as zero, so a
    // crafted payload of endless empty non-final blocks runs into an invalid block within a few hundred bits instead of decoding
    // on through the rest of the compressed window (zlib refuses such input in linear time as well).
    if (comp_left > (int64_t)csize + 8) comp_left = (int64_t)csize + 8;
    b.start(comp, comp_left);
    int64_t payload_off = 0;   // bytes of the payload in front of b.base (the reader is re-based behind every stored block)
    // st.pos bytes are stored, st.ns literals are decoded and not yet stored: lane k of mylit holds the k-th
    SymState st{0, 0, 0, 0, 0u};
    uint32_t mylit = 0;
    auto flush = [&]() {     // (callers have checked pos + ns <= usize)
        if (lane < st.ns) out[st.pos + lane] = (uint8_t)mylit;
        st.pos += st.ns; st.ns = 0;
    };
    auto copy = [&](int len, int dist) {   // out[pos + i] = out[pos - dist + i]; with dist < len the source repeats with period dist: only bytes in front of pos are read
        for (int i = lane; i < len; i += 64) out[st.pos + i] = out[st.pos - dist + (dist >= len ? i : i % dist)];
        st.pos += len;
    };
    for (bool last = false; !last;) {
        b.refill();
        last = b.take(1) != 0;
        const uint32_t type = b.take(2);
        if (type == 3) return false;
        if (type == 0) {   // stored: to the next byte edge, LEN, ~LEN, LEN bytes
            b.take(b.cnt & 7);
            b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ nlen) != 0xFFFFu || st.pos + st.ns + (int)len > usize) return false;
            flush();
            const int64_t src = b.byte_pos();
            if (payload_off + src + (int64_t)len > (int64_t)csize) return false;   // the bytes must lie inside this block's payload
            for (int i = lane; i < (int)len; i += 64) out[st.pos + i] = b.base[src + i];
            st.pos += (int)len;
            payload_off += src + len;
            b.start(b.base + src + len, b.limit - (src + len));
            continue;
        }
        Code ll, dd;
        if (type == 1) {   // fixed code (RFC 1951 3.2.6)
            for (int s = lane; s < 288; s += 64) lens[s] = s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8));
            if (lane < 32) lens[288 + lane] = 5;
            __builtin_amdgcn_wave_barrier();
            if (!build_code(lens, 288, sym_ll, ll) || !build_code(lens + 288, 30, sym_d, dd)) return false;
            build_lut2(ll, sym_ll, lens, lut2);
            build_dlut64(dd, sym_d, lens + 288, dlut);
        } else {           // dynamic code (3.2.7)
            const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return false;
            if (lane < 19) lens[lane] = 0;
            for (int i = 0; i < hclen; ++i) {
                b.refill();
                const uint32_t v = b.take(3);
                if (lane == 0) lens[CL_ORDER[i]] = (uint8_t)v;
            }
            __builtin_amdgcn_wave_barrier();
            Code cl;
            if (!build_code(lens, 19, sym_d, cl)) return false;   // (the distance table is free until its own build)
            int n = 0;
            uint32_t prev = 0;
            const int total = hlit + hdist;
            while (n < total) {
                b.refill();
                const int s = decode_sym(b, cl, sym_d);
                if (s < 0) return false;
                uint32_t val = 0; int rep = 1;
                if (s < 16) { val = (uint32_t)s; prev = val; }
                else if (s == 16) { if (n == 0) return false; val = prev; rep = 3 + (int)b.take(2); }
                else if (s == 17) { rep = 3 + (int)b.take(3); prev = 0; }
                else { rep = 11 + (int)b.take(7); prev = 0; }
                if (n + rep > total) return false;
                // lens of the two codes back to back at 32 (behind the 19 of the code length code, which is still in use)
                for (int i = lane; i < rep; i += 64) lens[32 + n + i] = (uint8_t)val;
                n += rep;
            }
            __builtin_amdgcn_wave_barrier();
            if (lens[32 + 256] == 0) return false;   // no end-of-block code
            if (!build_code(lens + 32, hlit, sym_ll, ll) || !build_code(lens + 32 + hlit, hdist, sym_d, dd)) return false;
            build_lut2(ll, sym_ll, lens + 32, lut2);
            build_dlut64(dd, sym_d, lens + 32 + hlit, dlut);
        }
        // the symbols: sym_run does the common cases without leaving its asm block; what it hands back is rare
        for (;;) {
            if (st.pos + st.ns > usize) return false;
            flush();   // (sym_run_ob knows no pending literals)
            const int why = sym_run_ob(b, lut2, dlut, obuf, out, usize, st);
            if (why != 4) st.dist = 0;   // (a distance handed in was used; 4 hands the match back: len, dist)
            if (why == 2) { b.refill(); continue; }              // (with st.len set it resumes in the distance half)
            if (why == 5) return false;
            if (why == 4) { copy(st.len, st.dist); st.len = 0; st.dist = 0; continue; }
            if (why == 3) {   // a distance code the direct table does not hold (longer than 8 bits, or none): the lane method judges it
                const int ds = decode_sym(b, dd, sym_d);
                if (ds < 0 || ds > 29) return false;
                st.dist = (int)(rdlane(dbase, ds) + b.take((int)rdlane(dext, ds)));   // sym_run_ob goes on with the copy (and its checks)
                continue;
            }
            // why == 0: end of block, a literal / length code longer than the table's index, or no code at all;
            // why == 6: fewer than two bytes of the block are left -- one symbol the long way
            const uint32_t e = why == 6 ? LUT_LONG : st.e & 0xFFFFu;
            int s;
            if (e != LUT_LONG) { s = (int)(e & 0xFFFu); const int l = (int)(e >> 12); b.buf >>= l; b.cnt -= l; }
            else { s = decode_sym(b, ll, sym_ll); if (s < 0) return false; }
            if (s < 256) {   // a literal with a long code
                mylit = wrlane((uint32_t)s, st.ns, mylit);
                if (++st.ns >= 63) {
                    if (st.pos + st.ns > usize) return false;
                    flush();
                }
                continue;
            }
            if (s == 256) break;
            if (s > 285) return false;
            st.len = (int)(rdlane(lbase, s - 257) + b.take((int)rdlane(lext, s - 257)));   // sym_run goes on with its distance
        }
    }
    if (st.pos + st.ns > usize) return false;
    flush();
    return st.pos == usize;
}

static __global__ __launch_bounds__(BLOCK) void k_bgzf_inflate(Args a) {
    __shared__ uint16_t s_ll[WAVES][288 + 32];   // sorted literal/length symbols, then the distance symbols
    __shared__ uint32_t s_ob[WAVES][OB_SLOTS];   // sym_run_ob's output buffer; between its calls (it leaves it empty) the code lengths of a block header
    static_assert(OB_FLUSH == 128 && OB_SLOTS >= OB_FLUSH + 64 && OB_SLOTS * 4 >= 320 + 64, "the code lengths share the output buffer");
    __shared__ uint32_t s_lut[WAVES][1 << LUT_BITS];
    __shared__ uint32_t s_dlut[WAVES][2 << DLUT_BITS];   // (64-bit entries: build_dlut64)
    __shared__ uint32_t s_crc_tab[256], s_x2n[32];
    crc_tables(s_crc_tab, s_x2n);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t bi = xcd_tile() * WAVES + wave;   // (neighbouring blocks on one XCD: bzq_device.hpp)
    if (bi >= a.n_blocks) return;
    const DevBlock blk = a.blocks[bi];
    bool ok = inflate_block(a.comp + blk.coff, (int64_t)(a.comp_bytes - blk.coff), (int)blk.csize, a.out + blk.uoff, (int)blk.usize,
                            s_ll[wave], s_ll[wave] + 288, reinterpret_cast<uint8_t*>(s_ob[wave]), s_lut[wave], s_dlut[wave], s_ob[wave]);
    // (the wave reads back what it stored itself: same L1, program order)
    if (ok) ok = block_crc32(a.out + blk.uoff, (int)blk.usize, s_crc_tab, s_x2n) == blk.crc;
    if (!ok && (threadIdx.x & 63) == 0) atomicMin(a.first_bad, (unsigned long long)bi);
}

} // namespace inf
} // namespace bzq
