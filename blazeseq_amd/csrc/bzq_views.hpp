// Views mode: the device analogue of `for view in parser.views()` (parser.mojo:253-258, _FastqParserViewIter 628-661;
// FastqView record.mojo:431-550): no column packing -- every record is delivered as its RecordOffsets
// (utils.mojo:37-93: header_start, seq_start, sep_start, qual_start, record_end) plus the stripped-id span, all as
// offsets into the chunk, which stays where it is (zero copy, like the reference's spans into its buffer).
// Algorithmic bytes per record: B read + 5 x 8 + 8 + 4 = B + 52 written (SURVEY.md 8d "offsets-only mode").
//
//   k_tile_count     pass A': newlines per tile, nothing else (no line classes are needed: no column offsets exist)
//   k_views          pass B': per line the offsets of its record, '@' / '+' checks, id span, optional validation of the
//                    bytes by role -- straight from the tile, nothing is moved
//   k_views_check    per record: sequence / quality length check (utils.mojo:458-461) from the offsets, the
//                    buffer-capacity refusal, and the chunk totals the host reads
#pragma once
#include "bzq_chain.hpp"
#include "bzq_fused.hpp"

namespace bzq {

__device__ inline void id_span(const ByteSrc& b, int64_t ls, int64_t le, int64_t& lo, int64_t& hi);

struct ViewArgs {
    const uint8_t* g;
    int64_t n;
    uint32_t prev_byte;
    int64_t tile_begin, tile_end;
    const int64_t* tileP;
    int64_t* o_hdr;
    int64_t* o_seq;
    int64_t* o_sep;
    int64_t* o_qual;
    int64_t* rec_end;
    int64_t* id_start;
    int32_t* id_len;
    int64_t rec_cap;
    ChunkState* st;
    uint32_t q_lower, q_upper;
    int32_t force_dense;
};

// ---- line entries: what pass A leaves behind for every newline, so that pass B never looks at the bytes again ----------
// bits 0-13  tile offset of the newline
// bit  14/15 the byte after it (the first byte of the next line) is '@' / '+'
// bits 16-22 number of POSIX-space bytes that follow that first byte (the next line's leading id spaces), saturating at 127
// bit  23    validation: a byte >= 0x80 between the previous newline OF THIS TILE (or the tile's first byte) and this one
// bits 24-30 number of POSIX-space bytes just before the newline (this line's trailing id spaces), saturating at 127
// bit  31    validation: a byte outside [q_lower, q_upper] in the same stretch
// (validation also keeps one byte per tile, tile_vf: the two flags of the stretch AFTER the tile's last newline -- the
// whole tile when it has none -- so that a line is judged from the entry of its newline, the tail byte of the tile its
// previous newline sits in, and the bytes of the newline-free tiles in between)
constexpr uint32_t ENT_SAT = 127u;
constexpr int ENT_STRIDE = 1024;   // entries per ordinary tile; slot 1023 of tile 0 describes the line that starts the chunk
constexpr int MAXE = 1020;         // a tile with more newlines (records of a few bytes) takes a 16384-entry slot of the pool
// tile_idc of a tile whose entries are NARROW: 16 bits each -- position and the two structure bits, i.e. the low half of an entry -- because
// none of them has anything in its upper half (no id space at a line's edge, no validation).  Ordinary reads: every tile.  Both
// forms end at the same byte of the slot (4 MAXE), so the join's speculative loads do not depend on the form.
constexpr u64 ENT_NARROW = 1ull << 63;

struct LineArgs {
    const uint8_t* g;
    int64_t n;
    uint32_t prev_byte;
    int64_t tile_begin, tile_end;
    uint32_t* tile_c;
    u64* tile_a;
    u64* tile_idc;         // views mode: 0 = entries in the tile's own slot, k + 1 = in pool slot k
    uint32_t* entries;     // [tiles][ENT_STRIDE]
    uint32_t* pool;        // [pool_slots][TILE]
    int64_t pool_slots;
    ViewsPool* pool_state;   // the pool's ticket (never the chunk state: pass A runs before its initial values have arrived)
    int32_t force_dense;   // test switch: every tile through the pool
    uint8_t* tile_vf;      // validation: per tile, bit 0 / 1 = non-ascii / out-of-range byte after the tile's last newline
    uint32_t q_lower, q_upper;
};

// 0x80 in every byte outside [lo, hi] (hi < 128)
__device__ __forceinline__ uint32_t out_of_range_flags(uint32_t x, uint32_t lo, uint32_t hi) {
    const uint32_t t = x & 0x7F7F7F7Fu;
    const uint32_t ge_lo = t + (0x80u - lo) * 0x01010101u, gt_hi = t + (0x7Fu - hi) * 0x01010101u;
    return (~ge_lo | gt_hi | x) & 0x80808080u;
}

// POSIX space other than '\n' (the byte walks of _strip_spaces stop at the line's own newlines; utils.mojo:221-242, 267-289)
__device__ __forceinline__ bool is_space_not_nl(uint32_t c) { return c <= 32u && ((0x170003A00ull >> c) & 1ull); }

// Piece (16 bytes) of load round s of this thread, WAVE-contiguous: wave w fetches bytes [4096 w, 4096 (w + 1)) of the tile -- the
// same bytes its lanes own as 64-byte stretches once the 16-bit masks have gone through LDS.  The transposition is therefore
// wave-local (a wave's DS operations execute in order: no workgroup barrier between its writes and its reads), and the wave's
// newline count is known before the tile's one barrier.
__device__ __forceinline__ int wave_piece(int s) { return (int)(threadIdx.x & ~63u) * 4 + s * 64 + (int)(threadIdx.x & 63u); }

template <bool NT>
__device__ __forceinline__ void tile_fetch_w(const uint8_t* __restrict__ g, int64_t n, int64_t t0, int valid, uint4 (&r)[4]) {
    if (valid == TILE) {   // every tile but the last: no guards
        const uint8_t* __restrict__ p = g + t0 + wave_piece(0) * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) r[s] = load16_any<NT>(p + 1024 * s);
        return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pos = wave_piece(s) * 16;
        r[s] = make_uint4(0u, 0u, 0u, 0u);
        if (pos < valid) r[s] = load16(g, t0 + pos, n);
    }
}

// Pass A of the metadata pipeline: the ONLY kernel that reads the input.  A workgroup walks LINES_TPW consecutive tiles; every
// thread owns the newlines of its 64 bytes.  The tile sits in LDS between a 16-byte halo on either side (the bytes before and
// after it; before the chunk: prev_byte behind newlines, after its end: zeros), so that everything an entry says about the
// bytes around a newline -- the byte before it, the two after it -- is one unguarded pair of LDS dwords; only a newline that
// does touch an id space walks on, and a run that leaves the halo is reported saturated (the join then reads it from the
// input).  The next tile's loads are issued as soon as this tile's registers have gone to LDS: they fly while the entries are
// written, so a workgroup's slot on the CU never sits without a request in the memory system.
#ifndef BZQ_LINES_TPW
#define BZQ_LINES_TPW 1
#endif
constexpr int LINES_TPW = BZQ_LINES_TPW;

template <bool VAL>
static __global__ __launch_bounds__(BLOCK) void k_tile_lines(LineArgs a) {
    constexpr int HALO = 16;
    __shared__ __attribute__((aligned(16))) uint8_t s_tile_raw[HALO + TILE + 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];
    __shared__ __attribute__((aligned(16))) uint16_t s_hi[VAL ? PIECES : 4], s_out[VAL ? PIECES : 4];
    __shared__ uint32_t s_chain[2][BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_w[BLOCK / 64];
    __shared__ int64_t s_slot;
    __shared__ uint32_t s_ent[ENT_STRIDE];
    __shared__ int s_wide;
    uint8_t* s_tile = s_tile_raw + HALO;
    const uint32_t* raw32 = reinterpret_cast<const uint32_t*>(s_tile_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t t = a.tile_begin + xcd_tile() * LINES_TPW;   // (neighbouring runs of tiles on one XCD: bzq_device.hpp)
    if (t >= a.tile_end) return;
    const int64_t t_end = t + LINES_TPW < a.tile_end ? t + LINES_TPW : a.tile_end;
    uint4 r[4], halo = make_uint4(0u, 0u, 0u, 0u);
    // tile tt into the registers: its 16 KiB, and on two lanes the 16 bytes before / after it
    auto fetch = [&](int64_t tt) {
        const int64_t t0 = tt * TILE;
        const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
        tile_fetch_w<true>(a.g, a.n, t0, valid, r);
        if (tid == 0) {
            halo = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x000A0A0Au | (a.prev_byte << 24));
            if (t0 > 0) halo = load16_any(a.g + t0 - HALO);
        } else if (tid == 64) {   // (zeros at and beyond the end of the data)
            halo = make_uint4(0u, 0u, 0u, 0u);
            if (valid == TILE) halo = load16(a.g, t0 + TILE, a.n);
        }
    };
    // what the bytes around the newline at tile offset pos (-1: the one before the tile) say: flags and id space runs of an entry
    auto around = [&](int pos) -> uint32_t {
        const int o = pos + HALO - 1;   // raw offset of the byte before the newline
        const uint32_t d0 = raw32[o >> 2], d1 = raw32[(o >> 2) + 1];
        const uint32_t w4 = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)o & 3u);   // bytes pos - 1, pos, pos + 1, pos + 2
        const uint32_t bp = w4 & 0xFFu, c0 = (w4 >> 16) & 0xFFu, c1 = w4 >> 24;
        uint32_t e = (c0 == 64u ? 1u << 14 : 0u) | (c0 == 43u ? 1u << 15 : 0u);
        if (is_space_not_nl(bp) | is_space_not_nl(c1)) {   // an id that _strip_spaces will shorten (or a CRLF file): walk the runs
            uint32_t trail = 0, lead = 0;
            int p = pos - 1;
            while (p >= -HALO && trail < ENT_SAT && is_space_not_nl(s_tile[p])) { ++trail; --p; }
            if (p < -HALO) trail = ENT_SAT;
            p = pos + 2;
            while (p < TILE + HALO && lead < ENT_SAT && is_space_not_nl(s_tile[p])) { ++lead; ++p; }
            if (p >= TILE + HALO) lead = ENT_SAT;
            e |= (lead << 16) | (trail << 24);
        }
        return e;
    };
    fetch(t);
    for (;;) {
        const int64_t t0 = t * TILE;
        const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
        if (tid == 0) { *reinterpret_cast<uint4*>(s_tile_raw) = halo; s_wide = VAL ? 1 : 0; }
        else if (tid == 64) *reinterpret_cast<uint4*>(s_tile + TILE) = halo;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int q = wave_piece(s), pos = q * 16;
            uint32_t m = nl_mask16(r[s]);
            uint32_t keep = 0xFFFFu;
            if (valid != TILE) {   // last tile: bytes at or beyond the end are not newlines
                const int rem = valid - pos;
                keep = rem >= 16 ? 0xFFFFu : (rem > 0 ? ((1u << rem) - 1u) : 0u);
                m &= keep;
            }
            s_mask[q] = (uint16_t)m;
            *reinterpret_cast<uint4*>(s_tile + pos) = r[s];
            if (VAL) {
                const uint32_t h = fa::flag_mask16(r[s].x & 0x80808080u, r[s].y & 0x80808080u, r[s].z & 0x80808080u, r[s].w & 0x80808080u);
                const uint32_t o = fa::flag_mask16(out_of_range_flags(r[s].x, a.q_lower, a.q_upper), out_of_range_flags(r[s].y, a.q_lower, a.q_upper),
                                                   out_of_range_flags(r[s].z, a.q_lower, a.q_upper), out_of_range_flags(r[s].w, a.q_lower, a.q_upper));
                s_hi[q] = (uint16_t)(h & keep); s_out[q] = (uint16_t)(o & keep);
            }
        }
        const bool more = t + 1 < t_end;
        if (more) fetch(t + 1);   // (the registers are free again)
        // wave-local transposition: the four pieces of this lane's 64 bytes were written by lanes of this wave
        // (acquire AND release: the reads below must not move above it either -- a wave's DS operations execute in order, so this is a
        // statement to the compiler, not a wait)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u64 m64 = reinterpret_cast<const u64*>(s_mask)[tid];
        const uint32_t cnt = (uint32_t)__popcll(m64);
        const uint32_t incl = dpp_scan_u32(cnt);
        if (lane == 63) s_w[wave] = incl;
        // validation: "such a byte since the last newline of this tile", read just before every newline
        u64 hi64 = 0, out64 = 0;
        fa::WaveChain wh{}, wo{};
        if (VAL) {
            hi64 = reinterpret_cast<const u64*>(s_hi)[tid]; out64 = reinterpret_cast<const u64*>(s_out)[tid] & ~m64;
            wh = fa::chain_wave<false>(hi64, m64, s_chain[0]);
            wo = fa::chain_wave<false>(out64, m64, s_chain[1]);
        }
        __syncthreads();   // the other waves' bytes, counts and chain codes, the halo
        const uint4 wc = *reinterpret_cast<const uint4*>(s_w);
        const uint32_t c = wc.x + wc.y + wc.z + wc.w;
        const uint32_t excl = (wave > 0 ? wc.x : 0u) + (wave > 1 ? wc.y : 0u) + (wave > 2 ? wc.z : 0u) + incl - cnt;
        u64 na_before = 0, or_before = 0;
        if (VAL) {
            const fa::Chain ch = fa::chain64(hi64, m64, fa::chain_cin<false>(wh, s_chain[0], 0u));
            const fa::Chain co = fa::chain64(out64, m64, fa::chain_cin<false>(wo, s_chain[1], 0u));
            na_before = ch.excl; or_before = co.excl;
            if (tid == BLOCK - 1) a.tile_vf[t] = (uint8_t)((ch.incl >> 63) | ((co.incl >> 63) << 1));
        }
        const bool pooled = ((int)c > MAXE) || (a.force_dense & 1);   // (the same for every thread)
        int64_t slot = -1;
        if (pooled) {
            if (tid == 0) {
                int64_t sl = (int64_t)atomicAdd(&a.pool_state->listed, 1ull);
                if (sl >= a.pool_slots) { sl = -2; a.pool_state->fallback = 1; }   // the host repeats the chunk on the byte-level kernels
                s_slot = sl;
            }
            __syncthreads();
            slot = s_slot;
        }
        if (tid == 0) {
            a.tile_c[t] = c; a.tile_a[t] = 0ull;
            if (slot != -1) a.tile_idc[t] = slot >= 0 ? (u64)(slot + 1) : 0ull;   // (an ordinary tile says below which form its entries have)
            if (t == 0) a.entries[ENT_STRIDE - 1] = around(-1) & 0x00FFC000u;   // the line that starts the chunk: its first byte and leading spaces
        }
        if (slot == -1) {
            // ordinary tile: the entries end at the END of the tile's slot (entry j of c at word MAXE - c + j, or at halfword
            // 2 MAXE - c + j in the narrow form) -- the join then finds the tile's LAST entries at fixed addresses without knowing c.
            // They go out through LDS: every wave gathers its own (its newlines are a contiguous run of the tile's), the tile
            // agrees on the form, and every wave stores its run as one contiguous stretch.
            uint32_t* sl = s_ent + (excl - (incl - cnt));   // this wave's run starts at its exclusive base
            u64 m = m64;
            uint32_t k = incl - cnt;                        // index inside the wave's run
            uint32_t upper = 0;
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                const int pos = tid * 64 + bit;
                uint32_t e = (uint32_t)pos | around(pos);
                if (VAL) e |= ((uint32_t)((na_before >> bit) & 1ull) << 23) | ((uint32_t)((or_before >> bit) & 1ull) << 31);
                sl[k] = e;
                upper |= e >> 16;
                ++k;
            }
            if (!VAL && upper) s_wide = 1;
            __syncthreads();
            const bool narrow = s_wide == 0;
            if (tid == 0) a.tile_idc[t] = narrow ? ENT_NARROW : 0ull;
            const uint32_t wbase = excl - (incl - cnt);
            const uint32_t wcnt = wave == 0 ? wc.x : wave == 1 ? wc.y : wave == 2 ? wc.z : wc.w;
            if (narrow) {
                uint16_t* out = reinterpret_cast<uint16_t*>(a.entries + t * ENT_STRIDE) + (2 * MAXE - (int)c) + wbase;
                for (uint32_t j = lane; j < wcnt; j += 64) out[j] = (uint16_t)s_ent[wbase + j];
            } else {
                uint32_t* out = a.entries + t * ENT_STRIDE + (MAXE - (int)c) + wbase;
                for (uint32_t j = lane; j < wcnt; j += 64) out[j] = s_ent[wbase + j];
            }
        } else if (slot >= 0) {   // a tile of tiny records: straight into its pool slot, from the slot's start
            uint32_t* out = a.pool + slot * TILE;
            u64 m = m64;
            uint32_t idx = excl;
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                const int pos = tid * 64 + bit;
                uint32_t e = (uint32_t)pos | around(pos);
                if (VAL) e |= ((uint32_t)((na_before >> bit) & 1ull) << 23) | ((uint32_t)((or_before >> bit) & 1ull) << 31);
                out[idx] = e;
                ++idx;
            }
        }
        if (!more) break;
        ++t;
        __syncthreads();   // everyone is done with this tile's LDS
    }
}

struct JoinArgs {
    const uint8_t* g;      // id space runs that saturated an entry; the bytes behind the chunk's last newline
    uint32_t prev_byte;
    int64_t n;
    int64_t tile_begin, tile_end, n_tiles;
    const uint32_t* tile_c;
    const u64* tile_slot;  // tile_idc
    const int64_t* tileP;
    const uint32_t* entries;
    const uint32_t* pool;
    int64_t* o_hdr;
    int64_t* o_seq;
    int64_t* o_sep;
    int64_t* o_qual;
    int64_t* rec_end;
    int64_t* id_start;
    int32_t* id_len;
    int64_t rec_cap, first_header, len_limit;
    ChunkState* st;
    const uint8_t* tile_vf;   // validation (see the entry layout)
    int32_t check_ascii, check_quality;
};

constexpr int JOIN_TILES = BLOCK / 64;                // tiles per workgroup of the join: one wave stages one tile's entries
constexpr int JOIN_ENT = JOIN_TILES * ENT_STRIDE;     // entries a window holds in LDS (a tile outside the pool has at most MAXE)

// Pass B of the metadata pipeline: one thread per RECORD.  Record r owns the newlines 4r-1 .. 4r+3 (global line index); the five
// entries give every offset, both structure bytes, the id span and the length / buffer checks -- 20 bytes read, 52 written.
// A workgroup takes the records whose LAST newline lies in its JOIN_TILES tiles.  Everything it needs from memory is asked for
// at once, before anything has come back: the window's prefixes (scalar loads), the LAST 256 entries of every tile's slot (one
// 16-byte load per lane, wave w takes tile w: a tile of ordinary reads has about 200; pass A aligns a tile's entries with the end
// of its slot so that these addresses do not depend on the tile's count), and the last four entries of the tile before the
// window (the lines of the record that straddles in).  The entries go to LDS in window order -- consecutive
// newline indices are consecutive words -- and every record reads its five from there.  One memory latency per workgroup; the
// walk through the prefixes (`locate`) remains for what this cannot serve: windows with pool tiles, records that began more
// than one tile before the window (long reads), the chunk's first record.
template <bool VAL>
static __global__ __launch_bounds__(BLOCK) void k_views_join(JoinArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_e[4 + JOIN_ENT];
    __shared__ int64_t s_tail;
    __shared__ int s_nb;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ta = a.tile_begin + xcd_tile() * JOIN_TILES;   // (neighbouring windows on one XCD)
    if (ta >= a.tile_end) return;
    const int64_t tb = ta + JOIN_TILES < a.tile_end ? ta + JOIN_TILES : a.tile_end;
    const int nt = (int)(tb - ta);
    // ---- requests: all of them before the first wait
    uint4 spec = make_uint4(0u, 0u, 0u, 0u);
    if (wave < nt) spec = *reinterpret_cast<const uint4*>(a.entries + (ta + wave) * ENT_STRIDE + (MAXE - 256) + 4 * lane);
    int64_t P[JOIN_TILES + 1];
    u64 slots = 0;
#pragma unroll
    for (int k = 0; k < JOIN_TILES; ++k) {
        P[k] = k < nt ? a.tileP[ta + k] : 0;
        slots |= k < nt ? (a.tile_slot[ta + k] & ~ENT_NARROW) : 0ull;   // (anything but the form bit: a pool tile)
    }
    const int64_t Pend = a.tileP[tb - 1] + (int64_t)a.tile_c[tb - 1];
#pragma unroll
    for (int k = 1; k <= JOIN_TILES; ++k) if (k >= nt) P[k] = Pend;
    const int64_t Pprev = ta > 0 ? a.tileP[ta - 1] : 0;
    uint4 l4 = make_uint4(0u, 0u, 0u, 0u);
    if (ta > 0) l4 = *reinterpret_cast<const uint4*>(a.entries + (ta - 1) * ENT_STRIDE + (MAXE - 4));
    const u64 slot_prev = ta > 0 ? a.tile_slot[ta - 1] : 1ull;
    const int64_t P0 = a.st->P0, lines = a.st->P;             // P0 + all newlines of the chunk
    const int64_t Gbeg = P[0], Gend = Pend;                   // newline indices [Gbeg, Gend) live in this window
    // the four newlines before the window are the last four of tile ta - 1 when that tile has that many
    const bool prev_ok = (slot_prev & ~ENT_NARROW) == 0ull && Gbeg - Pprev >= 4;
    const bool staged = slots == 0ull && Gend - Gbeg <= JOIN_ENT;
    u64 e_struct = ~0ull, e_buf = ~0ull, e_valid = ~0ull;
    bool overflow = false;
    int64_t loc_tile = -1;   // tile of the newline the last find() found (-1: the virtual one before the chunk)

    // ---- the window's entries into LDS, in newline order: s_e[4 + (G - Gbeg)]
    if (staged) {
        if (tid == 0) {   // (the last four entries of the tile before the window: its slot's last 16 bytes hold them in either form)
            if (slot_prev & ENT_NARROW) l4 = make_uint4(l4.z & 0xFFFFu, l4.z >> 16, l4.w & 0xFFFFu, l4.w >> 16);
            *reinterpret_cast<uint4*>(s_e) = l4;
        }
        if (wave < nt) {
            const int64_t Pw = wave == 0 ? P[0] : wave == 1 ? P[1] : wave == 2 ? P[2] : P[3];
            const int64_t Pn = wave == 0 ? P[1] : wave == 1 ? P[2] : wave == 2 ? P[3] : P[4];
            const int cw = (int)(Pn - Pw);
            uint32_t* dst = s_e + 4 + (int)(Pw - Gbeg);
            const u64 form = a.tile_slot[ta + wave];   // (scalar; it came back with the prefixes)
            if (form & ENT_NARROW) {   // 16 bits per entry: the speculative load took entries cw - 512 .. cw - 1, eight per lane
                const int j = cw - 512 + 8 * lane;
                const uint32_t w4[4] = {spec.x, spec.y, spec.z, spec.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (j + 2 * u >= 0) dst[j + 2 * u] = w4[u] & 0xFFFFu;
                    if (j + 2 * u + 1 >= 0) dst[j + 2 * u + 1] = w4[u] >> 16;
                }
                const uint16_t* src = reinterpret_cast<const uint16_t*>(a.entries + (ta + wave) * ENT_STRIDE) + (2 * MAXE - cw);
                for (int jj = lane; jj < cw - 512; jj += 64) dst[jj] = src[jj];   // (a tile of short records)
            } else {
                const int j = cw - 256 + 4 * lane;   // the speculative load took entries cw - 256 .. cw - 1
                if (j >= 0) dst[j] = spec.x;
                if (j + 1 >= 0) dst[j + 1] = spec.y;
                if (j + 2 >= 0) dst[j + 2] = spec.z;
                if (j + 3 >= 0) dst[j + 3] = spec.w;
                const uint32_t* src = a.entries + (ta + wave) * ENT_STRIDE + (MAXE - cw);
                for (int jj = lane; jj < cw - 256; jj += 64) dst[jj] = src[jj];   // (a tile of short records)
            }
        }
        __syncthreads();
    }
    const int B1 = (int)(P[1] - Gbeg), B2 = (int)(P[2] - Gbeg), B3 = (int)(P[3] - Gbeg);

    // entry and absolute position of newline G by the prefixes (P0 - 1 = the virtual newline before the chunk's first line)
    auto locate = [&](int64_t G, uint32_t& e) -> int64_t {
        if (G < P0) { e = a.entries[ENT_STRIDE - 1]; loc_tile = -1; return -1; }
        int64_t tt = tb - 1;
        while (a.tileP[tt] > G) --tt;
        const int64_t j = G - a.tileP[tt];
        const u64 slot = a.tile_slot[tt];
        if (slot & ENT_NARROW) e = reinterpret_cast<const uint16_t*>(a.entries + tt * ENT_STRIDE)[2 * MAXE - (int64_t)a.tile_c[tt] + j];
        else e = slot ? a.pool[(int64_t)(slot - 1) * TILE + j] : a.entries[tt * ENT_STRIDE + (MAXE - (int64_t)a.tile_c[tt]) + j];
        loc_tile = tt;
        return tt * TILE + (int64_t)(e & 0x3FFFu);
    };
    auto find = [&](int64_t G, uint32_t& e) -> int64_t {
        const int k = (int)(G - Gbeg);
        if (staged && G >= P0 && (k >= 0 || (prev_ok && k >= -4))) {
            e = s_e[4 + k];
            const int w = k < 0 ? -1 : (k >= B1) + (k >= B2) + (k >= B3);
            loc_tile = ta + w;
            return loc_tile * TILE + (int64_t)(e & 0x3FFFu);
        }
        return locate(G, e);
    };
    // validation flags (bit 0 non-ascii, bit 1 out of range) of the line between two consecutive newlines: the entry of
    // the closing one covers its own tile; the line's earlier tiles are the tail of the opening newline's tile and
    // newline-free tiles
    auto line_flags = [&](int64_t tile_open, int64_t tile_close, uint32_t e_close) -> uint32_t {
        uint32_t f = ((e_close >> 23) & 1u) | (((e_close >> 31) & 1u) << 1);
        for (int64_t t = tile_open < 0 ? 0 : tile_open; t < tile_close; ++t) f |= a.tile_vf[t];
        return f;
    };
    auto id_of = [&](int64_t hs, uint32_t e_hs, int64_t nl0, uint32_t e_nl0, int64_t& lo, int64_t& hi) {
        const uint32_t lead = (e_hs >> 16) & 0x7Fu, trail = (e_nl0 >> 24) & 0x7Fu;
        lo = hs + 1 + (int64_t)lead; hi = nl0 - (int64_t)trail;
        if (lead == ENT_SAT || trail == ENT_SAT) {           // a space run too long for an entry: from the bytes
            ByteSrc bs{a.g, a.n, a.prev_byte, nullptr, 0, 0};
            id_span(bs, hs, nl0, lo, hi);
        }
        if (hi < lo) hi = lo;                                // nothing but spaces: the two runs overlap
    };

    // complete records whose quality newline 4r+3 is in [Gbeg, Gend)
    const int64_t r_lo = Gbeg <= 3 ? 0 : (Gbeg - 3 + 3) >> 2;      // ceil((Gbeg - 3) / 4), >= 0
    const int64_t r_hi = (Gend - 4) >> 2;                           // floor((Gend - 4) / 4), may be < r_lo
    for (int64_t r = r_lo + tid; r <= r_hi; r += BLOCK) {
        if (r >= a.rec_cap) { overflow = true; continue; }
        const int64_t G3 = 4 * r + 3;
        uint32_t e3, e2, e1, e0, ep;
        const int64_t p3 = find(G3, e3); const int64_t T3 = loc_tile;
        const int64_t p2 = find(G3 - 1, e2); const int64_t T2 = loc_tile;
        const int64_t p1 = find(G3 - 2, e1); const int64_t T1 = loc_tile;
        const int64_t p0 = find(G3 - 3, e0); const int64_t T0 = loc_tile;
        const int64_t pp = find(G3 - 4, ep); const int64_t TP = loc_tile;
        const int64_t hs = pp + 1;
        a.o_hdr[r] = hs; a.o_seq[r] = p0 + 1; a.o_sep[r] = p1 + 1; a.o_qual[r] = p2 + 1; a.rec_end[r] = p3;
        int64_t lo = hs + 1 < p0 ? hs + 1 : p0, hi = lo;
        // (a record beyond the buffer limit is refused below; its id is not worth a walk over megabytes of spaces)
        if (p3 - (r ? pp : a.first_header - 1) <= a.len_limit) id_of(hs, ep, p0, e0, lo, hi);
        a.id_start[r] = lo; a.id_len[r] = (int32_t)(hi - lo);
        // utils.mojo:448-462 in the reference's order: '@', '+', lengths
        u64 k = ~0ull;
        if (!(ep & (1u << 14))) k = ((u64)r << 3) | 1ull;
        else if (!(e1 & (1u << 15))) k = ((u64)r << 3) | 2ull;
        else if ((p1 - p0 - 1) != (p3 - p2 - 1)) k = ((u64)r << 3) | 3ull;
        e_struct = k < e_struct ? k : e_struct;
        if (VAL) {   // Validator._validate (record.mojo:162-172): ascii over id, sequence, quality, then the quality range
            const uint32_t fh = line_flags(TP, T0, e0), fs = line_flags(T0, T1, e1), fq = line_flags(T2, T3, e3);
            u64 kv = ~0ull;
            if (a.check_ascii && ((fh | fs | fq) & 1u)) kv = ((u64)r << 3) | 4ull;
            else if (a.check_quality && (fq & 2u)) kv = ((u64)r << 3) | 5ull;
            e_valid = kv < e_valid ? kv : e_valid;
        }
        (void)T1; (void)T2; (void)T3; (void)T0; (void)TP;
        const int64_t prev_end = r ? pp : a.first_header - 1;
        if (p3 - prev_end > a.len_limit) { const u64 kb = (u64)r << 3; e_buf = kb < e_buf ? kb : e_buf; }
    }
    // the last workgroup: chunk totals, and the lines of the record that stays incomplete (an unterminated last record
    // may still be delivered, parser.mojo:464-475)
    if (tb == a.tile_end && tid == 0) {
        int64_t n_rec = lines > 0 ? (lines >> 2) : 0;
        if (n_rec > a.rec_cap) n_rec = a.rec_cap;
        uint32_t e;
        a.st->n_complete = n_rec;
        a.st->last_record_end = n_rec ? find(4 * n_rec - 1, e) : a.first_header - 1;
        a.st->last_ends = 0; a.st->last_id_ends = 0;
        const int64_t r = lines > 0 ? (lines >> 2) : 0;
        const int k = lines > 0 ? (int)(lines & 3) : 0;      // newlines of the incomplete record
        if (r < a.rec_cap && lines >= 0) {
            uint32_t ep, e0, e1, e2;
            const int64_t pp = find(4 * r - 1, ep);
            if (pp + 1 < a.n) a.o_hdr[r] = pp + 1;
            if (k >= 1) {
                const int64_t p0 = find(4 * r, e0);
                if (p0 + 1 < a.n) a.o_seq[r] = p0 + 1;
                int64_t lo, hi;
                id_of(pp + 1, ep, p0, e0, lo, hi);
                a.id_start[r] = lo; a.id_len[r] = (int32_t)(hi - lo);
                if (k >= 2) { const int64_t p1 = find(4 * r + 1, e1); if (p1 + 1 < a.n) a.o_sep[r] = p1 + 1; }
                if (k >= 3) { const int64_t p2 = find(4 * r + 2, e2); if (p2 + 1 < a.n) a.o_qual[r] = p2 + 1; }
                if (VAL && k == 3) {
                    // an unterminated last record may be delivered (parser.mojo:464-475) and is validated like any other:
                    // its quality line is everything after the chunk's last newline
                    uint32_t x;
                    (void)find(4 * r - 1, x); const int64_t tp = loc_tile;
                    (void)find(4 * r, x); const int64_t t0l = loc_tile;
                    (void)find(4 * r + 1, x); const int64_t t1l = loc_tile;
                    (void)find(4 * r + 2, x); const int64_t t2l = loc_tile;
                    const uint32_t fh = line_flags(tp, t0l, e0), fs = line_flags(t0l, t1l, e1);
                    uint32_t fq = 0;
                    for (int64_t t = t2l < 0 ? 0 : t2l; t < a.n_tiles; ++t) fq |= a.tile_vf[t];
                    u64 kv = ~0ull;
                    if (a.check_ascii && ((fh | fs | fq) & 1u)) kv = ((u64)r << 3) | 4ull;
                    else if (a.check_quality && (fq & 2u)) kv = ((u64)r << 3) | 5ull;
                    e_valid = kv < e_valid ? kv : e_valid;
                }
            }
        } else if (r >= a.rec_cap) overflow = true;
    }
    if (e_struct != ~0ull) atomicMin(&a.st->err_struct, e_struct);
    if (e_buf != ~0ull) atomicMin(&a.st->err_buf, e_buf);
    if (VAL && e_valid != ~0ull) atomicMin(&a.st->err_valid, e_valid);
    if (overflow) atomicOr(&a.st->rec_overflow, 1);
    // The chunk's last workgroup also looks behind the last newline: where the tail starts and whether it is more than blanks
    // (_check_end_qual, utils.mojo:292-329) -- what k_tail does for the other modes, here without a launch of its own.
    if (tb == a.n_tiles) {
        if (tid == 0) {
            uint32_t e;
            s_tail = lines > P0 ? find(lines - 1, e) + 1 : 0;
            s_nb = 0;
        }
        __syncthreads();
        const int64_t tail = s_tail;
        int nb = 0;
        for (int64_t p = tail + tid; p < a.n; p += BLOCK) {
            const uint32_t ch = a.g[p];
            if (ch != 10u && ch != 13u && ch != 32u && ch != 9u) { nb = 1; break; }
        }
        if (nb) atomicOr(&s_nb, 1);
        __syncthreads();
        if (tid == 0) { a.st->tail_start = tail; a.st->tail_nonblank = s_nb; }
    }
}

static __global__ __launch_bounds__(BLOCK) void k_tile_count(AggArgs a) {
    __shared__ uint32_t s_w[BLOCK / 64];
    const int tid = threadIdx.x;
    const int64_t t = a.tile_begin + xcd_tile();   // (neighbouring tiles on one XCD: bzq_device.hpp)
    if (t >= a.tile_end) return;
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    uint4 r[4];
    tile_fetch<true>(a.g, a.n, t0, valid, r);
    uint32_t cnt = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pos = (tid + BLOCK * s) * 16;
        uint32_t m = nl_mask16(r[s]);
        if (valid != TILE) {
            const int rem = valid - pos;
            if (rem < 16) m &= rem > 0 ? ((1u << rem) - 1u) : 0u;
        }
        cnt += (uint32_t)__popc(m);
    }
    const u64 w = wave_sum_u64((u64)cnt);
    if ((tid & 63) == 0) s_w[tid >> 6] = (uint32_t)w;
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0;
        for (int i = 0; i < BLOCK / 64; ++i) c += s_w[i];
        a.tile_c[t] = c; a.tile_a[t] = 0ull; a.tile_idc[t] = 0ull;
    }
}

// Stripped id of the header line [ls, le) (ls = the '@', le = its newline), as a span of the chunk
// (_strip_spaces on [header_start + 1, seq_start - 1), parser.mojo:355-366; all-space ids collapse to empty).
__device__ inline void id_span(const ByteSrc& b, int64_t ls, int64_t le, int64_t& lo, int64_t& hi) {
    lo = ls + 1 < le ? ls + 1 : le;
    hi = le;
    while (lo < hi && is_posix_space(b.at(lo))) ++lo;
    while (hi > lo && is_posix_space(b.at(hi - 1))) --hi;
}

// ascii / quality check of the tile bytes [a, b) (tile offsets) that belong to one line role
template <bool CA, bool CQ>
__device__ __forceinline__ void check_bytes(const uint8_t* s_tile, int a, int b, bool is_qual, int64_t rec, uint32_t qlo, uint32_t qhi,
                                            ErrAcc& err) {
    for (int p = a & ~15; p < b; p += 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(s_tile + p);
        const int x0 = a > p ? a - p : 0, x1 = b - p < 16 ? b - p : 16;
        const uint32_t m0 = byte_range_mask(0, x0, x1), m1 = byte_range_mask(1, x0, x1), m2 = byte_range_mask(2, x0, x1),
                       m3 = byte_range_mask(3, x0, x1);
        if (CA && any_non_ascii((v.x & m0) | (v.y & m1) | (v.z & m2) | (v.w & m3))) err.valid(rec, 4);
        if (CQ && is_qual) {
            const uint32_t fill = 0x01010101u * qlo;
            if (any_out_of_range((v.x & m0) | (fill & ~m0), qlo, qhi) | any_out_of_range((v.y & m1) | (fill & ~m1), qlo, qhi) |
                any_out_of_range((v.z & m2) | (fill & ~m2), qlo, qhi) | any_out_of_range((v.w & m3) | (fill & ~m3), qlo, qhi))
                err.valid(rec, 5);
        }
    }
}

constexpr int MAXL_V = 980;   // more newlines in a tile -> the serial path

template <bool CA, bool CQ>
static __global__ __launch_bounds__(BLOCK) void k_views(ViewArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile_raw[16 + TILE + 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];
    __shared__ uint16_t s_nl[MAXL_V];   // sized so that the kernel's LDS is 8 x 20480 B per CU
    __shared__ uint32_t s_w[BLOCK / 64];
    // validation only: line index at the first byte of every 16-byte piece
    __shared__ uint16_t s_pline[(CA || CQ) ? PIECES : 1];
    uint8_t* s_tile = s_tile_raw + 16;
    const int tid = threadIdx.x;
    const int64_t t = a.tile_begin + xcd_tile();   // (neighbouring tiles on one XCD: bzq_device.hpp)
    if (t >= a.tile_end) return;
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    const int64_t P = a.tileP[t];
    const uint32_t prevb = t0 > 0 ? (uint32_t)a.g[t0 - 1] : a.prev_byte;
    uint4 r[4];
    tile_fetch(a.g, a.n, t0, valid, r);
    tile_stage<true>(r, valid, s_mask, s_tile);
    ByteSrc bs{a.g, a.n, a.prev_byte, s_tile, t0, valid};
    const bool first_starts = prevb == 10u;
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(s_mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, BLOCK / 64>((uint32_t)__popcll(m64), s_w, c);
    const bool dense = ((int)c > MAXL_V) || a.force_dense;
    ErrAcc err{~0ull, ~0ull};
    bool overflow = false;
    if ((CA || CQ) && !dense) {
        const uint32_t l0 = excl, l1 = l0 + (uint32_t)__popc((uint32_t)m64 & 0xFFFFu),
                       l2 = l0 + (uint32_t)__popc((uint32_t)m64), l3 = l0 + (uint32_t)__popcll(m64 & 0xFFFFFFFFFFFFull);
        *reinterpret_cast<u64*>(&s_pline[4 * tid]) = (u64)l0 | ((u64)l1 << 16) | ((u64)l2 << 32) | ((u64)l3 << 48);
    }

    // one line: offsets of its record, structure bytes, id span, validation.  [start, end) are tile offsets;
    // sknown: the line starts at `start` (false only for a first line that continues from the previous tile)
    auto handle = [&](int j, int start, int end, bool sknown, bool end_in, bool check_here) {
        const int64_t L = P + j;
        if (L < 0) return;                       // a head line owned by the previous shard
        const int role = (int)(L & 3);
        const int64_t rec = L >> 2;
        const bool in_cap = rec < a.rec_cap;
        if (!in_cap) { overflow = true; }
        const bool sin = sknown && start < valid;
        const int64_t ls = t0 + start;
        if (role == 0) {
            if (sin) {
                if (s_tile[start] != 64) err.structure(rec, 1);   // '@', utils.mojo:454
                if (in_cap) a.o_hdr[rec] = ls;
            }
            if (end_in) {   // the header line ends here: its stripped id as a span of the chunk
                int64_t h0 = ls;
                if (!sknown) { h0 = t0; while (bs.at(h0 - 1) != 10u) --h0; }   // started in an earlier tile
                int64_t lo, hi;
                id_span(bs, h0, t0 + end, lo, hi);
                if (in_cap) { a.id_start[rec] = lo; a.id_len[rec] = (int32_t)(hi - lo); }
            }
            if (CA) {   // ascii covers the kept id bytes only: their range within this tile
                int64_t lo = ls, hi = ls;
                if (end > start) header_kept(bs, ls, t0 + end, sknown, end_in, t0 + valid, lo, hi);
                (void)check_here;   // headers are short: their kept bytes are always checked by the line's own thread
                check_bytes<CA, false>(s_tile, (int)(lo - t0), (int)(hi - t0), false, rec, a.q_lower, a.q_upper, err);
            }
        } else if (role == 2) {
            if (sin) {
                if (s_tile[start] != 43) err.structure(rec, 2);   // '+', utils.mojo:456
                if (in_cap) a.o_sep[rec] = ls;
            }
        } else {
            if (sin && in_cap) (role == 1 ? a.o_seq : a.o_qual)[rec] = ls;
            if ((CA || CQ) && end > start) {
                if (check_here) {
                    check_bytes<CA, CQ>(s_tile, start, end, role == 3, rec, a.q_lower, a.q_upper, err);
                } else {
                    // whole pieces without a newline are checked from registers by the bulk pass below; the line's own
                    // thread takes the (at most two) pieces it shares with a newline or with the end of the data
                    const int pa = start >> 4, pb = (end - 1) >> 4;
                    if (s_mask[pa] != 0 || (pa + 1) * 16 > valid) {
                        const int e1 = end < (pa + 1) * 16 ? end : (pa + 1) * 16;
                        check_bytes<CA, CQ>(s_tile, start, e1, role == 3, rec, a.q_lower, a.q_upper, err);
                    }
                    if (pb != pa && (s_mask[pb] != 0 || (pb + 1) * 16 > valid))
                        check_bytes<CA, CQ>(s_tile, pb * 16, end, role == 3, rec, a.q_lower, a.q_upper, err);
                }
            }
            if (role == 3 && end_in && in_cap) a.rec_end[rec] = t0 + end;
        }
    };

    if (dense) {   // tiles of tiny records: one thread walks them
        if (tid == 0) {
            int j = 0, line_start = 0;
            bool sknown = first_starts;
            for (int w = 0; w < BLOCK; ++w) {
                u64 m = s_mask64[w];
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const int nl = w * 64 + bit;
                    handle(j, line_start, nl, sknown, true, true);
                    line_start = nl + 1; sknown = true; ++j;
                }
            }
            handle(j, line_start, valid, sknown, false, true);
            atomicAdd((u64*)&a.st->dense_tiles, 1ull);
        }
    } else {
        u64 m = m64;
        int idx = 0;
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            s_nl[excl + idx] = (uint16_t)(tid * 64 + bit);
            ++idx;
        }
        __syncthreads();
        for (int j = tid; j <= (int)c; j += BLOCK) {
            const int start = j ? (int)s_nl[j - 1] + 1 : 0;
            const bool end_in = j < (int)c;
            const int end = end_in ? (int)s_nl[j] : valid;
            handle(j, start, end, j > 0 ? true : first_starts, end_in, false);
        }
        if (CA || CQ) {
            // bulk validation: every whole 16-byte piece that lies inside ONE sequence / quality line (no newline in it),
            // straight from the registers it was loaded into; pieces that hold a newline were taken by their lines above
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int q = tid + BLOCK * sidx, pos = q * 16;
                if (pos + 16 <= valid && s_mask[q] == 0) {
                    const int64_t L = P + (int64_t)s_pline[q];
                    const int role = (int)(L & 3);
                    if ((role & 1) && L >= 0) {
                        const uint4 v = r[sidx];
                        if (CA && any_non_ascii(v.x | v.y | v.z | v.w)) err.valid(L >> 2, 4);
                        if (CQ && role == 3 &&
                            (any_out_of_range(v.x, a.q_lower, a.q_upper) | any_out_of_range(v.y, a.q_lower, a.q_upper) |
                             any_out_of_range(v.z, a.q_lower, a.q_upper) | any_out_of_range(v.w, a.q_lower, a.q_upper)))
                            err.valid(L >> 2, 5);
                    }
                }
            }
        }
    }
    if (err.e_struct != ~0ull) atomicMin(&a.st->err_struct, err.e_struct);
    if (err.e_valid != ~0ull) atomicMin(&a.st->err_valid, err.e_valid);
    if (overflow) atomicOr(&a.st->rec_overflow, 1);
}

struct ViewCheckArgs {
    const int64_t* o_hdr;
    const int64_t* o_seq;
    const int64_t* o_sep;
    const int64_t* o_qual;
    const int64_t* rec_end;
    int64_t first_header, len_limit, rec_cap;
    ChunkState* st;
    const uint8_t* g;
    int32_t compat_w;
    uint32_t q_upper;
};

// Grid-stride over the complete records (their count comes from the device state, like k_rebase).
static __global__ __launch_bounds__(BLOCK) void k_views_check(ViewCheckArgs a) {
    const int64_t lines = a.st->P;
    int64_t n_rec = lines > 0 ? (lines >> 2) : 0;
    if (n_rec > a.rec_cap) n_rec = a.rec_cap;
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n_rec; r += (int64_t)gridDim.x * BLOCK) {
        const int64_t re = a.rec_end[r];
        const int64_t seq_len = a.o_sep[r] - a.o_seq[r] - 1, qual_len = re - a.o_qual[r];
        if (seq_len != qual_len) atomicMin(&a.st->err_struct, ((u64)r << 3) | 3ull);   // utils.mojo:458-461
        const int64_t prev = r ? a.rec_end[r - 1] : a.first_header - 1;
        if (re - prev > a.len_limit) atomicMin(&a.st->err_buf, (u64)r << 3);
        if (a.compat_w > 0) {   // SIMD-width quirk of the quality check, as in k_rebase
            const int64_t qs = re - qual_len, body = (qual_len / a.compat_w) * a.compat_w;
            for (int64_t i = 0; i < body; ++i)
                if (a.g[qs + i] == a.q_upper) { atomicMin(&a.st->err_valid, ((u64)r << 3) | 5ull); break; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.st->n_complete = n_rec;
        a.st->last_record_end = n_rec ? a.rec_end[n_rec - 1] : a.first_header - 1;
        a.st->last_ends = 0;
        a.st->last_id_ends = 0;
    }
}

} // namespace bzq
