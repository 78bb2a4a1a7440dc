// bzq_fasta.hpp -- gfx950 kernels of the FASTA record path.
//
// What this replaces in the reference (paths relative to the BlazeSeq tree):
//   FastaParser.next_record / _read_header_line   blazeseq/fasta/parser.mojo:122-203
//   LineIterator.next_line                        blazeseq/io/buffered.mojo:600-638
//   _strip_spaces / is_posix_space                blazeseq/utils.mojo:221-289
//   Validator._validate (ascii)                   blazeseq/fasta/parser.mojo:41-45
//
// Formulation.  The reference reads a line, strips it, and either opens a record ('>') or appends the line to the
// open record's sequence.  For a byte that is neither '\n' nor another posix space ("X" below) nothing more is
// needed than four per-line facts, each of them a "last event wins" state along the line:
//     Sb(p)  an X at or before p on this line        (set by X, cleared by '\n')
//     Sa(p)  an X at or after p on this line         (the same, read backwards)
//     hdr(p) the line's first X is '>'               (decided at that first X)
//     X2(p)  a second X at or before p on this line  (set by any X that is not the line's first)
// sequence bytes = ~hdr & Sb & Sa, id bytes = hdr & X2 & Sa; both columns are plain stream compactions of the input
// in input order, and record k's `ends` are the column ranks at the '>' of record k+1.  A "last event wins" state over
// 64 bytes is one 64-bit ADD: with a = ~clear and b = set the carry out of bit p is exactly the state after byte p
// (set: generate, neither: propagate, clear: kill), so each fact costs one add per thread plus a ballot to carry it
// across the 256 threads of a tile -- the same adder, applied to the ballots.
//
//   k_fa_tile_sums   pass 1: per 16 KiB tile, with nothing known about the bytes before or after it: column bytes and
//                    headers, the part of them that sits before the tile's first '\n' under each possible state of the
//                    line that enters the tile, the state the tile leaves, trailing spaces that are kept only if an X
//                    follows in a later tile.  Streaming read, 24 B out per tile.
//   k_fa_resolve     per tile: the entering state (walk back to the nearest tile with a '\n'), whether an X follows
//                    (look at the next tile that is not all spaces), lines of >= line_cap bytes; group sums.
//   k_fa_bases       exclusive scan of the resolved counts -> per tile column offsets and record index.
//   k_fa_emit        pass 2: re-reads the tile into registers + LDS, rebuilds the masks with the true carries, stores
//                    whole 16-byte pieces that lie inside one kept run straight from registers and the first/last 16
//                    bytes of every run from LDS (unaligned on both sides), writes ends / id_ends / '>' offsets per
//                    record, ascii check, "sequence before any header" check.
//   k_fa_finish      totals, the last record's ends at EOF, the carry point of a chunk that is not the last one.
//   k_fa_empty       records without sequence bytes (first one wins).
//
// HBM-bound byte work, no MFMA.  Algorithmic traffic: input once + both columns once (~2 B per input byte); this
// two-read design moves ~3 B.
#pragma once
#include "bzq_chain.hpp"

namespace bzq {
namespace fa {

constexpr unsigned long long NONE = ~0ull;

struct FaState {
    int64_t n_headers, seq_total, id_total, nl_total;   // k_fa_bases
    unsigned long long long_pos;    // smallest start offset of a line of >= line_cap bytes that has its '\n' (NONE)
    unsigned long long long_tile;   // smallest tile whose first '\n' ends a line that began more than line_cap bytes earlier (NONE)
    unsigned long long nohdr_pos;   // smallest offset of a sequence byte that comes before any header
    unsigned long long ascii_rec;   // smallest record with a byte >= 0x80 in id or sequence
    unsigned long long empty_rec;   // smallest closed record without sequence bytes
    int64_t n_closed;               // records closed inside this chunk (all at EOF; otherwise those followed by a header in a complete line)
    int64_t consumed, lines_consumed;
    int64_t last_line_start;        // offset after the last '\n' (0 if none)
    int64_t query[4];               // cold-path answers (k_fa_query)
};

// ---- byte classes -> bit masks ------------------------------------------------------------------------------------
// 0x80 in every byte that is a posix space (utils.mojo:267-289: 9-13, 28-30, 32), exact for all 256 values
__device__ __forceinline__ uint32_t space_flags(uint32_t x) {
    const uint32_t t = x & 0x7F7F7F7Fu;
    const uint32_t ge9 = t + 0x77777777u, ge14 = t + 0x72727272u, ge28 = t + 0x64646464u, ge31 = t + 0x61616161u,
                   ge32 = t + 0x60606060u, ge33 = t + 0x5F5F5F5Fu;
    return ((ge9 & ~ge14) | (ge28 & ~ge31) | (ge32 & ~ge33)) & ~x & 0x80808080u;
}
// 0x80 in every byte <= 32 (bytes >= 0x80 never flagged).  Two instructions (v_sub + v_bfi-like and-not) and conservative: the
// borrow of a byte below 0x21 may also flag a 0x21 right above it -- callers only use it to decide whether the exact
// classification is needed, so a false positive costs time, never a result.
__device__ __forceinline__ uint32_t le32_flags(uint32_t x) { return (x - 0x21212121u) & ~x & 0x80808080u; }
// 16-bit mask of the bytes equal to B (the v_perm/v_dot4 idiom of nl_mask16, bzq_device.hpp)
template <uint32_t B>
__device__ __forceinline__ uint32_t eq_mask16(uint4 v) {
    constexpr uint32_t K = (B ^ 12u) * 0x01010101u;
    auto f = [](uint32_t x) { return (int)__builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, x ^ K); };
    int lo = __builtin_amdgcn_sdot4(f(v.x), 0x08040201, 127, false);
    lo = __builtin_amdgcn_sdot4(f(v.y), (int)0x80402010, lo, false);
    int hi = __builtin_amdgcn_sdot4(f(v.z), 0x08040201, 127, false);
    hi = __builtin_amdgcn_sdot4(f(v.w), (int)0x80402010, hi, false);
    return (((uint32_t)hi << 8) | (uint32_t)lo) ^ 0x8080u;
}
// ---- tile front end: bytes -> per-thread 64-bit masks of the thread's 64 contiguous bytes --------------------------
struct Masks { u64 N, X, H; };   // '>' is looked up only where a line's first X sits (line_facts)

// newline mask and X (= not a posix space) mask of one 16-byte piece.  Almost every piece has no byte <= 32 other than
// its newlines: then X = ~N and the exact classification (13 ops per dword) is skipped for the whole wave.
__device__ __forceinline__ void piece_nx(const uint4 v, uint32_t& n, uint32_t& x) {
    // newline flags: 0x00 at '\n', 0xFF elsewhere (v_perm idiom, bzq_device.hpp)
    const uint32_t p0 = __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v.x ^ 0x06060606u), p1 = __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v.y ^ 0x06060606u),
                   p2 = __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v.z ^ 0x06060606u), p3 = __builtin_amdgcn_perm(0xFFFFFFFFu, 0xFFFFFFFFu, v.w ^ 0x06060606u);
    int lo = __builtin_amdgcn_sdot4((int)p0, 0x08040201, 127, false);
    lo = __builtin_amdgcn_sdot4((int)p1, (int)0x80402010, lo, false);
    int hi = __builtin_amdgcn_sdot4((int)p2, 0x08040201, 127, false);
    hi = __builtin_amdgcn_sdot4((int)p3, (int)0x80402010, hi, false);
    n = (((uint32_t)hi << 8) | (uint32_t)lo) ^ 0x8080u;
    x = ~n & 0xFFFFu;
    if (((le32_flags(v.x) & p0) | (le32_flags(v.y) & p1) | (le32_flags(v.z) & p2) | (le32_flags(v.w) & p3)) != 0u)
        x = ~flag_mask16(space_flags(v.x), space_flags(v.y), space_flags(v.z), space_flags(v.w)) & 0xFFFFu;
}

template <bool ASCII, bool STAGE>
__device__ __forceinline__ void tile_masks(const uint4 (&r)[4], int valid, uint16_t* s_n, uint16_t* s_x,
                                           uint16_t* s_h, uint8_t* s_tile) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int q = tid + BLOCK * s;
        const uint4 v = r[s];
        uint32_t n, x;
        piece_nx(v, n, x);
        uint32_t h = ASCII ? flag_mask16(v.x & 0x80808080u, v.y & 0x80808080u, v.z & 0x80808080u, v.w & 0x80808080u) : 0u;
        if (valid != TILE) {
            const int rem = valid - q * 16;
            const uint32_t keep = rem >= 16 ? 0xFFFFu : (rem > 0 ? ((1u << rem) - 1u) : 0u);
            n &= keep; x &= keep; h &= keep;
        }
        s_n[q] = (uint16_t)n; s_x[q] = (uint16_t)x;
        if (ASCII) s_h[q] = (uint16_t)h;
        if (STAGE) *reinterpret_cast<uint4*>(s_tile + q * 16) = v;
    }
}

// Pass 1: the tile is fetched coalesced (16 B per lane, piece q = tid + 256 s), laid down in LDS, and read back so that
// every thread holds ITS 64 contiguous bytes in registers: masks are then built in place (no mask transposition).  LDS layout: 256-byte rows 16 bytes apart, so that
// the 64-byte-stride read-back is free of bank conflicts (the gap rotates the banks by 4 per row).  Pass 1 only: pass 2
// keeps its pieces in fetch order, because its 16-byte stores must stay coalesced (owner-order stores scatter every
// wave's store over 64 cache lines: measured 1.36 -> 1.80 ms).
constexpr int LDS_TILE = TILE + (TILE / 256) * 16;
__device__ __forceinline__ int lds_at(int w) { return w + ((w >> 8) << 4); }
__device__ __forceinline__ void tile_to_own(const uint4 (&r)[4], uint8_t* s_tile, uint4 (&own)[4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int q = tid + BLOCK * s;
        *reinterpret_cast<uint4*>(s_tile + lds_at(q * 16)) = r[s];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) own[k] = *reinterpret_cast<const uint4*>(s_tile + lds_at(tid * 64 + k * 16));
}

template <bool ASCII>
__device__ __forceinline__ Masks own_masks(const uint4 (&own)[4], int valid) {
    Masks m{0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4 v = own[k];
        uint32_t n, x;
        piece_nx(v, n, x);
        m.N |= (u64)n << (16 * k); m.X |= (u64)x << (16 * k);
        if (ASCII) m.H |= (u64)flag_mask16(v.x & 0x80808080u, v.y & 0x80808080u, v.z & 0x80808080u, v.w & 0x80808080u) << (16 * k);
    }
    if (valid != TILE) {   // the last tile: bytes at or beyond the end of the chunk are nothing
        const int rem = valid - (int)threadIdx.x * 64;
        const u64 keep = rem >= 64 ? ~0ull : (rem > 0 ? ((1ull << rem) - 1ull) : 0ull);
        m.N &= keep; m.X &= keep; m.H &= keep;
    }
    return m;
}

// The four facts for this thread's 64 bytes.  cin_* are the tile-level carries (pass 1: all zero).
struct Facts { u64 seq, id, FG, Sb_incl, Sa_incl; uint32_t end_sb, end_hdr, end_x2; };

// byte_at(p): byte p (0..63) of this thread's 64 bytes, from LDS
template <typename ByteAt>
__device__ __forceinline__ Facts line_facts(const Masks& m, uint32_t cin_sb, uint32_t cin_hdr, uint32_t cin_x2, uint32_t cin_sa,
                                            uint32_t (*s_slot)[BLOCK / 64], ByteAt&& byte_at) {
    // round 1: Sb forwards, Sa backwards
    const WaveChain wb = chain_wave<false>(m.X, m.N, s_slot[0]);
    const WaveChain wa = chain_wave<true>(m.X, m.N, s_slot[1]);
    __syncthreads();
    const Chain sb = chain64(m.X, m.N, chain_cin<false>(wb, s_slot[0], cin_sb));
    const Chain sa = chain64_rev(m.X, m.N, chain_cin<true>(wa, s_slot[1], cin_sa));
    const u64 F = m.X & ~sb.excl;   // a line's first X
    // '>' matters only at a line's first X: one LDS byte per line instead of an equality mask over every byte
    u64 FG = 0;
    for (u64 ff = F; ff;) {
        const int p = __builtin_ctzll(ff);
        ff &= ff - 1;
        if (byte_at(p) == 62u) FG |= 1ull << p;
    }
    // round 2: hdr is decided at F; X2 is set by every other X
    const u64 hs = FG, hc = (F & ~FG) | m.N, xs = m.X & ~F;
    const WaveChain wh = chain_wave<false>(hs, hc, s_slot[2]);
    const WaveChain wx = chain_wave<false>(xs, m.N, s_slot[3]);
    __syncthreads();
    const Chain hd = chain64(hs, hc, chain_cin<false>(wh, s_slot[2], cin_hdr));
    const Chain x2 = chain64(xs, m.N, chain_cin<false>(wx, s_slot[3], cin_x2));
    Facts f;
    f.seq = ~hd.incl & sb.incl & sa.incl;
    f.id = hd.incl & x2.incl & sa.incl;
    f.FG = hs;
    f.Sb_incl = sb.incl; f.Sa_incl = sa.incl;
    f.end_sb = (uint32_t)(sb.incl >> 63); f.end_hdr = (uint32_t)(hd.incl >> 63); f.end_x2 = (uint32_t)(x2.incl >> 63);
    return f;
}

// state of the line at a tile edge: 0 no X yet, 1 header with only its '>', 2 header with id bytes, 3 sequence line
__device__ __forceinline__ uint32_t state_of(uint32_t sb, uint32_t hdr, uint32_t x2) { return sb ? (hdr ? (x2 ? 2u : 1u) : 3u) : 0u; }

// ---- pass 1 ------------------------------------------------------------------------------------------------------------
// sums[3t+0] = seq0 | id0 << 16 | hdr0 << 32 | newlines << 48           (everything as if the tile started a line)
// sums[3t+1] = c0seq | c0id << 16 | c23 << 32 | c1 << 48                (the part before the tile's first '\n')
// sums[3t+2] = pre | post << 16 | tail << 32 | flags << 48              (bytes before the first / after the last '\n',
//              trailing bytes after the last X or '\n'; flags: 1 c0hdr, 2 any X, 4 an X comes before the first '\n',
//              8|16 state at the tile end when it starts in state 0)
struct SumsArgs { const uint8_t* data; int64_t n; u64* sums; };

static __global__ __launch_bounds__(BLOCK) void k_fa_tile_sums(SumsArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[LDS_TILE];
    __shared__ uint32_t s_slot[4][BLOCK / 64];
    __shared__ u64 s_hasN[BLOCK / 64], s_hasE[BLOCK / 64];
    __shared__ uint32_t s_acc[BLOCK / 64][2];
    __shared__ uint32_t s_anyx[BLOCK / 64], s_edge[4];
    __shared__ uint32_t s_end;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t = xcd_tile(), t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    uint4 r[4], own[4];
    tile_fetch<true>(a.data, a.n, t0, valid, r);
    if (tid < 4) s_edge[tid] = 0u;
    tile_to_own(r, s_tile, own);
    const Masks m = own_masks<false>(own, valid);
    const u64 E = m.N | m.X;
    const u64 bN = __ballot(m.N != 0), bE = __ballot(E != 0);
    if (lane == 0) { s_hasN[wave] = bN; s_hasE[wave] = bE; }
    const int pos_own = tid * 64;
    const Facts f = line_facts(m, 0u, 0u, 0u, 0u, s_slot, [&](int p) { return (uint32_t)s_tile[lds_at(pos_own + p)]; });   // (its barriers also publish s_hasN / s_hasE)
    // bytes before the first '\n' of the tile, after its last '\n', after its last event
    const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull, above = lane < 63 ? (~0ull << (lane + 1)) : 0ull;
    bool n_before = (bN & below) != 0, n_after = (bN & above) != 0, e_after = (bE & above) != 0;
#pragma unroll
    for (int i = 0; i < BLOCK / 64; ++i) {
        if (i < wave) n_before |= s_hasN[i] != 0;
        if (i > wave) { n_after |= s_hasN[i] != 0; e_after |= s_hasE[i] != 0; }
    }
    const int pos0 = tid * 64;
    u64 vmask = ~0ull;   // bytes of this thread that exist
    if (valid != TILE) { const int rem = valid - pos0; vmask = rem >= 64 ? ~0ull : (rem > 0 ? ((1ull << rem) - 1ull) : 0ull); }
    auto pc = [](u64 x) { return (uint32_t)__builtin_popcountll(x); };
    auto below_first = [](u64 x) { return x ? ((x & (0 - x)) - 1ull) : ~0ull; };                     // bits under the lowest set bit
    auto above_last = [](u64 x) { return x ? ((x >> 63) ? 0ull : (~0ull << (64 - __builtin_clzll(x)))) : ~0ull; };
    // twelve counts, each <= 64 per thread and <= 16384 per tile, two per dword.  Four are sums over every thread (DPP scan per
    // wave); the eight edge counts live in the few threads in front of the tile's first '\n' and behind its last event: only
    // those lanes add theirs to LDS, and a wave without such a lane (waves 1 and 2, with lines shorter than 4 KiB) skips them.
    uint32_t w[2] = {pc(f.seq) | (pc(f.id) << 16), pc(f.FG) | (pc(m.N) << 16)};
    if (!n_before) {
        const u64 M1 = below_first(m.N) & vmask;
        if (M1) {
            atomicAdd(&s_edge[0], pc(f.seq & M1) | (pc(f.id & M1) << 16));
            atomicAdd(&s_edge[1], pc(f.Sa_incl & M1) | (pc(f.Sb_incl & f.Sa_incl & M1) << 16));
            atomicAdd(&s_edge[2], pc(M1));
            if (f.FG & M1) atomicAdd(&s_edge[3], pc(f.FG & M1) << 16);
        }
    }
    if (!n_after) { const u64 M2 = above_last(m.N) & vmask; if (M2) atomicAdd(&s_edge[2], pc(M2) << 16); }
    if (!e_after) { const u64 MT = above_last(E) & vmask; if (MT) atomicAdd(&s_edge[3], pc(MT)); }
    const bool wave_anyx = __ballot(m.X != 0) != 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        w[k] = dpp_scan_u32(w[k]);
        if (lane == 63) s_acc[wave][k] = w[k];
    }
    if (lane == 63) s_anyx[wave] = wave_anyx;
    if (tid == BLOCK - 1) s_end = state_of(f.end_sb, f.end_hdr, f.end_x2);
    const uint32_t lead_x = (uint32_t)(f.Sa_incl & 1ull);   // thread 0: an X comes before the first '\n'
    __syncthreads();
    if (tid == 0) {
        uint32_t t6[6] = {0u, 0u, s_edge[0], s_edge[1], s_edge[2], s_edge[3]};
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < BLOCK / 64; ++i) t6[k] += s_acc[i][k];
        uint32_t anyx = 0;
#pragma unroll
        for (int i = 0; i < BLOCK / 64; ++i) anyx |= s_anyx[i];
        const u64 flags = (u64)((t6[5] >> 16) & 1u) | ((u64)(anyx != 0u) << 1) | ((u64)lead_x << 2) | ((u64)s_end << 3);
        a.sums[3 * t + 0] = (u64)(t6[0] & 0xFFFF) | ((u64)(t6[0] >> 16) << 16) | ((u64)(t6[1] & 0xFFFF) << 32) | ((u64)(t6[1] >> 16) << 48);
        a.sums[3 * t + 1] = (u64)(t6[2] & 0xFFFF) | ((u64)(t6[2] >> 16) << 16) | ((u64)(t6[3] & 0xFFFF) << 32) | ((u64)(t6[3] >> 16) << 48);
        a.sums[3 * t + 2] = (u64)(t6[4] & 0xFFFF) | ((u64)(t6[4] >> 16) << 16) | ((u64)(t6[5] & 0xFFFF) << 32) | (flags << 48);
    }
}

// ---- resolve + scan -----------------------------------------------------------------------------------------------------
struct TileView {
    uint32_t seq0, id0, hdr0, nl, c0seq, c0id, c23, c1, pre, post, tail, c0hdr, anyx, lead_x, out0;
    __device__ __forceinline__ TileView(const u64* sums, int64_t t) {
        const u64 A = sums[3 * t], B = sums[3 * t + 1], C = sums[3 * t + 2];
        seq0 = (uint32_t)(A & 0xFFFF); id0 = (uint32_t)((A >> 16) & 0xFFFF); hdr0 = (uint32_t)((A >> 32) & 0xFFFF); nl = (uint32_t)(A >> 48);
        c0seq = (uint32_t)(B & 0xFFFF); c0id = (uint32_t)((B >> 16) & 0xFFFF); c23 = (uint32_t)((B >> 32) & 0xFFFF); c1 = (uint32_t)(B >> 48);
        pre = (uint32_t)(C & 0xFFFF); post = (uint32_t)((C >> 16) & 0xFFFF); tail = (uint32_t)((C >> 32) & 0xFFFF);
        const uint32_t fl = (uint32_t)(C >> 48);
        c0hdr = fl & 1u; anyx = (fl >> 1) & 1u; lead_x = (fl >> 2) & 1u; out0 = (fl >> 3) & 3u;
    }
    // state the tile leaves when the line that enters it is in state s
    __device__ __forceinline__ uint32_t out(uint32_t s) const { return nl ? out0 : (s == 0u ? out0 : (s == 1u ? (anyx ? 2u : 1u) : s)); }
    __device__ __forceinline__ bool all_space() const { return nl == 0u && anyx == 0u; }
};

struct ResolveArgs {
    const u64* sums; int64_t n_tiles; int64_t n; int32_t is_eof; int64_t line_cap;
    uint32_t* tile_in;      // per tile: entering state | x_follows << 2
    uint32_t* tile_cnt;     // per tile: 4 x uint32 {seq, id, hdr, nl}
    int64_t* grp;           // per group of BLOCK tiles: 4 x int64
    FaState* st;
};

static __global__ __launch_bounds__(BLOCK) void k_fa_resolve(ResolveArgs a) {
    __shared__ int64_t s_r[BLOCK / 64];
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    int64_t v[4] = {0, 0, 0, 0};
    if (t < a.n_tiles) {
        const TileView me(a.sums, t);
        // the state of the line that enters tile t: the nearest earlier tile with a '\n' fixes it, tiles without one map it.
        // Both walks stop after line_cap bytes: a line that long is an error whatever its state (the host then finds its
        // start with k_fa_long_start), and an unbounded walk would be quadratic on a file that is one giant line.
        const int64_t W = a.line_cap / TILE + 2;
        int64_t j = t - 1;
        while (j >= 0 && t - j <= W && (uint32_t)(a.sums[3 * j] >> 48) == 0u) --j;
        const bool lost = j >= 0 && (uint32_t)(a.sums[3 * j] >> 48) == 0u;
        uint32_t s = 0u;
        uint32_t post_j = 0u;
        if (!lost) {
            if (j >= 0) { const TileView tj(a.sums, j); s = tj.out0; post_j = tj.post; }
            for (int64_t k = j + 1; k < t; ++k) s = TileView(a.sums, k).out(s);
        }
        // a line that ends at this tile's first '\n' and is line_cap bytes or longer (buffered.mojo:634-636)
        if (me.nl) {
            if (lost) atomicMin(&a.st->long_tile, (unsigned long long)t);
            else {
                const int64_t len = (int64_t)me.pre + (t - 1 - j) * (int64_t)TILE + (int64_t)post_j;
                if (len >= a.line_cap) atomicMin(&a.st->long_pos, (unsigned long long)(t * (int64_t)TILE + me.pre - len));
            }
        }
        // does an X follow this tile's last byte before the next '\n'?  (skips tiles that are spaces only)
        bool xf = false;
        for (int64_t k = t + 1; k < a.n_tiles && k - t <= W; ++k) {
            const TileView tk(a.sums, k);
            if (tk.all_space()) continue;
            xf = tk.lead_x != 0u;
            break;
        }
        const uint32_t o = me.out(s);
        v[0] = (int64_t)me.seq0 - me.c0seq + (s == 3u ? me.c23 : (s == 0u ? me.c0seq : 0u)) + ((xf && o == 3u) ? me.tail : 0u);
        v[1] = (int64_t)me.id0 - me.c0id + (s == 2u ? me.c23 : (s == 1u ? me.c1 : (s == 0u ? me.c0id : 0u))) + ((xf && o == 2u) ? me.tail : 0u);
        v[2] = (int64_t)me.hdr0 - me.c0hdr + (s == 0u ? me.c0hdr : 0u);
        v[3] = me.nl;
        a.tile_in[t] = s | ((uint32_t)xf << 2);
        uint4 c = make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
        reinterpret_cast<uint4*>(a.tile_cnt)[t] = c;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t tot = block_sum_i64<BLOCK / 64>(v[k], s_r);
        if (threadIdx.x == 0) a.grp[(int64_t)blockIdx.x * 4 + k] = tot;
    }
}

struct BasesArgs {
    const uint32_t* tile_cnt; const int64_t* grp; int64_t n_tiles, n_groups;
    int64_t* base;   // per tile: 4 x int64 {seq, id, rec, nl} before the tile
    FaState* st;
};

static __global__ __launch_bounds__(BLOCK) void k_fa_bases(BasesArgs a) {
    __shared__ int64_t s_r[BLOCK / 64];
    __shared__ int64_t s_w[BLOCK / 64];
    const int64_t g = blockIdx.x, t = g * BLOCK + threadIdx.x;
    int64_t carry[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int64_t p = 0;
        for (int64_t i = threadIdx.x; i < g; i += BLOCK) p += a.grp[i * 4 + k];
        carry[k] = block_sum_i64<BLOCK / 64>(p, s_r);
    }
    uint4 c = make_uint4(0u, 0u, 0u, 0u);
    if (t < a.n_tiles) c = reinterpret_cast<const uint4*>(a.tile_cnt)[t];
    const int64_t v[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int64_t tot;
        const int64_t ex = block_exclusive_scan<int64_t, BLOCK / 64>(v[k], s_w, tot);
        if (t < a.n_tiles) a.base[t * 4 + k] = carry[k] + ex;
        if (g == a.n_groups - 1 && threadIdx.x == 0) {
            const int64_t total = carry[k] + tot;
            if (k == 0) a.st->seq_total = total;
            if (k == 1) a.st->id_total = total;
            if (k == 2) a.st->n_headers = total;
            if (k == 3) a.st->nl_total = total;
        }
    }
}

// ---- pass 2 ------------------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) P16 { uint32_t x, y, z, w; };

struct EmitArgs {
    const uint8_t* data; int64_t n;
    const uint32_t* tile_in; const int64_t* base;
    uint8_t* seq; uint8_t* id;
    int64_t* seq_ends; int64_t* id_ends; int64_t* hdr_pos; int64_t rec_cap;
    FaState* st;
};

// kept runs of one column inside this tile: whole 16-byte pieces come from registers (see the caller); this stores the
// first and last 16 bytes of every run that starts / ends in this thread's 64 bytes, from LDS
__device__ __forceinline__ void emit_run_edges(u64 own, u64 prev, u64 next, int pos0, int64_t rank0, uint8_t* __restrict__ col,
                                               const uint8_t* s_tile) {
    u64 starts = own & ~((own << 1) | (prev >> 63));
    while (starts) {
        const int p = __builtin_ctzll(starts);
        starts &= starts - 1;
        const u64 w = (own >> p) | (p ? (next << (64 - p)) : 0ull);
        const int len = (~w & 0xFFFFull) ? __builtin_ctzll(~w) : 16;
        uint8_t* d = col + rank0 + __builtin_popcountll(own & ((1ull << p) - 1ull));
        const uint8_t* s = s_tile + pos0 + p;
        if (len >= 16) *reinterpret_cast<P16*>(d) = *reinterpret_cast<const P16*>(s);
        else
            for (int i = 0; i < len; ++i) d[i] = s[i];
    }
    u64 ends = own & ~((own >> 1) | (next << 63));
    while (ends) {
        const int e = __builtin_ctzll(ends);
        ends &= ends - 1;
        const u64 w = (own << (63 - e)) | (e < 63 ? (prev >> (e + 1)) : 0ull);   // bit 63 = byte e, bit 48 = byte e-15
        if ((w >> 48) == 0xFFFFull) {
            uint8_t* d = col + rank0 + __builtin_popcountll(own & ((1ull << e) - 1ull)) - 15;
            *reinterpret_cast<P16*>(d) = *reinterpret_cast<const P16*>(s_tile + pos0 + e - 15);
        }
    }
}

template <bool ASCII>
static __global__ __launch_bounds__(BLOCK) void k_fa_emit(EmitArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[16 + TILE + 16];
    __shared__ __attribute__((aligned(16))) uint16_t s_n[TILE / 16], s_x[TILE / 16], s_h[ASCII ? TILE / 16 : 8];
    __shared__ uint32_t s_slot[4][BLOCK / 64];
    __shared__ u64 s_seq[BLOCK + 2], s_id[BLOCK + 2], s_rank[BLOCK];
    __shared__ u64 s_w[BLOCK / 64];
    uint8_t* s_tile = s_raw + 16;
    const int tid = threadIdx.x;
    const int64_t t = xcd_tile(), t0 = t * TILE;   // (neighbouring tiles on one XCD: bzq_device.hpp)
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    uint4 r[4];
    tile_fetch(a.data, a.n, t0, valid, r);
    tile_masks<ASCII, true>(r, valid, s_n, s_x, s_h, s_tile);
    const uint32_t tin = a.tile_in[t];
    const uint32_t s = tin & 3u, xf = (tin >> 2) & 1u;
    const int64_t seq_base = a.base[t * 4 + 0], id_base = a.base[t * 4 + 1], rec_base = a.base[t * 4 + 2];
    __syncthreads();
    Masks m;
    m.N = reinterpret_cast<const u64*>(s_n)[tid];
    m.X = reinterpret_cast<const u64*>(s_x)[tid];
    m.H = ASCII ? reinterpret_cast<const u64*>(s_h)[tid] : 0ull;
    const Facts f = line_facts(m, s != 0u, s == 1u || s == 2u, s == 2u, xf, s_slot, [&](int p) { return (uint32_t)s_tile[tid * 64 + p]; });
    // tile-local ranks of this thread's first byte in both columns and among the headers
    u64 tot;
    const u64 packed = (u64)__builtin_popcountll(f.seq) | ((u64)__builtin_popcountll(f.id) << 16) | ((u64)__builtin_popcountll(f.FG) << 32);
    const u64 rk = block_exclusive_scan<u64, BLOCK / 64>(packed, s_w, tot);
    const int64_t r_seq = (int64_t)(rk & 0xFFFF), r_id = (int64_t)((rk >> 16) & 0xFFFF), r_fg = (int64_t)((rk >> 32) & 0xFFFF);
    s_seq[tid + 1] = f.seq; s_id[tid + 1] = f.id; s_rank[tid] = rk;
    if (tid == 0) { s_seq[0] = 0; s_id[0] = 0; s_seq[BLOCK + 1] = 0; s_id[BLOCK + 1] = 0; }
    __syncthreads();
    // whole pieces from registers: piece q = tid + 256 s belongs to thread q >> 2's masks
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = tid + BLOCK * k, o = q >> 2, sh = (q & 3) * 16;
        const u64 ms = s_seq[o + 1], mi = s_id[o + 1];
        const uint32_t ps = (uint32_t)(ms >> sh) & 0xFFFFu, pi = (uint32_t)(mi >> sh) & 0xFFFFu;
        if (ps == 0xFFFFu || pi == 0xFFFFu) {
            const u64 ro = s_rank[o];
            const u64 lowbits = sh ? ((1ull << sh) - 1ull) : 0ull;
            const P16 v = {r[k].x, r[k].y, r[k].z, r[k].w};
            if (ps == 0xFFFFu) *reinterpret_cast<P16*>(a.seq + seq_base + (int64_t)(ro & 0xFFFF) + __builtin_popcountll(ms & lowbits)) = v;
            else *reinterpret_cast<P16*>(a.id + id_base + (int64_t)((ro >> 16) & 0xFFFF) + __builtin_popcountll(mi & lowbits)) = v;
        }
    }
    // run edges from LDS
    const int pos0 = tid * 64;
    if (f.seq) emit_run_edges(f.seq, s_seq[tid], s_seq[tid + 2], pos0, seq_base + r_seq, a.seq, s_tile);
    if (f.id) emit_run_edges(f.id, s_id[tid], s_id[tid + 2], pos0, id_base + r_id, a.id, s_tile);
    // records: the '>' of record k closes record k-1
    u64 fg = f.FG;
    while (fg) {
        const int p = __builtin_ctzll(fg);
        fg &= fg - 1;
        const u64 lowbits = (1ull << p) - 1ull;
        const int64_t k = rec_base + r_fg + __builtin_popcountll(f.FG & lowbits);
        if (k < a.rec_cap) a.hdr_pos[k] = t0 + pos0 + p;
        if (k >= 1 && k - 1 < a.rec_cap) {
            a.seq_ends[k - 1] = seq_base + r_seq + __builtin_popcountll(f.seq & lowbits);
            a.id_ends[k - 1] = id_base + r_id + __builtin_popcountll(f.id & lowbits);
        }
    }
    // a sequence byte before any header (parser.mojo:196-200)
    if (rec_base + r_fg == 0 && f.seq) {
        const u64 first_fg = f.FG ? (f.FG & (0 - f.FG)) : 0ull;
        const u64 before = f.seq & (first_fg ? (first_fg - 1ull) : ~0ull);
        if (before) atomicMin(&a.st->nohdr_pos, (unsigned long long)(t0 + pos0 + __builtin_ctzll(before)));
    }
    if (ASCII) {
        const u64 bad = m.H & (f.seq | f.id);
        if (bad) {
            const int p = __builtin_ctzll(bad);
            const int64_t rec = rec_base + r_fg + __builtin_popcountll(f.FG & ((2ull << p) - 1ull)) - 1;
            if (rec >= 0) atomicMin(&a.st->ascii_rec, (unsigned long long)rec);
        }
    }
}

// ---- after the passes -----------------------------------------------------------------------------------------------------
struct FinishArgs {
    const uint8_t* data; int64_t n; int32_t is_eof;
    const u64* sums; const int64_t* base; int64_t n_tiles;
    int64_t* seq_ends; int64_t* id_ends; const int64_t* hdr_pos; int64_t rec_cap;
    FaState* st;
    int64_t line_cap;
};

static __global__ __launch_bounds__(BLOCK) void k_fa_finish(FinishArgs a) {
    __shared__ int64_t s_r[BLOCK / 64];
    __shared__ int64_t s_p;
    FaState* st = a.st;
    const int64_t H = st->n_headers;
    // the last tile with a newline (all threads look, strided from the end; a file that is one giant line has none)
    __shared__ int64_t s_last;
    if (threadIdx.x == 0) s_last = -1;
    __syncthreads();
    for (int64_t base = a.n_tiles - 1; base >= 0 && s_last < 0; base -= BLOCK) {
        const int64_t j = base - threadIdx.x;
        if (j >= 0 && (uint32_t)(a.sums[3 * j] >> 48) != 0u) atomicMax((long long*)&s_last, (long long)j);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // offset after the last '\n'
        const int64_t j = s_last;
        int64_t lls = 0;
        if (j >= 0) {
            const int64_t vj = (a.n - j * (int64_t)TILE) < TILE ? (a.n - j * (int64_t)TILE) : TILE;
            lls = j * (int64_t)TILE + vj - (int64_t)((a.sums[3 * j + 2] >> 16) & 0xFFFF);
        }
        st->last_line_start = lls;
        // a chunk that is not the last one: a line without its '\n' yet is not looked at (it may still turn out too long),
        // so a header in it neither closes the record before it nor opens one
        // (counted by position: inside a line that outran the bounded walks of k_fa_resolve the states are not meaningful
        // and more than one '>' may have been taken for a header)
        int64_t Hc = H;
        if (!a.is_eof) {
            int64_t lo = 0, hi = H < a.rec_cap ? H : a.rec_cap;
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.hdr_pos[mid] < lls) lo = mid + 1; else hi = mid; }
            Hc = lo;
        }
        int64_t closed = a.is_eof ? H : (Hc > 0 ? Hc - 1 : 0);
        st->n_closed = closed;
        if (a.is_eof && H > 0 && H - 1 < a.rec_cap) { a.seq_ends[H - 1] = st->seq_total; a.id_ends[H - 1] = st->id_total; }
        // where the next chunk starts: the line of the last header (the open record), or after the last blank line
        int64_t p = a.n;
        if (!a.is_eof) {
            if (Hc > 0) {
                p = Hc - 1 < a.rec_cap ? a.hdr_pos[Hc - 1] : 0;
                // (at most line_cap spaces: a header line with more in front of its '>' is a too-long line, which the host reports)
                for (int64_t i = 0; i < a.line_cap && p > 0 && a.data[p - 1] != 10; ++i) --p;
            } else {
                p = lls;
            }
        }
        st->consumed = p;
        s_p = p;
    }
    __syncthreads();
    const int64_t p = s_p;
    const bool at_end = p >= a.n;
    const int64_t t = at_end ? 0 : p / TILE, off = at_end ? 0 : p - t * TILE;
    int64_t c = 0;
    for (int64_t i = threadIdx.x; i < off; i += BLOCK) c += a.data[t * TILE + i] == 10;
    const int64_t in_tile = block_sum_i64<BLOCK / 64>(c, s_r);
    const int64_t lines = at_end ? st->nl_total : a.base[t * 4 + 3] + in_tile;
    if (threadIdx.x == 0) st->lines_consumed = lines;
}

struct EmptyArgs { const int64_t* seq_ends; FaState* st; int64_t rec_cap; };
static __global__ __launch_bounds__(BLOCK) void k_fa_empty(EmptyArgs a) {
    const int64_t R = a.st->n_closed < a.rec_cap ? a.st->n_closed : a.rec_cap;
    for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < R; r += (int64_t)gridDim.x * BLOCK) {
        const int64_t lo = r ? a.seq_ends[r - 1] : 0;
        if (a.seq_ends[r] == lo) { atomicMin(&a.st->empty_rec, (unsigned long long)r); break; }
    }
}

// Cold path (error text): query[0] = newlines in [0, pos); query[1] = start of the line that holds pos;
// query[2] = headers whose '>' is before pos
struct QueryArgs { const uint8_t* data; int64_t n; const int64_t* base; const int64_t* hdr_pos; int64_t n_headers; int64_t pos; FaState* st; int64_t line_cap; };
static __global__ __launch_bounds__(BLOCK) void k_fa_query(QueryArgs a) {
    __shared__ int64_t s_r[BLOCK / 64];
    const int64_t pos = a.pos < a.n ? a.pos : a.n;
    const bool at_end = pos >= a.n;
    const int64_t t = at_end ? 0 : pos / TILE, off = at_end ? 0 : pos - t * TILE;
    int64_t c = 0;
    for (int64_t i = threadIdx.x; i < off; i += BLOCK) c += a.data[t * TILE + i] == 10;
    const int64_t in_tile = block_sum_i64<BLOCK / 64>(c, s_r);
    const int64_t lines = at_end ? a.st->nl_total : a.base[t * 4 + 3] + in_tile;
    if (threadIdx.x == 0) {
        a.st->query[0] = lines;
        int64_t p = pos;
        for (int64_t i = 0; i <= a.line_cap && p > 0 && a.data[p - 1] != 10; ++i) --p;   // (callers ask about lines shorter than line_cap)
        a.st->query[1] = p;
        int64_t lo = 0, hi = a.n_headers;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.hdr_pos[mid] < a.pos) lo = mid + 1; else hi = mid; }
        a.st->query[2] = lo;
    }
}

// Cold path: the start of the line that ends at the first newline of tile `tile` (k_fa_resolve gave up walking back):
// the nearest earlier tile with a newline, searched by the whole workgroup; query[3] = offset after that newline (0 if none)
struct LongStartArgs { const u64* sums; int64_t tile; FaState* st; };
static __global__ __launch_bounds__(BLOCK) void k_fa_long_start(LongStartArgs a) {
    __shared__ int64_t s_last;
    if (threadIdx.x == 0) s_last = -1;
    __syncthreads();
    for (int64_t base = a.tile - 1; base >= 0 && s_last < 0; base -= BLOCK) {
        const int64_t j = base - threadIdx.x;
        if (j >= 0 && (uint32_t)(a.sums[3 * j] >> 48) != 0u) atomicMax((long long*)&s_last, (long long)j);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int64_t j = s_last;
        a.st->query[3] = j < 0 ? 0 : j * (int64_t)TILE + TILE - (int64_t)((a.sums[3 * j + 2] >> 16) & 0xFFFF);
    }
}

// ---- byte-range shards (bzq_fasta_shard_stitch, bzq_api.hip) ------------------------------------------------------
// What one rank tells the others about its byte range so that every rank can place the record boundaries (new design;
// the reference is one sequential LineIterator, buffered.mojo:600-638).  A header line is a line whose first byte that
// is not a posix space is '>' (parser.mojo:181-203 after _strip_spaces); the line STARTS behind a '\n' (or at the
// stream's first byte), so a '>' is a header exactly when walking back from it over spaces ends at a '\n'.
//   first_header  smallest line start (offset behind a '\n' of this range) of a header line whose '>' is in this range; NONE
//   lead_kind     the bytes before the range's first '\n' (all of it when there is none): 0 the first non-space byte is
//                 not '>', 1 it is '>', 2 only spaces and then a '\n', 3 only spaces to the end of the range
//   tail_open     start of the range's last line (behind its last '\n') when that line is non-empty and all spaces so
//                 far -- whether it is a header line is decided by a later rank's lead_kind; -1 otherwise
struct ProbeOut { unsigned long long first_header; int64_t lead_kind, tail_open, last_byte, newlines; };
struct ProbeArgs { const uint8_t* data; int64_t n; ProbeOut* out; int64_t walk_cap; };

__device__ __forceinline__ bool is_space_byte(uint32_t c) { return c == 32u || (c - 9u) <= 4u || (c - 28u) <= 2u; }

// every '>' of tiles [tile_lo, tile_lo + gridDim.x).  Launched over growing tile ranges (8, 64, 512, ... tiles): a launch
// whose predecessors found a header returns at once (a later '>' cannot have an earlier line start: the walk back ends
// at the first '\n'), so a range with a header in its first kilobytes costs a few empty launches and a range without
// one is read once at streaming speed.  (One grid over everything with an early-exit flag does not work on this part:
// the flag is one address polled from 8 XCDs -- 0.45 ms for 2048 workgroups, longer than reading 3 GB.)
struct ProbeHdrArgs { const uint8_t* data; int64_t n; ProbeOut* out; int64_t tile_lo, walk_cap; };
static __global__ __launch_bounds__(BLOCK) void k_fa_probe_headers(ProbeHdrArgs a) {
    __shared__ unsigned long long s_min;
    const int tid = threadIdx.x;
    if (tid == 0) s_min = a.out->first_header;   // written by earlier launches only
    __syncthreads();
    if (s_min != NONE) return;
    __syncthreads();
    const int64_t t0 = (a.tile_lo + blockIdx.x) * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    uint4 r[4];
    tile_fetch<true>(a.data, a.n, t0, valid, r);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int pos = (tid + BLOCK * s) * 16;
        uint32_t m = eq_mask16<'>'>(r[s]);
        const int rem = valid - pos;
        if (rem < 16) m &= rem > 0 ? ((1u << rem) - 1u) : 0u;
        while (m) {
            const int bit = __builtin_ctz(m);
            m &= m - 1;
            // (the walk stops after line_capacity bytes: a line that long fails wherever it is parsed, and a hostile run of
            // spaces in front of a '>' must not cost a serial pass over it)
            int64_t q = t0 + pos + bit - 1;
            const int64_t stop = q - a.walk_cap;
            while (q >= 0 && q > stop && a.data[q] != 10 && is_space_byte(a.data[q])) --q;
            if (q >= 0 && a.data[q] == 10) atomicMin(&s_min, (unsigned long long)(q + 1));
        }
    }
    __syncthreads();
    if (tid == 0 && s_min != NONE) atomicMin(&a.out->first_header, s_min);
}

// the two ends of the range, one workgroup: forward to the first '\n' or non-space byte, backward to the last
struct EdgeScan { int first_nl, first_x, last_nl, last_x; };
static __global__ __launch_bounds__(BLOCK) void k_fa_probe_edges(ProbeArgs a) {
    __shared__ EdgeScan s_e;
    const int tid = threadIdx.x;
    constexpr int SPAN = BLOCK * 16;
    int64_t lead_kind = 3;
    // (both scans give up after line_capacity bytes of spaces: that line fails wherever it is parsed, whatever is reported here)
    for (int64_t b0 = 0; b0 < a.n && b0 <= a.walk_cap; b0 += SPAN) {
        if (tid == 0) { s_e.first_nl = SPAN; s_e.first_x = SPAN; }
        __syncthreads();
        const int64_t pos = b0 + tid * 16;
        if (pos < a.n) {
            uint32_t nm, xm;
            piece_nx(load16(a.data, pos, a.n), nm, xm);
            const int64_t rem = a.n - pos;
            if (rem < 16) { nm &= (1u << rem) - 1u; xm &= (1u << rem) - 1u; }
            if (nm) atomicMin(&s_e.first_nl, tid * 16 + __builtin_ctz(nm));
            if (xm) atomicMin(&s_e.first_x, tid * 16 + __builtin_ctz(xm));
        }
        __syncthreads();
        const int fn = s_e.first_nl, fx = s_e.first_x;
        __syncthreads();
        if (fx < fn) { lead_kind = a.data[b0 + fx] == '>' ? 1 : 0; break; }
        if (fn < SPAN) { lead_kind = 2; break; }
    }
    int64_t tail_open = -1;
    for (int64_t e0 = a.n; e0 > 0 && a.n - e0 <= a.walk_cap; e0 -= SPAN) {   // the span [e0 - SPAN, e0), clipped at 0
        if (tid == 0) { s_e.last_nl = -1; s_e.last_x = -1; }
        __syncthreads();
        const int64_t b0 = e0 - SPAN, pos = b0 + tid * 16;
        if (pos + 16 > 0) {
            uint32_t nm = 0, xm = 0;
            if (pos >= 0) {
                piece_nx(load16(a.data, pos, a.n), nm, xm);
                const int64_t rem = a.n - pos;
                if (rem < 16) { nm &= (1u << rem) - 1u; xm &= (1u << rem) - 1u; }
            } else {   // the piece that straddles offset 0: byte by byte
                for (int k = (int)-pos; k < 16; ++k) {
                    const uint32_t c = a.data[pos + k];
                    if (c == 10u) nm |= 1u << k; else if (!is_space_byte(c)) xm |= 1u << k;
                }
            }
            if (nm) atomicMax(&s_e.last_nl, tid * 16 + 31 - __builtin_clz(nm));
            if (xm) atomicMax(&s_e.last_x, tid * 16 + 31 - __builtin_clz(xm));
        }
        __syncthreads();
        const int ln = s_e.last_nl, lx = s_e.last_x;
        __syncthreads();
        if (lx > ln) break;                                    // the last line has a non-space byte
        if (ln >= 0) { if (b0 + ln + 1 < a.n) tail_open = b0 + ln + 1; break; }   // all spaces behind the last '\n' (none: empty line)
    }
    if (tid == 0) {
        a.out->lead_kind = lead_kind; a.out->tail_open = tail_open;
        a.out->last_byte = a.n > 0 ? a.data[a.n - 1] : 10;
    }
}

// cold path (error text needs stream-global line numbers): '\n' count of [0, n)
static __global__ __launch_bounds__(BLOCK) void k_fa_count_newlines(ProbeArgs a) {
    __shared__ int64_t s_r[BLOCK / 64];
    int64_t c = 0;
    for (int64_t pos = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 16; pos < a.n; pos += (int64_t)gridDim.x * BLOCK * 16) {
        uint32_t m = eq_mask16<10>(load16(a.data, pos, a.n));
        const int64_t rem = a.n - pos;
        if (rem < 16) m &= (1u << rem) - 1u;
        c += __popc(m);
    }
    const int64_t tot = block_sum_i64<BLOCK / 64>(c, s_r);
    if (threadIdx.x == 0 && tot) atomicAdd((unsigned long long*)&a.out->newlines, (unsigned long long)tot);
}

// generate_synthetic_fasta_buffer (utils.mojo:1033-1139): one thread per record
struct FaGenArgs {
    uint8_t* out; int64_t first, count; int32_t min_len, num_digits, line_width; int64_t len_range, period;
    const int64_t* size_prefix;   // [period + 1] prefix sums of one period's record sizes without the header (device), or null
};
static __global__ __launch_bounds__(BLOCK) void k_fa_generate(FaGenArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= a.count) return;
    const int64_t i = a.first + idx;
    const int64_t hdr = 6 + a.num_digits + 1;
    int64_t L = a.min_len, off;
    if (a.len_range > 1) {
        L = a.min_len + (int64_t)(((u64)i * 31ull + 7ull) % (u64)a.len_range);
        const int64_t q = i / a.period, r = i - q * a.period, q0 = a.first / a.period, r0 = a.first - q0 * a.period;
        off = idx * hdr + (q * a.size_prefix[a.period] + a.size_prefix[r]) - (q0 * a.size_prefix[a.period] + a.size_prefix[r0]);
    } else {
        off = idx * (hdr + L + L / a.line_width + (L % a.line_width ? 1 : 0));
    }
    uint8_t* o = a.out + off;
    const uint8_t lut[8] = {'G', 'C', 'G', 'C', 'A', 'T', 'A', 'T'};   // gc_bias = 0.5
    *o++ = '>'; *o++ = 'r'; *o++ = 'e'; *o++ = 'a'; *o++ = 'd'; *o++ = '_';
    {
        int64_t v = i;
        for (int d = a.num_digits - 1; d >= 0; --d) { o[d] = (uint8_t)('0' + (int)(v % 10)); v /= 10; }
        o += a.num_digits;
    }
    *o++ = '\n';
    const u64 MASK = 0x7FFFFFFFFFFFFFFFull;
    u64 st = ((u64)i * 6364136223846793005ull + 1442695040888963407ull) & MASK;
    int col = 0;
    for (int64_t b = 0; b < L; ++b) {
        st = (st * 6364136223846793005ull + 1442695040888963407ull) & MASK;
        *o++ = lut[(st >> 33) & 7];
        if (++col == a.line_width) { *o++ = '\n'; col = 0; }
    }
    if (col > 0) *o++ = '\n';
}

} // namespace fa
} // namespace bzq
