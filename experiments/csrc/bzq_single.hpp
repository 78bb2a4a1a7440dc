// bzq_single.hpp -- single-pass FASTQ batch-parse kernel with a prefix-service workgroup.
//
// One launch reads the input once.  The two cross-tile dependencies (line index of a tile's first
// byte; column offsets of its three byte streams) are resolved by ONE dedicated workgroup instead of
// per-tile look-backs: workgroups draw an atomic ticket when they start; ticket 0 becomes the prefix
// service, ticket t+1 processes tile t.
//
//   tile t                                   service (wave 0: lines, wave 1: columns)
//   ------                                   ---------------------------------------
//   load tile, newline bitmap, count  --dc[t]-->  scan 256 counts per step (DPP)  --pc[t]-->
//   meanwhile: line table + phase-agnostic analysis of its lines (every line as a potential header)
//   line index known: roles, segment tables, byte counts  --da[t]-->  scan  --ps/pq/pi[t]-->
//   per-record outputs, scatter of the three streams
//
// Every word is an 8-byte {flag, value} granule written once per launch with a relaxed agent-scope
// atomic store and polled with relaxed agent-scope loads (MI355X_MICROARCH.md "R2": the data is the
// flag, no fences).  A workgroup only waits on the service, the service only waits on tiles with
// smaller tickets, i.e. workgroups that have already started: no deadlock under any dispatch
// order.  Spins are bounded; on timeout ChunkState::lookback_timeout is set and the host re-runs the
// chunk on the two-pass kernels.
#pragma once
#include "bzq_fused.hpp"

namespace bzq {

constexpr int SVC_K = 4; // tiles per lane per service step (256 tiles per step)

__device__ __forceinline__ int64_t wait_granule(const u64* p, ChunkState* st) {
    int spins = 0;
    for (;;) {
        const u64 v = ld_agent(p);
        if ((v >> 62) == 2) return (int64_t)(v & DESC_VMASK) - DESC_BIAS;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { st->lookback_timeout = 1; return 0; }
    }
}

struct SingleArgs {
    FusedArgs f;      // g, n, prev_byte, n_tiles, columns, record arrays, state, schema bounds
    u64* dc;          // [n_tiles] tile -> service: newline count
    u64* pc;          // [n_tiles] service -> tile: exclusive line prefix
    u64* da;          // [n_tiles] tile -> service: packed (seq, qual, id) byte counts
    u64* ps;          // [n_tiles] service -> tile: exclusive column prefixes
    u64* pq;
    u64* pi;
    // MODE 1 (two-level look-back, no service workgroup): per 64-tile block, one word per quantity that
    // goes from {A, block sum} to {P, inclusive prefix through the block}
    u64* bc;          // [n_blocks] lines
    u64* bs;          // [n_blocks] seq / qual / id bytes
    u64* bq;
    u64* bi;
};

constexpr int HB = 64; // tiles per look-back block

// One component of the two-level look-back, executed by a full wave for tile t = HB*b + i:
//   prefix(t) = [inclusive prefix through block b-1] + sum of the aggregates of tiles HB*b .. t-1.
// `mine` is the lane's already-loaded-and-valid tile aggregate for predecessor t-1-lane (0 when that
// predecessor is outside the block).  Returns the exclusive prefix of tile t; when the tile is the last
// of its block it also publishes the block sum (before the block walk) and the block prefix (after).
__device__ inline int64_t block_lookback(u64* bd, int64_t b, int64_t in_block_sum, int64_t own, bool last_in_block,
                                         int64_t start_value, int lane, ChunkState* st) {
    if (last_in_block && lane == 0) st_agent(&bd[b], DESC_A | (u64)(in_block_sum + own));
    int64_t running = 0, base = b - 1;
    int spins = 0;
    for (;;) {
        const int64_t q = base - lane;
        const u64 g = q >= 0 ? ld_agent(&bd[q]) : (DESC_P | (u64)(start_value + DESC_BIAS));
        const int flag = (int)(g >> 62);
        const u64 pm = __ballot(flag == 2), xm = __ballot(flag == 0);
        const int f = pm ? __builtin_ctzll(pm) : 64;
        const u64 nearer = f >= 64 ? ~0ull : ((1ull << f) - 1ull);
        if (xm & nearer) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { if (lane == 0) st->lookback_timeout = 1; break; }
            continue;
        }
        int64_t v = (int64_t)(g & DESC_VMASK);
        if (flag == 2) v -= DESC_BIAS;
        running += wave_sum(lane <= f ? v : 0);
        if (f < 64) break;
        base -= 64;
    }
    if (last_in_block && lane == 0) st_agent(&bd[b], DESC_P | (u64)(running + in_block_sum + own + DESC_BIAS));
    return running + in_block_sum;
}

// Tile aggregates of the predecessors inside the own block: lane L waits for tile t-1-L (L < i).
__device__ inline u64 own_block_aggregate(const u64* ta, int64_t t, int i, int lane, ChunkState* st) {
    u64 v = 0;
    if (lane < i) {
        int spins = 0;
        for (;;) {
            const u64 g = ld_agent(&ta[t - 1 - lane]);
            if ((g >> 62) == 1) { v = g & DESC_VMASK; break; }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { st->lookback_timeout = 1; break; }
        }
    }
    return v;
}


// wave 0 of the service workgroup
__device__ inline void service_lines(const SingleArgs& a, int lane) {
    ChunkState* st = a.f.st;
    const int64_t nt = a.f.n_tiles;
    int64_t carry = st->P0, base = 0;
    int spins = 0;
    while (base < nt) {
        const int64_t i0 = base + (int64_t)lane * SVC_K;
        u64 v[SVC_K];
        bool ok = true;
        int64_t sum = 0;
#pragma unroll
        for (int k = 0; k < SVC_K; ++k) {
            v[k] = 0;
            if (i0 + k < nt) {
                const u64 g = ld_agent(&a.dc[i0 + k]);
                ok = ok && ((g >> 62) == 1);
                v[k] = g & DESC_VMASK;
                sum += (int64_t)v[k];
            }
        }
        const u64 okm = __ballot(ok);
        const int nv = (~okm) ? __builtin_ctzll(~okm) : 64; // leading lanes whose tiles have all published
        if (nv == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { if (lane == 0) st->lookback_timeout = 1; return; }
            continue;
        }
        const u64 mine = lane < nv ? (u64)sum : 0ull;
        const u64 incl = dpp_scan_u64(mine);
        int64_t ex = carry + (int64_t)(incl - mine);
        if (lane < nv) {
#pragma unroll
            for (int k = 0; k < SVC_K; ++k)
                if (i0 + k < nt) { st_agent(&a.pc[i0 + k], DESC_P | (u64)(ex + DESC_BIAS)); ex += (int64_t)v[k]; }
        }
        const uint32_t tl = __builtin_amdgcn_readlane((uint32_t)incl, 63), th = __builtin_amdgcn_readlane((uint32_t)(incl >> 32), 63);
        carry += (int64_t)(((u64)th << 32) | tl);
        base += (int64_t)nv * SVC_K;
    }
    if (lane == 0) st->P = carry;
}

// wave 1 of the service workgroup
__device__ inline void service_cols(const SingleArgs& a, int lane) {
    ChunkState* st = a.f.st;
    const int64_t nt = a.f.n_tiles;
    int64_t cS = st->S0, cQ = st->Q0, cI = st->I0, base = 0;
    int spins = 0;
    while (base < nt) {
        const int64_t i0 = base + (int64_t)lane * SVC_K;
        u64 v[SVC_K];
        bool ok = true;
        u64 sq = 0, dd = 0; // (seq | qual << 32), id
#pragma unroll
        for (int k = 0; k < SVC_K; ++k) {
            v[k] = 0;
            if (i0 + k < nt) {
                const u64 g = ld_agent(&a.da[i0 + k]);
                ok = ok && ((g >> 62) == 1);
                v[k] = g & DESC_VMASK;
                sq += (v[k] & 0xFFFFFull) | (((v[k] >> 20) & 0xFFFFFull) << 32);
                dd += (v[k] >> 40) & 0xFFFFFull;
            }
        }
        const u64 okm = __ballot(ok);
        const int nv = (~okm) ? __builtin_ctzll(~okm) : 64;
        if (nv == 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { if (lane == 0) st->lookback_timeout = 1; return; }
            continue;
        }
        const u64 msq = lane < nv ? sq : 0ull, mdd = lane < nv ? dd : 0ull;
        const u64 isq = dpp_scan_u64(msq), idd = dpp_scan_u64(mdd);
        const u64 esq = isq - msq;
        int64_t eS = cS + (int64_t)(esq & 0xFFFFFFFFull), eQ = cQ + (int64_t)(esq >> 32), eI = cI + (int64_t)(idd - mdd);
        if (lane < nv) {
#pragma unroll
            for (int k = 0; k < SVC_K; ++k)
                if (i0 + k < nt) {
                    st_agent(&a.ps[i0 + k], DESC_P | (u64)(eS + DESC_BIAS));
                    st_agent(&a.pq[i0 + k], DESC_P | (u64)(eQ + DESC_BIAS));
                    st_agent(&a.pi[i0 + k], DESC_P | (u64)(eI + DESC_BIAS));
                    eS += (int64_t)(v[k] & 0xFFFFFull); eQ += (int64_t)((v[k] >> 20) & 0xFFFFFull); eI += (int64_t)((v[k] >> 40) & 0xFFFFFull);
                }
        }
        const u64 tsq = ((u64)__builtin_amdgcn_readlane((uint32_t)(isq >> 32), 63) << 32) | __builtin_amdgcn_readlane((uint32_t)isq, 63);
        const u64 tdd = ((u64)__builtin_amdgcn_readlane((uint32_t)(idd >> 32), 63) << 32) | __builtin_amdgcn_readlane((uint32_t)idd, 63);
        cS += (int64_t)(tsq & 0xFFFFFFFFull); cQ += (int64_t)(tsq >> 32); cI += (int64_t)tdd;
        base += (int64_t)nv * SVC_K;
    }
    if (lane == 0) { st->S = cS; st->Q = cQ; st->I = cI; }
}

template <bool CA, bool CQ, bool OFFS, int MODE>
static __global__ __launch_bounds__(BLOCK) void k_single(SingleArgs sa) {
    const FusedArgs& a = sa.f;
    __shared__ __attribute__((aligned(16))) uint8_t s_tile_raw[16 + TILE + 32];
    __shared__ __attribute__((aligned(16))) uint16_t s_mask[PIECES];
    __shared__ uint16_t s_nl[MAXL + 4];
    __shared__ uint16_t s_src[3][SEGS], s_len[3][SEGS], s_dst[3][SEGS];
    __shared__ __attribute__((aligned(16))) uint16_t s_pline[PIECES];
    __shared__ u64 s_w64[4];
    __shared__ uint32_t s_w[4];
    __shared__ int64_t s_bcast[4];
    __shared__ int s_cnt[3];
    uint8_t* s_tile = s_tile_raw + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid == 0) s_bcast[0] = (int64_t)atomicAdd(a.ticket, 1ull);
    __syncthreads();
    const int64_t ticket = s_bcast[0];
    if (MODE == 0 && ticket == 0) { // prefix service (no barriers below this point in this workgroup)
        if (wave == 0) service_lines(sa, lane);
        else if (wave == 1) service_cols(sa, lane);
        return;
    }
    const int64_t t = MODE == 0 ? ticket - 1 : ticket;
    const int64_t hb = t / HB;
    const int hi = (int)(t - hb * HB);
    const int64_t t0 = t * TILE;
    const int valid = (int)((a.n - t0) < TILE ? (a.n - t0) : TILE);
    uint4 r[4];
    tile_fetch(a.g, a.n, t0, valid, r);
    tile_stage<true>(r, valid, s_mask, s_tile);
    ByteSrc bs{a.g, a.n, a.prev_byte, s_tile, t0, valid};
    const bool first_starts = (bs.at(t0 - 1) == 10u);
    __syncthreads();
    const u64* s_mask64 = reinterpret_cast<const u64*>(s_mask);
    const u64 m64 = s_mask64[tid];
    uint32_t c = 0;
    const uint32_t excl = block_exclusive_scan<uint32_t, 4>((uint32_t)__popcll(m64), s_w, c);
    if (tid == 0) st_agent(&sa.dc[t], DESC_A | (u64)c);           // -> service (lines)
    const bool dense = ((int)c > MAXL) || a.force_dense;
    {
        const uint32_t l0 = excl, l1 = l0 + (uint32_t)__popc((uint32_t)m64 & 0xFFFFu),
                       l2 = l0 + (uint32_t)__popc((uint32_t)m64), l3 = l0 + (uint32_t)__popcll(m64 & 0xFFFFFFFFFFFFull);
        *reinterpret_cast<u64*>(&s_pline[4 * tid]) = (u64)l0 | ((u64)l1 << 16) | ((u64)l2 << 32) | ((u64)l3 << 48);
    }
    if (!dense) {
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

    // ---- phase-agnostic line analysis (overlaps the wait for the line index): thread k owns lines
    // 4k..4k+3; every line is measured both raw and as a potential header (kept range after strip)
    int Ls[4], Ll[4], Ks[4], Kl[4];
    bool Lin[4];
    if (!dense) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int j = 4 * tid + rr;
            Ls[rr] = 0; Ll[rr] = 0; Ks[rr] = 0; Kl[rr] = 0; Lin[rr] = false;
            if (j <= (int)c) {
                const int start = j ? (int)s_nl[j - 1] + 1 : 0;
                const bool end_in = j < (int)c;
                const int end = end_in ? (int)s_nl[j] : valid;
                const bool sknown = j > 0 ? true : first_starts;
                Ls[rr] = start; Ll[rr] = end - start; Lin[rr] = sknown && start < valid;
                int64_t lo = t0 + start, hi = t0 + start;
                if (end > start) header_kept(bs, t0 + start, t0 + end, sknown, end_in, t0 + valid, lo, hi);
                Ks[rr] = (int)(lo - t0); Kl[rr] = (int)(hi - lo);
            }
        }
    }
    if (MODE == 0) {
        if (tid == 0) s_bcast[1] = wait_granule(&sa.pc[t], a.st);  // <- service: line index of the tile's first line
    } else if (wave == 3) { // an otherwise idle wave resolves the line index while wave 0 analyses lines
        const u64 mine = own_block_aggregate(sa.dc, t, hi, lane, a.st);
        const int64_t inb = wave_sum((int64_t)mine);
        const int64_t Pex = block_lookback(sa.bc, hb, inb, (int64_t)c, hi == HB - 1 || t == a.n_tiles - 1, a.st->P0, lane, a.st);
        if (lane == 0) {
            s_bcast[1] = Pex;
            if (t == a.n_tiles - 1) a.st->P = Pex + (int64_t)c;
        }
    }
    __syncthreads();
    const int64_t P = s_bcast[1];
    const int ph = (int)(P & 3);
    ErrAcc err{~0ull, ~0ull};
    bool overflow = false;

    auto dense_walk = [&](bool emit, int64_t S, int64_t Q, int64_t I, int64_t& ns, int64_t& nq, int64_t& ni) {
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
                if (emit && sin) {
                    if (s_tile[start] != 64) err.structure(rec, 1);
                    if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_hdr[rec] = ls;
                }
                int64_t lo = ls, hi = ls;
                if (end > start) header_kept(bs, ls, le, start_in, end_in, t0 + valid, lo, hi);
                if (emit)
                    for (int64_t p = lo; p < hi; ++p) {
                        const uint8_t ch = s_tile[p - t0];
                        if (CA && (ch & 0x80)) err.valid(rec, 4);
                        if (ri + (p - lo) >= 0) a.col_id[ri + (p - lo)] = ch;
                    }
                ri += hi - lo;
                if (emit && end_in && rec >= 0) { if (rec < a.rec_cap) a.id_ends[rec] = ri; else overflow = true; }
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
                if (emit && end_in && rec >= 0) {
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
        ns = rs - S; nq = rq - Q; ni = ri - I;
    };

    int n_id = 0, n_seq = 0, n_qual = 0;
    if (dense) {
        if (tid == 0) {
            int64_t ns, nq, ni;
            dense_walk(false, 0, 0, 0, ns, nq, ni);
            s_cnt[0] = (int)ni; s_cnt[1] = (int)ns; s_cnt[2] = (int)nq;
        }
        __syncthreads();
        n_id = s_cnt[0]; n_seq = s_cnt[1]; n_qual = s_cnt[2];
    } else {
        // ---- roles: line 4k+rr has role (ph+rr)&3 for every k (uniform per unrolled step) ----------
        uint32_t lh = 0, lsq = 0, lq = 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int j = 4 * tid + rr;
            const int role = (ph + rr) & 3;
            if (j <= (int)c) {
                const int64_t rec = (P + j) >> 2;
                const int64_t ls = t0 + Ls[rr];
                if (role == 0) {
                    if (Lin[rr]) {
                        if (s_tile[Ls[rr]] != 64) err.structure(rec, 1);   // '@', utils.mojo:454
                        if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_hdr[rec] = ls;
                    }
                    s_src[0][tid] = (uint16_t)Ks[rr];
                    lh = (uint32_t)Kl[rr];
                } else if (role == 1) {
                    if (Lin[rr] && OFFS && rec >= 0 && rec < a.rec_cap) a.o_seq[rec] = ls;
                    s_src[1][tid] = (uint16_t)Ls[rr];
                    lsq = (uint32_t)Ll[rr];
                } else if (role == 2) {
                    if (Lin[rr]) {
                        if (s_tile[Ls[rr]] != 43) err.structure(rec, 2);   // '+', utils.mojo:456
                        if (OFFS && rec >= 0 && rec < a.rec_cap) a.o_sep[rec] = ls;
                    }
                } else {
                    if (Lin[rr] && OFFS && rec >= 0 && rec < a.rec_cap) a.o_qual[rec] = ls;
                    s_src[2][tid] = (uint16_t)Ls[rr];
                    lq = (uint32_t)Ll[rr];
                }
            }
        }
        s_len[0][tid] = (uint16_t)lh; s_len[1][tid] = (uint16_t)lsq; s_len[2][tid] = (uint16_t)lq;
        const u64 packed = (u64)lh | ((u64)lsq << 21) | ((u64)lq << 42);
        u64 tot = 0;
        const u64 ex = block_exclusive_scan<u64, 4>(packed, s_w64, tot);
        s_dst[0][tid] = (uint16_t)(ex & 0x1FFFFFull);
        s_dst[1][tid] = (uint16_t)((ex >> 21) & 0x1FFFFFull);
        s_dst[2][tid] = (uint16_t)((ex >> 42) & 0x1FFFFFull);
        n_id = (int)(tot & 0x1FFFFFull); n_seq = (int)((tot >> 21) & 0x1FFFFFull); n_qual = (int)((tot >> 42) & 0x1FFFFFull);
    }
    if (MODE == 0) {
        if (tid == 0) {
            st_agent(&sa.da[t], DESC_A | (u64)n_seq | ((u64)n_qual << 20) | ((u64)n_id << 40));   // -> service (columns)
            if (c > 0) atomicMax((long long*)&a.st->last_nl_tile, (long long)t);
            s_bcast[1] = wait_granule(&sa.ps[t], a.st);                                              // <- service
            s_bcast[2] = wait_granule(&sa.pq[t], a.st);
            s_bcast[3] = wait_granule(&sa.pi[t], a.st);
        }
    } else if (wave == 3) {
        if (lane == 0) {
            st_agent(&sa.da[t], DESC_A | (u64)n_seq | ((u64)n_qual << 20) | ((u64)n_id << 40));
            if (c > 0) atomicMax((long long*)&a.st->last_nl_tile, (long long)t);
        }
        const u64 mine = own_block_aggregate(sa.da, t, hi, lane, a.st);
        const int64_t inS = wave_sum((int64_t)(mine & 0xFFFFFull)), inQ = wave_sum((int64_t)((mine >> 20) & 0xFFFFFull)),
                      inI = wave_sum((int64_t)((mine >> 40) & 0xFFFFFull));
        const bool last = hi == HB - 1 || t == a.n_tiles - 1;
        const int64_t Sx = block_lookback(sa.bs, hb, inS, n_seq, last, a.st->S0, lane, a.st);
        const int64_t Qx = block_lookback(sa.bq, hb, inQ, n_qual, last, a.st->Q0, lane, a.st);
        const int64_t Ix = block_lookback(sa.bi, hb, inI, n_id, last, a.st->I0, lane, a.st);
        if (lane == 0) {
            s_bcast[1] = Sx; s_bcast[2] = Qx; s_bcast[3] = Ix;
            if (t == a.n_tiles - 1) { a.st->S = Sx + n_seq; a.st->Q = Qx + n_qual; a.st->I = Ix + n_id; }
        }
    }
    __syncthreads();
    const int64_t S = s_bcast[1], Q = s_bcast[2], I = s_bcast[3];

    if (dense) {
        if (tid == 0) {
            int64_t ns, nq, ni;
            dense_walk(true, S, Q, I, ns, nq, ni);
            atomicAdd((u64*)&a.st->dense_tiles, 1ull);
        }
    } else {
        const int jh = (0 - ph) & 3, jq = (3 - ph) & 3;
        {
            const int j = 4 * tid + jh;
            const int64_t rec = (P + j) >> 2;
            if (j < (int)c && rec >= 0) {
                if (rec < a.rec_cap) a.id_ends[rec] = I + (int64_t)s_dst[0][tid] + (int64_t)s_len[0][tid];
                else overflow = true;
            }
        }
        {
            const int j = 4 * tid + jq;
            const int64_t rec = (P + j) >> 2;
            if (j < (int)c && rec >= 0) {
                const int64_t qe = Q + (int64_t)s_dst[2][tid] + (int64_t)s_len[2][tid];
                int64_t se = S;
                if (jq >= 2) se = S + (int64_t)s_dst[1][tid] + (int64_t)s_len[1][tid];
                else if (tid > 0) se = S + (int64_t)s_dst[1][tid - 1] + (int64_t)s_len[1][tid - 1];
                if (rec < a.rec_cap) { a.ends[rec] = qe; a.rec_end[rec] = t0 + (int64_t)s_nl[j]; }
                else overflow = true;
                if (se != qe) err.structure(rec, 3);
            }
        }
        // ---- scatter (same as k_fused) ------------------------------------------------------------
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int q = tid + BLOCK * sidx;
            const int pos = q * 16;
            if (pos < valid) {
                const int j = (int)s_pline[q];
                const int role = (ph + j) & 3;
                const int k = j >> 2;
                const int slot = role == 0 ? 0 : (role == 1 ? 1 : 2);
                const int src = (int)s_src[slot][k], len = (int)s_len[slot][k], dd = (int)s_dst[slot][k];
                const int64_t base = role == 0 ? I : (role == 1 ? S : Q);
                uint8_t* col = role == 0 ? a.col_id : (role == 1 ? a.col_seq : a.col_qual);
                const int64_t addr = base + dd + (pos - src);
                if (role != 2 && pos >= src && pos + 16 <= src + len && base + dd >= 0) {
                    const int64_t rec = (P + j) >> 2;
                    if (CA && any_non_ascii(r[sidx].x | r[sidx].y | r[sidx].z | r[sidx].w)) err.valid(rec, 4);
                    if (CQ && role == 3 &&
                        (any_out_of_range(r[sidx].x, a.q_lower, a.q_upper) | any_out_of_range(r[sidx].y, a.q_lower, a.q_upper) |
                         any_out_of_range(r[sidx].z, a.q_lower, a.q_upper) | any_out_of_range(r[sidx].w, a.q_lower, a.q_upper)))
                        err.valid(rec, 5);
                    U16B v{r[sidx].x, r[sidx].y, r[sidx].z, r[sidx].w};
                    *reinterpret_cast<U16B*>(col + addr) = v;
                }
            }
        }
        {
            const int jr[3] = {jh, (1 - ph) & 3, jq};
#pragma unroll
            for (int slot = 0; slot < 3; ++slot) {
                const int len = (int)s_len[slot][tid];
                const int64_t base = slot == 0 ? I : (slot == 1 ? S : Q);
                const int64_t d0 = base + (int64_t)s_dst[slot][tid];
                if (len > 0 && d0 >= 0) {
                    uint8_t* col = slot == 0 ? a.col_id : (slot == 1 ? a.col_seq : a.col_qual);
                    const int src = (int)s_src[slot][tid];
                    const int64_t rec = (P + 4 * tid + jr[slot]) >> 2;
                    const int au = (src + 15) & ~15;
                    const int a1 = au < src + len ? au : src + len;
                    const int ad = (src + len) & ~15;
                    const int b0 = ad > a1 ? ad : a1;
                    if (slot == 0) {
                        emit_part<0, CA, CQ>(col, d0, src, a1 - src, s_tile, rec, a.q_lower, a.q_upper, err);
                        emit_part<0, CA, CQ>(col, d0 + (b0 - src), b0, src + len - b0, s_tile, rec, a.q_lower, a.q_upper, err);
                    } else if (slot == 1) {
                        emit_part<1, CA, CQ>(col, d0, src, a1 - src, s_tile, rec, a.q_lower, a.q_upper, err);
                        emit_part<1, CA, CQ>(col, d0 + (b0 - src), b0, src + len - b0, s_tile, rec, a.q_lower, a.q_upper, err);
                    } else {
                        emit_part<3, CA, CQ>(col, d0, src, a1 - src, s_tile, rec, a.q_lower, a.q_upper, err);
                        emit_part<3, CA, CQ>(col, d0 + (b0 - src), b0, src + len - b0, s_tile, rec, a.q_lower, a.q_upper, err);
                    }
                }
            }
        }
    }
    if (err.e_struct != ~0ull) atomicMin(&a.st->err_struct, err.e_struct);
    if (err.e_valid != ~0ull) atomicMin(&a.st->err_valid, err.e_valid);
    if (overflow) atomicOr(&a.st->rec_overflow, 1);
}

} // namespace bzq
