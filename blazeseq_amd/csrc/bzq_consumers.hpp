// Device-side consumers of a DeviceFastqBatch: they read the SoA columns where the parser left
// them -- no host round trip -- through the same contract as the reference's example kernels
// (examples/nw_gpu/kernels.mojo:41-46: record r's bytes are [ends[r-1], ends[r]) of the sequence / quality buffer).
//
//   k_nw_scores        the nw_gpu example (examples/nw_gpu/kernels.mojo:21-89): global alignment score of every read
//                      against one reference, match +1 / mismatch -1 / gap -1.  The reference launches ONE THREAD per
//                      record with the two DP rows in global memory; here one wave64 owns a record, the DP row lives in
//                      registers (reference position = lane), and the serial insertion chain
//                      curr[i] = max(m[i], curr[i-1] - 1) becomes a wave prefix maximum of m[i] + i.
//   k_quality_sums     per-record sum of Phred scores (quality byte - offset): the "quality prefix-sum" consumer of
//                      the reference's v0.1 GPU path (CHANGELOG.md:73), as one segmented reduction per record.
//   k_byte_histogram   256-bin histogram of a column (base composition of the sequence column, quality distribution
//                      of the quality column): per-wave LDS histograms, 16 bytes per lane per step.
#pragma once

namespace bzq {

constexpr int NW_MAX_LEN = 256;   // MAX_REF_LEN / MAX_QUERY_LEN, examples/nw_gpu/kernels.mojo:15-16

// Inclusive prefix maximum over the wave on DPP (row_shr 1/2/4/8 inside rows of 16, then row_bcast:15 / :31 across
// rows), the max-analogue of dpp_scan_u32: lanes without a source keep the identity INT_MIN.
__device__ __forceinline__ int wave_prefix_max(int v, int lane) {
    (void)lane;
    constexpr int ID = (int)0x80000000;
#define BZQ_DPP_MAX(ctrl, rm)                                                           \
    do {                                                                                \
        const int o_ = __builtin_amdgcn_update_dpp(ID, v, ctrl, rm, 0xf, false);        \
        v = o_ > v ? o_ : v;                                                            \
    } while (0)
    BZQ_DPP_MAX(0x111, 0xf); BZQ_DPP_MAX(0x112, 0xf); BZQ_DPP_MAX(0x114, 0xf); BZQ_DPP_MAX(0x118, 0xf);
    BZQ_DPP_MAX(0x142, 0xa); BZQ_DPP_MAX(0x143, 0xc);
#undef BZQ_DPP_MAX
    return v;
}

// One wave per record; 4 records per workgroup.
static __global__ __launch_bounds__(BLOCK) void k_nw_scores(const uint8_t* __restrict__ ref, int ref_len,
                                                      const uint8_t* __restrict__ seq, const int64_t* __restrict__ ends,
                                                      int64_t num_records, int32_t* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int64_t rec = (int64_t)blockIdx.x * (BLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (uniform: scalar loads, scalar loop control)
    if (rec >= num_records) return;
    const int64_t q0 = rec ? ends[rec - 1] : 0;
    const int64_t qlen64 = ends[rec] - q0;
    if (qlen64 > NW_MAX_LEN || ref_len > NW_MAX_LEN) {   // kernels.mojo:48-50
        if (lane == 0) scores[rec] = 0;
        return;
    }
    const int qlen = __builtin_amdgcn_readfirstlane((int)qlen64);
    const int nseg = (ref_len + 63) >> 6;   // reference position i = 64*s + lane + 1
    uint32_t rb[4];
    int prev[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = 64 * s + lane + 1;
        rb[s] = i <= ref_len ? ref[i - 1] : 0x100u;   // never equal to a query byte
        prev[s] = -i;                                   // first row: gap * i (kernels.mojo:61-62)
    }
    // The query sits in registers, base 64 s + lane in qv[s] (four coalesced loads up front), and row j takes its base by v_readlane:
    // a load of seq[q0 + j - 1] at the top of every row put one memory latency on each of the 150 serial rows of a record (round 6:
    // the kernel was latency bound, ~75 us per record and wave; profiles/r6_pipeline.md).
    uint32_t qv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { const int k = 64 * s + lane; qv[s] = k < qlen ? (uint32_t)seq[q0 + k] : 0u; }
    int prev0 = 0;   // dp[j-1][0]
    for (int j = 1; j <= qlen; ++j) {
        const int jb = j - 1;                  // (uniform: the row counter)
        const uint32_t qsel = jb < 64 ? qv[0] : jb < 128 ? qv[1] : jb < 192 ? qv[2] : qv[3];
        const uint32_t qb = (uint32_t)__builtin_amdgcn_readlane((int)qsel, jb & 63);
        const int curr0 = -j;                  // dp[j][0] = gap * j
        int carry = curr0;                     // max over positions k < this segment of (curr-candidate[k] + k)
        int left_edge = prev0;                 // dp[j-1][i-1] for lane 0 of the segment
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < nseg) {
                const int i = 64 * s + lane + 1;
                int diag = __builtin_amdgcn_update_dpp(left_edge, prev[s], 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 keeps left_edge
                left_edge = __builtin_amdgcn_readlane(prev[s], 63);
                const int m0 = diag + (rb[s] == qb ? 1 : -1);
                const int m1 = prev[s] - 1;                       // deletion: dp[j-1][i] + gap
                const int m = m0 > m1 ? m0 : m1;
                int u = wave_prefix_max(m + i, lane);             // insertion chain as a prefix maximum
                u = u > carry ? u : carry;
                carry = __builtin_amdgcn_readlane(u, 63);
                prev[s] = u - i;
            }
        }
        prev0 = curr0;
    }
    int out = prev0;   // ref_len == 0
    if (ref_len > 0) {
        const int s = (ref_len - 1) >> 6, l = (ref_len - 1) & 63;
        int v = prev[0];
        if (s == 1) v = prev[1]; else if (s == 2) v = prev[2]; else if (s == 3) v = prev[3];
        out = __shfl(v, l, 64);
    }
    if (lane == 0) scores[rec] = out;
}

// The same scores with ONE THREAD per record, for references of at most 64 bases (the example's has 40).  The wave-per-record kernel
// above spends ~25 vector instructions on a ROW of the table (a prefix maximum over the lanes for the insertion chain) whatever the
// reference's length: 150 rows x 25 = 3 750 wave instructions per record, and the file -> records -> scores pipeline was bound by it
// (profiles/r6_pipeline.md).  Here a lane owns a record and keeps the table's current row in registers, and the recurrence is
// normalised so that a cell costs four instructions: with R[j][i] = H[j][i] + i + j (H = the example's table, gap = -1) the two gap
// moves cost NOTHING and the diagonal adds 3 for a match and 1 for a mismatch,
//     R[j][i] = max3(R[j-1][i-1] + (ref[i] == q[j] ? 3 : 1), R[j-1][i], R[j][i-1]),     R[0][i] = R[j][0] = 0,
// so a row is RL x (compare, select, add, max3) and the score is R[qlen][ref_len] - ref_len - qlen.  RL cells per row (a template
// parameter >= ref_len: the row lives in registers, indexed statically; cells beyond the reference compare against a byte that is
// never a base).  Lanes whose record is shorter than the wave's longest one sit out the remaining rows (EXEC mask, no per-cell cost).
// 64 records per wave: 40 x 150 x 4 = 24 000 wave instructions per 64 records = 375 per record, a tenth of the kernel above.
template <int RL>
static __global__ __launch_bounds__(BLOCK) void k_nw_scores_t(const uint8_t* __restrict__ ref, int ref_len, const uint8_t* __restrict__ seq,
                                                              const int64_t* __restrict__ ends, int64_t num_records, int64_t col_len,
                                                              int32_t* __restrict__ scores) {
    const int64_t rec = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = rec < num_records;
    int64_t q0 = 0;
    int qlen = 0;
    bool zero = false;   // kernels.mojo:48-50: a query longer than 256 scores 0
    if (live) {
        q0 = rec ? ends[rec - 1] : 0;
        const int64_t l = ends[rec] - q0;
        zero = l > NW_MAX_LEN;
        qlen = zero ? 0 : (int)l;
    }
    uint32_t rb[RL];
#pragma unroll
    for (int i = 0; i < RL; ++i) rb[i] = i < ref_len ? (uint32_t)ref[i] : 0x100u;   // (uniform: scalar loads)
    int row[RL + 1];
#pragma unroll
    for (int i = 0; i <= RL; ++i) row[i] = 0;
    int longest = qlen;   // rows the wave walks: its longest record
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(longest, off, 64); longest = o > longest ? o : longest; }
    longest = __builtin_amdgcn_readfirstlane(longest);
    for (int j0 = 0; j0 < longest; j0 += 4) {
        uint32_t q4 = 0;   // this lane's next four bases
        if (j0 < qlen) {
            const int64_t p = q0 + j0;
            if (p + 4 <= col_len) { struct __attribute__((packed, aligned(1))) U4 { uint32_t v; }; q4 = reinterpret_cast<const U4*>(seq + p)->v; }
            else for (int k = 0; k < 4 && p + k < col_len; ++k) q4 |= (uint32_t)seq[p + k] << (8 * k);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (j0 + jj < qlen) {
                const uint32_t qb = (q4 >> (8 * jj)) & 0xFFu;
                int diag = 0;   // R[j-1][0]
#pragma unroll
                for (int i = 1; i <= RL; ++i) {
                    const int up = row[i];
                    const int d = diag + (rb[i - 1] == qb ? 3 : 1);
                    const int m = up > row[i - 1] ? up : row[i - 1];
                    row[i] = d > m ? d : m;
                    diag = up;
                }
            }
        }
    }
    if (live) {
        int out = row[0];
#pragma unroll
        for (int i = 1; i <= RL; ++i) out = i == ref_len ? row[i] : out;
        scores[rec] = zero ? 0 : out - ref_len - qlen;
    }
}

// Sum of the bytes of four dwords (v_sad_u8 against zero: four bytes per instruction).
__device__ __forceinline__ uint32_t sum_bytes16(uint4 v, uint32_t acc) {
    acc = __builtin_amdgcn_sad_u8(v.x, 0u, acc);
    acc = __builtin_amdgcn_sad_u8(v.y, 0u, acc);
    acc = __builtin_amdgcn_sad_u8(v.z, 0u, acc);
    return __builtin_amdgcn_sad_u8(v.w, 0u, acc);
}
// 16 bytes at col[p..] where only bytes below col_len may be touched (the batch may end exactly at an allocation's end)
__device__ __forceinline__ uint4 load16_tail(const uint8_t* __restrict__ col, int64_t p, int64_t col_len) {
    if (p + 16 <= col_len) return load16_any(col + p);
    u64 lo = 0, hi = 0;
#pragma unroll 1
    for (int i = 0; p + i < col_len && i < 16; ++i) {
        const u64 b = col[p + i];
        if (i < 8) lo |= b << (8 * i); else hi |= b << (8 * (i - 8));
    }
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
// keep the first k bytes (0..16) of a 16-byte value
__device__ __forceinline__ uint4 keep_first(uint4 v, int k) {
    const u64 lo = (u64)v.x | ((u64)v.y << 32), hi = (u64)v.z | ((u64)v.w << 32);
    const u64 ml = k >= 8 ? ~0ull : ((1ull << (8 * k)) - 1ull);
    const u64 mh = k >= 16 ? ~0ull : (k > 8 ? ((1ull << (8 * (k - 8))) - 1ull) : 0ull);
    const u64 l = lo & ml, h = hi & mh;
    return make_uint4((uint32_t)l, (uint32_t)(l >> 32), (uint32_t)h, (uint32_t)(h >> 32));
}

// 1 in every byte that is G, C, g or c; 0 elsewhere (exact zero-byte test on x ^ 'g' and x ^ 'c' after folding the case bit)
__device__ __forceinline__ uint32_t gc_ones32(uint32_t x) {
    const uint32_t y = x | 0x20202020u, a = y ^ 0x67676767u, b = y ^ 0x63636363u;
    const uint32_t fa = ~(((a & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | a | 0x7F7F7F7Fu), fb = ~(((b & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | b | 0x7F7F7F7Fu);
    return (fa | fb) >> 7;
}
__device__ __forceinline__ uint4 gc_ones(uint4 v) { return make_uint4(gc_ones32(v.x), gc_ones32(v.y), gc_ones32(v.z), gc_ones32(v.w)); }

// Per-record sum of (quality byte - offset) -- or, with GC, per-record count of G/C bases (GC / base
// counts; any byte column with inclusive running sums as record boundaries: a FastqBatch's or a FASTA chunk's sequence
// column).  SHORT reads: one workgroup per 256 consecutive records.  Their bytes are
// one contiguous span of the column, read fully coalesced (16 bytes per lane per step); each 16-byte piece finds its
// record by a binary search over the 257 record boundaries kept in LDS, is split where it straddles boundaries, and is
// added to that record's LDS accumulator (v_sad_u8 sums four bytes per instruction).  A thread-per-record version with
// the same loads ran at 1.07 TB/s: neighbouring lanes are a record (150 B) apart, so every line was fetched ~8 times.
template <bool GC>
static __global__ __launch_bounds__(BLOCK) void k_quality_sums_block(const uint8_t* __restrict__ qual, const int64_t* __restrict__ ends,
                                                               int64_t num_records, int64_t col_len, int offset,
                                                               int64_t* __restrict__ sums) {
    __shared__ int64_t s_e[BLOCK + 1];   // s_e[k] = start of record r0 + k
    __shared__ u64 s_sum[BLOCK];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * BLOCK;
    const int nrec = (int)((num_records - r0) < BLOCK ? (num_records - r0) : BLOCK);
    if (tid < nrec) s_e[tid] = (r0 + tid) ? ends[r0 + tid - 1] : 0;
    if (tid == 0) s_e[nrec] = ends[r0 + nrec - 1];   // 257 boundaries, 256 threads
    s_sum[tid] = 0;
    __syncthreads();
    const int64_t b0 = s_e[0], b1 = s_e[nrec];
    // pieces are aligned to 16 bytes of the COLUMN (not of the span) so that loads are aligned whenever the column is
    for (int64_t p = (b0 & ~(int64_t)15) + 16 * tid; p < b1; p += 16 * BLOCK) {
        uint4 v = load16_tail(qual, p < 0 ? 0 : p, col_len);
        int64_t lo = p < b0 ? b0 : p;                        // first byte of this piece inside the span
        const int64_t hi = p + 16 < b1 ? p + 16 : b1;        // one past its last byte inside the span
        // record containing byte lo: largest k with s_e[k] <= lo (records may be empty: take the LAST such k)
        int k = 0;
        for (int step = BLOCK / 2; step > 0; step >>= 1)
            if (k + step <= nrec - 1 && s_e[k + step] <= lo) k += step;
        while (lo < hi) {
            while (s_e[k + 1] <= lo) ++k;                    // skip empty records
            const int64_t e = s_e[k + 1] < hi ? s_e[k + 1] : hi;
            // bytes [lo, e) of the piece: drop (lo - p) leading bytes, keep (e - lo)
            const int drop = (int)(lo - p), keep = (int)(e - lo);
            const u64 vlo = (u64)v.x | ((u64)v.y << 32), vhi = (u64)v.z | ((u64)v.w << 32);
            u64 slo, shi;
            if (drop == 0) { slo = vlo; shi = vhi; }
            else if (drop < 8) { slo = (vlo >> (8 * drop)) | (vhi << (64 - 8 * drop)); shi = vhi >> (8 * drop); }
            else { slo = vhi >> (8 * (drop - 8)); shi = 0; }
            const uint4 part = keep_first(make_uint4((uint32_t)slo, (uint32_t)(slo >> 32), (uint32_t)shi, (uint32_t)(shi >> 32)), keep);
            atomicAdd(&s_sum[k], (u64)sum_bytes16(GC ? gc_ones(part) : part, 0u));
            lo = e;
        }
    }
    __syncthreads();
    if (tid < nrec) sums[r0 + tid] = (int64_t)s_sum[tid] - (int64_t)offset * (s_e[tid + 1] - s_e[tid]);
}

// LONG reads: one wave per record, 16 bytes per lane per step (1 KiB per wave instruction).
template <bool GC>
static __global__ __launch_bounds__(BLOCK) void k_quality_sums(const uint8_t* __restrict__ qual, const int64_t* __restrict__ ends,
                                                         int64_t num_records, int64_t col_len, int offset, int64_t* __restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const int64_t rec = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (rec >= num_records) return;
    const int64_t q0 = rec ? ends[rec - 1] : 0, q1 = ends[rec];
    int64_t acc64 = 0;
    uint32_t acc = 0;
    int steps = 0;
    for (int64_t p = q0 + 16 * lane; p < q1; p += 1024) {
        const uint4 v = load16_tail(qual, p, col_len);
        const uint4 u = p + 16 <= q1 ? v : keep_first(v, (int)(q1 - p));
        acc = sum_bytes16(GC ? gc_ones(u) : u, acc);
        if (++steps == (1 << 19)) { acc64 += acc; acc = 0; steps = 0; }
    }
    acc64 += acc;
    const u64 tot = wave_sum_u64((u64)acc64);
    if (lane == 0) sums[rec] = (int64_t)tot - (int64_t)offset * (q1 - q0);
}

// hist[256] += byte counts of col[0, n).  Grid-stride, 16 bytes per lane per step.  LDS histograms: one per wave AND
// per lane-residue mod 8 (a 5-letter alphabet would otherwise pile 64 lanes onto 5 addresses); bank = bin + replica.
constexpr int HIST_REP = 8;
static __global__ __launch_bounds__(BLOCK) void k_byte_histogram(const uint8_t* __restrict__ col, int64_t n, u64* __restrict__ hist) {
    __shared__ uint32_t s_h[BLOCK / 64][HIST_REP][256 + 1];   // +1: replicas of one bin land in different banks
    const int tid = threadIdx.x, wave = tid >> 6, rep = tid & (HIST_REP - 1);
    for (int i = tid; i < (BLOCK / 64) * HIST_REP * 257; i += BLOCK) (&s_h[0][0][0])[i] = 0u;
    __syncthreads();
    uint32_t* h = &s_h[wave][rep][0];
    const int64_t stride = (int64_t)gridDim.x * BLOCK * 16;
    for (int64_t base = ((int64_t)blockIdx.x * BLOCK + tid) * 16; base < n; base += stride) {
        if (base + 16 <= n) {
            const uint4 v = load16_any(col + base);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&h[w[k] & 0xFFu], 1u);
                atomicAdd(&h[(w[k] >> 8) & 0xFFu], 1u);
                atomicAdd(&h[(w[k] >> 16) & 0xFFu], 1u);
                atomicAdd(&h[w[k] >> 24], 1u);
            }
        } else {
            for (int64_t p = base; p < n; ++p) atomicAdd(&h[col[p]], 1u);
        }
    }
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w)
#pragma unroll
        for (int r = 0; r < HIST_REP; ++r) t += s_h[w][r][tid];
    if (t) atomicAdd(&hist[tid], (u64)t);
}

// Quality distribution per READ POSITION (cycle): counts[p][v] = number of records whose quality byte at position p is v
// (the per-base quality plot of every FASTQ QC tool; the v0.1 `quality_distribution` example / quality prefix-sum kernel of
// the reference, CHANGELOG.md:73, on the DeviceFastqBatch layout).  A workgroup takes 64 consecutive positions and a block of
// records: lane = position, so a wave reads 64 contiguous quality bytes of one record per step (coalesced) and every lane
// adds into ITS OWN row of an LDS table -- rows are 129 words apart, which puts equal values of different lanes into different
// banks.  Bytes >= 128 count in bin 127.  One flush of the table per workgroup (global atomics on 64 x 128 counters).
constexpr int QP_POS = 64, QP_BINS = 128, QP_RECS = 2048;
static __global__ __launch_bounds__(BLOCK) void k_quality_by_position(const uint8_t* __restrict__ qual, const int64_t* __restrict__ ends,
                                                                      int64_t num_records, int max_pos, u64* __restrict__ counts) {
    __shared__ uint32_t s_h[QP_POS][QP_BINS + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < QP_POS * (QP_BINS + 1); i += BLOCK) (&s_h[0][0])[i] = 0u;
    __syncthreads();
    const int p = (int)blockIdx.x * QP_POS + lane;
    const int64_t r0 = (int64_t)blockIdx.y * QP_RECS, r1 = r0 + QP_RECS < num_records ? r0 + QP_RECS : num_records;
    for (int64_t r = r0 + wave; r < r1; r += BLOCK / 64) {
        const int64_t q0 = r ? ends[r - 1] : 0, q1 = ends[r];
        if (p < max_pos && q0 + p < q1) {
            const uint32_t v = qual[q0 + p];
            atomicAdd(&s_h[lane][v < 128u ? v : 127u], 1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < QP_POS * QP_BINS; i += BLOCK) {
        const int row = i / QP_BINS, bin = i % QP_BINS;
        const uint32_t t = s_h[row][bin];
        const int pp = (int)blockIdx.x * QP_POS + row;
        if (t && pp < max_pos) atomicAdd(&counts[(int64_t)pp * QP_BINS + bin], (u64)t);
    }
}

} // namespace bzq
