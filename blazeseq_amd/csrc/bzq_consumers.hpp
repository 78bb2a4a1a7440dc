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

__device__ __forceinline__ int wave_prefix_max(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (lane >= off) v = o > v ? o : v;
    }
    return v;
}

// One wave per record; 4 records per workgroup.
__global__ __launch_bounds__(BLOCK) void k_nw_scores(const uint8_t* __restrict__ ref, int ref_len,
                                                      const uint8_t* __restrict__ seq, const int64_t* __restrict__ ends,
                                                      int64_t num_records, int32_t* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int64_t rec = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (rec >= num_records) return;
    const int64_t q0 = rec ? ends[rec - 1] : 0;
    const int64_t qlen64 = ends[rec] - q0;
    if (qlen64 > NW_MAX_LEN || ref_len > NW_MAX_LEN) {   // kernels.mojo:48-50
        if (lane == 0) scores[rec] = 0;
        return;
    }
    const int qlen = (int)qlen64;
    const int nseg = (ref_len + 63) >> 6;   // reference position i = 64*s + lane + 1
    uint32_t rb[4];
    int prev[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = 64 * s + lane + 1;
        rb[s] = i <= ref_len ? ref[i - 1] : 0x100u;   // never equal to a query byte
        prev[s] = -i;                                   // first row: gap * i (kernels.mojo:61-62)
    }
    int prev0 = 0;   // dp[j-1][0]
    for (int j = 1; j <= qlen; ++j) {
        const uint32_t qb = seq[q0 + j - 1];   // uniform across the wave
        const int curr0 = -j;                  // dp[j][0] = gap * j
        int carry = curr0;                     // max over positions k < this segment of (curr-candidate[k] + k)
        int left_edge = prev0;                 // dp[j-1][i-1] for lane 0 of the segment
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < nseg) {
                const int i = 64 * s + lane + 1;
                int diag = __shfl_up(prev[s], 1, 64);
                if (lane == 0) diag = left_edge;
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

// One wave per record: sum over the record's quality bytes of (byte - offset), as int64.
__global__ __launch_bounds__(BLOCK) void k_quality_sums(const uint8_t* __restrict__ qual, const int64_t* __restrict__ ends,
                                                         int64_t num_records, int offset, int64_t* __restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const int64_t rec = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (rec >= num_records) return;
    const int64_t q0 = rec ? ends[rec - 1] : 0, q1 = ends[rec];
    int64_t acc = 0;
    for (int64_t p = q0 + lane; p < q1; p += 64) acc += (int64_t)qual[p] - offset;
    const u64 tot = wave_sum_u64((u64)acc);
    if (lane == 0) sums[rec] = (int64_t)tot;
}

// hist[256] += byte counts of col[0, n).  Grid-stride, 16 bytes per lane per step; one LDS histogram per wave.
__global__ __launch_bounds__(BLOCK) void k_byte_histogram(const uint8_t* __restrict__ col, int64_t n, u64* __restrict__ hist) {
    __shared__ uint32_t s_h[BLOCK / 64][256];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < (BLOCK / 64) * 256; i += BLOCK) (&s_h[0][0])[i] = 0u;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * BLOCK * 16;
    for (int64_t base = ((int64_t)blockIdx.x * BLOCK + tid) * 16; base < n; base += stride) {
        if (base + 16 <= n) {
            const uint4 v = load16_any(col + base);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                atomicAdd(&s_h[wave][w[k] & 0xFFu], 1u);
                atomicAdd(&s_h[wave][(w[k] >> 8) & 0xFFu], 1u);
                atomicAdd(&s_h[wave][(w[k] >> 16) & 0xFFu], 1u);
                atomicAdd(&s_h[wave][w[k] >> 24], 1u);
            }
        } else {
            for (int64_t p = base; p < n; ++p) atomicAdd(&s_h[wave][col[p]], 1u);
        }
    }
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) t += s_h[w][tid];
    if (t) atomicAdd(&hist[tid], (u64)t);
}

} // namespace bzq
