// match_kernels.hip — gfx950 Hamming matching kernels.
//
// Two kernels:
//   topk_window_kernel  one wave per query: scan the candidate keypoints in the grid columns GetFeaturesInArea
//                       would visit (code/src/Frame.cc:377-431; candidates are stored column by column, so that is
//                       one contiguous range, 64 candidates per step, coalesced SoA reads), apply its row / window /
//                       level tests as a per-pair mask, compute 256-bit Hamming distances for the survivors, and
//                       return the K best in the order the reference's sequential "dist < bestDist" scan induces:
//                       (distance, grid traversal rank).
//   hamming_top2_kernel one wave per query against every row of B (cross-agent keyframe search), best/second.
//
// Candidates are stored in grid-traversal order (cell x, cell y, keypoint index) by the host, so a candidate's
// array position IS its tie-break rank and key = dist << 16 | position sorts exactly like the reference visits.
#include "match_device.h"
#include <algorithm>

namespace so {

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
    // DescriptorDistance (code/src/ORBmatcher.cc:1511-1525) == popcount(a ^ b) over 256 bits (KAT-pinned)
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

constexpr int kListCap = 1024;  // per-wave LDS list of (dist<<16 | rank) keys

// The cells GetFeaturesInArea visits for one query (Frame.cc:382-396, KeyFrame.cc:784-798): columns as a range of
// candidate positions, rows as [cy0, cy1].
struct CellWindow {
    int lo, hi, cy0, cy1;
};

__device__ __forceinline__ CellWindow cell_window(const MatchFrameDev& F, const MatchQuery& Q) {
    CellWindow w;
    w.lo = 0; w.hi = F.n; w.cy0 = 0; w.cy1 = kMatchGridRows - 1;
    if (Q.flags & kQRange) {
        w.lo = Q.c_begin; w.hi = Q.c_end;
    } else if (F.col_start) {
        const int cx0 = max(0, (int)floorf((Q.u - F.min_x - Q.r) * F.grid_inv_w));
        const int cx1 = min(kMatchGridCols - 1, (int)ceilf((Q.u - F.min_x + Q.r) * F.grid_inv_w));
        w.cy0 = max(0, (int)floorf((Q.v - F.min_y - Q.r) * F.grid_inv_h));
        w.cy1 = min(kMatchGridRows - 1, (int)ceilf((Q.v - F.min_y + Q.r) * F.grid_inv_h));
        if (cx0 >= kMatchGridCols || cx1 < 0 || w.cy0 >= kMatchGridRows || w.cy1 < 0 || cx0 > cx1) {
            w.lo = w.hi = 0;
        } else {
            w.lo = F.col_start[cx0];
            w.hi = F.col_start[cx1 + 1];
        }
    }
    return w;
}

// Geometric / structural predicate of one (query, candidate) pair — everything except the descriptor distance.
__device__ __forceinline__ bool pair_pred(const MatchFrameDev& F, const MatchQuery& Q, const CellWindow& W, int c,
                                          bool check_levels) {
    const float2 xy = F.xy[c];
    const int o = F.octave[c];
    bool ok = true;
    if (!(Q.flags & kQRange)) {  // Frame::GetFeaturesInArea, code/src/Frame.cc:377-431
        const float dx = xy.x - Q.u, dy = xy.y - Q.v;
        ok = fabsf(dx) < Q.r && fabsf(dy) < Q.r;
        if (F.col_start) {  // the candidate's cell row (Frame::PosInGrid, :433-443) must be one of the visited rows
            const int py = (int)roundf((xy.y - F.min_y) * F.grid_inv_h);
            if (py < W.cy0 || py > W.cy1) ok = false;
        }
        if (check_levels) {
            if (o < Q.min_level) ok = false;
            if (Q.max_level >= 0 && o > Q.max_level) ok = false;
        }
    }
    if (ok && (Q.flags & kQChi2Gate)) {  // ORBmatcher::Fuse, code/src/ORBmatcher.cc:853-860
        const float ex = Q.u - xy.x, ey = Q.v - xy.y;
        const float e2 = ex * ex + ey * ey;
        if ((double)(e2 * F.inv_sigma2[o]) > 5.99) ok = false;
    }
    if (ok && (Q.flags & kQEpipolar)) {  // SearchForTriangulation :693-706 + CheckDistEpipolarLine :131-148
        const float distex = F.ex - xy.x, distey = F.ey - xy.y;
        if (distex * distex + distey * distey < 100 * F.scale[o]) ok = false;
        const float num = Q.la * xy.x + Q.lb * xy.y + Q.lc;
        const float den = Q.la * Q.la + Q.lb * Q.lb;
        if (den == 0) ok = false;
        const float dsqr = num * num / den;
        if (!((double)dsqr < 3.84 * (double)F.sigma2[o])) ok = false;
    }
    return ok;
}

__device__ __forceinline__ uint32_t make_key(const MatchQuery& Q, int dist, int c) {
    return ((uint32_t)dist << 16) | (uint32_t)((Q.flags & kQPreferLast) ? (0xFFFF - c) : c);
}

template <bool COMPACT>
__global__ __launch_bounds__(256) void topk_window_kernel(MatchFrameDev F, const void* __restrict__ q,
                                                           const uint4* __restrict__ qdesc, int nq, int K,
                                                           uint32_t* __restrict__ out_keys,
                                                           int32_t* __restrict__ out_count) {
    __shared__ uint32_t s_keys[4][kListCap];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int qi = blockIdx.x * 4 + w;
    if (qi >= nq) return;
    MatchQuery Q;
    if (COMPACT) {
        const MatchQueryW c = static_cast<const MatchQueryW*>(q)[qi];
        Q = MatchQuery{};
        Q.u = c.u; Q.v = c.v; Q.r = c.r;
        Q.min_level = c.min_level; Q.max_level = c.max_level;
        Q.active = c.active;
        Q.max_dist = 256;
    } else {
        Q = static_cast<const MatchQuery*>(q)[qi];
    }
    if (!Q.active) {
        if (lane == 0) out_count[qi] = 0;
        for (int k = lane; k < K; k += 64) out_keys[(size_t)qi * K + k] = 0xFFFFFFFFu;
        return;
    }
    const uint4 qd0 = qdesc[2 * qi], qd1 = qdesc[2 * qi + 1];
    const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const CellWindow W = cell_window(F, Q);
    const int lo = W.lo, hi = W.hi;
    int m = 0;
    for (int base = lo; base < hi; base += 64) {
        const int c = base + lane;
        bool ok = c < hi;
        int dist = 0;
        if (ok) ok = pair_pred(F, Q, W, c, check_levels);
        if (ok) {
            dist = hamming256(F.desc[2 * c], F.desc[2 * c + 1], qd0, qd1);
            if (dist > Q.max_dist) ok = false;
            if (F.limit && !(dist < F.limit[c])) ok = false;
        }
        const unsigned long long mask = __ballot(ok);
        if (mask) {
            const int pos = m + __popcll(mask & lt_mask);
            if (ok && pos < kListCap) s_keys[w][pos] = make_key(Q, dist, c);
            m += __popcll(mask);
        }
    }
    if (lane == 0) out_count[qi] = m;
    if (m <= kListCap) {
        // K smallest keys of the list: round k = min over keys greater than the previous minimum (keys are unique)
        uint32_t prev = 0;
        for (int k = 0; k < K; k++) {
            uint32_t cur = 0xFFFFFFFFu;
            for (int i = lane; i < m; i += 64) {
                const uint32_t key = s_keys[w][i];
                if ((k == 0 || key > prev) && key < cur) cur = key;
            }
            cur = wave_min_u32(cur);
            if (lane == 0) out_keys[(size_t)qi * K + k] = cur;
            if (cur == 0xFFFFFFFFu) {
                for (int k2 = k + 1 + lane; k2 < K; k2 += 64) out_keys[(size_t)qi * K + k2] = 0xFFFFFFFFu;
                break;
            }
            prev = cur;
        }
    } else {
        // window larger than the LDS list: rescan the candidates once per output rank (exact, slower)
        uint32_t prev = 0;
        for (int k = 0; k < K; k++) {
            uint32_t cur = 0xFFFFFFFFu;
            for (int base = lo; base < hi; base += 64) {
                const int c = base + lane;
                if (c < hi && pair_pred(F, Q, W, c, check_levels)) {
                    const int dist = hamming256(F.desc[2 * c], F.desc[2 * c + 1], qd0, qd1);
                    if (dist <= Q.max_dist && (!F.limit || dist < F.limit[c])) {
                        const uint32_t key = make_key(Q, dist, c);
                        if ((k == 0 || key > prev) && key < cur) cur = key;
                    }
                }
            }
            cur = wave_min_u32(cur);
            if (lane == 0) out_keys[(size_t)qi * K + k] = cur;
            prev = cur;
            if (cur == 0xFFFFFFFFu) {
                for (int k2 = k + 1 + lane; k2 < K; k2 += 64) out_keys[(size_t)qi * K + k2] = 0xFFFFFFFFu;
                break;
            }
        }
    }
}

// Copy the staged inputs from pinned host memory into HBM with a kernel on the matcher's own queue: a ~100 KB
// SDMA copy costs a cross-engine dependency (~15 us) in front of a 19 us kernel.
__global__ __launch_bounds__(256) void stage_in_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src,
                                                       size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}

void launch_stage_in(void* dst, const void* src_mapped, size_t bytes, hipStream_t s) {
    const size_t n16 = (bytes + 15) / 16;
    if (!n16) return;
    const int blocks = (int)std::min<size_t>((n16 + 255) / 256, 128);
    hipLaunchKernelGGL(stage_in_kernel, dim3(blocks), dim3(256), 0, s, (uint4*)dst, (const uint4*)src_mapped, n16);
}

void launch_topk_window(const MatchFrameDev& F, const void* d_q, bool compact, const uint4* d_qdesc, int nq, int K,
                        uint32_t* d_keys, int32_t* d_count, hipStream_t s) {
    if (nq <= 0) return;
    if (compact)
        hipLaunchKernelGGL(topk_window_kernel<true>, dim3((nq + 3) / 4), dim3(256), 0, s, F, d_q, d_qdesc, nq, K, d_keys,
                           d_count);
    else
        hipLaunchKernelGGL(topk_window_kernel<false>, dim3((nq + 3) / 4), dim3(256), 0, s, F, d_q, d_qdesc, nq, K, d_keys,
                           d_count);
}

// Brute-force best / second-best of each row of A against all rows of B; ties: lowest index in B.
// One wave per query row; each lane strides over B keeping its own (best, second) keys, then a wave merge.
__global__ __launch_bounds__(256) void hamming_top2_kernel(const uint4* __restrict__ A, int na,
                                                            const uint4* __restrict__ B, int nb,
                                                            int32_t* __restrict__ best_idx,
                                                            int32_t* __restrict__ best_dist,
                                                            int32_t* __restrict__ second_dist) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int qi = blockIdx.x * 4 + w;
    if (qi >= na) return;
    const uint4 a0 = A[2 * qi], a1 = A[2 * qi + 1];
    // key = dist << 20 | index (nb < 2^20); the two smallest keys give best + second in reference scan order
    uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    for (int c = lane; c < nb; c += 64) {
        const uint32_t key = ((uint32_t)hamming256(B[2 * c], B[2 * c + 1], a0, a1) << 20) | (uint32_t)c;
        if (key < k1) {
            k2 = k1;
            k1 = key;
        } else if (key < k2) {
            k2 = key;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t o1 = (uint32_t)__shfl_xor((int)k1, off), o2 = (uint32_t)__shfl_xor((int)k2, off);
        // merge two sorted pairs, keep the two smallest
        const uint32_t n1 = min(k1, o1);
        const uint32_t n2 = min(max(k1, o1), min(k2, o2));
        k1 = n1;
        k2 = n2;
    }
    if (lane == 0) {
        best_idx[qi] = k1 == 0xFFFFFFFFu ? -1 : (int32_t)(k1 & 0xFFFFFu);
        best_dist[qi] = k1 == 0xFFFFFFFFu ? 256 : (int32_t)(k1 >> 20);
        second_dist[qi] = k2 == 0xFFFFFFFFu ? 256 : (int32_t)(k2 >> 20);
    }
}

// ---------------- MapPoint::ComputeDistinctiveDescriptors (code/src/MapPoint.cc:323-392), batched ----------------
// One wave per map point.  The point's N observed descriptors sit in LDS; lane i owns row i of the N x N Hamming
// matrix and never materialises it: it histograms its N distances (self-distance 0 included, as the reference's
// vDists does) into a private 257-bin LDS histogram and walks the bins to the element of rank int(0.5 (N-1)) — the
// median the reference reads out of its sorted row.  The winner is the smallest (median, row) pair, i.e. the
// first row with the least median, found by a wave-min over median << 16 | row.
constexpr int kDdMaxObs = 512;  // observations per map point handled on the device

__global__ __launch_bounds__(64) void distinctive_desc_kernel(const uint4* __restrict__ desc, const int32_t* __restrict__ off,
                                                              int n_points, int32_t* __restrict__ best_idx,
                                                              int32_t* __restrict__ best_median) {
    __shared__ uint4 s_desc[2 * kDdMaxObs];
    __shared__ uint16_t s_hist[64][258];
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n_points) return;
    const int base = off[p], N = off[p + 1] - base;
    if (N <= 0) {
        if (lane == 0) { best_idx[p] = -1; best_median[p] = 0x7fffffff; }
        return;
    }
    for (int i = lane; i < 2 * N; i += 64) s_desc[i] = desc[2 * (size_t)base + i];
    __syncthreads();
    const int k = (int)(0.5 * (double)(N - 1));  // vDists[0.5*(N-1)]
    uint32_t best = 0xFFFFFFFFu;
    for (int row0 = 0; row0 < N; row0 += 64) {
        const int i = row0 + lane;
        for (int b = 0; b < 258; b++) s_hist[lane][b] = 0;
        if (i < N) {
            const uint4 a0 = s_desc[2 * i], a1 = s_desc[2 * i + 1];
            for (int j = 0; j < N; j++) s_hist[lane][hamming256(s_desc[2 * j], s_desc[2 * j + 1], a0, a1)]++;
            int cum = 0, med = 0;
            for (int b = 0; b <= 256; b++) {
                cum += s_hist[lane][b];
                if (cum > k) { med = b; break; }
            }
            const uint32_t key = ((uint32_t)med << 16) | (uint32_t)i;
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)best, o);
        best = other < best ? other : best;
    }
    if (lane == 0) {
        best_idx[p] = (int32_t)(best & 0xFFFFu);
        best_median[p] = (int32_t)(best >> 16);
    }
}

void launch_distinctive_desc(const uint4* d_desc, const int32_t* d_off, int n_points, int32_t* d_best_idx,
                             int32_t* d_best_median, hipStream_t s) {
    if (n_points <= 0) return;
    hipLaunchKernelGGL(distinctive_desc_kernel, dim3(n_points), dim3(64), 0, s, d_desc, d_off, n_points, d_best_idx,
                       d_best_median);
}

void launch_hamming_top2(const uint4* d_A, int na, const uint4* d_B, int nb, int32_t* d_best_idx,
                         int32_t* d_best_dist, int32_t* d_second_dist, hipStream_t s) {
    if (na <= 0) return;
    hipLaunchKernelGGL(hamming_top2_kernel, dim3((na + 3) / 4), dim3(256), 0, s, d_A, na, d_B, nb, d_best_idx,
                       d_best_dist, d_second_dist);
}

}  // namespace so
