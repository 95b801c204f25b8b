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
#include "pose_convert.h"
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
            const int py = (int)roundf((xy.y - F.grid_min_y) * F.grid_inv_h);
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

// Pc = mRcw * P + mtcw the way OpenCV's GEMM evaluates it for CV_32F: double accumulation, one rounding
__device__ __forceinline__ void track_camera_point(const float* T, const float* P, float* Pc) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double s = (double)T[4 * r] * (double)P[0] + (double)T[4 * r + 1] * (double)P[1] + (double)T[4 * r + 2] * (double)P[2];
        Pc[r] = (float)(s + (double)T[4 * r + 3]);
    }
}

__device__ __forceinline__ double track_log(double x) {  // == frame_log (frame_kernels.hip) == orc_log
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)u);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

// SearchByProjection(cur, last): the projection of one last-frame map point, code/src/ORBmatcher.cc:1251-1276
__device__ __forceinline__ MatchQuery track_query_last(const TrackQuerySrc& T, int qi, int& slot) {
    MatchQuery Q = MatchQuery{};
    Q.max_dist = 256;
    slot = T.slot ? T.slot[qi] : T.slot_base + qi;
    if (slot < 0 || slot >= T.n_slots) return Q;
    const int oct = T.last_octave[qi];
    if (oct < 0 || oct >= T.nlevels) return Q;
    const float P[3] = {T.Xw[3 * (size_t)slot], T.Xw[3 * (size_t)slot + 1], T.Xw[3 * (size_t)slot + 2]};
    float Pc[3];
    track_camera_point(T.Tcw, P, Pc);
    const float invzc = 1.0f / Pc[2];
    if (!(invzc >= 0.0f)) return Q;  // "if (invzc < 0) continue" (NaN never passes the bounds tests below either)
    const float u = T.fx * Pc[0] * invzc + T.cx;
    const float v = T.fy * Pc[1] * invzc + T.cy;
    if (u < T.bounds[0] || u > T.bounds[1]) return Q;
    if (v < T.bounds[2] || v > T.bounds[3]) return Q;
    if (!(u >= T.bounds[0] && v >= T.bounds[2])) return Q;
    Q.u = u;
    Q.v = v;
    Q.r = T.th * T.scale[oct];
    Q.min_level = oct - 1;
    Q.max_level = oct + 1;
    Q.active = 1;
    // a candidate farther than TH_HIGH can never be bound (ORBmatcher.cc:1311) and, there being no ratio test in this
    // search, never influences which one is: it need not travel to the host
    Q.max_dist = 100;
    return Q;
}

// SearchLocalPoints: Frame::isInFrustum + PredictScale for one local map point (code/src/Frame.cc:316-375,
// code/src/MapPoint.cc:466-485), then the window of SearchByProjection(F, vpMapPoints, th) (ORBmatcher.cc:52-70)
__device__ __forceinline__ MatchQuery track_query_local(const TrackQuerySrc& T, int qi, int& slot) {
    MatchQuery Q = MatchQuery{};
    Q.max_dist = 256;
    slot = T.slot ? T.slot[qi] : T.slot_base + qi;
    if (slot < 0 || slot >= T.n_slots) return Q;
    if (T.use_bits) {
        if ((T.skip_bits[qi >> 5] >> (qi & 31)) & 1u) return Q;
    } else if (T.skip && T.skip[qi]) {
        return Q;
    }
    const float* Tc = T.Tcw;
    float Ow[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const double s = (double)Tc[0 + j] * (double)Tc[3] + (double)Tc[4 + j] * (double)Tc[7] + (double)Tc[8 + j] * (double)Tc[11];
        Ow[j] = (float)(-s);
    }
    const float P[3] = {T.Xw[3 * (size_t)slot], T.Xw[3 * (size_t)slot + 1], T.Xw[3 * (size_t)slot + 2]};
    float Pc[3];
    track_camera_point(Tc, P, Pc);
    if (Pc[2] < 0.0f) return Q;
    const float invz = 1.0f / Pc[2];
    const float u = T.fx * Pc[0] * invz + T.cx;
    const float v = T.fy * Pc[1] * invz + T.cy;
    if (u < T.bounds[0] || u > T.bounds[1]) return Q;
    if (v < T.bounds[2] || v > T.bounds[3]) return Q;
    if (!(u >= T.bounds[0] && v >= T.bounds[2])) return Q;  // NaN
    const float max_d = T.max_dist[slot], min_d = T.min_dist[slot];
    const float maxD = 1.2f * max_d, minD = 0.8f * min_d;
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const double n2 = (double)PO[0] * (double)PO[0] + (double)PO[1] * (double)PO[1] + (double)PO[2] * (double)PO[2];
    const float dist = (float)sqrt(n2);
    if (dist < minD || dist > maxD) return Q;
    const double dot = (double)PO[0] * (double)T.normal[3 * (size_t)slot] + (double)PO[1] * (double)T.normal[3 * (size_t)slot + 1] +
                       (double)PO[2] * (double)T.normal[3 * (size_t)slot + 2];
    const float vc = (float)(dot / (double)dist);
    if (vc < T.cos_limit) return Q;
    const float ratio = max_d / dist;
    const float lr = (float)track_log((double)ratio);
    int nScale = (int)ceilf(lr / T.log_scale_factor);
    if (nScale > T.nlevels - 1) nScale = T.nlevels - 1;
    if (nScale < 0) nScale = 0;
    float r = vc > 0.998f ? 2.5f : 4.0f;  // RadiusByViewingCos, ORBmatcher.cc:123-128
    if (T.th != 1.0f) r *= T.th;           // bFactor
    Q.u = u;
    Q.v = v;
    Q.r = r * T.scale[nScale];
    Q.min_level = nScale - 1;
    Q.max_level = nScale;
    Q.active = 1;
    // the best candidate must be within TH_HIGH = 100 (ORBmatcher.cc:107); a second best beyond 100 / ratio can never
    // make the ratio test fail (bestDist <= 100 < ratio * second), so candidates beyond that bound are dropped here
    Q.max_dist = T.second_best_bound;
    return Q;
}

// MODE 0: MatchQuery records, 1: MatchQueryW records, 2: last-frame tracking search, 3: local-map tracking search,
// 4: batched jobs - q points to a BatchJobDev table, blockIdx.y selects the job (frame, queries and outputs from there)
// 5 / 6: modes 2 / 3 for a GROUP of agents (so_track_group): q points to a TrackGroupJob table in HBM, blockIdx.y selects the
// agent - its frame, its query source with the gates, its outputs; one launch searches for all of them
template <int MODE>
__global__ __launch_bounds__(256) void topk_window_kernel(MatchFrameDev F_arg, const void* __restrict__ q,
                                                           const uint4* __restrict__ qdesc, int nq, int K,
                                                           uint32_t* __restrict__ out_keys,
                                                           int32_t* __restrict__ out_count, TrackQuerySrc T_arg,
                                                           int q_first) {
    __shared__ uint32_t s_keys[4][kListCap];
    constexpr bool kTrack = MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6;
    constexpr bool kTrackLast = MODE == 2 || MODE == 5;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int blk = blockIdx.x;
    const TrackGroupJob* gj = MODE >= 5 ? static_cast<const TrackGroupJob*>(q) + blockIdx.y : nullptr;
    const TrackQuerySrc& T = MODE >= 5 ? gj->T : T_arg;
    const BatchJobDev* job = nullptr;
    if (MODE == 4) {  // qdesc carries the batch's grid table here: which job does this workgroup belong to?
        const BatchGridDev* G = reinterpret_cast<const BatchGridDev*>(qdesc);
        // Workgroups are dealt to the eight XCDs round-robin and every XCD has its own L2: with blk = blockIdx.x the
        // workgroups of one job (one resident keyframe against one query list) sat on all eight, and each XCD pulled every
        // keyframe's descriptors and cell tables across the fabric (FETCH_SIZE 4.5x the batch's inputs, round 4).  XCD x
        // takes the x-th eighth of the batch's workgroups instead - two or three whole jobs - so a keyframe is read by one
        // XCD, two at a seam.  Results do not depend on which workgroup computes a query.
        const int total = G->first_topk[G->n_jobs], chunk = (total + 7) >> 3;
        blk = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= chunk || blk >= total) return;
        int lo = 0, hi = G->n_jobs - 1;
        while (lo < hi) {  // largest j with first_topk[j] <= blk (uniform: scalar loads from one cached block)
            const int mid = (lo + hi + 1) >> 1;
            if (G->first_topk[mid] <= blk) lo = mid; else hi = mid - 1;
        }
        job = static_cast<const BatchJobDev*>(q) + lo;
        blk -= G->first_topk[lo];
    }
    const int qi = __builtin_amdgcn_readfirstlane(blk * 4 + w);  // wave-uniform: per-query data through scalar loads
    const MatchFrameDev& F = MODE == 4 ? job->F : (MODE >= 5 ? gj->F : F_arg);
    if (MODE == 4) {
        nq = job->nq; K = job->K; qdesc = job->qdesc; out_keys = job->keys; out_count = job->count;
    }
    if (MODE >= 5) {
        nq = gj->nq; K = gj->K; out_keys = gj->keys; out_count = gj->count; q_first = 0;
    }
    if (qi >= nq) return;
    const bool soa = kTrack && T.keys_soa;
    const size_t kq = soa ? (size_t)nq : 1, kk = soa ? 1 : (size_t)K;  // key (qi, k) lives at qi * kk + k * kq
    const bool bits = kTrack && T.use_bits;
    MatchQuery Q;
    int slot = -1;
    if (MODE == 1) {
        const MatchQueryW c = static_cast<const MatchQueryW*>(q)[qi];
        Q = MatchQuery{};
        Q.u = c.u; Q.v = c.v; Q.r = c.r;
        Q.min_level = c.min_level; Q.max_level = c.max_level;
        Q.active = c.active;
        Q.max_dist = 256;
    } else if (MODE == 0) {
        Q = static_cast<const MatchQuery*>(q)[qi];
    } else if (MODE == 4) {
        Q = job->q[qi];
    } else if (kTrackLast) {
        Q = track_query_last(T, q_first + qi, slot);
    } else {
        Q = track_query_local(T, q_first + qi, slot);
        if (T.in_view_out && lane == 0) T.in_view_out[q_first + qi] = (uint8_t)Q.active;
    }
    if (kTrack && T.slot_out && lane == 0) T.slot_out[q_first + qi] = slot;
    if (!Q.active) {
        if (lane == 0) out_count[qi] = 0;
        if (kTrack && T.count8_out && lane == 0) T.count8_out[q_first + qi] = 0;
        for (int k = lane; k < K; k += 64) out_keys[(size_t)qi * kk + k * kq] = 0xFFFFFFFFu;
        return;
    }
    // (a batched job whose map points live in a so_map: the descriptor is row qslot[qi] of the map's table; the query
    //  is only active when that slot is a row of the table)
    const size_t qrow = (MODE == 4 && job->qslot) ? (size_t)job->qslot[qi] : (size_t)qi;
    const uint4* qsrc = kTrack ? T.desc + 2 * (size_t)slot : qdesc + 2 * qrow;
    const uint4 qd0 = qsrc[0], qd1 = qsrc[1];
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
            if (bits) {
                if ((T.excl_bits[c >> 5] >> (c & 31)) & 1u) ok = false;
            } else if (F.limit && !(dist < F.limit[c])) {
                ok = false;
            }
        }
        const unsigned long long mask = __ballot(ok);
        if (mask) {
            const int pos = m + __popcll(mask & lt_mask);
            if (ok && pos < kListCap) s_keys[w][pos] = make_key(Q, dist, c);
            m += __popcll(mask);
        }
    }
    if (lane == 0) out_count[qi] = m;
    if (kTrack && T.count8_out && lane == 0) T.count8_out[q_first + qi] = (uint8_t)min(m, 255);
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
            if (lane == 0) out_keys[(size_t)qi * kk + k * kq] = cur;
            if (cur == 0xFFFFFFFFu) {
                for (int k2 = k + 1 + lane; k2 < K; k2 += 64) out_keys[(size_t)qi * kk + k2 * kq] = 0xFFFFFFFFu;
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
                    const bool gate_ok = bits ? !((T.excl_bits[c >> 5] >> (c & 31)) & 1u) : (!F.limit || dist < F.limit[c]);
                    if (dist <= Q.max_dist && gate_ok) {
                        const uint32_t key = make_key(Q, dist, c);
                        if ((k == 0 || key > prev) && key < cur) cur = key;
                    }
                }
            }
            cur = wave_min_u32(cur);
            if (lane == 0) out_keys[(size_t)qi * kk + k * kq] = cur;
            prev = cur;
            if (cur == 0xFFFFFFFFu) {
                for (int k2 = k + 1 + lane; k2 < K; k2 += 64) out_keys[(size_t)qi * kk + k2 * kq] = 0xFFFFFFFFu;
                break;
            }
        }
    }
}

// ---------------- projection + gating half of Fuse / SearchBySim3 / the keyframe-side SearchByProjection ----------------
// Thread per map point.  Every float / double operation is written in the order the reference's statements evaluate
// it (code/src/ORBmatcher.cc:776-815, :923-964, :293-333, :1063-1094, :1380-1410) with the cv::Mat conventions of
// oracle/project_oracle.h: one GEMM = double accumulation and one rounding, norm / dot in double.  A rejected point
// leaves an inactive query behind (topk_window_kernel writes "no candidate" for it).
__device__ __forceinline__ void project_one(const ProjectSrc& S, int i, MatchQuery* __restrict__ q_out,
                                            MatchQueryW* __restrict__ qw_out) {
    MatchQuery Q = MatchQuery{};
    Q.max_dist = S.q_max_dist;
    Q.flags = S.qflags;
    bool ok = S.valid[i] != 0;
    size_t j = (size_t)i;  // row of the point's fields
    if (S.slot) {
        const int sl = S.slot[i];
        if (sl < 0 || sl >= S.n_rows) ok = false;
        j = ok ? (size_t)sl : 0;
    }
    const float P[3] = {S.Xw[3 * j], S.Xw[3 * j + 1], S.Xw[3 * j + 2]};
    float Pc[3];
    track_camera_point(S.A, P, Pc);  // Rcw * p3Dw + tcw
    if (S.flags & kPChain) {         // p3Dc2 = sR21 * p3Dc1 + t21
        float P2[3];
        track_camera_point(S.B, Pc, P2);
        Pc[0] = P2[0]; Pc[1] = P2[1]; Pc[2] = P2[2];
    }
    float u, v;
    if (S.flags & kPFrameForm) {  // :1383-1393
        const float invzc = 1.0f / Pc[2];
        u = S.fx * Pc[0] * invzc + S.cx;
        v = S.fy * Pc[1] * invzc + S.cy;
        if (u < S.bounds[0] || u > S.bounds[1]) ok = false;
        if (v < S.bounds[2] || v > S.bounds[3]) ok = false;
        if (!(u >= S.bounds[0] && v >= S.bounds[2])) ok = false;  // NaN
    } else {
        if (Pc[2] < 0.0f) ok = false;  // Depth must be positive
        const float invz = 1.0f / Pc[2];
        const float x = Pc[0] * invz;
        const float y = Pc[1] * invz;
        u = S.fx * x + S.cx;
        v = S.fy * y + S.cy;
        if (!(u >= S.bounds[0] && u < S.bounds[1] && v >= S.bounds[2] && v < S.bounds[3])) ok = false;  // IsInImage
    }
    const float max_d = S.max_dist[j], min_d = S.min_dist[j];
    const float maxD = 1.2f * max_d, minD = 0.8f * min_d;
    float PO[3] = {Pc[0], Pc[1], Pc[2]};  // SearchBySim3 measures the camera-frame point
    if (!(S.flags & kPChain)) {
        PO[0] = P[0] - S.Ow[0]; PO[1] = P[1] - S.Ow[1]; PO[2] = P[2] - S.Ow[2];
    }
    const double n2 = (double)PO[0] * (double)PO[0] + (double)PO[1] * (double)PO[1] + (double)PO[2] * (double)PO[2];
    const float dist = (float)sqrt(n2);
    if (dist < minD || dist > maxD) ok = false;
    if (S.flags & kPAngleGate) {
        const double dot = (double)PO[0] * (double)S.normal[3 * j] + (double)PO[1] * (double)S.normal[3 * j + 1] +
                           (double)PO[2] * (double)S.normal[3 * j + 2];
        if (dot < 0.5 * (double)dist) ok = false;
    }
    int nScale = 0;
    if (ok) {  // MapPoint::PredictScale
        const float ratio = max_d / dist;
        const float lr = (float)track_log((double)ratio);
        nScale = (int)ceilf(lr / S.log_scale_factor);
        if (nScale > S.nlevels - 1) nScale = S.nlevels - 1;
        if (nScale < 0) nScale = 0;
        Q.u = u;
        Q.v = v;
        Q.r = S.th * S.scale[nScale];
        Q.min_level = nScale - 1;
        Q.max_level = nScale + S.level_above;
        Q.active = 1;
    }
    q_out[i] = Q;
    MatchQueryW W;
    W.u = Q.u; W.v = Q.v; W.r = Q.r;
    W.min_level = (int8_t)Q.min_level; W.max_level = (int8_t)Q.max_level;
    W.active = (uint8_t)Q.active; W.pad = 0;
    qw_out[i] = W;
}

__global__ __launch_bounds__(256) void project_queries_kernel(ProjectSrc S, MatchQuery* __restrict__ q_out,
                                                              MatchQueryW* __restrict__ qw_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S.n) return;
    project_one(S, i, q_out, qw_out);
}

__global__ __launch_bounds__(256) void project_queries_batch_kernel(const BatchGridDev* __restrict__ G,
                                                                    const BatchJobDev* __restrict__ jobs) {
    int blk = blockIdx.x, lo = 0, hi = G->n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (G->first_proj[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    const BatchJobDev& J = jobs[lo];
    const int i = (blk - G->first_proj[lo]) * 256 + threadIdx.x;
    if (J.project == 2) {
        // SearchForTriangulation's query of keyframe 1's feature at position i (code/src/ORBmatcher.cc:636-660): only features
        // without a map point, only against keyframe 2's features of the same vocabulary node; the epipolar line of the
        // keypoint in image 2 with the float expressions of CheckDistEpipolarLine (:131-137)
        if (i >= J.nq) return;
        MatchQuery q;
        q.u = q.v = q.r = 0.f;
        q.min_level = q.max_level = 0;
        q.active = 0;
        q.c_begin = q.c_end = 0;
        q.flags = 0;
        q.max_dist = 256;
        q.la = q.lb = q.lc = 0.f;
        q.pad = 0;
        const int2 cr = J.t_range[J.t_node[i]];
        if (J.t_free1[i] && cr.y > cr.x) {
            const float2 p = J.t_xy1[i];
            q.active = 1;
            q.flags = kQRange | kQEpipolar | kQPreferLast;
            q.max_dist = 50;  // TH_LOW
            q.c_begin = cr.x;
            q.c_end = cr.y;
            q.la = p.x * J.t_F[0] + p.y * J.t_F[3] + J.t_F[6];
            q.lb = p.x * J.t_F[1] + p.y * J.t_F[4] + J.t_F[7];
            q.lc = p.x * J.t_F[2] + p.y * J.t_F[5] + J.t_F[8];
        }
        J.q[i] = q;
        return;
    }
    if (!J.project || i >= J.S.n) return;
    project_one(J.S, i, J.q, J.qw);
}

void launch_project_queries(const ProjectSrc& S, MatchQuery* d_q, MatchQueryW* d_qw_mapped, hipStream_t s) {
    if (S.n <= 0) return;
    hipLaunchKernelGGL(project_queries_kernel, dim3((S.n + 255) / 256), dim3(256), 0, s, S, d_q, d_qw_mapped);
}

// ---------------- LocalMapping::CreateNewMapPoints, per match (code/src/LocalMapping.cc:263-420, monocular) ----------------
// Thread per match: parallax of the two rays, the 4 x 4 linear triangulation solved with OpenCV's one-sided Jacobi SVD
// (cv::SVD::compute, restated in oracle/mapping_oracle.h: the null vector is the row of V^T that ends up last), positive
// depth in both cameras, reprojection error against 5.991 sigma^2 in both, scale consistency.  Everything lives in
// registers: the pair loops are unrolled so that no array is indexed at run time; the sweeps diverge per thread.
__device__ __forceinline__ void tri_swap4(float* a, float* b) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float t = a[k];
        a[k] = b[k];
        b[k] = t;
    }
}

__device__ __forceinline__ void svd4_last_row(const float* A, float* v4) {
    float At[4][4], Vt[4][4];
    double W[4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) At[c][r] = A[r * 4 + c];
    const float eps = 1.1920928955078125e-07f * 2;  // FLT_EPSILON * 2
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += (double)At[i][k] * At[i][k];
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < 4; k++) Vt[i][k] = (i == k) ? 1.f : 0.f;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i + 1; j < 4; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < 4; k++) p += (double)At[i][k] * At[j][k];
                if (!(fabs(p) <= eps * sqrt(a * b))) {
                    p *= 2;
                    const double beta = a - b, gamma = sqrt(p * p + beta * beta);
                    float c, s;
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = (float)sqrt(delta / gamma);
                        c = (float)(p / (gamma * s * 2));
                    } else {
                        c = (float)sqrt((gamma + beta) / (gamma * 2));
                        s = (float)(p / (gamma * c * 2));
                    }
                    a = b = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float t0 = c * At[i][k] + s * At[j][k];
                        const float t1 = -s * At[i][k] + c * At[j][k];
                        At[i][k] = t0;
                        At[j][k] = t1;
                        a += (double)t0 * t0;
                        b += (double)t1 * t1;
                    }
                    W[i] = a;
                    W[j] = b;
                    changed = true;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float t0 = c * Vt[i][k] + s * Vt[j][k];
                        const float t1 = -s * Vt[i][k] + c * Vt[j][k];
                        Vt[i][k] = t0;
                        Vt[j][k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += (double)At[i][k] * At[i][k];
        W[i] = sqrt(sd);
    }
    // the selection sort (descending, first maximum on ties) with the rows of V^T swapped along
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int j = i;
#pragma unroll
        for (int k = i + 1; k < 4; k++) {
            const double wj = (j == 0) ? W[0] : (j == 1) ? W[1] : (j == 2) ? W[2] : W[3];
            if (wj < W[k]) j = k;
        }
#pragma unroll
        for (int k = i + 1; k < 4; k++)
            if (j == k) {
                const double tw = W[i];
                W[i] = W[k];
                W[k] = tw;
                tri_swap4(Vt[i], Vt[k]);
            }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) v4[k] = Vt[3][k];
}

__global__ __launch_bounds__(128) void triangulate_kernel(TriArgs A) {
    const int m = blockIdx.x * 128 + threadIdx.x;
    if (m >= A.n) return;
    const TriKeyframeDev& k1 = A.kf1;
    const TriKeyframeDev& k2 = A.kf2[A.kf2_of[m]];
    const float* T1 = k1.Tcw;
    const float* T2 = k2.Tcw;
    const float2 p1 = A.xy1[m], p2 = A.xy2[m];
    const int o1 = A.oct1[m], o2 = A.oct2[m];
    bool ok = true;
    const float xn1[3] = {(p1.x - k1.cx) * k1.invfx, (p1.y - k1.cy) * k1.invfy, 1.0f};
    const float xn2[3] = {(p2.x - k2.cx) * k2.invfx, (p2.y - k2.cy) * k2.invfy, 1.0f};
    float ray1[3], ray2[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        ray1[j] = (float)((double)T1[0 + j] * xn1[0] + (double)T1[4 + j] * xn1[1] + (double)T1[8 + j] * xn1[2]);
        ray2[j] = (float)((double)T2[0 + j] * xn2[0] + (double)T2[4 + j] * xn2[1] + (double)T2[8 + j] * xn2[2]);
    }
    const double dot = (double)ray1[0] * ray2[0] + (double)ray1[1] * ray2[1] + (double)ray1[2] * ray2[2];
    const double n1 = sqrt((double)ray1[0] * ray1[0] + (double)ray1[1] * ray1[1] + (double)ray1[2] * ray1[2]);
    const double n2 = sqrt((double)ray2[0] * ray2[0] + (double)ray2[1] * ray2[1] + (double)ray2[2] * ray2[2]);
    const float cosParallaxRays = (float)(dot / (n1 * n2));
    const float cosParallaxStereo = cosParallaxRays + 1;
    if (!(cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && (double)cosParallaxRays < 0.9998)) ok = false;
    float X[3] = {0.f, 0.f, 0.f};
    if (ok) {
        float M[16];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            M[0 + c] = xn1[0] * T1[8 + c] - T1[0 + c];
            M[4 + c] = xn1[1] * T1[8 + c] - T1[4 + c];
            M[8 + c] = xn2[0] * T2[8 + c] - T2[0 + c];
            M[12 + c] = xn2[1] * T2[8 + c] - T2[4 + c];
        }
        float v[4];
        svd4_last_row(M, v);
        if (v[3] == 0) ok = false;
        const float iw = (float)(1.0 / (double)v[3]);
        X[0] = v[0] * iw; X[1] = v[1] * iw; X[2] = v[2] * iw;
    }
    if (ok) {
        const float z1 = (float)(((double)T1[8] * X[0] + (double)T1[9] * X[1] + (double)T1[10] * X[2]) + (double)T1[11]);
        const float z2 = (float)(((double)T2[8] * X[0] + (double)T2[9] * X[1] + (double)T2[10] * X[2]) + (double)T2[11]);
        if (z1 <= 0 || z2 <= 0) ok = false;
        const float x1 = (float)(((double)T1[0] * X[0] + (double)T1[1] * X[1] + (double)T1[2] * X[2]) + (double)T1[3]);
        const float y1 = (float)(((double)T1[4] * X[0] + (double)T1[5] * X[1] + (double)T1[6] * X[2]) + (double)T1[7]);
        const float invz1 = (float)(1.0 / (double)z1);
        const float u1 = k1.fx * x1 * invz1 + k1.cx;
        const float v1 = k1.fy * y1 * invz1 + k1.cy;
        const float ex1 = u1 - p1.x, ey1 = v1 - p1.y;
        if ((double)(ex1 * ex1 + ey1 * ey1) > 5.991 * (double)k1.sigma2[o1]) ok = false;
        const float x2 = (float)(((double)T2[0] * X[0] + (double)T2[1] * X[1] + (double)T2[2] * X[2]) + (double)T2[3]);
        const float y2 = (float)(((double)T2[4] * X[0] + (double)T2[5] * X[1] + (double)T2[6] * X[2]) + (double)T2[7]);
        const float invz2 = (float)(1.0 / (double)z2);
        const float u2 = k2.fx * x2 * invz2 + k2.cx;
        const float v2 = k2.fy * y2 * invz2 + k2.cy;
        const float ex2 = u2 - p2.x, ey2 = v2 - p2.y;
        if ((double)(ex2 * ex2 + ey2 * ey2) > 5.991 * (double)k2.sigma2[o2]) ok = false;
        const float a1[3] = {X[0] - k1.Ow[0], X[1] - k1.Ow[1], X[2] - k1.Ow[2]};
        const float a2[3] = {X[0] - k2.Ow[0], X[1] - k2.Ow[1], X[2] - k2.Ow[2]};
        const float dist1 = (float)sqrt((double)a1[0] * a1[0] + (double)a1[1] * a1[1] + (double)a1[2] * a1[2]);
        const float dist2 = (float)sqrt((double)a2[0] * a2[0] + (double)a2[1] * a2[1] + (double)a2[2] * a2[2]);
        if (dist1 == 0 || dist2 == 0) ok = false;
        const float ratioDist = dist2 / dist1;
        const float ratioOctave = k1.scale[o1] / k2.scale[o2];
        if (ratioDist * A.ratio_factor < ratioOctave || ratioDist > ratioOctave * A.ratio_factor) ok = false;
    }
    A.ok[m] = ok ? 1 : 0;
    if (ok) {
        A.x3D[3 * (size_t)m] = X[0];
        A.x3D[3 * (size_t)m + 1] = X[1];
        A.x3D[3 * (size_t)m + 2] = X[2];
        if (A.normal) {  // the statements of normal_depth_kernel for observations (keyframe, neighbour), reference = keyframe
            float nsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < 2; o++) {
                const float* Ow = o == 0 ? k1.Ow : k2.Ow;
                const float d[3] = {X[0] - Ow[0], X[1] - Ow[1], X[2] - Ow[2]};
                const double nr = sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);
                const float inv = (float)(1.0 / nr);
#pragma unroll
                for (int j = 0; j < 3; j++) nsum[j] = nsum[j] + d[j] * inv;
            }
            const float PC[3] = {X[0] - k1.Ow[0], X[1] - k1.Ow[1], X[2] - k1.Ow[2]};
            const float dist = (float)sqrt((double)PC[0] * PC[0] + (double)PC[1] * PC[1] + (double)PC[2] * PC[2]);
            const float mx = dist * k1.scale[o1];
            A.max_dist[m] = mx;
            A.min_dist[m] = mx / A.last_scale;
            const float invn = (float)(1.0 / (double)2);
#pragma unroll
            for (int j = 0; j < 3; j++) A.normal[3 * (size_t)m + j] = nsum[j] * invn;
        }
    }
}

void launch_triangulate(const TriArgs& A, hipStream_t s) {
    if (A.n <= 0) return;
    hipLaunchKernelGGL(triangulate_kernel, dim3((A.n + 127) / 128), dim3(128), 0, s, A);
}

// ---------------- MapPoint::UpdateNormalAndDepth (code/src/MapPoint.cc:413-465), thread per map point ----------------
__global__ __launch_bounds__(256) void normal_depth_kernel(NormalDepthArgs A) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= A.n) return;
    const int a = A.off[p], b = A.off[p + 1];
    if (b <= a) return;
    const float Pos[3] = {A.Xw[3 * (size_t)p], A.Xw[3 * (size_t)p + 1], A.Xw[3 * (size_t)p + 2]};
    float nsum[3] = {0.f, 0.f, 0.f};
    int n = 0;
    for (int k = a; k < b; k++) {
        const float* O = A.kf_Ow ? A.kf_Ow + 3 * (size_t)A.obs_kf[k] : A.obs_Ow + 3 * (size_t)k;
        const float d[3] = {Pos[0] - O[0], Pos[1] - O[1], Pos[2] - O[2]};
        const double nr = sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);
        const float inv = (float)(1.0 / nr);
#pragma unroll
        for (int j = 0; j < 3; j++) nsum[j] = nsum[j] + d[j] * inv;
        n++;
    }
    const float* Or = A.kf_Ow ? A.kf_Ow + 3 * (size_t)A.ref_kf[p] : A.ref_Ow + 3 * (size_t)p;
    const float PC[3] = {Pos[0] - Or[0], Pos[1] - Or[1], Pos[2] - Or[2]};
    const float dist = (float)sqrt((double)PC[0] * PC[0] + (double)PC[1] * PC[1] + (double)PC[2] * PC[2]);
    const float mx = dist * A.ref_level_scale[p];
    A.max_dist[p] = mx;
    A.min_dist[p] = mx / A.ref_last_scale[p];
    const float invn = (float)(1.0 / (double)n);
#pragma unroll
    for (int j = 0; j < 3; j++) A.normal[3 * (size_t)p + j] = nsum[j] * invn;
}

void launch_normal_depth(const NormalDepthArgs& A, hipStream_t s) {
    if (A.n <= 0) return;
    hipLaunchKernelGGL(normal_depth_kernel, dim3((A.n + 255) / 256), dim3(256), 0, s, A);
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
    const TrackQuerySrc none{};
    if (compact)
        hipLaunchKernelGGL(topk_window_kernel<1>, dim3((nq + 3) / 4), dim3(256), 0, s, F, d_q, d_qdesc, nq, K, d_keys,
                           d_count, none, 0);
    else
        hipLaunchKernelGGL(topk_window_kernel<0>, dim3((nq + 3) / 4), dim3(256), 0, s, F, d_q, d_qdesc, nq, K, d_keys,
                           d_count, none, 0);
}

void launch_batch(const BatchGridDev* d_grid, const BatchJobDev* d_jobs, int n_jobs, int proj_blocks, int topk_blocks, hipStream_t s) {
    if (n_jobs <= 0) return;
    if (proj_blocks > 0) hipLaunchKernelGGL(project_queries_batch_kernel, dim3(proj_blocks), dim3(256), 0, s, d_grid, d_jobs);
    if (topk_blocks > 0) {
        const TrackQuerySrc none{};
        const MatchFrameDev unused{};
        hipLaunchKernelGGL(topk_window_kernel<4>, dim3(8 * ((topk_blocks + 7) / 8)), dim3(256), 0, s, unused, (const void*)d_jobs,
                           reinterpret_cast<const uint4*>(d_grid), 0, 0, (uint32_t*)nullptr, (int32_t*)nullptr, none, 0);
    }
}

void launch_topk_track(const MatchFrameDev& F, const TrackQuerySrc& T, int mode, int q_first, int nq, int K,
                       uint32_t* d_keys, int32_t* d_count, hipStream_t s) {
    if (nq <= 0) return;
    if (mode == 2)
        hipLaunchKernelGGL(topk_window_kernel<2>, dim3((nq + 3) / 4), dim3(256), 0, s, F, nullptr, nullptr, nq, K, d_keys,
                           d_count, T, q_first);
    else
        hipLaunchKernelGGL(topk_window_kernel<3>, dim3((nq + 3) / 4), dim3(256), 0, s, F, nullptr, nullptr, nq, K, d_keys,
                           d_count, T, q_first);
}

// the tracking search of a GROUP of agents: grid (workgroups of the largest search, agents); a member with fewer queries
// leaves its surplus workgroups at once
void launch_topk_track_group(const TrackGroupJob* d_jobs, int n_jobs, int mode, int max_nq, hipStream_t s) {
    if (n_jobs <= 0 || max_nq <= 0) return;
    const MatchFrameDev none{};
    static const TrackQuerySrc zero{};  // (the by-value argument of the single-agent modes: not read by the group modes)
    if (mode == 2)
        hipLaunchKernelGGL(topk_window_kernel<5>, dim3((max_nq + 3) / 4, n_jobs), dim3(256), 0, s, none, (const void*)d_jobs, nullptr, 0, 0,
                           nullptr, nullptr, zero, 0);
    else
        hipLaunchKernelGGL(topk_window_kernel<6>, dim3((max_nq + 3) / 4, n_jobs), dim3(256), 0, s, none, (const void*)d_jobs, nullptr, 0, 0,
                           nullptr, nullptr, zero, 0);
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

// One workgroup: copy the keyframe's descriptors into the exchange slot (zero padding behind them), checksum them,
// write the header row.  64 KB at most: a single CU streams it in a few microseconds and no second launch is needed.
__global__ __launch_bounds__(1024) void exchange_fill_slot_kernel(const uint4* __restrict__ desc, int n, int slot_keypoints,
                                                                  int rank, uint4* __restrict__ slot) {
    __shared__ unsigned long long s_part[16];
    const int tid = threadIdx.x;
    unsigned long long acc = 0;
    for (int j = tid; j < 2 * slot_keypoints; j += 1024) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (j < 2 * n) {
            v = desc[j];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const unsigned long long i = (unsigned long long)j * 16 + q * 4 + b;  // byte index
                    acc += (unsigned long long)((w[q] >> (8 * b)) & 0xFFu) * (i % 65521ull + 1ull);
                }
        }
        slot[2 + j] = v;  // descriptors start at row 1 (32 B = 2 uint4)
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((tid & 63) == 0) s_part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < 16; w++) tot += s_part[w];
        tot %= ((1ull << 61) - 1ull);
        slot[0] = make_uint4((uint32_t)n, (uint32_t)rank, (uint32_t)(tot & 0xFFFFFFFFull), (uint32_t)(tot >> 32));
        slot[1] = make_uint4(0, 0, 0, 0);
    }
}

void launch_exchange_fill_slot(const uint4* d_desc, int n, int slot_keypoints, int rank, uint4* d_slot, hipStream_t s) {
    hipLaunchKernelGGL(exchange_fill_slot_kernel, dim3(1), dim3(1024), 0, s, d_desc, n, slot_keypoints, rank, d_slot);
}

// ---------------- the tracking searches' order-dependent resolve on the device (match_device.h: TrackResolveArgs) ----------------
// One workgroup of 1024 threads; thread t owns the queries t, t + 1024, ... (a query's rank in the reference's walk is its
// index) and keeps their K-lists in registers for all rounds.  Everything the kernel reads from memory - the lists, the
// counts, the position -> keypoint map, the angles, the map slots, the bindings on entry (some of it in pinned host
// memory) - is requested up front, so the launch pays ONE memory latency before the rounds and none after them (the first
// version read them where it needed them: seven dependent latencies, 22 us per launch).
// LDS: claim / taken per candidate position (2 x 16 KB), the current frame's angle per position (16 KB), keypoint index per
// position (8 KB).
namespace {
constexpr int kResK = 8;
constexpr int kResTH_HIGH = 100, kResHisto = 30;  // ORBmatcher.cc:37-39

// ordered compaction helper: exclusive rank of `flag` among the threads of the block (thread order), block total in *total
template <int kResThreads>
__device__ __forceinline__ int block_rank(bool flag, int* s_wave /* >= 17 ints */, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[w] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < kResThreads / 64; i++) {
        const int c = s_wave[i];
        if (i < w) base += c;
        tot += c;
    }
    *total = tot;
    return base + before;
}
}  // namespace

// kResThreads x kResQPT queries: the host launches the instance that holds the search (1024 threads up to 2048 queries;
// beyond, 512 threads with twice the registers each - the K-lists of six or eight queries per thread stay out of scratch)
template <int kResThreads, int kResQPT>
__device__ __forceinline__ void track_resolve_body(const TrackResolveArgs& a) {
    constexpr int kResPPT = kResolveMaxCand / kResThreads;
    __shared__ int s_claim[kResolveMaxCand];
    __shared__ int s_taken[kResolveMaxCand];
    __shared__ float s_angle[kResolveMaxCand];
    __shared__ uint16_t s_item[kResolveMaxCand];
    __shared__ int8_t s_oct[kResolveMaxCand];
    __shared__ int s_unres[2];
    __shared__ int s_wave[kResThreads / 64 + 1];
    __shared__ int s_hist[kResHisto];
    __shared__ int s_flags[2];  // 0: a query ran out of list entries
    __shared__ int s_keep[3];
    const int tid = threadIdx.x;
    const int nc = a.n_cand, nk = a.n_kp, nq = a.nq;
    unsigned long long tk[6];  // (diagnostics: 100 MHz ticks at the phase boundaries, head_host[8..12])
    tk[0] = wall_clock64();
    const bool orient = a.mode == 2 && a.check_orientation != 0;
    const int F = a.mode == 3 ? 2 : 1;
    bool fallback = nc > kResolveMaxCand || nk > kResolveMaxCand || nq > kResQPT * kResThreads || a.K != kResK;
    // ---- the loads from device memory, issued together: K-lists, counts, position -> keypoint map, angles
    uint32_t key[kResQPT][kResK];
    bool un[kResQPT], many[kResQPT];
    float qang[kResQPT];
    int qslot[kResQPT], took[kResQPT];
    int n_active_local = 0;
#pragma unroll
    for (int u = 0; u < kResQPT; u++) {
        const int i = tid + u * kResThreads;
        const bool in = !fallback && i < nq;
        const int c8 = in ? (int)a.cnt8[i] : 0;
#pragma unroll
        for (int k = 0; k < kResK; k++) key[u][k] = in ? a.keys[(size_t)k * nq + i] : 0xFFFFFFFFu;
        qang[u] = (orient && in) ? a.q_angle[i] : 0.f;
        un[u] = c8 != 0;
        many[u] = c8 > kResK;
        took[u] = -1;
        n_active_local += un[u] ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < kResPPT; j++) {
        const int p = tid + j * kResThreads;
        int item = 0, oc = 0;
        float ang = 0.f;
        if (!fallback && p < nc) {
            item = a.cell_items[p];
            oc = a.s_octave[p];
            if (orient) ang = a.cur_angle[item];
        }
        s_item[p] = (uint16_t)item;
        s_oct[p] = (int8_t)oc;
        s_angle[p] = ang;
        s_claim[p] = 0;
        s_taken[p] = -1;
    }
    if (tid < 2) {
        s_flags[tid] = 0;
        s_unres[tid] = 0;
    }
    if (tid < kResHisto) s_hist[tid] = 0;
    __syncthreads();
    // ---- the loads from HOST memory (the queries' map slots sit in the search's pinned staging, the bindings on entry in the
    //      stage's host-mapped block): requested now, needed after the rounds - a PCIe read is 3-4 us, and waiting for a
    //      device load issued behind it would wait for it too (loads complete in order)
    int slot_in[kResPPT];
#pragma unroll
    for (int u = 0; u < kResQPT; u++) {
        const int i = tid + u * kResThreads;
        qslot[u] = (!fallback && i < nq) ? (a.q_slot ? a.q_slot[i] : a.slot_base + i) : -1;
    }
#pragma unroll
    for (int j = 0; j < kResPPT; j++) {
        const int p = tid + j * kResThreads;
        slot_in[j] = (a.kp_slot_in && p < nk) ? a.kp_slot_in[p] : -1;
    }
    if (n_active_local) atomicAdd(&s_flags[1], n_active_local);
    tk[1] = wall_clock64();
    int rounds = 0;
    while (!fallback) {
        // A: every unresolved query looks its list entries up (all reads in flight together) and claims the free ones.
        // A claim is (round, lowest rank) under atomicMax - the round in the high bits, so last round's claims need no clearing
        const int tag = (rounds + 1) << 13;
        bool fr[kResQPT][kResK];
#pragma unroll
        for (int u = 0; u < kResQPT; u++) {
            const int mine = tag | (8191 - (tid + u * kResThreads));
            int tk8[kResK];
            bool ended = false;
#pragma unroll
            for (int k = 0; k < kResK; k++) {
                const uint32_t kk = key[u][k];
                ended = ended || kk == 0xFFFFFFFFu;
                fr[u][k] = un[u] && !ended;
                tk8[k] = s_taken[fr[u][k] ? (int)(kk & 0xFFFFu) : 0];
            }
#pragma unroll
            for (int k = 0; k < kResK; k++) {
                fr[u][k] = fr[u][k] && tk8[k] < 0;  // ORBmatcher.cc:83-85 / 1294-1296: a taken keypoint is passed over
                if (fr[u][k]) atomicMax(&s_claim[key[u][k] & 0xFFFFu], mine);
            }
        }
        if (tid == 0) s_unres[(rounds + 1) & 1] = 0;  // (the other parity's counter: read last round, written next round)
        __syncthreads();
        // B: decisions of the queries nothing undecided can reach, applied at once (`taken` is only read again in the next
        // round's A, behind the barrier; claims are not touched here).  No early exits: every entry is looked at under
        // predicates (a loop with break / continue over the register-resident K-list came out of the compiler deciding
        // "no candidate" for every two-candidate query).
        bool any = false;
#pragma unroll
        for (int u = 0; u < kResQPT; u++) {
            const int mine = tag | (8191 - (tid + u * kResThreads));
            int found = 0, pos0 = 0, pos1 = 0, d0 = 256, d1 = 256;
            bool ended = false;
#pragma unroll
            for (int k = 0; k < kResK; k++) {
                const uint32_t kk = key[u][k];
                ended = ended || kk == 0xFFFFFFFFu;
                const bool free_ = fr[u][k] && found < F;
                if (free_ && found == 0) { pos0 = (int)(kk & 0xFFFFu); d0 = (int)(kk >> 16); }
                if (free_ && found == 1) { pos1 = (int)(kk & 0xFFFFu); d1 = (int)(kk >> 16); }
                found += free_ ? 1 : 0;
            }
            // fewer candidates than the decision reads, all K entries real, more in the window: the list is too short
            const bool exhausted = un[u] && found < F && !ended && many[u];
            if (exhausted) atomicOr(&s_flags[0], 1);
            const int cl0 = s_claim[found >= 1 ? pos0 : 0], cl1 = s_claim[found >= 2 ? pos1 : 0];
            const int l0 = s_oct[found >= 1 ? pos0 : 0], l1 = found >= 2 ? (int)s_oct[pos1] : -1;
            const bool c0 = found >= 1 && cl0 == mine, c1 = found >= 2 && cl1 == mine;
            const bool stable = un[u] && !exhausted && (found == 0 || (c0 && (found < 2 || c1)));
            bool ok = stable && found >= 1 && d0 <= kResTH_HIGH;
            if (F == 2 && ok && l0 == l1 && (float)d0 > a.nn_ratio * (float)d1) ok = false;  // ORBmatcher.cc:112-113
            if (stable || exhausted) un[u] = false;
            if (ok) {
                s_taken[pos0] = tid + u * kResThreads;
                took[u] = pos0;
            }
            any = any || un[u];
        }
        rounds++;
        // anybody left?  (one counter per round parity, one barrier)
        if (__ballot(any) != 0ull && (tid & 63) == 0) atomicAdd(&s_unres[rounds & 1], 1);
        __syncthreads();
        if (s_unres[rounds & 1] == 0) break;
        if (s_flags[0]) break;
    }
    __syncthreads();
    tk[2] = wall_clock64();
    fallback = fallback || s_flags[0] != 0;
    // ---- TrackWithMotionModel's rotation check (ORBmatcher.cc:1319-1350) by the queries' owners
    if (!fallback && orient) {
        int bin[kResQPT];
#pragma unroll
        for (int u = 0; u < kResQPT; u++) {
            bin[u] = -1;
            if (took[u] < 0) continue;
            float rot = qang[u] - s_angle[took[u]];
            if (rot < 0.0f) rot += 360.0f;
            int b = (int)roundf(rot * (1.0f / kResHisto));
            if (b == kResHisto) b = 0;
            bin[u] = b;
            atomicAdd(&s_hist[b], 1);
        }
        __syncthreads();
        if (tid < 64) {
            // ComputeThreeMaxima (ORBmatcher.cc:1475-1506): its strict comparisons pick the three largest non-empty bins,
            // the earlier bin first among equals - lane i ranks bin i against the others instead of one lane walking all thirty
            const int mine = tid < kResHisto ? s_hist[tid] : 0;
            int rank = 0;
            for (int j = 0; j < kResHisto; j++) {
                const int o = s_hist[j];
                rank += (o > mine || (o == mine && j < tid)) ? 1 : 0;
            }
            if (tid < 3) s_keep[tid] = -1;
            __builtin_amdgcn_wave_barrier();
            if (tid < kResHisto && mine > 0 && rank < 3) s_keep[rank] = tid;
            __builtin_amdgcn_wave_barrier();
            if (tid == 0) {
                const int i1 = s_keep[0], i2 = s_keep[1], i3 = s_keep[2];
                const int max1 = i1 >= 0 ? s_hist[i1] : 0, max2 = i2 >= 0 ? s_hist[i2] : 0, max3 = i3 >= 0 ? s_hist[i3] : 0;
                if ((float)max2 < 0.1f * (float)max1) { s_keep[1] = -1; s_keep[2] = -1; }
                else if ((float)max3 < 0.1f * (float)max1) { s_keep[2] = -1; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kResQPT; u++)
            if (bin[u] >= 0 && bin[u] != s_keep[0] && bin[u] != s_keep[1] && bin[u] != s_keep[2]) took[u] = -1;
    }
    // ---- by keypoint index: the query matched to it (s_claim) and that query's map slot (s_taken)
    __syncthreads();
    tk[3] = wall_clock64();
#pragma unroll
    for (int j = 0; j < kResPPT; j++) {
        s_claim[tid + j * kResThreads] = -1;
        s_taken[tid + j * kResThreads] = -1;
    }
    __syncthreads();
    if (!fallback) {
#pragma unroll
        for (int u = 0; u < kResQPT; u++)
            if (took[u] >= 0) {
                const int kp = s_item[took[u]];
                s_claim[kp] = tid + u * kResThreads;
                s_taken[kp] = qslot[u];
            }
    }
    __syncthreads();
    // ---- kp_to_q out, the frame's bindings, the edge list (keypoints with a map point, ascending index)
    tk[4] = wall_clock64();
    int n_edges = 0, n_match = 0;
#pragma unroll
    for (int j = 0; j < kResPPT; j++) {
        const int k = tid + j * kResThreads;
        if (j * kResThreads >= nk) break;  // (uniform)
        int q = -1, slot = -1;
        if (k < nk) {
            q = s_claim[k];
            a.kp_to_q[k] = q;
            slot = slot_in[j] >= 0 ? slot_in[j] : (q >= 0 ? s_taken[k] : -1);
            a.kp_slot_out[k] = slot;
        }
        int tot_e, tot_m;
        const int re = block_rank<kResThreads>(slot >= 0, s_wave, &tot_e);
        if (slot >= 0) {
            a.e_kp[n_edges + re] = k;
            a.e_slot[n_edges + re] = slot;
            a.e_kp_host[n_edges + re] = k;
        }
        (void)block_rank<kResThreads>(q >= 0, s_wave, &tot_m);
        n_edges += tot_e;
        n_match += tot_m;
    }
    if (tid == 0) {
        a.head[0] = n_edges; a.head[1] = n_match; a.head[2] = fallback ? 1 : 0; a.head[3] = rounds;
        a.head_host[0] = n_edges; a.head_host[1] = n_match; a.head_host[2] = fallback ? 1 : 0; a.head_host[3] = rounds;
        a.head_host[4] = s_flags[1];  // (diagnostics: queries with candidates)
        tk[5] = wall_clock64();
        for (int i = 0; i < 5; i++) a.head_host[8 + i] = (int)(tk[i + 1] - tk[i]);
    }
}

template <int kResThreads, int kResQPT>
__global__ __launch_bounds__(kResThreads) void track_resolve_kernel(TrackResolveArgs a) {
    track_resolve_body<kResThreads, kResQPT>(a);
}

// a GROUP of agents' resolves in one launch (so_track_group): workgroup x works on row x of the table
template <int kResThreads, int kResQPT>
__global__ __launch_bounds__(kResThreads) void track_resolve_group_kernel(const TrackResolveArgs* __restrict__ tab) {
    const TrackResolveArgs a = tab[blockIdx.x];
    track_resolve_body<kResThreads, kResQPT>(a);
}

// ---- stage 1 -> stage 2 on the device (match_device.h: TrackLinkArgs) ----
constexpr int kLinkHash = 8192;  // >= 2 x kResolveMaxCand: open addressing over the map slots bound in this frame
__global__ __launch_bounds__(1024) void track_link_kernel(TrackLinkArgs a) {
    __shared__ int s_key[kLinkHash];
    __shared__ unsigned s_skip[kTrackMaxQueryBits / 32];
    __shared__ unsigned s_excl[kTrackMaxCandBits / 32];
    const int tid = threadIdx.x;
    // the local points' slots may sit in pinned host memory (the search reads them there, one per wave): requested first, one
    // coalesced request per 1024 points, and needed last - a PCIe read is microseconds, a dependent chain of them per thread tens
    constexpr int kMaxPerThread = kTrackMaxQueryBits / 1024;
    int lslot[kMaxPerThread];
#pragma unroll
    for (int j = 0; j < kMaxPerThread; j++) {
        const int i = tid + j * 1024;
        lslot[j] = i < a.n_local ? (a.local_slot ? a.local_slot[i] : a.first_slot + i) : -1;
    }
    for (int i = tid; i < kLinkHash; i += 1024) s_key[i] = -1;
    for (int i = tid; i < kTrackMaxQueryBits / 32; i += 1024) s_skip[i] = 0;
    for (int i = tid; i < kTrackMaxCandBits / 32; i += 1024) s_excl[i] = 0;
    __syncthreads();
    // the slots bound behind stage 1 -> hash set
    for (int k = tid; k < a.n_kp; k += 1024) {
        const int slot = a.kp_slot[k];
        if (slot < 0) continue;
        unsigned h = ((unsigned)slot * 2654435761u) >> 19;
        for (;;) {
            const int old = atomicCAS(&s_key[h], -1, slot);
            if (old == -1 || old == slot) break;
            h = (h + 1) & (kLinkHash - 1);
        }
    }
    if (tid == 0) {  // the pose: double quaternion -> the frame's float pose -> the SE3Quat PoseOptimization starts from
        double q[4], t3[3];
        for (int i = 0; i < 4; i++) q[i] = a.pose1[i];
        for (int i = 0; i < 3; i++) t3[i] = a.pose1[4 + i];
        float T[12];
        pose_to_T12_hd(q, t3, T);
        for (int i = 0; i < 12; i++) a.job->T.Tcw[i] = T[i];
        pose_from_T12_hd(T, q, t3);
        for (int i = 0; i < 4; i++) a.pose2_init[i] = q[i];
        for (int i = 0; i < 3; i++) a.pose2_init[4 + i] = t3[i];
    }
    // candidate positions whose keypoint carries a map point: not eligible in SearchByProjection (ORBmatcher.cc:83-85)
    for (int p = tid; p < a.n_cand && p < kTrackMaxCandBits; p += 1024)
        if (a.kp_slot[a.cell_items[p]] >= 0) atomicOr(&s_excl[p >> 5], 1u << (p & 31));
    __syncthreads();
    // local points that are bound in this frame already: not searched (Tracking.cc:966-978)
#pragma unroll
    for (int j = 0; j < kMaxPerThread; j++) {
        const int slot = lslot[j];
        if (slot < 0) continue;
        const int i = tid + j * 1024;
        unsigned h = ((unsigned)slot * 2654435761u) >> 19;
        for (;;) {
            const int v = s_key[h];
            if (v == slot) { atomicOr(&s_skip[i >> 5], 1u << (i & 31)); break; }
            if (v == -1) break;
            h = (h + 1) & (kLinkHash - 1);
        }
    }
    __syncthreads();
    for (int w = tid; w < kTrackMaxCandBits / 32; w += 1024) a.job->T.excl_bits[w] = s_excl[w];
    for (int w = tid; w < (a.n_local + 31) / 32; w += 1024)
        if (s_skip[w]) a.job->T.skip_bits[w] |= s_skip[w];
}

void launch_track_link(const TrackLinkArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(track_link_kernel, dim3(1), dim3(1024), 0, s, a);
}

void launch_track_resolve_group(const TrackResolveArgs* d_tab, int n, int max_nq, hipStream_t s) {
    if (n <= 0) return;
    if (max_nq <= 1024) hipLaunchKernelGGL((track_resolve_group_kernel<1024, 1>), dim3(n), dim3(1024), 0, s, d_tab);
    else if (max_nq <= 2048) hipLaunchKernelGGL((track_resolve_group_kernel<1024, 2>), dim3(n), dim3(1024), 0, s, d_tab);
    else if (max_nq <= 3072) hipLaunchKernelGGL((track_resolve_group_kernel<512, 6>), dim3(n), dim3(512), 0, s, d_tab);
    else hipLaunchKernelGGL((track_resolve_group_kernel<512, 8>), dim3(n), dim3(512), 0, s, d_tab);
}

void launch_track_resolve(const TrackResolveArgs& a, hipStream_t s) {
    if (a.nq <= 1024) hipLaunchKernelGGL((track_resolve_kernel<1024, 1>), dim3(1), dim3(1024), 0, s, a);
    else if (a.nq <= 2048) hipLaunchKernelGGL((track_resolve_kernel<1024, 2>), dim3(1), dim3(1024), 0, s, a);
    else if (a.nq <= 3072) hipLaunchKernelGGL((track_resolve_kernel<512, 6>), dim3(1), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((track_resolve_kernel<512, 8>), dim3(1), dim3(512), 0, s, a);  // (more than 4096 queries: the kernel reports fallback)
}

void launch_hamming_top2(const uint4* d_A, int na, const uint4* d_B, int nb, int32_t* d_best_idx,
                         int32_t* d_best_dist, int32_t* d_second_dist, hipStream_t s) {
    if (na <= 0) return;
    hipLaunchKernelGGL(hamming_top2_kernel, dim3((na + 3) / 4), dim3(256), 0, s, d_A, na, d_B, nb, d_best_idx,
                       d_best_dist, d_second_dist);
}

}  // namespace so
