// frame_kernels.hip — gfx950 kernels of the Frame post-processing (code/src/Frame.cc:277-292, 316-375, 431-443,
// 454-514; code/src/MapPoint.cc:466-485).  Compiled contraction-free like the ORB kernels: every floating-point
// operation below is the one the CPU oracle performs, in the same order, so results are bit-identical.
// Conventions for the arithmetic the reference delegates to un-vendored OpenCV / libm (parity unpinned):
// see oracle/frame_oracle.h.
#include "frame_device.h"
#include "orb_device.h"

namespace so {

// ln x, fixed operation sequence (no libm): x = m 2^e, m in [sqrt(1/2), sqrt(2)), 2 atanh((m-1)/(m+1)) by 12 odd terms
__device__ __forceinline__ double frame_log(double x) {
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

// cv::undistortPoints(p, K, D, R = I, P = K), criteria COUNT 5
__device__ __forceinline__ void frame_undistort_point(const FrameCam& cam, float u, float v, float& uo, float& vo) {
    const double fx = (double)cam.fx, fy = (double)cam.fy, cx = (double)cam.cx, cy = (double)cam.cy;
    const double k1 = (double)cam.k1, k2 = (double)cam.k2, p1 = (double)cam.p1, p2 = (double)cam.p2, k3 = (double)cam.k3;
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
        const double deltaX = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
        const double deltaY = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    uo = (float)(fx * x + cx);
    vo = (float)(fy * y + cy);
}

// One workgroup: bounds (4 corner points), undistortion, cell of every keypoint, per-cell histogram + exclusive scan,
// and the grid lists: a cell's keypoints in index order, which is the push_back order of AssignFeaturesToGrid.
// The lists are a counting sort: every keypoint takes a slot of its cell's range in arrival order (an LDS atomic), then
// ranks itself among the cell's handful of entries by index - three barriers where the bitonic sort of
// (cell << 14 | index) keys this kernel started with needed fifty-five; the cell of a keypoint is worked out in the
// undistortion loop, from the value still in the register, instead of after a round trip through global memory.
__device__ __forceinline__ void frame_prepare_body(const FramePrepareArgs& a) {
    __shared__ int s_slot[kFrameMaxKeypoints];        // arrival-order slot -> keypoint index, then final position -> index
    __shared__ uint16_t s_cell[kFrameMaxKeypoints];   // cell of every keypoint (0xFFFF: outside the grid)
    __shared__ int s_hist[kFrameGridCols * kFrameGridRows + 1];
    __shared__ int s_start[kFrameGridCols * kFrameGridRows + 1];
    __shared__ int s_wsum[16];
    __shared__ float s_b[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // device-resident frame: the keypoint count is a device word written by the extractor's last kernel
    const int n = a.ex_total ? min(max(*a.ex_total, 0), a.n) : a.n;
    const SelectedKp* meta = static_cast<const SelectedKp*>(a.ex_meta);
    constexpr int ncell = kFrameGridCols * kFrameGridRows;
    static_assert(ncell == 3 * 1024, "three cells per thread in the scan below");
    if (a.do_bounds == 1) {
        __shared__ float s_c[4][2];
        if (tid < 4) {
            if (a.cam.k1 != 0.0f) {
                const float u = (tid & 1) ? (float)a.width : 0.0f, v = (tid & 2) ? (float)a.height : 0.0f;
                frame_undistort_point(a.cam, u, v, s_c[tid][0], s_c[tid][1]);
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (a.cam.k1 != 0.0f) {
                s_b[0] = fminf(s_c[0][0], s_c[2][0]);
                s_b[1] = fmaxf(s_c[1][0], s_c[3][0]);
                s_b[2] = fminf(s_c[0][1], s_c[1][1]);
                s_b[3] = fmaxf(s_c[2][1], s_c[3][1]);
            } else {
                s_b[0] = 0.0f; s_b[1] = (float)a.width; s_b[2] = 0.0f; s_b[3] = (float)a.height;
            }
            for (int i = 0; i < 4; i++) a.bounds[i] = s_b[i];
        }
    } else if (a.do_bounds == 2) {  // the caller knows them (they depend on the camera only): no corner undistortion, no load
        if (tid < 4) {
            s_b[tid] = a.bounds_value[tid];
            a.bounds[tid] = a.bounds_value[tid];
        }
    } else if (tid < 4) {
        s_b[tid] = a.bounds[tid];
    }
    for (int c = tid; c <= ncell; c += 1024) s_hist[c] = 0;
    __syncthreads();
    const float inv_w = (float)kFrameGridCols / (s_b[1] - s_b[0]);  // Frame.cc:259-260
    const float inv_h = (float)kFrameGridRows / (s_b[3] - s_b[2]);
    for (int i = tid; i < n; i += 1024) {
        float u, v;
        if (a.do_undistort) {
            if (meta) {  // ORBextractor::operator() (ORBextractor.cc:808-814): level coordinates -> level-0 pixels
                const SelectedKp k = meta[i];
                u = (float)k.x;
                v = (float)k.y;
                if (k.level != 0) {
                    u *= a.scale[k.level];
                    v *= a.scale[k.level];
                }
                a.octave[i] = (int8_t)k.level;
            } else {
                u = a.xy[2 * i];
                v = a.xy[2 * i + 1];
            }
            if (a.cam.k1 != 0.0f) frame_undistort_point(a.cam, u, v, u, v);  // else mvKeysUn = mvKeys
            a.xy_un[2 * i] = u;
            a.xy_un[2 * i + 1] = v;
            if (a.xy_un_host) {
                a.xy_un_host[2 * i] = u;
                a.xy_un_host[2 * i + 1] = v;
            }
        } else {
            u = a.xy_un[2 * i];
            v = a.xy_un[2 * i + 1];
        }
        if (a.do_grid) {
            const int px = (int)roundf((u - s_b[0]) * inv_w);  // PosInGrid
            const int py = (int)roundf((v - s_b[2]) * inv_h);
            int cell = -1;
            if (!(px < 0 || px >= kFrameGridCols || py < 0 || py >= kFrameGridRows)) {
                cell = px * kFrameGridRows + py;
                atomicAdd(&s_hist[cell], 1);
            }
            s_cell[i] = (uint16_t)cell;
            if (a.cell_of) a.cell_of[i] = cell;
        }
    }
    if (a.desc_by_index) {
        const uint4* src = reinterpret_cast<const uint4*>(a.ex_desc);
        for (int j = tid; j < 2 * n; j += 1024) a.desc_by_index[j] = src[j];
    }
    if (a.angle_by_index && a.ex_angle)
        for (int j = tid; j < n; j += 1024) a.angle_by_index[j] = a.ex_angle[j];
    if (a.header_host && tid == 0) {
        a.header_host[0] = n;
        a.header_host[2] = 0;
        a.header_host[3] = 0;
    }
    if (a.header_host && tid < 4) a.header_host[4 + tid] = __float_as_int(s_b[tid]);
    if (!a.do_grid) return;
    __threadfence_block();  // (the gathers at the end read xy_un / octave other threads wrote)
    __syncthreads();
    // exclusive scan of the 3072 cell counts: three cells per thread, a wave scan, the sixteen wave totals through LDS
    const int c0 = s_hist[3 * tid], c1 = s_hist[3 * tid + 1], c2 = s_hist[3 * tid + 2];
    const int mine = c0 + c1 + c2;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = 0, inside = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const int x = s_wsum[w];
        if (w < wave) base += x;
        inside += x;
    }
    const int before = base + incl - mine;
    s_start[3 * tid] = before;
    s_start[3 * tid + 1] = before + c0;
    s_start[3 * tid + 2] = before + c0 + c1;
    a.cell_start[3 * tid] = before;
    a.cell_start[3 * tid + 1] = before + c0;
    a.cell_start[3 * tid + 2] = before + c0 + c1;
    s_hist[3 * tid] = 0;  // (re-used as the arrival counters of the counting sort)
    s_hist[3 * tid + 1] = 0;
    s_hist[3 * tid + 2] = 0;
    if (tid == 1023) {
        a.cell_start[ncell] = inside;
        *a.n_inside = inside;
        if (a.header_host) a.header_host[1] = inside;
        if (a.col_start) a.col_start[kFrameGridCols] = inside;
    }
    if (a.col_start) {  // first position of each grid column = start of its first cell
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int c = 3 * tid + k;
            if (c % kFrameGridRows == 0) a.col_start[c / kFrameGridRows] = before + (k > 0 ? c0 : 0) + (k > 1 ? c1 : 0);
        }
    }
    __syncthreads();
    // counting sort, stable by index: arrival-order slots first ...
    for (int i = tid; i < n; i += 1024) {
        const int cell = s_cell[i];
        if (cell != 0xFFFF) s_slot[s_start[cell] + atomicAdd(&s_hist[cell], 1)] = i;
    }
    __syncthreads();
    // ... then every keypoint ranks itself among its cell's entries (a handful) by index
    int fin_pos[kFrameMaxKeypoints / 1024];
#pragma unroll
    for (int r = 0; r < kFrameMaxKeypoints / 1024; r++) {
        const int i = tid + 1024 * r;
        fin_pos[r] = -1;
        if (i < n) {
            const int cell = s_cell[i];
            if (cell != 0xFFFF) {
                const int lo = s_start[cell], cnt = s_hist[cell];
                int rank = 0;
                for (int k = 0; k < cnt; k++) rank += s_slot[lo + k] < i ? 1 : 0;
                fin_pos[r] = lo + rank;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kFrameMaxKeypoints / 1024; r++)
        if (fin_pos[r] >= 0) s_slot[fin_pos[r]] = tid + 1024 * r;
    __syncthreads();
    for (int i = tid; i < inside; i += 1024) {
        const int idx = s_slot[i];
        a.cell_items[i] = idx;
        if (a.perm_host) a.perm_host[i] = idx;
        if (a.s_xy) {  // the matcher's candidate layout (match_device.h), position = rank in grid-traversal order
            a.s_xy[i] = make_float2(a.xy_un[2 * idx], a.xy_un[2 * idx + 1]);
            a.s_octave[i] = a.octave[idx];
        }
    }
    if (a.s_desc) {
        const uint4* src = reinterpret_cast<const uint4*>(a.ex_desc);
        for (int j = tid; j < 2 * inside; j += 1024) a.s_desc[j] = src[2 * s_slot[j >> 1] + (j & 1)];
    }
}

__global__ __launch_bounds__(1024) void frame_prepare_kernel(FramePrepareArgs a) { frame_prepare_body(a); }
// several frames in one launch (so_dframe_group_submit): workgroup b prepares the frame of the extraction group's member b - row
// prep_sel of the member's `rot` rows, unless the member sat the chain out
__global__ __launch_bounds__(1024) void frame_prepare_batch_kernel(const FramePrepareArgs* __restrict__ A, int rot,
                                                                    const ExtractBatchMember* __restrict__ M) {
    const ExtractBatchMember& m = M[blockIdx.x];
    if (m.skip) return;
    frame_prepare_body(A[blockIdx.x * rot + m.prep_sel]);
}

void launch_frame_prepare(const FramePrepareArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(frame_prepare_kernel, dim3(1), dim3(1024), 0, s, a);
}
void launch_frame_prepare_batch(const FramePrepareArgs* d_args, int rot, const ExtractBatchMember* d_members, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(frame_prepare_batch_kernel, dim3(n), dim3(1024), 0, s, d_args, rot, d_members);
}

// Frame::isInFrustum + MapPoint::PredictScale, one thread per map point
__global__ __launch_bounds__(256) void frame_frustum_kernel(FrameFrustumArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const float* T = a.Tcw;
    float Ow[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {  // mOw = -mRcw.t()*mtcw: double accumulation, one rounding
        const double s = (double)T[0 + j] * (double)T[3] + (double)T[4 + j] * (double)T[7] + (double)T[8 + j] * (double)T[11];
        Ow[j] = (float)(-s);
    }
    uint8_t ok = 0;
    const float P[3] = {a.Xw[3 * i], a.Xw[3 * i + 1], a.Xw[3 * i + 2]};
    float Pc[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double s = (double)T[4 * r] * (double)P[0] + (double)T[4 * r + 1] * (double)P[1] + (double)T[4 * r + 2] * (double)P[2];
        Pc[r] = (float)(s + (double)T[4 * r + 3]);
    }
    do {
        if (Pc[2] < 0.0f) break;
        const float invz = 1.0f / Pc[2];
        const float u = a.cam.fx * Pc[0] * invz + a.cam.cx;
        const float v = a.cam.fy * Pc[1] * invz + a.cam.cy;
        if (u < a.bounds[0] || u > a.bounds[1]) break;
        if (v < a.bounds[2] || v > a.bounds[3]) break;
        const float maxD = 1.2f * a.max_dist[i], minD = 0.8f * a.min_dist[i];
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        const double n2 = (double)PO[0] * (double)PO[0] + (double)PO[1] * (double)PO[1] + (double)PO[2] * (double)PO[2];
        const float dist = (float)sqrt(n2);
        if (dist < minD || dist > maxD) break;
        const double dot = (double)PO[0] * (double)a.normal[3 * i] + (double)PO[1] * (double)a.normal[3 * i + 1] +
                           (double)PO[2] * (double)a.normal[3 * i + 2];
        const float vc = (float)(dot / (double)dist);
        if (vc < a.viewing_cos_limit) break;
        const float ratio = a.max_dist[i] / dist;
        const float lr = (float)frame_log((double)ratio);
        int nScale = (int)ceilf(lr / a.log_scale_factor);
        if (nScale > a.n_scale_levels - 1) nScale = a.n_scale_levels - 1;
        if (nScale < 0) nScale = 0;
        ok = 1;
        a.proj_x[i] = u;
        a.proj_y[i] = v;
        a.view_cos[i] = vc;
        a.pred_level[i] = nScale;
    } while (false);
    a.in_view[i] = ok;
}

__global__ __launch_bounds__(256) void map_scatter_positions_kernel(float* __restrict__ Xw, const int32_t* __restrict__ slots,
                                                                     const float* __restrict__ X, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = slots[i];
    Xw[3 * (size_t)s] = X[3 * i];
    Xw[3 * (size_t)s + 1] = X[3 * i + 1];
    Xw[3 * (size_t)s + 2] = X[3 * i + 2];
}

void launch_map_scatter_positions(float* d_Xw, const int32_t* slots, const float* X, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(map_scatter_positions_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_Xw, slots, X, n);
}

// rows slots[i] of the map table <- row i of the given arrays (any of X / N / mx / mn may be null = unchanged)
__global__ __launch_bounds__(256) void map_scatter_rows_kernel(float* __restrict__ Xw, float* __restrict__ normal, float* __restrict__ max_d,
                                                                float* __restrict__ min_d, const int32_t* __restrict__ slots,
                                                                const float* __restrict__ X, const float* __restrict__ N,
                                                                const float* __restrict__ mx, const float* __restrict__ mn, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t s = (size_t)slots[i];
    if (X) { Xw[3 * s] = X[3 * i]; Xw[3 * s + 1] = X[3 * i + 1]; Xw[3 * s + 2] = X[3 * i + 2]; }
    if (N) { normal[3 * s] = N[3 * i]; normal[3 * s + 1] = N[3 * i + 1]; normal[3 * s + 2] = N[3 * i + 2]; }
    if (mx) max_d[s] = mx[i];
    if (mn) min_d[s] = mn[i];
}

void launch_map_scatter_rows(float* d_Xw, float* d_normal, float* d_max, float* d_min, const int32_t* slots, const float* X,
                             const float* N, const float* mx, const float* mn, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(map_scatter_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_Xw, d_normal, d_max, d_min, slots, X, N, mx, mn, n);
}

// rows first .. first + n - 1 of the map table <- the five arrays of so_map_write, read in place from its pinned staging block
// (dword copies: 3 + 3 + 1 + 1 + 8 per row; any source may be null = that column is left alone).  One launch instead of five
// hipMemcpyAsync calls (five ~5 us copy kernels + their bookkeeping per keyframe).
__global__ __launch_bounds__(256) void map_write_range_kernel(uint32_t* __restrict__ Xw, uint32_t* __restrict__ normal, uint32_t* __restrict__ max_d,
                                                               uint32_t* __restrict__ min_d, uint32_t* __restrict__ desc,
                                                               const uint32_t* __restrict__ X, const uint32_t* __restrict__ N,
                                                               const uint32_t* __restrict__ mx, const uint32_t* __restrict__ mn,
                                                               const uint32_t* __restrict__ D, int first, int n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, f = (size_t)first, sn = (size_t)n;
    if (X && i < 3 * sn) Xw[3 * f + i] = X[i];
    if (N && i < 3 * sn) normal[3 * f + i] = N[i];
    if (mx && i < sn) max_d[f + i] = mx[i];
    if (mn && i < sn) min_d[f + i] = mn[i];
    if (D && i < 8 * sn) desc[8 * f + i] = D[i];
}

void launch_map_write_range(float* d_Xw, float* d_normal, float* d_max, float* d_min, uint8_t* d_desc, const float* X, const float* N,
                            const float* mx, const float* mn, const uint8_t* D, int first, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(map_write_range_kernel, dim3((8 * n + 255) / 256), dim3(256), 0, s, (uint32_t*)d_Xw, (uint32_t*)d_normal, (uint32_t*)d_max,
                       (uint32_t*)d_min, (uint32_t*)d_desc, (const uint32_t*)X, (const uint32_t*)N, (const uint32_t*)mx, (const uint32_t*)mn,
                       (const uint32_t*)D, first, n);
}

void launch_frame_frustum(const FrameFrustumArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(frame_frustum_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}

}  // namespace so
