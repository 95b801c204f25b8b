// ba_dense.hip — blocked dense Cholesky solve of a large reduced camera system (global bundle adjustment,
// Optimizer::BundleAdjustment over a whole map: hundreds to thousands of free keyframes), FP64, gfx950.
//
// Replaces LinearSolverEigen (SimplicialLDLT, linear_solver_eigen.h:94-124) for systems too large for the
// single-workgroup solvers in ba_kernels.hip: same solution up to rounding.
//
// S (np x np, row-major, leading dimension np = n rounded up to 96; the padding is an identity block) is factored
// in place, right-looking, in panels of 96 columns.  Per panel k, three launches:
//   dense_potrf_kernel   one workgroup: Cholesky of the 96x96 diagonal block in LDS, then its inverse by the
//                        recursive 2x2 block formula (6 -> 12 -> 24 -> 48 -> 96: small GEMMs, no serial solve)
//   dense_panel_kernel   L_ik = A_ik Linv_kk^T for every 96-row block below the diagonal: a GEMM, not a
//                        substitution (the diagonal-block-inverse TRSM GPU libraries use); one extra workgroup
//                        advances the forward substitution y_k = Linv_kk b_k
//   dense_update_kernel  A_ij -= L_ik L_jk^T for every 96x96 tile i >= j > k (the n^3/3 flops of the solve) plus
//                        b_j -= L_jk y_k in a few extra workgroups
// and after the last panel one launch per panel for the backward substitution x_k = Linv_kk^T (y_k - ...).
// The two GEMM kernels share one tile routine on the FP64 matrix cores (v_mfma_f64_16x16x4_f64): a 256-thread
// workgroup owns a 96x96 tile, each wave a 48x48 quadrant (3x3 MFMA tiles, 36 accumulators per lane); both
// operand panels are staged through LDS in two K-chunks of 48, row-major with a 50-double stride, which makes
// the MFMA operand fetch (16 rows x 2 k per half-wave) hit 64 distinct banks.
// HBM layout: S is read and written tile by tile exactly once per panel step; at 288 GB the 9000 x 9000 system of
// an eight-agent map (648 MB) stays resident next to the problem.
#include <cstdlib>

#include "ba_device.h"

#pragma clang fp contract(fast)

namespace so {

constexpr int kDNB = 96;        // panel width = tile edge
constexpr int kDChunk = 48;     // K-chunk staged in LDS
constexpr int kDStride = 50;    // LDS row stride (doubles) of a chunk: 2*50 mod 64 = 36 -> 16 rows x 2 k conflict-free

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double dense_rsqrt(double v) {
    double y = __builtin_amdgcn_rsq(v);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const double t = v * y;
        const double e = fma(-t, y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    return y;
}

// ---- padding: rows / columns n..np-1 form an identity block, the right-hand side is zero there ----
__global__ __launch_bounds__(256) void dense_pad_kernel(double* __restrict__ S, double* __restrict__ rhs, int n, int np) {
    const int pad = np - n;
    const size_t total = (size_t)pad * np;  // padded rows, full width
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = n + (int)(i / np), c = (int)(i % np);
        S[(size_t)r * np + c] = (r == c) ? 1.0 : 0.0;
        if (c < n) S[(size_t)c * np + r] = 0.0;  // padded columns of the real rows
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < pad; i += gridDim.x * 256) rhs[n + i] = 0.0;
}

// ---- diagonal block: Cholesky + inverse ----
__global__ __launch_bounds__(256) void dense_potrf_kernel(BaDev d, int k) {
    __shared__ double A[kDNB][kDNB + 1];
    __shared__ double X[kDNB][kDNB + 1];
    __shared__ double s_rinv[kDNB];
    __shared__ int s_bad;
    if (!d.lm->active) return;
    const int tid = threadIdx.x, ld = d.ldS;
    double* Sk = d.S + (size_t)k * kDNB * ld + (size_t)k * kDNB;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < kDNB * kDNB; i += 256) {
        const int r = i / kDNB, c = i - r * kDNB;
        A[r][c] = Sk[(size_t)r * ld + c];
        X[r][c] = 0.0;
    }
    __syncthreads();
    // Blocked inside the workgroup, 16 columns at a time: (1) the 16x16 diagonal sub-block on one wave, a row per
    // lane, pivots and multipliers by v_readlane (no LDS, no barrier); (2) every row below solves against it on its
    // own thread; (3) rank-16 update of what is left, 5x5 register patches per thread.  Three barriers per 16
    // columns instead of three per column.
    const int ty = tid >> 4, tx = tid & 15, lane = tid & 63, wave = tid >> 6;
    for (int jb = 0; jb < kDNB; jb += 16) {
        if (wave == 0) {
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; c++) x[c] = lane < 16 ? A[jb + lane][jb + c] : 0.0;
            double ys[16];
            bool bad = false;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(x[c]), c);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(x[c]), c);
                const double v = __hiloint2double(hi, lo);
                if (!(v > 0.0)) bad = true;
                const double y = dense_rsqrt(v);
                ys[c] = y;
                x[c] *= y;
#pragma unroll
                for (int c2 = c + 1; c2 < 16; c2++) {
                    const int l0 = __builtin_amdgcn_readlane(__double2loint(x[c]), c2);
                    const int l1 = __builtin_amdgcn_readlane(__double2hiint(x[c]), c2);
                    x[c2] = fma(-x[c], __hiloint2double(l1, l0), x[c2]);
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; c++) A[jb + lane][jb + c] = (c <= lane) ? x[c] : 0.0;
            }
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < 16; c++) s_rinv[jb + c] = ys[c];
                if (bad) s_bad = 1;
            }
        }
        __syncthreads();
        const int below = kDNB - jb - 16;
        if (tid < below) {  // x L_dd^T = a for one row
            const int r = jb + 16 + tid;
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; c++) x[c] = A[r][jb + c];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                double v = x[c];
#pragma unroll
                for (int m = 0; m < c; m++) v = fma(-x[m], A[jb + c][jb + m], v);
                x[c] = v * s_rinv[jb + c];
            }
#pragma unroll
            for (int c = 0; c < 16; c++) A[r][jb + c] = x[c];
        }
        __syncthreads();
        const int nb16 = below / 16, base = jb + 16;
        if (nb16 > 0) {
            double acc[5][5];
#pragma unroll
            for (int a = 0; a < 5; a++)
#pragma unroll
                for (int b = 0; b < 5; b++) acc[a][b] = 0.0;
            for (int kk = 0; kk < 16; kk++) {
                double pr[5], pc[5];
#pragma unroll
                for (int a = 0; a < 5; a++) {
                    pr[a] = a < nb16 ? A[base + ty + 16 * a][jb + kk] : 0.0;
                    pc[a] = a < nb16 ? A[base + tx + 16 * a][jb + kk] : 0.0;
                }
#pragma unroll
                for (int a = 0; a < 5; a++)
#pragma unroll
                    for (int b = 0; b <= a; b++) acc[a][b] = fma(pr[a], pc[b], acc[a][b]);
            }
#pragma unroll
            for (int a = 0; a < 5; a++)
#pragma unroll
                for (int b = 0; b <= a; b++)
                    if (a < nb16) A[base + ty + 16 * a][base + tx + 16 * b] -= acc[a][b];
        }
        __syncthreads();
    }
    // inverse of the lower-triangular factor: 6x6 leaves by substitution, then [X11 0; -X22 L21 X11, X22] doubling
    if (tid < kDNB / 6) {
        const int o = 6 * tid;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double x[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                if (i < c) { x[i] = 0.0; continue; }
                double v = (i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++)
                    if (m >= c && m < i) v = fma(-A[o + i][o + m], x[m], v);
                x[i] = v * s_rinv[o + i];
            }
#pragma unroll
            for (int i = 0; i < 6; i++) X[o + i][o + c] = x[i];
        }
    }
    __syncthreads();
    for (int s = 6; s < kDNB; s *= 2) {  // merge pairs of inverted s-blocks into 2s-blocks
        const int pairs = kDNB / (2 * s), per = s * s;
        // T = L21 X11 into the free upper-right block of the pair
        for (int e = tid; e < pairs * per; e += 256) {
            const int p = e / per, q = e - p * per, i = q / s, jx = q - i * s, o = 2 * s * p;
            double v = 0.0;
            for (int m = jx; m < s; m++) v = fma(A[o + s + i][o + m], X[o + m][o + jx], v);  // X11 lower: m >= jx
            X[o + i][o + s + jx] = v;
        }
        __syncthreads();
        // X21 = -X22 T
        for (int e = tid; e < pairs * per; e += 256) {
            const int p = e / per, q = e - p * per, i = q / s, jx = q - i * s, o = 2 * s * p;
            double v = 0.0;
            for (int m = 0; m <= i; m++) v = fma(X[o + s + i][o + s + m], X[o + m][o + s + jx], v);  // X22 lower: m <= i
            X[o + s + i][o + jx] = -v;
        }
        __syncthreads();
        for (int e = tid; e < pairs * per; e += 256) {  // the scratch block is part of the (zero) upper triangle
            const int p = e / per, q = e - p * per, i = q / s, jx = q - i * s, o = 2 * s * p;
            X[o + i][o + s + jx] = 0.0;
        }
        __syncthreads();
    }
    double* Linv = d.dense_ws + (size_t)k * kDNB * kDNB;
    for (int i = tid; i < kDNB * kDNB; i += 256) {
        const int r = i / kDNB, c = i - r * kDNB;
        Sk[(size_t)r * ld + c] = (c <= r) ? A[r][c] : 0.0;
        Linv[i] = X[r][c];
    }
    if (tid == 0 && s_bad) d.partial[kBaSolveOk] = 0.0;
}

// ---- 96x96 tile of C = PA * PB^T on the FP64 matrix cores ----
// PA, PB: 96 rows x 96 k, row-major with leading dimensions lda / ldb.  acc[rt][ct] is the wave's 48x48 quadrant
// as 3x3 MFMA tiles; element (row, col) of tile (rt, ct): col = lane & 15, row = (lane >> 4) + 4 * reg.
__device__ __forceinline__ void dense_tile_nt(const double* __restrict__ PA, int lda, const double* __restrict__ PB,
                                              int ldb, double (*sA)[kDStride], double (*sB)[kDStride], d4 acc[3][3]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 48, wc = (wave & 1) * 48;
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int ct = 0; ct < 3; ct++) acc[rt][ct] = d4{0.0, 0.0, 0.0, 0.0};
    for (int kc = 0; kc < kDNB; kc += kDChunk) {
        __syncthreads();  // the previous chunk is no longer read
        for (int idx = tid; idx < kDNB * (kDChunk / 2); idx += 256) {
            const int r = idx / (kDChunk / 2), v = idx - r * (kDChunk / 2);
            const double2 a = *reinterpret_cast<const double2*>(PA + (size_t)r * lda + kc + 2 * v);
            const double2 b = *reinterpret_cast<const double2*>(PB + (size_t)r * ldb + kc + 2 * v);
            sA[r][2 * v] = a.x; sA[r][2 * v + 1] = a.y;
            sB[r][2 * v] = b.x; sB[r][2 * v + 1] = b.y;
        }
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < kDChunk; kk += 4) {
            double a[3], b[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                a[t] = sA[wr + 16 * t + fr][kk + fk];
                b[t] = sB[wc + 16 * t + fr][kk + fk];
            }
#pragma unroll
            for (int rt = 0; rt < 3; rt++)
#pragma unroll
                for (int ct = 0; ct < 3; ct++)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b[ct], acc[rt][ct], 0, 0, 0);
        }
    }
}

// ---- panel: L_ik = A_ik Linv_kk^T (blocks 1..), forward substitution y_k = Linv_kk b_k (block 0) ----
__global__ __launch_bounds__(256) void dense_panel_kernel(BaDev d, int k) {
    __shared__ double sA[kDNB][kDStride];
    __shared__ double sB[kDNB][kDStride];
    if (!d.lm->active) return;
    const int tid = threadIdx.x, ld = d.ldS;
    const double* Linv = d.dense_ws + (size_t)k * kDNB * kDNB;
    if (blockIdx.x == 0) {
        double* b = d.bs + (size_t)k * kDNB;
        __shared__ double s_b[kDNB];
        if (tid < kDNB) s_b[tid] = b[tid];
        __syncthreads();
        if (tid < kDNB) {
            double v = 0.0;
            for (int m = 0; m <= tid; m++) v = fma(Linv[tid * kDNB + m], s_b[m], v);
            b[tid] = v;
        }
        return;
    }
    const int i = k + blockIdx.x;  // block row
    double* Aik = d.S + (size_t)i * kDNB * ld + (size_t)k * kDNB;
    d4 acc[3][3];
    dense_tile_nt(Aik, ld, Linv, kDNB, sA, sB, acc);
    __syncthreads();  // every wave has finished reading A_ik (through LDS) before it is overwritten
    const int lane = tid & 63, wave = tid >> 6, wr = (wave >> 1) * 48, wc = (wave & 1) * 48;
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int ct = 0; ct < 3; ct++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = wc + 16 * ct + (lane & 15);
                Aik[(size_t)r * ld + c] = acc[rt][ct][reg];
            }
}

// ---- trailing update: A_ij -= L_ik L_jk^T for i >= j > k; extra blocks: b_j -= L_jk y_k ----
// mode 0: every tile; mode 1: the tiles of block column k+1 (what the next panel needs) + the right-hand side;
// mode 2: the rest (runs while the side stream factors panel k+1)
__global__ __launch_bounds__(256) void dense_update_kernel(BaDev d, int k, int n_tiles, int mode) {
    __shared__ double sA[kDNB][kDStride];
    __shared__ double sB[kDNB][kDStride];
    if (!d.lm->active) return;
    const int tid = threadIdx.x, ld = d.ldS;
    const int T = d.ldS / kDNB, rem = T - k - 1;  // trailing block rows
    if ((int)blockIdx.x >= n_tiles) {  // right-hand side rows
        __shared__ double s_y[kDNB];
        if (tid < kDNB) s_y[tid] = d.bs[(size_t)k * kDNB + tid];
        __syncthreads();
        const int row = (k + 1) * kDNB + ((int)blockIdx.x - n_tiles) * 256 + tid;
        if (row < ld) {
            const double* L = d.S + (size_t)row * ld + (size_t)k * kDNB;
            double v = 0.0;
            for (int m = 0; m < kDNB; m += 2) {
                const double2 l = *reinterpret_cast<const double2*>(L + m);
                v = fma(l.x, s_y[m], v);
                v = fma(l.y, s_y[m + 1], v);
            }
            d.bs[row] -= v;
        }
        return;
    }
    // tile index -> (ti >= tj) within the trailing rem x rem block grid
    int ti, tj;
    if (mode == 1) {
        ti = (int)blockIdx.x;
        tj = 0;
    } else {
        int t = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);  // row by row of a lower triangle
        while ((t + 1) * (t + 2) / 2 <= (int)blockIdx.x) t++;
        while (t * (t + 1) / 2 > (int)blockIdx.x) t--;
        ti = t;
        tj = (int)blockIdx.x - t * (t + 1) / 2;
        if (mode == 2) {  // the triangle without its first column
            ti++;
            tj++;
        }
    }
    if (ti >= rem) return;
    const int I = k + 1 + ti, J = k + 1 + tj;
    const double* Pi = d.S + (size_t)I * kDNB * ld + (size_t)k * kDNB;
    const double* Pj = d.S + (size_t)J * kDNB * ld + (size_t)k * kDNB;
    double* C = d.S + (size_t)I * kDNB * ld + (size_t)J * kDNB;
    d4 acc[3][3];
    dense_tile_nt(Pi, ld, Pj, ld, sA, sB, acc);
    const int lane = tid & 63, wave = tid >> 6, wr = (wave >> 1) * 48, wc = (wave & 1) * 48;
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int ct = 0; ct < 3; ct++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = wc + 16 * ct + (lane & 15);
                C[(size_t)r * ld + c] -= acc[rt][ct][reg];
            }
}

// ---- backward substitution, one launch per panel from the last to the first ----
// every workgroup recomputes x_k = Linv_kk^T y_k (96x96 mat-vec) and then updates its 256 entries of y above
// the panel: y_c -= sum_m L[k*96+m][c] x_k[m]; workgroup 0 stores x_k.
__global__ __launch_bounds__(256) void dense_backward_kernel(BaDev d, int k) {
    __shared__ double s_y[kDNB], s_x[kDNB], s_part[2][kDNB];
    if (!d.lm->active) return;
    const int tid = threadIdx.x, ld = d.ldS;
    const double* Linv = d.dense_ws + (size_t)k * kDNB * kDNB;
    if (tid < kDNB) s_y[tid] = d.bs[(size_t)k * kDNB + tid];
    __syncthreads();
    // x_k = Linv^T y: thread (half, col) sums half of the rows of column col; 16 loads in flight per thread
    if (tid < 2 * kDNB) {
        const int col = tid % kDNB, half = tid / kDNB;
        const int m_lo = half == 0 ? col : (col < 48 ? 48 : col), m_hi = half == 0 ? (col < 48 ? 48 : col) : kDNB;
        double v = 0.0;
        for (int m0 = m_lo; m0 < m_hi; m0 += 16) {
            double l[16];
#pragma unroll
            for (int m = 0; m < 16; m++) l[m] = (m0 + m < m_hi) ? Linv[(size_t)(m0 + m) * kDNB + col] : 0.0;
#pragma unroll
            for (int m = 0; m < 16; m++) v = fma(l[m], (m0 + m < m_hi) ? s_y[m0 + m] : 0.0, v);
        }
        s_part[half][col] = v;
    }
    __syncthreads();
    if (tid < kDNB) s_x[tid] = s_part[0][tid] + s_part[1][tid];
    __syncthreads();
    const int c = blockIdx.x * 256 + tid;
    if (c < k * kDNB) {
        const double* L = d.S + (size_t)k * kDNB * ld + c;
        double v = 0.0;
        for (int m0 = 0; m0 < kDNB; m0 += 16) {  // 16 independent loads in flight per thread
            double l[16];
#pragma unroll
            for (int m = 0; m < 16; m++) l[m] = L[(size_t)(m0 + m) * ld];
#pragma unroll
            for (int m = 0; m < 16; m++) v = fma(l[m], s_x[m0 + m], v);
        }
        d.bs[c] -= v;
    }
    if (blockIdx.x == 0 && tid < kDNB) d.dense_x[(size_t)k * kDNB + tid] = s_x[tid];
}

__global__ __launch_bounds__(256) void dense_finish_kernel(BaDev d) {
    if (!d.lm->active) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 6 * d.n_free) d.bs[i] = d.dense_x[i];
    if (i == 0 && !(d.partial[kBaSolveOk] == 0.0)) d.partial[kBaSolveOk] = 1.0;
}

__global__ void dense_begin_kernel(BaDev d) {
    if (!d.lm->active) return;
    d.partial[kBaSolveOk] = 0.5;  // becomes 0 if a pivot fails, 1 at the end otherwise
}

void launch_ba_dense_pad(const BaDev& d, hipStream_t s) {
    const int n = 6 * d.n_free, np = d.ldS;
    if (np > n) hipLaunchKernelGGL(dense_pad_kernel, dim3(256), dim3(256), 0, s, d.S, d.bs, n, np);
}

// Look-ahead across launches: the trailing update of panel k is split into the tiles of block column k+1 (U1) and
// the rest (U2).  As soon as U1 is done the side stream factors the next diagonal block and solves the next panel
// while the main stream is still busy with U2 - the serial diagonal factor (80 us) hides behind the GEMM work for as
// long as the trailing matrix is large.  Only for T >= 32 panels: below that the two cross-stream waits per panel
// (~10 us each) cost more than the overlap returns (measured on GBA-1, 19 panels).
void launch_ba_dense_solve(const BaDev& d, hipStream_t s) {
    const int T = d.ldS / kDNB;
    hipStream_t side = d.dense_side;
    hipEvent_t* ev = d.dense_events;  // [0] start, [1 + 2k] U1(k) done, [2 + 2k] panel(k+1) done
    const bool lookahead = side && ev && T >= 32 && T <= kDenseMaxPanels && !getenv("SWARMORB_DENSE_NO_LOOKAHEAD");
    hipLaunchKernelGGL(dense_begin_kernel, dim3(1), dim3(1), 0, s, d);
    hipLaunchKernelGGL(dense_potrf_kernel, dim3(1), dim3(256), 0, s, d, 0);
    hipLaunchKernelGGL(dense_panel_kernel, dim3(1 + (T - 1)), dim3(256), 0, s, d, 0);
    for (int k = 0; k < T - 1; k++) {
        const int rem = T - k - 1;
        const int rhs_blocks = (rem * kDNB + 255) / 256;
        if (lookahead && rem >= 3) {
            hipLaunchKernelGGL(dense_update_kernel, dim3(rem + rhs_blocks), dim3(256), 0, s, d, k, rem, 1);
            (void)hipEventRecord(ev[1 + 2 * k], s);
            (void)hipStreamWaitEvent(side, ev[1 + 2 * k], 0);
            hipLaunchKernelGGL(dense_potrf_kernel, dim3(1), dim3(256), 0, side, d, k + 1);
            hipLaunchKernelGGL(dense_panel_kernel, dim3(1 + (rem - 1)), dim3(256), 0, side, d, k + 1);
            (void)hipEventRecord(ev[2 + 2 * k], side);
            const int rest = (rem - 1) * rem / 2;
            hipLaunchKernelGGL(dense_update_kernel, dim3(rest), dim3(256), 0, s, d, k, rest, 2);
            (void)hipStreamWaitEvent(s, ev[2 + 2 * k], 0);
        } else {
            const int n_tiles = rem * (rem + 1) / 2;
            hipLaunchKernelGGL(dense_update_kernel, dim3(n_tiles + rhs_blocks), dim3(256), 0, s, d, k, n_tiles, 0);
            hipLaunchKernelGGL(dense_potrf_kernel, dim3(1), dim3(256), 0, s, d, k + 1);
            hipLaunchKernelGGL(dense_panel_kernel, dim3(1 + (rem - 1)), dim3(256), 0, s, d, k + 1);
        }
    }
    for (int k = T - 1; k >= 0; k--) {
        const int blocks = (k * kDNB + 255) / 256;
        hipLaunchKernelGGL(dense_backward_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, d, k);
    }
    hipLaunchKernelGGL(dense_finish_kernel, dim3((6 * d.n_free + 255) / 256), dim3(256), 0, s, d);
}

}  // namespace so
