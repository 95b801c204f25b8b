// ba_dense.hip — blocked dense Cholesky solve of a large reduced camera system (global bundle adjustment,
// Optimizer::BundleAdjustment over a whole map: hundreds to thousands of free keyframes), FP64, gfx950.
//
// Replaces LinearSolverEigen (SimplicialLDLT, linear_solver_eigen.h:94-124) for systems too large for the
// single-workgroup solvers in ba_kernels.hip: same solution up to rounding.
//
// S (np x np, row-major, leading dimension np = n rounded up to 96; the padding is an identity block) is factored
// in place, right-looking, in panels of 96 columns taken two at a time (dense_chain / launch_ba_dense_solve):
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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "ba_device.h"

#pragma clang fp contract(fast)

namespace so {

constexpr int kDNB = 96;        // panel width = tile edge
constexpr int kDChunk = 48;     // K-chunk staged in LDS
constexpr int kDStride = 50;    // LDS row stride (doubles) of a chunk: 2*50 mod 64 = 36 -> 16 rows x 2 k conflict-free

#include "ba_solve_mfma.inc"  // d4, dense_rsqrt, kPS, the pivot-tile machinery, ba_solve_mfma_body

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

// ---- diagonal block: Cholesky + inverse, one workgroup, FP64 matrix cores for everything but the 16x16 pivots ----
// The lower triangle of the 96x96 block lives in LDS as 16x16 tiles (row stride 98 doubles: the 16 rows x 2 k of
// an MFMA operand fetch hit 32 distinct bank pairs).  Six steps of 16 columns, two barriers each:
//   panel   the row tiles below the diagonal tile become A_rd W^T (W = inverse of the 16x16 diagonal factor): one
//           16x16x16 MFMA product per tile instead of a substitution;
//   update  rank-16 update of the remaining lower triangle, one MFMA product per tile.  Wave 0 takes the next
//           diagonal tile first and then factors it - a row per lane, pivots and multipliers by v_readlane (no LDS,
//           no barrier), W by forward substitution on the same multipliers - while waves 1-3 finish the update and
//           compute block row jb of the inverse of the whole factor (X_ij = -W_i sum_m L_im X_mj, rows above it
//           are complete): the serial pivots hide everything else.
// Only the inverse goes back to HBM: the panel GEMM and both substitutions use Linv_kk, nothing reads L_kk.
constexpr int kDenseGroup = 3;   // panels per trailing update (K = 96 * kDenseGroup) when the look-ahead runs; measured on
                                 // GBA-2 (94 panels): 1 -> 125 ms, 2 -> 106, 3 -> 104, 4 -> 106, 6 -> 115; without look-ahead
                                 // (GBA-1, 19 panels) the serial chain dominates and single panels are fastest
constexpr int kPT = kDNB / 16;  // tiles per block edge

// Block row i (16 rows) of the inverse factor from LDS to the full row-major 96x96 block in HBM the panel GEMM reads
// (zeros above the diagonal: those tiles were never written in LDS), by `nthr` threads of which the caller is `t`.
__device__ __forceinline__ void potrf_store_inverse_row(double (*X)[kPS], double* __restrict__ Linv, int i, int t, int nthr) {
    for (int e = t; e < 16 * (kDNB / 2); e += nthr) {
        const int r = 16 * i + e / (kDNB / 2), c = 2 * (e % (kDNB / 2));
        double2 x;
        const bool low = (c >> 4) <= i;
        x.x = low ? X[r][c] : 0.0; x.y = low ? X[r][c + 1] : 0.0;
        *reinterpret_cast<double2*>(Linv + (size_t)r * kDNB + c) = x;
    }
}

// The block's lower 16x16 tiles are in A (LDS); on return X holds the inverse of its Cholesky factor (lower tiles; the
// diagonal tiles are zero above the diagonal, tiles above the diagonal are never written) and Linv (HBM) the same as
// a full 96x96 block: block row i leaves for HBM during step i + 1, from the waves that are not factoring, so only
// the last one's store is exposed (the whole block at the end was 4.2 k cycles of every panel's critical path).
// *s_bad must be 0 on entry (and a barrier between that store and the call); it becomes 1 if a pivot is not positive.
__device__ __forceinline__ void potrf_block_lds(double (*A)[kPS], double (*X)[kPS], int* s_bad_p, double* __restrict__ Linv) {
    // (the wave index as a scalar: what each wave works on is then scalar arithmetic and scalar branches - computed per
    //  lane, the tile indices of a pair of trailing tiles cost 1.1 k cycles, as much as its eight MFMAs)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int crow = lane >> 4, ccol = lane & 15;  // MFMA result layout: element (crow + 4 reg, ccol)
    int& s_bad = *s_bad_p;
    SO_POTRF_MARK(0);
    if (wave == 0 && !potrf_diag16<kPS>(&A[0][0], nullptr, &X[0][0], lane) && lane == 0) s_bad = 1;
    __syncthreads();
    for (int jb = 0; jb < kPT; jb++) {
        const int cb = 16 * jb;
        SO_POTRF_MARK(1 + 4 * jb);
        // panel: L_rd = A_rd W^T
        for (int rt = jb + 1 + wave; rt < kPT; rt += 4) {
            const d4 acc = potrf_mma_nt(&A[16 * rt][cb], &X[cb][cb], d4{0.0, 0.0, 0.0, 0.0}, lane);
            potrf_wave_sync();
#pragma unroll
            for (int reg = 0; reg < 4; reg++) A[16 * rt + crow + 4 * reg][cb + ccol] = acc[reg];
        }
        if (jb + 1 < kPT) __syncthreads();
        SO_POTRF_MARK(2 + 4 * jb);
        const int m = kPT - jb - 1;               // trailing tile rows
        const bool has_diag = m > 0;              // wave 0: next diagonal tile, then its factor
        if (has_diag && wave == 0) {
            const int nb = cb + 16;
            const d4 acc = potrf_mma_nt(&A[nb][cb], &A[nb][cb], d4{0.0, 0.0, 0.0, 0.0}, lane);
#pragma unroll
            for (int reg = 0; reg < 4; reg++) A[nb + crow + 4 * reg][nb + ccol] -= acc[reg];
            potrf_wave_sync();
            SO_POTRF_MARK(3 + 4 * jb);
            if (!potrf_diag16<kPS>(&A[nb][nb], nullptr, &X[nb][nb], lane) && lane == 0) s_bad = 1;
            SO_POTRF_MARK(4 + 4 * jb);
        } else {
            const int nw = has_diag ? 3 : 4, me = has_diag ? wave - 1 : wave;
            if (jb > 0) potrf_store_inverse_row(X, Linv, jb - 1, 64 * me + lane, 64 * nw);  // complete since the last barrier
            // Work items of the step, longest first, handed out in snake order (rounds alternate direction):
            //   m tiles of the trailing matrix (m >= 2): block column jb + 1 below its diagonal tile and the diagonal
            //     tile (jb + 2, jb + 2) - what the NEXT step reads.  They take all their updates at once,
            //     A_ij -= L_ik L_jk^T for k = 0..jb (jb + 1 products, one pass over the tile), everything to the right of
            //     column jb + 1 waits (left-looking).  Updating the whole trailing matrix every step (14 / 9 / 5 / 2
            //     single products, one read-modify-write and one set of tile indices each) kept three waves busy for
            //     5.5 k cycles in the first step against the 3.9 k of the wave that factors; this way the steps carry
            //     5 / 8 / 9 / 8 products in 5 / 4 / 3 / 2 tiles and every step hides behind the pivot tile;
            //   jb tiles of block row jb of the inverse, X_ij = -W_i sum_{m=j}^{i-1} L_im X_mj (jb - j + 1 products).
            const int n_upd = m >= 2 ? m : 0;
            const int n_items = n_upd + jb;
            for (int r = 0;; r++) {
                const int u = r * nw + ((r & 1) ? nw - 1 - me : me);
                if (u >= n_items) break;
                if (u < n_upd) {
                    const int ri = u + 1 < n_upd ? jb + 2 + u : jb + 2, ci = u + 1 < n_upd ? jb + 1 : jb + 2;
                    // (each product from a zero accumulator and subtracted on its own: the same roundings as a rank-16
                    //  update per step; one accumulator for all of them moved GBA-2's per-edge chi2 past 1e-5 relative)
                    d4 tot;
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) tot[reg] = A[16 * ri + crow + 4 * reg][16 * ci + ccol];
                    for (int k = 0; k <= jb; k++)
                        tot -= potrf_mma_nt(&A[16 * ri][16 * k], &A[16 * ci][16 * k], d4{0.0, 0.0, 0.0, 0.0}, lane);
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) A[16 * ri + crow + 4 * reg][16 * ci + ccol] = tot[reg];
                } else {
                    const int i = jb, j = u - n_upd;
                    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
                    for (int mm = j; mm < i; mm++) acc = potrf_mma_nn(&A[16 * i][16 * mm], &X[16 * mm][16 * j], acc, lane);
                    potrf_wave_sync();
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) X[16 * i + crow + 4 * reg][16 * j + ccol] = acc[reg];  // scratch
                    potrf_wave_sync();
                    acc = potrf_mma_nn(&X[16 * i][16 * i], &X[16 * i][16 * j], d4{0.0, 0.0, 0.0, 0.0}, lane);
                    potrf_wave_sync();
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) X[16 * i + crow + 4 * reg][16 * j + ccol] = -acc[reg];
                    potrf_wave_sync();
                }
            }
        }
        __syncthreads();
    }
    SO_POTRF_MARK(25);
    potrf_store_inverse_row(X, Linv, kPT - 1, tid, 256);
}

__global__ __launch_bounds__(256) void dense_potrf_kernel(BaDev d, int k) {
    __shared__ double A[kDNB][kPS];
    __shared__ double X[kDNB][kPS];
    __shared__ int s_bad;
    if (d.lm->active != d.stage) return;
    const int tid = threadIdx.x, ld = d.ldS;
    const double* Sk = d.S + (size_t)k * kDNB * ld + (size_t)k * kDNB;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < (kPT * (kPT + 1) / 2) * 128; i += 256) {  // lower tiles only, 128 double2 per tile
        int ti, tj;
        potrf_tri(i >> 7, ti, tj);
        const int e = i & 127, r = 16 * ti + (e >> 3), c = 16 * tj + 2 * (e & 7);
        const double2 v = *reinterpret_cast<const double2*>(Sk + (size_t)r * ld + c);
        A[r][c] = v.x; A[r][c + 1] = v.y;
    }
    __syncthreads();
    potrf_block_lds(A, X, &s_bad, d.dense_ws + (size_t)k * kDNB * kDNB);
    SO_POTRF_MARK(26);
    if (tid == 0 && s_bad) d.partial[kBaSolveOk] = 0.0;
}

// (the single-workgroup MFMA solve of a local window, ba_solve_mfma_body<THREADS>: ba_solve_mfma.inc)
__global__ __launch_bounds__(kSolveMfmaThreads) void ba_solve_mfma_kernel(BaDev d) {
    extern __shared__ double s_tiles[];        // NT (NT + 1) / 2 tiles
    ba_solve_mfma_body<kSolveMfmaThreads>(d, s_tiles);
}

// the single-workgroup solves of a GROUP of windows (so_ba_group): workgroup y solves the system of member y
__global__ __launch_bounds__(kSolveMfmaThreads) void ba_solve_mfma_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    extern __shared__ double s_tiles[];
    const BaDev d = rows[A.row[blockIdx.y]];
    ba_solve_mfma_body<kSolveMfmaThreads>(d, s_tiles);
}

void launch_ba_solve_mfma_group(const BaDev* d_rows, const BaGroupArgs& A, size_t lds, hipStream_t s) {
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_mfma_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(double) * (kSolveMfmaMaxTiles * (kSolveMfmaMaxTiles + 1) / 2) * kMTile));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(ba_solve_mfma_group_kernel, dim3(1, A.n), dim3(kSolveMfmaThreads), lds, s, d_rows, A);
}


// ---- the same solve for 30..43 free keyframes: 259 unknowns + right-hand side = 17 x 17 tiles, 306 KB - more than
// LDS holds but not more than the register file (512 KB per CU).  Sixteen waves: the strictly-lower tiles live in
// the MFMA result registers of waves 1-15 (ten tiles each at most; the three waves on wave 0's SIMD hold the
// leftmost block columns, which are final early), the pivot tiles (then W_k in their place) and the
// L tiles of the current block column in LDS, where the MFMA operands are fetched from.  Per 16 columns: owners of
// the column's tiles turn them into L = A W^T (through LDS once: result layout -> operand layout) and leave them in
// the panel buffer; every owner applies L_i L_j^T to its tiles without touching LDS for the result; wave 0 updates,
// factors and inverts the next pivot tile meanwhile.  The backward substitution sums each wave's tiles of a block
// column in registers and meets in LDS once per 16 unknowns.
constexpr int kRegMaxTiles = 17;
constexpr int kRegOwners = 15;
constexpr int kRegSlots = (kRegMaxTiles * (kRegMaxTiles - 1) / 2 + kRegOwners - 1) / kRegOwners;  // 10
constexpr int kRegThreads = 1024;
constexpr int kRegEarly = 3 * kRegSlots;  // tiles held by the three waves on wave 0's SIMD

__global__ __launch_bounds__(kRegThreads) void ba_solve_mfma_reg_kernel(BaDev d) {
    __shared__ double s_diag[kRegMaxTiles][kMTile];   // pivot tiles; W_k replaces tile k once it is factored
    __shared__ double s_panel[kRegMaxTiles][kMTile];  // L_i of the current block column, row-major (operand fetch)
    __shared__ double s_lastL[kMTile];
    __shared__ double s_y[kRegMaxTiles][16];          // forward-substituted right-hand side, block by block
    __shared__ double s_x[16 * kRegMaxTiles];
    __shared__ double s_zp[16][16];
    __shared__ double s_z[16];
    __shared__ unsigned char s_ti[kRegMaxTiles * (kRegMaxTiles - 1) / 2], s_tj[kRegMaxTiles * (kRegMaxTiles - 1) / 2];
    __shared__ int s_bad;
    if (d.lm->active != d.stage) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int crow = lane >> 4, ccol = lane & 15;
    const int n = 6 * d.n_free, ld = d.ldS;
    const int NT = (n + 1 + 15) / 16, tr = NT - 1, rr = n - 16 * tr;
    const int n_off = NT * (NT - 1) / 2;
    if (tid == 0) s_bad = 0;
    if (tid < n_off) {  // strictly-lower tiles, block column by block column (j ascending, i = j+1 .. NT-1)
        int j = 0, rem = tid;
        while (rem >= NT - 1 - j) {
            rem -= NT - 1 - j;
            j++;
        }
        s_ti[tid] = (unsigned char)(j + 1 + rem);
        s_tj[tid] = (unsigned char)j;
    }
    for (int i = tid; i < NT * 16; i += kRegThreads) s_x[i] = 0.0;
    for (int e = tid; e < NT * 256; e += kRegThreads) {  // pivot tiles
        const int k = e >> 8, r = (e >> 4) & 15, c = e & 15, gr = 16 * k + r, gc = 16 * k + c;
        double v = 0.0;
        if (gr < n && gc < n) v = d.S[(size_t)gr * ld + gc];
        else if (gr == n && gc < n) v = d.bs[gc];
        else if (gr == gc) v = (gr == n) ? 1e300 : 1.0;  // beta and the identity padding
        s_diag[k][r * kMS + c] = v;
    }
    __syncthreads();
    // Two code paths with the same barrier sequence: wave 0 only ever factors pivot tiles and runs the substitution
    // (its registers belong to potrf_diag16), waves 1-15 only ever hold tiles (their registers belong to C).
    if (wave == 0) {
        if (!potrf_diag16<kMS>(s_diag[0], NT == 1 ? s_lastL : nullptr, s_diag[0], lane) && lane == 0) s_bad = 1;
        __syncthreads();
        for (int jb = 0; jb + 1 < NT; jb++) {
            SO_POTRF_MARK(1 + 3 * jb);
            __syncthreads();  // the panel of block column jb is in LDS
            SO_POTRF_MARK(2 + 3 * jb);
            double* t = s_diag[jb + 1];
            const double* p = s_panel[jb + 1];
            const d4 acc = potrf_mma_nt<kMS>(p, p, d4{0.0, 0.0, 0.0, 0.0}, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) t[(crow + 4 * r) * kMS + ccol] -= acc[r];
            potrf_wave_sync();
            if (!potrf_diag16<kMS>(t, jb + 1 == tr ? s_lastL : nullptr, t, lane) && lane == 0) s_bad = 1;
            SO_POTRF_MARK(3 + 3 * jb);
            __syncthreads();
        }
        SO_POTRF_MARK(58);
        __syncthreads();  // the last block column's panel is done
        // backward substitution x_k = W_k^T (y_k - sum_{i>k} L_ik^T x_i); rows >= n of x stay zero
        for (int k = tr; k >= 0; k--) {
            __syncthreads();  // the owners' partial sums for block k are in s_zp
            const int mk = (k == tr) ? rr : 16;
            if (lane < 16) {
                double acc = 0.0;
#pragma unroll
                for (int w16 = 1; w16 < 16; w16++) acc += s_zp[w16][lane];
                const double yk = (k == tr) ? (lane < rr ? s_lastL[rr * kMS + lane] : 0.0) : s_y[k][lane];
                s_z[lane] = yk - acc;
            }
            potrf_wave_sync();
            const double* W = s_diag[k];
            double xv = 0.0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = crow + 4 * r;
                if (row < mk) xv = fma(W[row * kMS + ccol], s_z[row], xv);
            }
            xv = rows_sum(xv);
            if (lane < 16) {
                const bool real = lane < mk;
                s_x[16 * k + lane] = real ? xv : 0.0;
                if (real) d.bs[16 * k + lane] = xv;
            }
            __syncthreads();
        }
        SO_POTRF_MARK(59);
        if (lane == 0) d.partial[kBaSolveOk] = s_bad ? 0.0 : 1.0;
        return;
    }
    // owner waves.  Waves 4, 8, 12 share wave 0's SIMD: they take the first kRegEarly tiles in block-column order -
    // tiles of the leftmost columns, final after a step or two - so that the pivot chain has its SIMD to itself for
    // the rest of the factorisation; the other twelve waves share the remaining tiles.  Tile coordinates are
    // wave-uniform (SGPRs).
    const bool early = (wave & 3) == 0;
    const int others = wave - 1 - (wave >> 2);  // 0..11 among the waves with (wave & 3) != 0
    d4 C[kRegSlots];
    int my_i[kRegSlots], my_j[kRegSlots];
#pragma unroll
    for (int q = 0; q < kRegSlots; q++) {
        const int t = early ? 3 * q + (wave >> 2) - 1 : kRegEarly + 12 * q + others;
        const bool have = early ? t < (n_off < kRegEarly ? n_off : kRegEarly) : t < n_off;
        my_i[q] = __builtin_amdgcn_readfirstlane(have ? (int)s_ti[have ? t : 0] : -1);
        my_j[q] = __builtin_amdgcn_readfirstlane(have ? (int)s_tj[have ? t : 0] : -1);
        C[q] = d4{0.0, 0.0, 0.0, 0.0};
        if (have) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int gr = 16 * my_i[q] + crow + 4 * r, gc = 16 * my_j[q] + ccol;  // gc < gr: never on the diagonal
                double v = 0.0;
                if (gr < n) v = d.S[(size_t)gr * ld + gc];   // gc < gr < n
                else if (gr == n) v = d.bs[gc];              // the right-hand side row (gc < n)
                C[q][r] = v;
            }
        }
    }
    __syncthreads();  // W_0 is in LDS
    for (int jb = 0; jb < NT; jb++) {
        // panel: my tiles of block column jb become L = A W^T, stay in registers and go to the panel buffer
#pragma unroll
        for (int q = 0; q < kRegSlots; q++) {
            if (my_j[q] != jb) continue;
            double* t = s_panel[my_i[q]];
#pragma unroll
            for (int r = 0; r < 4; r++) t[(crow + 4 * r) * kMS + ccol] = C[q][r];
            potrf_wave_sync();
            const d4 acc = potrf_mma_nt<kMS>(t, s_diag[jb], d4{0.0, 0.0, 0.0, 0.0}, lane);
            potrf_wave_sync();
#pragma unroll
            for (int r = 0; r < 4; r++) t[(crow + 4 * r) * kMS + ccol] = acc[r];
            C[q] = acc;
            if (my_i[q] == tr && crow == (rr & 3)) {  // row rr of tile row tr is the right-hand side: y_jb
                const int reg = rr >> 2;
                s_y[jb][ccol] = reg == 0 ? acc[0] : reg == 1 ? acc[1] : reg == 2 ? acc[2] : acc[3];
            }
        }
        if (jb + 1 >= NT) break;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kRegSlots; q++) {  // my tiles to the right of the column
            if (my_j[q] <= jb) continue;
            const d4 acc = potrf_mma_nt<kMS>(s_panel[my_i[q]], s_panel[my_j[q]], d4{0.0, 0.0, 0.0, 0.0}, lane);
            C[q] -= acc;
        }
        for (int jj = jb + 2 + others; !early && jj < NT; jj += 12) {  // the later pivot tiles (jb + 1 is wave 0's)
            double* t = s_diag[jj];
            const d4 acc = potrf_mma_nt<kMS>(s_panel[jj], s_panel[jj], d4{0.0, 0.0, 0.0, 0.0}, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) t[(crow + 4 * r) * kMS + ccol] -= acc[r];
        }
        __syncthreads();
    }
    __syncthreads();
    for (int k = tr; k >= 0; k--) {
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < kRegSlots; q++) {
            if (my_j[q] != k) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) part = fma(C[q][r], s_x[16 * my_i[q] + crow + 4 * r], part);
        }
        part = rows_sum(part);
        if (lane < 16) s_zp[wave][lane] = part;
        __syncthreads();
        __syncthreads();  // x_k is in LDS
    }
}

bool launch_ba_solve_mfma(const BaDev& d, hipStream_t s) {
    const int n = 6 * d.n_free, NT = (n + 1 + 15) / 16;
    static const bool force_reg = getenv("SWARMORB_MFMA_REG") != nullptr;  // A/B switch for profiling
    if ((NT > kSolveMfmaMaxTiles || (force_reg && NT >= 2)) && NT <= kRegMaxTiles) {
        if (g_ba_recorder) {
            BaLaunchRec rec;
            rec.kind = kBaKSolo;
            rec.d = d;
            rec.solo = [d](hipStream_t st) { hipLaunchKernelGGL(ba_solve_mfma_reg_kernel, dim3(1), dim3(kRegThreads), 0, st, d); };
            rec.phase = g_ba_recorder->phase;
            g_ba_recorder->list.push_back(rec);
            return true;
        }
        hipLaunchKernelGGL(ba_solve_mfma_reg_kernel, dim3(1), dim3(kRegThreads), 0, s, d);
        return true;
    }
    if (NT > kSolveMfmaMaxTiles || NT < 2) return false;
    const size_t lds = sizeof(double) * (size_t)(NT * (NT + 1) / 2) * kMTile;
    if (g_ba_recorder) {  // a member of a so_ba_group: the solve goes out with the other members' (ba_kernels.hip)
        BaLaunchRec rec;
        rec.kind = kBaKSolveMfma;
        rec.d = d;
        rec.grid = 1;
        rec.lds = lds;
        rec.phase = g_ba_recorder->phase;
        g_ba_recorder->list.push_back(rec);
        return true;
    }
    static bool attr_set[64] = {};  // the attribute is per device; racing threads set the same value
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(double) * (kSolveMfmaMaxTiles * (kSolveMfmaMaxTiles + 1) / 2) * kMTile));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(ba_solve_mfma_kernel, dim3(1), dim3(kSolveMfmaThreads), lds, s, d);
    return true;
}

// ---- 96x96 tile of C = PA * PB^T on the FP64 matrix cores ----
// PA, PB: 96 rows x klen k (96 or 192: one or two panels), row-major with leading dimensions lda / ldb.  acc[rt][ct] is the wave's 48x48 quadrant
// as 3x3 MFMA tiles; element (row, col) of tile (rt, ct): col = lane & 15, row = (lane >> 4) + 4 * reg.
template <bool ZERO = true>
__device__ __forceinline__ void dense_tile_nt(const double* __restrict__ PA, int lda, const double* __restrict__ PB,
                                              int ldb, double (*sA)[kDStride], double (*sB)[kDStride], d4 acc[3][3],
                                              int klen = kDNB) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 48, wc = (wave & 1) * 48;
    const int fr = lane & 15, fk = lane >> 4;
    if (ZERO) {
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++) acc[rt][ct] = d4{0.0, 0.0, 0.0, 0.0};
    }
    // a K-chunk is 96 rows x 24 double2 per operand: 9 double2 per thread and operand.  The next chunk is fetched
    // into registers while the matrix cores work on the current one (global latency hides behind 108 MFMAs a wave).
    constexpr int kPer = kDNB * (kDChunk / 2) / 256;
    double2 ra[kPer], rb[kPer];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int idx = tid + 256 * q, r = idx / (kDChunk / 2), v = idx - r * (kDChunk / 2);
            ra[q] = *reinterpret_cast<const double2*>(PA + (size_t)r * lda + kc + 2 * v);
            rb[q] = *reinterpret_cast<const double2*>(PB + (size_t)r * ldb + kc + 2 * v);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int idx = tid + 256 * q, r = idx / (kDChunk / 2), v = idx - r * (kDChunk / 2);
            sA[r][2 * v] = ra[q].x; sA[r][2 * v + 1] = ra[q].y;
            sB[r][2 * v] = rb[q].x; sB[r][2 * v + 1] = rb[q].y;
        }
    };
    fetch(0);
    __syncthreads();  // the caller's previous use of sA / sB is over
    stage();
    __syncthreads();
    for (int kc = 0; kc < klen; kc += kDChunk) {
        const bool more = kc + kDChunk < klen;
        if (more) fetch(kc + kDChunk);
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
        if (more) {
            __syncthreads();  // everyone is done reading this chunk
            stage();
            __syncthreads();
        }
    }
}

// ---- panel: L_ik = A_ik Linv_kk^T (blocks 1..), forward substitution y_k = Linv_kk b_k (block 0) ----
__global__ __launch_bounds__(256) void dense_panel_kernel(BaDev d, int k) {
    __shared__ double sA[kDNB][kDStride];
    __shared__ double sB[kDNB][kDStride];
    if (d.lm->active != d.stage) return;
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
    if (d.tile_first[i] > k) return;  // A_ik lies left of row i's envelope: structurally zero, and stays zero
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

// ---- trailing update: A_ij -= sum over kw panels from k of L_i. L_j.^T; extra blocks: b_j -= L_j,p y_p ----
// Panels are consumed in groups (kw panels, K = 96 kw): every 96x96 tile of the trailing matrix is read and written once
// per group instead of once per 96 columns - that read-modify-write of C, not the MFMA rate, bounds the solve of a
// 9000 x 9000 system.  Tiles: ncols == 0: every tile I >= J >= j0; ncols > 0: the block columns j0 .. j0+ncols-1
// only (what the next panels' factor needs first).  rhs_panel >= 0: n_rhs more workgroups apply that panel's y.
__global__ __launch_bounds__(256) void dense_update_kernel(BaDev d, int k, int kw, const int2* __restrict__ tiles,
                                                           int n_tiles, int rhs_panel) {
    __shared__ double sA[kDNB][kDStride];
    __shared__ double sB[kDNB][kDStride];
    if (d.lm->active != d.stage) return;
    const int tid = threadIdx.x, ld = d.ldS;
    if ((int)blockIdx.x >= n_tiles) {  // right-hand side rows below panel rhs_panel
        const int row0 = (rhs_panel + 1) * kDNB + ((int)blockIdx.x - n_tiles) * 256;
        // the 256 rows of this workgroup span at most four tiles; L(row, rhs_panel) is zero left of the row's envelope
        const int row = row0 + tid;
        if (row >= ld || d.tile_first[row / kDNB] > rhs_panel) return;
        const double* y = d.bs + (size_t)rhs_panel * kDNB;
        const double* L = d.S + (size_t)row * ld + (size_t)rhs_panel * kDNB;
        double v = 0.0;
        for (int m = 0; m < kDNB; m += 2) {
            const double2 l = *reinterpret_cast<const double2*>(L + m);
            v = fma(l.x, y[m], v);
            v = fma(l.y, y[m + 1], v);
        }
        d.bs[row] -= v;
        return;
    }
    const int2 t = tiles[blockIdx.x];
    const int I = t.x, J = t.y;
    // panels of the group both rows have inside their envelopes: [lo, k + kw)
    const int lo = max(k, max(d.tile_first[I], d.tile_first[J]));
    const int klen = (k + kw - lo) * kDNB;
    if (klen <= 0) return;
    const double* Pi = d.S + (size_t)I * kDNB * ld + (size_t)lo * kDNB;
    const double* Pj = d.S + (size_t)J * kDNB * ld + (size_t)lo * kDNB;
    double* C = d.S + (size_t)I * kDNB * ld + (size_t)J * kDNB;
    d4 acc[3][3];
    dense_tile_nt(Pi, ld, Pj, ld, sA, sB, acc, klen);
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
    if (d.lm->active != d.stage) return;
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
    if (c < k * kDNB && c / kDNB >= d.tile_first[k]) {  // L(k, c / 96) is zero left of row k's envelope
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
    if (d.lm->active != d.stage) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 6 * d.n_free) d.bs[i] = d.dense_x[i];
    if (i == 0 && !(d.partial[kBaSolveOk] == 0.0)) d.partial[kBaSolveOk] = 1.0;
}

__global__ void dense_begin_kernel(BaDev d) {
    if (d.lm->active != d.stage) return;
    d.partial[kBaSolveOk] = 0.5;  // becomes 0 if a pivot fails, 1 at the end otherwise
}

// ---- single-launch tile dataflow (mid-size systems: local windows of 44..128 free keyframes, small global maps) ----
// The blocked factorisation above costs 3 launches per panel plus one per panel for the backward substitution; with 3-8
// panels every launch is a 5-25 us island and a 64-keyframe window pays 17 of them per LM trial.  Here ONE launch runs
// the whole solve: a 256-thread workgroup per 96x96 tile of the block skyline, all of them resident at once (at most
// kFlowMaxTiles, one per CU: 152 KB of LDS each), meeting through epoch-stamped flags in HBM (agent-scope release /
// acquire; the epoch is a kernel argument that grows with every solve, so nothing is ever reset).
//   tile (I, J), J < I   left-looking: acc = S_IJ - sum_k L_Ik L_Jk^T as the column tiles k become ready (whatever prefix
//                        is ready goes through the matrix cores in one K-loop), waits for the inverse factor of diagonal
//                        block J, L_IJ = acc Linv_J^T straight from LDS, publishes L_IJ; then the vector the forward
//                        substitution needs from this tile, L_IJ y_J
//   tile (J, J)          the same accumulation - plus a private copy of the sub-diagonal tile (J, J-1), so that the step
//                        on the critical path (Linv_{J-1} arrives -> L_{J,J-1} -> rank-96 update of (J, J)) runs from LDS
//                        without a trip through HBM and a second flag -, Cholesky + inverse in LDS (potrf_block_lds),
//                        publishes Linv_J; y_J = Linv_J (b_J - sum_k [L_Jk y_k]) from the row's forward vectors in
//                        ascending k; then x_J = Linv_J^T (y_J - sum_{i>J} L_iJ^T x_i), rows folded in from the bottom up
//                        as their x_i arrive (the last one, L_{J+1,J}, waits in LDS): fixed summation orders,
//                        deterministic.  b, y and x share d.bs (every reader of y_J is done before x_J exists).
// Critical path per panel (tools/flow_probe.py, 64 keyframes): factor + inverse 17 us, inverse to HBM + flag 3 us,
// L_{J+1,J} 8.4 us, update 7 us; the backward substitution adds ~6.5 us per panel at the end.
constexpr int kFlowDiagLeadDefault = 12, kFlowDiagLeadMax = 32;  // ticketed kernel: levels a diagonal tile is picked up early
constexpr int kFlowMaxTiles = 231;   // 21 panels dense (2016 unknowns, 336 keyframes); a skyline may reach further
constexpr int kFlowSlots = 256;      // flag / vector slots per kind: tile (I, J) -> I (I + 1) / 2 + J
constexpr int kFlowFlagTile = 0, kFlowFlagFwd = kFlowSlots, kFlowFlagY = 2 * kFlowSlots, kFlowFlagX = 2 * kFlowSlots + 32,
              kFlowFlagBad = 2 * kFlowSlots + 64, kFlowFlagAbort = 2 * kFlowSlots + 65;
constexpr int kFlowLdsDoubles = 2 * kDNB * kDStride + kDNB * kPS;  // GEMM chunks + one full tile >= the factor's A and X
static_assert(kFlowLdsDoubles >= 2 * kDNB * kPS, "potrf_block_lds needs two padded blocks");
static_assert(kFlowFlagAbort < kFlowFlagWords, "flag words");
static_assert(kFlowMaxTiles <= kFlowSlots && 22 * 21 / 2 <= kFlowSlots, "tile slots");

// -DSO_FLOW_PROBE (developer builds only): wall-clock marks per workgroup and stage, read back by tools/flow_probe.py
#ifdef SO_FLOW_PROBE
__device__ unsigned long long g_flow_marks[256][16];
#define SO_FLOW_MARK(i) do { if (threadIdx.x == 0) g_flow_marks[blockIdx.x][(i)] = wall_clock64(); } while (0)
__device__ unsigned long long g_flow_diag[10][512];  // per block column: factor published, y published, x published
#define SO_FLOW_DIAG(which, j) do { if (threadIdx.x == 0) g_flow_diag[(which)][(j)] = wall_clock64(); } while (0)
#define SO_FLOW_WNOTE(v) do { if ((threadIdx.x & 63) == 0) g_flow_marks[blockIdx.x][12 + (threadIdx.x >> 6)] = (unsigned long long)(v); } while (0)
#define SO_FLOW_NOTE(i, v) do { if (threadIdx.x == 0) g_flow_marks[blockIdx.x][(i)] = (unsigned long long)(v); } while (0)
#else
#define SO_FLOW_MARK(i)
#define SO_FLOW_NOTE(i, v)
#define SO_FLOW_WNOTE(v)
#define SO_FLOW_DIAG(which, j)
#endif

__device__ __forceinline__ bool flow_ready(const unsigned* f, unsigned epoch) {
    return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
}
// Every wait of the dataflow kernels is bounded.  The workgroups of a launch wait for each other, so all of them must be
// resident; the host checks that with the runtime's occupancy figure and its own budget (ba.cpp), but neither sees another
// process on the GPU, a CU mask or a device whose partitioning changed.  So a wait that lasts longer than the budget
// (wall clock, d.flow_timeout_ticks at 100 MHz) raises the ABORT stamp of this solve - in the flag block, for the other
// workgroups, and in host-mapped memory, for the host - and returns; from then on every wait of every workgroup returns
// at once (after at most 256 polls), the kernel runs to its end on whatever it finds in memory, writes "solve failed",
// and the host repeats the call on the chain-of-launches path (ba.cpp), whose kernels do not wait for each other.
struct FlowWatch {
    unsigned* abort_w;           // device word of the flag block: == epoch when some workgroup gave up
    unsigned* abort_host;        // host-mapped copy (may be null)
    unsigned long long deadline; // wall_clock64() value after which a wait gives up
    unsigned epoch;
    bool dead;
};
__device__ __forceinline__ FlowWatch flow_watch(const BaDev& d, unsigned* abort_w, unsigned epoch) {
    FlowWatch W;
    W.abort_w = abort_w;
    W.abort_host = d.flow_abort_host;
    W.deadline = wall_clock64() + d.flow_timeout_ticks;
    W.epoch = epoch;
    // A solve of this call that already gave up (the host zeroes the word per call and only looks at it after the whole
    // chain of enqueued trials): the trials behind it would each wait their full budget again before the host can fall
    // back - they start dead instead and run through at once.
    W.dead = W.abort_host != nullptr && __hip_atomic_load(W.abort_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
    return W;
}
// true when the solve has been given up (by this workgroup just now, or by another one): stop waiting
__device__ __forceinline__ bool flow_expired(FlowWatch& W) {
    if (flow_ready(W.abort_w, W.epoch)) {
        W.dead = true;
    } else if (wall_clock64() > W.deadline) {
        __hip_atomic_store(W.abort_w, W.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (W.abort_host) __hip_atomic_store(W.abort_host, W.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        W.dead = true;
    }
    return W.dead;
}
__device__ __forceinline__ void flow_wait(const unsigned* f, unsigned epoch, FlowWatch& W) {  // one thread; flow_acquire() after the barrier
    if (W.dead) return;
    unsigned polls = 0;
    while (!flow_ready(f, epoch)) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 255u) == 0 && flow_expired(W)) return;
    }
}
__device__ __forceinline__ void flow_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// every thread's global stores so far become visible device-wide, then the flag goes up
__device__ __forceinline__ void flow_publish(unsigned* f, unsigned epoch) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// The 96-vectors of the substitutions (y_J, x_J, a tile's L_IJ y_J) travel without cache maintenance: agent-scope atomic
// stores write through to memory, agent-scope atomic loads read from there, so neither side pays the L2 write-back /
// invalidate of the tile-sized hand-overs (measured per backward step: 6.5 -> see DESIGN 5).
__device__ __forceinline__ void flow_vec_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double flow_vec_load(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the workgroup's flow_vec_store()s have been acknowledged, then the flag goes up
__device__ __forceinline__ void flow_publish_vec(unsigned* f, unsigned epoch) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc = sT (96 x 96 in LDS, row stride kPS) * PB^T with PB = Linv, a lower-triangular 96 x 96 block, row-major in HBM:
// the panel GEMM with its left operand already on chip.  Column tile c of the product only needs k < 16 (c + 1); to keep
// the four waves level each wave half takes the column tiles {0, 3, 5} or {1, 2, 4} (11 and 10 sixteenths of the full K
// instead of 6 and 15): acc[rt][ct] is row tile (wave >> 1) * 3 + rt, column tile flow_trsm_ctile(wave, ct).
__device__ __forceinline__ int flow_trsm_ctile(int wave, int ct) {
    return (wave & 1) ? (ct == 0 ? 1 : ct == 1 ? 2 : 4) : (ct == 0 ? 0 : ct == 1 ? 3 : 5);
}
__device__ __forceinline__ void dense_tile_lds_nt(const double (*sT)[kPS], const double* __restrict__ PB, int ldb,
                                                  double (*sB)[kDStride], d4 acc[3][3]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 48;
    const int fr = lane & 15, fk = lane >> 4;
    int crow[3], klim[3];
#pragma unroll
    for (int ct = 0; ct < 3; ct++) {
        const int c = flow_trsm_ctile(wave, ct);
        crow[ct] = 16 * c + fr;
        klim[ct] = 16 * (c + 1);
    }
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int ct = 0; ct < 3; ct++) acc[rt][ct] = d4{0.0, 0.0, 0.0, 0.0};
    constexpr int kPer = kDNB * (kDChunk / 2) / 256;
    double2 rb[kPer];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int idx = tid + 256 * q, r = idx / (kDChunk / 2), v = idx - r * (kDChunk / 2);
            rb[q] = *reinterpret_cast<const double2*>(PB + (size_t)r * ldb + kc + 2 * v);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int idx = tid + 256 * q, r = idx / (kDChunk / 2), v = idx - r * (kDChunk / 2);
            sB[r][2 * v] = rb[q].x; sB[r][2 * v + 1] = rb[q].y;
        }
    };
    fetch(0);
    __syncthreads();  // the caller's previous use of sB is over, its stores to sT are complete
    stage();
    __syncthreads();
    for (int kc = 0; kc < kDNB; kc += kDChunk) {
        const bool more = kc + kDChunk < kDNB;
        if (more) fetch(kc + kDChunk);
#pragma unroll 2
        for (int kk = 0; kk < kDChunk; kk += 4) {
            double a[3];
#pragma unroll
            for (int t = 0; t < 3; t++) a[t] = sT[wr + 16 * t + fr][kc + kk + fk];
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
                if (kc + kk < klim[ct]) {  // wave-uniform
                    const double b = sB[crow[ct]][kk + fk];
#pragma unroll
                    for (int rt = 0; rt < 3; rt++)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b, acc[rt][ct], 0, 0, 0);
                }
        }
        if (more) {
            __syncthreads();
            stage();
            __syncthreads();
        }
    }
}

// acc += sT sT^T, both operands from the same 96 x 96 LDS tile: the diagonal tile's last rank-96 update, no staging.
// (Only the lower triangle is needed; dealing its 21 MFMA tiles out 6 / 6 / 6 / 3 to the waves, each with its own code
// path, was measured slower - 8.0 against 7.1 us - than the full product in the quadrant layout of the accumulator.)
__device__ __forceinline__ void dense_tile_lds_syrk(const double (*sT)[kPS], d4 acc[3][3]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = (wave >> 1) * 48, wc = (wave & 1) * 48;
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll 2
    for (int kk = 0; kk < kDNB; kk += 4) {
        double a[3], b[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            a[t] = sT[wr + 16 * t + fr][kk + fk];
            b[t] = sT[wc + 16 * t + fr][kk + fk];
        }
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
                acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b[ct], acc[rt][ct], 0, 0, 0);
    }
}

// Tile (I, J) from the accumulation to the published result, shared by the two flow kernels.  On return
//   J < I   L_IJ is in HBM (flag up) and in LDS (sT = flow_lds + 2 * kDNB * kDStride, row stride kPS)
//   J == I  Linv_J is in HBM (flag up) and in LDS (X = flow_lds + kDNB * kPS); the A block (flow_lds) is free
// `flags`: the tile flags, slot I (I + 1) / 2 + J; `bad`: the word a failed pivot stamps.
__device__ __forceinline__ void flow_factor_tile(const BaDev& d, int I, int J, unsigned epoch, unsigned* flags, unsigned* bad,
                                                 double* flow_lds, int* s_m_p, int* s_bad_p, FlowWatch& W) {
    const int tid = threadIdx.x, ld = d.ldS;
    const int self = I * (I + 1) / 2 + J;
    int& s_m = *s_m_p;
    int& s_bad = *s_bad_p;
    double (*sA)[kDStride] = reinterpret_cast<double (*)[kDStride]>(flow_lds);
    double (*sB)[kDStride] = reinterpret_cast<double (*)[kDStride]>(flow_lds + kDNB * kDStride);
    const int lane = tid & 63, wave = tid >> 6, wr = (wave >> 1) * 48, wc = (wave & 1) * 48;
    double* C = d.S + (size_t)I * kDNB * ld + (size_t)J * kDNB;
    // tot = sum_k L_Ik L_Jk^T - S_IJ (the matrix cores only add)
    d4 tot[3][3];
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int ct = 0; ct < 3; ct++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = wc + 16 * ct + (lane & 15);
                tot[rt][ct][reg] = -C[(size_t)r * ld + c];
            }
    SO_FLOW_MARK(0);
    const int lo = __builtin_amdgcn_readfirstlane(max(d.tile_first[I], d.tile_first[J]));
    const double* Pi = d.S + (size_t)I * kDNB * ld;
    const double* Pj = d.S + (size_t)J * kDNB * ld;
    // The diagonal workgroup owns the step that sits on the critical path: it keeps its own copy of the sub-diagonal
    // tile (J, J-1) - tot2, the same accumulation workgroup (J, J-1) does - so that when Linv_{J-1} arrives it forms
    // L_{J,J-1} in LDS and applies it to its tile without a round trip through HBM and a second flag.
    const bool own_sub = I == J && J > 0 && __builtin_amdgcn_readfirstlane(d.tile_first[J]) <= J - 1;
    const int k_end = own_sub ? J - 1 : J;
    const int sub_first = own_sub ? __builtin_amdgcn_readfirstlane(d.tile_first[J - 1]) : 0;  // tile (J-1, k) exists from here on
    const double* Ps = own_sub ? d.S + (size_t)(J - 1) * kDNB * ld : nullptr;
    d4 tot2[3][3];
    if (own_sub) {
        // S_{J,J-1} is read from its mirror image in the UPPER triangle (the Schur gather writes both, the factorisation
        // only ever overwrites lower tiles): workgroup (J, J-1) replaces the lower copy by L_{J,J-1} in place, possibly
        // before this workgroup gets to run - reading it there was a race that showed as one wrong solve in ~15 test runs
        const double* C2t = d.S + (size_t)(J - 1) * kDNB * ld + (size_t)J * kDNB;  // tile (J-1, J) = S_{J,J-1}^T
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = wc + 16 * ct + (lane & 15);
                    tot2[rt][ct][reg] = -C2t[(size_t)c * ld + r];
                }
    }
    for (int k = lo; k < k_end;) {
        if (tid == 0) {
            int m = 0;
            unsigned polls = 0;
            for (;;) {  // the ready prefix of the remaining column tiles, at least one
                while (k + m < k_end && flow_ready(&flags[I * (I + 1) / 2 + k + m], epoch) &&
                       flow_ready(&flags[J * (J + 1) / 2 + k + m], epoch) &&
                       (!own_sub || k + m < sub_first || flow_ready(&flags[(J - 1) * J / 2 + k + m], epoch)))
                    m++;
                if (m > 0) break;
                if (W.dead || ((++polls & 255u) == 0 && flow_expired(W))) {  // the solve is void: walk through what is left
                    m = k_end - k;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_m = m;
        }
        __syncthreads();
        const int m = __builtin_amdgcn_readfirstlane(s_m);  // (the GEMM's barriers separate this read from the next store)
        flow_acquire();
        dense_tile_nt<false>(Pi + (size_t)k * kDNB, ld, Pj + (size_t)k * kDNB, ld, sA, sB, tot, m * kDNB);
        if (own_sub && k + m > sub_first) {
            // (both products in one pass over K - 18 MFMA per step, the row operand staged once - was measured: 1 % on
            // the ticketed kernel, which then spills, and -1.5 % on the one-tile-per-workgroup kernel)
            const int k2 = max(k, sub_first);
            dense_tile_nt<false>(Pi + (size_t)k2 * kDNB, ld, Ps + (size_t)k2 * kDNB, ld, sA, sB, tot2, (k + m - k2) * kDNB);
        }
        k += m;
    }
    SO_FLOW_MARK(1);
    if (I == J) SO_FLOW_DIAG(4, J);
    if (I == J + 1) SO_FLOW_DIAG(7, J);
    if (I != J) {
        double (*sT)[kPS] = reinterpret_cast<double (*)[kPS]>(flow_lds + 2 * kDNB * kDStride);  // disjoint from sA / sB
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = wc + 16 * ct + (lane & 15);
                    sT[r][c] = -tot[rt][ct][reg];
                }
        if (tid == 0) flow_wait(&flags[J * (J + 1) / 2 + J], epoch, W);
        __syncthreads();
        flow_acquire();
        SO_FLOW_MARK(2);
        if (I == J + 1) SO_FLOW_DIAG(8, J);
        d4 acc[3][3];
        dense_tile_lds_nt(sT, d.dense_ws + (size_t)J * kDNB * kDNB, kDNB, sB, acc);
        __syncthreads();  // every wave is done reading sT: it now takes L_IJ
        SO_FLOW_MARK(3);
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = 16 * flow_trsm_ctile(wave, ct) + (lane & 15);
                    C[(size_t)r * ld + c] = acc[rt][ct][reg];
                    sT[r][c] = acc[rt][ct][reg];
                }
        flow_publish(&flags[self], epoch);
        SO_FLOW_MARK(4);
        if (I == J + 1) SO_FLOW_DIAG(9, J);
        return;
    }
    // diagonal tile
    if (own_sub) {
        double (*sT)[kPS] = reinterpret_cast<double (*)[kPS]>(flow_lds + 2 * kDNB * kDStride);
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = wc + 16 * ct + (lane & 15);
                    sT[r][c] = -tot2[rt][ct][reg];
                }
        if (tid == 0) flow_wait(&flags[(J - 1) * J / 2 + (J - 1)], epoch, W);
        __syncthreads();
        flow_acquire();
        SO_FLOW_MARK(2);
        SO_FLOW_DIAG(5, J);
        dense_tile_lds_nt(sT, d.dense_ws + (size_t)(J - 1) * kDNB * kDNB, kDNB, sB, tot2);  // L_{J,J-1}
        SO_FLOW_MARK(3);
        __syncthreads();  // every wave is done reading sT
#pragma unroll
        for (int rt = 0; rt < 3; rt++)
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = 16 * flow_trsm_ctile(wave, ct) + (lane & 15);
                    sT[r][c] = tot2[rt][ct][reg];
                }
        __syncthreads();
        dense_tile_lds_syrk(sT, tot);
    }
    SO_FLOW_MARK(5);
    double (*A)[kPS] = reinterpret_cast<double (*)[kPS]>(flow_lds);
    double (*X)[kPS] = reinterpret_cast<double (*)[kPS]>(flow_lds + kDNB * kPS);
    __syncthreads();  // the last GEMM's reads of sA / sB / sT (which A and X overlay) are over
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
        for (int ct = 0; ct < 3; ct++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r = wr + 16 * rt + (lane >> 4) + 4 * reg, c = wc + 16 * ct + (lane & 15);
                if ((c >> 4) <= (r >> 4)) A[r][c] = -tot[rt][ct][reg];
            }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    potrf_block_lds(A, X, &s_bad, d.dense_ws + (size_t)J * kDNB * kDNB);
    SO_FLOW_MARK(6);
    if (tid == 0 && s_bad) __hip_atomic_store(bad, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flow_publish(&flags[self], epoch);
    SO_FLOW_MARK(7);
}

// a 96 x 96 tile of S (row-major, leading dimension ld) into LDS (row stride kPS)
__device__ __forceinline__ void flow_stage_tile(const double* __restrict__ src, int ld, double (*dst)[kPS]) {
    for (int i = threadIdx.x; i < kDNB * (kDNB / 2); i += 256) {
        const int r = i / (kDNB / 2), c = 2 * (i - r * (kDNB / 2));
        const double2 v = *reinterpret_cast<const double2*>(src + (size_t)r * ld + c);
        dst[r][c] = v.x; dst[r][c + 1] = v.y;
    }
}

__global__ __launch_bounds__(256) void dense_flow_kernel(BaDev d, unsigned epoch) {
    extern __shared__ __align__(16) double flow_lds[];
    __shared__ int s_m, s_bad;
    __shared__ double s_v[kDNB], s_u[kDNB];
    if (d.lm->active != d.stage) return;
    const int tid = threadIdx.x, ld = d.ldS, T = ld / kDNB;
    const int2 t = d.flow_tiles[blockIdx.x];
    const int I = __builtin_amdgcn_readfirstlane(t.x), J = __builtin_amdgcn_readfirstlane(t.y), self = I * (I + 1) / 2 + J;
    unsigned* flags = d.flow_flags;
    FlowWatch W = flow_watch(d, &flags[kFlowFlagAbort], epoch);
    flow_factor_tile(d, I, J, epoch, flags + kFlowFlagTile, &flags[kFlowFlagBad], flow_lds, &s_m, &s_bad, W);
    double* vec_fwd = d.flow_vec;
    if (I != J) {
        const double (*sT)[kPS] = reinterpret_cast<const double (*)[kPS]>(flow_lds + 2 * kDNB * kDStride);
        // forward: L_IJ y_J
        if (tid == 0) flow_wait(&flags[kFlowFlagY + J], epoch, W);
        __syncthreads();
        if (tid < kDNB) s_v[tid] = flow_vec_load(&d.bs[(size_t)J * kDNB + tid]);
        __syncthreads();
        if (tid < kDNB) {
            double v = 0.0;
#pragma unroll 8
            for (int m = 0; m < kDNB; m++) v = fma(sT[tid][m], s_v[m], v);
            flow_vec_store(&vec_fwd[(size_t)self * kDNB + tid], v);
        }
        flow_publish_vec(&flags[kFlowFlagFwd + self], epoch);
        SO_FLOW_MARK(8);
        return;
    }
    double (*A)[kPS] = reinterpret_cast<double (*)[kPS]>(flow_lds);
    double (*X)[kPS] = reinterpret_cast<double (*)[kPS]>(flow_lds + kDNB * kPS);
    // forward substitution of this block row
    const int first = d.tile_first[J];
    if (tid == 0)
        for (int k = first; k < J; k++) flow_wait(&flags[kFlowFlagFwd + J * (J + 1) / 2 + k], epoch, W);
    __syncthreads();
    if (tid < kDNB) {
        double v = d.bs[(size_t)J * kDNB + tid];  // b_J: written before the launch
        for (int k = first; k < J; k++) v -= flow_vec_load(&vec_fwd[(size_t)(J * (J + 1) / 2 + k) * kDNB + tid]);
        s_v[tid] = v;
    }
    __syncthreads();
    if (tid < kDNB) {
        double y = 0.0;
        for (int m = 0; m <= tid; m++) y = fma(X[tid][m], s_v[m], y);
        flow_vec_store(&d.bs[(size_t)J * kDNB + tid], y);
        s_u[tid] = y;
    }
    flow_publish_vec(&flags[kFlowFlagY + J], epoch);
    SO_FLOW_MARK(8);
    // backward substitution: x_J = Linv_J^T (y_J - sum_{i>J} L_iJ^T x_i), rows from the bottom up.  x_{J+1} is the last to
    // arrive and the only one on the critical path: its tile L_{J+1,J} waits in LDS (the factor's A block is free now),
    // the rows further down are folded in from HBM as their x_i show up.
    double (*Lsub)[kPS] = A;
    const bool has_next = J + 1 < T && d.tile_first[J + 1] <= J;
    if (tid == 0)  // every tile of the column (all published long ago): one acquire covers the reads below
        for (int i = J + 1; i < T; i++)
            if (d.tile_first[i] <= J) flow_wait(&flags[kFlowFlagTile + i * (i + 1) / 2 + J], epoch, W);
    __syncthreads();
    flow_acquire();
    if (has_next) flow_stage_tile(d.S + (size_t)(J + 1) * kDNB * ld + (size_t)J * kDNB, ld, Lsub);
    double z = tid < kDNB ? s_u[tid] : 0.0;
    for (int i = T - 1; i > J; i--) {
        if (d.tile_first[i] > J) continue;
        if (tid == 0) flow_wait(&flags[kFlowFlagX + i], epoch, W);
        __syncthreads();  // (also: Lsub is complete, the previous round's reads of s_v are over)
        if (tid < kDNB) s_v[tid] = flow_vec_load(&d.bs[(size_t)i * kDNB + tid]);
        __syncthreads();
        if (tid < kDNB) {
            if (i == J + 1) {
#pragma unroll 8
                for (int m = 0; m < kDNB; m++) z = fma(-Lsub[m][tid], s_v[m], z);
            } else {
                const double* L = d.S + (size_t)i * kDNB * ld + (size_t)J * kDNB + tid;
                for (int m0 = 0; m0 < kDNB; m0 += 16) {  // 16 independent loads in flight per thread
                    double l[16];
#pragma unroll
                    for (int m = 0; m < 16; m++) l[m] = L[(size_t)(m0 + m) * ld];
#pragma unroll
                    for (int m = 0; m < 16; m++) z = fma(-l[m], s_v[m0 + m], z);
                }
            }
        }
    }
    __syncthreads();
    if (tid < kDNB) s_v[tid] = z;
    __syncthreads();
    if (tid < kDNB) {
        double x = 0.0;
        for (int m = tid; m < kDNB; m++) x = fma(X[m][tid], s_v[m], x);
        flow_vec_store(&d.bs[(size_t)J * kDNB + tid], x);
    }
    flow_publish_vec(&flags[kFlowFlagX + J], epoch);
    SO_FLOW_MARK(9);
    if (J == 0 && tid == 0) {  // the last block row to finish (every x_J is out before the verdict is written)
        for (int j = 1; j < T; j++) flow_wait(&flags[kFlowFlagX + j], epoch, W);
        d.partial[kBaSolveOk] = (flow_ready(&flags[kFlowFlagBad], epoch) || flow_ready(&flags[kFlowFlagAbort], epoch)) ? 0.0 : 1.0;
    }
}

// ---- the same dataflow for a skyline of any size: resident workgroups take the tiles by ticket ----
// Column by column, diagonal tile first - an order in which every tile's inputs carry smaller tickets, so a workgroup
// that waits always waits for a tile somebody resident is working on.  The grid is as large as the residency budget
// allows (one workgroup per CU); a tile costs its whole left-looking K-loop, so late columns keep every CU on the matrix
// cores while the early ones race ahead of the critical path (diagonal factor -> L_{J+1,J} -> update, 36 us per panel
// instead of the multi-launch chain's ~80).  Substitutions: the diagonal workgroup finishes y_J right after its factor
// (row tiles staged through LDS as the y_k arrive); when the tickets run out the workgroups take the block rows of the
// backward substitution from the bottom up (Linv_J and the column's tiles staged through LDS, x_i as they arrive).
// flags: [0, nslots) tiles | 512 y | 512 x | failed-pivot stamp | 2 ticket counters (zeroed by the host before the launch)
__global__ __launch_bounds__(256) void dense_flow_big_kernel(BaDev d, unsigned epoch, int n_tiles) {
    extern __shared__ __align__(16) double flow_lds[];
    __shared__ int s_m, s_bad;
    __shared__ unsigned s_ticket;
    __shared__ double s_v[kDNB];
    if (d.lm->active != d.stage) return;
    const int tid = threadIdx.x, ld = d.ldS, T = ld / kDNB;
    unsigned* flags = d.flow_flags;
    unsigned* fY = flags + d.flow_nslots;
    unsigned* fX = fY + kDenseMaxPanels;
    unsigned* bad = fX + kDenseMaxPanels;
    unsigned* counter = bad + 1;
    FlowWatch W = flow_watch(d, counter + 2, epoch);  // (the word behind the two ticket counters)
    double (*A)[kPS] = reinterpret_cast<double (*)[kPS]>(flow_lds);
    double (*X)[kPS] = reinterpret_cast<double (*)[kPS]>(flow_lds + kDNB * kPS);
    for (;;) {  // factorisation + forward substitution
        if (tid == 0) s_ticket = atomicAdd(&counter[0], 1u);
        __syncthreads();
        // (workgroup-uniform values go through readfirstlane: the branches around the barriers below are then scalar
        // branches - with per-lane copies the compiler predicates them, and the first version of this kernel hung)
        const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket);
        __syncthreads();  // (everybody has read the ticket before thread 0 takes the next one)
        if (t >= (unsigned)n_tiles) break;
        const int2 tile = d.flow_tiles[t];
        const int I = __builtin_amdgcn_readfirstlane(tile.x), J = __builtin_amdgcn_readfirstlane(tile.y);
        SO_FLOW_NOTE(10, 1000000 + I * 1000 + J);
        if (I == J) SO_FLOW_DIAG(3, J);
        if (I == J + 1) SO_FLOW_DIAG(6, J);
        flow_factor_tile(d, I, J, epoch, flags, bad, flow_lds, &s_m, &s_bad, W);
        SO_FLOW_NOTE(10, 2000000 + I * 1000 + J);
        if (I == J) SO_FLOW_DIAG(0, J);
        // forward substitution of block row J by its diagonal workgroup.  Every workgroup walks the same barriers (the
        // off-diagonal ones with an empty range and nothing to store): no barrier sits under a branch.
        const bool diag = I == J;
        const int first = diag ? __builtin_amdgcn_readfirstlane(d.tile_first[J]) : 0, last = diag ? J : 0;
        if (tid == 0)  // (J, J-1) was formed locally: its copy in HBM comes from workgroup (J, J-1)
            for (int k = first; k < last; k++) flow_wait(&flags[J * (J + 1) / 2 + k], epoch, W);
        __syncthreads();
        flow_acquire();
        double v = diag && tid < kDNB ? d.bs[(size_t)J * kDNB + tid] : 0.0;  // b_J: written before the launch
        for (int k = first; k < last; k++) {
            flow_stage_tile(d.S + (size_t)J * kDNB * ld + (size_t)k * kDNB, ld, A);
            SO_FLOW_NOTE(10, 3000000 + J * 1000 + k);
            if (tid == 0) flow_wait(&fY[k], epoch, W);
            __syncthreads();
            if (tid < kDNB) s_v[tid] = flow_vec_load(&d.bs[(size_t)k * kDNB + tid]);
            __syncthreads();
            if (tid < kDNB) {
#pragma unroll 8
                for (int m = 0; m < kDNB; m++) v = fma(-A[tid][m], s_v[m], v);
            }
            __syncthreads();  // A and s_v are free again
        }
        if (tid < kDNB) s_v[tid] = v;
        __syncthreads();
        if (diag && tid < kDNB) {
            double y = 0.0;
            for (int m = 0; m <= tid; m++) y = fma(X[tid][m], s_v[m], y);
            flow_vec_store(&d.bs[(size_t)J * kDNB + tid], y);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (diag && tid == 0) __hip_atomic_store(&fY[J], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (diag) SO_FLOW_DIAG(1, J);
    }
    for (;;) {  // backward substitution, block rows from the bottom up
        if (tid == 0) s_ticket = atomicAdd(&counter[1], 1u);
        __syncthreads();
        const unsigned u = __builtin_amdgcn_readfirstlane(s_ticket);
        __syncthreads();
        if (u >= (unsigned)T) break;
        const int J = __builtin_amdgcn_readfirstlane(d.flow_tiles[n_tiles + (int)u].x);  // deepest chains first (build_dense_plan)
        SO_FLOW_NOTE(10, 4000000 + J);
        if (tid == 0) {  // the factor of block J, its column below, y_J
            flow_wait(&flags[J * (J + 1) / 2 + J], epoch, W);
            for (int i = J + 1; i < T; i++)
                if (d.tile_first[i] <= J) flow_wait(&flags[i * (i + 1) / 2 + J], epoch, W);
            flow_wait(&fY[J], epoch, W);
        }
        __syncthreads();
        flow_acquire();
        flow_stage_tile(d.dense_ws + (size_t)J * kDNB * kDNB, kDNB, X);
        double z = tid < kDNB ? flow_vec_load(&d.bs[(size_t)J * kDNB + tid]) : 0.0;
        for (int i = T - 1; i > J; i--) {
            if (__builtin_amdgcn_readfirstlane(d.tile_first[i]) > J) continue;
            flow_stage_tile(d.S + (size_t)i * kDNB * ld + (size_t)J * kDNB, ld, A);
            SO_FLOW_NOTE(10, 5000000 + J * 1000 + i);
            if (tid == 0) flow_wait(&fX[i], epoch, W);
            __syncthreads();
            if (tid < kDNB) s_v[tid] = flow_vec_load(&d.bs[(size_t)i * kDNB + tid]);
            __syncthreads();
            if (tid < kDNB) {
#pragma unroll 8
                for (int m = 0; m < kDNB; m++) z = fma(-A[m][tid], s_v[m], z);
            }
            __syncthreads();
        }
        if (tid < kDNB) s_v[tid] = z;
        __syncthreads();  // (also: X is complete)
        if (tid < kDNB) {
            double x = 0.0;
            for (int m = tid; m < kDNB; m++) x = fma(X[m][tid], s_v[m], x);
            flow_vec_store(&d.bs[(size_t)J * kDNB + tid], x);
        }
        flow_publish_vec(&fX[J], epoch);
        SO_FLOW_DIAG(2, J);
        if (J == 0 && tid == 0) {  // every x_J is out before the verdict is written
            for (int j = 1; j < T; j++) flow_wait(&fX[j], epoch, W);
            d.partial[kBaSolveOk] = (flow_ready(bad, epoch) || flow_ready(counter + 2, epoch)) ? 0.0 : 1.0;
        }
        __syncthreads();
    }
    SO_FLOW_NOTE(10, 9000000);
}

#ifdef SO_FLOW_PROBE
extern "C" int so_debug_flow_diag(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flow_diag), sizeof(unsigned long long) * 10 * 512);
}
extern "C" int so_debug_flow_marks(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flow_marks), sizeof(unsigned long long) * 256 * 16);
}
#endif

void launch_ba_dense_pad(const BaDev& d, hipStream_t s) {
    const int n = 6 * d.n_free, np = d.ldS;
    if (np > n) hipLaunchKernelGGL(dense_pad_kernel, dim3(256), dim3(256), 0, s, d.S, d.bs, n, np);
}

// ---- launch plan ----
// The factorisation is right-looking in groups of G panels.  Per group: a serial chain over its panels ([update of the
// panel's block column with the group's earlier panels, K = 96 p] -> diagonal factor -> panel solve), then the
// trailing matrix from block k + G on takes the whole group in one K = 96 G update.  With look-ahead the trailing
// update is split into the G block columns the next group needs (A) and the rest (B): as soon as A is done the side
// stream runs the next group's chain while the main stream is still busy with B.
// build_dense_plan enumerates, in exactly that order, the tiles every update launch touches: tile (I, J) takes part in
// the update by panels [k, k + kw) iff it lies inside row I's envelope (J >= first[I]) and both rows have one of those
// panels inside theirs (max(first[I], first[J]) < k + kw).  A dense matrix (first == 0) reproduces the old launches.
namespace {

struct PlanBuilder {
    DensePlan* P;
    const int* first;
    void update(int k, int kw, int j0, int ncols, int rhs_panel) {  // ncols == 0: every tile I >= J >= j0
        DensePlan::Update u;
        u.k = k; u.kw = kw; u.rhs_panel = rhs_panel;
        u.first_tile = (int)P->tiles.size();
        const int T = P->T;
        const int jend = ncols > 0 ? std::min(T, j0 + ncols) : T;
        for (int J = j0; J < jend; J++)
            for (int I = J; I < T; I++) {
                if (J < first[I]) continue;
                const int lo = std::max(k, std::max(first[I], first[J]));
                if (lo >= k + kw) continue;
                P->tiles.push_back(make_int2(I, J));
                P->flop_structural += 2.0 * kDNB * kDNB * (double)((k + kw - lo) * kDNB);
            }
        u.n_tiles = (int)P->tiles.size() - u.first_tile;
        u.n_rhs_blocks = rhs_panel >= 0 ? ((T - (rhs_panel + 1)) * kDNB + 255) / 256 : 0;
        P->updates.push_back(u);
    }
    void chain(int k, int g) {
        const int T = P->T;
        for (int p = 0; p < g && k + p < T; p++) {
            const int c = k + p;
            if (p > 0) update(k, p, c, 1, c - 1);
            P->flop_structural += (double)kDNB * kDNB * kDNB / 3.0;  // diagonal factor
            for (int i = c + 1; i < T; i++)
                if (first[i] <= c) P->flop_structural += 2.0 * kDNB * kDNB * kDNB;  // panel solve as a GEMM with Linv
        }
    }
};

}  // namespace

void build_dense_plan(int T, const int* tile_first, bool has_side_stream, DensePlan* plan, int flow_grid) {
    static const bool no_lookahead = getenv("SWARMORB_DENSE_NO_LOOKAHEAD") != nullptr;
    static const int g_env = getenv("SWARMORB_DENSE_GROUP") ? atoi(getenv("SWARMORB_DENSE_GROUP")) : 0;
    plan->T = T;
    plan->lookahead = has_side_stream && T >= 32 && T <= kDenseMaxPanels && !no_lookahead;
    plan->G = g_env >= 1 && g_env <= 8 ? g_env : (plan->lookahead ? kDenseGroup : 1);
    plan->updates.clear();
    plan->tiles.clear();
    plan->flop_structural = 0.0;
    const double n = (double)T * kDNB;
    plan->flop_dense = n * n * n / 3.0 + 2.0 * n * n;
    plan->nnz_tiles = 0;
    for (int I = 0; I < T; I++) plan->nnz_tiles += I - tile_first[I] + 1;
    PlanBuilder B{plan, tile_first};
    const int G = plan->G;
    B.chain(0, G);
    for (int k = 0; k + G < T; k += G) {
        const int j0 = k + G, rem2 = T - j0;
        if (plan->lookahead && rem2 >= G + 2) {
            B.update(k, G, j0, G, j0 - 1);  // A: the next group's block columns (+ right-hand side)
            B.update(k, G, j0 + G, 0, -1);  // B: the rest
        } else {
            B.update(k, G, j0, 0, j0 - 1);
        }
        B.chain(j0, G);
    }
    plan->flop_structural += 2.0 * 2.0 * kDNB * kDNB * (double)plan->nnz_tiles;  // forward + backward substitution
    // single-launch dataflow solve: one workgroup per tile of the skyline, column by column (diagonal tile first)
    static const bool no_flow = getenv("SWARMORB_DENSE_NO_FLOW") != nullptr;
    static const int flow_max = getenv("SWARMORB_DENSE_FLOW_MAX_TILES") ? atoi(getenv("SWARMORB_DENSE_FLOW_MAX_TILES")) : kFlowDefaultMaxTiles;
    static const bool no_flow_big = getenv("SWARMORB_DENSE_NO_FLOW_BIG") != nullptr;
    plan->flow_first_tile = (int)plan->tiles.size();
    plan->flow_n_tiles = 0;
    plan->flow_big = !(T <= 21 && plan->nnz_tiles <= std::min(flow_max, kFlowMaxTiles));
    if (flow_grid > 0 && !no_flow && !(plan->flow_big && no_flow_big)) {
        // Ticket order.  level[J] = length of the chain of diagonal factors column J hangs on: 0 if its row has no tile to
        // the left, else 1 + the highest level inside the row's envelope.  A dense map gives level[J] = J (column by column);
        // a merged multi-agent map, numbered agent by agent and linked only where trajectories meet, gives every agent's
        // band its own chain: the columns of all agents at the same depth share a level, and the tickets walk the levels -
        // eight chains of 12 panels advance together instead of one chain of 94 (LinearSolverEigen gets the same effect
        // from its elimination tree, linear_solver_eigen.h:94-124).  Every input of a tile lies in a column of a lower
        // level or is the diagonal tile of its own column (first in the column), so a workgroup that waits always waits
        // for a tile a resident workgroup holds.  Exception by design: a tile starts from nothing when it is picked up and
        // has its whole left-looking history to catch up on, which is too late for the diagonal tiles (they carry the
        // chains and accumulate two tiles' worth): those get their tickets `lead` levels early and grow with the
        // factorisation.  Early tiles can wait for tickets not yet handed out - at most lead x (columns per level) of them,
        // which is kept under a third of the launch's workgroups (flow_grid: ba.cpp reserves them before it asks for the plan).
        // (Early tickets for the tiles next to the diagonal as well - a triangle of them - were measured: no better.)
        static const int lead_env = getenv("SWARMORB_DENSE_FLOW_DIAG_LEAD") ? atoi(getenv("SWARMORB_DENSE_FLOW_DIAG_LEAD")) : -1;
        std::vector<int> level((size_t)T, 0), per_level((size_t)T + 1, 0);
        int max_cols = 1;
        for (int J = 0; J < T; J++) {
            int mx = -1;
            for (int k = tile_first[J]; k < J; k++) mx = std::max(mx, level[(size_t)k]);
            level[(size_t)J] = mx + 1;
            max_cols = std::max(max_cols, ++per_level[(size_t)level[(size_t)J]]);
        }
        int lead = 0;
        if (plan->flow_big) {
            lead = lead_env >= 0 ? std::min(lead_env, kFlowDiagLeadMax) : kFlowDiagLeadDefault;
            lead = std::max(1, std::min(lead, (flow_grid / 3) / max_cols));
            if (lead_env == 0) lead = 0;
        }
        if (getenv("SWARMORB_BA_TRACE")) {
            int depth = 0;
            for (int J = 0; J < T; J++) depth = std::max(depth, level[(size_t)J]);
            fprintf(stderr, "[ba] dense plan: %d panels, %lld tiles, chain depth %d, at most %d columns per level, diagonal lead %d; first:", T,
                    plan->nnz_tiles, depth + 1, max_cols, lead);
            for (int J = 0; J < T; J++) fprintf(stderr, " %d", tile_first[J]);
            fprintf(stderr, "\n");
        }
        std::vector<std::pair<long long, int2>> order;
        for (int J = 0; J < T; J++)
            for (int I = J; I < T; I++)
                if (J >= tile_first[I]) {
                    const int lv = level[(size_t)J], key = I == J ? std::max(lv - lead, 0) : lv;
                    // ticket level, then (within a level) early diagonal tiles of deeper levels after the level's own
                    // tiles, then column, diagonal first
                    const long long early = (I == J && key != lv) ? 1 : 0;
                    order.push_back({((long long)key << 44) | (early << 43) | ((long long)J << 21) | (long long)(I - J), make_int2(I, J)});
                }
        std::sort(order.begin(), order.end(), [](const std::pair<long long, int2>& x, const std::pair<long long, int2>& y) { return x.first < y.first; });
        for (const auto& o : order) plan->tiles.push_back(o.second);
        plan->flow_n_tiles = (int)plan->tiles.size() - plan->flow_first_tile;
        // backward substitution of the ticketed kernel: block rows by the length of THEIR chain (x_J needs the x_i of the
        // tiles below it in column J), deepest dependencies first; T more entries (.x = block row) behind the tiles
        if (plan->flow_big) {
            std::vector<int> blevel((size_t)T, 0);
            for (int J = T - 1; J >= 0; J--) {
                int mx = -1;
                for (int i = J + 1; i < T; i++)
                    if (tile_first[i] <= J) mx = std::max(mx, blevel[(size_t)i]);
                blevel[(size_t)J] = mx + 1;
            }
            std::vector<std::pair<int, int>> back;
            for (int J = 0; J < T; J++) back.push_back({blevel[(size_t)J], -J});  // level, then bottom-up inside a level
            std::sort(back.begin(), back.end());
            for (const auto& bj : back) plan->tiles.push_back(make_int2(-bj.second, 0));
        }
    }
}

int dense_flow_max_tiles() { return kFlowMaxTiles; }

// Workgroups of the dataflow kernels the device can hold at the same time, from the runtime's occupancy calculator
// (block size, registers and the 152 KB of dynamic LDS against what a CU has) times the CU count - 0 when a workgroup
// does not fit at all.  ba.cpp never launches more than this (minus its reserve) in one dataflow launch.
int dense_flow_resident_capacity(int device) {
    constexpr int lds = (int)(sizeof(double) * kFlowLdsDoubles);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_flow_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_flow_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int per_cu_small = 0, per_cu_big = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_small, dense_flow_kernel, 256, (size_t)lds) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_big, dense_flow_big_kernel, 256, (size_t)lds) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    const int per_cu = per_cu_small < per_cu_big ? per_cu_small : per_cu_big;
    return per_cu > 0 ? per_cu * cus : 0;
}

namespace {

struct PlanCursor {
    const BaDev& d;
    size_t next = 0;
    void update(hipStream_t s) {
        const DensePlan::Update& u = d.plan->updates[next++];
        const int blocks = u.n_tiles + u.n_rhs_blocks;
        if (blocks > 0)
            hipLaunchKernelGGL(dense_update_kernel, dim3(blocks), dim3(256), 0, s, d, u.k, u.kw, d.plan_tiles + u.first_tile,
                               u.n_tiles, u.rhs_panel);
    }
    void chain(int k, int g, hipStream_t s) {
        const int T = d.plan->T;
        for (int p = 0; p < g && k + p < T; p++) {
            const int c = k + p, rem = T - c;
            if (p > 0) update(s);
            hipLaunchKernelGGL(dense_potrf_kernel, dim3(1), dim3(256), 0, s, d, c);
            hipLaunchKernelGGL(dense_panel_kernel, dim3(1 + (rem - 1)), dim3(256), 0, s, d, c);
        }
    }
};

}  // namespace

// Look-ahead only for T >= 32 panels: below that the cross-stream waits (~10 us each) cost more than the overlap
// returns (measured on GBA-1, 19 panels).
void launch_ba_dense_solve(const BaDev& d, hipStream_t s) {
    const DensePlan& P = *d.plan;
    if (d.flow_tiles) {  // ba.cpp hands the tile list over only when every workgroup of the launch can be resident
        static bool attr_set[64] = {false};
        int dev = 0;
        (void)hipGetDevice(&dev);
        constexpr int lds = (int)(sizeof(double) * kFlowLdsDoubles);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_flow_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_set[dev] = true;
        }
        if (!P.flow_big) {
            hipLaunchKernelGGL(dense_flow_kernel, dim3(P.flow_n_tiles), dim3(256), lds, s, d, ++*d.flow_epoch);
            return;
        }
        static bool attr_big[64] = {false};
        if (dev >= 0 && dev < 64 && !attr_big[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_flow_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_big[dev] = true;
        }
        unsigned* counters = d.flow_flags + d.flow_nslots + 2 * kDenseMaxPanels + 1;
        (void)hipMemsetAsync(counters, 0, 2 * sizeof(unsigned), s);
        hipLaunchKernelGGL(dense_flow_big_kernel, dim3(d.flow_grid), dim3(256), lds, s, d, ++*d.flow_epoch, P.flow_n_tiles);
        return;
    }
    const int T = P.T, G = P.G;
    hipStream_t side = d.dense_side;
    hipEvent_t* ev = d.dense_events;
    PlanCursor C{d};
    hipLaunchKernelGGL(dense_begin_kernel, dim3(1), dim3(1), 0, s, d);
    C.chain(0, G, s);
    int n_ev = 0;
    for (int k = 0; k + G < T; k += G) {
        const int j0 = k + G, rem2 = T - j0;  // block rows that take the group's update
        if (P.lookahead && rem2 >= G + 2) {
            C.update(s);
            (void)hipEventRecord(ev[n_ev], s);
            (void)hipStreamWaitEvent(side, ev[n_ev], 0);
            n_ev++;
            C.update(s);
            C.chain(j0, G, side);
            (void)hipEventRecord(ev[n_ev], side);
            (void)hipStreamWaitEvent(s, ev[n_ev], 0);
            n_ev++;
        } else {
            C.update(s);
            C.chain(j0, G, s);
        }
    }
    for (int k = T - 1; k >= 0; k--) {
        const int blocks = (k * kDNB + 255) / 256;
        hipLaunchKernelGGL(dense_backward_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, d, k);
    }
    hipLaunchKernelGGL(dense_finish_kernel, dim3((6 * d.n_free + 255) / 256), dim3(256), 0, s, d);
}


// Diagnostic load (so_runtime_occupy, include/swarmorb.h): `workgroups` workgroups that each pin `lds_bytes` of LDS - with
// 152 KB a whole CU - and do nothing but watch the wall clock for `ms` milliseconds.  What another process on the GPU
// looks like to the dataflow solves; tests/test_ba_gpu.py starts it under a solve to see the time-out path work.
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned occ_lds[];
    occ_lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (occ_lds[threadIdx.x] == 0xFFFFFFFFu) *sink = 1;  // (keeps the LDS allocation alive)
}

int launch_occupy(int workgroups, int lds_bytes, int ms, unsigned* d_sink, hipStream_t s) {
    if (workgroups <= 0 || lds_bytes < 256 || lds_bytes > 160 * 1024 || ms <= 0 || ms > 10000) return 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(64), (size_t)lds_bytes, s, (unsigned long long)ms * 100000ull, d_sink);
    return hipGetLastError() == hipSuccess ? 1 : 0;
}

}  // namespace so
