// ba_pcg.hip — block-Jacobi preconditioned conjugate gradients on the reduced camera system (the "PCG solve" BASELINE.json's
// north_star names), as an alternative to the direct block-skyline Cholesky of ba_dense.hip on large maps.
//
// The reference solves S x = b directly (LinearSolverEigen = SimplicialLDLT, code/Thirdparty/g2o/g2o/solvers/
// linear_solver_eigen.h:94-124); g2o also ships LinearSolverPCG (g2o/solvers/linear_solver_pcg.h: block-Jacobi
// preconditioner, relative residual tolerance, iteration cap) - this is that algorithm on the GPU, selected with
// so_ba_set_linear_solver.  Measured before it was built (tools/pcg_study.py, tools/probe/pcg_spmv_probe.hip,
// profiles/r5_pcg_*): on a 1500-keyframe map an iteration costs 16-45 us against 4.4-5.0 ms for one direct solve, and the
// damped systems of the first LM iterations converge in 10-40 iterations, those of the last ones in 60-300.
//
// Layout.  S stays where the Schur gather writes it - dense row-major, both triangles, leading dimension ldS - and is
// read through a 6 x 6 block-sparse index built once per problem from the gather's own pair lists (block (i1, i2) of S is
// nonzero iff a landmark is seen from both keyframes, block_solver.hpp:386-460): a map whose keyframes see each other
// along streets has 3 % of its blocks set (GBA-2r: 77 k of 2.26 M), and only those are touched.  One iteration = three
// launches, scalars stay on the device:
//   spmv       a wavefront per block row: lane l takes block (l / 6) of a group of ten, row (l % 6) of it - 48 contiguous
//              bytes per lane -, fixed summation order; + the partial sums of p.Sp.  The blocks are read from a COMPACT copy
//              (36 doubles per nonzero block, in index order, made once per solve by pcg_compact_kernel): in S itself the six rows
//              of a block are ldS doubles apart (72 KB on GBA-2r), so a wavefront's ten blocks were sixty separate 48-byte
//              pieces; compact they are 2.9 KB in a row (NOTES F.3 measured 110 -> 45 us per iteration on the probe)
//   update     alpha = rz / p.Sp; x += alpha p; r -= alpha Sp; z = M^-1 r (6 x 6 blocks); partial sums of r.z and r.r
//   direction  beta = rz' / rz; p = z + beta p; convergence: r.r <= tol^2 b.b; the status word goes to host-mapped memory
// The host keeps two chunks of iterations enqueued and looks at the status word between chunks (a converged or failed
// solve turns the launches still queued into no-ops).  All sums have a fixed order: the solve is deterministic.
#include <hip/hip_runtime.h>

#include <chrono>
#include <thread>

#include "ba_device.h"

namespace so {

namespace {

constexpr unsigned long long kStConverged = 1ull, kStFailed = 2ull;
__host__ __device__ inline unsigned long long pcg_status(unsigned seq, unsigned it, unsigned long long flags) {
    return ((unsigned long long)seq << 32) | ((unsigned long long)(it & 0x3FFFFFFFu) << 2) | flags;
}

__device__ __forceinline__ bool pcg_done(const BaPcgDev& q) { return (*q.status & (kStConverged | kStFailed)) != 0; }

__device__ __forceinline__ int pair_count(const BaDev& d, int a, int b) {  // block (a, b), a != b
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    const size_t g = (size_t)hi * ((size_t)hi + 1) / 2 + (size_t)lo;
    return d.pr_off[g + 1] - d.pr_off[g];
}

// ---- block structure of S from the pair lists (once per problem) ----
__global__ __launch_bounds__(256) void pcg_count_kernel(BaDev d, int* counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n_free) return;
    int c = 1;
    for (int j = 0; j < d.n_free; j++)
        if (j != i && pair_count(d, i, j) > 0) c++;
    counts[i] = c;
}

__global__ void pcg_scan_kernel(const int* counts, int* indptr, int n) {  // (n <= 8192: one thread, once per problem)
    int s = 0;
    for (int i = 0; i < n; i++) {
        indptr[i] = s;
        s += counts[i];
    }
    indptr[n] = s;
}

__global__ __launch_bounds__(256) void pcg_fill_kernel(BaDev d, const int* indptr, int* indices) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n_free) return;
    int at = indptr[i];
    for (int j = 0; j < d.n_free; j++)
        if (j == i || pair_count(d, i, j) > 0) indices[at++] = j;
}

// ---- start of a solve: M^-1 (inverse of the 6 x 6 diagonal blocks), x = 0, r = b, z = M^-1 r, p = z ----
__global__ __launch_bounds__(256) void pcg_init_kernel(BaDev d, BaPcgDev q) {
    __shared__ double sh[2][256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double rz = 0.0, bb = 0.0;
    if (i < d.n_free && d.lm->active == d.stage) {
        double A[6][6], L[6][6], Li[6][6];
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) A[r][c] = d.S[(size_t)(6 * i + r) * d.ldS + 6 * i + c];
        bool ok = true;
        for (int c = 0; c < 6; c++) {  // Cholesky of the (damped, positive definite) diagonal block
            for (int r = c; r < 6; r++) {
                double s = A[r][c];
                for (int k = 0; k < c; k++) s -= L[r][k] * L[c][k];
                if (r == c) {
                    ok = ok && s > 0.0;
                    L[c][c] = sqrt(s > 0.0 ? s : 1.0);
                } else {
                    L[r][c] = s / L[c][c];
                }
            }
        }
        for (int c = 0; c < 6; c++) {  // L^-1, column by column
            for (int r = 0; r < 6; r++) Li[r][c] = 0.0;
            Li[c][c] = 1.0 / L[c][c];
            for (int r = c + 1; r < 6; r++) {
                double s = 0.0;
                for (int k = c; k < r; k++) s -= L[r][k] * Li[k][c];
                Li[r][c] = s / L[r][r];
            }
        }
        double* M = q.Minv + 36 * (size_t)i;
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) {
                double s = 0.0;
                for (int k = (r > c ? r : c); k < 6; k++) s += Li[k][r] * Li[k][c];  // L^-T L^-1
                M[6 * r + c] = ok ? s : (r == c ? 1.0 : 0.0);
            }
        double rr[6];
        for (int k = 0; k < 6; k++) {
            rr[k] = d.bs[6 * (size_t)i + k];
            q.x[6 * (size_t)i + k] = 0.0;
            q.r[6 * (size_t)i + k] = rr[k];
            bb += rr[k] * rr[k];
        }
        for (int r = 0; r < 6; r++) {
            double v = 0.0;
            for (int c = 0; c < 6; c++) v += M[6 * r + c] * rr[c];
            q.z[6 * (size_t)i + r] = v;
            q.p[6 * (size_t)i + r] = v;
            rz += v * rr[r];
        }
    }
    sh[0][threadIdx.x] = rz;
    sh[1][threadIdx.x] = bb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        q.partB[blockIdx.x] = sh[0][0];
        q.partB[gridDim.x + blockIdx.x] = sh[1][0];
    }
}

__global__ __launch_bounds__(64) void pcg_init2_kernel(BaDev d, BaPcgDev q, int nB, unsigned seq, double tol) {
    double rz = 0.0, bb = 0.0;
    for (int i = 0; i < nB; i++) {  // (a handful of partials: fixed order)
        rz += q.partB[i];
        bb += q.partB[nB + i];
    }
    if (threadIdx.x != 0) return;
    q.scal[0] = rz;
    q.scal[1] = 0.0;
    q.scal[2] = bb;
    q.scal[3] = tol * tol * bb;
    unsigned long long flags = 0;
    if (d.lm->active != d.stage || !(bb > 0.0)) flags = kStConverged;  // nothing to do: the trial is not wanted, or b = 0 (x = 0)
    else if (!(rz > 0.0)) flags = kStFailed;
    const unsigned long long st = pcg_status(seq, 0, flags);
    *q.status = st;
    __hip_atomic_store(q.status_host, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the nonzero blocks of this trial's S side by side (same lane layout as the product that reads them) ----
__global__ __launch_bounds__(256) void pcg_compact_kernel(BaDev d, BaPcgDev q) {
    if (d.lm->active != d.stage) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const int sub = lane / 6, r = lane % 6;
    if (row >= d.n_free || lane >= 60) return;
    const int lo = q.indptr[row], hi = q.indptr[row + 1];
    const double* Srow = d.S + (size_t)(6 * row + r) * d.ldS;
    for (int k = lo + sub; k < hi; k += 10) {
        const double2* B = reinterpret_cast<const double2*>(Srow + 6 * (size_t)q.indices[k]);
        double2* C = reinterpret_cast<double2*>(q.Sc + 36 * (size_t)k + 6 * r);
        const double2 b0 = B[0], b1 = B[1], b2 = B[2];
        C[0] = b0; C[1] = b1; C[2] = b2;
    }
}

// ---- one iteration ----
__global__ __launch_bounds__(256) void pcg_spmv_kernel(BaDev d, BaPcgDev q) {
    if (pcg_done(q)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const int sub = lane / 6, r = lane % 6;
    double acc = 0.0;
    if (row < d.n_free && lane < 60) {
        const int lo = q.indptr[row], hi = q.indptr[row + 1];
        const double* Srow = d.S + (size_t)(6 * row + r) * d.ldS;
        for (int k = lo + sub; k < hi; k += 10) {
            const int j = q.indices[k];
            const double2* B = q.Sc ? reinterpret_cast<const double2*>(q.Sc + 36 * (size_t)k + 6 * r)
                                    : reinterpret_cast<const double2*>(Srow + 6 * (size_t)j);
            const double2* x = reinterpret_cast<const double2*>(q.p + 6 * (size_t)j);
            const double2 b0 = B[0], b1 = B[1], b2 = B[2], x0 = x[0], x1 = x[1], x2 = x[2];
            acc += ((b0.x * x0.x + b0.y * x0.y) + (b1.x * x1.x + b1.y * x1.y)) + (b2.x * x2.x + b2.y * x2.y);
        }
    }
    // the ten sub-sums of row r sit in lanes r, r + 6, ..., r + 54: added in that order
    double tot = 0.0;
    for (int s = 0; s < 10; s++) tot += __shfl(acc, r + 6 * s);
    double dot = 0.0;
    if (row < d.n_free && lane < 6) {
        q.Sp[6 * (size_t)row + lane] = tot;
        dot = tot * q.p[6 * (size_t)row + lane];
    }
    dot += __shfl_down(dot, 4);  // lanes 0..5 -> lane 0 (fixed tree)
    dot += __shfl_down(dot, 2);
    dot += __shfl_down(dot, 1);
    __shared__ double wsum[4];
    if (lane == 0) wsum[wave] = dot;
    __syncthreads();
    if (threadIdx.x == 0) q.partA[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// The same product with a WORKGROUP per block row (dense maps: GBA-2 has 516 blocks per row, and 1108 wavefronts walking 52
// steps each do not keep enough loads in flight): wave w takes the groups of ten w, w + 4, ...; the four waves' sums of a row
// are added as (w0 + w1) + (w2 + w3).  Fixed order, so deterministic - but not the order of the one-wave kernel: which of the
// two a problem gets depends on its structure only (BaPcgHost::wide), never on timing.
__global__ __launch_bounds__(256) void pcg_spmv_wide_kernel(BaDev d, BaPcgDev q) {
    if (pcg_done(q)) return;
    __shared__ double s_tot[4][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const int sub = lane / 6, r = lane % 6;
    double acc = 0.0;
    if (lane < 60) {
        const int lo = q.indptr[row], hi = q.indptr[row + 1];
        const double* Srow = d.S + (size_t)(6 * row + r) * d.ldS;
        for (int k = lo + sub + 10 * wave; k < hi; k += 40) {
            const int j = q.indices[k];
            const double2* B = q.Sc ? reinterpret_cast<const double2*>(q.Sc + 36 * (size_t)k + 6 * r)
                                    : reinterpret_cast<const double2*>(Srow + 6 * (size_t)j);
            const double2* x = reinterpret_cast<const double2*>(q.p + 6 * (size_t)j);
            const double2 b0 = B[0], b1 = B[1], b2 = B[2], x0 = x[0], x1 = x[1], x2 = x[2];
            acc += ((b0.x * x0.x + b0.y * x0.y) + (b1.x * x1.x + b1.y * x1.y)) + (b2.x * x2.x + b2.y * x2.y);
        }
    }
    double tot = 0.0;
    for (int s = 0; s < 10; s++) tot += __shfl(acc, r + 6 * s);
    if (lane < 6) s_tot[wave][lane] = tot;
    __syncthreads();
    if (wave != 0) return;
    double dot = 0.0;
    if (lane < 6) {
        const double t = (s_tot[0][lane] + s_tot[1][lane]) + (s_tot[2][lane] + s_tot[3][lane]);
        q.Sp[6 * (size_t)row + lane] = t;
        dot = t * q.p[6 * (size_t)row + lane];
    }
    dot += __shfl_down(dot, 4);  // lanes 0..5 -> lane 0 (fixed tree)
    dot += __shfl_down(dot, 2);
    dot += __shfl_down(dot, 1);
    if (lane == 0) q.partA[row] = dot;
}

__device__ __forceinline__ double block_sum_256(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double s = sh[0];
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(256) void pcg_update_kernel(BaDev d, BaPcgDev q, int nA, int par) {
    if (pcg_done(q)) return;
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nA; i += 256) s += q.partA[i];
    const double pSp = block_sum_256(s, sh);
    const double alpha = q.scal[par] / pSp;  // (p.Sp <= 0: the direction kernel reports the failure, nothing below is used)
    const int i = blockIdx.x * 256 + threadIdx.x;
    double rz = 0.0, rr2 = 0.0;
    if (i < d.n_free && pSp > 0.0) {
        double rr[6];
        for (int k = 0; k < 6; k++) {
            const size_t e = 6 * (size_t)i + k;
            q.x[e] += alpha * q.p[e];
            rr[k] = q.r[e] - alpha * q.Sp[e];
            q.r[e] = rr[k];
            rr2 += rr[k] * rr[k];
        }
        const double* M = q.Minv + 36 * (size_t)i;
        for (int r = 0; r < 6; r++) {
            double v = 0.0;
            for (int c = 0; c < 6; c++) v += M[6 * r + c] * rr[c];
            q.z[6 * (size_t)i + r] = v;
            rz += v * rr[r];
        }
    }
    const double a = block_sum_256(rz, sh), b = block_sum_256(rr2, sh);
    if (threadIdx.x == 0) {
        q.partB[blockIdx.x] = a;
        q.partB[gridDim.x + blockIdx.x] = b;
        if (blockIdx.x == 0) q.scal[4] = pSp;
    }
}

__global__ __launch_bounds__(256) void pcg_direction_kernel(BaDev d, BaPcgDev q, int nB, int par, unsigned seq, unsigned it_done, int last) {
    if (pcg_done(q)) return;
    double rz_new = 0.0, rr = 0.0;
    for (int i = 0; i < nB; i++) {  // (nB <= 32)
        rz_new += q.partB[i];
        rr += q.partB[nB + i];
    }
    const double pSp = q.scal[4];
    const double beta = rz_new / q.scal[par];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 6 * d.n_free && pSp > 0.0) q.p[i] = q.z[i] + beta * q.p[i];
    if (i == 0) {
        q.scal[par ^ 1] = rz_new;
        unsigned long long flags = 0;
        if (!(pSp > 0.0) || !(rz_new == rz_new)) flags = kStFailed;  // not positive definite / NaN
        else if (rr <= q.scal[3] || !(rz_new > 0.0)) flags = kStConverged;
        else if (last) flags = kStConverged;  // iteration cap: the iterate is used as it is (LinearSolverPCG does the same)
        const unsigned long long st = pcg_status(seq, it_done, flags);
        __hip_atomic_store(q.status_host, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (flags) *q.status = st;  // (read by the NEXT launches only)
        q.scal[5] = rr;
    }
}

__global__ __launch_bounds__(256) void pcg_finish_kernel(BaDev d, BaPcgDev q, int host_gave_up) {
    if (d.lm->active != d.stage) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 6 * d.n_free) d.bs[i] = q.x[i];
    // (host_gave_up: the host loop left without seeing a flag - x is whatever iteration it had reached: not a solve)
    if (i == 0) d.partial[kBaSolveOk] = ((*q.status & kStFailed) || host_gave_up) ? 0.0 : 1.0;
}

}  // namespace

void launch_ba_pcg_structure(const BaDev& d, int* counts, int* indptr, int* indices_or_null, hipStream_t s) {
    const int nb = (d.n_free + 255) / 256;
    if (!indices_or_null) {
        hipLaunchKernelGGL(pcg_count_kernel, dim3(nb), dim3(256), 0, s, d, counts);
        hipLaunchKernelGGL(pcg_scan_kernel, dim3(1), dim3(1), 0, s, counts, indptr, d.n_free);
    } else {
        hipLaunchKernelGGL(pcg_fill_kernel, dim3(nb), dim3(256), 0, s, d, indptr, indices_or_null);
    }
}

// One solve: S x = bs, x -> bs.  Enqueues on `s` and looks at the host-mapped status word between chunks of iterations.
void launch_ba_pcg_solve(const BaDev& d, hipStream_t s) {
    BaPcgHost& H = *d.pcg_host;
    const BaPcgDev& q = H.dev;
    const int nf = d.n_free, nA = H.wide ? nf : (nf + 3) / 4, nB = (nf + 255) / 256, nC = (6 * nf + 255) / 256;
    const unsigned seq = ++H.seq;
    if (q.Sc) hipLaunchKernelGGL(pcg_compact_kernel, dim3((nf + 3) / 4), dim3(256), 0, s, d, q);
    hipLaunchKernelGGL(pcg_init_kernel, dim3(nB), dim3(256), 0, s, d, q);
    hipLaunchKernelGGL(pcg_init2_kernel, dim3(1), dim3(64), 0, s, d, q, nB, seq, H.tol);
    constexpr int kChunk = 8;
    const int max_it = H.max_it > 0 ? H.max_it : 1;
    int enq = 0;
    auto chunk = [&]() {
        for (int c = 0; c < kChunk && enq < max_it; c++, enq++) {
            const int par = enq & 1;
            if (H.wide) hipLaunchKernelGGL(pcg_spmv_wide_kernel, dim3(nA), dim3(256), 0, s, d, q);
            else hipLaunchKernelGGL(pcg_spmv_kernel, dim3(nA), dim3(256), 0, s, d, q);
            hipLaunchKernelGGL(pcg_update_kernel, dim3(nB), dim3(256), 0, s, d, q, nA, par);
            hipLaunchKernelGGL(pcg_direction_kernel, dim3(nC), dim3(256), 0, s, d, q, nB, par, seq, (unsigned)(enq + 1), enq + 1 == max_it ? 1 : 0);
        }
    };
    chunk();
    chunk();
    volatile unsigned long long* st = H.status_host;
    int waited_for = kChunk;  // iterations the host has seen complete (or the solve end)
    unsigned it_seen = 0;
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0, gave_up = 0;
    for (;;) {
        const unsigned long long v = *st;
        if ((unsigned)(v >> 32) == seq) {
            it_seen = (unsigned)((v >> 2) & 0x3FFFFFFFu);
            if (v & (kStConverged | kStFailed)) break;
            if ((int)it_seen >= std::min(waited_for, max_it)) {
                if (enq >= max_it && (int)it_seen >= max_it) break;  // (cannot happen: the last iteration sets a flag)
                waited_for += kChunk;
                chunk();
                continue;
            }
        }
        if (++spins > 2000) {  // not a busy spin for ever: nap, and give up on a dead stream
            std::this_thread::sleep_for(std::chrono::microseconds(5));
            const hipError_t qs = hipStreamQuery(s);
            if (qs == hipSuccess) {
                // the stream has drained without this solve's stamp: either the stamp landed after the read above, or
                // the launches belonged to a stage that is over (every kernel returned at once: nothing to solve)
                const unsigned long long v2 = *st;
                if ((unsigned)(v2 >> 32) == seq && !(v2 & (kStConverged | kStFailed))) gave_up = 2;  // begun, never finished
                if ((unsigned)(v2 >> 32) == seq) it_seen = (unsigned)((v2 >> 2) & 0x3FFFFFFFu);
                break;
            }
            if (qs != hipErrorNotReady) {  // the stream is in error
                gave_up = 2;
                break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
                gave_up = 1;
                break;
            }
        }
    }
    if (gave_up && !H.fault) H.fault = gave_up;  // surfaced by so_bundle_adjust as SO_ERR_TIMEOUT / SO_ERR_HIP
    H.iterations += (long long)it_seen;
    H.solves++;
    hipLaunchKernelGGL(pcg_finish_kernel, dim3(nC), dim3(256), 0, s, d, q, gave_up);
}

}  // namespace so
