// ba_kernels.hip — gfx950 kernels of the bundle-adjustment hot path (FP64).
//
// One Levenberg-Marquardt trial = schur (per-landmark 3x3 inverse + per-block gather of the reduced camera
// system) -> dense Cholesky solve of the reduced system -> back-substitution + manifold update into the trial
// buffers -> residuals + chi2 of the trial state -> accept / reject (ba_trial_decide_kernel).  The LM state
// (lambda, which buffer is current, iteration counters) lives in HBM (BaLm); every kernel reads it and returns
// at once when the optimize() call has finished, so the host enqueues a whole stage without a round trip.
// One iteration additionally linearises (build).
// Reductions are segmented by vertex through host-built CSR lists (edges pre-sorted by landmark): no
// floating-point atomics, fixed summation order, deterministic results.
//
// Reference behaviour followed (files under /root/reference/code/Thirdparty/g2o/g2o):
//   types/types_six_dof_expmap.{h,cpp}  EdgeSE3ProjectXYZ error + analytic Jacobians, VertexSE3Expmap oplus
//   types/se3quat.h                     SE3Quat exp / product / map
//   core/base_binary_edge.hpp:55-120    constructQuadraticForm (+ Huber weights, robust_kernel_impl.cpp:78-91)
//   core/block_solver.hpp:354-486       Schur complement, back-substitution
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include <hipcub/hipcub.hpp>

#include "ba_device.h"

// FP64 solver: parity with the oracle is tolerance-based (see tests/test_ba_gpu.py), so let the compiler fuse
// multiply-adds here (the ORB translation unit stays contraction-free for bit parity).
#pragma clang fp contract(fast)

namespace so {

#include "ba_solve_mfma.inc"  // ba_solve_mfma_body<256> for the resident trial loop (the solve's own kernels: ba_dense.hip)

// ---------------- SE3Quat pieces (se3quat.h), same formulas as the CPU oracle ----------------
__device__ __forceinline__ void quat_rotate(const double* q, const double* v, double* out) {
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    out[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    out[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    out[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

__device__ __forceinline__ void quat_from_R(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        // Eigen's i / j = (i + 1) % 3 / k = (j + 1) % 3 walk, written out per case: with run-time indices R and q live in
        // scratch memory (19 scratch_* instructions in every register-resident PoseOptimization kernel, the stores ahead
        // of the branch whether it is taken or not); the arithmetic per case is the same, operand for operand
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i == 1 ? R[4] : R[0])) i = 2;
        if (i == 0) {         // j = 1, k = 2
            t = sqrt(R[0] - R[4] - R[8] + 1.0);
            q[0] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (R[7] - R[5]) * t;
            q[1] = (R[3] + R[1]) * t;
            q[2] = (R[6] + R[2]) * t;
        } else if (i == 1) {  // j = 2, k = 0
            t = sqrt(R[4] - R[8] - R[0] + 1.0);
            q[1] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (R[2] - R[6]) * t;
            q[2] = (R[7] + R[5]) * t;
            q[0] = (R[1] + R[3]) * t;
        } else {              // j = 0, k = 1
            t = sqrt(R[8] - R[0] - R[4] + 1.0);
            q[2] = 0.5 * t;
            t = 0.5 / t;
            q[3] = (R[3] - R[1]) * t;
            q[0] = (R[2] + R[6]) * t;
            q[1] = (R[5] + R[7]) * t;
        }
    }
}

__device__ __forceinline__ void quat_normalize_rotation(double* q) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// pose <- SE3Quat::exp(u) * pose   (VertexSE3Expmap::oplusImpl)
__device__ void se3_exp_mul(const double* u, const BaPose& in, BaPose& out) {
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double Om[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double Om2[9], R[9], V[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Om2[i * 3 + j] = Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j] + Om[i * 3 + 2] * Om[6 + j];
    if (theta < 0.00001) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
            V[i] = R[i];
        }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / (theta * theta * theta);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
        }
    }
    double qa[4], ta[3], rt[3];
    quat_from_R(R, qa);
    quat_normalize_rotation(qa);
#pragma unroll
    for (int i = 0; i < 3; i++) ta[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    quat_rotate(qa, in.t, rt);
    out.t[0] = ta[0] + rt[0]; out.t[1] = ta[1] + rt[1]; out.t[2] = ta[2] + rt[2];
    const double* b = in.q;
    double qn[4];
    qn[3] = qa[3] * b[3] - qa[0] * b[0] - qa[1] * b[1] - qa[2] * b[2];
    qn[0] = qa[3] * b[0] + qa[0] * b[3] + qa[1] * b[2] - qa[2] * b[1];
    qn[1] = qa[3] * b[1] + qa[1] * b[3] + qa[2] * b[0] - qa[0] * b[2];
    qn[2] = qa[3] * b[2] + qa[2] * b[3] + qa[0] * b[1] - qa[1] * b[0];
    quat_normalize_rotation(qn);
    out.q[0] = qn[0]; out.q[1] = qn[1]; out.q[2] = qn[2]; out.q[3] = qn[3];
    out.pad = 0;
}

struct EdgeLin {
    double err0, err1;
    double Jp[6];   // 2x3 d err / d point
    double Jc[12];  // 2x6 d err / d pose [omega, upsilon]
    double z;
};

__device__ __forceinline__ void camera_point(const BaPose& P, const double* X, double* pc) {
    quat_rotate(P.q, X, pc);
    pc[0] += P.t[0]; pc[1] += P.t[1]; pc[2] += P.t[2];
}

__device__ __forceinline__ void edge_error(const BaPose& P, const double* X, const double* obs, const double* K,
                                           double& e0, double& e1) {
    double pc[3];
    camera_point(P, X, pc);
    e0 = obs[0] - (pc[0] / pc[2] * K[0] + K[2]);
    e1 = obs[1] - (pc[1] / pc[2] * K[1] + K[3]);
}

__device__ __forceinline__ void edge_jacobians(const BaPose& P, const double* X, const double* K, double* Jp, double* Jc) {
    double pc[3], R[9];
    camera_point(P, X, pc);
    quat_to_R(P.q, R);
    const double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z, fx = K[0], fy = K[1];
    const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            Jp[r * 3 + c] = -1. / z * (tmp[r * 3] * R[c] + tmp[r * 3 + 1] * R[3 + c] + tmp[r * 3 + 2] * R[6 + c]);
    Jc[0] = x * y / z_2 * fx;       Jc[1] = -(1 + (x * x / z_2)) * fx; Jc[2] = y / z * fx;
    Jc[3] = -1. / z * fx;           Jc[4] = 0;                         Jc[5] = x / z_2 * fx;
    Jc[6] = (1 + y * y / z_2) * fy; Jc[7] = -x * y / z_2 * fy;         Jc[8] = -x / z * fy;
    Jc[9] = 0;                      Jc[10] = -1. / z * fy;             Jc[11] = y / z_2 * fy;
}

__device__ __forceinline__ double huber_rho0(double e, double delta, float dsqr) {
    return (e <= dsqr) ? e : 2 * sqrt(e) * delta - dsqr;
}
__device__ __forceinline__ double huber_rho1(double e, double delta, float dsqr) {
    return (e <= dsqr) ? 1.0 : delta / sqrt(e);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// the two widest exchanges (lane ^ 32, lane ^ 16) use gfx950's v_permlane32_swap / v_permlane16_swap: the first
// operand's upper half (odd 16-lane rows) trades places with the second operand's lower half (even rows), so
// a' + b' is "my half of the values plus my partner's copy of the same half" without touching LDS
__device__ __forceinline__ double po_swap_add32(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double po_swap_add16(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

template <int COUNT>
__device__ __forceinline__ void po_halve(double* v, int off, int lane) {
    const bool upper = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < COUNT; i++) {
        const double keep = upper ? v[i + COUNT] : v[i];
        const double send = upper ? v[i] : v[i + COUNT];
        v[i] = keep + __shfl_xor(send, off);
    }
}

// block-wide deterministic sum (fixed tree: xor-butterfly inside a wave, then waves in index order)
__device__ __forceinline__ double block_sum(double v, double* s_tmp /* >= 16 doubles */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) s_tmp[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; i++) t += s_tmp[i];
    return t;
}

// ---------------- residuals + chi2 (computeActiveErrors + activeRobustChi2) ----------------
__device__ __forceinline__ void ba_errors_body(const BaDev& d, int which, int gated, const unsigned BX, const unsigned GX) {
    __shared__ double s_tmp[16];
    const BaLm lm = *d.lm;
    if (gated == 1 && lm.active != d.stage) return;
    if (gated == 2 && (lm.active || lm.stages_begun != d.stage - 1)) return;  // chained behind the previous stage's trials: only once that stage is over
    const BaPose* __restrict__ poses = d.pose[lm.cur ^ which];
    const double* __restrict__ points = d.pt[lm.cur ^ which];
    double acc = 0.0;
    for (int e = BX * 256 + threadIdx.x; e < d.n_edges; e += GX * 256) {
        if (!d.e_active[e]) continue;
        const int ip = d.e_pose[e], il = d.e_point[e];
        double e0, e1;
        const double obs[2] = {d.e_obs[2 * e], d.e_obs[2 * e + 1]};
        const double X[3] = {points[3 * il], points[3 * il + 1], points[3 * il + 2]};
        edge_error(poses[ip], X, obs, d.intr + 4 * ip, e0, e1);
        const double w = d.e_w[e];
        const double chi2 = e0 * (w * e0) + e1 * (w * e1);
        d.e_err[2 * e] = e0;
        d.e_err[2 * e + 1] = e1;
        d.e_chi2[e] = chi2;
        acc += d.robust ? huber_rho0(chi2, d.huber_delta, d.huber_dsqr) : chi2;
    }
    const double t = block_sum(acc, s_tmp);
    if (threadIdx.x == 0) d.partial[kBaPartialChi + BX] = t;
}
__global__ __launch_bounds__(256) void ba_errors_kernel(BaDev d, int which, int gated) {
    ba_errors_body(d, which, gated, blockIdx.x, gridDim.x);
}


thread_local BaRecorder* g_ba_recorder = nullptr;

namespace {
BaLaunchRec& ba_record(int kind, const BaDev& d, int grid) {
    g_ba_recorder->list.emplace_back();
    BaLaunchRec& r = g_ba_recorder->list.back();
    r.kind = kind;
    r.d = d;
    r.grid = grid;
    r.phase = g_ba_recorder->phase;
    return r;
}
}  // namespace

void launch_ba_errors(const BaDev& d, int which, int gate, int n_blocks, hipStream_t s) {
    if (g_ba_recorder) {
        BaLaunchRec& r = ba_record(kBaKErrors, d, n_blocks);
        r.i0 = which;
        r.i1 = gate;
        return;
    }
    hipLaunchKernelGGL(ba_errors_kernel, dim3(n_blocks), dim3(256), 0, s, d, which, gate);
}

// edge_tab[hessian index][landmark] = the edge joining them (at most one: a keyframe observes a landmark once).
// One row per keyframe: the blocks (., i2) of the gather, dispatched back to back, all read row i2, which stays in L2.
__global__ __launch_bounds__(256) void ba_edge_table_kernel(BaDev d) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= d.n_edges) return;
    const int h = d.pose_hidx[d.e_pose[e]];
    if (h >= 0) d.edge_tab[(size_t)h * d.n_points + d.e_point[e]] = e;
}

// The per-call fills (stored errors, chi2, partials, LM state to zero; edge table to -1) as ONE launch: five runtime
// fills in a row cost ~5 us each on the stream before the first kernel of the solve can start.
__global__ __launch_bounds__(256) void ba_clear_kernel(BaClearList L) {
    for (int k = 0; k < L.n; k++) {
        const BaClearItem it = L.item[k];
        uint32_t* p = static_cast<uint32_t*>(it.p);
        const size_t n4 = it.bytes / 16, words = it.bytes / 4;
        const uint4 v = make_uint4(it.value, it.value, it.value, it.value);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) reinterpret_cast<uint4*>(p)[i] = v;
        for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = it.value;
    }
}

void launch_ba_clear(const BaClearList& L, hipStream_t s) {
    size_t most = 0;
    for (int k = 0; k < L.n; k++) most = L.item[k].bytes > most ? L.item[k].bytes : most;
    const int blocks = (int)std::min<size_t>(1024, std::max<size_t>(1, most / (16 * 256 * 4)));
    hipLaunchKernelGGL(ba_clear_kernel, dim3(blocks), dim3(256), 0, s, L);
}

void launch_ba_edge_table(const BaDev& d, hipStream_t s) {
    if (d.n_edges <= 0 || d.n_free <= 0) return;
    hipLaunchKernelGGL(ba_edge_table_kernel, dim3((d.n_edges + 255) / 256), dim3(256), 0, s, d);
}

// ---------------- pair lists for large maps (global bundle adjustment) ----------------
// A keyframe pair of a big map shares a handful of landmarks or none; walking a keyframe's ~500 edges per block to
// find them (the local-window gather) is 200x more lookups than there are pairs.  Per landmark, every pair of its
// observing free keyframes (h1 < h2) is one entry of block g = h2 (h2 + 1) / 2 + h1.  Count -> scan -> fill; the
// order inside a block comes out of atomics and is therefore arbitrary: the consumer sorts each short list by
// landmark before it sums, so results stay deterministic.
template <bool FILL>
__global__ __launch_bounds__(256) void ba_pairs_kernel(BaDev d) {
    const int il = blockIdx.x * 256 + threadIdx.x;
    if (il >= d.n_points) return;
    const int e0 = d.pt_off[il], e1 = d.pt_off[il + 1];
    for (int a = e0; a < e1; a++) {
        const int ha = d.pose_hidx[d.e_pose[a]];
        if (ha < 0) continue;
        for (int c = a + 1; c < e1; c++) {
            const int hc = d.pose_hidx[d.e_pose[c]];
            if (hc < 0) continue;
            const int h1 = ha < hc ? ha : hc, h2 = ha < hc ? hc : ha;
            const int g = h2 * (h2 + 1) / 2 + h1;
            if (!FILL) {
                atomicAdd(&d.pr_cur[g], 1);
            } else {
                const int pos = d.pr_off[g] + atomicAdd(&d.pr_cur[g], 1);
                d.pr_l[pos] = il;
                d.pr_k1[pos] = ha < hc ? a : c;  // edge of the keyframe with the smaller hessian index
                d.pr_k2[pos] = ha < hc ? c : a;
            }
        }
    }
}

__global__ __launch_bounds__(256) void ba_pairs_classify_kernel(BaDev d, int n_blk) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_blk) return;
    d.pr_cur[g] = 0;  // becomes the fill cursor
    int i2 = (int)((sqrt(8.0 * (double)g + 1.0) - 1.0) * 0.5);
    while ((i2 + 1) * (i2 + 2) / 2 <= g) i2++;
    while (i2 * (i2 + 1) / 2 > g) i2--;
    const int i1 = g - i2 * (i2 + 1) / 2;
    if (i1 == i2 || d.pr_off[g + 1] - d.pr_off[g] > kBaSmallBlockPairs) d.big_list[atomicAdd(d.big_n, 1)] = g;
}

// Once per problem: every block's list sorted by landmark (out of place, rank by counting: the landmark of a pair is
// unique inside a block), one wave per block.  After this the per-trial kernels read the lists in a fixed order.
constexpr int kPairSortMax = 4096;  // pairs of one block held in LDS while it is sorted

__global__ __launch_bounds__(64) void ba_pairs_sort_kernel(BaDev d, int n_blk, int* __restrict__ out_l,
                                                           int* __restrict__ out_k1, int* __restrict__ out_k2) {
    __shared__ int s_l[kPairSortMax];
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= n_blk) return;
    const int o = d.pr_off[g], np = d.pr_off[g + 1] - o;
    if (np <= 0) return;
    if (np > kPairSortMax) {  // cannot happen below ~4096 shared landmarks; keep the arrival order (still a valid list)
        for (int i = lane; i < np; i += 64) { out_l[o + i] = d.pr_l[o + i]; out_k1[o + i] = d.pr_k1[o + i]; out_k2[o + i] = d.pr_k2[o + i]; }
        return;
    }
    for (int i = lane; i < np; i += 64) s_l[i] = d.pr_l[o + i];
    __syncthreads();
    for (int i = lane; i < np; i += 64) {
        const int li = s_l[i];
        int rank = 0;
        for (int j = 0; j < np; j++) rank += s_l[j] < li;
        out_l[o + rank] = li;
        out_k1[o + rank] = d.pr_k1[o + i];
        out_k2[o + rank] = d.pr_k2[o + i];
    }
}

size_t ba_pairs_scan_temp_bytes(int n_blk) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int*)nullptr, (int*)nullptr, n_blk + 1);
    return bytes;
}

void launch_ba_build_pairs(const BaDev& d, void* scan_temp, size_t scan_temp_bytes, hipStream_t s) {
    const int nf = d.n_free, n_blk = nf * (nf + 1) / 2;
    if (n_blk <= 0 || d.n_points <= 0) return;
    (void)hipMemsetAsync(d.pr_cur, 0, sizeof(int) * ((size_t)n_blk + 1), s);
    (void)hipMemsetAsync(d.big_n, 0, sizeof(int), s);
    const dim3 grid((d.n_points + 255) / 256);
    hipLaunchKernelGGL(ba_pairs_kernel<false>, grid, dim3(256), 0, s, d);
    (void)hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_temp_bytes, d.pr_cur, d.pr_off, n_blk + 1, s);
    hipLaunchKernelGGL(ba_pairs_classify_kernel, dim3((n_blk + 255) / 256), dim3(256), 0, s, d, n_blk);
    hipLaunchKernelGGL(ba_pairs_kernel<true>, grid, dim3(256), 0, s, d);
    hipLaunchKernelGGL(ba_pairs_sort_kernel, dim3(n_blk), dim3(64), 0, s, d, n_blk, d.ps_l, d.ps_k1, d.ps_k2);
}

__device__ __forceinline__ void ba_finish_body(const BaDev& d, double chi2_threshold, BaPose* __restrict__ pose_out,
                                                         double* __restrict__ pt_out, double* __restrict__ chi2_out,
                                                         uint8_t* __restrict__ outlier_out, const unsigned BX, const unsigned GX) {
    const int i = BX * 256 + threadIdx.x;
    const int cur = d.lm->cur;
    const BaPose* poses = d.pose[cur];
    const double* points = d.pt[cur];
    if (i < d.n_edges) {
        const int il = d.e_point[i];
        const double X[3] = {points[3 * il], points[3 * il + 1], points[3 * il + 2]};
        double pc[3];
        camera_point(poses[d.e_pose[i]], X, pc);  // isDepthPositive()
        const double chi2 = d.e_chi2[i];
        chi2_out[i] = chi2;
        outlier_out[i] = (chi2 > chi2_threshold || !(pc[2] > 0.0)) ? 1 : 0;
    }
    if (i < d.n_poses) pose_out[i] = poses[i];
    if (i < 3 * d.n_points) pt_out[i] = points[i];
}
__global__ __launch_bounds__(256) void ba_finish_kernel(BaDev d, double chi2_threshold, BaPose* __restrict__ pose_out,
                                                         double* __restrict__ pt_out, double* __restrict__ chi2_out,
                                                         uint8_t* __restrict__ outlier_out) {
    ba_finish_body(d, chi2_threshold, pose_out, pt_out, chi2_out, outlier_out, blockIdx.x, gridDim.x);
}


// Completion word behind an epilogue: everything enqueued before this launch has run and its writes into host-mapped memory
// are out, so the host can spin on the word instead of querying the stream.
__global__ void ba_signal_kernel(int* __restrict__ word, int value) {
    __threadfence_system();
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void launch_ba_signal(int* word_host_mapped, int value, hipStream_t s) {
    if (g_ba_recorder) {
        BaLaunchRec& r = ba_record(kBaKSignal, BaDev{}, 1);
        r.p0 = word_host_mapped;
        r.i0 = value;
        return;
    }
    hipLaunchKernelGGL(ba_signal_kernel, dim3(1), dim3(1), 0, s, word_host_mapped, value);
}

void launch_ba_finish(const BaDev& d, double chi2_threshold, BaPose* pose_out, double* pt_out, double* chi2_out,
                      uint8_t* outlier_out, hipStream_t s) {
    int nthreads = d.n_edges > d.n_poses ? d.n_edges : d.n_poses;
    if (3 * d.n_points > nthreads) nthreads = 3 * d.n_points;
    if (nthreads <= 0) return;
    if (g_ba_recorder) {
        BaLaunchRec& r = ba_record(kBaKFinish, d, (nthreads + 255) / 256);
        r.f0 = chi2_threshold;
        r.p0 = pose_out; r.p1 = pt_out; r.p2 = chi2_out; r.p3 = outlier_out;
        return;
    }
    hipLaunchKernelGGL(ba_finish_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, s, d, chi2_threshold, pose_out,
                       pt_out, chi2_out, outlier_out);
}

// Between the two stages of LocalBundleAdjustment (Optimizer.cc:644-656): one thread per landmark walks its
// (contiguous) edges; an active edge whose stored chi2 exceeds the threshold or whose point is not in front of
// the camera in the current estimate is dropped (setLevel(1)); a landmark without active edges drops out too.
// (eight lanes per landmark, each looking at every eighth edge: a thread per landmark walked ~15 edges of dependent loads,
//  32 us on a 26 k-edge window; flags only, so no summation order is involved)
__device__ __forceinline__ void ba_mark_outliers_body(const BaDev& d, double chi2_threshold, int gate, const unsigned BX, const unsigned GX) {
    const int il = BX * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    if (gate == 2 && (d.lm->active || d.lm->stages_begun != d.stage - 1)) return;
    const bool live = il < d.n_points;
    const int cur = d.lm->cur;
    int alive = 0;
    if (live) {
        const double* points = d.pt[cur];
        const double X[3] = {points[3 * il], points[3 * il + 1], points[3 * il + 2]};
        for (int e = d.pt_off[il] + sub; e < d.pt_off[il + 1]; e += 8) {
            if (!d.e_active[e]) continue;
            double pc[3];
            camera_point(d.pose[cur][d.e_pose[e]], X, pc);
            if (d.e_chi2[e] > chi2_threshold || !(pc[2] > 0.0)) {
                d.e_active[e] = 0;
                const int h = d.use_pairs ? -1 : d.pose_hidx[d.e_pose[e]];
                if (h >= 0) d.edge_tab[(size_t)h * d.n_points + il] = -1;  // the table only lists active edges
            } else {
                alive++;
            }
        }
    }
    alive += __shfl_xor(alive, 1);
    alive += __shfl_xor(alive, 2);
    alive += __shfl_xor(alive, 4);
    if (live && sub == 0) d.pt_active[il] = alive > 0;
}
__global__ __launch_bounds__(256) void ba_mark_outliers_kernel(BaDev d, double chi2_threshold, int gate) {
    ba_mark_outliers_body(d, chi2_threshold, gate, blockIdx.x, gridDim.x);
}


void launch_ba_mark_outliers(const BaDev& d, double chi2_threshold, int gate, hipStream_t s) {
    if (d.n_points <= 0) return;
    if (g_ba_recorder) {
        BaLaunchRec& r = ba_record(kBaKMarkOutliers, d, (d.n_points + 31) / 32);
        r.f0 = chi2_threshold;
        r.i0 = gate;
        return;
    }
    hipLaunchKernelGGL(ba_mark_outliers_kernel, dim3((d.n_points + 31) / 32), dim3(256), 0, s, d, chi2_threshold, gate);
}

// ---------------- linearisation: Hpp/bp per free pose (one workgroup each), Hll/bl/W per landmark ----------------
// The residuals are recomputed here from the current estimate (the same function on the same state as
// computeActiveErrors, so the same values) rather than read from the stored errors, which describe the last
// TRIAL and are stale after a rejected one.
__device__ __forceinline__ void ba_build_body(const BaDev& d, int gated, const unsigned BX, const unsigned GX) {
    __shared__ double s_red[4][28];
    const BaLm lm = *d.lm;
    if (gated == 1 && !(lm.active == d.stage && lm.need_build)) return;
    if (gated == 2 && (lm.active || lm.stages_begun != d.stage - 1)) return;
    const BaPose* __restrict__ poses = d.pose[lm.cur];
    const double* __restrict__ points = d.pt[lm.cur];
    if ((int)BX < d.n_free) {
        // pose role: reduce J_c^T w J_c (upper 21) and J_c^T omega_r (6) over this pose's active edges
        const int hi = BX, ip = d.free_pose[hi];
        const BaPose P = poses[ip];
        const double* K = d.intr + 4 * ip;
        double acc[27];
#pragma unroll
        for (int i = 0; i < 27; i++) acc[i] = 0.0;
        for (int k = d.pose_off[hi] + threadIdx.x; k < d.pose_off[hi + 1]; k += 256) {
            const int e = d.pose_edges[k];
            if (!d.e_active[e]) continue;
            const int il = d.e_point[e];
            const double X[3] = {points[3 * il], points[3 * il + 1], points[3 * il + 2]};
            double Jp[6], Jc[12], e0, e1;
            edge_jacobians(P, X, K, Jp, Jc);
            const double obs[2] = {d.e_obs[2 * e], d.e_obs[2 * e + 1]};
            edge_error(P, X, obs, K, e0, e1);
            const double om = d.e_w[e];
            const double r1 = d.robust ? huber_rho1(e0 * (om * e0) + e1 * (om * e1), d.huber_delta, d.huber_dsqr) : 1.0;
            const double w = r1 * om;
            const double o0 = -om * e0 * r1, o1 = -om * e1 * r1;
            int t = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = r; c < 6; c++) acc[t++] += Jc[r] * w * Jc[c] + Jc[6 + r] * w * Jc[6 + c];
#pragma unroll
            for (int r = 0; r < 6; r++) acc[21 + r] += Jc[r] * o0 + Jc[6 + r] * o1;
        }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 27; i++) {
            const double v = wave_sum(acc[i]);
            if (lane == 0) s_red[wv][i] = v;
        }
        __syncthreads();
        if (threadIdx.x < 27) {
            const double v = ((s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + s_red[2][threadIdx.x]) + s_red[3][threadIdx.x];
            if (threadIdx.x < 21) {
                int r = 0, t = threadIdx.x;  // unrank the upper-triangular index
                while (t >= 6 - r) { t -= 6 - r; r++; }
                const int c = r + t;
                d.Hpp[36 * (size_t)hi + r * 6 + c] = v;
                d.Hpp[36 * (size_t)hi + c * 6 + r] = v;
            } else {
                d.bp[6 * (size_t)hi + (threadIdx.x - 21)] = v;
            }
        }
        return;
    }
    // landmark role: 8 lanes per landmark (32 landmarks per workgroup); a landmark's edges are contiguous, each
    // lane linearises every 8th one, then a fixed xor-butterfly over the 8 lanes sums Hll (6 unique) and bl (3)
    const int il = (BX - d.n_free) * 32 + (threadIdx.x >> 3);
    const int sub = threadIdx.x & 7;
    const bool live = il < d.n_points && d.pt_active[il];
    double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    if (live) {
        const double X[3] = {points[3 * il], points[3 * il + 1], points[3 * il + 2]};
        for (int e = d.pt_off[il] + sub; e < d.pt_off[il + 1]; e += 8) {
            if (!d.e_active[e]) continue;
            const int ip = d.e_pose[e];
            double Jp[6], Jc[12], e0, e1;
            const BaPose P = poses[ip];
            edge_jacobians(P, X, d.intr + 4 * ip, Jp, Jc);
            const double obs[2] = {d.e_obs[2 * e], d.e_obs[2 * e + 1]};
            edge_error(P, X, obs, d.intr + 4 * ip, e0, e1);
            const double om = d.e_w[e];
            const double r1 = d.robust ? huber_rho1(e0 * (om * e0) + e1 * (om * e1), d.huber_delta, d.huber_dsqr) : 1.0;
            const double w = r1 * om;
            const double o0 = -om * e0 * r1, o1 = -om * e1 * r1;
            H[0] += Jp[0] * w * Jp[0] + Jp[3] * w * Jp[3];
            H[1] += Jp[0] * w * Jp[1] + Jp[3] * w * Jp[4];
            H[2] += Jp[0] * w * Jp[2] + Jp[3] * w * Jp[5];
            H[3] += Jp[1] * w * Jp[1] + Jp[4] * w * Jp[4];
            H[4] += Jp[1] * w * Jp[2] + Jp[4] * w * Jp[5];
            H[5] += Jp[2] * w * Jp[2] + Jp[5] * w * Jp[5];
#pragma unroll
            for (int r = 0; r < 3; r++) b[r] += Jp[r] * o0 + Jp[3 + r] * o1;
            if (d.pose_hidx[ip] >= 0) {
                double* W = d.W + 18 * (size_t)e;  // pose x point = J_c^T w J_p
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) W[r * 3 + c] = Jc[r] * w * Jp[c] + Jc[6 + r] * w * Jp[3 + c];
            }
        }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 6; k++) H[k] += __shfl_xor(H[k], off);
#pragma unroll
        for (int k = 0; k < 3; k++) b[k] += __shfl_xor(b[k], off);
    }
    if (live && sub == 0) {
        double* Hl = d.Hll + 9 * (size_t)il;
        Hl[0] = H[0]; Hl[1] = H[1]; Hl[2] = H[2];
        Hl[3] = H[1]; Hl[4] = H[3]; Hl[5] = H[4];
        Hl[6] = H[2]; Hl[7] = H[4]; Hl[8] = H[5];
        d.bl[3 * (size_t)il] = b[0]; d.bl[3 * (size_t)il + 1] = b[1]; d.bl[3 * (size_t)il + 2] = b[2];
    }
}
__global__ __launch_bounds__(256) void ba_build_kernel(BaDev d, int gated) {
    ba_build_body(d, gated, blockIdx.x, gridDim.x);
}


void launch_ba_build(const BaDev& d, int gate, hipStream_t s) {
    const int nb = d.n_free + (d.n_points + 31) / 32;
    if (nb <= 0) return;
    if (g_ba_recorder) {
        ba_record(kBaKBuild, d, nb).i0 = gate;
        return;
    }
    hipLaunchKernelGGL(ba_build_kernel, dim3(nb), dim3(256), 0, s, d, gate);
}

// ---------------- Schur complement ----------------
__device__ __forceinline__ void damped_inverse3(const double* Hl, double lambda, double* Di) {
    const double m0 = Hl[0] + lambda, m1 = Hl[1], m2 = Hl[2], m3 = Hl[3], m4 = Hl[4] + lambda, m5 = Hl[5], m6 = Hl[6],
                 m7 = Hl[7], m8 = Hl[8] + lambda;
    const double c00 = m4 * m8 - m5 * m7, c01 = m5 * m6 - m3 * m8, c02 = m3 * m7 - m4 * m6;
    const double invdet = 1.0 / (m0 * c00 + m1 * c01 + m2 * c02);
    Di[0] = c00 * invdet; Di[1] = (m2 * m7 - m1 * m8) * invdet; Di[2] = (m1 * m5 - m2 * m4) * invdet;
    Di[3] = c01 * invdet; Di[4] = (m0 * m8 - m2 * m6) * invdet; Di[5] = (m2 * m3 - m0 * m5) * invdet;
    Di[6] = c02 * invdet; Di[7] = (m1 * m6 - m0 * m7) * invdet; Di[8] = (m0 * m4 - m1 * m3) * invdet;
}

// prep: thread i < n_points: Dinv = (Hll + lambda I)^-1 (cofactors, like Eigen's 3x3 inverse), db = Dinv bl;
//       thread i < n_edges : BDinv_e = W_e Dinv (recomputing the 3x3 inverse of its landmark: no dependency
//       between the two roles, so everything is one memory latency deep)
__global__ __launch_bounds__(256) void ba_schur_prep_kernel(BaDev d) {
    if (d.lm->active != d.stage) return;
    const double lambda = d.lm->lambda;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d.n_points && d.pt_active[i]) {
        double Di[9];
        damped_inverse3(d.Hll + 9 * (size_t)i, lambda, Di);
        double* Do = d.Dinv + 9 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 9; k++) Do[k] = Di[k];
        const double* bl = d.bl + 3 * (size_t)i;
#pragma unroll
        for (int r = 0; r < 3; r++) d.db[3 * (size_t)i + r] = Di[r * 3] * bl[0] + Di[r * 3 + 1] * bl[1] + Di[r * 3 + 2] * bl[2];
    }
    if (i < d.n_edges && d.e_active[i] && d.pose_hidx[d.e_pose[i]] >= 0) {
        double Di[9];
        damped_inverse3(d.Hll + 9 * (size_t)d.e_point[i], lambda, Di);
        const double* W = d.W + 18 * (size_t)i;
        double* B = d.BDinv + 18 * (size_t)i;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) B[r * 3 + c] = W[r * 3] * Di[c] + W[r * 3 + 1] * Di[3 + c] + W[r * 3 + 2] * Di[6 + c];
    }
}

// gather: one group of WAVES waves per upper block (i1 <= i2) of the reduced camera system (WAVES = 4: a whole
//         workgroup, local windows; WAVES = 1: large maps, most blocks are short).  The threads stride over pose
//         i1's edges; the partner edge of pose i2 on the same landmark comes from edge_tab (the edge itself on the
//         diagonal).  Each lane accumulates a full 6x6 partial, a fixed xor-butterfly sums the lanes of a wave and
//         the four waves meet in LDS (fixed order):
//         S(i1,i2) = [i1 == i2] (Hpp + lambda I) - sum over shared landmarks BDinv_{e1} W_{e2}^T ; mirrored.
//         A pose of a local window has a few hundred edges: with a wave per block the kernel was five serial rounds
//         of four dependent global loads (21 us on LBA-M); a workgroup per block needs one or two.
//         extra workgroups: b_schur(i) = bp(i) - sum over the pose's edges W_e db_{landmark(e)}
// With pair lists (large maps) only the blocks of big_list are walked here (list_cap = launch bound of the list,
// *big_n of its entries are real); the others belong to ba_schur_gather_small_kernel.
template <int WAVES>
__device__ __forceinline__ void ba_schur_gather_body(const BaDev& d, int list_cap, const unsigned BX, const unsigned GX) {
    __shared__ double s_part[4][36];
    if (d.lm->active != d.stage) return;
    const double lambda = d.lm->lambda;
    const int lane = threadIdx.x & 63, wave = WAVES == 4 ? (int)(threadIdx.x >> 6) : 0;
    const int tid = WAVES == 4 ? (int)threadIdx.x : lane;  // index inside the group
    constexpr int kStride = 64 * WAVES;
    int g = WAVES == 4 ? (int)BX : (int)(BX * 4 + (threadIdx.x >> 6));
    const int nf = d.n_free;
    int n_blk = nf * (nf + 1) / 2;
    if (WAVES == 4) {
        // Workgroups go to the eight XCDs round-robin, each with its own L2: with g = BX every XCD met every
        // keyframe's W blocks (FETCH_SIZE 4.1x the inputs, round 4).  XCD x takes the x-th eighth of the column-ordered
        // block list instead - two or three whole columns i2, so W of those keyframes' edges stays in one L2; the i1 side
        // is shared by all columns whatever the deal (any split of all pairs of 25 keyframes over eight caches re-reads
        // W about three times: DESIGN.md 5).  S(i1,i2) does not depend on which workgroup forms it.
        const int total = n_blk + nf, chunk = (total + 7) >> 3, j = (int)(BX >> 3);
        g = (int)(BX & 7) * chunk + j;
        if (j >= chunk || g >= total) return;
    }
    if (d.use_pairs) {
        if (g < list_cap) {
            if (g >= *d.big_n) return;
            g = d.big_list[g];
        } else {
            g = n_blk + (g - list_cap);  // the right-hand-side workgroups follow the list
        }
    }
    if (g < n_blk) {
        // upper blocks column by column: (0,0) (0,1) (1,1) (0,2) (1,2) (2,2) ... - consecutive blocks share i2
        int i2 = (int)((sqrt(8.0 * (double)g + 1.0) - 1.0) * 0.5);
        while ((i2 + 1) * (i2 + 2) / 2 <= g) i2++;
        while (i2 * (i2 + 1) / 2 > g) i2--;
        const int i1 = g - i2 * (i2 + 1) / 2;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; k++) acc[k] = 0.0;
        const bool from_list = d.use_pairs && i1 != i2;  // large maps: the block's own (landmark-sorted) pair list
        const int p_lo = from_list ? d.pr_off[g] : d.pose_off[i1], p_hi = from_list ? d.pr_off[g + 1] : d.pose_off[i1 + 1];
        // lmk >= 0 (local windows: no prep launch): BDinv_{k1} = W_{k1} (Hll + lambda I)^-1 is formed here, by the very
        // expressions ba_schur_prep_kernel uses (same bits); lmk < 0: read what that kernel stored
        auto add_pair = [&](int k1, int k2, int lmk) {
            const double* W = d.W + 18 * (size_t)k2;
            double b[18], w[18];
            if (lmk >= 0) {
                double Di[9];
                damped_inverse3(d.Hll + 9 * (size_t)lmk, lambda, Di);
                const double* W1 = d.W + 18 * (size_t)k1;
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) b[r * 3 + c] = W1[r * 3] * Di[c] + W1[r * 3 + 1] * Di[3 + c] + W1[r * 3 + 2] * Di[6 + c];
            } else {
                const double* B = d.BDinv + 18 * (size_t)k1;
#pragma unroll
                for (int k = 0; k < 18; k++) b[k] = B[k];
            }
#pragma unroll
            for (int k = 0; k < 18; k++) w[k] = W[k];
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c < 6; c++)
                    acc[r * 6 + c] += b[r * 3] * w[c * 3] + b[r * 3 + 1] * w[c * 3 + 1] + b[r * 3 + 2] * w[c * 3 + 2];
        };
        if (from_list) {
            for (int p = p_lo + tid; p < p_hi; p += kStride) {
                const int k1 = d.ps_k1[p], k2 = d.ps_k2[p];
                if (!d.e_active[k1] || !d.e_active[k2]) continue;  // dropped between the stages
                add_pair(k1, k2, -1);
            }
        } else {
            // pose i1's edges, four per thread and round: the look-ups of a round go out together (edge + landmark ->
            // partner edge from the table, which only lists active edges), so a round is three dependent loads deep
            // however few of its edges pose i2 shares
            const int* tab = d.edge_tab + (size_t)i2 * d.n_points;
            for (int base = p_lo; base < p_hi; base += 4 * kStride) {
                int k1[4], k2[4], lmk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int p = base + u * kStride + tid;
                    k1[u] = p < p_hi ? d.pose_edges[p] : -1;
                    lmk[u] = p < p_hi ? d.pose_edge_point[p] : 0;  // the edge's landmark
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool live = k1[u] >= 0 && d.e_active[k1[u]];
                    const int partner = (i1 == i2) ? k1[u] : (k1[u] >= 0 ? tab[lmk[u]] : -1);
                    k2[u] = live ? partner : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (k2[u] >= 0) add_pair(k1[u], k2[u], d.fold_prep ? lmk[u] : -1);
            }
        }
        // wave totals by transposition (the pose kernel's scheme): values 0..31 end up in lane pairs, 32..35 by butterfly
        {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = po_swap_add32(acc[i], acc[i + 16]);
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = po_swap_add16(acc[i], acc[i + 8]);
            po_halve<4>(acc, 8, lane);
            po_halve<2>(acc, 4, lane);
            po_halve<1>(acc, 2, lane);
            const double tot = acc[0] + __shfl_xor(acc[0], 1);  // the wave total of value (lane >> 1)
#pragma unroll
            for (int k = 32; k < 36; k++) acc[k] = wave_sum(acc[k]);
            const double extra = lane == 0 ? acc[32] : lane == 1 ? acc[33] : lane == 2 ? acc[34] : acc[35];
            if (WAVES == 4) {
                if ((lane & 1) == 0) s_part[wave][lane >> 1] = tot;
                if (lane < 4) s_part[wave][32 + lane] = extra;
            } else {
                // one wave per block: value k to lane k
                const double low = __shfl(tot, 2 * (lane & 31));
                const double high = __shfl(extra, lane & 3);
                acc[0] = lane < 32 ? low : high;
            }
        }
        if (WAVES == 4) __syncthreads();
        if (tid < 36) {
            const int r = tid / 6, c = tid - 6 * r;
            double out = WAVES == 4 ? -(((s_part[0][tid] + s_part[1][tid]) + s_part[2][tid]) + s_part[3][tid]) : -acc[0];
            if (i1 == i2) out += d.Hpp[36 * (size_t)i1 + tid] + (r == c ? lambda : 0.0);
            d.S[(size_t)(6 * i1 + r) * d.ldS + 6 * i2 + c] = out;
            if (i1 != i2) d.S[(size_t)(6 * i2 + c) * d.ldS + 6 * i1 + r] = out;
        }
        return;
    }
    const int hi = g - n_blk;
    if (hi >= d.n_free) return;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int k = d.pose_off[hi] + tid; k < d.pose_off[hi + 1]; k += kStride) {
        const int e = d.pose_edges[k];
        if (!d.e_active[e]) continue;
        const double* W = d.W + 18 * (size_t)e;
        double db[3];
        if (!d.fold_prep) {
            const double* dbp = d.db + 3 * (size_t)d.e_point[e];
            db[0] = dbp[0]; db[1] = dbp[1]; db[2] = dbp[2];
        } else {  // db = Dinv bl as ba_schur_prep_kernel forms it
            double Di[9];
            damped_inverse3(d.Hll + 9 * (size_t)d.e_point[e], lambda, Di);
            const double* bl = d.bl + 3 * (size_t)d.e_point[e];
#pragma unroll
            for (int r = 0; r < 3; r++) db[r] = Di[r * 3] * bl[0] + Di[r * 3 + 1] * bl[1] + Di[r * 3 + 2] * bl[2];
        }
#pragma unroll
        for (int r = 0; r < 6; r++) acc[r] += W[r * 3] * db[0] + W[r * 3 + 1] * db[1] + W[r * 3 + 2] * db[2];
    }
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const double v = wave_sum(acc[r]);
        if (WAVES == 4) {
            if (lane == 0) s_part[wave][r] = v;
        } else if (lane == 0) {
            d.bs[6 * (size_t)hi + r] = d.bp[6 * (size_t)hi + r] - v;
        }
    }
    if (WAVES == 4) {
        __syncthreads();
        if (tid < 6) d.bs[6 * (size_t)hi + tid] = d.bp[6 * (size_t)hi + tid] - (((s_part[0][tid] + s_part[1][tid]) + s_part[2][tid]) + s_part[3][tid]);
    }
}
template <int WAVES>
__global__ __launch_bounds__(256) void ba_schur_gather_kernel(BaDev d, int list_cap) {
    ba_schur_gather_body<WAVES>(d, list_cap, blockIdx.x, gridDim.x);
}


// ---------------- dense Cholesky solve of the reduced camera system (single workgroup) ----------------
// Replaces LinearSolverEigen's SimplicialLDLT (linear_solver_eigen.h:94-124): same solution up to rounding.
//
// Fast path (n_free <= 43): 6x6-block right-looking Cholesky held entirely in REGISTERS.  The lower block
// triangle of S plus one extra block row carrying the right-hand side is dealt out in block-COLUMN order, one
// block (36 doubles) per thread slot, so the blocks of a finished column form a prefix of the thread range and
// whole waves retire as the factorisation advances.  Per block step k: the owner of (k,k) factors its block
// (right-looking inside the block, reciprocal square roots from v_rsq_f64 + Newton steps instead of the IEEE
// sqrt/div sequences: the step's serial chain is what bounds this kernel), the owners of column k solve
// against it and publish their blocks through LDS, and every owner of a trailing block applies its rank-6
// update from two LDS blocks.  Two barriers per 6 columns, no global memory between the initial load and the
// final store.  Larger systems (global bundle adjustment) go to the blocked multi-workgroup solver in ba_dense.hip.  The forward substitution falls out of the extra block row; the backward substitution multiplies
// by the inverted diagonal blocks (inverted off the critical path, all at once after the factorisation).
constexpr int kSolveMaxNB = 44;

// tools/probe/solve_probe.hip defines SO_SOLVE_MARK to log clock64() per phase; the product build compiles it away
#ifndef SO_SOLVE_MARK
#define SO_SOLVE_MARK(k, phase)
#define SO_SOLVE_MARK_DECL
#endif

// 1/sqrt(v) to double precision: v_rsq_f64 is good to about 2^-23, so one third-order step
// y (1 + e/2 + 3 e^2/8), e = 1 - v y^2, is enough - four dependent operations instead of the six of two Newton steps
__device__ __forceinline__ double rsqrt_newton(double v) {
    const double y = __builtin_amdgcn_rsq(v);
    const double t = v * y;
    const double e = fma(-t, y, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(y * e, p, y);
}

template <int THREADS, int BPT>
__global__ __launch_bounds__(THREADS) void ba_solve_reg_kernel(BaDev d) {
    __shared__ double s_diag[36];
    __shared__ double s_dinv[kSolveMaxNB][6];
    __shared__ double s_panel[kSolveMaxNB][37];  // 37: lanes reading the same element of 32 different blocks hit distinct banks
    __shared__ double s_Linv[kSolveMaxNB][36];
    __shared__ double s_y[kSolveMaxNB][6];
    __shared__ double s_x[6];
    __shared__ int s_fail;
    if (d.lm->active != d.stage) return;
    SO_SOLVE_MARK_DECL;
    const int tid = threadIdx.x;
    const int nf = d.n_free, NB = nf + 1, nblk = NB * (NB + 1) / 2;
    double a[BPT][36];
    int bI[BPT], bJ[BPT];
    if (tid == 0) s_fail = 0;
#pragma unroll
    for (int s = 0; s < BPT; s++) {
        const int p = tid + s * THREADS;
        bI[s] = -1;
        bJ[s] = -1;
#pragma unroll
        for (int k = 0; k < 36; k++) a[s][k] = 0.0;
        if (p < nblk - 1) {  // the last block would be (nf, nf): the right-hand side has no diagonal block
            int J = 0, rem = p;
            while (rem >= NB - J) {
                rem -= NB - J;
                J++;
            }
            const int I = J + rem;
            bI[s] = I;
            bJ[s] = J;
            if (I < nf) {
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 6; c++) a[s][r * 6 + c] = d.S[(size_t)(6 * I + r) * d.ldS + 6 * J + c];
            } else {
#pragma unroll
                for (int c = 0; c < 6; c++) a[s][c] = d.bs[6 * J + c];  // right-hand side rides as row 0
            }
        }
    }
    __syncthreads();
    SO_SOLVE_MARK(0, 0);
    for (int k = 0; k < nf; k++) {
        SO_SOLVE_MARK(k, 1);
#pragma unroll
        for (int s = 0; s < BPT; s++) {
            if (bI[s] == k && bJ[s] == k) {  // factor the diagonal block in place (lower), publish it
                double* A = a[s];
                bool bad = false;
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const double v = A[c * 6 + c];
                    if (!(v > 0.0)) bad = true;
                    const double y = rsqrt_newton(v);
                    double l = v * y;
                    l = fma(fma(-l, l, v), 0.5 * y, l);  // one correction step: l = sqrt(v) to the last bit or so
                    A[c * 6 + c] = l;
                    s_dinv[k][c] = y;
#pragma unroll
                    for (int r = c + 1; r < 6; r++) A[r * 6 + c] *= y;
#pragma unroll
                    for (int r = c + 1; r < 6; r++)
#pragma unroll
                        for (int c2 = c + 1; c2 <= r; c2++) A[r * 6 + c2] -= A[r * 6 + c] * A[c2 * 6 + c];
#pragma unroll
                    for (int r = 0; r < c; r++) A[r * 6 + c] = 0.0;
                }
#pragma unroll
                for (int q = 0; q < 36; q++) s_diag[q] = A[q];
                if (bad) s_fail = 1;
            }
        }
        __syncthreads();
        SO_SOLVE_MARK(k, 2);
        if (s_fail) break;
#pragma unroll
        for (int s = 0; s < BPT; s++) {
            if (bJ[s] == k && bI[s] > k) {  // X = A L_kk^-T, column by column (right-looking)
                double* A = a[s];
                double L[36], ri[6];
#pragma unroll
                for (int q = 0; q < 36; q++) L[q] = s_diag[q];
#pragma unroll
                for (int c = 0; c < 6; c++) ri[c] = s_dinv[k][c];
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        const double x = A[r * 6 + c] * ri[c];
                        A[r * 6 + c] = x;
#pragma unroll
                        for (int c2 = c + 1; c2 < 6; c2++) A[r * 6 + c2] -= x * L[c2 * 6 + c];
                    }
#pragma unroll
                for (int q = 0; q < 36; q++) s_panel[bI[s]][q] = A[q];
                if (bI[s] == nf) {
#pragma unroll
                    for (int c = 0; c < 6; c++) s_y[k][c] = A[c];
                }
            }
        }
        __syncthreads();
        SO_SOLVE_MARK(k, 3);
#pragma unroll
        for (int s = 0; s < BPT; s++) {
            if (bJ[s] > k) {  // trailing update A_IJ -= L_Ik L_Jk^T  (bI >= bJ > k)
                double* A = a[s];
                const double* Pj = s_panel[bJ[s]];
                const double* Pi = s_panel[bI[s]];
                double pj[36];
#pragma unroll
                for (int q = 0; q < 36; q++) pj[q] = Pj[q];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    double pi[6];
#pragma unroll
                    for (int m = 0; m < 6; m++) pi[m] = Pi[r * 6 + m];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double v = A[r * 6 + c];
#pragma unroll
                        for (int m = 0; m < 6; m++) v = fma(-pi[m], pj[c * 6 + m], v);
                        A[r * 6 + c] = v;
                    }
                }
            }
        }
        // no barrier here: s_panel / s_diag are rewritten only after the next barrier pair
        SO_SOLVE_MARK(k, 4);
    }
    __syncthreads();
    SO_SOLVE_MARK(0, 5);
    if (!s_fail) {
#pragma unroll
        for (int s = 0; s < BPT; s++) {
            if (bI[s] == bJ[s] && bI[s] >= 0) {  // invert the factored diagonal block (lower triangular)
                const double* L = a[s];
                const int K = bI[s];
                double ri[6], X[36];
#pragma unroll
                for (int c = 0; c < 6; c++) ri[c] = s_dinv[K][c];
#pragma unroll
                for (int q = 0; q < 36; q++) X[q] = 0.0;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    X[j * 6 + j] = ri[j];
#pragma unroll
                    for (int i = j + 1; i < 6; i++) {
                        double v = 0.0;
#pragma unroll
                        for (int m = j; m < i; m++) v = fma(L[i * 6 + m], X[m * 6 + j], v);
                        X[i * 6 + j] = -v * ri[i];
                    }
                }
#pragma unroll
                for (int q = 0; q < 36; q++) s_Linv[K][q] = X[q];
            }
        }
        __syncthreads();
        for (int K = nf - 1; K >= 0; K--) {  // L^T x = y, block columns right to left
            if (tid < 6) {                   // x_K = L_KK^-T y_K
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < 6; r++) v = fma(s_Linv[K][r * 6 + tid], s_y[K][r], v);  // zeros above the diagonal
                s_x[tid] = v;
                d.bs[6 * K + tid] = v;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < BPT; s++) {
                if (bI[s] == K && bJ[s] < K) {  // y_J -= L_KJ^T x_K (one block per J in this step)
                    const double* A = a[s];
                    double x[6];
#pragma unroll
                    for (int r = 0; r < 6; r++) x[r] = s_x[r];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double v = s_y[bJ[s]][c];
#pragma unroll
                        for (int r = 0; r < 6; r++) v = fma(-A[r * 6 + c], x[r], v);
                        s_y[bJ[s]][c] = v;
                    }
                }
            }
            __syncthreads();
        }
    }
    SO_SOLVE_MARK(0, 6);
    if (tid == 0) d.partial[kBaSolveOk] = s_fail ? 0.0 : 1.0;
}

// ---- look-ahead variant (n_free <= 30): the factorisation's serial chain gets its own waves ----
// The kernel above spends most of a block step waiting: the diagonal factor (one lane) -> barrier -> the panel
// solve (one lane per block) -> barrier -> the trailing update (everyone).  Here two teams of waves run
// decoupled, synchronised through LDS counters instead of workgroup barriers:
//   panel team  (PW waves, one matrix ROW of the current block column per lane; lanes 0-5 of every panel wave
//               hold the six rows of the diagonal block, replicated): loads its rows of column k from the LDS
//               stage, applies the one update they still lack (that of column k-1), then factors and solves in
//               one pass - per column c the pivot and the five multipliers below it travel by v_readlane, so
//               there is no cross-lane memory traffic and no barrier - and publishes the finished panel.
//   update team (UW waves, one 6x6 block per lane in registers, block-column order): waits for panel k,
//               applies it to its block, and the owners of column k+2 copy their block to the stage, one step
//               AHEAD of its use, so the panel team never waits for a trailing update that is not its own.
// Counters are per step / per column: a running total would let waves that are ahead cover for one behind.
constexpr int kLaMaxNB = 31;

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void lds_wait_ge(int* flag, int target) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}

__device__ __forceinline__ void lds_signal(int* flag) {  // one increment per wave, after the wave's LDS stores
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int UW, int PW>
__global__ __launch_bounds__((UW + PW) * 64) void ba_solve_la_kernel(BaDev d) {
    __shared__ double s_col[2][kLaMaxNB][37];    // stage: block column k (parity k & 1), lacking the update of k-1
    __shared__ double s_panel[2][kLaMaxNB][37];  // finished panel of column k (parity k & 1), rows of blocks I > k
    __shared__ double s_Ldiag[kLaMaxNB][36];
    __shared__ double s_dinv[kLaMaxNB][6];
    __shared__ double s_Linv[kLaMaxNB][36];
    __shared__ double s_y[kLaMaxNB][6];
    __shared__ double s_x[6];
    __shared__ int s_colcnt[kLaMaxNB + 2];  // update-team waves that have staged their part of column J
    __shared__ int s_updcnt[kLaMaxNB + 2];  // update-team waves that have finished step k
    __shared__ int s_panel_ready, s_fail;
    if (d.lm->active != d.stage) return;
    SO_SOLVE_MARK_DECL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nf = d.n_free, NB = nf + 1, nblk = NB * (NB + 1) / 2 - 1;
    const bool panel_team = wave < PW;
    for (int i = tid; i < kLaMaxNB + 2; i += (UW + PW) * 64) {
        s_colcnt[i] = 0;
        s_updcnt[i] = 0;
    }
    for (int i = tid; i < 2 * 37; i += (UW + PW) * 64) {  // rows 1-5 of the right-hand-side "block" stay zero
        s_panel[i / 37][nf][i % 37] = 0.0;
    }
    if (tid == 0) {
        s_panel_ready = 0;
        s_fail = 0;
    }
    // ---- update team: own one block (I, J), J <= I, blocks numbered column by column ----
    double a[36];
    int bI = -1, bJ = -1;
#pragma unroll
    for (int q = 0; q < 36; q++) a[q] = 0.0;
    if (!panel_team) {
        const int p = tid - PW * 64;
        if (p < nblk) {
            int J = 0, rem = p;
            while (rem >= NB - J) {
                rem -= NB - J;
                J++;
            }
            bI = J + rem;
            bJ = J;
            if (bI < nf) {
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 6; c++) a[r * 6 + c] = d.S[(size_t)(6 * bI + r) * d.ldS + 6 * bJ + c];
            } else {
#pragma unroll
                for (int c = 0; c < 6; c++) a[c] = d.bs[6 * bJ + c];  // right-hand side rides as row 0
            }
        }
    }
    __syncthreads();  // counters and zero rows are in place
    SO_SOLVE_MARK(40, 2);
    if (!panel_team) {
        // columns 0 and 1 go to the stage as loaded (column 1 lacks the update of column 0, as the stage expects)
        if (bJ == 0 || bJ == 1) {
#pragma unroll
            for (int q = 0; q < 36; q++) s_col[bJ][bI][q] = a[q];
        }
        if (__ballot(bJ == 0)) lds_signal(&s_colcnt[0]);
        if (__ballot(bJ == 1)) lds_signal(&s_colcnt[1]);
        for (int k = 0; k < nf; k++) {
            SO_SOLVE_MARK(k, 0);
            lds_wait_ge(&s_panel_ready, PW * (k + 1));
            SO_SOLVE_MARK(k, 1);
            if (__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            const int buf = k & 1;
            if (bJ == k && bI > k) {  // my block is final now: keep L_Ik for the backward substitution
#pragma unroll
                for (int q = 0; q < 36; q++) a[q] = s_panel[buf][bI][q];
            }
            if (bJ >= k + 2) {  // trailing update A_IJ -= L_Ik L_Jk^T  (column k+1 gets it from the panel team)
                const double* Pj = s_panel[buf][bJ];
                const double* Pi = s_panel[buf][bI];
                double pj[36];
#pragma unroll
                for (int q = 0; q < 36; q++) pj[q] = Pj[q];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    double pi[6];
#pragma unroll
                    for (int m = 0; m < 6; m++) pi[m] = Pi[r * 6 + m];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double v = a[r * 6 + c];
#pragma unroll
                        for (int m = 0; m < 6; m++) v = fma(-pi[m], pj[c * 6 + m], v);
                        a[r * 6 + c] = v;
                    }
                }
                if (bJ == k + 2) {
#pragma unroll
                    for (int q = 0; q < 36; q++) s_col[buf][bI][q] = a[q];  // (k + 2) & 1 == k & 1
                }
            }
            if (__ballot(bJ == k + 2)) lds_signal(&s_colcnt[k + 2]);
            lds_signal(&s_updcnt[k]);
            SO_SOLVE_MARK(k, 2);
        }
    } else {
        __builtin_amdgcn_s_setprio(3);
        const int rho = wave * 58 + lane - 6;  // panel row handled by this lane (lanes 0-5: diagonal rows)
        const int rq = rho / 6, rr = rho - 6 * rq;
        int col_start = 0;
        for (int k = 0; k < nf; k++) {
            // number of update-team waves that hold a piece of column k (its blocks are contiguous in p)
            const int len = NB - k;
            const int owners = (col_start + len - 1) / 64 - col_start / 64 + 1;
            col_start += len;
            const int buf = k & 1;
            const bool diag_lane = lane < 6;
            const bool live = diag_lane || (lane >= 6 && rho <= 6 * (nf - k - 1));
            const int I = diag_lane ? k : k + 1 + rq, r = diag_lane ? lane : rr;
            SO_SOLVE_MARK(k, 0);
            lds_wait_ge(&s_colcnt[k], owners);
            SO_SOLVE_MARK(k, 1);
            double x[6];
#pragma unroll
            for (int c = 0; c < 6; c++) x[c] = live ? s_col[buf][I][r * 6 + c] : 0.0;
            if (k > 0) {  // the update of column k-1, which the stage copy does not have yet
                lds_wait_ge(&s_panel_ready, PW * k);
                SO_SOLVE_MARK(k, 2);
                const double* Pk = s_panel[buf ^ 1][k];
                double own[6];
#pragma unroll
                for (int m = 0; m < 6; m++) own[m] = live ? s_panel[buf ^ 1][I][r * 6 + m] : 0.0;
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double v = x[c];
#pragma unroll
                    for (int m = 0; m < 6; m++) v = fma(-own[m], Pk[c * 6 + m], v);
                    x[c] = v;
                }
            }
            SO_SOLVE_MARK(k, 3);
            // factor the diagonal block (lanes 0-5) and solve the rows below it in the same pass
            bool bad = false;
            double ys[6];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const double v = readlane_f64(x[c], c);
                if (!(v > 0.0)) bad = true;
                const double y = rsqrt_newton(v);
                ys[c] = y;
                x[c] *= y;
#pragma unroll
                for (int c2 = c + 1; c2 < 6; c2++) {
                    const double l = readlane_f64(x[c], c2);  // L(c2, c)
                    x[c2] = fma(-x[c], l, x[c2]);
                }
            }
            if (wave == 0 && diag_lane) {
#pragma unroll
                for (int c = 0; c < 6; c++) s_Ldiag[k][lane * 6 + c] = (c <= lane) ? x[c] : 0.0;
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 6; c++) s_dinv[k][c] = ys[c];
                    if (bad) __hip_atomic_store(&s_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            SO_SOLVE_MARK(k, 4);
            if (k >= 2) lds_wait_ge(&s_updcnt[k - 2], UW);  // nobody reads panel k-2 any more
            SO_SOLVE_MARK(k, 5);
            if (live && !diag_lane) {
#pragma unroll
                for (int c = 0; c < 6; c++) s_panel[buf][I][r * 6 + c] = x[c];
                if (I == nf) {
#pragma unroll
                    for (int c = 0; c < 6; c++) s_y[k][c] = x[c];
                }
            }
            lds_signal(&s_panel_ready);
            SO_SOLVE_MARK(k, 6);
            if (bad) break;  // uniform: every panel wave factors the same diagonal block
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    SO_SOLVE_MARK(40, 0);
    const bool failed = s_fail != 0;
    if (!failed) {
        if (tid < nf) {  // invert the diagonal blocks (lower triangular), one per thread
            const double* L = s_Ldiag[tid];
            double ri[6], X[36];
#pragma unroll
            for (int c = 0; c < 6; c++) ri[c] = s_dinv[tid][c];
#pragma unroll
            for (int q = 0; q < 36; q++) X[q] = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                X[j * 6 + j] = ri[j];
#pragma unroll
                for (int i = j + 1; i < 6; i++) {
                    double v = 0.0;
#pragma unroll
                    for (int m = j; m < i; m++) v = fma(L[i * 6 + m], X[m * 6 + j], v);
                    X[i * 6 + j] = -v * ri[i];
                }
            }
#pragma unroll
            for (int q = 0; q < 36; q++) s_Linv[tid][q] = X[q];
        }
        __syncthreads();
        for (int K = nf - 1; K >= 0; K--) {  // L^T x = y, block columns right to left
            if (tid < 6) {                   // x_K = L_KK^-T y_K
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < 6; r++) v = fma(s_Linv[K][r * 6 + tid], s_y[K][r], v);
                s_x[tid] = v;
                d.bs[6 * K + tid] = v;
            }
            __syncthreads();
            if (bI == K && bJ < K) {  // y_J -= L_KJ^T x_K (one block per J in this step)
                double xk[6];
#pragma unroll
                for (int r = 0; r < 6; r++) xk[r] = s_x[r];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double v = s_y[bJ][c];
#pragma unroll
                    for (int r = 0; r < 6; r++) v = fma(-a[r * 6 + c], xk[r], v);
                    s_y[bJ][c] = v;
                }
            }
            __syncthreads();
        }
    }
    SO_SOLVE_MARK(40, 1);
    if (tid == 0) d.partial[kBaSolveOk] = failed ? 0.0 : 1.0;
}

// One thread per upper block with at most kBaSmallBlockPairs pairs (the common case of a large map, including
// "no landmark in common"): sums -BDinv W^T over its (landmark-sorted) pairs, writes the block and its mirror.
__global__ __launch_bounds__(256) void ba_schur_gather_small_kernel(BaDev d, int n_blk) {
    if (d.lm->active != d.stage) return;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_blk) return;
    const int o = d.pr_off[g], np = d.pr_off[g + 1] - o;
    if (np > kBaSmallBlockPairs) return;
    int i2 = (int)((sqrt(8.0 * (double)g + 1.0) - 1.0) * 0.5);
    while ((i2 + 1) * (i2 + 2) / 2 <= g) i2++;
    while (i2 * (i2 + 1) / 2 > g) i2--;
    const int i1 = g - i2 * (i2 + 1) / 2;
    if (i1 == i2) return;
    int k1[kBaSmallBlockPairs], k2[kBaSmallBlockPairs];  // the list is sorted by landmark (ba_pairs_sort_kernel)
#pragma unroll
    for (int q = 0; q < kBaSmallBlockPairs; q++) {
        const bool in = q < np;
        k1[q] = in ? d.ps_k1[o + q] : -1;
        k2[q] = in ? d.ps_k2[o + q] : -1;
    }
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = 0.0;
#pragma unroll
    for (int q = 0; q < kBaSmallBlockPairs; q++) {
        if (k1[q] < 0 || !(d.e_active[k1[q]] && d.e_active[k2[q]])) continue;
        const double* B = d.BDinv + 18 * (size_t)k1[q];
        const double* W = d.W + 18 * (size_t)k2[q];
        double b[18], w[18];
#pragma unroll
        for (int k = 0; k < 18; k++) { b[k] = B[k]; w[k] = W[k]; }
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++)
                acc[r * 6 + c] += b[r * 3] * w[c * 3] + b[r * 3 + 1] * w[c * 3 + 1] + b[r * 3 + 2] * w[c * 3 + 2];
    }
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double out = -acc[r * 6 + c];
            d.S[(size_t)(6 * i1 + r) * d.ldS + 6 * i2 + c] = out;
            d.S[(size_t)(6 * i2 + c) * d.ldS + 6 * i1 + r] = out;
        }
}

static void launch_ba_schur(const BaDev& d, hipStream_t s) {
    const int nthreads = d.n_points > d.n_edges ? d.n_points : d.n_edges;
    const int n_blk = d.n_free * (d.n_free + 1) / 2;
    if (g_ba_recorder) {
        const int groups = n_blk + d.n_free;
        if (d.fold_prep && !d.use_pairs && groups > 0) {  // a local window: the one launch that has a grouped form
            ba_record(kBaKGather4, d, 8 * ((groups + 7) / 8));
        } else {
            BaRecorder* rec = g_ba_recorder;
            BaLaunchRec& r = ba_record(kBaKSolo, d, 1);
            r.solo = [d](hipStream_t st) { launch_ba_schur(d, st); };
            (void)rec;
        }
        return;
    }
    // Local windows (no pair lists, at most 43 free keyframes: d.fold_prep) have no prep launch: the gather and the update form (Hll + lambda I)^-1, W Dinv and
    // Dinv bl themselves where they need them - the same expressions, hence the same bits, one launch (9.5 us of an 87 us
    // trial on LBA-M) less.  Larger windows and pair-list maps read each product many times over: they keep the stored copies.
    if (!d.fold_prep && nthreads > 0) hipLaunchKernelGGL(ba_schur_prep_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, s, d);
    if (d.use_pairs) {
        if (n_blk > 0) hipLaunchKernelGGL(ba_schur_gather_small_kernel, dim3((n_blk + 255) / 256), dim3(256), 0, s, d, n_blk);
        const int waves = d.big_cap + d.n_free;  // the walked blocks, then the right-hand sides
        hipLaunchKernelGGL(ba_schur_gather_kernel<1>, dim3((waves + 3) / 4), dim3(256), 0, s, d, d.big_cap);
        return;
    }
    const int groups = n_blk + d.n_free;  // a local window: a workgroup per block
    if (groups > 0) hipLaunchKernelGGL(ba_schur_gather_kernel<4>, dim3(8 * ((groups + 7) / 8)), dim3(256), 0, s, d, 0);
}

static void launch_ba_solve(const BaDev& d, hipStream_t s) {
    const int NB = d.n_free + 1, nblk = NB * (NB + 1) / 2 - 1, rows = 6 * (d.n_free - 1) + 1;
    static const bool classic = getenv("SWARMORB_BA_CLASSIC_SOLVER") != nullptr;  // A/B switches for profiling
    static const bool no_mfma = getenv("SWARMORB_BA_NO_MFMA_SOLVER") != nullptr;
    if (!classic && !no_mfma && d.n_free >= kBaMfmaSolverMinFree && launch_ba_solve_mfma(d, s)) return;  // (records by itself)
    if (g_ba_recorder) {  // every other solver: as it is, behind the group's grouped launches
        BaLaunchRec& r = ba_record(kBaKSolo, d, 1);
        r.solo = [d](hipStream_t st) { launch_ba_solve(d, st); };
        return;
    }
    if (!classic && nblk <= 256 && rows <= 2 * 58)
        hipLaunchKernelGGL((ba_solve_la_kernel<4, 2>), dim3(1), dim3(6 * 64), 0, s, d);
    else if (!classic && nblk <= 384 && rows <= 3 * 58)
        hipLaunchKernelGGL((ba_solve_la_kernel<6, 3>), dim3(1), dim3(9 * 64), 0, s, d);
    else if (!classic && nblk <= 512 && rows <= 4 * 58 && NB <= kLaMaxNB)
        hipLaunchKernelGGL((ba_solve_la_kernel<8, 4>), dim3(1), dim3(12 * 64), 0, s, d);
    else if (nblk <= 256)
        hipLaunchKernelGGL((ba_solve_reg_kernel<256, 1>), dim3(1), dim3(256), 0, s, d);
    else if (nblk <= 512)
        hipLaunchKernelGGL((ba_solve_reg_kernel<512, 1>), dim3(1), dim3(512), 0, s, d);
    else if (NB <= kSolveMaxNB)
        hipLaunchKernelGGL((ba_solve_reg_kernel<512, 2>), dim3(1), dim3(512), 0, s, d);
    else if (d.use_pcg)
        launch_ba_pcg_solve(d, s);    // ba_pcg.hip: block-Jacobi PCG over the nonzero 6 x 6 blocks (so_ba_set_linear_solver)
    else
        launch_ba_dense_solve(d, s);  // ba_dense.hip: blocked Cholesky on the FP64 matrix cores
}

// ---------------- back-substitution + manifold update into the trial buffers + scale partials ----------------
__global__ __launch_bounds__(256) void ba_update_kernel(BaDev d) {
    __shared__ double s_tmp[16];
    const BaLm lm = *d.lm;
    if (lm.active != d.stage) return;
    const double lambda = lm.lambda;
    const BaPose* __restrict__ poses = d.pose[lm.cur];
    const double* __restrict__ points = d.pt[lm.cur];
    BaPose* __restrict__ poses_trial = d.pose[lm.cur ^ 1];
    double* __restrict__ points_trial = d.pt[lm.cur ^ 1];
    double scale = 0.0;  // computeScale: sum x (lambda x + b)
    const double* xp = d.bs;
    // landmarks: 8 lanes each (xl = Dinv (bl - sum_e W_e^T x_pose(e))), then one thread per keyframe
    const int total = 8 * d.n_points + d.n_poses;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        if (i < 8 * d.n_points) {
            const int il = i >> 3, sub = i & 7;
            const bool act = d.pt_active[il];
            double cl[3] = {0, 0, 0};
            if (act) {
                for (int e = d.pt_off[il] + sub; e < d.pt_off[il + 1]; e += 8) {
                    if (!d.e_active[e]) continue;
                    const int hi = d.pose_hidx[d.e_pose[e]];
                    if (hi < 0) continue;
                    const double* W = d.W + 18 * (size_t)e;
                    const double* x = xp + 6 * (size_t)hi;
#pragma unroll
                    for (int c = 0; c < 3; c++)
#pragma unroll
                        for (int r = 0; r < 6; r++) cl[c] -= W[r * 3 + c] * x[r];
                }
            }
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
#pragma unroll
                for (int c = 0; c < 3; c++) cl[c] += __shfl_xor(cl[c], off);
            }
            if (sub == 0) {
                double X[3] = {points[3 * il], points[3 * il + 1], points[3 * il + 2]};
                if (act) {
                    const double* bl = d.bl + 3 * (size_t)il;
                    cl[0] += bl[0]; cl[1] += bl[1]; cl[2] += bl[2];
                    double Di[9];  // (a copy in registers either way: a pointer that selects between a local array and
                                   //  global memory put the array into scratch memory)
                    if (d.fold_prep) {
                        damped_inverse3(d.Hll + 9 * (size_t)il, lambda, Di);  // (no prep launch: same bits)
                    } else {
#pragma unroll
                        for (int q = 0; q < 9; q++) Di[q] = d.Dinv[9 * (size_t)il + q];
                    }
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const double xl = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
                        d.xl[3 * (size_t)il + r] = xl;
                        scale += xl * (lambda * xl + bl[r]);
                        X[r] += xl;
                    }
                }
                points_trial[3 * il] = X[0]; points_trial[3 * il + 1] = X[1]; points_trial[3 * il + 2] = X[2];
            }
        } else {
            const int ip = i - 8 * d.n_points;
            const int hi = d.pose_hidx[ip];
            if (hi >= 0) {
                const double* x = xp + 6 * (size_t)hi;
                const double u[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
                BaPose out;
                se3_exp_mul(u, poses[ip], out);
                poses_trial[ip] = out;
#pragma unroll
                for (int r = 0; r < 6; r++) scale += u[r] * (lambda * u[r] + d.bp[6 * (size_t)hi + r]);
            } else {
                poses_trial[ip] = poses[ip];
            }
        }
    }
    const double t = block_sum(scale, s_tmp);
    if (threadIdx.x == 0) d.partial[kBaPartialScale + blockIdx.x] = t;
}

// ---------------- the same + the trial's residuals in ONE launch (local windows) ----------------
// ba_update_kernel followed by ba_errors_kernel(which = 1) costs two launch floors (~4.5 us each on a 7 + 5 us pair); the
// residuals of an edge need the TRIAL pose and the TRIAL point of its two ends, which the update writes from different
// workgroups.  Here every workgroup forms all trial poses itself (a window has at most 128 keyframes: 25 exponential maps by
// 25 threads, the fixed ones copied) into LDS, then walks landmarks the way the update does - eight lanes each - and, once the
// landmark's trial position is known, the same eight lanes compute the residuals of that landmark's edges.  Same
// expressions per pose / point / edge as the two kernels (same bits in the trial buffers, e_err, e_chi2); the chi2
// partials are per workgroup of THIS grid (the decision kernel sums nb_upd of them, in index order: deterministic).
constexpr int kBaFusedMaxPoses = 128;
__device__ __forceinline__ void ba_update_errors_body(const BaDev& d, const unsigned BX, const unsigned GX) {
    __shared__ BaPose s_pose[kBaFusedMaxPoses];
    __shared__ double s_tmp[16];
    const BaLm lm = *d.lm;
    if (lm.active != d.stage) return;
    const double lambda = lm.lambda;
    const BaPose* __restrict__ poses = d.pose[lm.cur];
    const double* __restrict__ points = d.pt[lm.cur];
    BaPose* __restrict__ poses_trial = d.pose[lm.cur ^ 1];
    double* __restrict__ points_trial = d.pt[lm.cur ^ 1];
    const double* xp = d.bs;
    double scale = 0.0, chi = 0.0;
    for (int ip = threadIdx.x; ip < d.n_poses; ip += 256) {
        const int hi = d.pose_hidx[ip];
        BaPose out = poses[ip];
        if (hi >= 0) {
            const double* x = xp + 6 * (size_t)hi;
            const double u[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
            se3_exp_mul(u, poses[ip], out);
            if (BX == 0) {
#pragma unroll
                for (int r = 0; r < 6; r++) scale += u[r] * (lambda * u[r] + d.bp[6 * (size_t)hi + r]);
            }
        }
        s_pose[ip] = out;
        if (BX == 0) poses_trial[ip] = out;
    }
    __syncthreads();
    const int total = 8 * d.n_points;
    for (int i = BX * 256 + threadIdx.x; i < total; i += GX * 256) {
        const int il = i >> 3, sub = i & 7;
        const bool act = d.pt_active[il];
        double cl[3] = {0, 0, 0};
        if (act) {
            for (int e = d.pt_off[il] + sub; e < d.pt_off[il + 1]; e += 8) {
                if (!d.e_active[e]) continue;
                const int hi = d.pose_hidx[d.e_pose[e]];
                if (hi < 0) continue;
                const double* W = d.W + 18 * (size_t)e;
                const double* x = xp + 6 * (size_t)hi;
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int r = 0; r < 6; r++) cl[c] -= W[r * 3 + c] * x[r];
            }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) cl[c] += __shfl_xor(cl[c], off);
        }
        double X[3] = {points[3 * il], points[3 * il + 1], points[3 * il + 2]};
        if (sub == 0) {
            if (act) {
                const double* bl = d.bl + 3 * (size_t)il;
                cl[0] += bl[0]; cl[1] += bl[1]; cl[2] += bl[2];
                double Di[9];
                damped_inverse3(d.Hll + 9 * (size_t)il, lambda, Di);  // (local windows: no prep launch, as in ba_update_kernel)
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const double xl = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
                    d.xl[3 * (size_t)il + r] = xl;
                    scale += xl * (lambda * xl + bl[r]);
                    X[r] += xl;
                }
            }
            points_trial[3 * il] = X[0]; points_trial[3 * il + 1] = X[1]; points_trial[3 * il + 2] = X[2];
        }
        // the landmark's trial position to its eight lanes, then its edges' residuals at the trial state
        const int lead = (int)(threadIdx.x & 63) & ~7;
#pragma unroll
        for (int c = 0; c < 3; c++) X[c] = __shfl(X[c], lead);
        for (int e = d.pt_off[il] + sub; e < d.pt_off[il + 1]; e += 8) {
            if (!d.e_active[e]) continue;
            const int ip = d.e_pose[e];
            double e0, e1;
            const double obs[2] = {d.e_obs[2 * e], d.e_obs[2 * e + 1]};
            edge_error(s_pose[ip], X, obs, d.intr + 4 * ip, e0, e1);
            const double w = d.e_w[e];
            const double chi2 = e0 * (w * e0) + e1 * (w * e1);
            d.e_err[2 * e] = e0;
            d.e_err[2 * e + 1] = e1;
            d.e_chi2[e] = chi2;
            chi += d.robust ? huber_rho0(chi2, d.huber_delta, d.huber_dsqr) : chi2;
        }
    }
    const double ts = block_sum(scale, s_tmp);
    const double tc = block_sum(chi, s_tmp);
    if (threadIdx.x == 0) {
        d.partial[kBaPartialScale + BX] = ts;
        d.partial[kBaPartialChi + BX] = tc;
    }
}
__global__ __launch_bounds__(256) void ba_update_errors_kernel(BaDev d) {
    ba_update_errors_body(d, blockIdx.x, gridDim.x);
}


// ---------------- Levenberg-Marquardt control on the device ----------------
// Start of SparseOptimizer::optimize(iterations): chi2 of the current estimate (the error kernel ran on it),
// computeLambdaInit (optimization_algorithm_levenberg.cpp:166-180: tau * max diagonal, tau = 1e-5).
// Start of a stage (SparseOptimizer::optimize up to the first solve): chi2 of the estimate from the partials of the
// error pass, lambda = 1e-5 max |diagonal| over Hpp and Hll (computeLambdaInit,
// optimization_algorithm_levenberg.cpp:166-180; max is order-independent), LM state reset.  One workgroup; the maximum
// used to be a launch of its own that walked the 3 n_points diagonal entries one dependent load at a time (19 us for
// 9600 landmarks) - here a thread takes whole landmarks, three independent loads each.
__device__ __forceinline__ void ba_stage_begin_body(const BaDev& d, int nb_err, int iterations, BaLm* __restrict__ lm_host, int gate,
                                                               const uint8_t* __restrict__ abort_flag, const unsigned BX, const unsigned GX) {
    __shared__ double s_tmp[16];
    __shared__ double s_m[16];
    // gate 2: chained behind the previous stage's trials - the stage starts on the device as soon as that one is over
    // (and not at all if it is still running, or a stop was requested: the host then starts it the ordinary way)
    if (gate == 2 && (d.lm->active || d.lm->stages_begun != d.stage - 1 ||
                      (abort_flag && *reinterpret_cast<const volatile uint8_t*>(abort_flag))))
        return;
    double m = 0.0;
    for (int i = threadIdx.x; i < d.n_free; i += 1024) {
        const double* H = d.Hpp + 36 * (size_t)i;
        m = fmax(m, fmax(fmax(fmax(fabs(H[0]), fabs(H[7])), fmax(fabs(H[14]), fabs(H[21]))), fmax(fabs(H[28]), fabs(H[35]))));
    }
    for (int i = threadIdx.x; i < d.n_points; i += 1024)
        if (d.pt_active[i]) {
            const double* H = d.Hll + 9 * (size_t)i;
            m = fmax(m, fmax(fmax(fabs(H[0]), fabs(H[4])), fabs(H[8])));
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    double v = 0.0;  // summed by the first 256 threads, as when this kernel had 256 (same association, same bits)
    if (threadIdx.x < 256)
        for (int i = threadIdx.x; i < nb_err; i += 256) v += d.partial[kBaPartialChi + i];
    const double chi = block_sum(v, s_tmp);  // (its barriers also cover s_m)
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; i++) t = fmax(t, s_m[i]);
        d.partial[kBaMaxDiag] = t;
        BaLm& lm = *d.lm;
        lm.prev_done = lm.done;        // what the stage before this one ended with (the host may look only after this launch)
        lm.prev_chi_out = lm.chi_out;
        lm.prev_chi_begin = lm.chi_begin;
        lm.stages_begun++;
        lm.currentChi = chi;
        lm.iniChi = chi;
        lm.tempChi = chi;
        lm.chi_out = chi;
        lm.chi_begin = chi;
        lm.rho = 0.0;
        lm.lambda = 1e-5 * t;
        lm.ni = 2.0;
        lm.nBad = 0;
        lm.it = 0;
        lm.iterations = iterations;
        lm.qmax = 0;
        lm.done = 0;
        lm.need_build = 0;
        lm.active = iterations > 0 ? d.stage : 0;  // the stage's tag: kernels launched for another stage return at once
        if (lm_host) *lm_host = lm;
    }
}
__global__ __launch_bounds__(1024) void ba_stage_begin_kernel(BaDev d, int nb_err, int iterations, BaLm* __restrict__ lm_host, int gate,
                                                               const uint8_t* __restrict__ abort_flag) {
    ba_stage_begin_body(d, nb_err, iterations, lm_host, gate, abort_flag, blockIdx.x, gridDim.x);
}


void launch_ba_stage_begin(const BaDev& d, int nb_err, int iterations, BaLm* lm_host, int gate, const uint8_t* abort_flag, hipStream_t s) {
    if (g_ba_recorder) {
        BaLaunchRec& r = ba_record(kBaKStageBegin, d, 1);
        r.i0 = nb_err; r.i1 = iterations; r.i2 = gate;
        r.p0 = lm_host; r.p1 = const_cast<uint8_t*>(abort_flag);
        return;
    }
    hipLaunchKernelGGL(ba_stage_begin_kernel, dim3(1), dim3(1024), 0, s, d, nb_err, iterations, lm_host, gate, abort_flag);
}

// End of a trial: OptimizationAlgorithmLevenberg::solve's accept / reject (optimization_algorithm_levenberg.cpp:
// 95-148) and SparseOptimizer::optimize's stopping rules (sparse_optimizer.cpp:355-420).
__device__ __forceinline__ void ba_trial_decide_body(const BaDev& d, int nb_err, int nb_upd,
                                                              const uint8_t* __restrict__ abort_flag,
                                                              BaLm* __restrict__ lm_host, const unsigned BX, const unsigned GX) {
    __shared__ double s_tmp[16];
    if (d.lm->active != d.stage) return;
    double c = 0.0, sc = 0.0;
    for (int i = threadIdx.x; i < nb_err; i += 256) c += d.partial[kBaPartialChi + i];
    for (int i = threadIdx.x; i < nb_upd; i += 256) sc += d.partial[kBaPartialScale + i];
    const double chi = block_sum(c, s_tmp);
    const double scale_sum = block_sum(sc, s_tmp);
    if (threadIdx.x != 0) return;
    BaLm lm = *d.lm;
    const bool abort = abort_flag && *reinterpret_cast<const volatile uint8_t*>(abort_flag);
    const bool solved = d.partial[kBaSolveOk] != 0.0;
    double tempChi = solved ? chi : 1.7976931348623157e308;
    double rho = (lm.currentChi - tempChi) / (scale_sum + 1e-3);
    const bool accepted = rho > 0 && isfinite(tempChi);
    if (accepted) {
        double alpha = 1. - pow(2 * rho - 1, 3);
        alpha = fmin(alpha, 2. / 3.);
        lm.lambda *= fmax(1. / 3., alpha);
        lm.ni = 2;
        lm.currentChi = tempChi;
        lm.cur ^= 1;  // discardTop: the trial becomes the estimate
    } else {
        lm.lambda *= lm.ni;  // pop: the estimate stays, the stored errors describe the rejected trial
        lm.ni *= 2;
    }
    lm.tempChi = tempChi;
    lm.rho = rho;
    lm.qmax++;
    lm.trials++;
    if (!(rho < 0 && lm.qmax < 10 && !abort)) {  // the iteration is over
        lm.done++;
        lm.chi_out = tempChi;
        bool ok = true;
        if (lm.qmax == 10 || rho == 0) {
            ok = false;
        } else {
            if ((lm.iniChi - lm.currentChi) * 1e3 < lm.iniChi) lm.nBad++; else lm.nBad = 0;
            if (lm.nBad >= 3) ok = false;
        }
        lm.it++;
        if (lm.it < lm.iterations && !abort && ok) {
            lm.need_build = accepted ? 1 : 0;
            lm.iniChi = lm.currentChi;
            lm.qmax = 0;
        } else {
            lm.active = 0;
        }
    } else {
        lm.need_build = 0;
    }
    *d.lm = lm;
    if (lm_host) *lm_host = lm;
}
__global__ __launch_bounds__(256) void ba_trial_decide_kernel(BaDev d, int nb_err, int nb_upd,
                                                              const uint8_t* __restrict__ abort_flag,
                                                              BaLm* __restrict__ lm_host) {
    ba_trial_decide_body(d, nb_err, nb_upd, abort_flag, lm_host, blockIdx.x, gridDim.x);
}


// =====================================================================================================
// A stage's trials as ONE resident launch.  The chain above costs a window ~70 dependent launches; alone on the GPU they follow
// each other within a microsecond or two, but with several agents' chains, tracking stages and extraction chains on the same
// device every one of them queues behind whatever the command processor is dispatching, and a window that takes 1.0 ms alone
// takes 3 ms next to seven others (NOTES G.8).  Here the workgroups stay: n_workgroups of 256 threads walk the virtual blocks of
// each phase (the bodies take their block index as an argument: same partial sums, same bits as the launches), meet at a grid
// barrier (device-scope release / acquire around an arrival counter), workgroup 0 runs the MFMA solve on four waves and the
// decision, and the loop goes on until the stage is over - lm->active leaves the stage's tag - or max_trials are done.
// Every wait is bounded (BaDev::flow_timeout_ticks, as in the dataflow solves of ba_dense.hip): a workgroup that is not resident
// makes the others give up, the host repeats the call on the chain of launches and keeps that path (so_bundle_adjust).
// Residency is the caller's business: the workgroups hold the solve's LDS (one per CU), ba.cpp limits how many windows run
// resident at once.
// =====================================================================================================
struct ResWatch {
    BaResidentSync* y;
    unsigned* abort_host;
    unsigned long long deadline;
    unsigned epoch;
};
__device__ __forceinline__ bool res_expired(const ResWatch& W) {  // one thread
    if (__hip_atomic_load(&W.y->abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == W.epoch) return true;
    if (wall_clock64() > W.deadline) {
        __hip_atomic_store(&W.y->abort_w, W.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (W.abort_host) __hip_atomic_store(W.abort_host, W.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
    return false;
}
// every thread of every workgroup; `ordinal` counts the launch's barriers from 1.  Returns false once the launch has been given up.
__device__ __forceinline__ bool res_barrier(const ResWatch& W, unsigned n_wg, unsigned ordinal, int* s_dead) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this workgroup's part of the phase is out ...
    __syncthreads();
    if (threadIdx.x == 0 && !*s_dead) {
        const unsigned target = (W.epoch << 11) | ordinal;
        if (__hip_atomic_fetch_add(&W.y->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_wg - 1) {
            __hip_atomic_store(&W.y->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&W.y->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned polls = 0;
            while (__hip_atomic_load(&W.y->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++polls & 255u) == 0 && res_expired(W)) {
                    *s_dead = 1;
                    break;
                }
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ... and everybody else's is visible
    return !*s_dead;
}

__global__ __launch_bounds__(256) void ba_lm_resident_kernel(BaDev d, int nb_build, int nb_gather, int nb_upd, int max_trials,
                                                             const uint8_t* __restrict__ abort_flag, BaLm* __restrict__ lm_host,
                                                             BaResidentSync* __restrict__ sync, unsigned epoch) {
    extern __shared__ double s_tiles[];  // the solve's tiles (workgroup 0; the others carry the allocation: one workgroup per CU)
    __shared__ int s_dead;
    const unsigned G = gridDim.x;
    ResWatch W;
    W.y = sync;
    W.abort_host = d.flow_abort_host;
    W.deadline = wall_clock64() + d.flow_timeout_ticks;
    W.epoch = epoch;
    if (threadIdx.x == 0)  // (a launch of this call that gave up already: run through)
        s_dead = W.abort_host != nullptr && __hip_atomic_load(W.abort_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ? 1 : 0;
    __syncthreads();
    unsigned ordinal = 0;
    for (int t = 0; t < max_trials && !s_dead; t++) {
        // the LM state as of the last barrier (the launch boundary for the first trial): the same value in every workgroup
        if (__hip_atomic_load(&d.lm->active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != d.stage) break;
        for (unsigned bx = blockIdx.x; bx < (unsigned)nb_build; bx += G) {
            ba_build_body(d, kBaGateActive, bx, (unsigned)nb_build);
            __syncthreads();  // (the bodies' LDS scratch is reused by the next virtual block)
        }
        if (!res_barrier(W, G, ++ordinal, &s_dead)) break;
        for (unsigned bx = blockIdx.x; bx < (unsigned)nb_gather; bx += G) {
            ba_schur_gather_body<4>(d, 0, bx, (unsigned)nb_gather);
            __syncthreads();
        }
        if (!res_barrier(W, G, ++ordinal, &s_dead)) break;
        if (blockIdx.x == 0) ba_solve_mfma_body<256>(d, s_tiles);
        if (!res_barrier(W, G, ++ordinal, &s_dead)) break;
        for (unsigned bx = blockIdx.x; bx < (unsigned)nb_upd; bx += G) {
            ba_update_errors_body(d, bx, (unsigned)nb_upd);
            __syncthreads();
        }
        if (!res_barrier(W, G, ++ordinal, &s_dead)) break;
        if (blockIdx.x == 0) ba_trial_decide_body(d, nb_upd, nb_upd, abort_flag, lm_host, 0, 1);
        if (!res_barrier(W, G, ++ordinal, &s_dead)) break;
    }
}

bool launch_ba_trials_resident(const BaDev& d, int nb_upd, int max_trials, const uint8_t* abort_flag, BaLm* lm_host,
                               BaResidentSync* sync, unsigned epoch, int n_workgroups, hipStream_t s) {
    // the windows the fused update + the MFMA solve in LDS cover (what launch_ba_trial sends down the five-launch path)
    if (!(d.fold_prep && d.n_poses <= kBaFusedMaxPoses && !d.use_pairs && !d.use_pcg) || d.n_free < kBaMfmaSolverMinFree) return false;
    const int n = 6 * d.n_free, NT = (n + 1 + 15) / 16;
    constexpr int kResidentMaxTiles = 10;  // 55 tiles = 127 KB beside the bodies' ~20 KB of static LDS
    if (NT < 2 || NT > kResidentMaxTiles || n_workgroups < 2 || !sync) return false;
    const int nb_build = d.n_free + (d.n_points + 31) / 32;
    const int groups = d.n_free * (d.n_free + 1) / 2 + d.n_free, nb_gather = 8 * ((groups + 7) / 8);
    if (nb_build <= 0 || nb_upd <= 0) return false;
    const size_t lds = sizeof(double) * (size_t)(NT * (NT + 1) / 2) * kMTile;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ba_lm_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(double) * (kResidentMaxTiles * (kResidentMaxTiles + 1) / 2) * kMTile)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[dev] = true;
    }
    max_trials = std::min(max_trials, kBaResidentMaxTrials);
    hipLaunchKernelGGL(ba_lm_resident_kernel, dim3((unsigned)n_workgroups), dim3(256), lds, s, d, nb_build, nb_gather, nb_upd, max_trials,
                       abort_flag, lm_host, sync, epoch);
    return true;
}

void launch_ba_trial(const BaDev& d, int nb_err, int nb_upd, const uint8_t* abort_flag, BaLm* lm_host, hipEvent_t ev0,
                     hipEvent_t ev1, hipStream_t s) {
    launch_ba_build(d, kBaGateActive, s);
    launch_ba_schur(d, s);
    if (ev0) (void)hipEventRecord(ev0, s);
    launch_ba_solve(d, s);
    if (ev1) (void)hipEventRecord(ev1, s);
    // local windows: back-substitution, manifold update and the trial's residuals in one launch (SWARMORB_BA_FUSE_UPDATE=0:
    // the two kernels)
    static const bool fuse_env = !(getenv("SWARMORB_BA_FUSE_UPDATE") && atoi(getenv("SWARMORB_BA_FUSE_UPDATE")) == 0);
    if (fuse_env && d.fold_prep && d.n_poses <= kBaFusedMaxPoses && !d.use_pairs) {
        if (g_ba_recorder) {
            ba_record(kBaKUpdateErrors, d, nb_upd);
            BaLaunchRec& r = ba_record(kBaKDecide, d, 1);
            r.i0 = nb_upd; r.i1 = nb_upd;
            r.p0 = const_cast<uint8_t*>(abort_flag); r.p1 = lm_host;
            return;
        }
        hipLaunchKernelGGL(ba_update_errors_kernel, dim3(nb_upd), dim3(256), 0, s, d);
        hipLaunchKernelGGL(ba_trial_decide_kernel, dim3(1), dim3(256), 0, s, d, nb_upd, nb_upd, abort_flag, lm_host);
        return;
    }
    if (g_ba_recorder) {
        BaLaunchRec& u = ba_record(kBaKSolo, d, 1);
        u.solo = [d, nb_upd](hipStream_t st) { hipLaunchKernelGGL(ba_update_kernel, dim3(nb_upd), dim3(256), 0, st, d); };
        launch_ba_errors(d, 1, kBaGateActive, nb_err, s);
        BaLaunchRec& r = ba_record(kBaKDecide, d, 1);
        r.i0 = nb_err; r.i1 = nb_upd;
        r.p0 = const_cast<uint8_t*>(abort_flag); r.p1 = lm_host;
        return;
    }
    hipLaunchKernelGGL(ba_update_kernel, dim3(nb_upd), dim3(256), 0, s, d);
    launch_ba_errors(d, 1, kBaGateActive, nb_err, s);
    hipLaunchKernelGGL(ba_trial_decide_kernel, dim3(1), dim3(256), 0, s, d, nb_err, nb_upd, abort_flag, lm_host);
}

// ---- grouped forms (so_ba_group): member y of the launch = row A.row[y] of the BaDev table, its own grid A.grid[y] ----
#define BA_GROUP_PROLOGUE                                    \
    const int m_ = blockIdx.y;                               \
    if ((int)blockIdx.x >= A.grid[m_]) return;               \
    const BaDev d = rows[A.row[m_]];                         \
    const unsigned BX = blockIdx.x, GX = (unsigned)A.grid[m_];

__global__ __launch_bounds__(256) void ba_errors_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    BA_GROUP_PROLOGUE
    ba_errors_body(d, A.i0[m_], A.i1[m_], BX, GX);
}
__global__ __launch_bounds__(256) void ba_build_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    BA_GROUP_PROLOGUE
    ba_build_body(d, A.i0[m_], BX, GX);
}
__global__ __launch_bounds__(1024) void ba_stage_begin_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    BA_GROUP_PROLOGUE
    ba_stage_begin_body(d, A.i0[m_], A.i1[m_], static_cast<BaLm*>(A.p0[m_]), A.i2[m_], static_cast<const uint8_t*>(A.p1[m_]), BX, GX);
}
__global__ __launch_bounds__(256) void ba_schur_gather4_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    BA_GROUP_PROLOGUE
    ba_schur_gather_body<4>(d, 0, BX, GX);
}
__global__ __launch_bounds__(256) void ba_update_errors_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    BA_GROUP_PROLOGUE
    ba_update_errors_body(d, BX, GX);
}
__global__ __launch_bounds__(256) void ba_trial_decide_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    BA_GROUP_PROLOGUE
    ba_trial_decide_body(d, A.i0[m_], A.i1[m_], static_cast<const uint8_t*>(A.p0[m_]), static_cast<BaLm*>(A.p1[m_]), BX, GX);
}
__global__ __launch_bounds__(256) void ba_mark_outliers_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    BA_GROUP_PROLOGUE
    ba_mark_outliers_body(d, A.f0[m_], A.i0[m_], BX, GX);
}
__global__ __launch_bounds__(256) void ba_finish_group_kernel(const BaDev* __restrict__ rows, BaGroupArgs A) {
    BA_GROUP_PROLOGUE
    ba_finish_body(d, A.f0[m_], static_cast<BaPose*>(A.p0[m_]), static_cast<double*>(A.p1[m_]), static_cast<double*>(A.p2[m_]),
                   static_cast<uint8_t*>(A.p3[m_]), BX, GX);
}
__global__ void ba_signal_group_kernel(BaGroupArgs A) {
    const int m_ = threadIdx.x;
    if (m_ >= A.n) return;
    __threadfence_system();
    __hip_atomic_store(static_cast<int*>(A.p0[m_]), A.i0[m_], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#undef BA_GROUP_PROLOGUE

void launch_ba_group(int kind, const BaDev* d_rows, const BaGroupArgs& A, int max_grid, size_t lds, hipStream_t s) {
    const dim3 grid((unsigned)std::max(max_grid, 1), (unsigned)A.n);
    switch (kind) {
        case kBaKErrors: hipLaunchKernelGGL(ba_errors_group_kernel, grid, dim3(256), 0, s, d_rows, A); break;
        case kBaKBuild: hipLaunchKernelGGL(ba_build_group_kernel, grid, dim3(256), 0, s, d_rows, A); break;
        case kBaKStageBegin: hipLaunchKernelGGL(ba_stage_begin_group_kernel, grid, dim3(1024), 0, s, d_rows, A); break;
        case kBaKGather4: hipLaunchKernelGGL(ba_schur_gather4_group_kernel, grid, dim3(256), 0, s, d_rows, A); break;
        case kBaKSolveMfma: launch_ba_solve_mfma_group(d_rows, A, lds, s); break;
        case kBaKUpdateErrors: hipLaunchKernelGGL(ba_update_errors_group_kernel, grid, dim3(256), 0, s, d_rows, A); break;
        case kBaKDecide: hipLaunchKernelGGL(ba_trial_decide_group_kernel, grid, dim3(256), 0, s, d_rows, A); break;
        case kBaKMarkOutliers: hipLaunchKernelGGL(ba_mark_outliers_group_kernel, grid, dim3(256), 0, s, d_rows, A); break;
        case kBaKFinish: hipLaunchKernelGGL(ba_finish_group_kernel, grid, dim3(256), 0, s, d_rows, A); break;
        case kBaKSignal: hipLaunchKernelGGL(ba_signal_group_kernel, dim3(1), dim3(64), 0, s, A); break;
        default: break;
    }
}

}  // namespace so

// =====================================================================================================
// Optimizer::PoseOptimization (code/src/Optimizer.cc:239-434): ONE workgroup runs the whole schedule —
// 4 rounds x optimize(10) of Levenberg-Marquardt on a single SE3 vertex with n unary
// EdgeSE3ProjectXYZOnlyPose edges (types_six_dof_expmap.{h:143-171,cpp:266-296}), Huber kernel, outlier
// re-classification between rounds — without a single host round trip.  The 6x6 normal equations are block
// reductions over the edges (fixed order), the 6x6 solve and the SE3 update run on one lane.
// =====================================================================================================
namespace so {

__device__ __forceinline__ void po_edge_err(const PoseOptArgs& a, int e, const BaPose& T, double& e0, double& e1) {
    const double X[3] = {(double)a.Xw[3 * e], (double)a.Xw[3 * e + 1], (double)a.Xw[3 * e + 2]};
    double pc[3];
    camera_point(T, X, pc);
    e0 = (double)a.obs[2 * e] - (pc[0] / pc[2] * a.K[0] + a.K[2]);
    e1 = (double)a.obs[2 * e + 1] - (pc[1] / pc[2] * a.K[1] + a.K[3]);
}

__device__ __forceinline__ void po_edge_jac(const PoseOptArgs& a, int e, const BaPose& T, double* J) {
    const double X[3] = {(double)a.Xw[3 * e], (double)a.Xw[3 * e + 1], (double)a.Xw[3 * e + 2]};
    double pc[3];
    camera_point(T, X, pc);
    const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz, fx = a.K[0], fy = a.K[1];
    J[0] = x * y * invz_2 * fx;       J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx;
    J[3] = -invz * fx;                J[4] = 0;                            J[5] = x * invz_2 * fx;
    J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy;         J[8] = -x * invz * fy;
    J[9] = 0;                         J[10] = -invz * fy;                  J[11] = y * invz_2 * fy;
}

constexpr int kPoThreads = 256;

__global__ __launch_bounds__(kPoThreads) void pose_opt_kernel(PoseOptArgs a) {
    __shared__ double s_tmp[16];
    __shared__ double s_red[4][28];
    __shared__ double s_H[36], s_b[6], s_x[6];
    __shared__ BaPose s_cur, s_trial;
    __shared__ int s_ctrl[4];  // [0] trial accepted, [1] keep trying, [2] stop optimize(), [3] solve ok
    __shared__ double s_lm[4]; // lambda, ni, currentChi, scale
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.n;
    const double delta = (double)sqrtf(5.991f);  // const float deltaMono = sqrt(5.991)
    const float dsqr = (float)(delta * delta);
    int robust = 1, nBad = 0, its_total = 0, trials_total = 0;

    for (int e = tid; e < n; e += kPoThreads) a.outlier[e] = 0;
    __syncthreads();

    // residuals of the active edges for pose T -> stored error; returns the robustified chi2 (block sum)
    auto errors = [&](const BaPose& T) -> double {
        double acc = 0.0;
        for (int e = tid; e < n; e += kPoThreads) {
            if (a.outlier[e]) continue;  // level 1
            double e0, e1;
            po_edge_err(a, e, T, e0, e1);
            a.err[2 * e] = e0;
            a.err[2 * e + 1] = e1;
            const double w = (double)a.inv_sigma2[e];
            const double chi2 = e0 * (w * e0) + e1 * (w * e1);
            acc += robust ? huber_rho0(chi2, delta, dsqr) : chi2;
        }
        return block_sum(acc, s_tmp);
    };

    for (int round = 0; round < 4; round++) {
        if (tid == 0) s_cur = a.init;  // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw))
        int n_active_local = 0;
        for (int e = tid; e < n; e += kPoThreads) n_active_local += !a.outlier[e];
        const int n_active = (int)block_sum((double)n_active_local, s_tmp);
        __syncthreads();
        if (n_active > 0) {
            bool fresh = false;  // stored errors describe s_cur
            double carried = 0.0;
            if (tid == 0) { s_ctrl[2] = 0; }
            int nBadLM = 0;
            __syncthreads();
            for (int it = 0; it < 10; it++) {
                if (s_ctrl[2]) break;
                const BaPose cur = s_cur;
                double currentChi = fresh ? carried : errors(cur);
                const double iniChi = currentChi;
                // build the 6x6 system: upper 21 of J^T w J and the 6 of -rho' J^T Omega e
                double acc[27];
#pragma unroll
                for (int k = 0; k < 27; k++) acc[k] = 0.0;
                for (int e = tid; e < n; e += kPoThreads) {
                    if (a.outlier[e]) continue;
                    double J[12];
                    po_edge_jac(a, e, cur, J);
                    const double w = (double)a.inv_sigma2[e];
                    const double e0 = a.err[2 * e], e1 = a.err[2 * e + 1];
                    const double r1 = robust ? huber_rho1(e0 * (w * e0) + e1 * (w * e1), delta, dsqr) : 1.0;
                    const double wo = r1 * w;
                    int t = 0;
#pragma unroll
                    for (int r = 0; r < 6; r++)
#pragma unroll
                        for (int c = r; c < 6; c++) acc[t++] += J[r] * wo * J[c] + J[6 + r] * wo * J[6 + c];
#pragma unroll
                    for (int r = 0; r < 6; r++) acc[21 + r] -= r1 * (J[r] * (w * e0) + J[6 + r] * (w * e1));
                }
#pragma unroll
                for (int k = 0; k < 27; k++) {
                    const double v = wave_sum(acc[k]);
                    if (lane == 0) s_red[wv][k] = v;
                }
                __syncthreads();
                if (tid < 27) {
                    const double v = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
                    if (tid < 21) {
                        int r = 0, t = tid;
                        while (t >= 6 - r) { t -= 6 - r; r++; }
                        const int c = r + t;
                        s_H[r * 6 + c] = v;
                        s_H[c * 6 + r] = v;
                    } else {
                        s_b[tid - 21] = v;
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    if (it == 0) {  // computeLambdaInit
                        double md = 0.0;
                        for (int j = 0; j < 6; j++) md = fmax(md, fabs(s_H[7 * j]));
                        s_lm[0] = 1e-5 * md;
                        s_lm[1] = 2.0;
                    }
                    s_lm[2] = currentChi;
                }
                if (it == 0) nBadLM = 0;
                __syncthreads();
                int qmax = 0;
                double rho = 0.0;
                do {
                    if (tid == 0) {  // (H + lambda I) x = b by Cholesky, then trial = exp(x) * cur
                        double A[36], x[6];
                        const double lambda = s_lm[0];
                        for (int q = 0; q < 36; q++) A[q] = s_H[q];
                        for (int j = 0; j < 6; j++) A[7 * j] += lambda;
                        for (int j = 0; j < 6; j++) x[j] = s_b[j];
                        int ok = 1;
                        for (int j = 0; j < 6 && ok; j++) {
                            double dj = A[j * 6 + j];
                            for (int k = 0; k < j; k++) dj -= A[j * 6 + k] * A[j * 6 + k];
                            if (!(dj > 0.0)) { ok = 0; break; }
                            dj = sqrt(dj);
                            A[j * 6 + j] = dj;
                            for (int i = j + 1; i < 6; i++) {
                                double v = A[i * 6 + j];
                                for (int k = 0; k < j; k++) v -= A[i * 6 + k] * A[j * 6 + k];
                                A[i * 6 + j] = v / dj;
                            }
                        }
                        if (ok) {
                            for (int i = 0; i < 6; i++) {
                                double v = x[i];
                                for (int k = 0; k < i; k++) v -= A[i * 6 + k] * x[k];
                                x[i] = v / A[i * 6 + i];
                            }
                            for (int i = 5; i >= 0; i--) {
                                double v = x[i];
                                for (int k = i + 1; k < 6; k++) v -= A[k * 6 + i] * x[k];
                                x[i] = v / A[i * 6 + i];
                            }
                        }
                        double scale = 0.0;
                        for (int j = 0; j < 6; j++) {
                            s_x[j] = x[j];
                            scale += x[j] * (lambda * x[j] + s_b[j]);
                        }
                        s_lm[3] = scale + 1e-3;
                        s_ctrl[3] = ok;
                        BaPose tr;
                        se3_exp_mul(x, s_cur, tr);
                        s_trial = tr;
                    }
                    __syncthreads();
                    const BaPose trial = s_trial;
                    double tempChi = errors(trial);
                    if (!s_ctrl[3]) tempChi = 1.7976931348623157e308;
                    if (tid == 0) {
                        double r = (s_lm[2] - tempChi) / s_lm[3];
                        if (a.trace && trials_total < 256) {
                            a.trace[4 * trials_total] = s_lm[0];
                            a.trace[4 * trials_total + 1] = tempChi;
                            a.trace[4 * trials_total + 2] = r;
                            a.trace[4 * trials_total + 3] = s_lm[2];
                        }
                        if (r > 0 && isfinite(tempChi)) {
                            double alpha = 1. - pow((2 * r - 1), 3);
                            alpha = fmin(alpha, 2. / 3.);
                            s_lm[0] *= fmax(1. / 3., alpha);
                            s_lm[1] = 2.0;
                            s_lm[2] = tempChi;
                            s_cur = s_trial;  // discardTop
                            s_ctrl[0] = 1;
                        } else {
                            s_lm[0] *= s_lm[1];
                            s_lm[1] *= 2.0;
                            s_ctrl[0] = 0;  // pop
                        }
                        s_red[0][27] = r;
                    }
                    __syncthreads();
                    rho = s_red[0][27];
                    fresh = s_ctrl[0] != 0;
                    if (fresh) carried = tempChi;
                    qmax++;
                    trials_total++;
                    __syncthreads();
                } while (rho < 0 && qmax < 10);
                its_total++;
                currentChi = s_lm[2];
                bool stop = false;
                if (qmax == 10 || rho == 0) stop = true;
                else {
                    if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
                    if (nBadLM >= 3) stop = true;
                }
                __syncthreads();
                if (tid == 0) s_ctrl[2] = stop ? 1 : 0;
                __syncthreads();
            }
        }
        __syncthreads();
        // classify (Optimizer.cc:357-380): outliers of the previous round get a fresh error, inliers keep the stored one
        const BaPose fin = s_cur;
        int bad_local = 0;
        for (int e = tid; e < n; e += kPoThreads) {
            if (a.outlier[e]) {
                double e0, e1;
                po_edge_err(a, e, fin, e0, e1);
                a.err[2 * e] = e0;
                a.err[2 * e + 1] = e1;
            }
            const double w = (double)a.inv_sigma2[e];
            const double e0 = a.err[2 * e], e1 = a.err[2 * e + 1];
            const float chi2 = (float)(e0 * (w * e0) + e1 * (w * e1));
            const bool out = chi2 > 5.991f;
            a.outlier[e] = out ? 1 : 0;
            bad_local += out;
        }
        nBad = (int)block_sum((double)bad_local, s_tmp);
        if (round == 2) robust = 0;
        __syncthreads();
        if (n < 10) break;
    }
    if (tid == 0) {
        *a.pose_out = s_cur;
        a.info[0] = nBad;
        a.info[1] = its_total;
        a.info[2] = trials_total;
    }
}

// results of a PoseOptimization launch are complete in (host-mapped) memory: raise the completion word behind them.
// The system-scope fences are needed: with s_waitcnt + a relaxed store instead (no L2 write-back) the word reaches the
// host before the results do (tests/test_trajectory_gpu.py fails); they cost ~4 us of kernel time per launch.
__device__ __forceinline__ void pose_publish(const PoseOptArgs& a, const BaPose& pose, int nbad, int its, int trials) {
    if (a.done_seq) __threadfence_system();  // this thread's outlier flags
    __syncthreads();
    if (threadIdx.x == 0) {
        *a.pose_out = pose;
        a.info[0] = nbad;
        a.info[1] = its;
        a.info[2] = trials;
        if (a.done_seq) {
            __threadfence_system();
            __hip_atomic_store(&a.info[3], a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- LDS-resident version (n <= 3072 matched points) ----
// The kernel above keeps the stored errors and outlier flags in global memory and rebuilds the normal equations
// in a phase of their own; with ~25 LM trials per call every phase is a memory round trip plus barriers (18 us
// per trial measured).  Here the edges (inputs, stored errors, flags) live in LDS for the whole schedule, ONE
// pass over the edges yields the trial's chi2 AND its normal equations (a rejected trial wastes them, an accepted
// one - the common case - has the next iteration's system ready), the 29 block sums ride a transposed butterfly
// (32 shuffles per wave instead of 29 x 6), and lane 0 decides and solves for the next trial in one go: two
// barriers per trial.
constexpr int kPoLdsMax = kPoseOptLdsMax;

// tools/probe/pose_probe.hip defines SO_POSE_TICK to accumulate clock64() per phase; the product build compiles it away
#ifndef SO_POSE_TICK
#define SO_POSE_TICK(i)
#define SO_POSE_TICK_DECL
#define SO_POSE_TICK_FLUSH
#endif


// pose <- SE3Quat::exp(u) * pose for the LM steps of PoseOptimization, off the libm path: lane 0 evaluates this
// once per trial while 511 threads wait, so sin / cos / sqrt / the divisions of se3_exp_mul are replaced by the
// Taylor series of sin(t)/t, (1-cos t)/t^2, (t-sin t)/t^3 in t^2 (|t| < 0.25 rad: 8 terms reach double precision
// and avoid the cancellation of the closed forms) and by reciprocal-square-root normalisations.  Larger steps
// (never seen in tracking: 0.25 rad is 14 degrees between consecutive LM trials) take the closed forms.
__device__ void se3_exp_mul_small(const double* u, const BaPose& in, BaPose& out) {
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double t2 = w0 * w0 + w1 * w1 + w2 * w2;
    if (t2 > 0.0625) {
        se3_exp_mul(u, in, out);
        return;
    }
    double a = 1.0 / 1307674368000.0, b = 1.0 / 20922789888000.0, c = 1.0 / 355687428096000.0;  // 1/15!, 1/16!, 1/17!
    a = fma(-a, t2, 1.0 / 6227020800.0);  b = fma(-b, t2, 1.0 / 87178291200.0);  c = fma(-c, t2, 1.0 / 1307674368000.0);
    a = fma(-a, t2, 1.0 / 39916800.0);    b = fma(-b, t2, 1.0 / 479001600.0);    c = fma(-c, t2, 1.0 / 6227020800.0);
    a = fma(-a, t2, 1.0 / 362880.0);      b = fma(-b, t2, 1.0 / 3628800.0);      c = fma(-c, t2, 1.0 / 39916800.0);
    a = fma(-a, t2, 1.0 / 5040.0);        b = fma(-b, t2, 1.0 / 40320.0);        c = fma(-c, t2, 1.0 / 362880.0);
    a = fma(-a, t2, 1.0 / 120.0);         b = fma(-b, t2, 1.0 / 720.0);          c = fma(-c, t2, 1.0 / 5040.0);
    a = fma(-a, t2, 1.0 / 6.0);           b = fma(-b, t2, 1.0 / 24.0);           c = fma(-c, t2, 1.0 / 120.0);
    a = fma(-a, t2, 1.0);                 b = fma(-b, t2, 0.5);                  c = fma(-c, t2, 1.0 / 6.0);
    const double Om[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    double Om2[9], R[9], V[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Om2[i * 3 + j] = Om[i * 3] * Om[j] + Om[i * 3 + 1] * Om[3 + j] + Om[i * 3 + 2] * Om[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
        V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
    }
    // Quaterniond(R) for a rotation this close to the identity: trace > 0 branch
    double qa[4];
    {
        const double tr = R[0] + R[4] + R[8] + 1.0;  // = 4 w^2
        const double y = rsqrt_newton(tr);
        qa[3] = 0.5 * tr * y;
        const double h = 0.5 * y;
        qa[0] = (R[7] - R[5]) * h;
        qa[1] = (R[2] - R[6]) * h;
        qa[2] = (R[3] - R[1]) * h;
        const double n = rsqrt_newton(qa[0] * qa[0] + qa[1] * qa[1] + qa[2] * qa[2] + qa[3] * qa[3]);
        qa[0] *= n; qa[1] *= n; qa[2] *= n; qa[3] *= n;
    }
    double ta[3], rt[3];
#pragma unroll
    for (int i = 0; i < 3; i++) ta[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    quat_rotate(qa, in.t, rt);
    out.t[0] = ta[0] + rt[0]; out.t[1] = ta[1] + rt[1]; out.t[2] = ta[2] + rt[2];
    const double* q = in.q;
    double qn[4];
    qn[3] = qa[3] * q[3] - qa[0] * q[0] - qa[1] * q[1] - qa[2] * q[2];
    qn[0] = qa[3] * q[0] + qa[0] * q[3] + qa[1] * q[2] - qa[2] * q[1];
    qn[1] = qa[3] * q[1] + qa[1] * q[3] + qa[2] * q[0] - qa[0] * q[2];
    qn[2] = qa[3] * q[2] + qa[2] * q[3] + qa[0] * q[1] - qa[1] * q[0];
    if (qn[3] < 0) { qn[0] = -qn[0]; qn[1] = -qn[1]; qn[2] = -qn[2]; qn[3] = -qn[3]; }
    const double n = rsqrt_newton(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    out.q[0] = qn[0] * n; out.q[1] = qn[1] * n; out.q[2] = qn[2] * n; out.q[3] = qn[3] * n;
    out.pad = 0;
}

// 6x6 SPD solve (H + lambda I) x = b by Cholesky with reciprocal square roots; also computeScale.  Right-looking
// with the right-hand side carried as a seventh row: after column j only one reciprocal square root, one scale and
// one update lie on the dependent chain (about 9 operations per column instead of 2 j + 8), and the forward
// substitution has happened by the time the factor is complete.
__device__ void po_solve6(const double* H, const double* b, double lambda, double* x, double& scale, int& ok) {
    double A[42], ri[6];  // rows 0-5: the matrix (lower part used), row 6: the right-hand side
#pragma unroll
    for (int q = 0; q < 36; q++) A[q] = H[q];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        A[7 * j] += lambda;
        A[36 + j] = b[j];
    }
    ok = 1;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double dj = A[j * 6 + j];
        if (!(dj > 0.0)) ok = 0;
        const double y = rsqrt_newton(dj);
        ri[j] = y;
#pragma unroll
        for (int i = j + 1; i < 7; i++) A[i * 6 + j] *= y;
#pragma unroll
        for (int i = j + 1; i < 7; i++)
#pragma unroll
            for (int c = j + 1; c < 6; c++)
                if (c <= i) A[i * 6 + c] = fma(-A[i * 6 + j], A[c * 6 + j], A[i * 6 + c]);
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {  // L^T x = y, y = row 6
        double v = A[36 + i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) v = fma(-A[k * 6 + i], x[k], v);
        x[i] = v * ri[i];
    }
    scale = 0.0;
#pragma unroll
    for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
    if (!ok) {
#pragma unroll
        for (int j = 0; j < 6; j++) x[j] = 0.0;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void pose_opt_lds_kernel(PoseOptArgs a) {
    extern __shared__ double s_dyn[];
    constexpr int kPoThreads = THREADS;  // shadows the classic kernel's constant inside this kernel
    __shared__ double s_red[THREADS / 64][32];
    __shared__ double s_sys[2][48];  // [0] system of the current estimate, [1] of the trial: H 36 | b 6 | chi | n_active
    __shared__ BaPose s_cur, s_trial;
    __shared__ int s_ctrl[2];        // [0] 0 stop, 1 run the trial in s_trial, 2 re-evaluate the current estimate first
                                     // [1] which of s_sys holds the current estimate's system (the other takes the trial's)
    __shared__ int s_info[3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.n;
    SO_POSE_TICK_DECL;
    double* s_err = s_dyn;                                   // 2n
    float* s_X = reinterpret_cast<float*>(s_dyn + 2 * n);    // 3n
    float* s_obs = s_X + 3 * n;                              // 2n
    float* s_w = s_obs + 2 * n;                              // n
    uint8_t* s_out = reinterpret_cast<uint8_t*>(s_w + n);    // n
    const double delta = (double)sqrtf(5.991f);  // const float deltaMono = sqrt(5.991)
    const float dsqr = (float)(delta * delta);
    const double fx = a.K[0], fy = a.K[1], cx = a.K[2], cy = a.K[3];
    for (int e = tid; e < n; e += kPoThreads) {
        s_X[3 * e] = a.Xw[3 * e]; s_X[3 * e + 1] = a.Xw[3 * e + 1]; s_X[3 * e + 2] = a.Xw[3 * e + 2];
        s_obs[2 * e] = a.obs[2 * e]; s_obs[2 * e + 1] = a.obs[2 * e + 1];
        s_w[e] = a.inv_sigma2[e];
        s_out[e] = 0;
        s_err[2 * e] = 0.0; s_err[2 * e + 1] = 0.0;
    }
    if (tid == 0) { s_info[0] = 0; s_info[1] = 0; s_info[2] = 0; }
    __syncthreads();
    int robust = 1;

    // One pass over the active edges at pose T: stored errors, robustified chi2, J^T w J (upper 21), -rho' J^T Omega e
    // (6) and the number of active edges, reduced into s_sys[which].  One barrier; the caller's barrier after lane 0's
    // decision is the second.
    auto edge_phase = [&](const BaPose& T, int which) {
        SO_POSE_TICK(6);  // everything since the last tick: lane 0's decision + barrier, round set-up
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = 0.0;
        for (int e = tid; e < n; e += kPoThreads) {
            if (s_out[e]) continue;  // level 1
            const double X[3] = {(double)s_X[3 * e], (double)s_X[3 * e + 1], (double)s_X[3 * e + 2]};
            double pc[3];
            camera_point(T, X, pc);
            double invz = __builtin_amdgcn_rcp(pc[2]);  // reciprocal + two Newton steps instead of the IEEE division
            invz = fma(fma(-pc[2], invz, 1.0), invz, invz);
            invz = fma(fma(-pc[2], invz, 1.0), invz, invz);
            const double x = pc[0], y = pc[1], invz_2 = invz * invz;
            const double e0 = (double)s_obs[2 * e] - (x * invz * fx + cx);
            const double e1 = (double)s_obs[2 * e + 1] - (y * invz * fy + cy);
            s_err[2 * e] = e0;
            s_err[2 * e + 1] = e1;
            const double w = (double)s_w[e];
            const double chi2 = e0 * (w * e0) + e1 * (w * e1);
            // Huber through one reciprocal square root: rho = 2 sqrt(e) delta - delta^2, rho' = delta / sqrt(e)
            const bool clipped = robust && !(chi2 <= (double)dsqr);
            const double rs = clipped ? rsqrt_newton(chi2) : 0.0;
            v[27] += clipped ? 2.0 * (chi2 * rs) * delta - (double)dsqr : chi2;
            v[28] += 1.0;
            double J[12];
            J[0] = x * y * invz_2 * fx;       J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx;
            J[3] = -invz * fx;                J[4] = 0;                            J[5] = x * invz_2 * fx;
            J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy;         J[8] = -x * invz * fy;
            J[9] = 0;                         J[10] = -invz * fy;                  J[11] = y * invz_2 * fy;
            const double r1 = robust ? huber_rho1(chi2, delta, dsqr) : 1.0;
            const double wo = r1 * w;
            // J[4] and J[9] are structurally zero: their products are left out (same sums, fewer FMAs)
            double Ju[6], Jv[6];
#pragma unroll
            for (int r = 0; r < 6; r++) { Ju[r] = J[r] * wo; Jv[r] = J[6 + r] * wo; }
            int t = 0;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = r; c < 6; c++) {
                    double acc = v[t];
                    if (r != 4 && c != 4) acc = fma(Ju[r], J[c], acc);
                    if (r != 3 && c != 3) acc = fma(Jv[r], J[6 + c], acc);
                    v[t++] = acc;
                }
            const double we0 = r1 * (w * e0), we1 = r1 * (w * e1);
#pragma unroll
            for (int r = 0; r < 6; r++) {
                double acc = v[21 + r];
                if (r != 4) acc = fma(-J[r], we0, acc);
                if (r != 3) acc = fma(-J[6 + r], we1, acc);
                v[21 + r] = acc;
            }
        }
        SO_POSE_TICK(0);  // edge loop
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = po_swap_add32(v[i], v[i + 16]);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = po_swap_add16(v[i], v[i + 8]);
        po_halve<4>(v, 8, lane);
        po_halve<2>(v, 4, lane);
        po_halve<1>(v, 2, lane);
        const double tot = v[0] + __shfl_xor(v[0], 1);  // lane holds the wave total of value (lane >> 1)
        if ((lane & 1) == 0) s_red[wv][lane >> 1] = tot;
        SO_POSE_TICK(1);  // wave reduction
        __syncthreads();
        SO_POSE_TICK(2);  // barrier
        if (tid < 29) {
            double sum = s_red[0][tid];
#pragma unroll
            for (int w2 = 1; w2 < THREADS / 64; w2++) sum += s_red[w2][tid];
            double* S = s_sys[which];
            if (tid < 21) {
                int r = 0, t = tid;  // unrank the upper-triangular index
                while (t >= 6 - r) { t -= 6 - r; r++; }
                const int c = r + t;
                S[r * 6 + c] = sum;
                S[c * 6 + r] = sum;
            } else {
                S[36 + (tid - 21)] = sum;  // b 36..41, chi 42, n_active 43
            }
        }
        // no barrier here: only lane 0 (same wave as the 29 summing lanes, LDS in order) reads s_sys before the
        // barrier that ends lane 0's decision
        __threadfence_block();
        SO_POSE_TICK(3);  // cross-wave sum into s_sys
    };

    // lane 0: solve the current system with the current lambda and publish the trial pose
    double lambda = 0.0, ni = 2.0, currentChi = 0.0, iniChi = 0.0, scale = 1.0;
    int qmax = 0, it = 0, nBadLM = 0, solve_ok = 1, its_total = 0, trials_total = 0;
    auto propose = [&]() {
        double x[6];
        po_solve6(s_sys[s_ctrl[1]], s_sys[s_ctrl[1]] + 36, lambda, x, scale, solve_ok);
        scale += 1e-3;
        BaPose tr;
        se3_exp_mul_small(x, s_cur, tr);
        s_trial = tr;
        s_ctrl[0] = 1;
    };

    for (int round = 0; round < 4; round++) {
        if (tid == 0) {
            s_cur = a.init;  // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw))
            s_ctrl[1] = 0;
        }
        __syncthreads();
        {
            const BaPose cur = s_cur;
            edge_phase(cur, 0);
        }
        if (tid == 0) {
            if (s_sys[0][43] > 0.0) {  // optimize(10) with at least one active edge
                double md = 0.0;
                for (int j = 0; j < 6; j++) md = fmax(md, fabs(s_sys[0][7 * j]));
                lambda = 1e-5 * md;  // computeLambdaInit
                ni = 2.0;
                currentChi = iniChi = s_sys[0][42];
                qmax = 0; it = 0; nBadLM = 0;
                propose();
            } else {
                s_ctrl[0] = 0;
            }
        }
        __syncthreads();
        while (s_ctrl[0] != 0) {
            if (s_ctrl[0] == 2) {  // a trial was neither accepted nor retried: stored errors back to the estimate
                const BaPose cur = s_cur;
                edge_phase(cur, s_ctrl[1]);
                if (tid == 0) propose();
                __syncthreads();
                continue;
            }
            const BaPose trial = s_trial;
            const int tsys = s_ctrl[1] ^ 1;
            edge_phase(trial, tsys);
            if (tid == 0) {
                double tempChi = solve_ok ? s_sys[tsys][42] : 1.7976931348623157e308;
                const double rho = (currentChi - tempChi) / scale;
                if (a.trace && trials_total < 256) {
                    a.trace[4 * trials_total] = lambda;
                    a.trace[4 * trials_total + 1] = tempChi;
                    a.trace[4 * trials_total + 2] = rho;
                    a.trace[4 * trials_total + 3] = currentChi;
                }
                const bool accepted = rho > 0 && isfinite(tempChi);
                if (accepted) {
                    const double tr = 2 * rho - 1;
                    double alpha = 1. - tr * tr * tr;
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2.0;
                    currentChi = tempChi;
                    s_cur = s_trial;     // discardTop;
                    s_ctrl[1] = tsys;    // the trial's system becomes the current one
                } else {
                    lambda *= ni;  // pop
                    ni *= 2.0;
                }
                qmax++;
                trials_total++;
                if (rho < 0 && qmax < 10) {
                    propose();  // same system, larger lambda
                } else {
                    its_total++;
                    bool stop = false;
                    if (qmax == 10 || rho == 0) {
                        stop = true;
                    } else {
                        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
                        if (nBadLM >= 3) stop = true;
                    }
                    it++;
                    if (stop || it >= 10) {
                        s_ctrl[0] = 0;
                    } else {
                        qmax = 0;
                        iniChi = currentChi;
                        if (accepted) propose();
                        else s_ctrl[0] = 2;
                    }
                }
            }
            __syncthreads();
        }
        // classify (Optimizer.cc:357-380): outliers of the previous round get a fresh error, inliers keep the stored one
        const BaPose fin = s_cur;
        double bad_local = 0.0;
        for (int e = tid; e < n; e += kPoThreads) {
            double e0 = s_err[2 * e], e1 = s_err[2 * e + 1];
            if (s_out[e]) {
                const double X[3] = {(double)s_X[3 * e], (double)s_X[3 * e + 1], (double)s_X[3 * e + 2]};
                double pc[3];
                camera_point(fin, X, pc);
                e0 = (double)s_obs[2 * e] - (pc[0] / pc[2] * fx + cx);
                e1 = (double)s_obs[2 * e + 1] - (pc[1] / pc[2] * fy + cy);
                s_err[2 * e] = e0;
                s_err[2 * e + 1] = e1;
            }
            const double w = (double)s_w[e];
            const float chi2 = (float)(e0 * (w * e0) + e1 * (w * e1));
            const bool out = chi2 > 5.991f;
            s_out[e] = out ? 1 : 0;
            bad_local += out ? 1.0 : 0.0;
        }
        const double nbad = block_sum(bad_local, s_red[0]);  // s_red[0] has 32 slots, block_sum needs THREADS / 64
        if (tid == 0) s_info[0] = (int)nbad;
        if (round == 2) robust = 0;
        __syncthreads();
        if (n < 10) break;
    }
    SO_POSE_TICK(6);
    SO_POSE_TICK_FLUSH;
    for (int e = tid; e < n; e += kPoThreads) a.outlier[e] = s_out[e];
    pose_publish(a, s_cur, s_info[0], its_total, trials_total);
}

// ---- short-dependency-chain versions of the 6x6 solve and of the SE3 update for the register-resident kernel ----
// Every lane of pose_opt_reg_kernel runs the LM control flow redundantly, so what matters is the LENGTH of the
// dependent FP64 chain (a dependent DP instruction issues every ~16 cycles with one wave per SIMD), not the
// operation count.  Cholesky has ~75 dependent steps for a 6x6 system; the block form below has ~40.
__device__ __forceinline__ double po_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}

// adjugate of a symmetric 3x3 (m00 m01 m02 m11 m12 m22) and its determinant; pd = all leading minors positive
__device__ __forceinline__ void po_adj3(const double* m, double* c, double& det, bool& pd) {
    c[0] = fma(m[3], m[5], -(m[4] * m[4]));
    c[1] = fma(m[2], m[4], -(m[1] * m[5]));
    c[2] = fma(m[1], m[4], -(m[2] * m[3]));
    c[3] = fma(m[0], m[5], -(m[2] * m[2]));
    c[4] = fma(m[1], m[2], -(m[0] * m[4]));
    c[5] = fma(m[0], m[3], -(m[1] * m[1]));
    det = fma(m[0], c[0], fma(m[1], c[1], m[2] * c[2]));
    pd = m[0] > 0.0 && c[5] > 0.0 && det > 0.0;
}

// (H + lambda I) x = b for the 6x6 system of PoseOptimization through its 3x3 blocks [[A B],[B^T C]]:
// A^-1 by adjugate, S = C - B^T A^-1 B, x2 = S^-1 (b2 - B^T A^-1 b1), x1 = A^-1 b1 - A^-1 B x2; also computeScale.
// S: 21 upper-triangular entries, row-major.  ok = 0 when A or S is not positive definite (then H + lambda I is not).
__device__ __forceinline__ void po_solve6_blocks(const double* S, const double* b, double lambda, double* x, double& scale,
                                                 int& ok) {
    const double A[6] = {S[0] + lambda, S[1], S[2], S[6] + lambda, S[7], S[11] + lambda};
    const double B[9] = {S[3], S[4], S[5], S[8], S[9], S[10], S[12], S[13], S[14]};  // rows 0-2, columns 3-5
    const double C[6] = {S[15] + lambda, S[16], S[17], S[18] + lambda, S[19], S[20] + lambda};
    double aA[6], detA;
    bool pdA;
    po_adj3(A, aA, detA, pdA);
    const double iA = po_rcp(detA);
    // full symmetric adjugate rows
    const double a00 = aA[0], a01 = aA[1], a02 = aA[2], a11 = aA[3], a12 = aA[4], a22 = aA[5];
    double Y[9];  // adj(A) B (unscaled)
#pragma unroll
    for (int j = 0; j < 3; j++) {
        Y[0 + j] = fma(a00, B[j], fma(a01, B[3 + j], a02 * B[6 + j]));
        Y[3 + j] = fma(a01, B[j], fma(a11, B[3 + j], a12 * B[6 + j]));
        Y[6 + j] = fma(a02, B[j], fma(a12, B[3 + j], a22 * B[6 + j]));
    }
    const double y1[3] = {fma(a00, b[0], fma(a01, b[1], a02 * b[2])), fma(a01, b[0], fma(a11, b[1], a12 * b[2])),
                          fma(a02, b[0], fma(a12, b[1], a22 * b[2]))};  // adj(A) b1 (unscaled)
    double Sc[6], r2[3];
    {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = i; j < 3; j++) {
                const double d = fma(B[i], Y[j], fma(B[3 + i], Y[3 + j], B[6 + i] * Y[6 + j]));  // (B^T adj(A) B)_ij
                Sc[t] = fma(-d, iA, C[t]);
                t++;
            }
            const double e = fma(B[i], y1[0], fma(B[3 + i], y1[1], B[6 + i] * y1[2]));
            r2[i] = fma(-e, iA, b[3 + i]);
        }
    }
    double aS[6], detS;
    bool pdS;
    po_adj3(Sc, aS, detS, pdS);
    const double iS = po_rcp(detS);
    x[3] = fma(aS[0], r2[0], fma(aS[1], r2[1], aS[2] * r2[2])) * iS;
    x[4] = fma(aS[1], r2[0], fma(aS[3], r2[1], aS[4] * r2[2])) * iS;
    x[5] = fma(aS[2], r2[0], fma(aS[4], r2[1], aS[5] * r2[2])) * iS;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double d = fma(Y[3 * i], x[3], fma(Y[3 * i + 1], x[4], Y[3 * i + 2] * x[5]));
        x[i] = (y1[i] - d) * iA;
    }
    ok = (pdA && pdS) ? 1 : 0;
    double sc = 0.0;
#pragma unroll
    for (int j = 0; j < 6; j++) sc += x[j] * fma(lambda, x[j], b[j]);
    scale = sc;
    if (!ok) {
#pragma unroll
        for (int j = 0; j < 6; j++) x[j] = 0.0;
    }
}

// pose <- SE3Quat::exp(u) * pose with the unit quaternion of exp(omega) written down directly
// ((sin(t/2)/t) omega, cos(t/2), series in t^2) instead of Rodrigues' matrix turned back into a quaternion
// (code/Thirdparty/g2o/g2o/types/se3quat.h:223-257): same value up to rounding, half the dependent chain.
// Steps beyond 0.25 rad take the closed forms of se3_exp_mul.
__device__ __forceinline__ void se3_exp_mul_direct(const double* u, const BaPose& in, BaPose& out) {
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double t2 = fma(w0, w0, fma(w1, w1, w2 * w2));
    if (t2 > 0.0625) {
        se3_exp_mul(u, in, out);
        return;
    }
    const double h = 0.25 * t2;  // (theta / 2)^2
    // sin(theta/2)/theta = 1/2 (1 - h/3! + h^2/5! - ...), cos(theta/2) = 1 - h/2! + h^2/4! - ...
    double sn = -1.0 / 6227020800.0, cs = -1.0 / 87178291200.0;  // -1/13!, -1/14!
    sn = fma(sn, h, 1.0 / 39916800.0);   cs = fma(cs, h, 1.0 / 479001600.0);
    sn = fma(sn, h, -1.0 / 362880.0);    cs = fma(cs, h, -1.0 / 3628800.0);
    sn = fma(sn, h, 1.0 / 5040.0);       cs = fma(cs, h, 1.0 / 40320.0);
    sn = fma(sn, h, -1.0 / 120.0);       cs = fma(cs, h, -1.0 / 720.0);
    sn = fma(sn, h, 1.0 / 6.0);          cs = fma(cs, h, 1.0 / 24.0);
    sn = fma(-sn, h, 1.0);               cs = fma(cs, h, -0.5);
    sn = 0.5 * sn;                       cs = fma(cs, h, 1.0);
    // V = I + b Omega + c Omega^2, b = (1 - cos t)/t^2, c = (t - sin t)/t^3
    double b = 1.0 / 20922789888000.0, c = 1.0 / 355687428096000.0;  // 1/16!, 1/17!
    b = fma(-b, t2, 1.0 / 87178291200.0);  c = fma(-c, t2, 1.0 / 1307674368000.0);
    b = fma(-b, t2, 1.0 / 479001600.0);    c = fma(-c, t2, 1.0 / 6227020800.0);
    b = fma(-b, t2, 1.0 / 3628800.0);      c = fma(-c, t2, 1.0 / 39916800.0);
    b = fma(-b, t2, 1.0 / 40320.0);        c = fma(-c, t2, 1.0 / 362880.0);
    b = fma(-b, t2, 1.0 / 720.0);          c = fma(-c, t2, 1.0 / 5040.0);
    b = fma(-b, t2, 1.0 / 24.0);           c = fma(-c, t2, 1.0 / 120.0);
    b = fma(-b, t2, 0.5);                  c = fma(-c, t2, 1.0 / 6.0);
    const double v0 = u[3], v1 = u[4], v2 = u[5];
    const double k0 = fma(w1, v2, -(w2 * v1)), k1 = fma(w2, v0, -(w0 * v2)), k2 = fma(w0, v1, -(w1 * v0));  // omega x upsilon
    const double m0 = fma(w1, k2, -(w2 * k1)), m1 = fma(w2, k0, -(w0 * k2)), m2 = fma(w0, k1, -(w1 * k0));  // omega x (omega x upsilon)
    const double ta[3] = {fma(c, m0, fma(b, k0, v0)), fma(c, m1, fma(b, k1, v1)), fma(c, m2, fma(b, k2, v2))};
    const double qa[4] = {sn * w0, sn * w1, sn * w2, cs};
    double rt[3];
    quat_rotate(qa, in.t, rt);
    out.t[0] = ta[0] + rt[0]; out.t[1] = ta[1] + rt[1]; out.t[2] = ta[2] + rt[2];
    const double* q = in.q;
    double qn[4];
    qn[3] = fma(qa[3], q[3], -fma(qa[0], q[0], fma(qa[1], q[1], qa[2] * q[2])));
    qn[0] = fma(qa[3], q[0], fma(qa[0], q[3], fma(qa[1], q[2], -(qa[2] * q[1]))));
    qn[1] = fma(qa[3], q[1], fma(qa[1], q[3], fma(qa[2], q[0], -(qa[0] * q[2]))));
    qn[2] = fma(qa[3], q[2], fma(qa[2], q[3], fma(qa[0], q[1], -(qa[1] * q[0]))));
    double nrm = rsqrt_newton(fma(qn[0], qn[0], fma(qn[1], qn[1], fma(qn[2], qn[2], qn[3] * qn[3]))));
    if (qn[3] < 0) nrm = -nrm;
    out.q[0] = qn[0] * nrm; out.q[1] = qn[1] * nrm; out.q[2] = qn[2] * nrm; out.q[3] = qn[3] * nrm;
    out.pad = 0;
}

// ---- register-resident version (n <= THREADS x EPT matched points) ----
// What bounded the LDS-resident kernel above (tools/probe/pose_probe.hip): a dependent FP64 chain per edge with one
// wave per SIMD, and 3.7 k cycles per pass on lane 0 (decision + 6x6 solve + SE3 exponential) fenced by two barriers
// while the other 255 threads wait.  Here
//   * every thread keeps its EPT edges (point, observation, weight, stored error, outlier flag) in registers for the
//     whole schedule - no LDS traffic and no float -> double conversion per pass - and walks them unrolled, so the
//     scheduler interleaves the independent chains of different edges;
//   * the rotation matrix is formed once per pass instead of a quaternion rotation per edge, the Jacobian rows share
//     their sub-products, Huber's rho and rho' come from ONE reciprocal square root (no division);
//   * the LM control flow is replicated: after the single barrier of a pass every wave sums the per-wave partials
//     itself (fixed order: identical in all waves) and EVERY lane decides, solves and exponentiates redundantly in
//     registers - the SIMD lanes were idle anyway - so the trial pose never travels through LDS and the second
//     barrier of a pass disappears.  The partial-sum buffer is double-buffered by pass parity: a wave can be at most
//     one pass ahead of the slowest one.
// Same schedule, same accept / reject rules, same stored-error semantics as above (g2o's optimize(10) x 4).
// float -> double where the value is USED: an opaque conversion (the compiler would otherwise widen a loop-invariant float once
// and keep the double in registers for the whole schedule, which is what the float-resident inputs are there to avoid)
__device__ __forceinline__ double pose_widen(double v) { return v; }
__device__ __forceinline__ double pose_widen(float v) {
    double d;
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(v));
    return d;
}

template <int THREADS, int EPT>
__device__ __forceinline__ void pose_opt_reg_body(const PoseOptArgs& a, const int n) {
    constexpr int NW = THREADS / 64;
    __shared__ double s_red[2][NW][32];
    __shared__ double s_sysw[NW][2][32];  // per wave: the two systems; 21 H (upper) | 6 b | chi | n_active
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double delta = (double)sqrtf(5.991f);  // const float deltaMono = sqrt(5.991)
    const float dsqr = (float)(delta * delta);
    const double fx = a.K[0], fy = a.K[1], cx = a.K[2], cy = a.K[3];
    // The inputs are float32 at the ABI: from eight edges per thread on they stay float in the registers (6 instead of 12 per
    // edge) and are widened where they are used - the same doubles, so the same bits; below eight the widened copies fit.
    using In = typename std::conditional<(EPT >= 8), float, double>::type;
    In X[EPT][3], ob[EPT][2], w[EPT];
    double er[EPT][2];
    bool live[EPT], outl[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int e = tid + k * THREADS;
        live[k] = e < n;
        outl[k] = false;
        const int ee = live[k] ? e : 0;
        if (a.e_kp) {  // indexed: the frame's keypoint and the map table's row, in place
            const int kp = a.e_kp[ee], sl = a.e_slot[ee];
            X[k][0] = (In)a.map_Xw[3 * (size_t)sl]; X[k][1] = (In)a.map_Xw[3 * (size_t)sl + 1]; X[k][2] = (In)a.map_Xw[3 * (size_t)sl + 2];
            const float2 o = a.kp_xy_un[kp];
            ob[k][0] = (In)o.x; ob[k][1] = (In)o.y;
            const int oc = a.kp_octave[kp];
            float ws = a.lvl_inv_sigma2[0];  // (a select chain: a run-time index into the argument block would go through scratch)
#pragma unroll
            for (int l = 1; l < 8; l++) ws = oc == l ? a.lvl_inv_sigma2[l] : ws;
            w[k] = (In)ws;
        } else {
            X[k][0] = (In)a.Xw[3 * ee]; X[k][1] = (In)a.Xw[3 * ee + 1]; X[k][2] = (In)a.Xw[3 * ee + 2];
            ob[k][0] = (In)a.obs[2 * ee]; ob[k][1] = (In)a.obs[2 * ee + 1];
            w[k] = (In)a.inv_sigma2[ee];
        }
        er[k][0] = 0.0; er[k][1] = 0.0;
    }
    int robust = 1, parity = 0;
    double pass_chi = 0.0;  // chi2 total of the last edge pass
    SO_POSE_TICK_DECL;

    // One pass over the active edges at pose T; the block totals land in this wave's s_sysw[wv][which].
    auto edge_phase = [&](const BaPose& T, int which) __attribute__((always_inline)) {
        SO_POSE_TICK(6);  // everything since the last tick: decision + solve + SE3 update, round set-up
        double R[9];
        quat_to_R(T.q, R);
        double v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            // Branch-free: an edge that is not active (beyond n, or level 1) runs the same instructions on harmless
            // values and enters every sum with weight 0.  With `if (!active) continue` every edge was a basic block of
            // its own behind an exec-mask change, and the scheduler could not interleave the edges' dependent chains.
            const bool act = live[k] && !outl[k];
            const double X0 = pose_widen(X[k][0]), X1 = pose_widen(X[k][1]), X2 = pose_widen(X[k][2]);
            const double x = fma(R[0], X0, fma(R[1], X1, fma(R[2], X2, T.t[0])));
            const double y = fma(R[3], X0, fma(R[4], X1, fma(R[5], X2, T.t[1])));
            const double zr = fma(R[6], X0, fma(R[7], X1, fma(R[8], X2, T.t[2])));
            const double z = act ? zr : 1.0;
            double c = __builtin_amdgcn_rcp(z);  // reciprocal + two Newton steps instead of the IEEE division
            c = fma(fma(-z, c, 1.0), c, c);
            c = fma(fma(-z, c, 1.0), c, c);
            const double p = x * c, q = y * c;  // normalised image coordinates
            const double pf = p * fx, qg = q * fy;
            const double e0 = pose_widen(ob[k][0]) - (pf + cx);
            const double e1 = pose_widen(ob[k][1]) - (qg + cy);
            er[k][0] = act ? e0 : er[k][0];
            er[k][1] = act ? e1 : er[k][1];
            const double wk = act ? pose_widen(w[k]) : 0.0;
            const double we0 = wk * e0, we1 = wk * e1;
            const double chi2 = fma(e0, we0, e1 * we1);
            // Huber: rho = 2 sqrt(e) delta - delta^2, rho' = delta / sqrt(e), both from one reciprocal square root
            const bool clipped = robust && !(chi2 <= (double)dsqr);
            const double rs = rsqrt_newton(clipped ? chi2 : 1.0);
            v[27] += clipped ? fma(2.0 * delta, chi2 * rs, -(double)dsqr) : chi2;
            v[28] += act ? 1.0 : 0.0;
            const double r1 = clipped ? delta * rs : 1.0;
            const double wo = r1 * wk;
            // EdgeSE3ProjectXYZOnlyPose::linearizeOplus (types_six_dof_expmap.cpp:266-288), shared sub-products;
            // Ju[4] and Jv[3] are structurally zero
            const double cf = c * fx, cg = c * fy;
            const double Ju0 = pf * q, Ju1 = -fma(pf, p, fx), Ju2 = q * fx, Ju3 = -cf, Ju5 = pf * c;
            const double Jv0 = fma(qg, q, fy), Jv1 = -(qg * p), Jv2 = -(p * fy), Jv4 = -cg, Jv5 = qg * c;
            const double a0 = Ju0 * wo, a1 = Ju1 * wo, a2 = Ju2 * wo, a3 = Ju3 * wo, a5 = Ju5 * wo;
            const double b0 = Jv0 * wo, b1 = Jv1 * wo, b2 = Jv2 * wo, b4 = Jv4 * wo, b5 = Jv5 * wo;
            // upper triangle, row-major: (0,0..5) (1,1..5) (2,2..5) (3,3..5) (4,4..5) (5,5)
            v[0] = fma(a0, Ju0, fma(b0, Jv0, v[0]));
            v[1] = fma(a0, Ju1, fma(b0, Jv1, v[1]));
            v[2] = fma(a0, Ju2, fma(b0, Jv2, v[2]));
            v[3] = fma(a0, Ju3, v[3]);
            v[4] = fma(b0, Jv4, v[4]);
            v[5] = fma(a0, Ju5, fma(b0, Jv5, v[5]));
            v[6] = fma(a1, Ju1, fma(b1, Jv1, v[6]));
            v[7] = fma(a1, Ju2, fma(b1, Jv2, v[7]));
            v[8] = fma(a1, Ju3, v[8]);
            v[9] = fma(b1, Jv4, v[9]);
            v[10] = fma(a1, Ju5, fma(b1, Jv5, v[10]));
            v[11] = fma(a2, Ju2, fma(b2, Jv2, v[11]));
            v[12] = fma(a2, Ju3, v[12]);
            v[13] = fma(b2, Jv4, v[13]);
            v[14] = fma(a2, Ju5, fma(b2, Jv5, v[14]));
            v[15] = fma(a3, Ju3, v[15]);
            // (3,4) is structurally zero: v[16] stays 0
            v[17] = fma(a3, Ju5, v[17]);
            v[18] = fma(b4, Jv4, v[18]);
            v[19] = fma(b4, Jv5, v[19]);
            v[20] = fma(a5, Ju5, fma(b5, Jv5, v[20]));
            const double g0 = r1 * we0, g1 = r1 * we1;  // -rho' J^T Omega e
            v[21] = fma(-Ju0, g0, fma(-Jv0, g1, v[21]));
            v[22] = fma(-Ju1, g0, fma(-Jv1, g1, v[22]));
            v[23] = fma(-Ju2, g0, fma(-Jv2, g1, v[23]));
            v[24] = fma(-Ju3, g0, v[24]);
            v[25] = fma(-Jv4, g1, v[25]);
            v[26] = fma(-Ju5, g0, fma(-Jv5, g1, v[26]));
        }
        SO_POSE_TICK(0);  // edge loop
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = po_swap_add32(v[i], v[i + 16]);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = po_swap_add16(v[i], v[i + 8]);
        po_halve<4>(v, 8, lane);
        po_halve<2>(v, 4, lane);
        po_halve<1>(v, 2, lane);
        const double tot = v[0] + __shfl_xor(v[0], 1);  // lane holds the wave total of value (lane >> 1)
        if ((lane & 1) == 0) s_red[parity][wv][lane >> 1] = tot;
        SO_POSE_TICK(1);  // wave reduction
        __syncthreads();  // the only barrier of a pass
        SO_POSE_TICK(2);  // barrier
        {
            double sum = s_red[parity][0][lane & 31];
#pragma unroll
            for (int w2 = 1; w2 < NW; w2++) sum += s_red[parity][w2][lane & 31];
            if (lane < 32) s_sysw[wv][which][lane] = sum;
            // chi2 of this pass straight from lane 27's register: the decision does not wait for the LDS round trip
            pass_chi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sum), 27), __builtin_amdgcn_readlane(__double2loint(sum), 27));
        }
        parity ^= 1;
        __threadfence_block();  // wave-private rows: in-order LDS traffic of this wave + completed writes are enough
        SO_POSE_TICK(3);  // cross-wave sum
    };

    // replicated LM state (identical in every lane of every wave)
    BaPose cur = a.init, trial = a.init;
    double lambda = 0.0, ni = 2.0, currentChi = 0.0, iniChi = 0.0, scale = 1.0, inv_scale = 1.0;
    int qmax = 0, it = 0, nBadLM = 0, solve_ok = 1, its_total = 0, trials_total = 0, cursys = 0, nbad_total = 0;
    auto propose = [&]() __attribute__((always_inline)) {
        SO_POSE_TICK(4);  // decision (since the cross-wave sum)
        const double* S = s_sysw[wv][cursys];
        double Su[21], b[6], x[6];
#pragma unroll
        for (int r = 0; r < 21; r++) Su[r] = S[r];
#pragma unroll
        for (int r = 0; r < 6; r++) b[r] = S[21 + r];
        po_solve6_blocks(Su, b, lambda, x, scale, solve_ok);
        inv_scale = po_rcp(scale + 1e-3);  // ready long before the trial's chi2 arrives
        SO_POSE_TICK(5);  // 6x6 solve
        se3_exp_mul_direct(x, cur, trial);
        SO_POSE_TICK(7);  // SE3 update

    };

    // One instance of the edge pass, of the decision and of the proposal in the instruction stream (a small state
    // machine instead of inlining them at every place of g2o's control flow): the unrolled kernel was ~60 KB of code and
    // every pass paid instruction-cache misses between its pieces (1.2 k cycles from the end of the proposal to the
    // first edge, tools/probe/pose_probe.hip).
    enum { kFirst = 0, kTrial = 1, kReeval = 2 };
    for (int round = 0; round < 4; round++) {
        cur = a.init;  // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw))
        cursys = 0;
        int mode = kFirst, which = 0;
        BaPose eval = cur;
        for (;;) {
            edge_phase(eval, which);
            bool stop = false, want_proposal = false;
            if (mode == kFirst) {
                const double* S = s_sysw[wv][0];
                if (S[28] > 0.0) {  // optimize(10) with at least one active edge
                    const double md = fmax(fmax(fmax(fabs(S[0]), fabs(S[6])), fmax(fabs(S[11]), fabs(S[15]))), fmax(fabs(S[18]), fabs(S[20])));
                    lambda = 1e-5 * md;  // computeLambdaInit
                    ni = 2.0;
                    currentChi = iniChi = S[27];
                    qmax = 0; it = 0; nBadLM = 0;
                    want_proposal = true;
                } else {
                    stop = true;
                }
            } else if (mode == kReeval) {  // stored errors are back at the estimate: next iteration's first trial
                want_proposal = true;
            } else {
                const int tsys = cursys ^ 1;
                const double tempChi = solve_ok ? pass_chi : 1.7976931348623157e308;
                const double rho = (currentChi - tempChi) * inv_scale;
                if (a.trace && tid == 0 && trials_total < 256) {
                    a.trace[4 * trials_total] = lambda;
                    a.trace[4 * trials_total + 1] = tempChi;
                    a.trace[4 * trials_total + 2] = rho;
                    a.trace[4 * trials_total + 3] = currentChi;
                }
                const bool accepted = rho > 0 && isfinite(tempChi);
                if (accepted) {
                    const double tr = 2 * rho - 1;
                    double alpha = 1. - tr * tr * tr;
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2.0;
                    currentChi = tempChi;
                    cur = trial;     // discardTop
                    cursys = tsys;   // the trial's system becomes the current one
                } else {
                    lambda *= ni;  // pop
                    ni *= 2.0;
                }
                qmax++;
                trials_total++;
                if (rho < 0 && qmax < 10) {
                    want_proposal = true;  // same system, larger lambda
                } else {
                    its_total++;
                    bool end_it = false;
                    if (qmax == 10 || rho == 0) {
                        end_it = true;
                    } else {
                        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
                        if (nBadLM >= 3) end_it = true;
                    }
                    it++;
                    if (end_it || it >= 10) {
                        stop = true;
                    } else {
                        qmax = 0;
                        iniChi = currentChi;
                        want_proposal = accepted;  // a rejected last trial: re-evaluate the estimate first
                    }
                }
            }
            if (stop) break;
            if (want_proposal) {
                propose();
                mode = kTrial;
                eval = trial;
                which = cursys ^ 1;
            } else {
                mode = kReeval;
                eval = cur;
                which = cursys;
            }
        }
        // classify (Optimizer.cc:357-380): outliers of the previous round get a fresh error, inliers keep the stored one
        double R[9];
        quat_to_R(cur.q, R);
        double bad_local = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            if (!live[k]) continue;
            if (outl[k]) {
                const double X0 = pose_widen(X[k][0]), X1 = pose_widen(X[k][1]), X2 = pose_widen(X[k][2]);
                const double x = fma(R[0], X0, fma(R[1], X1, fma(R[2], X2, cur.t[0])));
                const double y = fma(R[3], X0, fma(R[4], X1, fma(R[5], X2, cur.t[1])));
                const double z = fma(R[6], X0, fma(R[7], X1, fma(R[8], X2, cur.t[2])));
                er[k][0] = pose_widen(ob[k][0]) - (x / z * fx + cx);
                er[k][1] = pose_widen(ob[k][1]) - (y / z * fy + cy);
            }
            const double wd = pose_widen(w[k]);
            const float chi2 = (float)(er[k][0] * (wd * er[k][0]) + er[k][1] * (wd * er[k][1]));
            outl[k] = chi2 > 5.991f;
            bad_local += outl[k] ? 1.0 : 0.0;
        }
        // block total in the same replicated fashion (one barrier): every wave ends up with the count
        {
            const double wsum = wave_sum(bad_local);
            if (lane == 0) s_red[parity][wv][0] = wsum;
            __syncthreads();
            double tot = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) tot += s_red[parity][w2][0];
            parity ^= 1;
            nbad_total = (int)tot;
        }
        if (round == 2) robust = 0;
        if (n < 10) break;
    }
    SO_POSE_TICK(6);
    SO_POSE_TICK_FLUSH;
    // results: the outlier flags meet in LDS and wave 0 writes everything (flags as dwords, pose, counters), so that ONE
    // wave pays ONE system-scope fence in front of the completion word instead of every wave one plus thread 0 a second
    __shared__ uint32_t s_outl[THREADS * EPT / 4];
    uint8_t* s_outl8 = reinterpret_cast<uint8_t*>(s_outl);
#pragma unroll
    for (int k = 0; k < EPT; k++)
        if (live[k]) {
            s_outl8[tid + k * THREADS] = outl[k] ? 1 : 0;
            if (a.kp_slot_clear && outl[k]) a.kp_slot_clear[a.e_kp[tid + k * THREADS]] = -1;
        }
    __syncthreads();
    if (tid < 64) {
        const int full = n >> 2;  // a.outlier is 16-byte aligned in every caller (offset 80 of a 64-byte aligned block)
        uint32_t* out32 = reinterpret_cast<uint32_t*>(a.outlier);
        for (int i = tid; i < full; i += 64) out32[i] = s_outl[i];
        if (tid < (n & 3)) a.outlier[4 * full + tid] = s_outl8[4 * full + tid];
        if (tid == 0) {
            *a.pose_out = cur;
            a.info[0] = nbad_total;
            a.info[1] = its_total;
            a.info[2] = trials_total;
        }
        if (a.done_seq) {
            __threadfence_system();  // (needed: see pose_publish)
            if (tid == 0) __hip_atomic_store(&a.info[3], a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int THREADS, int EPT>
__global__ __launch_bounds__(THREADS) void pose_opt_reg_kernel(PoseOptArgs a0, const PoseOptArgs* __restrict__ batch) {
    // batch != null: workgroup b solves problem batch[b] (so_pose_optimization_batch: several agents' frames, or the
    // independent problems of one frame, in ONE launch - a workgroup per problem, nothing shared between them)
    const PoseOptArgs a = batch ? batch[blockIdx.x] : a0;
    pose_opt_reg_body<THREADS, EPT>(a, a.n);
}

// The PoseOptimization call of a tracking stage whose search was resolved on the device: the number of edges is a device
// word (TrackResolveArgs::head), so the body that fits it is chosen here instead of at the launch.  Three bodies per
// kernel: with four and more in one function the compiler no longer keeps the 256-byte argument block out of scratch
// memory.  RANGE 0: up to 1024 edges (2 / 3 / 4 per thread), RANGE 1: 1025 .. 1792 (5 / 6 / 7); the host launches the range
// the stage's last call needed, and a count outside it is answered with info[0] = -1 (the caller takes the host path).
// other_range_launched (grouped launches): the group also launches the other RANGE, which takes the counts this one does not
template <int RANGE>
__device__ __forceinline__ void pose_opt_chain_body(const PoseOptArgs& a0, bool other_range_launched) {
    const PoseOptArgs a = a0;
    const int n = a0.head[0];
    const bool fits = RANGE == 0 ? n <= 1024 : (n > 1024 && n <= kPoseChainMaxEdges);
    if (!fits && other_range_launched && a0.head[2] == 0 && n >= 3 && n <= kPoseChainMaxEdges) return;
    if (a0.head[2] != 0 || n < 3 || !fits) {
        // nothing optimised: the resolve gave up / the count is outside this kernel's range (-1), or fewer than three
        // edges (-2: Optimizer.cc:358-359 returns without touching the frame)
        if (threadIdx.x == 0) {
            *a.pose_out = a.init;
            a.info[0] = (a0.head[2] == 0 && n < 3) ? -2 : -1;
            a.info[1] = 0;
            a.info[2] = 0;
            if (a.done_seq) {
                __threadfence_system();
                __hip_atomic_store(&a.info[3], a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    const int ept = (n + 255) >> 8;
    if (RANGE == 0) {
        if (ept <= 2) pose_opt_reg_body<256, 2>(a, n);
        else if (ept == 3) pose_opt_reg_body<256, 3>(a, n);
        else pose_opt_reg_body<256, 4>(a, n);
    } else {
        if (ept == 5) pose_opt_reg_body<256, 5>(a, n);
        else if (ept == 6) pose_opt_reg_body<256, 6>(a, n);
        else pose_opt_reg_body<256, 7>(a, n);
    }
}

template <int RANGE>
__global__ __launch_bounds__(256) void pose_opt_chain_kernel(PoseOptArgs a0) {
    pose_opt_chain_body<RANGE>(a0, false);
}

// a GROUP of agents' PoseOptimization calls behind their resolves (so_track_group): workgroup x = row x of the table
template <int RANGE>
__global__ __launch_bounds__(256) void pose_opt_chain_group_kernel(const PoseOptArgs* __restrict__ tab, int other_range_launched) {
    const PoseOptArgs a0 = tab[blockIdx.x];
    pose_opt_chain_body<RANGE>(a0, other_range_launched != 0);
}

// range_mask: bit r = some member's stage needed RANGE r last time; both bits: two launches, each problem runs in one
void launch_pose_opt_chain_group(const PoseOptArgs* d_tab, int n, int range_mask, hipStream_t s) {
    if (n <= 0) return;
    const int both = (range_mask & 3) == 3 ? 1 : 0;
    if ((range_mask & 1) || !(range_mask & 2))
        hipLaunchKernelGGL(pose_opt_chain_group_kernel<0>, dim3(n), dim3(256), 0, s, d_tab, both);
    if (range_mask & 2) hipLaunchKernelGGL(pose_opt_chain_group_kernel<1>, dim3(n), dim3(256), 0, s, d_tab, both);
}

void launch_pose_opt_chain(const PoseOptArgs& a, int range, hipStream_t s) {
    if (range == 0) hipLaunchKernelGGL(pose_opt_chain_kernel<0>, dim3(1), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pose_opt_chain_kernel<1>, dim3(1), dim3(256), 0, s, a);
}

template <int THREADS, int EPT>
static void launch_pose_reg(const PoseOptArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((pose_opt_reg_kernel<THREADS, EPT>), dim3(1), dim3(THREADS), 0, s, a, (const PoseOptArgs*)nullptr);
}

template <int EPT>
static void launch_pose_reg_batch(const PoseOptArgs* d_args, int n_problems, hipStream_t s) {
    const PoseOptArgs none{};
    hipLaunchKernelGGL((pose_opt_reg_kernel<256, EPT>), dim3(n_problems), dim3(256), 0, s, none, d_args);
}

bool launch_pose_opt_batch(const PoseOptArgs* d_args, int n_problems, int max_n, hipStream_t s) {
    if (n_problems <= 0) return true;
    if (max_n > 3072) return false;  // the register-resident kernel holds 12 edges per thread
    switch ((max_n + 255) / 256) {
        // (up to 512 points: two edges per thread - pose_opt_chain_kernel has no one-edge body, and the single, the batched
        //  and the chained launch of a problem must add their sums in the same order: the lockstep / chain tests compare bits)
        case 0: case 1: case 2: launch_pose_reg_batch<2>(d_args, n_problems, s); break;
        case 3: launch_pose_reg_batch<3>(d_args, n_problems, s); break;
        case 4: launch_pose_reg_batch<4>(d_args, n_problems, s); break;
        case 5: launch_pose_reg_batch<5>(d_args, n_problems, s); break;
        case 6: launch_pose_reg_batch<6>(d_args, n_problems, s); break;
        case 7: launch_pose_reg_batch<7>(d_args, n_problems, s); break;
        case 8: launch_pose_reg_batch<8>(d_args, n_problems, s); break;
        case 9: case 10: launch_pose_reg_batch<10>(d_args, n_problems, s); break;
        case 11: launch_pose_reg_batch<11>(d_args, n_problems, s); break;
        default: launch_pose_reg_batch<12>(d_args, n_problems, s); break;
    }
    return true;
}

void launch_pose_opt(const PoseOptArgs& a, hipStream_t s) {
    static const bool classic = getenv("SWARMORB_POSE_CLASSIC") != nullptr;  // A/B switch for profiling
    static const bool lds_only = getenv("SWARMORB_POSE_LDS") != nullptr;    // A/B: the LDS-resident kernel for every size
    static const int reg_max = getenv("SWARMORB_POSE_REG_MAX") ? atoi(getenv("SWARMORB_POSE_REG_MAX")) : 3072;  // A/B: 1024 hands 1025..3072 points to the LDS-resident kernel
    if (!classic && !lds_only && a.n > 1024 && a.n <= reg_max && a.n <= 3072) {
        const int ept = (a.n + 255) / 256;
        if (ept == 5) launch_pose_reg<256, 5>(a, s);
        else if (ept == 6) launch_pose_reg<256, 6>(a, s);
        else if (ept == 7) launch_pose_reg<256, 7>(a, s);
        else if (ept == 8) launch_pose_reg<256, 8>(a, s);
        else if (ept <= 10) launch_pose_reg<256, 10>(a, s);
        else if (ept == 11) launch_pose_reg<256, 11>(a, s);
        else launch_pose_reg<256, 12>(a, s);
        return;
    }
    if (!classic && !lds_only && a.n <= 1024) {
        // register-resident kernel, 256 threads (one wave per SIMD, 512 registers per lane) and 2..4 edges per thread.  (A
        // 512-thread variant - two waves per SIMD, 256 registers each - was measured in round 2 and lost, 104 vs 73 us at 800
        // points, NOTES B; its instances spilled by construction and are gone.)
        const int ept = (a.n + 255) / 256;
        if (ept <= 2) launch_pose_reg<256, 2>(a, s);  // (<= 256 points too: see launch_pose_opt_batch)
        else if (ept == 3) launch_pose_reg<256, 3>(a, s);
        else launch_pose_reg<256, 4>(a, s);
        return;
    }
    if (a.n <= kPoLdsMax && !classic) {
        const size_t lds = sizeof(double) * 2 * (size_t)a.n + sizeof(float) * 6 * (size_t)a.n + (size_t)a.n + 16;
        static bool big_lds[64] = {};  // per device; 41 B per edge: 3072 edges = 126 KB of the CU's 160 KB
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !big_lds[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pose_opt_lds_kernel<512>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            big_lds[dev] = true;
        }
        // 256 threads (one wave per SIMD, 2-3 edges per thread) up to 640 points: measured 117 us vs 122 us with 512
        // and 144 us with 128 threads at n = 500
        if (a.n <= 640) hipLaunchKernelGGL(pose_opt_lds_kernel<256>, dim3(1), dim3(256), lds, s, a);
        else hipLaunchKernelGGL(pose_opt_lds_kernel<512>, dim3(1), dim3(512), lds, s, a);
    } else {
        hipLaunchKernelGGL(pose_opt_kernel, dim3(1), dim3(kPoThreads), 0, s, a);
    }
}

}  // namespace so
