// ba_device.h — device-side layout of the bundle-adjustment solver (FP64 throughout, like g2o).
//
// HBM layout (one so_ba = one agent's local-mapping solver context; buffers grow on demand, stay resident):
//   poses     2 buffers (current / trial) of Pose{q[4] xyzw, t[3], pad}      n_poses x 64 B
//   points    2 buffers (current / trial) of double[3]                        n_points x 24 B
//   edges     sorted by point (stable): pose idx, point idx, obs[2], inv_sigma2, active flag
//             + stored error[2] (EdgeSE3ProjectXYZ::_error) that only active edges refresh
//   CSR       point -> its edges (contiguous after the sort); free pose -> its edges; upper block (i1,i2) of the
//             reduced camera system -> the (edge, edge) pairs that share a landmark (built per stage on the host)
//   system    Hpp[n_free][36], bp[n_free][6], Hll[n_pt][9], bl[n_pt][3], W[edge][18] = J_pose^T w J_point,
//             Dinv[n_pt][9], db[n_pt][3], BDinv[edge][18], S[(6 n_free)^2] dense, bs / x_p[6 n_free], x_l[n_pt][3]
// Summation orders are fixed (CSR order, fixed reduction trees, no floating-point atomics) so results are
// deterministic run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace so {

struct BaPose {
    double q[4];  // x y z w
    double t[3];
    double pad;
};

struct BaDev {
    // problem
    int n_poses, n_points, n_edges;
    int n_free;          // poses with a hessian index in the current stage
    int n_active_points;  // informational
    const double* intr;       // n_poses x 4
    const int* e_pose;        // sorted edges
    const int* e_point;
    const double* e_obs;      // x2
    const double* e_w;        // inv_sigma2
    const uint8_t* e_active;  // level == 0 in this stage
    double* e_err;            // x2 stored error
    double* e_chi2;           // stored chi2 (refreshed with the error)
    const int* pt_off;        // n_points + 1
    const uint8_t* pt_active; // n_points
    const int* pose_hidx;     // n_poses: hessian index or -1
    const int* free_pose;     // n_free: pose index of hessian index i
    const int* pose_off;      // n_free + 1
    const int* pose_edges;    // edge ids per free pose
    const int* blk_off;       // n_blk + 1, n_blk = n_free (n_free + 1) / 2, block (i1,i2) at i1*n_free - i1(i1-1)/2 + (i2-i1)
    const int* pair_k1;
    const int* pair_k2;
    // system
    double* Hpp; double* bp; double* Hll; double* bl; double* W;
    double* Dinv; double* db; double* BDinv;
    double* S; double* bs; double* xl;
    double* partial;  // reduction partials (chi2 | scale) + flags
    int robust;
    double huber_delta;
    float huber_dsqr;  // stored as float in the reference (robust_kernel_impl.h:84)
};

constexpr int kBaPartialChi = 0;       // [0, 1024): chi2 partials of the error kernel
constexpr int kBaPartialScale = 1024;  // [1024, 2048): scale partials of the update kernel
constexpr int kBaMaxDiag = 2048;       // [2048]: max |diag|
constexpr int kBaSolveOk = 2049;       // [2049]: 1.0 if the Cholesky succeeded
constexpr int kBaPartialCount = 2056;

void launch_ba_errors(const BaDev& d, const BaPose* poses, const double* points, int n_blocks, hipStream_t s);
void launch_ba_build(const BaDev& d, const BaPose* poses, const double* points, hipStream_t s);
void launch_ba_maxdiag(const BaDev& d, hipStream_t s);
void launch_ba_schur(const BaDev& d, double lambda, const int* blk_i1, const int* blk_i2, int n_blk, hipStream_t s);
void launch_ba_solve(const BaDev& d, hipStream_t s);
void launch_ba_update(const BaDev& d, double lambda, const BaPose* poses, const double* points, BaPose* poses_trial,
                      double* points_trial, int n_blocks, hipStream_t s);
void launch_ba_depth(const BaDev& d, const BaPose* poses, const double* points, double* depth, hipStream_t s);

struct PoseOptArgs {  // Optimizer::PoseOptimization, one workgroup (ba_kernels.hip)
    const float* Xw;          // n x 3
    const float* obs;         // n x 2
    const float* inv_sigma2;  // n
    double K[4];
    BaPose init;
    int n;
    double* err;       // n x 2 scratch (stored _error)
    uint8_t* outlier;  // n (out)
    BaPose* pose_out;
    int* info;         // [0] nBad, [1] LM iterations, [2] LM trials
    double* trace;     // optional: 4 doubles per LM trial (lambda, tempChi, rho, currentChi), 256 trials max
};
void launch_pose_opt(const PoseOptArgs& a, hipStream_t s);

}  // namespace so
