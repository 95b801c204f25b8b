/*
 * ba_oracle.h — CPU restatement (parity oracle) of SwarmMap's bundle adjustment on a flattened problem:
 * Optimizer::LocalBundleAdjustment / BundleAdjustment (code/src/Optimizer.cc:42-237,436-740) with the g2o
 * machinery they drive (Levenberg-Marquardt, BlockSolver<6,3> Schur complement, Huber kernel,
 * EdgeSE3ProjectXYZ, SE3Quat).  TEST INFRASTRUCTURE ONLY (see orb_oracle.h).
 *
 * Parity status: "parity unpinned" at the linear-algebra boundary — g2o needs Eigen (absent from the image,
 * un-vendored, version unpinned: CMakeLists.txt:81), so the reference cannot be compiled here and ships no
 * golden vectors.  What IS pinned (tests/test_ba_oracle.py): the analytic Jacobians against central finite
 * differences (g2o's own numeric fallback, base_binary_edge.hpp:147-197), the Huber kernel formulas
 * (robust_kernel_impl.cpp:78-91), SE3 exp against the matrix exponential series, the Schur solve against a
 * full dense solve of the same normal equations, and convergence to the generating parameters on noise-free
 * synthetic windows.  Eigen pieces restated from their published algorithms: Quaterniond(Matrix3d),
 * Quaternion * vector, toRotationMatrix, 3x3 inverse (cofactors), SPD solve (dense Cholesky instead of
 * SimplicialLDLT — same solution up to rounding).
 * Monocular edges only (EdgeSE3ProjectXYZ), which is all SwarmMap builds.
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t n_poses;       /* keyframe vertices, listed in ascending vertex id (KeyFrame::mnId) */
    const float* Tcw;      /* n_poses x 12, row-major [R|t] of KeyFrame::GetPose() (cv::Mat CV_32F) */
    const uint8_t* fixed;  /* n_poses: setFixed(...) */
    const float* intr;     /* n_poses x 4: fx, fy, cx, cy of the keyframe (copied into every edge) */
    int32_t n_points;      /* map point vertices, ascending vertex id */
    const float* Xw;       /* n_points x 3, MapPoint::GetWorldPos() */
    int32_t n_edges;       /* EdgeSE3ProjectXYZ, in insertion order (= edge id order) */
    const int32_t* edge_pose;
    const int32_t* edge_point;
    const float* obs;         /* n_edges x 2: kpUn.pt */
    const float* inv_sigma2;  /* n_edges: mvInvLevelSigma2[kpUn.octave] */
} orc_ba_problem;

typedef struct {
    int32_t its_stage1;   /* optimize(5) / optimize(nIterations) */
    int32_t its_stage2;   /* optimize(10) after the outlier pass; 0 = single stage (BundleAdjustment) */
    int32_t robust;       /* Huber kernel on every edge in stage 1 */
    float huber_delta;    /* sqrt(5.991) as float (thHuberMono / thHuber2D) */
    float chi2_threshold; /* 5.991 */
} orc_ba_options;

typedef struct {
    double chi2_initial, chi2_final; /* activeRobustChi2 before / after the last executed stage */
    double lambda_final;
    int32_t iterations_stage1, iterations_stage2; /* LM iterations actually run */
    int32_t lm_trials;                             /* total inner trials */
    int32_t aborted;                               /* stop flag observed */
    int32_t n_outliers;
} orc_ba_info;

/* Returns 0 on success.  Tcw_out: n_poses x 12 (Converter::toCvMat of every pose vertex), Xw_out: n_points x 3,
 * edge_outlier: n_edges (chi2 > threshold || depth <= 0 at the end, from the edges' stored errors exactly as
 * Optimizer.cc:682-695 reads them), edge_chi2: n_edges (stored chi2).  stop may be NULL.
 * If *stop is set before the first optimize the outputs are the inputs converted and info->aborted = 1. */
int orc_bundle_adjust(const orc_ba_problem* p, const orc_ba_options* opt, const volatile uint8_t* stop,
                      float* Tcw_out, float* Xw_out, uint8_t* edge_outlier, double* edge_chi2, orc_ba_info* info);

/* pieces exposed for the known-answer tests */
void orc_se3_from_Tcw(const float* Tcw12, double* q_xyzw, double* t);       /* Converter::toSE3Quat */
void orc_se3_to_Tcw(const double* q_xyzw, const double* t, float* Tcw12);   /* Converter::toCvMat(SE3Quat) */
void orc_se3_exp_mul(const double* update6, double* q_xyzw, double* t);     /* T <- SE3Quat::exp(update) * T */
/* EdgeSE3ProjectXYZ::computeError + linearizeOplus: err[2], Jpoint[2x3], Jpose[2x6] row-major; returns depth */
double orc_edge_project(const double* q_xyzw, const double* t, const double* X, const double* obs,
                        const double* intr, double* err, double* Jpoint, double* Jpose);
void orc_huber(double e, double delta, double* rho3);                       /* RobustKernelHuber::robustify */

#ifdef __cplusplus
}
#endif
#endif

/* ---- Optimizer::PoseOptimization (code/src/Optimizer.cc:239-434), monocular, on flattened inputs ---- */
#ifndef BA_ORACLE_POSE_H
#define BA_ORACLE_POSE_H
#ifdef __cplusplus
extern "C" {
#endif
/* One SE3 vertex, n unary EdgeSE3ProjectXYZOnlyPose edges (types_six_dof_expmap.h:143-171, .cpp:266-296), Huber
 * sqrt(5.991), 4 rounds of optimize(10) each restarted from the input pose, inlier/outlier re-classification after
 * every round (float chi2 > 5.991f), kernel dropped after round 3, dense 6x6 solve (LinearSolverDense).
 * Tcw12: row-major [R|t] float (pFrame->mTcw); intr: fx, fy, cx, cy; Xw n x 3, obs n x 2, inv_sigma2 n (floats).
 * Returns nInitialCorrespondences - nBad (0 and untouched outputs when n < 3).
 * outlier[i] = pFrame->mvbOutlier of the i-th correspondence; info[0] = LM iterations run, info[1] = LM trials. */
int orc_pose_optimization(const float* Tcw12, const float* intr, int32_t n, const float* Xw, const float* obs,
                          const float* inv_sigma2, float* Tcw_out12, uint8_t* outlier, int32_t* info);
#ifdef __cplusplus
}
#endif
#endif
