/*
 * mapping_oracle.h — CPU restatement (parity oracle) of the two data-parallel loops the local-mapping thread runs
 * between the matcher and local BA:
 *     the per-match body of LocalMapping::CreateNewMapPoints   code/src/LocalMapping.cc:263-420 (monocular branch)
 *     MapPoint::UpdateNormalAndDepth                            code/src/MapPoint.cc:413-465
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.h): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 *
 * Parity status: control flow, gates, thresholds and float / double expression order are the reference's.  PARITY
 * UNPINNED for what lives in un-vendored OpenCV; the conventions are those of project_oracle.h plus
 *   cv::SVD::compute(A, w, u, vt, MODIFY_A | FULL_UV) of the 4 x 4 CV_32F system   OpenCV's published one-sided Jacobi
 *       (modules/core/src/lapack.cpp, JacobiSVDImpl_<float>): works on A^T (rows = columns of A), squared norms and the
 *       dot product of a column pair accumulated in double, pair (i, j) skipped when |p| <= eps sqrt(a b) with
 *       eps = 2 FLT_EPSILON, rotation (c, s) formed in double and rounded to float, rows rotated in float
 *       (c * x + s * y: two products, one sum), at most max(m, 30) sweeps, singular values sorted descending by a
 *       selection sort that swaps the rows of V^T along; hypot(p, beta) is taken as sqrt(p p + beta beta) in double so
 *       that CPU and GPU agree bit for bit;
 *   s * row - row (the rows of A), Mat / s, a + b / s   element-wise float: one product (by (float)alpha), one sum.
 * MapPoint::UpdateNormalAndDepth walks std::map<KeyFrame*, size_t> - pointer order, different from run to run in the
 * reference itself: the order of the caller's observation list defines the summation order here.
 */
#ifndef MAPPING_ORACLE_H
#define MAPPING_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* what CreateNewMapPoints reads of a keyframe */
typedef struct {
    float Tcw[12];               /* [Rcw | tcw] row-major */
    float fx, fy, cx, cy, invfx, invfy;
    const float* scale_factors;  /* mvScaleFactors */
    const float* level_sigma2;   /* mvLevelSigma2 */
} orc_tri_keyframe;

/* right singular vector of the smallest singular value of a 4 x 4 float matrix (row-major), OpenCV's Jacobi SVD as
 * described above: v4 = vt.row(3) */
void orc_svd4_last_row(const float* A16, float* v4);

/* The per-match body of CreateNewMapPoints for n matches between keyframe 1 (mpCurrentKeyFrame) and keyframe 2:
 * parallax of the rays, linear triangulation, positive depth in both, reprojection error <= 5.991 sigma2 in both, scale
 * consistency with ratio_factor = 1.5f * mfScaleFactor.  ok[k] = 1 where a MapPoint would be created, x3D[3k..] its
 * position (untouched otherwise). */
void orc_triangulate_matches(const orc_tri_keyframe* kf1, const orc_tri_keyframe* kf2, float ratio_factor, int32_t n,
                             const float* xy1, const int32_t* octave1, const float* xy2, const int32_t* octave2,
                             uint8_t* ok, float* x3D);

/* MapPoint::UpdateNormalAndDepth for a batch: point p is observed from camera centres obs_Ow[offsets[p] .. offsets[p+1])
 * (3 floats each, in the order the caller walks mObservations); ref_Ow = pRefKF->GetCameraCenter(), ref_level_scale =
 * pRefKF->mvScaleFactors[level of the point's keypoint in pRefKF], ref_last_scale = pRefKF->mvScaleFactors[nLevels - 1].
 * A point without observations keeps its outputs untouched. */
void orc_update_normal_and_depth(int32_t n_points, const int32_t* offsets, const float* obs_Ow, const float* Xw,
                                 const float* ref_Ow, const float* ref_level_scale, const float* ref_last_scale,
                                 float* normal, float* max_dist, float* min_dist);

#ifdef __cplusplus
}
#endif
#endif
