/*
 * frame_oracle.h — CPU restatement (parity oracle) of the Frame post-processing between the extractor and the
 * matcher: Frame::UndistortKeyPoints, ComputeImageBounds, AssignFeaturesToGrid, isInFrustum (+ MapPoint::PredictScale).
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.h): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
 * use it.
 *
 * Parity status: code/src/Frame.cc:277-292, 316-375, 431-443, 454-514 and code/src/MapPoint.cc:466-485 are followed
 * literally.  Two pieces of arithmetic live in un-vendored OpenCV (version not pinned by the reference)
 * and in libm: PARITY UNPINNED for them; the conventions adopted here are
 *   cv::undistortPoints(src, dst, K, D, noArray(), K)   the published algorithm of cvUndistortPointsInternal: in
 *       double, x = (u-cx)/fx, five fixed-point iterations of the radial-tangential model (default TermCriteria
 *       COUNT 5), then u' = fx x + cx, rounded to float;
 *   cv::Mat algebra of isInFrustum (mRcw*P+mtcw, cv::norm, Mat::dot)   float inputs, double accumulation, one
 *       rounding to float per result, as OpenCV's GEMM / norm / dot kernels do for CV_32F;
 *   log() in PredictScale   a correctly rounded-to-float natural logarithm computed in double by a fixed series
 *       (orc_log), so that CPU and GPU agree bit for bit; glibc's logf differs from it only on exact ties.
 * Pinned by tests/test_frame_oracle.py: distort(undistort(p)) == p to 1e-3 px for EuRoC's coefficients, the
 * zero-distortion identities, orc_log against math.log to 1 ulp of float, grid lists against a literal
 * vector<vector<>> fill.
 */
#ifndef FRAME_ORACLE_H
#define FRAME_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float fx, fy, cx, cy;     /* mK */
    float k1, k2, p1, p2, k3; /* mDistCoef (k3 = 0 when the yaml has four coefficients) */
} orc_camera;

double orc_log(double x); /* natural logarithm, fixed operation sequence (no libm) */

/* Frame::UndistortKeyPoints, code/src/Frame.cc:454-486: xy interleaved (x0 y0 x1 y1 ...) */
void orc_undistort_keypoints(const orc_camera* cam, int32_t n, const float* xy, float* xy_un);

/* Frame::ComputeImageBounds, code/src/Frame.cc:488-514: bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY} */
void orc_image_bounds(const orc_camera* cam, int32_t width, int32_t height, float* bounds4);

/* Frame::AssignFeaturesToGrid + PosInGrid, code/src/Frame.cc:277-292, 431-443.  cell = x * 48 + y (mGrid[x][y]);
 * cell_of[i] = -1 outside the grid; cell_start has 64*48+1 entries; cell_items lists the keypoints of each cell in
 * insertion (= index) order.  Returns the number of keypoints inside the grid. */
int32_t orc_assign_features_to_grid(int32_t n, const float* xy_un, const float* bounds4, int32_t* cell_of,
                                    int32_t* cell_start, int32_t* cell_items);

/* Frame::UpdatePoseMatrices' mOw = -mRcw.t()*mtcw, code/src/Frame.cc:306-313 (Tcw: 3x4 row-major float) */
void orc_camera_center(const float* Tcw12, float* Ow3);

/* Frame::isInFrustum for a batch of map points, code/src/Frame.cc:316-375 with MapPoint::PredictScale
 * (code/src/MapPoint.cc:476-485) and Get{Min,Max}DistanceInvariance (:466-474).  max_dist / min_dist are the
 * map points' mfMaxDistance / mfMinDistance.  Outputs are written only where in_view[i] = 1 (mbTrackInView). */
void orc_is_in_frustum(const orc_camera* cam, const float* bounds4, const float* Tcw12, int32_t n, const float* Xw,
                       const float* normal, const float* max_dist, const float* min_dist, float viewing_cos_limit,
                       float log_scale_factor, int32_t n_scale_levels, uint8_t* in_view, float* proj_x,
                       float* proj_y, float* view_cos, int32_t* pred_level);

/* The projection step of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), code/src/ORBmatcher.cc:
 * 1251-1270, for a batch of the last frame's map points: x3Dc = Rcw * x3Dw + tcw (cv::Mat algebra: double
 * accumulation, one rounding), invzc = 1.0 / z, u = fx * xc * invzc + cx; valid[i] = has_point[i] && invzc >= 0 &&
 * (u, v) inside the image bounds.  u / v are written only where valid[i] = 1. */
void orc_project_last_frame(const orc_camera* cam, const float* bounds4, const float* Tcw12, int32_t n, const float* Xw,
                            const uint8_t* has_point, uint8_t* valid, float* u, float* v);

#ifdef __cplusplus
}
#endif
#endif
