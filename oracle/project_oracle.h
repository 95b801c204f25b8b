/*
 * project_oracle.h — CPU restatement (parity oracle) of the projection + gating half of the map-point searches of
 * ORBmatcher that LocalMapping, loop closing, map merging and relocalisation call, and of
 * the five routines end to end on flattened inputs:
 *     Fuse(KeyFrame*, const vector<MapPoint*>&, th)                         code/src/ORBmatcher.cc:751-891
 *     Fuse(KeyFrame*, cv::Mat Scw, vpPoints, th, vpReplacePoint)            :893-1009
 *     SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)              :1011-1221
 *     SearchByProjection(KeyFrame*, cv::Mat Scw, vpPoints, vpMatched, th)   :264-373
 *     SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist, bGlobal)  :1356-1473
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.h): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
 * use it.
 *
 * Parity status: the control flow, gates, thresholds and float expression order are the reference's, statement by
 * statement.  The cv::Mat algebra (un-vendored OpenCV, version not pinned by the reference) is PARITY UNPINNED; the
 * conventions are those of frame_oracle.h plus, for the similarity transforms:
 *   A * x + b, -A.t() * b, A * b      one GEMM call: float inputs, double accumulation, alpha / beta applied in double,
 *                                     one rounding to float per element (OpenCV's GEMMSingleMul<float, double>);
 *   Mat::dot, cv::norm                double accumulation of double products (dot stays double, norm = sqrt in double
 *                                     and is rounded when assigned to a float);
 *   Mat / s, s * Mat, s * Mat.t()     MatExpr scaling = convertTo(alpha): alpha is formed in double (1.0 / s for the
 *                                     division), cast to float, and multiplied element-wise in float (cvt_32f scale);
 *   log() in PredictScale             orc_log, as in frame_oracle.h.
 * The window search behind every projection is orc_search_window_best / _greedy (matcher_oracle.h), already pinned
 * by the grid and DescriptorDistance KATs.
 */
#ifndef PROJECT_ORACLE_H
#define PROJECT_ORACLE_H
#include <stdint.h>

#include "frame_oracle.h"
#include "matcher_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The MapPoint fields these searches read (code/include/MapPoint.h), one entry per element of vpMapPoints. */
typedef struct {
    int32_t n;
    const float* Xw;       /* n x 3: GetWorldPos() (GetGlobalPos() where the routine says so) */
    const float* normal;   /* n x 3: GetNormal(); may be NULL for the routines without the viewing-angle gate */
    const float* max_dist; /* mfMaxDistance (GetMaxDistanceInvariance() = 1.2f * this) */
    const float* min_dist; /* mfMinDistance (GetMinDistanceInvariance() = 0.8f * this) */
    const uint8_t* desc;   /* n x 32: GetDescriptor() */
    const uint8_t* valid;  /* the routine's object-graph gates: pMP && !pMP->isBad() && !IsInKeyFrame / !already found */
} orc_mappoint_view;

/* What the projection half hands to the window search, per map point (written for every i; u / v / radius / level
 * are 0 where active[i] = 0). */
typedef struct {
    uint8_t* active;
    float* u;
    float* v;
    float* radius;  /* th * mvScaleFactors[nPredictedLevel] */
    int32_t* level; /* nPredictedLevel */
} orc_window_queries;

/* Scw -> Rcw, tcw, Ow (ORBmatcher.cc:272-277 = :901-906).  Scw12: 3x4 row-major. */
void orc_sim3_decompose(const float* Scw12, float* Rcw9, float* tcw3, float* Ow3);
/* sR12 = s12 * R12, sR21 = (1.0 / s12) * R12.t(), t21 = -sR21 * t12 (ORBmatcher.cc:1027-1029). */
void orc_sim3_relative(float s12, const float* R12, const float* t12, float* sR12, float* sR21, float* t21);

/* Fuse(pKF, vpMapPoints, th), projection half: ORBmatcher.cc:767-815.  Tcw12 = [Rcw | tcw] of pKF. */
void orc_fuse_queries(const orc_frame_view* KF, const orc_camera* cam, const float* Tcw12, float log_scale_factor,
                      const orc_mappoint_view* mp, float th, const orc_window_queries* out);
/* Fuse(pKF, Scw, ...) :916-964 and SearchByProjection(pKF, Scw, ...) :286-333 (the same statements). */
void orc_sim3_world_queries(const orc_frame_view* KF, const orc_camera* cam, const float* Scw12, float log_scale_factor,
                            const orc_mappoint_view* mp, float th, const orc_window_queries* out);
/* One direction of SearchBySim3, :1054-1094 (= :1130-1170 with the roles swapped): map points of the SOURCE keyframe
 * (pose Tsw12) go through p3Dc_src = Rsw * p3Dw + tsw, p3Dc_dst = sR * p3Dc_src + t and are projected into the
 * TARGET keyframe `KF` with `cam` (the reference uses pKF1's intrinsics in both directions, :1013-1016). */
void orc_sim3_pair_queries(const orc_frame_view* KF, const orc_camera* cam, const float* Tsw12, const float* sR9,
                           const float* t3, float log_scale_factor, const orc_mappoint_view* mp, float th,
                           const orc_window_queries* out);
/* SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist, bGlobal), :1374-1410.  Tcw12 = CurrentFrame.mTcw. */
void orc_frame_kf_queries(const orc_frame_view* F, const orc_camera* cam, const float* Tcw12, float log_scale_factor,
                          const orc_mappoint_view* mp, float th, const orc_window_queries* out);

/* ---- the five routines end to end ---- */

/* Fuse(pKF, vpMapPoints, th) up to the map side effects: best_idx[i] = bestIdx where bestDist <= TH_LOW (:873), else
 * -1; best_dist[i] = bestDist (256 when no candidate).  inv_level_sigma2 = pKF->mvInvLevelSigma2.  Returns nFused. */
int orc_fuse(const orc_frame_view* KF, const orc_camera* cam, const float* Tcw12, float log_scale_factor,
             const float* inv_level_sigma2, const orc_mappoint_view* mp, float th, int32_t* best_idx,
             int32_t* best_dist);
/* Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) likewise (:995). */
int orc_fuse_sim3(const orc_frame_view* KF, const orc_camera* cam, const float* Scw12, float log_scale_factor,
                  const orc_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist);
/* SearchBySim3: mp1.valid[i1] = pMP && !vbAlreadyMatched1[i1] && !isBad (:1057-1061), mp2 likewise; match12[i1] = idx2
 * of the agreeing pair or -1 (:1205-1218).  mp1.n = N1 = keypoints of KF1, mp2.n = N2.  Returns nFound. */
int orc_search_by_sim3(const orc_frame_view* KF1, const orc_frame_view* KF2, const orc_camera* cam, const float* T1w12,
                       const float* T2w12, float s12, const float* R12, const float* t12, float log_scale_factor1,
                       float log_scale_factor2, const orc_mappoint_view* mp1, const orc_mappoint_view* mp2, float th,
                       int32_t* match12);
/* SearchByProjection(pKF, Scw, vpPoints, vpMatched, th): KF->excluded[k] = vpMatched[k] != NULL on entry;
 * kp_to_point[k] = index into vpPoints bound to keypoint k by this call or -1.  Returns nmatches. */
int orc_search_by_projection_sim3(const orc_frame_view* KF, const orc_camera* cam, const float* Scw12,
                                  float log_scale_factor, const orc_mappoint_view* mp, int th, int32_t* kp_to_point);
/* SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist): F->excluded[k] = CurrentFrame.mvpMapPoints[k] !=
 * NULL on entry; mp_angle[i] = pKF->mvKeysUn[i].angle.  Returns nmatches (after the rotation histogram). */
int orc_search_by_projection_frame_kf(const orc_frame_view* F, const orc_camera* cam, const float* Tcw12,
                                      float log_scale_factor, const orc_mappoint_view* mp, const float* mp_angle,
                                      float th, int orb_dist, int check_orientation, int32_t* kp_to_point);

#ifdef __cplusplus
}
#endif
#endif
