/*
 * matcher_oracle.h — CPU restatement (parity oracle) of SwarmMap's ORBmatcher tracking routines on
 * flattened inputs.  TEST INFRASTRUCTURE ONLY (see orb_oracle.h).
 *
 * Parity status: integer work, fully determined by the reference source (no third-party arithmetic):
 * code/src/ORBmatcher.cc + Frame grid helpers code/src/Frame.cc:277-292,377-443.  Pinned by the KATs in
 * tests/test_matcher_oracle.py (DescriptorDistance == popcount(a^b); literal 64x48 grid traversal).
 * Monocular only (mvuRight < 0 everywhere), which is all SwarmMap builds (Examples/Monocular).
 */
#ifndef MATCHER_ORACLE_H
#define MATCHER_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_GRID_COLS 64 /* FRAME_GRID_COLS, code/include/Frame.h:38 */
#define ORC_GRID_ROWS 48 /* FRAME_GRID_ROWS, code/include/Frame.h:37 */

/* Flattened view of the parts of ORB_SLAM2::Frame the matcher reads. */
typedef struct {
    int32_t n;              /* N */
    const float* x;         /* mvKeysUn[i].pt.x */
    const float* y;         /* mvKeysUn[i].pt.y */
    const int32_t* octave;  /* mvKeysUn[i].octave */
    const float* angle;     /* mvKeysUn[i].angle (degrees) */
    const uint8_t* desc;    /* mDescriptors, n x 32 */
    const uint8_t* excluded; /* 1 iff mvpMapPoints[i] && mvpMapPoints[i]->Observations() > 0 on entry; may be NULL */
    float min_x, max_x, min_y, max_y; /* mnMinX, mnMaxX, mnMinY, mnMaxY */
    float grid_inv_w, grid_inv_h;     /* mfGridElementWidthInv / HeightInv */
    const float* scale_factors;       /* mvScaleFactors */
    int32_t nlevels;
    /* A KeyFrame keeps the Frame's grid (filled with the Frame's float origin, KeyFrame.cc:66-72) but queries it with
     * its own int-truncated bounds (KeyFrame.h:220, KeyFrame.cc:779-818): has_grid_origin != 0 -> PosInGrid used
     * (grid_min_x, grid_min_y), GetFeaturesInArea / IsInImage use min_x .. max_y. */
    int32_t has_grid_origin;
    float grid_min_x, grid_min_y;
} orc_frame_view;

/* ORBmatcher::DescriptorDistance, code/src/ORBmatcher.cc:1511-1525 (SWAR popcount, literal) */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);

/* Frame::GetFeaturesInArea through a literal 64x48 grid, code/src/Frame.cc:377-443.  Returns count. */
int orc_features_in_area(const orc_frame_view* F, float x, float y, float r, int min_level, int max_level,
                         int32_t* out_idx, int cap);

/* ORBmatcher::ComputeThreeMaxima, code/src/ORBmatcher.cc:1475-1506 (on bin populations) */
void orc_three_maxima(const int32_t* histo_sizes, int L, int* ind1, int* ind2, int* ind3);

/* M1: ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), code/src/ORBmatcher.cc:44-121.
 * in_view[i] = pMP->mbTrackInView && !pMP->isBad().  kp_to_mp[k] = index of the map point bound to
 * keypoint k by this call, or -1.  Returns nmatches. */
int orc_search_by_projection_mappoints(const orc_frame_view* F, int32_t n_mp, const uint8_t* in_view,
                                       const float* proj_x, const float* proj_y, const float* view_cos,
                                       const int32_t* pred_level, const uint8_t* mp_desc,
                                       const uint8_t* mp_has_obs, float th, float nn_ratio, int32_t* kp_to_mp);

/* M2: ORBmatcher::SearchByProjection(Frame& cur, const Frame& last, th, bMono=true),
 * code/src/ORBmatcher.cc:1223-1354, from the projection onwards: valid[i] = pMP && !outlier && invzc >= 0 &&
 * (u,v) inside the image bounds.  kp_to_last[k] = index i in the last frame, or -1. */
int orc_search_by_projection_lastframe(const orc_frame_view* cur, int32_t n_last, const uint8_t* valid,
                                       const float* u, const float* v, const int32_t* last_octave,
                                       const float* last_angle, const uint8_t* mp_desc,
                                       const uint8_t* mp_has_obs, float th, int check_orientation,
                                       int32_t* kp_to_last);

/* M4: ORBmatcher::SearchForInitialization, code/src/ORBmatcher.cc:375-479.
 * prev_matched: n1 x 2 floats, updated in place.  matches12: n1 ints out. */
int orc_search_for_initialization(const orc_frame_view* F1, const orc_frame_view* F2, float* prev_matched,
                                  int window, float nn_ratio, int check_orientation, int32_t* matches12);

/* Brute-force best / second-best Hamming match of every row of A against all rows of B (the cross-agent
 * keyframe search after the descriptor all-gather).  Ties: lowest index in B wins. */
void orc_hamming_top2(const uint8_t* A, int na, const uint8_t* B, int nb, int32_t* best_idx, int32_t* best_dist,
                      int32_t* second_dist);

#ifdef __cplusplus
}
#endif
#endif

/* ---- additional routines (M3, M5, M6, M7); declared after the include guard's closing on purpose ---- */
#ifndef MATCHER_ORACLE_EXT_H
#define MATCHER_ORACLE_EXT_H
#ifdef __cplusplus
extern "C" {
#endif

/* DBoW2::FeatureVector flattened: nodes ascending by id, node k owns idx[off[k] .. off[k+1]) (feature indices in
 * the order DBoW2 stored them). */
typedef struct {
    int32_t n_nodes;
    const int32_t* node_id;
    const int32_t* off; /* n_nodes + 1 */
    const int32_t* idx;
} orc_featvec;

/* M3: ORBmatcher::SearchByBoW.  kf_frame = 0: (KeyFrame*, Frame&) variant, code/src/ORBmatcher.cc:150-262
 * (accept best <= TH_LOW; a target is taken once matched).  kf_frame = 1: (KeyFrame*, KeyFrame*) variant,
 * :481-597 (accept best < TH_LOW; targets need valid2[idx2] = pMP2 && !pMP2->isBad()).
 * valid1[i] = pMP1 && !pMP1->isBad().  Variant 0 returns match_of_2[k2] = index in set 1 bound to target k2;
 * variant 1 returns match_of_1[k1] = target index.  Unused output may be NULL. */
int orc_search_by_bow(int variant, int32_t n1, const uint8_t* desc1, const float* angle1, const uint8_t* valid1,
                      const orc_featvec* fv1, int32_t n2, const uint8_t* desc2, const float* angle2,
                      const uint8_t* valid2, const orc_featvec* fv2, float nn_ratio, int check_orientation,
                      int32_t* match_of_2, int32_t* match_of_1);

/* M5: ORBmatcher::SearchForTriangulation (monocular), code/src/ORBmatcher.cc:599-749 + CheckDistEpipolarLine
 * :131-148.  free1/free2[i] = !GetMapPoint(i).  F12 row-major 3x3 float; (ex, ey) epipole in image 2.
 * matches12[i1] = index in 2 or -1.  Returns nmatches. */
int orc_search_for_triangulation(int32_t n1, const float* x1, const float* y1, const float* angle1,
                                 const uint8_t* desc1, const uint8_t* free1, const orc_featvec* fv1, int32_t n2,
                                 const float* x2, const float* y2, const int32_t* octave2, const float* angle2,
                                 const uint8_t* desc2, const uint8_t* free2, const orc_featvec* fv2,
                                 const float* F12, float ex, float ey, const float* scale_factors2,
                                 const float* level_sigma2_2, int check_orientation, int32_t* matches12);

/* M6 / M7 core: for every valid query the first minimum-distance keypoint of KeyFrame::GetFeaturesInArea(u,v,r)
 * (code/src/KeyFrame.cc:779-814) whose octave is in [pred-1, pred]; chi2_gate: additionally
 * e2 * inv_sigma2[octave] <= 5.99 (ORBmatcher::Fuse, :829-861).  Independent queries.
 * Used by Fuse (:751-891, :893-1009) and both passes of SearchBySim3 (:1011-1221). */
void orc_search_window_best(const orc_frame_view* KF, int32_t nq, const uint8_t* valid, const float* u,
                            const float* v, const float* radius, const int32_t* pred_level, const uint8_t* qdesc,
                            int chi2_gate, const float* inv_sigma2, int32_t* best_idx, int32_t* best_dist);

/* M7 (and M2 generalised): sequential greedy window search — SearchByProjection(KeyFrame*, Scw, ...) :264-373
 * (levels [pred-1,pred], TH_LOW, no orientation) and SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th,
 * ORBdist) :1356-1473 (levels [pred-1,pred+1], ORBdist, orientation).  A keypoint is skipped when F->excluded
 * or when an earlier query of this call took it.  kp_to_query[k] = query bound to keypoint k or -1. */
int orc_search_window_greedy(const orc_frame_view* F, int32_t nq, const uint8_t* valid, const float* u,
                             const float* v, const float* radius, const int32_t* min_level,
                             const int32_t* max_level, const uint8_t* qdesc, const float* q_angle, int max_dist,
                             int check_orientation, int32_t* kp_to_query);

/* MapPoint::ComputeDistinctiveDescriptors, code/src/MapPoint.cc:361-391, on the N >= 1 descriptors of one map
 * point: returns BestIdx (the first row with the least median), *median_out = BestMedian. */
int orc_distinctive_descriptor(const uint8_t* descs, int32_t n, int32_t* median_out);

#ifdef __cplusplus
}
#endif
#endif
