/*
 * swarmorb.h — C ABI of libswarmorb.so, the MI355X (gfx950) implementation of SwarmMap's per-frame
 * ORB front-end, Hamming matcher and bundle-adjustment hot path.
 *
 * Plain C types only (pointers + sizes); no torch / OpenCV / Eigen types cross this boundary.
 * Each entry point cites the reference interface (file:line under /root/reference) it replaces.
 * The reference-side bindings a maintainer would add are shown in INTEGRATION.md; C++ adapter
 * classes with the reference's own signatures live in swarmmap_amd/host/.
 *
 * Threading: a handle is NOT re-entrant (like the reference's extractor, which owns CUDA streams and
 * fixed-size device buffers, code/src/cuda/Fast_gpu.cu:342-369); different handles may be used
 * concurrently from different threads (N agents in one process).
 */
#ifndef SWARMORB_H
#define SWARMORB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the reference exit()s on CUDA errors via checkCudaErrors; we return) ---- */
enum {
    SO_OK = 0,
    SO_ERR_INVALID_ARG = 1,
    SO_ERR_NO_DEVICE = 2,    /* no HIP device / HIP runtime error at init */
    SO_ERR_HIP = 3,          /* HIP runtime error during a call (see so_last_error) */
    SO_ERR_CAPACITY = 4,     /* caller buffer too small */
    SO_ERR_SIZE_CHANGED = 5, /* image size differs from the first frame (reference: "WILL BREAK") */
    SO_ERR_NUMERIC = 6,      /* BA: linear solve failed in every LM trial */
    SO_ERR_TIMEOUT = 7       /* exchange: the collective of a tick did not complete within the budget; the handle is dead */
};
const char* so_status_string(int status);
/* last HIP error text on the calling thread ("" if none) */
const char* so_last_error(void);
/* number of visible HIP devices (0 when there is no GPU); never throws */
int so_device_count(void);
/* The host CPUs next to a device, as a cpulist ("0-7,128-135").  slot < 0: the whole NUMA node the device's PCI function
 * hangs off (/sys/bus/pci/devices/<bus id>/numa_node, /sys/devices/system/node/node<N>/cpulist).  slot >= 0: ONE group
 * of that node's CPUs that share a last-level cache (cache/index3/shared_cpu_list: a CCD of 8 cores on the EPYC hosts),
 * group number slot modulo the number of groups - so that the threads feeding a GPU (tracking, local mapping, the
 * runtime's helpers) sit behind one L3 and next to the device, and several agents get different groups.  Measured with
 * bench.py on a two-socket EPYC 9575F box: unpinned 2.21-2.25 k frames/s, the whole node the same, one CCD 2.34-2.35 k,
 * and a third of the run-to-run spread.  The replay loop of bench.py pins its two threads this way (SWARMORB_NO_PIN=1:
 * leave placement to the OS).  SO_ERR_NO_DEVICE if the device does not exist, SO_ERR_INVALID_ARG for a null / too small
 * buffer, SO_ERR_NUMERIC when the kernel does not say (single-node hosts report node -1: nothing to pin to).  No
 * reference counterpart (the reference leaves thread placement to the OS). */
int so_device_host_cpus(int device, int slot, char* cpulist, int capacity);
/* By default the extractors created by one thread share one HIP stream and its matchers / frame contexts another (the
 * per-frame path of ONE agent is a chain: more streams only spread it over hardware queues).  A thread that drives
 * SEVERAL agents in lockstep wants their launches to overlap: after so_runtime_private_streams(1) every extractor and
 * matcher created (by any thread) gets a stream of its own. */
int so_runtime_private_streams(int enabled);
/* Diagnostic load: `workgroups` workgroups that pin `lds_bytes` of LDS each (152 KB = a whole CU) and busy-wait for
 * `milliseconds`, on a stream of their own; returns as soon as the launch is queued.  What a second process on the GPU
 * looks like to the single-launch bundle-adjustment solves, whose workgroups must all be resident (tests). */
int so_runtime_occupy(int device, int workgroups, int lds_bytes, int milliseconds);

/* ------------------------------------------------------------------------------------------------
 * ORB extractor  — replaces ORB_SLAM2::ORBextractor (code/include/ORBextractor.h:49-127,
 * code/src/ORBextractor.cc:340-855) incl. cuda::GpuFast / IC_Angle / GpuOrb
 * (code/src/cuda/Fast_gpu.cu, Orb_gpu.cu) and the cv::cuda resize/border/blur calls.
 * ---------------------------------------------------------------------------------------------- */
/* identical 28-byte layout to cv::KeyPoint {pt.x, pt.y, size, angle, response, octave, class_id} */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} so_keypoint;

typedef struct {
    int32_t nfeatures;   /* ORBextractor ctor args, code/include/ORBextractor.h:54 */
    float scale_factor;
    int32_t nlevels;     /* <= SO_MAX_LEVELS */
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t device;      /* HIP device ordinal (one agent per GPU) */
} so_extractor_config;

#define SO_MAX_LEVELS 8
#define SO_FAST_CAP 10000 /* GpuFast maxKeypoints, code/include/cuda/Fast.hpp:32 */

typedef struct so_extractor so_extractor;

/* ORBextractor::ORBextractor (code/src/ORBextractor.cc:340-405) */
int so_extractor_create(const so_extractor_config* cfg, so_extractor** out);
void so_extractor_destroy(so_extractor* ex);

/* Output capacity the caller must provide: nfeatures + 3*nlevels (the quadtree may overshoot its
 * per-level quota by up to 3, code/src/ORBextractor.cc:656-661). */
int so_extractor_capacity(const so_extractor* ex);

/* Asynchronous form of operator(): so_extractor_submit[_device] enqueues the whole frame on the extractor's stream
 * and returns; so_extractor_collect waits for it and hands out the keypoints / descriptors.  Between the two the
 * calling thread is free — the tracking thread submits frame t+1 before it matches and optimises frame t, so the
 * extraction (which depends on nothing but the image) runs under the matcher / PoseOptimization kernels.  One frame
 * in flight per extractor; the image must stay valid until collect.  Results are identical to so_extractor_run. */
int so_extractor_submit(so_extractor* ex, const uint8_t* image, int width, int height, int stride);

/* Several extractors, ONE chain of launches (in-kernel batching of agents that share a GPU): so_extractor_group_submit is
 * so_extractor_submit on every member with its own image - each member is collected as usual, results are identical -
 * but every kernel of the chain runs once, with the member as one more grid dimension.  A lone frame leaves the GPU
 * nearly idle (one workgroup per pyramid level in the quadtree, ~1000 small workgroups elsewhere), so n frames take
 * hardly longer than one.  Members: same configuration, same device, device quadtree path; images: tightly packed,
 * device-visible (pinned host memory or device memory), all of one size; at most SO_EXTRACTOR_GROUP_MAX members.  The
 * group does not own its members; destroy it before them.  images[i] == NULL: member i sits this chain out (nothing of it
 * is read, written or expected back - its earlier frame may still be uncollected); at least one member must take part. */
#define SO_EXTRACTOR_GROUP_MAX 64
typedef struct so_extractor_group so_extractor_group;
int so_extractor_group_create(so_extractor* const* members, int n, so_extractor_group** out);
void so_extractor_group_destroy(so_extractor_group* group);
int so_extractor_group_submit(so_extractor_group* group, const uint8_t* const* images, int width, int height, int stride);
int so_extractor_submit_device(so_extractor* ex, const uint8_t* d_image, int width, int height, int stride);
int so_extractor_collect(so_extractor* ex, so_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out);
/* collect in two steps: so_extractor_wait blocks until the frame is done and tells the keypoint count; the results stay
 * in the context until so_extractor_collect copies them out (no further wait).  Lets the caller start work that only
 * needs the frame on the DEVICE (so_dframe_wait) before spending time on the host copies. */
int so_extractor_wait(so_extractor* ex, int* n_out);

/* DistributeOctTree (code/src/ORBextractor.cc:534-744) placement after the first frame sized the context:
 * 1 = HIP kernel (one workgroup per level, whole tree in LDS; one host sync per frame), 0 = host tree
 * (a level quota above 1020 keypoints or more than 4 root cells, or SWARMORB_HOST_QUADTREE set when the
 * context was sized).  Both produce identical output. */
int so_extractor_quadtree_on_device(const so_extractor* ex);

/* ORBextractor::operator() (code/src/ORBextractor.cc:746-819): CV_8UC1 host image in, keypoints
 * (level-0 pixel units, levels concatenated 0..n-1) and 32-byte descriptors out.  An empty image
 * (NULL / w<=0 / h<=0) returns SO_OK with *n_out = 0 and outputs untouched. */
int so_extractor_run(so_extractor* ex, const uint8_t* image, int width, int height, int stride,
                     so_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out);
/* Same, image already resident in device memory (HBM) on the extractor's device. */
int so_extractor_run_device(so_extractor* ex, const uint8_t* d_image, int width, int height, int stride,
                            so_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out);

/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (code/include/ORBextractor.h:64-87) + mnFeaturesPerLevel; each array has nlevels entries. */
int so_extractor_tables(const so_extractor* ex, float* scale, float* inv_scale, float* sigma2,
                        float* inv_sigma2, int32_t* features_per_level);

/* Stage outputs of the LAST run, for stage-wise parity tests (not used by the SLAM threads):
 *  - level image `level` (un-blurred), tightly packed w*h bytes (mvImagePyramid, ORBextractor.h:90)
 *  - FAST candidates of `level` before the quadtree (ROI-relative x,y; integer score), raster order
 *    (GpuFast::joinDetectAsync output, code/src/cuda/Fast_gpu.cu:379-388). */
int so_extractor_level_size(const so_extractor* ex, int level, int* w, int* h);
int so_extractor_get_level(so_extractor* ex, int level, uint8_t* out, int out_bytes);
int so_extractor_get_candidates(so_extractor* ex, int level, int16_t* xs, int16_t* ys, uint8_t* scores,
                                int capacity, int* n_out);

/* Per-stage GPU timing with HIP events on the extractor's own stream (off by default).
 * GPU stages (HIP events): 0 upload+pyramid, 1 FAST score+NMS (high threshold), 2 FAST low-threshold pass +
 * row counts, 3 candidate emission, 4 angle+blur+rBRIEF.  Host stages (steady clock): 5 whole call,
 * 6 enqueue of phase 1, 7 wait for the candidates, 8 quadtree, 9 phase 2 (H2D + describe + D2H + wait),
 * 10 keypoint assembly.  All in ms. */
#define SO_EXTRACTOR_N_STAGES 11
int so_extractor_set_profiling(so_extractor* ex, int enabled);
int so_extractor_get_profile(so_extractor* ex, float* ms_per_stage /* [SO_EXTRACTOR_N_STAGES] */);

/* PMC calibration helper (tools/pmc_calibrate.py): streams n_bytes of device memory with one aligned 4-byte
 * load per lane (the access shape of the FAST tile staging) so rocprofv3's FETCH_SIZE can be compared with a
 * known byte count.  d_sink_4k: 4 KiB of device scratch.  Not part of the SLAM path. */
int so_debug_stream_read(const void* d_src, unsigned long long n_bytes, void* d_sink_4k);

/* ------------------------------------------------------------------------------------------------
 * Hamming matcher — replaces the data-parallel part of ORB_SLAM2::ORBmatcher (code/include/ORBmatcher.h:41-83,
 * code/src/ORBmatcher.cc) for the tracking thread: candidate gathering (Frame::GetFeaturesInArea,
 * code/src/Frame.cc:377-431), DescriptorDistance (ORBmatcher.cc:1511-1525) and best/second selection run
 * on the GPU for ALL queries of a call at once; the order-dependent greedy resolve, ratio tests and the
 * rotation histogram (ComputeThreeMaxima, ORBmatcher.cc:1475-1506) run on the host inside the library,
 * exactly in the reference's order.  Monocular only (mvuRight < 0), as every SwarmMap binary is.
 * Object-graph side effects (F.mvpMapPoints[idx] = pMP ...) stay in the caller's adapter.
 * ---------------------------------------------------------------------------------------------- */
typedef struct so_matcher so_matcher;

/* The parts of ORB_SLAM2::Frame the matcher reads (code/include/Frame.h), flattened. */
typedef struct {
    int32_t n;               /* N */
    const float* x;          /* mvKeysUn[i].pt.x */
    const float* y;          /* mvKeysUn[i].pt.y */
    const int32_t* octave;   /* mvKeysUn[i].octave */
    const float* angle;      /* mvKeysUn[i].angle (degrees); may be NULL when orientation is not checked */
    const uint8_t* desc;     /* mDescriptors, n x 32 */
    const uint8_t* excluded; /* 1 iff mvpMapPoints[i] && mvpMapPoints[i]->Observations() > 0 on entry; may be NULL */
    float min_x, max_x, min_y, max_y; /* mnMinX, mnMaxX, mnMinY, mnMaxY */
    float grid_inv_w, grid_inv_h;     /* mfGridElementWidthInv, mfGridElementHeightInv */
    const float* scale_factors;       /* mvScaleFactors */
    int32_t nlevels;
    /* KeyFrame quirk (code/include/KeyFrame.h:220: `const int mnMinX, mnMinY, mnMaxX, mnMaxY`): a KeyFrame truncates the
     * Frame's float bounds to int, so its IsInImage / GetFeaturesInArea (code/src/KeyFrame.cc:779-818) work with the
     * truncated values - pass those as min_x .. max_y - while the grid it searches was copied from the Frame
     * (KeyFrame.cc:66-72), i.e. filled by Frame::PosInGrid with the float origin.  has_grid_origin != 0: the cells were
     * assigned with (grid_min_x, grid_min_y) instead of (min_x, min_y).  A keyframe whose grid was rebuilt by
     * KeyFrame::AssignFeaturesToGrid (KeyFrame.cc:1012-1038, after deserialisation) and every Frame leave it 0. */
    int32_t has_grid_origin;
    float grid_min_x, grid_min_y;
} so_frame_view;

/* Pinhole intrinsics + distortion of a Frame / KeyFrame (mK, mDistCoef). */
typedef struct so_camera {
    float fx, fy, cx, cy;     /* mK */
    float k1, k2, p1, p2, k3; /* mDistCoef; k1 == 0 means "no distortion" exactly like Frame.cc:456,490 */
} so_camera;

int so_matcher_create(int device, so_matcher** out);
void so_matcher_destroy(so_matcher* m);

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) — code/src/ORBmatcher.cc:44-121.
 * in_view[i] = pMP->mbTrackInView && !pMP->isBad(); proj_x/proj_y/view_cos/pred_level = mTrackProjX/Y,
 * mTrackViewCos, mnTrackScaleLevel; mp_desc = GetDescriptor() (n_mp x 32); mp_has_obs[i] = Observations() > 0.
 * nn_ratio = mfNNratio.  Out: kp_to_mp[k] = map point bound to keypoint k by this call, or -1. */
int so_search_by_projection_mappoints(so_matcher* m, const so_frame_view* F, int32_t n_mp, const uint8_t* in_view,
                                      const float* proj_x, const float* proj_y, const float* view_cos,
                                      const int32_t* pred_level, const uint8_t* mp_desc, const uint8_t* mp_has_obs,
                                      float th, float nn_ratio, int32_t* kp_to_mp, int32_t* nmatches);

/* ORBmatcher::SearchByProjection(Frame& cur, const Frame& last, th, bMono=true) — ORBmatcher.cc:1223-1354,
 * from the projection onwards: valid[i] = pMP && !mvbOutlier[i] && invzc >= 0 && (u,v) inside the bounds.
 * Out: kp_to_last[k] = index in the last frame whose map point is bound to keypoint k, or -1. */
int so_search_by_projection_lastframe(so_matcher* m, const so_frame_view* cur, int32_t n_last, const uint8_t* valid,
                                      const float* u, const float* v, const int32_t* last_octave,
                                      const float* last_angle, const uint8_t* mp_desc, const uint8_t* mp_has_obs,
                                      float th, int check_orientation, int32_t* kp_to_last, int32_t* nmatches);

/* ORBmatcher::SearchForInitialization — ORBmatcher.cc:375-479.  prev_matched: n1 x 2 floats (vbPrevMatched),
 * updated in place; matches12: n1 ints (vnMatches12). */
int so_search_for_initialization(so_matcher* m, const so_frame_view* F1, const so_frame_view* F2,
                                 float* prev_matched, int window, float nn_ratio, int check_orientation,
                                 int32_t* matches12, int32_t* nmatches);

/* Building block exposed for the other matcher routines and for tests: for every query the K best
 * candidates of `F` inside the GetFeaturesInArea window (u,v,r,[min_level,max_level]), ordered as the
 * reference's sequential scan would rank them (distance, then grid traversal order).
 * limit (optional, n ints): candidate k only competes when dist < limit[k] (0 excludes it).
 * out_idx/out_dist: nq x K (idx -1 / dist 256 padding); out_count: candidates inside each window. */
int so_matcher_topk(so_matcher* m, const so_frame_view* F, const int32_t* limit, int32_t nq, const float* u,
                    const float* v, const float* r, const int32_t* min_level, const int32_t* max_level,
                    const uint8_t* active, const uint8_t* qdesc, int32_t K, int32_t* out_idx, int32_t* out_dist,
                    int32_t* out_count);

/* Brute-force best / second-best Hamming match of every row of A (na x 32) against all rows of B (nb x 32);
 * ties go to the lowest index in B.  Used for the cross-agent keyframe search after the RCCL descriptor
 * all-gather (replaces the BoW candidate query of code/src/AgentMediator.cc:177-191). */
int so_hamming_top2(so_matcher* m, const uint8_t* A, int32_t na, const uint8_t* B, int32_t nb, int32_t* best_idx,
                    int32_t* best_dist, int32_t* second_dist);
/* Same with A and B already in device memory (e.g. the all-gathered descriptor slots). */
int so_hamming_top2_device(so_matcher* m, const uint8_t* d_A, int32_t na, const uint8_t* d_B, int32_t nb,
                           int32_t* best_idx, int32_t* best_dist, int32_t* second_dist);

/* ---- the remaining ORBmatcher routines (LocalMapping / loop closing / relocalisation), same division of labour:
 *      distances + candidate selection on the GPU, order-dependent resolve in the library, object-graph side
 *      effects in the caller. ---- */

/* DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened: nodes ascending by id; node k owns
 * idx[off[k] .. off[k+1]) in the order DBoW2 stored the feature indices. */
typedef struct {
    int32_t n_nodes;
    const int32_t* node_id;
    const int32_t* off; /* n_nodes + 1 */
    const int32_t* idx;
} so_featvec;

/* ORBmatcher::SearchByBoW.  variant 0 = (KeyFrame* pKF, Frame& F, vpMapPointMatches), code/src/ORBmatcher.cc:150-262;
 * variant 1 = (KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12), :481-597.  Set 1 is the keyframe whose map points
 * are being matched (valid1[i] = pMP && !pMP->isBad()); set 2 the target (variant 1: valid2[i] likewise; variant
 * 0: ignored, may be NULL).  match_of_2[k2] = index in set 1 bound to target feature k2 (variant 0's
 * vpMapPointMatches), match_of_1[k1] = target feature of k1 (variant 1's vpMatches12); either may be NULL.  The two
 * arrays always describe the same set of pairs: a pair dropped by the rotation histogram leaves both. */
int so_search_by_bow(so_matcher* m, int variant, int32_t n1, const uint8_t* desc1, const float* angle1,
                     const uint8_t* valid1, const so_featvec* fv1, int32_t n2, const uint8_t* desc2,
                     const float* angle2, const uint8_t* valid2, const so_featvec* fv2, float nn_ratio,
                     int check_orientation, int32_t* match_of_2, int32_t* match_of_1, int32_t* nmatches);

/* ORBmatcher::SearchForTriangulation (monocular) — ORBmatcher.cc:599-749 with CheckDistEpipolarLine :131-148.
 * free1/free2[i] = !pKF->GetMapPoint(i); F12 row-major 3x3; (ex, ey) the epipole in image 2 (:607-613);
 * scale_factors2 / level_sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2 (nlevels2 entries, <= 8).
 * matches12[i1] = index in keyframe 2 or -1 (vMatchedPairs). */
int so_search_for_triangulation(so_matcher* m, int32_t n1, const float* x1, const float* y1, const float* angle1,
                                const uint8_t* desc1, const uint8_t* free1, const so_featvec* fv1, int32_t n2,
                                const float* x2, const float* y2, const int32_t* octave2, const float* angle2,
                                const uint8_t* desc2, const uint8_t* free2, const so_featvec* fv2, const float* F12,
                                float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                int32_t nlevels2, int check_orientation, int32_t* matches12, int32_t* nmatches);

/* Core of ORBmatcher::Fuse (both overloads, ORBmatcher.cc:751-891 and :893-1009) and of each pass of
 * SearchBySim3 (:1011-1221): for every valid query the first minimum-distance keypoint of
 * KeyFrame::GetFeaturesInArea(u, v, radius) (code/src/KeyFrame.cc:779-814) with octave in [pred_level-1,
 * pred_level]; chi2_gate != 0 additionally requires e2 * inv_sigma2[octave] <= 5.99 (:853-860).  Queries are
 * independent; thresholds (TH_LOW / TH_HIGH) and Replace/AddObservation stay with the caller.
 * best_idx[i] = keypoint or -1, best_dist[i] = distance (256 when none). */
int so_search_window_best(so_matcher* m, const so_frame_view* KF, int32_t nq, const uint8_t* valid, const float* u,
                          const float* v, const float* radius, const int32_t* pred_level, const uint8_t* qdesc,
                          int chi2_gate, const float* inv_sigma2, int32_t* best_idx, int32_t* best_dist);

/* Sequential greedy window search: SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) — :264-373
 * (levels [pred-1, pred], max_dist TH_LOW, no orientation) — and SearchByProjection(Frame&, KeyFrame*,
 * sAlreadyFound, th, ORBdist) — :1356-1473 (levels [pred-1, pred+1], max_dist ORBdist, orientation).
 * A keypoint is skipped when F->excluded[k] (bound on entry) or when an earlier query of this call took it.
 * kp_to_query[k] = query bound to keypoint k, or -1. */
int so_search_window_greedy(so_matcher* m, const so_frame_view* F, int32_t nq, const uint8_t* valid, const float* u,
                            const float* v, const float* radius, const int32_t* min_level, const int32_t* max_level,
                            const uint8_t* qdesc, const float* q_angle, int32_t max_dist, int check_orientation,
                            int32_t* kp_to_query, int32_t* nmatches);

/* ---- Fuse, SearchBySim3 and the keyframe-side SearchByProjection overloads with the projection on the device ----
 * The five routines below take the map points as the reference's loops read them and run, in one enqueue without a host
 * hop in between, (1) the projection + gating statements of the routine for every map point (thread per point:
 * Rcw * X + tcw or the Scw / Sim3 chain, positive depth, KeyFrame::IsInImage, Get{Min,Max}DistanceInvariance, the
 * viewing-angle gate PO.dot(Pn) < 0.5 * dist3D, MapPoint::PredictScale, radius = th * mvScaleFactors[level]) and
 * (2) the window search of so_search_window_best / _greedy over the produced queries.  cv::Mat arithmetic follows the
 * conventions of oracle/project_oracle.h (one GEMM = double accumulation + one rounding).  Object-graph side effects
 * (AddObservation / AddMapPoint / Replace, vpReplacePoint, vpMatched, mvpMapPoints) stay with the caller, which walks
 * the returned bindings in map-point order exactly as the reference's loop does (host/glue/ORBmatcher_glue.cc). */

/* The MapPoint fields these searches read (code/include/MapPoint.h), one entry per element of vpMapPoints. */
typedef struct so_mappoint_view {
    int32_t n;
    const float* Xw;       /* n x 3: GetWorldPos() (GetGlobalPos() where the routine says so) */
    const float* normal;   /* n x 3: GetNormal(); may be NULL for SearchBySim3 / SearchByProjection(Frame, KeyFrame) */
    const float* max_dist; /* mfMaxDistance (GetMaxDistanceInvariance() = 1.2f * this) */
    const float* min_dist; /* mfMinDistance (GetMinDistanceInvariance() = 0.8f * this) */
    const uint8_t* desc;   /* n x 32: GetDescriptor() */
    const uint8_t* valid;  /* the routine's object-graph gates (pMP && !isBad() && !IsInKeyFrame ...); NULL = all 1 */
} so_mappoint_view;

/* Optional diagnostics of stage (1), per map point; any pointer (or the struct) may be NULL.  u / v / radius / level
 * are 0 where active[i] = 0. */
typedef struct so_window_queries {
    uint8_t* active;
    float* u;
    float* v;
    float* radius;  /* th * mvScaleFactors[nPredictedLevel] */
    int32_t* level; /* nPredictedLevel */
} so_window_queries;

/* ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th) — code/src/ORBmatcher.cc:751-891, up to
 * the map side effects.  Tcw12 = [pKF->GetRotation() | GetTranslation()] row-major, log_scale_factor =
 * pKF->mfLogScaleFactor, inv_level_sigma2 = pKF->mvInvLevelSigma2 (KF->nlevels entries), cam: fx fy cx cy of pKF.
 * best_idx[i] = bestIdx where bestDist <= TH_LOW (:873) else -1; best_dist[i] = bestDist (256: no candidate);
 * *n_fused = number of i with best_idx[i] >= 0 (the reference's return value when no point turns bad meanwhile). */
int so_fuse(so_matcher* m, const so_frame_view* KF, const so_camera* cam, const float* Tcw12, float log_scale_factor,
            const float* inv_level_sigma2, const so_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist,
            int32_t* n_fused, const so_window_queries* queries_out);
/* ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) — :893-1009.  Scw12: rows 0-2 of Scw. */
int so_fuse_sim3(so_matcher* m, const so_frame_view* KF, const so_camera* cam, const float* Scw12, float log_scale_factor,
                 const so_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist, int32_t* n_fused,
                 const so_window_queries* queries_out);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) — :1011-1221.  mp1 = pKF1->GetMapPointMatches()
 * (n = N1 keypoints of KF1; valid[i1] = pMP && !vbAlreadyMatched1[i1] && !isBad(), :1040-1061), mp2 likewise with
 * vbAlreadyMatched2; T1w12 / T2w12 the keyframes' poses; R12 3x3 row-major, t12 3; cam = pKF1's intrinsics (used for
 * both directions, :1013-1016).  match12[i1] = idx2 where both directions agree (:1205-1218) else -1. */
int so_search_by_sim3(so_matcher* m, const so_frame_view* KF1, const so_frame_view* KF2, const so_camera* cam,
                      const float* T1w12, const float* T2w12, float s12, const float* R12, const float* t12,
                      float log_scale_factor1, float log_scale_factor2, const so_mappoint_view* mp1,
                      const so_mappoint_view* mp2, float th, int32_t* match12, int32_t* n_found,
                      const so_window_queries* queries1_out, const so_window_queries* queries2_out);
/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, int th) — :264-373.
 * KF->excluded[k] = vpMatched[k] != NULL on entry; mp->valid[i] = !isBad() && !spAlreadyFound.count(pMP).
 * kp_to_point[k] = index into vpPoints the call binds to keypoint k (vpMatched[k] = vpPoints[...]) or -1. */
int so_search_by_projection_sim3(so_matcher* m, const so_frame_view* KF, const so_camera* cam, const float* Scw12,
                                 float log_scale_factor, const so_mappoint_view* mp, int th, int32_t* kp_to_point,
                                 int32_t* nmatches, const so_window_queries* queries_out);
/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist, bGlobal) —
 * :1356-1473.  mp = pKF->GetMapPointMatches() (valid[i] = pMP && !isBad() && !sAlreadyFound.count(pMP); Xw = GetGlobalPos
 * when bGlobal), mp_angle[i] = pKF->mvKeysUn[i].angle, Tcw12 = CurrentFrame.mTcw, F->excluded[k] =
 * CurrentFrame.mvpMapPoints[k] != NULL on entry.  kp_to_point[k] = i bound to keypoint k (after the rotation
 * histogram) or -1. */
int so_search_by_projection_keyframe(so_matcher* m, const so_frame_view* F, const so_camera* cam, const float* Tcw12,
                                     float log_scale_factor, const so_mappoint_view* mp, const float* mp_angle, float th,
                                     int32_t orb_dist, int check_orientation, int32_t* kp_to_point, int32_t* nmatches,
                                     const so_window_queries* queries_out);

/* Batched calls.  LocalMapping handles a new keyframe with one SearchForTriangulation per neighbour (<= 20,
 * code/src/LocalMapping.cc:197-246) and one Fuse per neighbour plus one back (:451-481): some 40 independent searches of
 * ~1000 queries each.  Between so_matcher_batch_begin and so_matcher_batch_end the calls so_search_for_triangulation,
 * so_fuse and so_fuse_sim3 on this handle only stage their inputs and return SO_OK; so_matcher_batch_end enqueues all of
 * them as ONE staging copy, ONE projection launch and ONE search launch (the job is a grid dimension), waits once and
 * then writes every call's outputs - the arrays the calls were given, which must stay valid until then (the inputs are
 * copied at call time, except angle1 / angle2 of so_search_for_triangulation, which the resolve reads).  Results are
 * identical to the same calls made one by one.  At most 64 calls are held; further ones flush the batch first (their
 * predecessors' outputs are then complete early).  Any other matcher call inside a batch fails with
 * SO_ERR_INVALID_ARG.  so_matcher_last_kernel_ms / _last_stats after so_matcher_batch_end cover the whole batch. */
int so_matcher_batch_begin(so_matcher* m);
int so_matcher_batch_end(so_matcher* m);
/* Leaves a batch without running it: the deferred calls are dropped (their output arrays stay untouched unless an automatic
 * flush already completed them) and the handle takes plain calls again.  For the error path of a caller whose call inside
 * the batch failed; a handle that is not batching is left alone.  so_matcher_batch_end leaves the batch on every path,
 * failures included. */
int so_matcher_batch_abort(so_matcher* m);

/* HBM-resident keyframes.  A keyframe is searched again and again - by every later keyframe's SearchForTriangulation
 * and Fuse while it is among the <= 20 covisible neighbours (code/src/LocalMapping.cc:197-246, 451-481) - and its
 * keypoints never change: so_kframe_create uploads them ONCE, in GetFeaturesInArea order and (with a feature vector)
 * in vocabulary-node order; the two calls below then stage only the queries.  Same results as so_fuse /
 * so_search_for_triangulation with the same keyframe as so_frame_view; both may be part of a batch.
 * KF: the keyframe's view (bounds as a KeyFrame has them, see so_frame_view); fv / level_sigma2 (mvLevelSigma2): needed
 * for so_search_for_triangulation_kframe only, may be NULL.  The handle only lends its staging; the keyframe may be
 * used with any so_matcher of the same device and must outlive the batches that reference it. */
typedef struct so_kframe so_kframe;
int so_kframe_create(so_matcher* m, const so_frame_view* KF, const so_featvec* fv, const float* level_sigma2, so_kframe** out);
void so_kframe_destroy(so_kframe* k);
int so_fuse_kframe(so_matcher* m, const so_kframe* KF, const so_camera* cam, const float* Tcw12, float log_scale_factor,
                   const float* inv_level_sigma2, const so_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist,
                   int32_t* n_fused, const so_window_queries* queries_out);
/* so_fuse_kframe with the map points read where they already are: rows slots[i] of a device-resident map table
 * (so_map below: mWorldPos, mNormalVector, mfMax/MinDistance, mDescriptor by slot).  A keyframe's Fuse calls then stage
 * 5 bytes per map point (slot + valid) instead of 65.  slots[i] outside [0, so_map_size) makes point i inactive; valid
 * may be NULL (= all 1).  The map may be written by another thread meanwhile as long as rows this call names are not
 * (appends are fine; the call holds the table in place while its kernels run).  Same results as so_fuse_kframe with the
 * rows' values as arrays.  Batchable. */
typedef struct so_map so_map;
int so_fuse_kframe_map(so_matcher* m, const so_kframe* KF, const so_camera* cam, const float* Tcw12, float log_scale_factor,
                       const float* inv_level_sigma2, const so_map* map, int32_t n, const int32_t* slots, const uint8_t* valid,
                       float th, int32_t* best_idx, int32_t* best_dist, int32_t* n_fused, const so_window_queries* queries_out);
/* keyframe 2 resident (its angles, octaves, descriptors, feature vector, scale tables were given at creation); free2[i]
 * = !pKF2->GetMapPoint(i) at the time of the call */
int so_search_for_triangulation_kframe(so_matcher* m, int32_t n1, const float* x1, const float* y1, const float* angle1,
                                       const uint8_t* desc1, const uint8_t* free1, const so_featvec* fv1, const so_kframe* kf2,
                                       const uint8_t* free2, const float* F12, float ex, float ey, int check_orientation,
                                       int32_t* matches12, int32_t* nmatches);

/* CreateNewMapPoints' searches of ONE new keyframe against its neighbours (code/src/LocalMapping.cc:197-246: up to twenty
 * SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, ...) calls) with BOTH sides resident: kf1 was created with its feature
 * vector like the neighbours, and the queries - which of its features have no map point, the vocabulary node each one
 * searches, its epipolar line in the neighbour's image - are built on the device by the batch's first launch; per neighbour
 * ~6 KB go up instead of ~40 KB of query records.  free1[i] = !kf1->GetMapPoint(i) for all of them (the caller applies what
 * changes between neighbours when it walks the results in order); per neighbour: the keyframe, free2 as
 * in so_search_for_triangulation_kframe, F12 (row-major 3 x 3), the epipole, the outputs.  Results identical to n_neighbours
 * calls of so_search_for_triangulation_kframe with the same arguments.  Batchable (inside so_matcher_batch_begin / _end the
 * outputs are complete after _end); outside a batch it is one. */
typedef struct so_tri_neighbour {
    const so_kframe* kf2;
    const uint8_t* free2;
    float F12[9];
    float ex, ey;
    int32_t* matches12; /* kf1->n entries */
    int32_t* nmatches;
} so_tri_neighbour;
int so_search_for_triangulation_kframes(so_matcher* m, const so_kframe* kf1, const uint8_t* free1, int32_t n_neighbours,
                                        const so_tri_neighbour* neighbours, int check_orientation);

/* ---- the local-mapping thread's two per-point loops between the matcher and local BA (widening beyond SURVEY 8f) ----
 * What LocalMapping::CreateNewMapPoints reads of a keyframe. */
typedef struct so_tri_keyframe {
    float Tcw[12];               /* [GetRotation() | GetTranslation()] row-major */
    float fx, fy, cx, cy, invfx, invfy;
    const float* scale_factors;  /* mvScaleFactors */
    const float* level_sigma2;   /* mvLevelSigma2 */
    int32_t nlevels;             /* <= 8 */
} so_tri_keyframe;
/* The per-match body of LocalMapping::CreateNewMapPoints (code/src/LocalMapping.cc:263-420, monocular), for the matches
 * of mpCurrentKeyFrame (kf1) with n_kf2 neighbours in ONE launch (thread per match): parallax of the rays (0 < cos <
 * 0.9998), linear triangulation by cv::SVD::compute of the 4 x 4 system (OpenCV's one-sided Jacobi, restated in
 * oracle/mapping_oracle.h), positive depth in both cameras, reprojection error <= 5.991 sigma2 in both, scale
 * consistency with ratio_factor = 1.5f * mfScaleFactor.  Match k pairs keypoint (xy1, octave1)[k] of kf1 with (xy2,
 * octave2)[k] of kf2[kf2_of_match[k]] (mvKeysUn).  ok[k] = 1 where the reference creates a MapPoint, x3D[3k..] its
 * position (untouched otherwise).  The baseline / median-depth test per neighbour (:228-241), `new MapPoint`,
 * AddObservation / AddMapPoint and the bookkeeping stay with the caller. */
int so_triangulate_matches(so_matcher* m, const so_tri_keyframe* kf1, int32_t n_kf2, const so_tri_keyframe* kf2, float ratio_factor,
                           int32_t n, const int32_t* kf2_of_match, const float* xy1, const int32_t* octave1, const float* xy2,
                           const int32_t* octave2, uint8_t* ok, float* x3D);
/* so_triangulate_matches + what CreateNewMapPoints does with each new point right away (LocalMapping.cc:403-414:
 * AddObservation x 2, ComputeDistinctiveDescriptors, UpdateNormalAndDepth): mNormalVector, mfMaxDistance and
 * mfMinDistance of the accepted matches' points in the SAME launch - two observations (kf1, then the neighbour), kf1 the
 * reference keyframe, level = octave1[k].  Same values as so_update_normal_and_depth on those inputs; normal (3 per
 * match), max_dist, min_dist are written where ok[k] = 1 only.  One launch and one wait instead of two. */
int so_triangulate_new_points(so_matcher* m, const so_tri_keyframe* kf1, int32_t n_kf2, const so_tri_keyframe* kf2, float ratio_factor,
                              int32_t n, const int32_t* kf2_of_match, const float* xy1, const int32_t* octave1, const float* xy2,
                              const int32_t* octave2, uint8_t* ok, float* x3D, float* normal, float* max_dist, float* min_dist);
/* MapPoint::UpdateNormalAndDepth (code/src/MapPoint.cc:413-465) for a batch of map points (thread per point): point p is
 * observed from the camera centres obs_Ow[offsets[p] .. offsets[p + 1]) (3 floats each, in the order the caller walks
 * mObservations - the reference's own order is the pointer order of a std::map and differs from run to run); ref_Ow =
 * pRefKF->GetCameraCenter(), ref_level_scale = pRefKF->mvScaleFactors[octave of the point's keypoint in pRefKF],
 * ref_last_scale = pRefKF->mvScaleFactors[mnScaleLevels - 1].  normal / max_dist / min_dist (mNormalVector,
 * mfMaxDistance, mfMinDistance) are in / out: a point without observations keeps its values. */
int so_update_normal_and_depth(so_matcher* m, int32_t n_points, const int32_t* offsets, const float* obs_Ow, const float* Xw,
                               const float* ref_Ow, const float* ref_level_scale, const float* ref_last_scale, float* normal,
                               float* max_dist, float* min_dist);

/* The same with the observers given by index: observation k of point p (offsets[p] <= k < offsets[p + 1]) is seen from
 * keyframe obs_kf[k], whose camera centre is kf_Ow[3 obs_kf[k] ..]; the reference keyframe of point p is ref_kf[p].  What the
 * write-back of local bundle adjustment has at hand (Optimizer.cc:729-737: a window's points, observed from a few dozen
 * keyframes): 4 bytes per observation cross PCIe instead of 12.  Same results as so_update_normal_and_depth on the expanded
 * arrays. */
int so_update_normal_and_depth_indexed(so_matcher* m, int32_t n_points, const int32_t* offsets, const int32_t* obs_kf, int32_t n_kf,
                                       const float* kf_Ow, const float* Xw, const int32_t* ref_kf, const float* ref_level_scale,
                                       const float* ref_last_scale, float* normal, float* max_dist, float* min_dist);

/* MapPoint::ComputeDistinctiveDescriptors (code/src/MapPoint.cc:323-392) for a batch of map points (SURVEY 8f rank
 * 4): point p owns descriptors [offsets[p], offsets[p+1]) (the rows the reference collects from its observing
 * keyframes, in map order); best_idx[p] = index within the point's own list of the descriptor with the least
 * median Hamming distance to the others (first one on ties, median = sorted row [int(0.5 (N-1))] with the zero
 * self-distance included), -1 for a point without descriptors; best_median may be NULL.  At most 512
 * observations per point. */
int so_distinctive_descriptors(so_matcher* m, int32_t n_points, const int32_t* offsets, const uint8_t* descriptors,
                               int32_t* best_idx, int32_t* best_median);

/* Allocates the staging of so_track_search_local_map / _last_frame for up to n_queries map points now instead of on
 * demand (a local map that grows keyframe by keyframe otherwise pays a pinned re-allocation, 0.1-0.3 ms, every time it
 * outgrows the slack).  Optional; the searches still grow the buffers when a call needs more. */
int so_matcher_reserve(so_matcher* m, int32_t n_queries);
/* HIP-event time (ms) of the kernels of the last matcher call on the matcher's stream (0 while the events are switched
 * off: so_matcher_set_profiling(m, 0) saves the two event records per search, ~2 us of host time each and a
 * timestamp packet on the queue; on by default). */
int so_matcher_last_kernel_ms(so_matcher* m, float* ms);
int so_matcher_set_profiling(so_matcher* m, int enabled);

/* Tracking searches the same frame twice in a row (SearchByProjection against the last frame, Tracking.cc:731, then
 * against the local map, :1006).  Calling this before the second search tells the handle that the next call's frame
 * view is the one of the previous call (same keypoints, descriptors and bounds; `excluded` may differ and is re-read):
 * the grid ordering and the candidate upload are skipped.  One-shot: it applies to the next search call only, and
 * is ignored if the handle no longer holds that frame. */
int so_matcher_reuse_frame(so_matcher* m);

/* Host-side view of the last matcher call: stats4[0] = ms spent enqueueing (staging copy + launch), [1] = ms
 * blocked in stream syncs, [2] = kernel launches (1 + exact re-runs of single queries), [3] = bytes staged
 * host -> device. */
int so_matcher_last_stats(so_matcher* m, double* stats4);

/* ------------------------------------------------------------------------------------------------
 * Frame post-processing between extractor and matcher — replaces, on flattened data,
 * Frame::UndistortKeyPoints (code/src/Frame.cc:454-486), Frame::ComputeImageBounds (:488-514),
 * Frame::AssignFeaturesToGrid + PosInGrid (:277-292, 431-443) and Frame::isInFrustum (:316-375) with
 * MapPoint::PredictScale / Get{Min,Max}DistanceInvariance (code/src/MapPoint.cc:466-485).
 * Conventions for the arithmetic the reference leaves to un-vendored OpenCV / libm: oracle/frame_oracle.h.
 * ------------------------------------------------------------------------------------------------ */
/* (so_camera is declared with the matcher's types above) */

typedef struct so_frame_ctx so_frame_ctx;
int so_frame_create(int device, so_frame_ctx** out);
void so_frame_destroy(so_frame_ctx* f);

/* The Frame constructor's steps after ExtractORB (Frame.cc:183-192, 230-274), one launch: xy (x0 y0 x1 y1 ...) ->
 * xy_un; bounds4 = {mnMinX, mnMaxX, mnMinY, mnMaxY} is computed when compute_bounds != 0 (first frame / after a
 * calibration change, Frame.cc:247-263) and read otherwise; grid outputs (all four non-NULL or all NULL):
 * cell_of[i] = x * 48 + y or -1 outside, cell_start[64*48+1], cell_items = the cells' keypoints in push_back
 * order, *n_inside.  At most 16384 keypoints. */
int so_frame_prepare(so_frame_ctx* f, const so_camera* cam, int32_t width, int32_t height, int compute_bounds,
                     int32_t n, const float* xy, float* xy_un, float* bounds4, int32_t* cell_of, int32_t* cell_start,
                     int32_t* cell_items, int32_t* n_inside);

/* Frame::isInFrustum for every local map point (Tracking::SearchLocalPoints, code/src/Tracking.cc:985-996):
 * in_view[i] = mbTrackInView; proj_x / proj_y / view_cos / pred_level (mTrackProjX, mTrackProjY, mTrackViewCos,
 * mnTrackScaleLevel) are written only where in_view[i] = 1, as the reference leaves them untouched otherwise.
 * max_dist / min_dist are the map points' mfMaxDistance / mfMinDistance (the 1.2 / 0.8 factors are applied inside). */
int so_frame_is_in_frustum(so_frame_ctx* f, const so_camera* cam, const float* bounds4, const float* Tcw12, int32_t n,
                           const float* Xw, const float* normal, const float* max_dist, const float* min_dist,
                           float viewing_cos_limit, float log_scale_factor, int32_t n_scale_levels, uint8_t* in_view,
                           float* proj_x, float* proj_y, float* view_cos, int32_t* pred_level);

/* ------------------------------------------------------------------------------------------------
 * Device-resident frame, map-point table and the two tracking searches over them.
 *
 * The host-array entry points above hand keypoints / descriptors back to the caller after every operator and take
 * them in again at the next one.  On the tracking thread that round trip (extractor -> Frame::UndistortKeyPoints /
 * AssignFeaturesToGrid -> ORBmatcher, code/src/Frame.cc:218-275 -> code/src/Tracking.cc:714-768, 964-1007) is pure
 * overhead: these entry points keep the frame in HBM from the image upload to the matcher's K-lists.  The host still
 * receives copies of the keypoints, undistorted positions and descriptors (Frame's members stay usable by the CPU
 * parts of SLAM), written by the kernels into host-mapped memory.  Results are identical to chaining
 * so_extractor_run -> so_frame_prepare -> so_frame_is_in_frustum -> so_search_by_projection_*.
 * ---------------------------------------------------------------------------------------------- */
typedef struct so_dframe so_dframe;

/* One device-resident Frame bound to an extractor (two alternate per agent: frame t+1 is extracted while frame t is
 * matched).  cam = mK / mDistCoef of the Frame constructor. */
int so_dframe_create(so_extractor* ex, const so_camera* cam, so_dframe** out);
void so_dframe_destroy(so_dframe* f);

/* Frame::Frame(imGray, ..., extractor, K, distCoef, ...) monocular (code/src/Frame.cc:218-275), asynchronous:
 * image upload (host pointer, pinned or pageable; so_dframe_submit_device: image already in HBM) -> ExtractORB ->
 * UndistortKeyPoints -> ComputeImageBounds -> AssignFeaturesToGrid -> the matcher's candidate layout, enqueued on the
 * extractor's stream; returns without waiting.  One frame in flight per extractor. */
int so_dframe_submit(so_dframe* f, const uint8_t* image, int width, int height, int stride);
/* so_dframe_submit for frame i on member i of an extractor group, all in one chain of launches (the Frame constructors'
 * kernel too: one workgroup per frame).  Every frame is waited for / collected as usual.  images[i] == NULL: member i sits
 * this chain out and frames[i] is ignored (agents that share a GPU but not a clock: so_fleet_run's elastic ticks). */
int so_dframe_group_submit(so_extractor_group* group, so_dframe* const* frames, const uint8_t* const* images, int width,
                           int height, int stride);
int so_dframe_submit_device(so_dframe* f, const uint8_t* d_image, int width, int height, int stride);
/* Waits for the frame and hands out the host copies: keypoints (mvKeys), xy_un (mvKeysUn[i].pt, 2 floats each; may
 * be NULL), descriptors (mDescriptors), bounds4 = {mnMinX, mnMaxX, mnMinY, mnMaxY} (may be NULL).
 * capacity >= so_extractor_capacity(). */
int so_dframe_collect(so_dframe* f, so_keypoint* keypoints, float* xy_un, uint8_t* descriptors, int capacity,
                      int* n_out, float* bounds4);

/* so_dframe_collect in two steps: so_dframe_wait blocks until the frame is complete on the device - from then on it can
 * be searched (so_track_search_*_submit) - and so_dframe_collect afterwards only copies the host mirrors out.  The
 * tracking thread issues the motion-model search between the two: the search kernel runs under the copies and under
 * the next frame's submission. */
int so_dframe_wait(so_dframe* f, int* n_out, float* bounds4);

/* Device pointers of a collected frame (valid until the handle's next submit). */
typedef struct {
    int32_t n, n_inside;            /* N; keypoints inside the 64 x 48 grid */
    float bounds[4];
    const float* xy_un;             /* n x 2, by keypoint index */
    const int8_t* octave;           /* n */
    const uint8_t* descriptors;     /* n x 32 */
    const int32_t* cell_start;      /* 64*48+1, cell = x * 48 + y (mGrid[x][y]) */
    const int32_t* cell_items;      /* n_inside keypoint indices, cells in order, push_back order inside a cell */
    const float* sorted_xy;         /* the same keypoints in cell_items order: the matcher's candidate layout */
    const int8_t* sorted_octave;
    const uint8_t* sorted_descriptors;
    const int32_t* col_start;       /* 65: first position of every grid column */
} so_dframe_view;
int so_dframe_device_view(const so_dframe* f, so_dframe_view* out);
/* host copies of the grid lists, for parity tests */
int so_dframe_get_grid(so_dframe* f, int32_t* cell_start, int32_t* cell_items, int32_t* n_inside);

/* so_search_by_projection_mappoints / _lastframe with the candidates read in place from a device-resident frame
 * (no host sort, no candidate upload); excluded: n bytes as in so_frame_view, may be NULL. */
int so_search_by_projection_mappoints_dframe(so_matcher* m, const so_dframe* F, const uint8_t* excluded, int32_t n_mp,
                                             const uint8_t* in_view, const float* proj_x, const float* proj_y,
                                             const float* view_cos, const int32_t* pred_level, const uint8_t* mp_desc,
                                             const uint8_t* mp_has_obs, float th, float nn_ratio, int32_t* kp_to_mp,
                                             int32_t* nmatches);
int so_search_by_projection_lastframe_dframe(so_matcher* m, const so_dframe* cur, const uint8_t* excluded,
                                             int32_t n_last, const uint8_t* valid, const float* u, const float* v,
                                             const int32_t* last_octave, const float* last_angle,
                                             const uint8_t* mp_desc, const uint8_t* mp_has_obs, float th,
                                             int check_orientation, int32_t* kp_to_last, int32_t* nmatches);

/* Device-resident table of the MapPoint fields the per-frame operators read (code/include/MapPoint.h: mWorldPos,
 * mNormalVector, mfMaxDistance, mfMinDistance, mDescriptor), indexed by a slot the caller assigns (append order).
 * Written at keyframe rate (new points, SetWorldPos / UpdateNormalAndDepth / ComputeDistinctiveDescriptors results),
 * read every frame by the tracking searches below.  Writes are complete when the call returns. */
/* (so_map is declared above, with so_fuse_kframe_map) */
int so_map_create(int device, so_map** out);
void so_map_destroy(so_map* map);
int so_map_size(const so_map* map);
/* slots [first, first + n) <- the given rows; first <= size (rows beyond the current size are appended and must be
 * given whole; for existing rows any array may be NULL = unchanged). */
int so_map_write(so_map* map, int32_t first, int32_t n, const float* Xw, const float* normal, const float* max_dist,
                 const float* min_dist, const uint8_t* desc);
/* Xw[slots[i]] <- Xw row i (MapPoint::SetWorldPos after bundle adjustment, code/src/Optimizer.cc:729-737) */
int so_map_write_positions(so_map* map, int32_t n, const int32_t* slots, const float* Xw);
/* rows slots[i] <- row i of the given arrays, any of which may be NULL = unchanged: the write-back of local bundle
 * adjustment for the points it moved - SetWorldPos and the UpdateNormalAndDepth that follows it (code/src/Optimizer.cc:
 * 729-737; mWorldPos, mNormalVector, mfMaxDistance, mfMinDistance) - in one launch */
int so_map_write_rows(so_map* map, int32_t n, const int32_t* slots, const float* Xw, const float* normal, const float* max_dist,
                      const float* min_dist);
int so_map_read(so_map* map, int32_t first, int32_t n, float* Xw, uint8_t* desc); /* either may be NULL */

/* TrackWithMotionModel's search: ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono = true) —
 * code/src/ORBmatcher.cc:1223-1354, the WHOLE function including the projection of the last frame's map points with
 * the current pose guess (:1242-1276).  Tcw12 = CurrentFrame.mTcw; last_slot[i] (last->n entries) = map slot of
 * LastFrame.mvpMapPoints[i], or -1 when it is null or mvbOutlier[i]; slot_has_obs[i] = Observations() > 0 of that
 * point (NULL = all 1); cur_excluded as so_frame_view.excluded (may be NULL).
 * Out: kp_to_last[k] (cur->n entries) = index in the last frame whose map point is bound to keypoint k, or -1. */
int so_track_search_last_frame(so_matcher* m, const so_dframe* cur, const uint8_t* cur_excluded, const so_dframe* last,
                               const so_map* map, const float* Tcw12, const int32_t* last_slot,
                               const uint8_t* slot_has_obs, float th, int check_orientation, int32_t* kp_to_last,
                               int32_t* nmatches);

/* The two tracking searches in two halves, for a thread that drives several agents (or that has other work while the
 * kernel runs): _submit stages the gates and enqueues the launch on the matcher's stream, _wait blocks for it and runs
 * the order-dependent resolve.  One search in flight per matcher; cur_excluded / the frames / the map must stay
 * untouched until the wait.  so_track_search_last_frame / _local_map are submit followed by wait. */
int so_track_search_last_frame_submit(so_matcher* m, const so_dframe* cur, const uint8_t* cur_excluded, const so_dframe* last,
                                      const so_map* map, const float* Tcw12, const int32_t* last_slot, float th);
int so_track_search_last_frame_wait(so_matcher* m, const uint8_t* slot_has_obs, int check_orientation, int32_t* kp_to_last,
                                    int32_t* nmatches);
int so_track_search_local_map_submit(so_matcher* m, const so_dframe* cur, const uint8_t* cur_excluded, const so_map* map,
                                     const float* Tcw12, int32_t n_local, const int32_t* local_slot, int32_t first_slot,
                                     const uint8_t* skip, float th, float nn_ratio, float viewing_cos_limit,
                                     float log_scale_factor);
int so_track_search_local_map_wait(so_matcher* m, const uint8_t* slot_has_obs, uint8_t* in_view, int32_t* kp_to_local,
                                   int32_t* nmatches);

/* TrackLocalMap's search: Tracking::SearchLocalPoints (code/src/Tracking.cc:964-1007) — Frame::isInFrustum(pMP,
 * viewing_cos_limit) (code/src/Frame.cc:316-375, MapPoint::PredictScale code/src/MapPoint.cc:476-485) for each of
 * the n_local local map points, then ORBmatcher::SearchByProjection(F, vpMapPoints, th) (code/src/ORBmatcher.cc:
 * 44-121) over the visible ones.  local_slot[i] = map slot of mvpLocalMapPoints[i] (NULL: the contiguous slots
 * first_slot .. first_slot + n_local - 1);
 * skip[i] != 0: the point is already in mCurrentFrame.mvpMapPoints or isBad() and is not searched (may be NULL).
 * Out: in_view[i] = mbTrackInView (may be NULL), kp_to_local[k] = index into the local list bound to keypoint k. */
int so_track_search_local_map(so_matcher* m, const so_dframe* cur, const uint8_t* cur_excluded, const so_map* map,
                              const float* Tcw12, int32_t n_local, const int32_t* local_slot, int32_t first_slot,
                              const uint8_t* skip, const uint8_t* slot_has_obs, float th, float nn_ratio,
                              float viewing_cos_limit, float log_scale_factor, uint8_t* in_view, int32_t* kp_to_local,
                              int32_t* nmatches);

/* A tracking stage WITHOUT a host hop between the search and the pose: what Tracking::TrackWithMotionModel
 * (code/src/Tracking.cc:714-768: SearchByProjection(mCurrentFrame, mLastFrame, th, mono) then
 * Optimizer::PoseOptimization(&mCurrentFrame)) and Tracking::TrackLocalMap (:770-807: SearchLocalPoints() then
 * PoseOptimization) do back to back, as one chain of launches on the matcher's stream:
 *   search (the same kernel as so_track_search_*) -> the order-dependent resolve ON THE DEVICE (an exact parallel form of
 *   the reference's sequential walk, code/src/ORBmatcher.cc:83-85 and :1294-1296, incl. the rotation histogram of
 *   :1319-1350) -> PoseOptimization (code/src/Optimizer.cc:239-434) over the frame's bindings, its edges read in place from
 *   the device-resident frame and map table.
 * The host waits ONCE, for the pose.  Semantics of the arguments as in so_track_search_last_frame / _local_map and
 * so_pose_optimization; additionally
 *   intr4 = fx, fy, cx, cy;  level_inv_sigma2 = mvInvLevelSigma2 (cur's number of levels);
 *   kp_slot (local-map stage, cur->n entries) = map slot bound to keypoint k on entry, < 0: none - those keypoints are the
 *   search's `excluded` set AND edges of the pose problem; bound points must carry skip[] = 1 as for the plain search.
 *   kp_slot_is_last_stage != 0: kp_slot is exactly what this matcher's last stage left for this frame - the bindings it
 *   reported, minus the outliers of its pose when it was the last-frame stage (code/src/Tracking.cc:745-760) - so the device
 *   copy of that stage is used and nothing is read from host memory inside the chain.  The library can only tell that a copy
 *   for THIS frame exists, not that the caller's bindings still equal it: pass 0 (or call so_track_stage_invalidate first)
 *   whenever mvpMapPoints changed on the host since that stage - a host re-search with a wider window, a fall-back onto
 *   TrackReferenceKeyFrame / Relocalization (code/src/Tracking.cc:321-327), a binding nulled by the isBad() test of
 *   SearchLocalPoints (:972-975).  With 0 the bindings are uploaded from kp_slot.
 * All map points are taken to have observations (slot_has_obs = NULL of the plain calls): the stage API is MONOCULAR-only -
 * the temporal points UpdateLastFrame creates for stereo / RGB-D (Observations() == 0, code/src/Tracking.cc:664-711) must go
 * through so_track_search_* with a slot_has_obs plane.
 * so_track_stage_wait - out, all required unless noted:
 *   kp_to_q[k] (cur->n) = query (index into the last frame / the local list) matched to keypoint k by THIS stage's search;
 *   in_view (local-map stage; may be NULL);  n_edges, edge_kp[e] = keypoint of edge e (ascending), edge_outlier[e]
 *   (capacity cur->n each);  Tcw_out12, n_inliers, info2 as so_pose_optimization_wait.
 * Returns SO_OK, or SO_RETRY_ON_HOST (100, not an error; nothing but kp_to_q = -1 is defined then): the stage could not be
 * finished on the device - a query ran out of K-list entries with more candidates in its window, more than 4096 keypoints or
 * more than 4096 queries in all (kResolveMaxQueries, csrc/match_device.h; the count is of queries, with or without candidates),
 * or more edges than the launched PoseOptimization variant holds - and the caller runs
 * so_track_search_* + so_pose_optimization instead (same results by construction: tests/test_track_chain_gpu.py).
 * so_track_stage_pose_again_submit: PoseOptimization over the SAME edges from another start pose (the edge list of the
 * last stage is still on the device); wait with so_track_stage_wait (kp_to_q / in_view are not written). */
#define SO_RETRY_ON_HOST 100
int so_track_stage_last_frame_submit(so_matcher* m, const so_dframe* cur, const so_dframe* last, const so_map* map,
                                     const float* Tcw12, const int32_t* last_slot, float th, int check_orientation,
                                     const float* intr4, const float* level_inv_sigma2);
int so_track_stage_local_map_submit(so_matcher* m, const so_dframe* cur, const int32_t* kp_slot, int kp_slot_is_last_stage,
                                    const so_map* map, const float* Tcw12, int32_t n_local, const int32_t* local_slot,
                                    int32_t first_slot, const uint8_t* skip, float th, float nn_ratio, float viewing_cos_limit,
                                    float log_scale_factor, const float* intr4, const float* level_inv_sigma2);
int so_track_stage_pose_again_submit(so_matcher* m, const float* Tcw12);
/* The local-map stage enqueued BEHIND the last-frame stage without the host in between (round 6): `first` is another matcher of
 * the same stream (created on the same thread) whose so_track_stage_last_frame_submit for `cur` is in flight.  What the host did
 * between the two stages of a frame (code/src/Tracking.cc:743-760, :964-1007, :779) happens on the device, in one small launch
 * between the chains: stage 1's pose becomes the float pose Frame::SetPose would hold and, converted back, the start of stage 2's
 * PoseOptimization (Converter::toCvMat / toSE3Quat, the same double operations as on the host); the keypoints bound behind stage 1
 * - its matches minus its pose's outliers - are the search's excluded set and edges of the pose problem; the local points whose map
 * slot is bound already are not searched.  skip_static[i] != 0: local point i is not searched for a reason the host knows
 * beforehand (isBad()); may be NULL.  Everything else as so_track_stage_local_map_submit.  Wait for `first` (so_track_stage_wait:
 * its results are complete long before), hand its pose to so_track_stage_set_start_pose(m, Tcw12) - only used when the stage has
 * fewer than three edges - and wait for `m`; results equal those of the two stages run one after the other with the host in
 * between, to the bit (tests/test_track_chain_gpu.py).  If the first stage turns out unusable (SO_RETRY_ON_HOST, or fewer than
 * 20 matches and the caller wants the wider window), wait for `m` anyway, discard its result and run the stages the plain way.
 * SO_RETRY_ON_HOST at submit (nothing enqueued): gates too large for the argument block. */
int so_track_stage_local_map_submit_after(so_matcher* m, so_matcher* first, const so_dframe* cur, const so_map* map, int32_t n_local,
                                          const int32_t* local_slot, int32_t first_slot, const uint8_t* skip_static, float th,
                                          float nn_ratio, float viewing_cos_limit, float log_scale_factor, const float* intr4,
                                          const float* level_inv_sigma2);
int so_track_stage_set_start_pose(so_matcher* m, const float* Tcw12);
/* Forget the device copy of the last stage's bindings: the next local-map stage uploads kp_slot whatever its flag says. */
int so_track_stage_invalidate(so_matcher* m);
int so_track_stage_wait(so_matcher* m, int32_t* kp_to_q, int32_t* nmatches, uint8_t* in_view, int32_t* n_edges,
                        int32_t* edge_kp, uint8_t* edge_outlier, float* Tcw_out12, int32_t* n_inliers, int32_t* info2);
/* The tracking stages of SEVERAL agents in one chain of launches - several agents per GPU, the concurrency of one process per
 * agent (code/Examples/Monocular/swarm_map.cc:329-337) brought inside the launches: a frame's chain is three kernels of one
 * workgroup (resolve, pose) or a few hundred (search) on a chip of 256 CUs, and chains of different agents on different
 * streams take turns on the hardware queues.  A matcher that is a member of a group (so_matcher_set_track_group) RECORDS the
 * launches of its so_track_stage_last_frame_submit / _local_map_submit / _pose_again_submit instead of issuing them - inputs
 * staged, completion word armed, map table held as always -; so_track_group_launch then issues, for all members recorded
 * since the last launch: one copy of their argument rows into HBM, ONE search launch (blockIdx.y = member), ONE resolve launch
 * and ONE PoseOptimization launch (a workgroup per member).  Every member waits with its own so_track_stage_wait and gets
 * exactly the results of a solo stage (same kernels' bodies, same summation orders: tests/test_track_group_gpu.py).
 * Rules: the members' matchers share one stream (create them on one thread, without so_runtime_private_streams); the rows of
 * one launch are stages of one kind (all last-frame, all local-map or all pose-again); a member whose submit returns
 * SO_RETRY_ON_HOST is not in the launch - its caller runs the plain calls, which launch at once on the same stream.
 * so_track_group_last_kernel_ms: HIP-event times of the last launch's search and PoseOptimization kernels (0 unless a
 * member's profiling is on); the members' own kernel times are not recorded for grouped stages. */
typedef struct so_track_group so_track_group;
int so_track_group_create(int device, so_track_group** out);
void so_track_group_destroy(so_track_group* g);
int so_matcher_set_track_group(so_matcher* m, so_track_group* g_or_null);
int so_track_group_pending(so_track_group* g);
/* An opaque number that is equal for two matchers exactly when they issue their work on the same HIP stream (the rule for
 * members of one group); 0 for a null handle. */
uint64_t so_matcher_stream_id(const so_matcher* m);
/* Stream placement of a matcher (idle handles only).  By default the handles a thread creates share that thread's stream.
 * so_matcher_private_stream: the handle gets a new stream of its own.  so_matcher_share_stream: the handle issues its work on
 * `other`'s stream from now on (`other` must outlive it) - e.g. the local-mapping matchers of the agents of one GPU on ONE stream
 * beside the tracking stream: a handful of busy streams map onto the GPU's few hardware queues without sharing one with a
 * long chain of another agent. */
int so_matcher_private_stream(so_matcher* m);
int so_map_share_stream(so_map* map, const so_matcher* with); /* the table's (synchronous) writes go out on that matcher's stream */
int so_matcher_share_stream(so_matcher* m, const so_matcher* other);
int so_track_group_launch(so_track_group* g);
int so_track_group_last_kernel_ms(so_track_group* g, float* search_ms, float* pose_ms);
/* rounds the last stage's device resolve took, and how many of its queries had candidates (diagnostics) */
int so_track_stage_last_rounds(so_matcher* m, int32_t* rounds, int32_t* active_queries);
/* HIP-event time of the stage's PoseOptimization kernel (0 unless so_matcher_set_profiling is on) */
int so_track_stage_last_pose_kernel_ms(so_matcher* m, float* ms);

/* ------------------------------------------------------------------------------------------------
 * Cross-agent keyframe exchange — one agent per GPU, RCCL all-gather over xGMI.  Replaces, for agents
 * sharded across the GPUs of a node, the server-side candidate query AgentMediator::CheckOverlapCandidates
 * (code/src/AgentMediator.cc:140-202: every new keyframe is looked up in every other agent's BoW inverted index):
 * each tick every rank contributes one fixed-capacity slot with its newest keyframe's descriptors, one ncclAllGather
 * delivers all slots to all ranks, and each rank matches its slot against every peer's with the Hamming top-2
 * kernel on the gathered buffer.  RCCL is loaded at run time (librccl.so.1).
 * ---------------------------------------------------------------------------------------------- */
typedef struct so_exchange so_exchange;
#define SO_EXCHANGE_ID_BYTES 128
/* ncclGetUniqueId: called by ONE rank; the 128 bytes reach the other ranks by whatever transport the host has
 * (SwarmMap: its WebSocket layer; bench.py: torch.distributed) */
int so_exchange_unique_id(uint8_t* id128);
/* ncclCommInitRank + slot buffers; collective: every rank of the group calls it with the same id.
 * slot_keypoints = descriptors a slot can hold (nfeatures + 3 * nlevels covers every frame) */
int so_exchange_create(int device, int rank, int world, const uint8_t* id128, int slot_keypoints, so_exchange** out);
void so_exchange_destroy(so_exchange* x);
/* One tick (collective).  The slot is filled on the device from the frame's descriptors (so_exchange_tick_dframe: no
 * host hop) or from host memory (so_exchange_tick).  Out, both `world` entries and either may be NULL: peer_counts[p]
 * = keypoints in rank p's slot; peer_candidates[p] = descriptors of this rank's slot whose best match in rank p's slot
 * has distance <= max_dist and < ratio * second best (0 for p == rank): what the host merger is told. */
int so_exchange_tick_dframe(so_exchange* x, const so_dframe* f, int max_dist, float ratio, int32_t* peer_counts,
                            int32_t* peer_candidates);
int so_exchange_tick(so_exchange* x, const uint8_t* descriptors, int n, int max_dist, float ratio, int32_t* peer_counts,
                     int32_t* peer_candidates);
/* rank `peer`'s slot as gathered by the last tick: descriptors (may be NULL), *n_out, header checksum (may be NULL) */
int so_exchange_read_slot(so_exchange* x, int peer, uint8_t* descriptors, int capacity, int* n_out, uint64_t* checksum);

/* ------------------------------------------------------------------------------------------------
 * Keyframe record — the compact binary form of what a peer needs from a keyframe for the
 * loop / merge candidate search, replacing the Boost text archive of code/src/MapUpdater.cc:190-230 /
 * code/include/KeyFrame.h:310-406 on the agent-to-agent path.  Layout (little-endian):
 *   [so_keyframe_header, 128 B][descriptors n x 32 B][geometry n x {x f32, y f32, angle f32, octave i32}]
 * The descriptor block starts on a 32-byte row so the Hamming kernels read it in place inside an all-gathered
 * slot.  Host-side functions, no GPU involved.
 * ------------------------------------------------------------------------------------------------ */
typedef struct so_keyframe_header {
    uint32_t magic;        /* "SOKF", filled by pack */
    uint16_t version;      /* 1, filled by pack */
    uint16_t header_bytes; /* sizeof(so_keyframe_header), filled by pack */
    int32_t agent_id;      /* mnClientId */
    int32_t n_keypoints;   /* N */
    uint64_t keyframe_id;  /* mnId */
    double timestamp;      /* mTimeStamp */
    uint64_t checksum;     /* of the descriptor + geometry blocks, filled by pack */
    float Tcw[12];         /* pose, 3x4 row-major */
    float K[4];            /* fx fy cx cy */
    uint32_t flags;        /* SO_KF_FLAG_*; 0 in a version-1 record */
    int32_t n_map_points;  /* version 2: keypoints with map_point_id >= 0, filled by pack2 */
    uint8_t reserved[16];
} so_keyframe_header;
#define SO_KF_FLAG_MAP_POINTS 1u /* a map-point block follows the geometry (record version 2) */

size_t so_keyframe_record_size(int32_t n_keypoints); /* 128 + 48 n */
int so_keyframe_record_pack(const so_keyframe_header* hdr, const float* xy, const float* angle, const int32_t* octave,
                            const uint8_t* descriptors, uint8_t* out, size_t capacity);
/* Validates magic / version / checksum, fills *hdr, then copies out up to `capacity` keypoints (SO_ERR_CAPACITY
 * with *hdr filled when the record holds more); output arrays may be NULL. */
int so_keyframe_record_unpack(const uint8_t* rec, size_t length, so_keyframe_header* hdr, float* xy, float* angle,
                              int32_t* octave, uint8_t* descriptors, int32_t capacity);
/* Record version 2 = version 1 + one i32 per keypoint behind the geometry: the id of the map point bound to the
 * keypoint (KeyFrame::mvpMapPoints[i]->mnId, code/include/KeyFrame.h:381; -1 = none or isBad()), padded to a multiple
 * of 32 bytes.  It is what the candidate search needs beyond version 1: SearchByBoW(KF, KF) only matches keypoints that
 * carry a good map point (code/src/ORBmatcher.cc:517-521,535-541) and hands back the map points of its pairs. */
size_t so_keyframe_record_size2(int32_t n_keypoints); /* 128 + 52 n, rounded up to 32 */
int so_keyframe_record_pack2(const so_keyframe_header* hdr, const float* xy, const float* angle, const int32_t* octave,
                             const uint8_t* descriptors, const int32_t* map_point_id, uint8_t* out, size_t capacity);
/* accepts both versions (a version-1 record yields map_point_id[i] = 0: every keypoint counts as bound) */
int so_keyframe_record_unpack2(const uint8_t* rec, size_t length, so_keyframe_header* hdr, float* xy, float* angle,
                               int32_t* octave, uint8_t* descriptors, int32_t* map_point_id, int32_t capacity);

/* ------------------------------------------------------------------------------------------------
 * Keyframe store + cross-agent candidate search — the device-side counterpart of the per-agent
 * KeyFrameDatabase the reference's server queries (AgentMediator::CheckOverlapCandidates,
 * code/src/AgentMediator.cc:140-202: EVERY new keyframe is looked up in EVERY other agent's whole database) and of the
 * per-candidate matching that follows (AgentMediator::GetSim3, :204-262: ORBmatcher(0.75, true).SearchByBoW(pCurrentKF,
 * pKF, ...) and the `nmatches < 20` gate of :259).
 *
 * The store is a ring of keyframe records in HBM (capacity_keyframes x records of up to slot_keypoints keypoints); next
 * to every record it keeps the descriptors of the keypoints that carry a map point, compacted in keypoint order - the
 * only ones SearchByBoW(KF, KF) looks at.  A search has two phases:
 *   1. detection (replaces KeyFrameDatabase::DetectLoopCandidates' BoW scoring; ORBvoc.bin is not part of the
 *      reference checkout and the north star asks for brute-force Hamming + ratio test instead): ONE scan of the whole
 *      store; votes[k] = number of the query's bound keypoints whose best match among keyframe k's bound keypoints has
 *      best < th_low and (float)best < nn_ratio * (float)second  (the acceptance test of ORBmatcher.cc:550-551, first
 *      minimum wins ties).  Keyframes of the query's own agent are skipped (AgentMediator.cc:186).
 *   2. the keyframes with votes >= min_votes, by (votes descending, slot ascending), max_candidates at most, are
 *      matched with the exact SearchByBoW(KF1, KF2) semantics of ORBmatcher.cc:481-597 with every feature in ONE
 *      vocabulary node (the limit levelsup -> root of KeyFrame::ComputeBoW's transform): sequential over the query's
 *      keypoints, targets already taken are skipped, `<` TH_LOW, ratio test, rotation histogram.  Candidates with
 *      n_matches >= min_matches (the 20 of AgentMediator.cc:259) are reported with their pairs.
 * Distances come from the GPU in both phases; the order-dependent resolve of phase 2 runs on the host inside the
 * library, like every other matcher routine here.
 * ---------------------------------------------------------------------------------------------- */
typedef struct so_kfstore so_kfstore;
typedef struct so_kf_search_params {
    int32_t th_low;            /* 50: ORBmatcher::TH_LOW, code/src/ORBmatcher.cc:38 */
    float nn_ratio;            /* 0.75: AgentMediator.cc:210 */
    int32_t check_orientation; /* 1 */
    int32_t min_votes;         /* detection gate of phase 1 */
    int32_t min_matches;       /* 20: AgentMediator.cc:259 */
    int32_t max_candidates;    /* keyframes phase 2 looks at per query (<= SO_KF_MAX_CANDIDATES) */
} so_kf_search_params;
#define SO_KF_MAX_CANDIDATES 64
typedef struct so_kf_candidate {
    int32_t slot;         /* position in the store (so_kfstore_read) */
    int32_t agent_id;     /* header fields of the stored keyframe */
    uint64_t keyframe_id;
    int32_t n_keypoints;
    int32_t votes;        /* phase 1 */
    int32_t n_matches;    /* phase 2, after the rotation histogram */
    int32_t reserved;
} so_kf_candidate;
int so_kfstore_create(int device, int capacity_keyframes, int slot_keypoints, so_kfstore** out);
void so_kfstore_destroy(so_kfstore* s);
/* n version-1/2 records in host memory, `stride` bytes apart; a full store overwrites its oldest keyframes.
 * slots_out[i] (may be NULL) = where record i went. */
int so_kfstore_append(so_kfstore* s, const uint8_t* records, size_t stride, int32_t n_records, int32_t* slots_out);
/* keyframes held, and the bound-keypoint descriptors one scan reads */
int so_kfstore_size(const so_kfstore* s, int32_t* n_keyframes, int64_t* n_descriptors);
/* phase 1 alone: votes[slot] for slot < capacity (-1: empty slot or the query's own agent) */
int so_kfstore_votes(so_kfstore* s, const uint8_t* query_record, size_t length, int32_t th_low, float nn_ratio,
                     int32_t* votes);
/* both phases.  out: max_candidates entries; pairs (may be NULL): max_candidates x n_keypoints(query), pairs[c][i1] =
 * keypoint of candidate c matched to the query's keypoint i1, or -1 (vpMatches12 of ORBmatcher.cc:481 as indices);
 * *n_out = candidates that passed min_matches.  n_evaluated (may be NULL) = keyframes phase 2 looked at. */
int so_kfstore_search(so_kfstore* s, const uint8_t* query_record, size_t length, const so_kf_search_params* p,
                      so_kf_candidate* out, int32_t* pairs, int32_t* n_out, int32_t* n_evaluated);
/* phase 2 alone on the store slots the caller names (<= SO_KF_MAX_CANDIDATES): for a host that filters the detection
 * result itself before matching - the reference's covisibility-consistency groups (AgentMediator::DetectLoop,
 * code/src/AgentMediator.cc:384-456) sit between DetectLoopCandidates and GetSim3.  so_kfstore_votes gives the scores,
 * the host picks, this matches; out[c].votes is 0.  Same outputs as so_kfstore_search otherwise. */
int so_kfstore_match(so_kfstore* s, const uint8_t* query_record, size_t length, const int32_t* slots, int32_t n_slots,
                     const so_kf_search_params* p, so_kf_candidate* out, int32_t* pairs, int32_t* n_out);
/* the record stored at `slot`, as appended (the merger needs the geometry and pose of a candidate) */
int so_kfstore_read(so_kfstore* s, int32_t slot, uint8_t* record, size_t capacity, size_t* length);
/* stats of the last search: [0] scan kernel ms (HIP events), [1] descriptor pairs the scan compared, [2] keyframes
 * scanned, [3] phase-2 kernel ms, [4] keyframes phase 2 evaluated, [5] phase-2 queries re-run on the GPU */
int so_kfstore_last_stats(so_kfstore* s, double* stats6);

/* The exchange as the reference's candidate search (code/src/AgentMediator.cc:177-191,204-262): a communicator whose
 * slot holds up to records_per_tick keyframe records (version 2) and which owns a keyframe store of store_keyframes
 * records.  so_exchange_tick_records (collective): this rank's new keyframes go out, one ncclAllGather delivers every
 * rank's, the peers' records are appended to the store on the device (no host hop), and each of this rank's new
 * keyframes is searched against the WHOLE store (everything every peer has sent so far, this tick included) with
 * so_kfstore_search's two phases.  out / pairs / n_out: n_records blocks of max_candidates entries / max_candidates x
 * slot_keypoints entries / one count, in record order (pairs may be NULL).  Inside a record's pairs block the rows are
 * packed by the QUERY's keypoint count, as so_kfstore_search packs them: candidate c's row starts at c x n_keypoints(query
 * record) - not at c x slot_keypoints - and the rest of the block is unused.  n_records may be 0 (the rank only receives). */
int so_exchange_create_store(int device, int rank, int world, const uint8_t* id128, int slot_keypoints,
                             int records_per_tick, int store_keyframes, so_exchange** out);
/* The same exchange over the HOST's transport instead of RCCL - agents that do not share a node (the reference's agents
 * reach their server over WebSockets, code/src/WebSocket.cc, code/src/ClientService.cc) or a build without RCCL.
 * allgather(user, send, recv, bytes_per_rank) delivers every rank's `bytes_per_rank` to every rank, in rank order
 * (recv = world x bytes_per_rank), and returns 0; it is called once per tick, from the ticking thread, with pinned host
 * buffers.  Everything else (records, store, search, the collective nature of a tick) is as above. */
typedef int (*so_exchange_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
int so_exchange_create_store_host(int device, int rank, int world, so_exchange_allgather_fn allgather, void* user,
                                  int slot_keypoints, int records_per_tick, int store_keyframes, so_exchange** out);
int so_exchange_tick_records(so_exchange* x, const uint8_t* records, size_t stride, int32_t n_records,
                             const so_kf_search_params* p, so_kf_candidate* out, int32_t* pairs, int32_t* n_out);
/* The same for ONE keyframe whose descriptors and undistorted keypoints are device-resident (the frame tracked last):
 * the record is assembled in the slot by a kernel; hdr supplies ids / pose, map_point_id[n] (host) the bindings. */
int so_exchange_tick_keyframe(so_exchange* x, const so_dframe* f, const so_keyframe_header* hdr,
                              const int32_t* map_point_id, const so_kf_search_params* p, so_kf_candidate* out,
                              int32_t* pairs, int32_t* n_out);
/* Failure behaviour of the ticks (all of them are collective).  A rank-local problem - frame not ready, malformed record,
 * missing argument - does not keep the rank out of the collective: it takes part with zero records / an empty slot and
 * returns SO_ERR_INVALID_ARG afterwards, the other ranks are not affected.  Only a null handle returns before the
 * collective.  The wait for the collective is bounded (so_exchange_set_timeout; default SWARMORB_COLLECTIVE_TIMEOUT_MS
 * or 30000 ms; 0 = unbounded; a handle's FIRST collective, which also pays RCCL's lazy connection set-up and the ranks'
 * start-up skew, gets at least 120 s): when a peer never enters the tick the survivors get SO_ERR_TIMEOUT, the handle is
 * DEAD - every later tick returns SO_ERR_TIMEOUT at once, so_exchange_destroy does not wait for the stuck stream - and
 * the group has to be re-created (in a fresh child process, never by re-executing one that holds the GPU).  The budget
 * covers RCCL collectives; with the host transport (so_exchange_create_store_host) the caller's all-gather callback owns
 * its time-outs.  so_exchange_debug_stall (test hook) delays this rank's next collective by a spinning
 * workgroup on the tick's stream. */
int so_exchange_set_timeout(so_exchange* x, int milliseconds);
int so_exchange_is_dead(const so_exchange* x);
int so_exchange_debug_stall(so_exchange* x, int milliseconds);
/* record `index` (< records_per_tick) of rank `peer` as the last tick's all-gather delivered it; *length = 0 for an
 * unused position.  record may be NULL (length only). */
int so_exchange_read_record(so_exchange* x, int peer, int index, uint8_t* record, size_t capacity, size_t* length);
/* the store behind the communicator (owned by it; NULL for one made by so_exchange_create) */
so_kfstore* so_exchange_store(so_exchange* x);

/* ------------------------------------------------------------------------------------------------
 * Bundle adjustment — replaces Optimizer::LocalBundleAdjustment / BundleAdjustment / GlobalBundleAdjustment
 * (code/include/Optimizer.h:41-46, code/src/Optimizer.cc:42-237,436-740) and the g2o machinery they drive
 * (OptimizationAlgorithmLevenberg, BlockSolver_6_3 with Schur complement, LinearSolverEigen, RobustKernelHuber,
 * EdgeSE3ProjectXYZ, VertexSE3Expmap, VertexSBAPointXYZ; code/Thirdparty/g2o/g2o/...) on a FLATTENED problem.
 * The caller's adapter gathers the local window from the KeyFrame/MapPoint graph (Optimizer.cc:437-482) and
 * writes the results back under the map mutex (Optimizer.cc:713-739); everything in between runs here, on
 * the GPU in FP64.  Map storage types cross the boundary unchanged (float poses / points / observations).
 * Monocular edges only, which is all SwarmMap builds.
 * ---------------------------------------------------------------------------------------------- */
typedef struct so_ba so_ba;

typedef struct {
    int32_t n_poses;       /* keyframe vertices, ascending vertex id (KeyFrame::mnId): defines the Hessian order */
    const float* Tcw;      /* n_poses x 12, row-major [R|t] of KeyFrame::GetPose() */
    const uint8_t* fixed;  /* n_poses: vSE3->setFixed(...) (fixed keyframes and pKFi->isFirst()) */
    const float* intr;     /* n_poses x 4: fx, fy, cx, cy */
    int32_t n_points;      /* map point vertices, ascending vertex id */
    const float* Xw;       /* n_points x 3, MapPoint::GetWorldPos() */
    int32_t n_edges;       /* EdgeSE3ProjectXYZ in insertion order */
    const int32_t* edge_pose;
    const int32_t* edge_point;
    const float* obs;         /* n_edges x 2: kpUn.pt */
    const float* inv_sigma2;  /* n_edges: pKFi->mvInvLevelSigma2[kpUn.octave] */
} so_ba_problem;

typedef struct {
    int32_t its_stage1;   /* optimizer.optimize(5)  (LocalBA) / optimize(nIterations) (BundleAdjustment) */
    int32_t its_stage2;   /* optimizer.optimize(10) after the outlier pass; 0 = single stage */
    int32_t robust;       /* Huber kernel in stage 1 (bRobust) */
    float huber_delta;    /* sqrt(5.991) as float */
    float chi2_threshold; /* 5.991 */
} so_ba_options;

typedef struct {
    double chi2_initial, chi2_final;
    double lambda_final;
    int32_t iterations_stage1, iterations_stage2;
    int32_t lm_trials;
    int32_t aborted;    /* *stop was observed set */
    int32_t n_outliers;
    float gpu_ms;       /* HIP-event time from the first to the last kernel of the call */
    float wall_ms;      /* host wall time of the call */
    float solve_ms;     /* HIP-event time summed over the reduced-system solve kernel launches */
    int32_t n_solves;
    /* blocked solver (more than 43 free keyframes), 0 otherwise: FP64 work of ONE solve of the reduced camera system
     * over the nonzero 96 x 96 tiles of its block skyline, the same for a dense matrix of that size, and the number
     * of tiles inside the skyline (lower triangle) */
    double solve_gflop_structural, solve_gflop_dense, nnz_tiles;
    /* which solver factored the reduced camera system: 0 single-workgroup kernels (up to 43 free keyframes), 1 blocked
     * Cholesky as a chain of launches, 2 blocked Cholesky as ONE launch of tile workgroups (up to 231 skyline tiles, when
     * the device can keep them all resident) */
    int32_t solver_path;
    int32_t n_free_keyframes; /* keyframes that got a hessian index: not fixed and observed by at least one edge */
    /* single-launch solves of this solver context whose workgroups gave up waiting for each other (the launch was not
     * fully resident: another process on the GPU, a CU mask); the call was then repeated on the multi-launch path, which
     * the context keeps from then on (solver_path 1) */
    int32_t flow_timeouts;
    int32_t pcg_iterations;  /* solver_path 3 (so_ba_set_linear_solver: PCG): conjugate-gradient iterations of all solves of the call
                              * (n_solves / lm_trials solves; nnz_tiles then holds the nonzero 6 x 6 blocks of S) */
} so_ba_info;

int so_ba_create(int device, so_ba** out);
void so_ba_destroy(so_ba* ba);
/* LocalBundleAdjustment's schedule: {5, 10, robust, sqrt(5.991)f, 5.991f} (Optimizer.cc:547,635-660) */
void so_ba_options_local(so_ba_options* opt);
/* BundleAdjustment's schedule: one stage of n_iterations, optional Huber (Optimizer.cc:49-52,190-192) */
void so_ba_options_global(so_ba_options* opt, int32_t n_iterations, int32_t robust);

/* stop: the reference's bool* pbStopFlag (may be NULL); polled between LM iterations and trials.
 * Outputs: Tcw_out n_poses x 12 (Converter::toCvMat of every pose vertex), Xw_out n_points x 3,
 * edge_outlier n_edges (e->chi2() > threshold || !e->isDepthPositive(), Optimizer.cc:682-695),
 * edge_chi2 n_edges (may be NULL). */
/* Optimizer::PoseOptimization(Frame* pFrame, bGlobal) — code/src/Optimizer.cc:239-434 (monocular): motion-only BA
 * of one frame, called 2-3 times per frame by Tracking.  The whole schedule (4 rounds x 10 LM iterations, Huber,
 * outlier re-classification) runs in ONE kernel launch.  Inputs are the n keypoints that have a map point:
 * Tcw12 = pFrame->mTcw (row-major [R|t]), intr = fx, fy, cx, cy, Xw n x 3 (GetWorldPos / GetGlobalPos),
 * obs n x 2 (mvKeysUn[i].pt), inv_sigma2 n (mvInvLevelSigma2[octave]).  Outputs: Tcw_out12 (SetPose argument),
 * outlier[i] (mvbOutlier), *n_inliers = nInitialCorrespondences - nBad (0 with untouched outputs when n < 3). */
int so_pose_optimization(so_ba* ba, const float* Tcw12, const float* intr, int32_t n, const float* Xw,
                         const float* obs, const float* inv_sigma2, float* Tcw_out12, uint8_t* outlier,
                         int32_t* n_inliers, int32_t* info /* [iterations, lm_trials], may be NULL */);
/* The same call in two halves: _submit stages the inputs and launches, _wait returns the results (one call may be in
 * flight per handle; SO_ERR_INVALID_ARG otherwise).  What a tracking thread does in between runs under the kernel -
 * e.g. handing the next frame to the extractor. */
int so_pose_optimization_submit(so_ba* ba, const float* Tcw12, const float* intr, int32_t n, const float* Xw,
                                const float* obs, const float* inv_sigma2);
int so_pose_optimization_wait(so_ba* ba, float* Tcw_out12, uint8_t* outlier, int32_t* n_inliers, int32_t* info);

/* Several independent PoseOptimization problems in ONE launch (a workgroup per problem), e.g. the frames of several
 * agents a thread drives in lockstep.  Fields as the arguments of so_pose_optimization; results identical to calling it
 * per problem. */
typedef struct {
    const float* Tcw12;
    const float* intr;
    int32_t n;
    const float* Xw;
    const float* obs;
    const float* inv_sigma2;
    float* Tcw_out12;
    uint8_t* outlier;
    int32_t* n_inliers;
    int32_t* info; /* [iterations, lm_trials], may be NULL */
} so_pose_problem;
int so_pose_optimization_batch(so_ba* ba, int32_t n_problems, const so_pose_problem* problems);
/* HIP events around the PoseOptimization kernel on / off (on by default; a caller that samples the kernel time on some
 * frames only switches them off in between: two event records per call). */
int so_pose_optimization_set_timing(so_ba* ba, int enabled);
/* HIP-event time (ms) of the kernel of the last so_pose_optimization call on this handle. */
int so_pose_optimization_last_kernel_ms(so_ba* ba, float* ms);

int so_bundle_adjust(so_ba* ba, const so_ba_problem* problem, const so_ba_options* options,
                     const volatile uint8_t* stop, float* Tcw_out, float* Xw_out, uint8_t* edge_outlier,
                     double* edge_chi2, so_ba_info* info);

/* Local bundle adjustments of SEVERAL agents as one chain of launches (several agents per GPU; the reference runs one
 * LocalMapping thread per agent process, code/Examples/Monocular/swarm_map.cc:329-337, code/src/LocalMapping.cc:53-110).  A
 * window's Levenberg-Marquardt chain is ~65 small dependent launches; eight agents' chains on eight streams are bound by the
 * rate at which the GPU's command processor dispatches kernels, not by its CUs.  Members of a group (one so_ba per agent, each
 * called from its own thread as before) stage and upload their windows on their own streams, RECORD the launches of the chain
 * and hand the list in; the first member of a round waits `window_us` (<= 0: 250) for the others, merges the lists phase by
 * phase - launches of one kind become ONE launch with the member as blockIdx.y, their argument blocks rows of a table in HBM
 * - and issues the round on the group's stream.  Every member waits for its own completion word and gets the results of a
 * solo call bit by bit (same kernel bodies, same grids per member, same summation orders: tests/test_ba_group_gpu.py).
 * A member that arrives late makes the next round; nobody waits for a member that does not come.  Grouped: local windows
 * (<= 43 free keyframes, edge-table gather); larger problems, solve timing and SWARMORB_BA_NO_CHAIN run ungrouped.
 * so_ba_group_stats: rounds, members summed over rounds, grouped launches, ungrouped launches, rows summed over grouped
 * launches, members registered, window_us, 0. */
typedef struct so_ba_group so_ba_group;
int so_ba_group_create(int device, double window_us, so_ba_group** out);
void so_ba_group_destroy(so_ba_group* g);
int so_ba_set_group(so_ba* b, so_ba_group* g_or_null);
int so_ba_group_stats(so_ba_group* g, double* out8);
/* Experiment kept for the record (NOTES.md G.10), off by default: SWARMORB_BA_RESIDENT=1 runs the Levenberg-Marquardt trials of a
 * local window's stage (<= 26 free keyframes, ungrouped) as ONE resident launch - workgroups that walk the phases' virtual blocks
 * and meet at grid barriers, the MFMA solve on four waves of workgroup 0 - instead of five launches per trial.  Same bits
 * (tests/test_ba_group_gpu.py), but slower on this hardware, alone and next to other agents.  so_ba_resident_stages: stages of
 * this context that ran that way. */
int so_ba_resident_stages(const so_ba* b, long long* n_out);
/* How the reduced camera system S x = b of every LM trial is solved on LARGE maps (80 free keyframes and more; smaller
 * systems always take the single-workgroup / blocked direct solvers).
 *   SO_BA_SOLVER_DIRECT (default): block-skyline Cholesky on FP64 MFMA tiles - what the reference's LinearSolverEigen /
 *     SimplicialLDLT computes (code/Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124), to rounding.
 *   SO_BA_SOLVER_PCG: block-Jacobi preconditioned conjugate gradients over the nonzero 6 x 6 blocks of S (g2o's
 *     LinearSolverPCG, g2o/solvers/linear_solver_pcg.h; BASELINE.json's north_star names it) down to a relative residual
 *     |r| / |b| <= rel_tolerance (<= 0: keep, default 1e-7) within max_iterations (<= 0: keep, default 4000; the iterate
 *     is used as it is when the cap is hit).  An iterative solve: results agree with the direct solver's within the
 *     tolerance the tests state (tests/test_ba_gpu.py), not to rounding.  Deterministic run to run. */
#define SO_BA_SOLVER_DIRECT 0
#define SO_BA_SOLVER_PCG 1
int so_ba_set_linear_solver(so_ba* ba, int solver, double rel_tolerance, int max_iterations);
/* HIP events around the reduced-camera-system solve of every LM trial (so_ba_info.solve_ms / n_solves) on / off.  Off by
 * default: an event record in front of and behind a kernel idles the stream ~6 us each, 4 % of a 64-keyframe window
 * (profiles/r3_lba_trial_sequence.txt).  With timing off solve_ms and n_solves are 0. */
int so_bundle_adjust_set_solve_timing(so_ba* ba, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* SWARMORB_H */
