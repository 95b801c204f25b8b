// match_device.h — device-side layout of the Hamming matcher.
//
// HBM layout (one so_matcher = one agent's matcher context, buffers grow on demand and stay resident):
//   candidate frame, SoA in grid-traversal order (cell x, cell y, keypoint index):
//     xy      float2  undistorted keypoint position (mvKeysUn[i].pt)
//     octave  i8
//     desc    2 x uint4 (32 B) per keypoint
//     limit   i32 optional dynamic gate: candidate is eligible iff dist < limit (0 = taken, INT_MAX = free)
//     cols    i32 x 65 first position of each grid column (window queries scan only their GetFeaturesInArea columns)
//   queries: MatchQuery (56 B) or MatchQueryW (16 B) + 32-B descriptor each; results: K u32 keys (dist << 16 | position) per query.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace so {

constexpr int kMatchGridCols = 64, kMatchGridRows = 48;  // FRAME_GRID_COLS / ROWS, code/include/Frame.h:38-39

struct MatchFrameDev {
    const float2* xy;
    const int8_t* octave;
    const uint4* desc;
    const int32_t* limit;  // may be null
    int n;
    // per-level tables of the candidate frame and the epipole, used by the optional gates
    float inv_sigma2[8];  // mvInvLevelSigma2 (Fuse chi2 gate)
    float sigma2[8];      // mvLevelSigma2    (CheckDistEpipolarLine)
    float scale[8];       // mvScaleFactors   (epipole distance test)
    float ex, ey;
    // the frame's 64 x 48 grid (Frame.cc:259-260): first candidate position of every cell column (65 entries) when
    // the candidates are in grid-traversal order, else null (vocabulary-node order: queries carry explicit ranges)
    const int32_t* col_start;
    float min_x, min_y, grid_inv_w, grid_inv_h;
    float grid_min_y;  // origin the cell ROWS were assigned with (differs from min_y only for a KeyFrame's int bounds)
};

enum : uint32_t {
    kQRange = 1u,       // candidates = positions [c_begin, c_end) (vocabulary-node segment); no window / level test
    kQChi2Gate = 2u,    // Fuse: e2 * inv_sigma2[octave] > 5.99 -> skip (ORBmatcher.cc:845-861)
    kQEpipolar = 4u,    // SearchForTriangulation: epipole distance + epipolar line tests (ORBmatcher.cc:131-148,693-706)
    kQPreferLast = 8u,  // ties go to the LAST candidate in scan order ("dist > bestDist -> continue", :688)
};

struct MatchQuery {
    float u, v, r;
    int32_t min_level, max_level;
    int32_t active;
    int32_t c_begin, c_end;
    uint32_t flags;
    int32_t max_dist;   // candidates with dist > max_dist never compete (256 = no limit)
    float la, lb, lc;   // epipolar line of the query keypoint in the candidate image
    int32_t pad;
};

// The two tracking searches (SearchByProjection against the local map / the last frame) only need the window and
// the level range: 16 B per query instead of 56 on the way up (thousands of queries per call).
struct MatchQueryW {
    float u, v, r;
    int8_t min_level, max_level;
    uint8_t active, pad;
};

// Tracking searches on device-resident data (so_track_search_*): the query of wave i is not read from a staged
// record but built in the kernel from the map-point table and the frame pose — the projection of
// SearchByProjection(cur, last) (code/src/ORBmatcher.cc:1242-1276) or Frame::isInFrustum + the window of
// SearchByProjection(F, vpMapPoints) (code/src/Frame.cc:316-375, ORBmatcher.cc:52-70).  Same float / double
// operation sequence as frame_frustum_kernel and the CPU oracle.
constexpr int kTrackMaxCandBits = 4096, kTrackMaxQueryBits = 16384;
struct TrackQuerySrc {
    // map-point table (so_map), indexed by slot
    const float* Xw;
    const float* normal;
    const float* max_dist;
    const float* min_dist;
    const uint4* desc;
    // per query
    const int32_t* slot;         // map slot, < 0: no map point (null = slot_base + query index); may point into pinned
                                 // host memory (one read per wave)
    int slot_base;
    const uint8_t* skip;         // local-map search: 1 = not searched (already matched in this frame / bad); may be null
    const int8_t* last_octave;   // last-frame search: lastFrame.mvKeys[i].octave
    uint8_t* in_view_out;        // local-map search: mbTrackInView per query (may be null)
    uint8_t* count8_out;         // candidates per query clamped to 255 (0: inactive / none): the one plane the host's
                                 // resolve scans for every query (may be null)
    int32_t* slot_out;           // the query's map slot as the search read it (device memory; may be null): the device-side
                                 // resolve takes it from here instead of reading the pinned host list again
    float Tcw[12];
    float fx, fy, cx, cy;
    float bounds[4];             // mnMinX, mnMaxX, mnMinY, mnMaxY
    float scale[8];              // mvScaleFactors of the current frame
    int nlevels;
    float th;
    float cos_limit, log_scale_factor;  // local-map search
    int second_best_bound;       // local-map search: ceil(TH_HIGH / nn_ratio), candidates beyond it cannot matter
    int n_slots;                 // size of the table (slots beyond it are treated as "no map point")
    // Small gates travel in the kernel arguments instead of a staged buffer (no copy launch in front of the search):
    // bit c of excl_bits = candidate POSITION c is not eligible (bound on entry / taken); bit i of skip_bits = query i
    // is not searched.  use_bits selects them over `skip` and MatchFrameDev::limit.
    int use_bits;
    int keys_soa;                // 1: K-lists are written [k][query] (the host reads ranks 0 and 1 as two streams)
    uint32_t excl_bits[kTrackMaxCandBits / 32];
    uint32_t skip_bits[kTrackMaxQueryBits / 32];
};

// The order-dependent half of the two tracking searches on the device (round 5).  The reference walks the map points in order and
// a point takes the best candidate keypoint no EARLIER point has taken (ORBmatcher.cc:83-85 in SearchByProjection(Frame,
// vpMapPoints), :1294-1296 in SearchByProjection(CurrentFrame, LastFrame)); the host loops of so_track_search_*_wait do
// exactly that over the K-lists.  track_resolve_kernel reproduces the sequential result in parallel rounds:
//   every unresolved query claims all the still-free entries of its K-list (atomicMin of its rank); a query whose first
//   one / two free entries - the ones its decision reads - are claimed by no query of lower rank cannot be affected by
//   anything still undecided (taken entries only accumulate, and a lower query can only ever take an entry it claims), so
//   its decision is final: distance / ratio / level tests as in the reference, and its keypoint is marked taken.  The lowest
//   unresolved query is always final, so the rounds end; conflicts are local (queries whose windows overlap), a handful of
//   rounds in practice.  A query that runs out of list entries with more candidates in its window raises `fallback`
//   (the host then resolves the call the old way).
// Behind the matches: TrackWithMotionModel's rotation-consistency check (ORBmatcher.cc:1319-1350), the frame's bindings
// (entry bindings + new matches) and the edge list of the PoseOptimization call that follows (keypoints with a map point,
// ascending index - pose_opt_chain_kernel reads it in place).
constexpr int kResolveMaxCand = 4096;     // keypoints of the current frame the kernel holds in LDS
constexpr int kResolveMaxQueries = 4096;  // queries (map points of the last frame / of the local map) of one search
struct TrackResolveArgs {
    const uint32_t* keys;        // [K][nq]
    const uint8_t* cnt8;         // candidates per query, clamped to 255
    int nq, K;
    int mode;                    // 2: last-frame search (one candidate, <= TH_HIGH), 3: local-map search (two + ratio / level test)
    float nn_ratio;
    int n_cand;                  // candidate positions (the current frame's keypoints inside the grid)
    int n_kp;                    // keypoints of the current frame
    const int8_t* s_octave;      // by candidate position
    const int32_t* cell_items;   // candidate position -> keypoint index
    int check_orientation;
    const float* q_angle;        // last frame's keypoint angles (query i = keypoint i of the last frame)
    const float* cur_angle;      // current frame's, by keypoint index
    const int32_t* q_slot;       // query -> map slot (null: slot_base + query)
    int slot_base;
    const int32_t* kp_slot_in;   // bindings on entry by keypoint index (null: none)
    int32_t* kp_slot_out;        // bindings behind this stage by keypoint index (device; the next stage's kp_slot_in)
    int32_t* kp_to_q;            // [n_kp] out (host-mapped): query matched to keypoint k, -1 none
    int32_t* e_kp;               // edge list out (device)
    int32_t* e_slot;
    int32_t* e_kp_host;          // host-mapped copy of e_kp
    int32_t* head;               // device: {n_edges, nmatches, fallback, rounds}
    int32_t* head_host;          // host-mapped copy
};
void launch_track_resolve(const TrackResolveArgs& a, hipStream_t s);
// one row per agent of a so_track_group, table in HBM; the instance is chosen for the largest search of the group (the
// resolve is exact: its result does not depend on the instance)
void launch_track_resolve_group(const TrackResolveArgs* d_tab, int n, int max_nq, hipStream_t s);

// The tracking search of one agent inside a grouped launch (topk_window_kernel<5 / 6>, blockIdx.y = row)
struct TrackGroupJob {
    MatchFrameDev F;
    TrackQuerySrc T;
    uint32_t* keys;
    int32_t* count;
    int nq, K;
};
void launch_topk_track_group(const TrackGroupJob* d_jobs, int n_jobs, int mode, int max_nq, hipStream_t s);

// The first tracking stage hands over to the second ON THE DEVICE (so_track_stage_local_map_submit_after): between the two
// chains of a frame the host used to wait for stage 1's pose, drop its outliers' bindings, mark the keypoints that are bound and the
// local points that are matched already, and pass the pose on as the start of stage 2 (code/src/Tracking.cc:743-760 ->
// :964-1007 -> :779).  track_link_kernel (one workgroup) does that hand-over on the stage-2 rows that are already in HBM:
//   T.Tcw of the search row   = stage 1's pose as the float [R|t] Frame::SetPose would hold (pose_convert.h),
//   init of the pose row      = that float pose back as an SE3Quat (Converter::toSE3Quat), the start of PoseOptimization,
//   T.excl_bits               = candidate positions whose keypoint is bound behind stage 1 (minus its pose's outliers),
//   T.skip_bits              |= local points whose map slot is bound in this frame already (mbTrackInView = false, :966-978).
struct TrackLinkArgs {
    const double* pose1;         // stage 1's result: q[4] (x y z w), t[3] - host-mapped memory its pose kernel wrote
    const int32_t* kp_slot;      // bindings by keypoint index behind stage 1 (device)
    const int32_t* cell_items;   // candidate position -> keypoint index
    int n_cand, n_kp;
    const int32_t* local_slot;   // n_local map slots (null: first_slot + i); may be pinned host memory
    int first_slot, n_local;
    TrackGroupJob* job;          // stage 2's search row (device)
    double* pose2_init;          // &PoseOptArgs::init of stage 2's pose row (device): q[4], t[3]
};
void launch_track_link(const TrackLinkArgs& a, hipStream_t s);

// Projection + gating half of the keyframe-side map-point searches: Fuse (code/src/
// ORBmatcher.cc:767-815), Fuse / SearchByProjection with a Sim3 (:916-964, :286-333), one direction of SearchBySim3
// (:1054-1094, :1130-1170) and SearchByProjection(Frame, KeyFrame, ...) (:1374-1410).  project_queries_kernel turns
// every map point into the MatchQuery topk_window_kernel<0> consumes, in HBM, thread per point; the compact copy
// (MatchQueryW) goes to host-mapped memory for the host's resolve loops and the parity tests.
enum : uint32_t {
    kPChain = 1u,       // SearchBySim3: Pc = B * (A * P + a) + b, dist3D = |Pc|, no viewing-angle gate
    kPFrameForm = 2u,   // Frame overload: no depth test, u = fx * xc * invzc + cx, closed bounds, no viewing-angle gate
    kPAngleGate = 4u,   // PO.dot(Pn) < 0.5 * dist3D -> skip
};
struct ProjectSrc {
    const float* Xw;        // n x 3
    const float* normal;    // n x 3 (kPAngleGate)
    const float* max_dist;  // mfMaxDistance
    const float* min_dist;  // mfMinDistance
    const uint8_t* valid;   // the caller's object-graph gates
    const int32_t* slot;    // null: point i is row i of the arrays above; else row slot[i] (the arrays are a so_map's
    int n_rows;             // tables with n_rows rows; a slot outside [0, n_rows) makes the point inactive)
    int n;
    float A[12], B[12];     // [R | t] rows; B only with kPChain
    float Ow[3];            // camera centre the distances are measured from (not with kPChain)
    float fx, fy, cx, cy;
    float bounds[4];        // mnMinX, mnMaxX, mnMinY, mnMaxY of the target
    float scale[8];         // mvScaleFactors of the target
    int nlevels;
    float log_scale_factor, th;
    uint32_t flags;
    int level_above;        // octave window [pred - 1, pred + level_above]
    uint32_t qflags;        // MatchQuery::flags of the produced queries (kQChi2Gate for Fuse)
    int q_max_dist;         // MatchQuery::max_dist
};
void launch_project_queries(const ProjectSrc& S, MatchQuery* d_q, MatchQueryW* d_qw_mapped, hipStream_t s);

// Batched matcher calls (so_matcher_batch_begin / _end): several independent searches - each with its own candidate
// frame, queries and outputs - as ONE projection launch and ONE search launch, blockIdx.y = job.  The table sits in HBM.
struct BatchJobDev {
    MatchFrameDev F;
    ProjectSrc S;            // used when project != 0: the job's queries are produced on the device
    MatchQuery* q;           // nq records (read by the search; written by the projection when project != 0)
    MatchQueryW* qw;         // compact copies for the host (project != 0), host-mapped
    const uint4* qdesc;      // nq x 32 B (row qslot[i] of it when qslot is not null: a so_map's descriptor table)
    const int32_t* qslot;
    uint32_t* keys;          // nq x K, host-mapped
    int32_t* count;          // nq, host-mapped
    int nq, K, project, pad;  // project: 0 queries staged, 1 projected map points (S), 2 SearchForTriangulation queries (t_*)
    // project == 2: the queries of SearchForTriangulation(kf1, kf2) are generated on the device from the RESIDENT keyframe 1:
    // query p = position p of its vocabulary-node order (its descriptor: row p of qdesc = keyframe 1's resident table)
    const float2* t_xy1;      // keyframe 1's mvKeysUn in that order (resident)
    const uint8_t* t_free1;   // by position: no map point yet (staged)
    const uint16_t* t_node;   // by position: which node of keyframe 1's node list (staged)
    const int2* t_range;      // per node of keyframe 1: its candidates [begin, end) in keyframe 2's node order, begin = end if
                              // keyframe 2 has no feature in that node (staged)
    float t_F[9];             // F12
    int t_pad;
};
// Grid layout of a batch: the jobs' workgroups back to back (a job with 4000 queries next to one with 700 must not make
// every job launch 1000 workgroups): first_proj / first_topk[j] = first workgroup of job j, [n_jobs] = the grid size.
constexpr int kBatchMaxJobsDev = 64;
struct BatchGridDev {
    int n_jobs, pad[3];
    int first_proj[kBatchMaxJobsDev + 1];
    int first_topk[kBatchMaxJobsDev + 1];
};
void launch_batch(const BatchGridDev* d_grid, const BatchJobDev* d_jobs, int n_jobs, int proj_blocks, int topk_blocks, hipStream_t s);

void launch_stage_in(void* dst, const void* src_mapped, size_t bytes, hipStream_t s);
// mode 2 = last-frame search, 3 = local-map search; queries [q_first, q_first + nq) write keys / counts at [0, nq)
void launch_topk_track(const MatchFrameDev& F, const TrackQuerySrc& T, int mode, int q_first, int nq, int K,
                       uint32_t* d_keys, int32_t* d_count, hipStream_t s);
// compact: d_q points to MatchQueryW records instead of MatchQuery
void launch_topk_window(const MatchFrameDev& F, const void* d_q, bool compact, const uint4* d_qdesc, int nq, int K,
                        uint32_t* d_keys, int32_t* d_count, hipStream_t s);
// ---- the local-mapping thread's two per-point loops between the matcher and local BA ----
// CreateNewMapPoints' per-match body (code/src/LocalMapping.cc:263-420, monocular): thread per match
struct TriKeyframeDev {
    float Tcw[12], Ow[3];
    float fx, fy, cx, cy, invfx, invfy;
    float scale[8], sigma2[8];
};
struct TriArgs {
    TriKeyframeDev kf1;          // mpCurrentKeyFrame
    const TriKeyframeDev* kf2;   // the neighbour keyframes of this call
    const int32_t* kf2_of;       // per match: which neighbour
    const float2* xy1;           // mvKeysUn[idx1].pt
    const float2* xy2;
    const int32_t* oct1;
    const int32_t* oct2;
    uint8_t* ok;                 // host-mapped
    float* x3D;                  // host-mapped, 3 per match
    // not null: the accepted matches' new map points also get MapPoint::UpdateNormalAndDepth's three fields - two
    // observations (the keyframe, then the neighbour), the keyframe as reference, its octave `oct1` - in the same launch
    float* normal;               // host-mapped, 3 per match
    float* max_dist;             // host-mapped
    float* min_dist;             // host-mapped
    float last_scale;            // mvScaleFactors[nLevels - 1] of the keyframe
    float ratio_factor;
    int n;
};
void launch_triangulate(const TriArgs& A, hipStream_t s);
// MapPoint::UpdateNormalAndDepth (code/src/MapPoint.cc:413-465) for a batch of map points: thread per point
struct NormalDepthArgs {
    const int32_t* off;          // n + 1
    const float* obs_Ow;         // 3 per observation
    const float* Xw;             // 3 per point
    const float* ref_Ow;         // 3 per point
    const float* ref_level_scale;
    const float* ref_last_scale;
    float* normal;               // host-mapped, 3 per point
    float* max_dist;             // host-mapped
    float* min_dist;             // host-mapped
    int n;
    // indexed form (so_update_normal_and_depth_indexed): kf_Ow != nullptr - observation k is seen from camera centre
    // kf_Ow[3 obs_kf[k]], the reference keyframe's centre is kf_Ow[3 ref_kf[p]]; obs_Ow / ref_Ow are not read
    const float* kf_Ow;
    const int32_t* obs_kf;
    const int32_t* ref_kf;
};
void launch_normal_depth(const NormalDepthArgs& A, hipStream_t s);

constexpr int kDistinctiveMaxObs = 512;
void launch_distinctive_desc(const uint4* d_desc, const int32_t* d_off, int n_points, int32_t* d_best_idx,
                             int32_t* d_best_median, hipStream_t s);
void launch_hamming_top2(const uint4* d_A, int na, const uint4* d_B, int nb, int32_t* d_best_idx, int32_t* d_best_dist,
                         int32_t* d_second_dist, hipStream_t s);
// exchange slot = [header row: i32 n, i32 rank, u64 checksum, 16 B zero][slot_keypoints x 32 B descriptors, zero padded];
// checksum = sum over the n x 32 descriptor bytes of byte[i] * ((i mod 65521) + 1), mod 2^61 - 1
void launch_exchange_fill_slot(const uint4* d_desc, int n, int slot_keypoints, int rank, uint4* d_slot, hipStream_t s);

}  // namespace so
