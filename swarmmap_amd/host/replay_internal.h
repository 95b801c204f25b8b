// replay_internal.h - the state of the replay loop (replay.cc: the tracking thread's steps and the open-loop local-mapping
// job; closedloop.cc: the closed loop's map model and local-mapping job).  Internal to libswarmorb_replay.so.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "../../include/swarmorb.h"


constexpr int kEventEvery = 4;

static inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct BaWindow {
    std::vector<float> Tcw, intr, Xw, obs, w;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> epose, epoint;
};

struct M4 {  // 4x4 double, row-major
    double a[16];
    static M4 eye() {
        M4 m;
        for (int i = 0; i < 16; i++) m.a[i] = (i % 5 == 0) ? 1.0 : 0.0;
        return m;
    }
};

static inline M4 mul(const M4& A, const M4& B) {
    M4 C;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += A.a[4 * i + k] * B.a[4 * k + j];
            C.a[4 * i + j] = s;
        }
    return C;
}

static inline M4 rigid_inverse_general(const M4& T) {  // general 4x4 inverse by Gauss-Jordan with partial pivoting
    double m[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            m[i][j] = T.a[4 * i + j];
            m[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(m[r][c]) > std::fabs(m[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 8; j++) std::swap(m[p][j], m[c][j]);
        const double d = 1.0 / m[c][c];
        for (int j = 0; j < 8; j++) m[c][j] *= d;
        for (int r = 0; r < 4; r++)
            if (r != c) {
                const double f = m[r][c];
                for (int j = 0; j < 8; j++) m[r][j] -= f * m[c][j];
            }
    }
    M4 R;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) R.a[4 * i + j] = m[i][4 + j];
    return R;
}

static inline void to_f12(const M4& T, float* o) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) o[4 * r + c] = (float)T.a[4 * r + c];
}

static inline M4 from_f12(const float* p) {
    M4 T = M4::eye();
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) T.a[4 * r + c] = (double)p[4 * r + c];
    return T;
}


// ---- the closed loop (closedloop.cc; swarmmap_amd/closedloop.py is the same logic in Python) ----
// What a keyframe's local-mapping job hands back to the tracking side; applied between two frames.
struct LmPacket {
    int kf = 0, first_new = 0, n_points = 0;  // rows [first_new, n_points) of the map are new
    std::vector<float> new_X;
    std::vector<int32_t> moved;  // slots local BA moved (SetWorldPos + UpdateNormalAndDepth)
    std::vector<float> moved_X, moved_N, moved_mx, moved_mn;
    std::vector<int32_t> bad, bad_repl;  // points that went bad in this job and what replaced them (-1: nothing)
    float kf_T[12] = {0};                // the keyframe's pose after local BA
    std::vector<int32_t> local_slots;    // the local map for Tracking::SearchLocalPoints from now on
    int recent_from = 0;                 // smallest slot still on mlpRecentAddedMapPoints (the counters the next keyframe brings start there)
};
struct KfSnap;
struct ClosedLoop {
    int kf_every = 5, delay = 5, n_free = 25, n_fixed = 40;
    int policy = 0;  // 0: deterministic schedule (results arrive `delay` frames after the keyframe); 1: the reference's policy
    // ---- the map as local mapping sees it (local-mapping thread only) ----
    std::vector<float> X, N, mx, mn;
    std::vector<uint8_t> D, bad;
    std::vector<int32_t> repl, ref_kf, first_kf;
    std::vector<std::vector<std::pair<int32_t, int32_t>>> obs;  // per point: (keyframe id, keypoint index), insertion order
    std::vector<std::shared_ptr<KfSnap>> kfs;
    std::vector<int32_t> recent;  // mlpRecentAddedMapPoints
    std::vector<int32_t> newly_bad;
    std::vector<int32_t> stamp;   // scratch: slot -> id of the pass that marked it
    int stamp_id = 0;
    std::vector<int64_t> lm_log;  // 12 per job (closedloop.LM_LOG_COLUMNS)
    int windows = 0, aborted = 0, skipped = 0;
    // ---- the tracking side's view (tracking thread only; positions: so_replay::mp_X) ----
    std::vector<uint8_t> tv_bad;
    std::vector<int32_t> tv_repl, tv_local;
    std::vector<uint32_t> tv_vis, tv_found;  // mnVisible / mnFound (IncreaseVisible in SearchLocalPoints, IncreaseFound after TrackLocalMap)
    std::vector<int32_t> tv_seen;            // slot -> index of the frame that holds it already (MapPoint::mnLastFrameSeen): no per-frame clearing
    int recent_from = 0;
    M4 T_ref = M4::eye(), Tlr = M4::eye();
    int last_kf_t = 0, n_kf = 0, apply_at = 0, interrupts = 0;
    bool job_pending = false;
    double wait_ms = 0.0;
    double t_packet = 0.0, handover_ms = 0.0;  // when the last packet arrived; summed time from there to the next keyframe's hand-over
    int n_handover = 0;
    double apply_ms = 0.0;  // tracking thread: applying the packets (map rows, flags, last-frame fix-ups)
    std::vector<int32_t> ref_log, kf_t;  // per frame: id of its reference keyframe; per keyframe: frame index
    std::vector<double> Tcr_log;         // per frame: pose relative to the reference keyframe (16)
    // ---- hand-over ----
    std::mutex mu;
    std::condition_variable cv;
    std::deque<LmPacket> outbox;
    bool failed = false;        // under `mu`: a local-mapping job ended in an error (so_replay::error, written under so_replay::mu,
                                // is NOT read by the waiting tracking thread: that was a race on a std::string and a lost wake-up)
    std::atomic<uint8_t> stop{0};  // mbAbortBA: set by the tracking thread (InterruptBA), cleared by the local-mapping thread before a window,
                                   // polled by so_bundle_adjust through the ABI's `const volatile uint8_t*` (stop_flag())
    const volatile uint8_t* stop_flag() const { return reinterpret_cast<const volatile uint8_t*>(&stop); }
};
// What the local-mapping thread keeps of a tracked frame that became a keyframe (KeyFrame::KeyFrame(Frame&, ...),
// code/src/KeyFrame.cc:47-72): keypoints, descriptors, pose, bindings to map points that existed before the frame
// (those created AT the frame play the part of the still untriangulated features: -1), the bound points' fields.
struct KfSnap {
    int n = 0;
    std::vector<float> x, y, angle;
    std::vector<int32_t> octave, mp;
    std::vector<uint8_t> desc;
    std::vector<float> mpX, mpN, mpMax, mpMin;
    std::vector<uint8_t> mpDesc;
    float T[12] = {0}, bounds[4] = {0, 0, 0, 0};
    int t = 0;
    int id = 0;  // closed loop: index in ClosedLoop::kfs
    // closed loop: Tracking's mnVisible / mnFound of the slots from cnt_from on, as they stand when the keyframe is handed over
    int cnt_from = 0;
    std::vector<uint32_t> cnt_vis, cnt_found;
    // DBoW2::FeatureVector stand-in, filled by the local-mapping thread
    std::vector<int32_t> node_id, off, idx;
    // the keyframe's matcher-side data in HBM (so_kframe_create), uploaded once when it joins the local-mapping thread's
    // ring and read by every later keyframe's searches
    so_kframe* dev = nullptr;
    ~KfSnap() { so_kframe_destroy(dev); }
};

struct LmJob {
    int timed = 0;
    std::shared_ptr<KfSnap> kf;  // null: window only (warm-up / no vocabulary)
    bool pre = false;            // closed loop: only the keyframe's feature vector + HBM upload (the frame is still being tracked)
};

struct so_replay {
    int device = 0, width = 0, height = 0, lba_every = 5;
    std::string host_cpus;  // the CPUs this agent's threads are pinned to (so_device_host_cpus); empty: no pinning
    int pin_slot = -1;      // which cache group of the device's node (among the live agents of this process)
    int keyframe_every = 8, local_keyframes = 0, third_pose = 1;
    double keyframe_ratio = 0.7, plane_z = 2.0;
    so_camera cam{};
    so_extractor* ex = nullptr;
    // three device-resident frames rotate: the last frame (read by the motion-model search), the current one and the
    // one being extracted ahead
    so_dframe* fr[3] = {nullptr, nullptr, nullptr};
    so_matcher* matcher = nullptr;
    so_matcher* matcher2 = nullptr;  // the local-map stage when it is enqueued behind the last-frame stage (so_track_stage_local_map_submit_after)
    so_map* map = nullptr;
    so_ba* tracker_opt = nullptr;
    so_ba* mapper_opt = nullptr;
    so_matcher* mapper_matcher = nullptr;  // the local-mapping thread's own matcher context
    std::vector<uint8_t> vocab;            // n_vocab x 32 centroid descriptors (so_replay_set_vocabulary)
    int lm_neighbours = 20;                // nn = 20, LocalMapping.cc:207,455 (monocular)
    bool lm_resident = true;               // neighbours are searched in HBM-resident form (SWARMORB_LM_RESIDENT=0: host views)
    bool lm_batch = true;                  // all searches of a keyframe as one so_matcher batch (SWARMORB_LM_BATCH=0: one by one)
    bool lm_from_map = true;               // Fuse reads the map points from the resident map by slot (SWARMORB_LM_MAP=0: staged arrays)
    std::vector<int32_t> lm_cslot;         // vpFuseCandidates as map slots
    std::deque<std::shared_ptr<KfSnap>> lm_ring;  // (local-mapping thread only)
    std::vector<int32_t> lm_stamp, lm_cstamp;     // slot -> id of the keyframe / job that marked it
    std::vector<float> lm_cX, lm_cN, lm_cmax, lm_cmin;  // vpFuseCandidates of the keyframe being processed
    std::vector<uint8_t> lm_cD, lm_cok;
    int lm_stamp_id = 0;
    double lm_stat[40] = {0};  // kLm* below; [24..31] closed loop: process, Fuse-batch kernels, apply, window gather, solver call, write-back, local BA, whole job
    std::vector<int32_t> lm_log;           // 6 ints per job: t, neighbours, triangulation matches, fused, fused back, new map points
    std::vector<int32_t> lm_tof, lm_to1, lm_to2, lm_noff;  // scratch of the triangulation step (capacity kept)
    std::vector<float> lm_txy1, lm_txy2, lm_tX, lm_nobs, lm_nX, lm_nref, lm_nls, lm_nll, lm_nnrm, lm_nmax, lm_nmin;
    std::vector<uint8_t> lm_tok;
    std::vector<const uint8_t*> frames;
    bool frames_on_device = false;
    // so_fleet_run: the extractors of the fleet this agent leads, as one group (one extraction chain for all agents)
    so_extractor_group* fleet_group = nullptr;
    so_track_group* fleet_track_group = nullptr;  // (owned by the fleet's first agent) the agents' tracking stages as one chain of launches
    so_ba_group* fleet_ba_group = nullptr;        // (owned by the fleet's first agent) the agents' local bundle adjustments as one chain of launches
    bool fleet_lm_stream_set = false;
    int fleet_offset = 0;  // so_replay_set_fleet_offset: this agent's frame index = the fleet's tick + offset
    long long fleet_ticks = 0, fleet_tick_slots = 0;  // (fleet lead) ticks so_fleet_run has driven, agents they took (so_replay_fleet_ticks)
    bool fleet_chain = false;                     // so_fleet_run drives this agent AND its stages go out with the fleet's group
    std::vector<so_extractor*> fleet_members;
    BaWindow window;
    float scale[8] = {0}, inv_sigma2[8] = {0};
    int nlevels = 8;
    float log_sf = 0.f;
    int cap = 0;
    // frame state: [0] / [1] alternate as current / last
    struct FrameHost {
        std::vector<so_keypoint> kps;
        std::vector<float> xy_un;
        std::vector<uint8_t> desc;
        std::vector<int32_t> kp_mp;
        std::vector<uint8_t> outlier;
        int n = 0;
    } fh[2];
    int cur = 0;
    int submitted = -1;  // handle index holding the frame in flight
    int last_tracked = -1;  // handle index of the frame tracked last
    bool in_flight = false;
    bool live = false;   // so_replay_run_live: no frame is extracted ahead
    int track_chain = -1;   // so_replay_set_track_chain: 1 / 0 = the tracking stages as device chains / as separate calls; -1: SWARMORB_TRACK_CHAIN (default on)
    bool lockstep = false;  // so_fleet_run drives this agent: its stages stay separate calls (the fleet batches the PoseOptimization calls of all agents)
    int n_tracked = 0;   // frames tracked so far (0: the next frame initialises the map)
    float bounds[4] = {0, 0, 0, 0};
    M4 T_last = M4::eye(), velocity = M4::eye();
    int kf_inliers = 0;
    // host copy of the map positions (PoseOptimization inputs are gathered here) + keyframe bookkeeping
    std::vector<float> mp_X;
    std::vector<float> mp_N, mp_max, mp_min;  // normal, mfMaxDistance, mfMinDistance as written to the device table
    std::vector<uint8_t> mp_desc;
    std::vector<int32_t> kf_first_slot;
    // scratch
    std::vector<int32_t> last_slot, k2l, k2m, idx, local_slot;
    std::vector<uint8_t> skip, skip_static, excluded, pose_out;
    std::vector<float> pX, pobs, pw, new_X, new_N, new_max, new_min;
    std::vector<uint8_t> new_desc;
    // log (every tracked frame)
    std::vector<float> poses;  // 12 per frame
    std::vector<int32_t> n_m2, n_m1, n_inl, n_map;
    // local-mapping thread
    std::thread mapper;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<LmJob> queue;  // timed = 1: counted window, 0: warm-up window
    int running = 0;
    bool quit = false;
    std::string error;
    // the frame being tracked (stage functions below)
    struct Step {
        double t0 = 0, t1 = 0, tm2 = 0, tp1 = 0, tm1 = 0, tp2 = 0, tp3 = 0, tmap = 0;
        double match_kernel = 0, pose_kernel = 0, pose_trials = 0, pose_points = 0, mstat[4] = {0, 0, 0, 0}, reruns = 0, wide_m2 = 0;
        int pose_calls = 0, pose_timed_calls = 0, nm2 = 0, nm1 = 0, n_local = 0, n_view = 0, keyframe = 0, hcur = 0, first_slot = 0;
        int32_t n_in = 0;
        int map_size_at_begin = 0;
        int t = 0;  // the frame's index
        bool kf_under_pose = false;  // the keyframe is made on the tracking thread while a PoseOptimization kernel runs
        bool kf_queued = false;  // closed loop: the frame went to local mapping as a keyframe already (under its last PoseOptimization)
        bool first = false, m2_submitted = false, timed_kernels = true, next_submitted = false;
        bool stage1_dev = false, stage2_dev = false;  // the stage's search -> resolve -> pose chain is on the device (so_track_stage_*)
        bool stage2_linked = false;  // ... and stage 2 went out right behind stage 1, on matcher2, before stage 1 was waited for
        int m2_rc = 0;
        float Tp[12] = {0}, Ta[12] = {0}, Tb[12] = {0}, Tc[12] = {0}, Tl[12] = {0};
        bool third_dev = false;  // fleet: the third PoseOptimization went out with the group (so_track_stage_pose_again_submit)
        M4 T = M4::eye();
    } step;
    float K4[4] = {0, 0, 0, 0};
    // statistics (timed steps only)
    double stat[48] = {0};
    unsigned lba_windows_run = 0;  // (local-mapping thread only)
    std::vector<float> frame_ms;
    std::vector<float> ba_Tcw, ba_Xw;
    std::vector<uint8_t> ba_out;
    std::unique_ptr<ClosedLoop> cl;  // non-null: the closed loop (so_replay_set_closed_loop)
    int step_timed = 0;              // the frame being tracked counts for the statistics
    std::shared_ptr<KfSnap> pre_kf;  // closed loop: the frame being tracked will be a keyframe; local mapping prepares it meanwhile
};

// closedloop.cc
int cl_lm_job(so_replay* r, const std::shared_ptr<KfSnap>& c, bool timed, so_ba_info* info);  // local-mapping thread
bool cl_frame_ready(so_replay* r, int t);  // tracking thread: would cl_frame_begin(r, t) return without waiting for a packet?
int cl_frame_begin(so_replay* r, int t);  // tracking thread, before the frame's first search: applies what local mapping handed back
void cl_keyframe_queued(so_replay* r, int t, const std::shared_ptr<KfSnap>& snap);
int cl_keyframe_featvec_upload(so_replay* r, so_matcher* m, KfSnap& c);
void cl_frame_end(so_replay* r, int t);

enum {  // indices of so_replay::stat, mirrored in bench.py
    kSteps = 0, kExtractMs, kM2Ms, kPose1Ms, kM1Ms, kPose2Ms, kPose3Ms, kMapMs, kSubmitWaitMs, kKp, kM2, kM1, kInliers,
    kMatchKernelMs, kPoseKernelMs, kPoseTrials, kPoseCalls, kPosePoints, kLbaWindows, kLbaBusyMs, kLbaGpuMs, kLbaSolveMs,
    kLbaSolves, kLocalPoints, kInView, kKeyframes, kMapPoints, kM2EnqMs, kM2WaitMs, kM1EnqMs, kM1WaitMs, kPoseTimedCalls,
    kTimedFrames, kStage0 /* 11 extractor stages */, kReruns = kStage0 + SO_EXTRACTOR_N_STAGES /* search launches: 2 per frame + exact re-runs of exhausted K-lists */,
    kWideM2 /* motion-model searches repeated with the wider window */,
    kLbaTrials /* LM trials (= reduced-system solves) of all windows; kLbaSolves / kLbaSolveMs cover the event-timed ones */
};
enum { kLmJobs = 0, kLmWallMs, kLmNodeMs, kLmTriCalls, kLmTriMs, kLmTriKernelMs, kLmTriMatches, kLmFuseCalls, kLmFuseMs,
       kLmFuseKernelMs, kLmFused, kLmFusePoints, kLmTriQueries, kLmBatchMs, kLmBatchEndMs, kLmBatchKernelMs, kLmTriangMs, kLmTriangKernelMs, kLmNewPoints,
       kLmStageTriMs, kLmStageFuseMs, kLmStageBackMs, kLmBatchEnqueueMs, kLmBatchWaitMs };

static inline so_frame_view keyframe_view(const so_replay* r, const KfSnap& k) {
    so_frame_view v;
    memset(&v, 0, sizeof(v));
    v.n = k.n;
    v.x = k.x.data(); v.y = k.y.data(); v.octave = k.octave.data(); v.angle = k.angle.data(); v.desc = k.desc.data();
    // a KeyFrame's bounds are the Frame's truncated to int, its grid is the Frame's (include/swarmorb.h, so_frame_view)
    v.min_x = (float)(int)k.bounds[0]; v.max_x = (float)(int)k.bounds[1];
    v.min_y = (float)(int)k.bounds[2]; v.max_y = (float)(int)k.bounds[3];
    v.grid_inv_w = 64.0f / (k.bounds[1] - k.bounds[0]);
    v.grid_inv_h = 48.0f / (k.bounds[3] - k.bounds[2]);
    v.has_grid_origin = 1;
    v.grid_min_x = k.bounds[0];
    v.grid_min_y = k.bounds[2];
    v.scale_factors = r->scale;
    v.nlevels = r->nlevels;
    return v;
}

// LocalMapping::ComputeF12 (code/src/LocalMapping.cc:593-609) and the epipole of SearchForTriangulation
// (code/src/ORBmatcher.cc:605-613) from the two poses; both keyframes share K.
static inline void fundamental_and_epipole(const so_replay* r, const float* T1, const float* T2, float* F12, float* ex, float* ey) {
    double R1[9], R2[9], t1[3], t2[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) { R1[3 * i + j] = T1[4 * i + j]; R2[3 * i + j] = T2[4 * i + j]; }
        t1[i] = T1[4 * i + 3]; t2[i] = T2[4 * i + 3];
    }
    double R12[9], t12[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R12[3 * i + j] = R1[3 * i] * R2[3 * j] + R1[3 * i + 1] * R2[3 * j + 1] + R1[3 * i + 2] * R2[3 * j + 2];
    for (int i = 0; i < 3; i++) t12[i] = -(R12[3 * i] * t2[0] + R12[3 * i + 1] * t2[1] + R12[3 * i + 2] * t2[2]) + t1[i];
    const double tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
    double A[9];  // t12x * R12
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A[3 * i + j] = tx[3 * i] * R12[j] + tx[3 * i + 1] * R12[3 + j] + tx[3 * i + 2] * R12[6 + j];
    const double fx = r->cam.fx, fy = r->cam.fy, cx = r->cam.cx, cy = r->cam.cy;
    const double Kit[9] = {1 / fx, 0, 0, 0, 1 / fy, 0, -cx / fx, -cy / fy, 1};  // K^-T
    const double Ki[9] = {1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1};   // K^-1
    double B[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) B[3 * i + j] = Kit[3 * i] * A[j] + Kit[3 * i + 1] * A[3 + j] + Kit[3 * i + 2] * A[6 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) F12[3 * i + j] = (float)(B[3 * i] * Ki[j] + B[3 * i + 1] * Ki[3 + j] + B[3 * i + 2] * Ki[6 + j]);
    double Cw[3], C2[3];
    for (int j = 0; j < 3; j++) Cw[j] = -(R1[j] * t1[0] + R1[3 + j] * t1[1] + R1[6 + j] * t1[2]);
    for (int i = 0; i < 3; i++) C2[i] = R2[3 * i] * Cw[0] + R2[3 * i + 1] * Cw[1] + R2[3 * i + 2] * Cw[2] + t2[i];
    *ex = (float)(fx * C2[0] / C2[2] + cx);
    *ey = (float)(fy * C2[1] / C2[2] + cy);
}
