// Tracking_glue.cc — the two tracking stages of code/src/Tracking.cc as a SwarmMap maintainer would write them over the
// device-resident frame: Tracking::TrackWithMotionModel (:714-768) and Tracking::TrackLocalMap's search + pose
// (:770-807 with SearchLocalPoints, :964-1007), each as ONE chain of launches (so_track_stage_*: search -> the
// order-dependent resolve on the device -> Optimizer::PoseOptimization over edges read in place) with ONE wait, and the
// fall-back onto the separate calls (so_track_search_* + so_pose_optimization) when a stage is handed back.
// The reference's OWN classes and members (code/include/Tracking.h, Frame.h, MapPoint.h); every object-graph side effect -
// mvpMapPoints, mvbOutlier, IncreaseVisible / IncreaseFound, mbTrackInView, mnLastFrameSeen - stays on the host where the
// reference has it, in the reference's order.  What the reference classes do not have is kept in side tables here
// (DeviceSide): the frames' so_dframe handles (three rotate: last / current / being extracted), the map's so_map and a map
// point's row in it - a maintainer would make them members (Frame::mpDeviceFrame, MapPoint::mnDeviceSlot).
//
// Compiled INSIDE the reference tree in place of the two functions (link libswarmorb.so); here it is type-checked against
// the reference's headers by tests/test_glue_typecheck.py (g++ -fsyntax-only, compile-only stand-ins for OpenCV etc.).
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "Optimizer.h"
#include "ORBmatcher.h"
#include "Tracking.h"
#include "swarmorb.h"

namespace ORB_SLAM2 {

namespace {

struct DeviceSide {  // what Frame / Map / MapPoint would carry as members
    so_matcher* matcher = nullptr;                 // the tracking thread's (one per thread, as ORBmatcher_glue.cc keeps it)
    so_ba* solver = nullptr;                       // for the separate-calls path
    so_map* map = nullptr;                         // the device table of mpMap's points
    std::unordered_map<long unsigned int, so_dframe*> frame;  // Frame::mnId -> its device-resident twin
    std::unordered_map<MapPoint*, int32_t> slot;   // MapPoint -> row of the table (so_map_write when the point is created / moved)
    // The frame whose bindings TrackWithMotionModel's stage left on the device AND whose host bindings still equal them
    // (what host/replay.cc keeps as S.stage1_dev): set only when that stage ran on the device, its result was applied and the
    // function returned true; anything that rebinds mvpMapPoints afterwards - the wider host re-search, Track()'s fall-back
    // onto TrackReferenceKeyFrame / Relocalization (code/src/Tracking.cc:321-327), the isBad() test of SearchLocalPoints -
    // leaves it unset or clears it, and TrackLocalMap then uploads the host bindings.
    bool stage1_dev = false;
    long unsigned int stage1_frame = 0;
    int32_t slot_of(MapPoint* p) const {
        if (!p) return -1;
        auto it = slot.find(p);
        return it == slot.end() ? -1 : it->second;
    }
};

DeviceSide& device_side() {
    static thread_local DeviceSide d;
    return d;
}

void pose_rows(const cv::Mat& Tcw, float* T12) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) T12[4 * r + c] = Tcw.at<float>(r, c);
}

cv::Mat pose_mat(const float* T12) {
    cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) T.at<float>(r, c) = T12[4 * r + c];
    return T;
}

// Optimizer::PoseOptimization's write-back (code/src/Optimizer.cc:377-391, 429-431) from a stage's results
void apply_pose(Frame& F, const float* Tcw12, int n_edges, const int32_t* edge_kp, const uint8_t* edge_outlier) {
    for (int e = 0; e < n_edges; e++) F.mvbOutlier[edge_kp[e]] = edge_outlier[e] != 0;
    F.SetPose(pose_mat(Tcw12));
}

}  // namespace

// code/src/Tracking.cc:714-768
bool Tracking::TrackWithMotionModel() {
    DeviceSide& D = device_side();
    D.stage1_dev = false;
    UpdateLastFrame();                                            // :719
    mCurrentFrame.SetPose(mVelocity * mLastFrame.mTcw);           // :721
    std::fill(mCurrentFrame.mvpMapPoints.begin(), mCurrentFrame.mvpMapPoints.end(), static_cast<MapPoint*>(NULL));  // :723
    const float th = mSensor != System::STEREO ? 15.f : 7.f;      // :726-730
    so_dframe* cur = D.frame[mCurrentFrame.mnId];
    so_dframe* last = D.frame[mLastFrame.mnId];
    const int N = mCurrentFrame.N;
    std::vector<int32_t> last_slot((size_t)mLastFrame.N, -1);
    for (int i = 0; i < mLastFrame.N; i++)  // ORBmatcher.cc:1248-1250: a map point that is not an outlier of the last frame
        if (mLastFrame.mvpMapPoints[i] && !mLastFrame.mvbOutlier[i]) last_slot[(size_t)i] = D.slot_of(mLastFrame.mvpMapPoints[i]);
    float Tcw[12], Tout[12];
    pose_rows(mCurrentFrame.mTcw, Tcw);
    const float K4[4] = {mCurrentFrame.fx, mCurrentFrame.fy, mCurrentFrame.cx, mCurrentFrame.cy};
    std::vector<int32_t> kp_to_last((size_t)N, -1), edge_kp((size_t)N);
    std::vector<uint8_t> edge_outlier((size_t)N);
    int32_t nmatches = 0, n_edges = 0, n_inliers = 0, info2[2];
    // search -> resolve -> PoseOptimization as one chain (:731 and :743 without the host in between)
    int rc = so_track_stage_last_frame_submit(D.matcher, cur, last, D.map, Tcw, last_slot.data(), th, /*mbCheckOrientation*/ 1, K4,
                                              mCurrentFrame.mvInvLevelSigma2.data());
    if (rc == SO_OK)
        rc = so_track_stage_wait(D.matcher, kp_to_last.data(), &nmatches, nullptr, &n_edges, edge_kp.data(), edge_outlier.data(), Tout,
                                 &n_inliers, info2);
    bool on_device = rc == SO_OK && nmatches >= 20;
    if (!on_device) {
        // handed back (a K-list ran out, sizes beyond the kernels), or :733-737's wider window: the separate calls
        if (rc != SO_OK && rc != SO_RETRY_ON_HOST) return false;
        float use_th = rc == SO_OK ? 2 * th : th;
        if (so_track_search_last_frame(D.matcher, cur, nullptr, last, D.map, Tcw, last_slot.data(), nullptr, use_th, 1, kp_to_last.data(),
                                       &nmatches) != SO_OK)
            return false;
        if (rc != SO_OK && nmatches < 20 &&
            so_track_search_last_frame(D.matcher, cur, nullptr, last, D.map, Tcw, last_slot.data(), nullptr, 2 * th, 1, kp_to_last.data(),
                                       &nmatches) != SO_OK)
            return false;
    }
    for (int k = 0; k < N; k++)
        if (kp_to_last[(size_t)k] >= 0) mCurrentFrame.mvpMapPoints[k] = mLastFrame.mvpMapPoints[kp_to_last[(size_t)k]];
    if (nmatches < 20) return false;                              // :739-740
    if (on_device) apply_pose(mCurrentFrame, Tout, n_edges, edge_kp.data(), edge_outlier.data());
    else Optimizer::PoseOptimization(&mCurrentFrame);             // :743 (Optimizer_glue.cc: so_pose_optimization)
    int nmatchesMap = 0;                                          // :745-760
    for (int i = 0; i < N; i++) {
        if (!mCurrentFrame.mvpMapPoints[i]) continue;
        if (mCurrentFrame.mvbOutlier[i]) {
            MapPoint* pMP = mCurrentFrame.mvpMapPoints[i];
            mCurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);
            mCurrentFrame.mvbOutlier[i] = false;
            pMP->mbTrackInView = false;
            pMP->mnLastFrameSeen = mCurrentFrame.mnId;
            nmatches--;
        } else if (mCurrentFrame.mvpMapPoints[i]->Observations() > 0) {
            nmatchesMap++;
        }
    }
    if (mbOnlyTracking) {                                         // :762-765 (localisation mode may run Relocalization beside this
        mbVO = nmatchesMap < 10;                                  //  function and keep either result, :348-391: no device copy is trusted)
        return nmatches > 20;
    }
    const bool ok = nmatchesMap >= 10;
    // a false return sends Track() to TrackReferenceKeyFrame, which rebinds the frame (:321-327)
    D.stage1_dev = on_device && ok;
    D.stage1_frame = mCurrentFrame.mnId;
    return ok;
}

// code/src/Tracking.cc:770-807 with SearchLocalPoints (:964-1007) folded in: the frustum test of every local point, the search
// and the pose are one chain; the points' counters are updated from what the chain reports
bool Tracking::TrackLocalMap() {
    DeviceSide& D = device_side();
    UpdateLocalMap();                                             // :774
    const int N = mCurrentFrame.N;
    // :966-978: points already matched are not searched again; they are edges of the pose problem and `excluded` keypoints
    std::vector<int32_t> kp_slot((size_t)N, -1);
    // the last-frame stage's device copy of the bindings is good for THIS frame only, and only while the host agrees with it
    bool kp_slot_on_device = D.stage1_dev && D.stage1_frame == mCurrentFrame.mnId;
    D.stage1_dev = false;
    for (int i = 0; i < N; i++) {
        MapPoint* pMP = mCurrentFrame.mvpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad()) {
            mCurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);
            kp_slot_on_device = false;  // (the device copy still carries this binding)
        } else {
            pMP->IncreaseVisible();
            pMP->mnLastFrameSeen = mCurrentFrame.mnId;
            pMP->mbTrackInView = false;
            kp_slot[(size_t)i] = D.slot_of(pMP);
        }
    }
    const int n_local = (int)mvpLocalMapPoints.size();
    std::vector<int32_t> local_slot((size_t)n_local);
    std::vector<uint8_t> skip((size_t)n_local), in_view((size_t)n_local, 0);
    for (int i = 0; i < n_local; i++) {                           // :982-990: already seen in this frame, or bad
        MapPoint* pMP = mvpLocalMapPoints[i];
        local_slot[(size_t)i] = D.slot_of(pMP);
        skip[(size_t)i] = (pMP->mnLastFrameSeen == mCurrentFrame.mnId || pMP->isBad()) ? 1 : 0;
    }
    float th = 1.f;                                               // :999-1005
    if (mSensor == System::RGBD) th = 3.f;
    if (mCurrentFrame.mnId < mnLastRelocFrameId + 2) th = 5.f;
    float Tcw[12], Tout[12];
    pose_rows(mCurrentFrame.mTcw, Tcw);
    const float K4[4] = {mCurrentFrame.fx, mCurrentFrame.fy, mCurrentFrame.cx, mCurrentFrame.cy};
    so_dframe* cur = D.frame[mCurrentFrame.mnId];
    std::vector<int32_t> kp_to_local((size_t)N, -1), edge_kp((size_t)N);
    std::vector<uint8_t> edge_outlier((size_t)N), excluded((size_t)N);
    int32_t nmatches = 0, n_edges = 0, n_inliers = 0, info2[2];
    // when kp_slot is exactly what TrackWithMotionModel's stage left on the device for this frame the chain reads it in
    // place; otherwise (host re-search, another tracking routine bound the frame, a bad point dropped) it is uploaded
    int rc = so_track_stage_local_map_submit(D.matcher, cur, kp_slot.data(), kp_slot_on_device ? 1 : 0, D.map, Tcw, n_local, local_slot.data(), 0, skip.data(), th, 0.8f,
                                             0.5f, std::log(mCurrentFrame.mfScaleFactor), K4, mCurrentFrame.mvInvLevelSigma2.data());
    if (rc == SO_OK)
        rc = so_track_stage_wait(D.matcher, kp_to_local.data(), &nmatches, in_view.data(), &n_edges, edge_kp.data(), edge_outlier.data(), Tout,
                                 &n_inliers, info2);
    const bool on_device = rc == SO_OK;
    if (!on_device) {
        if (rc != SO_RETRY_ON_HOST) return false;
        for (int i = 0; i < N; i++) excluded[(size_t)i] = kp_slot[(size_t)i] >= 0 ? 1 : 0;
        if (so_track_search_local_map(D.matcher, cur, excluded.data(), D.map, Tcw, n_local, local_slot.data(), 0, skip.data(), nullptr, th, 0.8f,
                                      0.5f, std::log(mCurrentFrame.mfScaleFactor), in_view.data(), kp_to_local.data(), &nmatches) != SO_OK)
            return false;
    }
    for (int i = 0; i < n_local; i++)                             // :991-994: isInFrustum -> IncreaseVisible
        if (in_view[(size_t)i]) mvpLocalMapPoints[i]->IncreaseVisible();
    for (int k = 0; k < N; k++)                                   // ORBmatcher.cc:115-116
        if (kp_to_local[(size_t)k] >= 0) mCurrentFrame.mvpMapPoints[k] = mvpLocalMapPoints[kp_to_local[(size_t)k]];
    if (on_device) apply_pose(mCurrentFrame, Tout, n_edges, edge_kp.data(), edge_outlier.data());
    else Optimizer::PoseOptimization(&mCurrentFrame);             // :779
    mnMatchesInliers = 0;                                         // :780-796
    for (int i = 0; i < N; i++) {
        if (!mCurrentFrame.mvpMapPoints[i]) continue;
        if (!mCurrentFrame.mvbOutlier[i]) {
            mCurrentFrame.mvpMapPoints[i]->IncreaseFound();
            if (!mbOnlyTracking) {
                if (mCurrentFrame.mvpMapPoints[i]->Observations() > 0) mnMatchesInliers++;
            } else {
                mnMatchesInliers++;
            }
        } else if (mSensor == System::STEREO) {
            mCurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);
        }
    }
    if (mCurrentFrame.mnId < mnLastRelocFrameId + mMaxFrames && mnMatchesInliers < 50) return false;  // :800-801
    return mnMatchesInliers >= 30;                                // :803-806
}

}  // namespace ORB_SLAM2
