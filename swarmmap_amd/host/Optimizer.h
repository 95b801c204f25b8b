// Optimizer.h — drop-in for the bundle-adjustment entry points of ORB_SLAM2::Optimizer
// (code/include/Optimizer.h:41-46) on libswarmorb.so.  The reference signatures take KeyFrame* / Map*; the
// adapter takes the gathered window (what Optimizer.cc:437-629 collects before it touches g2o) and returns what
// Optimizer.cc:682-739 writes back, so the glue in LocalMapping / MediatorScheduler stays a flatten + write-back.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/swarmorb.h"

namespace ORB_SLAM2 {

struct BAWindow {                 // flattened local window / whole map
    std::vector<float> Tcw;       // n_kf x 12 row-major [R|t] (KeyFrame::GetPose()), ascending mnId
    std::vector<uint8_t> fixed;   // lFixedCameras and pKFi->isFirst()
    std::vector<float> intr;      // n_kf x 4 (fx, fy, cx, cy)
    std::vector<float> Xw;        // n_mp x 3 (MapPoint::GetWorldPos()), ascending mnId
    std::vector<int32_t> edge_kf, edge_mp;  // one per observation, in insertion order
    std::vector<float> obs;       // kpUn.pt (x, y)
    std::vector<float> inv_sigma2;  // mvInvLevelSigma2[kpUn.octave]
};

struct BAResult {
    std::vector<float> Tcw;            // Converter::toCvMat(SE3quat) for every keyframe vertex
    std::vector<float> Xw;             // Converter::toCvMat(vPoint->estimate())
    std::vector<uint8_t> edge_outlier; // (pKFi, pMP) pairs to erase (Optimizer.cc:682-711)
    so_ba_info info{};
};

class Optimizer {
public:
    explicit Optimizer(int device = 0);
    ~Optimizer();
    Optimizer(const Optimizer&) = delete;
    Optimizer& operator=(const Optimizer&) = delete;
    // The reference's methods are static (call sites read Optimizer::PoseOptimization(&mCurrentFrame),
    // Optimizer::LocalBundleAdjustment(mpCurrentKeyFrame, &mbAbortBA, mpMap)); the solver context has to live
    // somewhere.  ThreadInstance() is the calling thread's context for a device, created on first use and kept until
    // ReleaseThreadInstance(): Tracking's PoseOptimization and LocalMapping's bundle adjustment get one each (and
    // with it their own streams), with no object to thread through the reference's call sites.
    static Optimizer& ThreadInstance(int device = 0);
    static void ReleaseThreadInstance();

    // void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap)
    void LocalBundleAdjustment(const BAWindow& window, bool* pbStopFlag, BAResult& out);
    // void static BundleAdjustment(vpKFs, vpMP, int nIterations = 5, bool* pbStopFlag = NULL,
    //                              const unsigned long nLoopKF = 0, const bool bRobust = true)
    void BundleAdjustment(const BAWindow& map, int nIterations, bool* pbStopFlag, bool bRobust, BAResult& out);
    // void static GlobalBundleAdjustment(Map* pMap, int nIterations, bool* pbStopFlag, nLoopKF, bRobust)
    void GlobalBundleAdjustment(const BAWindow& map, int nIterations, bool* pbStopFlag, bool bRobust, BAResult& out) {
        BundleAdjustment(map, nIterations, pbStopFlag, bRobust, out);
    }
    // int static PoseOptimization(Frame* pFrame) (Optimizer.cc:239-434): Tcw in / out is pFrame->mTcw ([R|t] 3x4),
    // one entry per matched keypoint i with pFrame->mvpMapPoints[i] != NULL: world position, mvKeysUn[i].pt,
    // mvInvLevelSigma2[octave]; mvbOutlier comes back per entry.  Returns nInitialCorrespondences - nBad.
    int PoseOptimization(float Tcw[12], const float K[4], const std::vector<float>& worldPos,
                         const std::vector<float>& obs, const std::vector<float>& invSigma2,
                         std::vector<uint8_t>& mvbOutlier);

private:
    void solve(const BAWindow& w, const so_ba_options& opt, bool* pbStopFlag, BAResult& out);
    so_ba* handle_ = nullptr;
};

}  // namespace ORB_SLAM2
