// Optimizer_glue.cc — the bodies a SwarmMap maintainer puts in place of code/src/Optimizer.cc:42-237 (BundleAdjustment,
// GlobalBundleAdjustment), :239-434 (PoseOptimization) and :436-740 (LocalBundleAdjustment): the reference's OWN
// signatures (code/include/Optimizer.h:41-47) and its OWN KeyFrame / MapPoint / Map / Frame classes, g2o replaced by one
// call into libswarmorb.so each.  Gathering the window from the object graph and writing the results back under the map
// mutex stay on the host exactly where the reference has them (:437-482, :713-739); graph construction, both optimize()
// calls and the outlier pass in between are so_bundle_adjust.
//
// This file is compiled INSIDE the reference tree (add it to libslam_core's sources instead of the three functions, link
// libswarmorb.so).  Here it is type-checked against the reference's headers by tests/test_glue_typecheck.py
// (g++ -fsyntax-only; the image has no OpenCV / Eigen / Boost, so compile-only stand-ins declare their names).
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "Optimizer.h"
#include "swarmorb.h"

namespace ORB_SLAM2 {

namespace {

// one solver context per calling thread and device: the reference's functions are static and are called from the
// LocalMapping thread of every agent, the MediatorScheduler thread, Tracking (initialisation) and the GBA thread
so_ba* thread_solver() {
    static thread_local so_ba* h = nullptr;
    if (!h && so_ba_create(0, &h) != SO_OK) h = nullptr;
    return h;
}

void pose_rows(const cv::Mat& T, std::vector<float>& out) {  // CV_32F 4x4 -> 12 floats [R|t]
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) out.push_back(T.at<float>(r, c));
}

cv::Mat pose_mat(const float* T12) {  // Converter::toCvMat(SE3Quat): 4x4 CV_32F
    cv::Mat T = cv::Mat::eye(4, 4, CV_32F);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) T.at<float>(r, c) = T12[4 * r + c];
    return T;
}

cv::Mat point_mat(const float* X3) {  // Converter::toCvMat(Vector3d): 3x1 CV_32F
    cv::Mat X(3, 1, CV_32F);
    for (int r = 0; r < 3; r++) X.at<float>(r) = X3[r];
    return X;
}

struct Flat {  // the flattened problem and who owns every edge
    std::vector<KeyFrame*> kfs;  // ascending mnId: the Hessian order of sparse_optimizer.cpp:166-190
    std::vector<MapPoint*> mps;  // ascending mnId
    std::map<KeyFrame*, int> kf_index;
    std::vector<float> Tcw, intr, Xw, obs, inv_sigma2;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> edge_kf, edge_mp;
    std::vector<std::pair<KeyFrame*, MapPoint*> > edge_owner;
    so_ba_problem problem() const {
        so_ba_problem p;
        p.n_poses = (int32_t)kfs.size();
        p.Tcw = Tcw.data();
        p.fixed = fixed.data();
        p.intr = intr.data();
        p.n_points = (int32_t)mps.size();
        p.Xw = Xw.data();
        p.n_edges = (int32_t)edge_kf.size();
        p.edge_pose = edge_kf.data();
        p.edge_point = edge_mp.data();
        p.obs = obs.data();
        p.inv_sigma2 = inv_sigma2.data();
        return p;
    }
};

void add_observations(Flat& w, size_t j, MapPoint* pMP, unsigned long maxKFid, bool bound_by_id) {
    const std::map<KeyFrame*, size_t> observations = pMP->GetObservations();
    for (std::map<KeyFrame*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
        KeyFrame* pKFi = mit->first;
        if (pKFi->isBad() || (bound_by_id && pKFi->mnId > maxKFid)) continue;  // Optimizer.cc:103-104 / :564
        std::map<KeyFrame*, int>::const_iterator it = w.kf_index.find(pKFi);
        if (it == w.kf_index.end()) continue;
        if (pKFi->mvuRight[mit->second] >= 0) continue;  // stereo observation: SwarmMap builds monocular only
        const cv::KeyPoint& kpUn = pKFi->mvKeysUn[mit->second];
        w.edge_kf.push_back(it->second);
        w.edge_mp.push_back((int32_t)j);
        w.obs.push_back(kpUn.pt.x);
        w.obs.push_back(kpUn.pt.y);
        w.inv_sigma2.push_back(pKFi->mvInvLevelSigma2[kpUn.octave]);
        w.edge_owner.push_back(std::make_pair(pKFi, pMP));
    }
}

}  // namespace

// code/src/Optimizer.cc:436-740
void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap) {
    // Local KeyFrames: first breadth search from the current keyframe (:437-449)
    std::list<KeyFrame*> lLocalKeyFrames;
    lLocalKeyFrames.push_back(pKF);
    std::set<unsigned long> sLocalKeyFrameIds;
    sLocalKeyFrameIds.insert(pKF->mnId);
    const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (size_t i = 0; i < vNeighKFs.size(); i++) {
        KeyFrame* pKFi = vNeighKFs[i];
        sLocalKeyFrameIds.insert(pKFi->mnId);
        if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
    }
    // Local MapPoints seen in local keyframes (:451-465)
    std::list<MapPoint*> lLocalMapPoints;
    std::set<unsigned long> sLocalMapPointIds;
    for (std::list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); ++lit) {
        const std::vector<MapPoint*> vpMPs = (*lit)->GetMapPointMatches();
        for (size_t i = 0; i < vpMPs.size(); i++) {
            MapPoint* pMP = vpMPs[i];
            if (!pMP || pMP->isBad()) continue;
            if (sLocalMapPointIds.insert(pMP->mnId).second) lLocalMapPoints.push_back(pMP);
        }
    }
    // Fixed keyframes: they see local map points but are not local keyframes (:467-482)
    std::list<KeyFrame*> lFixedCameras;
    std::set<unsigned long> sFixedKeyFrameIds;
    for (std::list<MapPoint*>::iterator lit = lLocalMapPoints.begin(); lit != lLocalMapPoints.end(); ++lit) {
        const std::map<KeyFrame*, size_t> observations = (*lit)->GetObservations();
        for (std::map<KeyFrame*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
            KeyFrame* pKFi = mit->first;
            if (sLocalKeyFrameIds.count(pKFi->mnId) == 0 && sFixedKeyFrameIds.count(pKFi->mnId) == 0) {
                sFixedKeyFrameIds.insert(pKFi->mnId);
                if (pKFi->isBad()) continue;
                lFixedCameras.push_back(pKFi);
            }
        }
    }
    // flatten: vertices in ascending id order (that is the Hessian order g2o derives), edges in the reference's
    // insertion order (:545-628: map points in list order, their observations in std::map order)
    Flat w;
    w.kfs.assign(lLocalKeyFrames.begin(), lLocalKeyFrames.end());
    w.kfs.insert(w.kfs.end(), lFixedCameras.begin(), lFixedCameras.end());
    std::sort(w.kfs.begin(), w.kfs.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
    for (size_t i = 0; i < w.kfs.size(); i++) {
        KeyFrame* kf = w.kfs[i];
        w.kf_index[kf] = (int)i;
        pose_rows(kf->GetPose(), w.Tcw);  // Converter::toSE3Quat happens behind the C ABI, on the same floats
        w.fixed.push_back((kf->isFirst() || sFixedKeyFrameIds.count(kf->mnId)) ? 1 : 0);  // :504, :516
        w.intr.push_back(kf->fx);
        w.intr.push_back(kf->fy);
        w.intr.push_back(kf->cx);
        w.intr.push_back(kf->cy);
    }
    w.mps.assign(lLocalMapPoints.begin(), lLocalMapPoints.end());
    std::sort(w.mps.begin(), w.mps.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
    for (size_t j = 0; j < w.mps.size(); j++) {
        const cv::Mat X = w.mps[j]->GetWorldPos();
        for (int r = 0; r < 3; r++) w.Xw.push_back(X.at<float>(r));
        add_observations(w, j, w.mps[j], 0, false);
    }
    if (pbStopFlag && *pbStopFlag) return;  // :630-632
    so_ba* h = thread_solver();
    if (!h) return;
    so_ba_options opt;
    so_ba_options_local(&opt);  // optimize(5), outlier pass at 5.991, optimize(10), Huber sqrt(5.991)
    std::vector<float> Tcw_out(w.Tcw.size()), Xw_out(w.Xw.size());
    std::vector<uint8_t> edge_outlier(w.edge_kf.size());
    so_ba_info info;
    const so_ba_problem p = w.problem();
    // bool is one byte holding 0 / 1 on every ABI SwarmMap builds for: the flag is polled as the reference's g2o does
    if (so_bundle_adjust(h, &p, &opt, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), Tcw_out.data(), Xw_out.data(),
                         edge_outlier.data(), nullptr, &info) != SO_OK)
        return;
    // write-back under the map mutex (:711-739)
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
    for (size_t e = 0; e < w.edge_owner.size(); e++) {
        MapPoint* pMP = w.edge_owner[e].second;
        if (pMP->isBad() || !edge_outlier[e]) continue;  // :686-695
        KeyFrame* pKFi = w.edge_owner[e].first;
        pKFi->EraseMapPointMatch(pMP);
        pMP->EraseObservation(pKFi);
    }
    for (std::list<KeyFrame*>::iterator lit = lLocalKeyFrames.begin(); lit != lLocalKeyFrames.end(); ++lit)
        (*lit)->SetPose(pose_mat(&Tcw_out[12 * (size_t)w.kf_index[*lit]]));
    for (size_t j = 0; j < w.mps.size(); j++) {
        w.mps[j]->SetWorldPos(point_mat(&Xw_out[3 * j]));
        w.mps[j]->UpdateNormalAndDepth();
    }
}

// code/src/Optimizer.cc:42-47
void Optimizer::GlobalBundleAdjustment(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF,
                                       const bool bRobust) {
    std::vector<KeyFrame*> vpKFs = pMap->GetAllKeyFrames();
    std::vector<MapPoint*> vpMP = pMap->GetAllMapPoints();
    BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}

// code/src/Optimizer.cc:50-237
void Optimizer::BundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations,
                                 bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust, const bool bGlobal) {
    Flat w;
    unsigned long maxKFid = 0;
    for (size_t i = 0; i < vpKFs.size(); i++)
        if (!vpKFs[i]->isBad()) {
            w.kfs.push_back(vpKFs[i]);
            maxKFid = std::max(maxKFid, (unsigned long)vpKFs[i]->mnId);
        }
    std::sort(w.kfs.begin(), w.kfs.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
    for (size_t i = 0; i < w.kfs.size(); i++) {
        KeyFrame* kf = w.kfs[i];
        w.kf_index[kf] = (int)i;
        pose_rows(bGlobal ? kf->GetGlobalPose() : kf->GetPose(), w.Tcw);  // :70
        w.fixed.push_back(kf->isFirst() ? 1 : 0);                            // :73
        w.intr.push_back(kf->fx);
        w.intr.push_back(kf->fy);
        w.intr.push_back(kf->cx);
        w.intr.push_back(kf->cy);
    }
    std::vector<size_t> order;  // indices into vpMP of the points that become vertices, ascending mnId
    for (size_t i = 0; i < vpMP.size(); i++)
        if (!vpMP[i]->isBad()) order.push_back(i);
    std::sort(order.begin(), order.end(), [&vpMP](size_t a, size_t b) { return vpMP[a]->mnId < vpMP[b]->mnId; });
    std::vector<uint8_t> included;
    for (size_t k = 0; k < order.size(); k++) {
        MapPoint* pMP = vpMP[order[k]];
        const cv::Mat X = bGlobal ? pMP->GetGlobalPos() : pMP->GetWorldPos();  // :90
        w.mps.push_back(pMP);
        for (int r = 0; r < 3; r++) w.Xw.push_back(X.at<float>(r));
        const size_t before = w.edge_kf.size();
        add_observations(w, k, pMP, maxKFid, true);
        included.push_back(w.edge_kf.size() > before ? 1 : 0);  // vbNotIncludedMP, :171-176
    }
    so_ba* h = thread_solver();
    if (!h) return;
    so_ba_options opt;
    so_ba_options_global(&opt, nIterations, bRobust ? 1 : 0);  // optimize(nIterations), Huber sqrt(5.99) when bRobust
    std::vector<float> Tcw_out(w.Tcw.size()), Xw_out(w.Xw.size());
    so_ba_info info;
    const so_ba_problem p = w.problem();
    if (so_bundle_adjust(h, &p, &opt, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), Tcw_out.data(), Xw_out.data(),
                         nullptr, nullptr, &info) != SO_OK)
        return;
    // recover optimised data (:186-236)
    for (size_t i = 0; i < w.kfs.size(); i++) {
        KeyFrame* kf = w.kfs[i];
        const cv::Mat T = pose_mat(&Tcw_out[12 * i]);
        if (nLoopKF == 0) {
            if (bGlobal) kf->SetGlobalPose(T); else kf->SetPose(T);
        } else {
            kf->mTcwGBA.create(4, 4, CV_32F);
            T.copyTo(kf->mTcwGBA);
            kf->mnBAGlobalForKF = nLoopKF;
        }
    }
    for (size_t k = 0; k < w.mps.size(); k++) {
        if (!included[k]) continue;
        MapPoint* pMP = w.mps[k];
        const cv::Mat X = point_mat(&Xw_out[3 * k]);
        if (nLoopKF == 0) {
            if (bGlobal) pMP->SetGlobalPos(X); else pMP->SetWorldPos(X);
            pMP->UpdateNormalAndDepth();
        } else {
            pMP->mPosGBA.create(3, 1, CV_32F);
            X.copyTo(pMP->mPosGBA);
            pMP->mnBAGlobalForKF = nLoopKF;
        }
    }
}

// code/src/Optimizer.cc:239-434: motion-only BA of one frame; four rounds of optimize(10) with outlier re-classification
int Optimizer::PoseOptimization(Frame* pFrame, const bool bGlobal) {
    const int N = pFrame->N;
    std::vector<int> idx;
    std::vector<float> Xw, obs, w;
    {
        std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);  // :283
        for (int i = 0; i < N; i++) {
            MapPoint* pMP = pFrame->mvpMapPoints[i];
            if (!pMP || pFrame->mvuRight[i] >= 0) continue;
            pFrame->mvbOutlier[i] = false;  // :290
            const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
            const cv::Mat X = bGlobal ? pMP->GetGlobalPos() : pMP->GetWorldPos();
            idx.push_back(i);
            for (int r = 0; r < 3; r++) Xw.push_back(X.at<float>(r));
            obs.push_back(kpUn.pt.x);
            obs.push_back(kpUn.pt.y);
            w.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
        }
    }
    const int n = (int)idx.size();
    if (n < 3) return 0;  // :358-359
    so_ba* h = thread_solver();
    if (!h) return 0;
    std::vector<float> Tin, Tout(12);
    pose_rows(pFrame->mTcw, Tin);
    const float intr[4] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy};
    std::vector<uint8_t> outlier((size_t)n);
    int32_t n_inliers = 0;
    if (so_pose_optimization(h, Tin.data(), intr, n, Xw.data(), obs.data(), w.data(), Tout.data(), outlier.data(), &n_inliers,
                             nullptr) != SO_OK)
        return 0;
    for (int k = 0; k < n; k++) pFrame->mvbOutlier[idx[k]] = outlier[k] != 0;  // :377-391
    pFrame->SetPose(pose_mat(Tout.data()));                                    // :429-431
    return n_inliers;                                                          // nInitialCorrespondences - nBad
}

}  // namespace ORB_SLAM2
