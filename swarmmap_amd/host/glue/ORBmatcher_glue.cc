// ORBmatcher_glue.cc — the bodies a SwarmMap maintainer puts in place of code/src/ORBmatcher.cc's ten search routines:
//   SearchByProjection(Frame&, const vector<MapPoint*>&, th) (:44-121), SearchByProjection(Frame&, const Frame&, th, bMono)
//   (:1223-1354), SearchByBoW x2 (:150-262, :481-597), SearchForInitialization (:375-479), SearchForTriangulation
//   (:599-749), Fuse x2 (:751-891, :893-1009), SearchBySim3 (:1011-1221), SearchByProjection(KeyFrame*, Scw, ...)
//   (:264-373) and SearchByProjection(Frame&, KeyFrame*, set, ...) (:1356-1473)
// - the reference's OWN signatures (code/include/ORBmatcher.h:41-83) over its OWN Frame / KeyFrame / MapPoint classes:
// flatten -> one call into libswarmorb.so -> write the bindings back.  isBad() / already-found gates and every
// object-graph side effect (mvpMapPoints, AddObservation / AddMapPoint / Replace, vpReplacePoint, vpMatched) stay on the
// host where the reference has them, walked in the reference's order; candidate gathering (GetFeaturesInArea),
// DescriptorDistance, best / second selection, the greedy resolve, the ratio tests, the rotation histogram and - for
// the keyframe-side routines - the projection + gating of every map point are behind the C ABI.
//
// Compiled INSIDE the reference tree (instead of the ten functions; link libswarmorb.so).  Here it is type-checked
// against the reference's headers by tests/test_glue_typecheck.py (g++ -fsyntax-only with compile-only stand-ins for
// the OpenCV / Eigen / Boost headers the image lacks).
#include <cstring>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "ORBmatcher.h"
#include "swarmorb.h"

namespace ORB_SLAM2 {

namespace {

// The reference builds an ORBmatcher on the stack at every call site (Tracking.cc:470,623,715,998,1153,1187, ...): the
// device context cannot live in the object.  One so_matcher per calling thread: pinned staging, HBM block and stream
// survive from call to call.
so_matcher* thread_matcher() {
    static thread_local so_matcher* h = nullptr;
    if (!h && so_matcher_create(0, &h) != SO_OK) h = nullptr;
    return h;
}

struct FrameArrays {  // the parts of Frame the matcher reads, flattened (so_frame_view)
    std::vector<float> x, y, angle;
    std::vector<int32_t> octave;
    std::vector<uint8_t> excluded;
    so_frame_view view;
};

void flatten(const Frame& F, FrameArrays& A) {
    const int N = F.N;
    A.x.resize(N); A.y.resize(N); A.angle.resize(N); A.octave.resize(N); A.excluded.resize(N);
    for (int i = 0; i < N; i++) {
        const cv::KeyPoint& kp = F.mvKeysUn[i];
        A.x[i] = kp.pt.x;
        A.y[i] = kp.pt.y;
        A.octave[i] = kp.octave;
        A.angle[i] = kp.angle;
        MapPoint* p = F.mvpMapPoints[i];
        A.excluded[i] = (p && p->Observations() > 0) ? 1 : 0;  // ORBmatcher.cc:83-85, :1289-1291
    }
    so_frame_view& v = A.view;
    memset(&v, 0, sizeof(v));
    v.n = N;
    v.x = A.x.data();
    v.y = A.y.data();
    v.octave = A.octave.data();
    v.angle = A.angle.data();
    v.desc = F.mDescriptors.data;  // N x 32, CV_8U, continuous (ORBextractor.cc:773)
    v.excluded = A.excluded.data();
    v.min_x = Frame::mnMinX; v.max_x = Frame::mnMaxX; v.min_y = Frame::mnMinY; v.max_y = Frame::mnMaxY;
    v.grid_inv_w = Frame::mfGridElementWidthInv;
    v.grid_inv_h = Frame::mfGridElementHeightInv;
    v.scale_factors = F.mvScaleFactors.data();
    v.nlevels = F.mnScaleLevels;
}

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) -> so_featvec
struct FlatFeatVec {
    std::vector<int32_t> node_id, off, idx;
    so_featvec fv;
    explicit FlatFeatVec(const DBoW2::FeatureVector& f) {
        off.push_back(0);
        for (DBoW2::FeatureVector::const_iterator it = f.begin(); it != f.end(); ++it) {
            node_id.push_back((int32_t)it->first);
            for (size_t k = 0; k < it->second.size(); k++) idx.push_back((int32_t)it->second[k]);
            off.push_back((int32_t)idx.size());
        }
        fv.n_nodes = (int32_t)node_id.size();
        fv.node_id = node_id.data();
        fv.off = off.data();
        fv.idx = idx.data();
    }
};

void bound_flags(const std::vector<MapPoint*>& v, std::vector<uint8_t>& out) {
    out.resize(v.size());
    for (size_t i = 0; i < v.size(); i++) out[i] = (v[i] && !v[i]->isBad()) ? 1 : 0;
}

// The parts of KeyFrame the keyframe-side searches read.  A KeyFrame truncates the image bounds to int (code/include/
// KeyFrame.h:220) and uses those in IsInImage / GetFeaturesInArea (code/src/KeyFrame.cc:779-818), while its grid is the
// Frame's (copied at construction, KeyFrame.cc:66-72, and serialised as is, KeyFrame.h:377): cells assigned with the
// Frame's float origin.  (Only MapEnhancer's synthetic keyframes, KeyFrame.cc:74-110, rebuild the grid with the int
// origin; they never reach these routines' callers in the agent / server pipelines.)
struct KeyFrameArrays {
    std::vector<float> x, y, angle;
    std::vector<int32_t> octave;
    std::vector<uint8_t> excluded;
    so_frame_view view;
};

void flatten(KeyFrame* pKF, KeyFrameArrays& A) {
    const int N = pKF->N;
    A.x.resize(N); A.y.resize(N); A.angle.resize(N); A.octave.resize(N);
    for (int i = 0; i < N; i++) {
        const cv::KeyPoint& kp = pKF->mvKeysUn[i];
        A.x[i] = kp.pt.x;
        A.y[i] = kp.pt.y;
        A.octave[i] = kp.octave;
        A.angle[i] = kp.angle;
    }
    so_frame_view& v = A.view;
    memset(&v, 0, sizeof(v));
    v.n = N;
    v.x = A.x.data();
    v.y = A.y.data();
    v.octave = A.octave.data();
    v.angle = A.angle.data();
    v.desc = pKF->mDescriptors.data;
    v.excluded = nullptr;
    v.min_x = (float)pKF->mnMinX; v.max_x = (float)pKF->mnMaxX; v.min_y = (float)pKF->mnMinY; v.max_y = (float)pKF->mnMaxY;
    v.grid_inv_w = pKF->mfGridElementWidthInv;
    v.grid_inv_h = pKF->mfGridElementHeightInv;
    v.scale_factors = pKF->mvScaleFactors.data();
    v.nlevels = pKF->mnScaleLevels;
    v.has_grid_origin = 1;
    v.grid_min_x = Frame::mnMinX;
    v.grid_min_y = Frame::mnMinY;
}

// MapPoint keeps mfMaxDistance / mfMinDistance protected and only publishes 1.2f * / 0.8f * of them
// (code/src/MapPoint.cc:466-474); PredictScale (:476-485) needs the raw value.  A using-declaration in a derived class
// yields accessible pointers to the base's members without touching MapPoint.h (a maintainer may prefer two getters).
struct MapPointPeek : MapPoint {
    using MapPoint::mfMaxDistance;
    using MapPoint::mfMinDistance;
    using MapPoint::mMutexPos;
};

struct PointArrays {  // so_mappoint_view over a vector<MapPoint*>
    std::vector<float> Xw, normal, max_dist, min_dist;
    std::vector<uint8_t> desc, valid;
    so_mappoint_view view;
    explicit PointArrays(size_t n) : Xw(3 * n, 0.f), normal(3 * n, 0.f), max_dist(n, 0.f), min_dist(n, 0.f), desc(32 * n, 0), valid(n, 0) {
        view.n = (int32_t)n;
        view.Xw = Xw.data(); view.normal = normal.data(); view.max_dist = max_dist.data(); view.min_dist = min_dist.data();
        view.desc = desc.data(); view.valid = valid.data();
    }
    void set(size_t i, MapPoint* pMP, bool global_pos) {
        const cv::Mat p = global_pos ? pMP->GetGlobalPos() : pMP->GetWorldPos();
        const cv::Mat nrm = pMP->GetNormal();
        for (int c = 0; c < 3; c++) {
            Xw[3 * i + c] = p.at<float>(c);
            normal[3 * i + c] = nrm.at<float>(c);
        }
        {
            std::unique_lock<std::mutex> lock(pMP->*(&MapPointPeek::mMutexPos));
            max_dist[i] = pMP->*(&MapPointPeek::mfMaxDistance);
            min_dist[i] = pMP->*(&MapPointPeek::mfMinDistance);
        }
        const cv::Mat d = pMP->GetDescriptor();
        memcpy(&desc[32 * i], d.data, 32);
    }
};

void pose12(const cv::Mat& R, const cv::Mat& t, float* T) {
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = R.at<float>(r, c);
        T[4 * r + 3] = t.at<float>(r);
    }
}

void rows12(const cv::Mat& S, float* T) {  // rows 0-2 of a 4 x 4 (or 3 x 4) CV_32F
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) T[4 * r + c] = S.at<float>(r, c);
}

so_camera camera_of(float fx, float fy, float cx, float cy) {
    so_camera cam;
    memset(&cam, 0, sizeof(cam));
    cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy;
    return cam;
}

}  // namespace

// code/src/ORBmatcher.cc:44-121 (TrackLocalMap's SearchLocalPoints, Tracking.cc:1153)
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th) {
    so_matcher* h = thread_matcher();
    if (!h) return 0;
    const int M = (int)vpMapPoints.size();
    std::vector<uint8_t> in_view(M), has_obs(M), desc(32 * (size_t)M);
    std::vector<float> px(M), py(M), vc(M);
    std::vector<int32_t> lvl(M);
    for (int i = 0; i < M; i++) {
        MapPoint* p = vpMapPoints[i];
        in_view[i] = (p->mbTrackInView && !p->isBad()) ? 1 : 0;  // :52-56
        px[i] = p->mTrackProjX;
        py[i] = p->mTrackProjY;
        vc[i] = p->mTrackViewCos;
        lvl[i] = p->mnTrackScaleLevel;
        const cv::Mat d = p->GetDescriptor();
        memcpy(&desc[32 * (size_t)i], d.data, 32);
        has_obs[i] = p->Observations() > 0 ? 1 : 0;
    }
    FrameArrays A;
    flatten(F, A);
    std::vector<int32_t> kp_to_mp(F.N, -1);
    int32_t nmatches = 0;
    if (so_search_by_projection_mappoints(h, &A.view, M, in_view.data(), px.data(), py.data(), vc.data(), lvl.data(), desc.data(),
                                          has_obs.data(), th, mfNNratio, kp_to_mp.data(), &nmatches) != SO_OK)
        return 0;
    for (int k = 0; k < F.N; k++)
        if (kp_to_mp[k] >= 0) F.mvpMapPoints[k] = vpMapPoints[kp_to_mp[k]];  // :116
    return nmatches;
}

// code/src/ORBmatcher.cc:1223-1354 (TrackWithMotionModel, Tracking.cc:731,735), monocular
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
    so_matcher* h = thread_matcher();
    if (!h || !bMono) return 0;  // SwarmMap builds Examples/Monocular only
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const int L = LastFrame.N;
    std::vector<uint8_t> valid(L, 0), has_obs(L, 0), desc(32 * (size_t)L);
    std::vector<float> u(L, 0.f), v(L, 0.f), ang(L);
    std::vector<int32_t> oct(L);
    for (int i = 0; i < L; i++) {
        oct[i] = LastFrame.mvKeys[i].octave;      // :1272
        ang[i] = LastFrame.mvKeysUn[i].angle;     // :1322
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;  // :1247-1250
        const cv::Mat x3Dw = pMP->GetWorldPos();
        const cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0), yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        if (invzc < 0) continue;
        const float uu = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float vv = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (uu < CurrentFrame.mnMinX || uu > CurrentFrame.mnMaxX) continue;
        if (vv < CurrentFrame.mnMinY || vv > CurrentFrame.mnMaxY) continue;
        valid[i] = 1;
        u[i] = uu;
        v[i] = vv;
        const cv::Mat d = pMP->GetDescriptor();
        memcpy(&desc[32 * (size_t)i], d.data, 32);
        has_obs[i] = pMP->Observations() > 0 ? 1 : 0;  // (:1289-1291 once the point is bound to a keypoint)
    }
    FrameArrays A;
    flatten(CurrentFrame, A);
    std::vector<int32_t> kp_to_last(CurrentFrame.N, -1);
    int32_t nmatches = 0;
    if (so_search_by_projection_lastframe(h, &A.view, L, valid.data(), u.data(), v.data(), oct.data(), ang.data(), desc.data(),
                                          has_obs.data(), th, mbCheckOrientation ? 1 : 0, kp_to_last.data(), &nmatches) != SO_OK)
        return 0;
    for (int k = 0; k < CurrentFrame.N; k++)
        if (kp_to_last[k] >= 0) CurrentFrame.mvpMapPoints[k] = LastFrame.mvpMapPoints[kp_to_last[k]];  // :1314
    return nmatches;
}

// code/src/ORBmatcher.cc:150-262 (TrackReferenceKeyFrame, Relocalization)
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    so_matcher* h = thread_matcher();
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    if (!h) return 0;
    std::vector<uint8_t> valid1;
    bound_flags(vpMapPointsKF, valid1);  // :176-182
    const int n1 = (int)vpMapPointsKF.size(), n2 = F.N;
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; i++) a1[i] = pKF->mvKeysUn[i].angle;
    for (int i = 0; i < n2; i++) a2[i] = F.mvKeys[i].angle;  // :218
    const FlatFeatVec f1(pKF->mFeatVec), f2(F.mFeatVec);
    std::vector<int32_t> match_of_2(n2, -1);
    int32_t nmatches = 0;
    if (so_search_by_bow(h, 0, n1, pKF->mDescriptors.data, a1.data(), valid1.data(), &f1.fv, n2, F.mDescriptors.data, a2.data(),
                         nullptr, &f2.fv, mfNNratio, mbCheckOrientation ? 1 : 0, match_of_2.data(), nullptr, &nmatches) != SO_OK)
        return 0;
    for (int k = 0; k < n2; k++)
        if (match_of_2[k] >= 0) vpMapPointMatches[k] = vpMapPointsKF[match_of_2[k]];  // :214
    return nmatches;
}

// code/src/ORBmatcher.cc:481-597 (loop closing; AgentMediator::GetSim3, code/src/AgentMediator.cc:252)
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    so_matcher* h = thread_matcher();
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
    if (!h) return 0;
    std::vector<uint8_t> valid1, valid2;
    bound_flags(vpMapPoints1, valid1);  // :517-521
    bound_flags(vpMapPoints2, valid2);  // :535-541
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; i++) a1[i] = pKF1->mvKeysUn[i].angle;
    for (int i = 0; i < n2; i++) a2[i] = pKF2->mvKeysUn[i].angle;
    const FlatFeatVec f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
    std::vector<int32_t> match_of_1(n1, -1);
    int32_t nmatches = 0;
    if (so_search_by_bow(h, 1, n1, pKF1->mDescriptors.data, a1.data(), valid1.data(), &f1.fv, n2, pKF2->mDescriptors.data, a2.data(),
                         valid2.data(), &f2.fv, mfNNratio, mbCheckOrientation ? 1 : 0, nullptr, match_of_1.data(), &nmatches) != SO_OK)
        return 0;
    for (int k = 0; k < n1; k++)
        if (match_of_1[k] >= 0) vpMatches12[k] = vpMapPoints2[match_of_1[k]];  // :552
    return nmatches;
}

// code/src/ORBmatcher.cc:375-479 (Tracking::MonocularInitialization, Tracking.cc:471)
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched,
                                        std::vector<int>& vnMatches12, int windowSize) {
    so_matcher* h = thread_matcher();
    vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1);
    if (!h) return 0;
    FrameArrays A1, A2;
    flatten(F1, A1);
    flatten(F2, A2);
    A1.view.excluded = nullptr;  // no map points exist yet
    A2.view.excluded = nullptr;
    std::vector<float> prev(2 * vbPrevMatched.size());
    for (size_t i = 0; i < vbPrevMatched.size(); i++) {
        prev[2 * i] = vbPrevMatched[i].x;
        prev[2 * i + 1] = vbPrevMatched[i].y;
    }
    std::vector<int32_t> m12(F1.mvKeysUn.size(), -1);
    int32_t nmatches = 0;
    if (so_search_for_initialization(h, &A1.view, &A2.view, prev.data(), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                     m12.data(), &nmatches) != SO_OK)
        return 0;
    for (size_t i = 0; i < m12.size(); i++) vnMatches12[i] = m12[i];
    for (size_t i = 0; i < vbPrevMatched.size(); i++) vbPrevMatched[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]);  // :473-476
    return nmatches;
}

// code/src/ORBmatcher.cc:599-749 (LocalMapping::CreateNewMapPoints, LocalMapping.cc:246), monocular
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
                                       std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo) {
    so_matcher* h = thread_matcher();
    vMatchedPairs.clear();
    if (!h || bOnlyStereo) return 0;  // every keypoint is monocular (mvuRight < 0): "if (bOnlyStereo) if (!bStereo1) continue"
    // Compute epipole in second image (:605-613)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    KeyFrameArrays A1, A2;
    flatten(pKF1, A1);
    flatten(pKF2, A2);
    const int n1 = pKF1->N, n2 = pKF2->N;
    std::vector<uint8_t> free1(n1), free2(n2);
    for (int i = 0; i < n1; i++) free1[i] = pKF1->GetMapPoint(i) ? 0 : 1;  // :636-641
    for (int i = 0; i < n2; i++) free2[i] = pKF2->GetMapPoint(i) ? 0 : 1;  // :663-668
    const FlatFeatVec f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
    float F[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
    std::vector<int32_t> m12(n1, -1);
    int32_t nmatches = 0;
    if (so_search_for_triangulation(h, n1, A1.x.data(), A1.y.data(), A1.angle.data(), pKF1->mDescriptors.data, free1.data(), &f1.fv,
                                    n2, A2.x.data(), A2.y.data(), A2.octave.data(), A2.angle.data(), pKF2->mDescriptors.data,
                                    free2.data(), &f2.fv, F, ex, ey, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
                                    pKF2->mnScaleLevels, mbCheckOrientation ? 1 : 0, m12.data(), &nmatches) != SO_OK)
        return 0;
    vMatchedPairs.reserve(nmatches);  // :737-745
    for (int i = 0; i < n1; i++)
        if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    return nmatches;
}

// code/src/ORBmatcher.cc:751-891 (LocalMapping::SearchInNeighbors, LocalMapping.cc:456,481; MapManager.cc:127)
int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th) {
    so_matcher* h = thread_matcher();
    if (!h) return 0;
    float T[12];
    pose12(pKF->GetRotation(), pKF->GetTranslation(), T);
    const so_camera cam = camera_of(pKF->fx, pKF->fy, pKF->cx, pKF->cy);
    KeyFrameArrays A;
    flatten(pKF, A);
    const int nMPs = (int)vpMapPoints.size();
    PointArrays P(nMPs);
    for (int i = 0; i < nMPs; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP) continue;                                      // :770
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;    // :773
        P.valid[i] = 1;
        P.set(i, pMP, false);
    }
    std::vector<int32_t> best(nMPs, -1), dist(nMPs, 256);
    int32_t n = 0;
    if (so_fuse(h, &A.view, &cam, T, pKF->mfLogScaleFactor, pKF->mvInvLevelSigma2.data(), &P.view, th, best.data(), dist.data(), &n,
                nullptr) != SO_OK)
        return 0;
    // The searches of different map points do not read anything this loop writes; the map side effects do depend on
    // each other and run here in the reference's order, with its entry test repeated at the point's turn (an earlier
    // Replace() may have turned a later point bad).
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        const int bestIdx = best[i];
        if (bestIdx < 0) continue;  // bestDist > TH_LOW or rejected by a gate
        MapPoint* pMP = vpMapPoints[i];
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);  // :874-887
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations())
                    pMP->Replace(pMPinKF);
                else
                    pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// code/src/ORBmatcher.cc:893-1009 (LoopClosing::SearchAndFuse, LoopClosing.cc:563; MapManager)
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th,
                     std::vector<MapPoint*>& vpReplacePoint) {
    so_matcher* h = thread_matcher();
    if (!h) return 0;
    float S[12];
    rows12(Scw, S);
    const so_camera cam = camera_of(pKF->fx, pKF->fy, pKF->cx, pKF->cy);
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();  // :909
    KeyFrameArrays A;
    flatten(pKF, A);
    const int nPoints = (int)vpPoints.size();
    PointArrays P(nPoints);
    for (int i = 0; i < nPoints; i++) {
        MapPoint* pMP = vpPoints[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;  // :920
        P.valid[i] = 1;
        P.set(i, pMP, true);  // GetGlobalPos(), :924
    }
    std::vector<int32_t> best(nPoints, -1), dist(nPoints, 256);
    int32_t n = 0;
    if (so_fuse_sim3(h, &A.view, &cam, S, pKF->mfLogScaleFactor, &P.view, th, best.data(), dist.data(), &n, nullptr) != SO_OK)
        return 0;
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {  // :995-1005, in order: AddMapPoint changes what GetMapPoint returns later
        const int bestIdx = best[iMP];
        if (bestIdx < 0) continue;
        MapPoint* pMP = vpPoints[iMP];
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// code/src/ORBmatcher.cc:1011-1221 (AgentMediator::GetSim3, AgentMediator.cc:337; LoopClosing::ComputeSim3)
int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12,
                             const cv::Mat& R12, const cv::Mat& t12, const float th) {
    so_matcher* h = thread_matcher();
    if (!h) return 0;
    const so_camera cam = camera_of(pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy);  // :1013-1016
    float T1w[12], T2w[12], R[9], t[3];
    pose12(pKF1->GetRotation(), pKF1->GetTranslation(), T1w);
    pose12(pKF2->GetRotation(), pKF2->GetTranslation(), T2w);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) R[3 * r + c] = R12.at<float>(r, c);
        t[r] = t12.at<float>(r);
    }
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {  // :1040-1048
        MapPoint* pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    PointArrays P1(N1), P2(N2);
    for (int i = 0; i < N1; i++) {
        MapPoint* pMP = vpMapPoints1[i];
        if (!pMP || vbAlreadyMatched1[i] || pMP->isBad()) continue;  // :1057-1061
        P1.valid[i] = 1;
        P1.set(i, pMP, false);
    }
    for (int i = 0; i < N2; i++) {
        MapPoint* pMP = vpMapPoints2[i];
        if (!pMP || vbAlreadyMatched2[i] || pMP->isBad()) continue;  // :1133-1137
        P2.valid[i] = 1;
        P2.set(i, pMP, false);
    }
    KeyFrameArrays A1, A2;
    flatten(pKF1, A1);
    flatten(pKF2, A2);
    std::vector<int32_t> m12(N1, -1);
    int32_t nFound = 0;
    if (so_search_by_sim3(h, &A1.view, &A2.view, &cam, T1w, T2w, s12, R, t, pKF1->mfLogScaleFactor, pKF2->mfLogScaleFactor, &P1.view,
                          &P2.view, th, m12.data(), &nFound, nullptr, nullptr) != SO_OK)
        return 0;
    for (int i1 = 0; i1 < N1; i1++)
        if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];  // :1214
    return nFound;
}

// code/src/ORBmatcher.cc:264-373 (LoopClosing::ComputeSim3, LoopClosing.cc:347; AgentMediator)
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints,
                                   std::vector<MapPoint*>& vpMatched, int th) {
    so_matcher* h = thread_matcher();
    if (!h) return 0;
    float S[12];
    rows12(Scw, S);
    const so_camera cam = camera_of(pKF->fx, pKF->fy, pKF->cx, pKF->cy);
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());  // :280-281
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    KeyFrameArrays A;
    flatten(pKF, A);
    A.excluded.resize(pKF->N);
    for (int k = 0; k < pKF->N; k++) A.excluded[k] = vpMatched[k] ? 1 : 0;  // :347
    A.view.excluded = A.excluded.data();
    const int nPoints = (int)vpPoints.size();
    PointArrays P(nPoints);
    for (int i = 0; i < nPoints; i++) {
        MapPoint* pMP = vpPoints[i];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;  // :290
        P.valid[i] = 1;
        P.set(i, pMP, false);
    }
    std::vector<int32_t> kp_to_point(pKF->N, -1);
    int32_t nmatches = 0;
    if (so_search_by_projection_sim3(h, &A.view, &cam, S, pKF->mfLogScaleFactor, &P.view, th, kp_to_point.data(), &nmatches,
                                     nullptr) != SO_OK)
        return 0;
    for (int k = 0; k < pKF->N; k++)
        if (kp_to_point[k] >= 0) vpMatched[k] = vpPoints[kp_to_point[k]];  // :366
    return nmatches;
}

// code/src/ORBmatcher.cc:1356-1473 (Tracking::Relocalization, Tracking.cc:1187-)
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                                   const int ORBdist, const bool bGlobal) const {
    so_matcher* h = thread_matcher();
    if (!h) return 0;
    float T[12];
    pose12(CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3), CurrentFrame.mTcw.rowRange(0, 3).col(3), T);
    const so_camera cam = camera_of(Frame::fx, Frame::fy, Frame::cx, Frame::cy);
    FrameArrays A;
    flatten(CurrentFrame, A);
    for (int k = 0; k < CurrentFrame.N; k++) A.excluded[k] = CurrentFrame.mvpMapPoints[k] ? 1 : 0;  // :1425
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    const int n = (int)vpMPs.size();
    PointArrays P(n);
    std::vector<float> angle(n, 0.f);
    for (int i = 0; i < n; i++) {
        angle[i] = pKF->mvKeysUn[i].angle;  // :1443
        MapPoint* pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;  // :1377
        P.valid[i] = 1;
        P.set(i, pMP, bGlobal);  // :1380
    }
    std::vector<int32_t> kp_to_point(CurrentFrame.N, -1);
    int32_t nmatches = 0;
    if (so_search_by_projection_keyframe(h, &A.view, &cam, T, CurrentFrame.mfLogScaleFactor, &P.view, angle.data(), th, ORBdist,
                                         mbCheckOrientation ? 1 : 0, kp_to_point.data(), &nmatches, nullptr) != SO_OK)
        return 0;
    for (int k = 0; k < CurrentFrame.N; k++)
        if (kp_to_point[k] >= 0) CurrentFrame.mvpMapPoints[k] = vpMPs[kp_to_point[k]];  // :1439 (minus the histogram's :1465)
    return nmatches;
}

}  // namespace ORB_SLAM2
