// ORBmatcher_glue.cc — the bodies a SwarmMap maintainer puts in place of the tracking-thread searches of
// code/src/ORBmatcher.cc: SearchByProjection(Frame&, const vector<MapPoint*>&, th) (:44-121),
// SearchByProjection(Frame&, const Frame&, th, bMono) (:1223-1354), and the two SearchByBoW overloads (:150-262,
// :481-597) - the reference's OWN signatures (code/include/ORBmatcher.h:41-83) over its OWN Frame / KeyFrame / MapPoint
// classes: flatten -> one call into libswarmorb.so -> write the bindings back.  The projections, the isBad() / outlier
// gates and every object-graph side effect stay on the host where the reference has them; candidate gathering
// (GetFeaturesInArea), DescriptorDistance, best / second selection, the greedy resolve, the ratio tests and the
// rotation histogram are behind the C ABI.  The remaining routines (SearchForInitialization, SearchForTriangulation,
// Fuse x2, SearchBySim3, the two relocalisation / loop searches) follow the same pattern - INTEGRATION.md, table in 2 -
// and exist as tested adapters over flattened views in swarmmap_amd/host/ORBmatcher.{h,cc}.
//
// Compiled INSIDE the reference tree (instead of the four functions; link libswarmorb.so).  Here it is type-checked
// against the reference's headers by tests/test_glue_typecheck.py (g++ -fsyntax-only with compile-only stand-ins for
// the OpenCV / Eigen / Boost headers the image lacks).
#include <cstring>
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

// code/src/ORBmatcher.cc:1223-1354 (TrackWithMotionModel, Tracking.cc:998,1014), monocular
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

}  // namespace ORB_SLAM2
