// ORBmatcher.cc — see ORBmatcher.h.
#include "ORBmatcher.h"

#include <cstring>
#include <stdexcept>
#include <string>

namespace ORB_SLAM2 {

static void check(int status, const char* what) {
    if (status != SO_OK)
        throw std::runtime_error(std::string(what) + ": " + so_status_string(status) + " (" + so_last_error() + ")");
}

namespace {
constexpr int kMaxDevices = 16;
thread_local so_matcher* t_context[kMaxDevices] = {};  // one matcher context per (thread, device)
}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri, int device) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {
    if (device < 0 || device >= kMaxDevices) throw std::runtime_error("ORBmatcher: device index out of range");
    if (!t_context[device]) check(so_matcher_create(device, &t_context[device]), "so_matcher_create");
    handle_ = t_context[device];
}

ORBmatcher::~ORBmatcher() {}  // the context belongs to the thread

void ORBmatcher::ReleaseThreadContext() {
    for (so_matcher*& h : t_context) {
        if (h) so_matcher_destroy(h);
        h = nullptr;
    }
}

void ORBmatcher::SameFrameAsPreviousSearch() { check(so_matcher_reuse_frame(handle_), "so_matcher_reuse_frame"); }

int ORBmatcher::DescriptorDistance(const uint8_t* a, const uint8_t* b) {  // ORBmatcher.cc:1511-1525
    int dist = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        std::memcpy(&x, a + 8 * i, 8);
        std::memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

int ORBmatcher::SearchByProjection(const so_frame_view& F, const MapPointViews& mp, float th,
                                   std::vector<int32_t>& kp_to_mp) {
    kp_to_mp.assign((size_t)F.n, -1);
    int32_t nmatches = 0;
    check(so_search_by_projection_mappoints(handle_, &F, (int32_t)mp.proj_x.size(), mp.in_view.data(), mp.proj_x.data(),
                                            mp.proj_y.data(), mp.view_cos.data(), mp.pred_level.data(), mp.desc.data(),
                                            mp.has_obs.data(), th, mfNNratio, kp_to_mp.data(), &nmatches),
          "so_search_by_projection_mappoints");
    return nmatches;
}

int ORBmatcher::SearchByProjection(const so_frame_view& cur, const LastFrameViews& last, float th,
                                   std::vector<int32_t>& kp_to_last) {
    kp_to_last.assign((size_t)cur.n, -1);
    int32_t nmatches = 0;
    check(so_search_by_projection_lastframe(handle_, &cur, (int32_t)last.u.size(), last.valid.data(), last.u.data(),
                                            last.v.data(), last.octave.data(), last.angle.data(), last.desc.data(),
                                            last.has_obs.data(), th, mbCheckOrientation ? 1 : 0, kp_to_last.data(),
                                            &nmatches),
          "so_search_by_projection_lastframe");
    return nmatches;
}

int ORBmatcher::SearchForInitialization(const so_frame_view& F1, const so_frame_view& F2,
                                        std::vector<float>& vbPrevMatched, std::vector<int32_t>& vnMatches12,
                                        int windowSize) {
    vnMatches12.assign((size_t)F1.n, -1);
    int32_t nmatches = 0;
    check(so_search_for_initialization(handle_, &F1, &F2, vbPrevMatched.data(), windowSize, mfNNratio,
                                       mbCheckOrientation ? 1 : 0, vnMatches12.data(), &nmatches),
          "so_search_for_initialization");
    return nmatches;
}

int ORBmatcher::SearchByBoW(int variant, int n1, const uint8_t* desc1, const float* angle1, const uint8_t* valid1,
                            const so_featvec& fv1, int n2, const uint8_t* desc2, const float* angle2,
                            const uint8_t* valid2, const so_featvec& fv2, std::vector<int32_t>& match_of_2,
                            std::vector<int32_t>& match_of_1) {
    match_of_2.assign((size_t)n2, -1);
    match_of_1.assign((size_t)n1, -1);
    int32_t nmatches = 0;
    check(so_search_by_bow(handle_, variant, n1, desc1, angle1, valid1, &fv1, n2, desc2, angle2, valid2, &fv2,
                           mfNNratio, mbCheckOrientation ? 1 : 0, match_of_2.data(), match_of_1.data(), &nmatches),
          "so_search_by_bow");
    return nmatches;
}

void ORBmatcher::SearchWindowBest(const so_frame_view& KF, int nq, const uint8_t* valid, const float* u, const float* v,
                                  const float* radius, const int32_t* pred_level, const uint8_t* desc, bool chi2_gate,
                                  const float* inv_sigma2, std::vector<int32_t>& best_idx,
                                  std::vector<int32_t>& best_dist) {
    best_idx.assign((size_t)nq, -1);
    best_dist.assign((size_t)nq, 256);
    check(so_search_window_best(handle_, &KF, nq, valid, u, v, radius, pred_level, desc, chi2_gate ? 1 : 0, inv_sigma2,
                                best_idx.data(), best_dist.data()),
          "so_search_window_best");
}

int ORBmatcher::SearchForTriangulation(const KeyFrameFeatures& kf1, const so_featvec& fv1, const KeyFrameFeatures& kf2,
                                       const so_featvec& fv2, const float F12[9], float ex, float ey,
                                       const std::vector<float>& sf2, const std::vector<float>& ls2,
                                       std::vector<std::pair<size_t, size_t>>& vMatchedPairs) {
    std::vector<int32_t> m12((size_t)kf1.n, -1);
    int32_t nmatches = 0;
    check(so_search_for_triangulation(handle_, kf1.n, kf1.x, kf1.y, kf1.angle, kf1.desc, kf1.free_, &fv1, kf2.n, kf2.x,
                                      kf2.y, kf2.octave, kf2.angle, kf2.desc, kf2.free_, &fv2, F12, ex, ey, sf2.data(),
                                      ls2.data(), (int32_t)sf2.size(), mbCheckOrientation ? 1 : 0, m12.data(), &nmatches),
          "so_search_for_triangulation");
    vMatchedPairs.clear();  // :737-745
    vMatchedPairs.reserve((size_t)nmatches);
    for (size_t i = 0; i < m12.size(); i++)
        if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
    return nmatches;
}

static int count_within(const std::vector<int32_t>& idx, const std::vector<int32_t>& dist, int th) {
    int n = 0;
    for (size_t i = 0; i < idx.size(); i++) n += (idx[i] >= 0 && dist[i] <= th) ? 1 : 0;
    return n;
}

int ORBmatcher::Fuse(const so_frame_view& KF, const WindowQueries& q, const std::vector<float>& inv_sigma2,
                     std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) {
    SearchWindowBest(KF, q.size(), q.valid.data(), q.u.data(), q.v.data(), q.radius.data(), q.pred_level.data(),
                     q.desc.data(), true, inv_sigma2.data(), bestIdx, bestDist);
    return count_within(bestIdx, bestDist, TH_LOW);  // "if(bestDist<=TH_LOW)", :871
}

int ORBmatcher::Fuse(const so_frame_view& KF, const WindowQueries& q, std::vector<int32_t>& bestIdx,
                     std::vector<int32_t>& bestDist) {
    SearchWindowBest(KF, q.size(), q.valid.data(), q.u.data(), q.v.data(), q.radius.data(), q.pred_level.data(),
                     q.desc.data(), false, nullptr, bestIdx, bestDist);
    return count_within(bestIdx, bestDist, TH_LOW);  // :995
}

int ORBmatcher::SearchBySim3(const so_frame_view& KF1, const so_frame_view& KF2, const WindowQueries& q12,
                             const WindowQueries& q21, std::vector<int32_t>& vnMatch12) {
    std::vector<int32_t> i2, d2, i1, d1;
    SearchWindowBest(KF2, q12.size(), q12.valid.data(), q12.u.data(), q12.v.data(), q12.radius.data(),
                     q12.pred_level.data(), q12.desc.data(), false, nullptr, i2, d2);
    SearchWindowBest(KF1, q21.size(), q21.valid.data(), q21.u.data(), q21.v.data(), q21.radius.data(),
                     q21.pred_level.data(), q21.desc.data(), false, nullptr, i1, d1);
    std::vector<int32_t> vnMatch1((size_t)q12.size(), -1), vnMatch2((size_t)q21.size(), -1);
    for (size_t i = 0; i < vnMatch1.size(); i++)
        if (i2[i] >= 0 && d2[i] <= TH_HIGH) vnMatch1[i] = i2[i];  // :1113-1116
    for (size_t i = 0; i < vnMatch2.size(); i++)
        if (i1[i] >= 0 && d1[i] <= TH_HIGH) vnMatch2[i] = i1[i];  // :1192-1194
    vnMatch12.assign(vnMatch1.size(), -1);
    int nFound = 0;
    for (size_t k = 0; k < vnMatch1.size(); k++) {  // "Check agreement", :1199-1211
        const int idx2 = vnMatch1[k];
        if (idx2 >= 0 && idx2 < (int)vnMatch2.size() && vnMatch2[(size_t)idx2] == (int)k) {
            vnMatch12[k] = idx2;
            nFound++;
        }
    }
    return nFound;
}

int ORBmatcher::SearchByProjection(const so_frame_view& KF, const WindowQueries& q, std::vector<int32_t>& kp_to_point) {
    kp_to_point.assign((size_t)KF.n, -1);
    int32_t nmatches = 0;
    check(so_search_window_greedy(handle_, &KF, q.size(), q.valid.data(), q.u.data(), q.v.data(), q.radius.data(),
                                  q.min_level.data(), q.max_level.data(), q.desc.data(), nullptr, TH_LOW, 0,
                                  kp_to_point.data(), &nmatches),
          "so_search_window_greedy");
    return nmatches;
}

int ORBmatcher::SearchByProjection(const so_frame_view& F, const WindowQueries& q, int ORBdist,
                                   std::vector<int32_t>& kp_to_point) {
    kp_to_point.assign((size_t)F.n, -1);
    int32_t nmatches = 0;
    check(so_search_window_greedy(handle_, &F, q.size(), q.valid.data(), q.u.data(), q.v.data(), q.radius.data(),
                                  q.min_level.data(), q.max_level.data(), q.desc.data(), q.angle.data(), ORBdist,
                                  mbCheckOrientation ? 1 : 0, kp_to_point.data(), &nmatches),
          "so_search_window_greedy");
    return nmatches;
}

static so_camera camera_of(const ORBmatcher::Calibration& c) {
    so_camera cam{};
    cam.fx = c.fx; cam.fy = c.fy; cam.cx = c.cx; cam.cy = c.cy;
    return cam;
}

int ORBmatcher::Fuse(const so_frame_view& KF, const Calibration& cal, const Pose& Tcw, const MapPointFields& mp, float th,
                     std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) {
    const so_camera cam = camera_of(cal);
    const so_mappoint_view v = mp.view();
    bestIdx.assign((size_t)mp.size(), -1);
    bestDist.assign((size_t)mp.size(), 256);
    int32_t nFused = 0;
    check(so_fuse(handle_, &KF, &cam, Tcw.m, cal.mfLogScaleFactor, cal.mvInvLevelSigma2.data(), &v, th, bestIdx.data(),
                  bestDist.data(), &nFused, nullptr),
          "so_fuse");
    return nFused;
}

int ORBmatcher::Fuse(const so_frame_view& KF, const Calibration& cal, const Sim3& Scw, const MapPointFields& mp, float th,
                     std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) {
    const so_camera cam = camera_of(cal);
    const so_mappoint_view v = mp.view();
    bestIdx.assign((size_t)mp.size(), -1);
    bestDist.assign((size_t)mp.size(), 256);
    int32_t nFused = 0;
    check(so_fuse_sim3(handle_, &KF, &cam, Scw.m, cal.mfLogScaleFactor, &v, th, bestIdx.data(), bestDist.data(), &nFused,
                       nullptr),
          "so_fuse_sim3");
    return nFused;
}

int ORBmatcher::SearchBySim3(const so_frame_view& KF1, const Calibration& cal1, const Pose& T1w, const so_frame_view& KF2,
                             const Calibration& cal2, const Pose& T2w, const MapPointFields& mp1, const MapPointFields& mp2,
                             float s12, const float R12[9], const float t12[3], float th, std::vector<int32_t>& vnMatch12) {
    const so_camera cam = camera_of(cal1);  // pKF1's intrinsics serve both directions, :1013-1016
    const so_mappoint_view v1 = mp1.view(), v2 = mp2.view();
    vnMatch12.assign((size_t)mp1.size(), -1);
    int32_t nFound = 0;
    check(so_search_by_sim3(handle_, &KF1, &KF2, &cam, T1w.m, T2w.m, s12, R12, t12, cal1.mfLogScaleFactor,
                            cal2.mfLogScaleFactor, &v1, &v2, th, vnMatch12.data(), &nFound, nullptr, nullptr),
          "so_search_by_sim3");
    return nFound;
}

int ORBmatcher::SearchByProjection(const so_frame_view& KF, const Calibration& cal, const Sim3& Scw, const MapPointFields& mp,
                                   int th, std::vector<int32_t>& kp_to_point) {
    const so_camera cam = camera_of(cal);
    const so_mappoint_view v = mp.view();
    kp_to_point.assign((size_t)KF.n, -1);
    int32_t nmatches = 0;
    check(so_search_by_projection_sim3(handle_, &KF, &cam, Scw.m, cal.mfLogScaleFactor, &v, th, kp_to_point.data(), &nmatches,
                                       nullptr),
          "so_search_by_projection_sim3");
    return nmatches;
}

int ORBmatcher::SearchByProjection(const so_frame_view& F, const Calibration& cal, const Pose& Tcw, const MapPointFields& mp,
                                   float th, int ORBdist, std::vector<int32_t>& kp_to_point) {
    const so_camera cam = camera_of(cal);
    const so_mappoint_view v = mp.view();
    kp_to_point.assign((size_t)F.n, -1);
    int32_t nmatches = 0;
    check(so_search_by_projection_keyframe(handle_, &F, &cam, Tcw.m, cal.mfLogScaleFactor, &v, mp.angle.data(), th, ORBdist,
                                           mbCheckOrientation ? 1 : 0, kp_to_point.data(), &nmatches, nullptr),
          "so_search_by_projection_keyframe");
    return nmatches;
}

std::vector<int32_t> ORBmatcher::ComputeDistinctiveDescriptors(const std::vector<int32_t>& offsets,
                                                               const std::vector<uint8_t>& descriptors) {
    const int n = (int)offsets.size() - 1;
    std::vector<int32_t> best((size_t)(n > 0 ? n : 0), -1);
    if (n > 0)
        check(so_distinctive_descriptors(handle_, n, offsets.data(), descriptors.data(), best.data(), nullptr),
              "so_distinctive_descriptors");
    return best;
}

}  // namespace ORB_SLAM2
