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

}  // namespace ORB_SLAM2
