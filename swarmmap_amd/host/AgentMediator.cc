// AgentMediator.cc — see AgentMediator.h.
#include "AgentMediator.h"

#include <stdexcept>
#include <string>

namespace ORB_SLAM2 {

static void check(int status, const char* what) {
    if (status != SO_OK)
        throw std::runtime_error(std::string(what) + ": " + so_status_string(status) + " (" + so_last_error() + ")");
}

AgentMediator::AgentMediator(int capacityKeyFrames, int maxKeypoints, int device) : max_keypoints_(maxKeypoints) {
    check(so_kfstore_create(device, capacityKeyFrames, maxKeypoints, &handle_), "so_kfstore_create");
}

AgentMediator::~AgentMediator() { so_kfstore_destroy(handle_); }

std::vector<uint8_t> AgentMediator::pack(const KeyFrameView& kf) const {
    so_keyframe_header h{};
    h.agent_id = kf.mnClientId;
    h.n_keypoints = kf.N;
    h.keyframe_id = kf.mnId;
    h.timestamp = kf.mTimeStamp;
    for (int i = 0; i < 12; i++) h.Tcw[i] = kf.Tcw ? kf.Tcw[i] : 0.f;
    for (int i = 0; i < 4; i++) h.K[i] = kf.K ? kf.K[i] : 0.f;
    std::vector<uint8_t> rec(so_keyframe_record_size2(kf.N));
    check(so_keyframe_record_pack2(&h, kf.xy, kf.angle, kf.octave, kf.descriptors, kf.mapPointId, rec.data(), rec.size()),
          "so_keyframe_record_pack2");
    return rec;
}

int AgentMediator::AddKeyFrame(const KeyFrameView& kf) {
    const std::vector<uint8_t> rec = pack(kf);
    int32_t slot = -1;
    check(so_kfstore_append(handle_, rec.data(), rec.size(), 1, &slot), "so_kfstore_append");
    return slot;
}

int AgentMediator::KeyFramesInStore() const {
    int32_t n = 0;
    check(so_kfstore_size(handle_, &n, nullptr), "so_kfstore_size");
    return n;
}

std::vector<OverlapCandidate> AgentMediator::CheckOverlapCandidates(const KeyFrameView& kf, float nnratio, bool checkOri,
                                                                    int minVotes, int minMatches, int maxCandidates) {
    const std::vector<uint8_t> rec = pack(kf);
    so_kf_search_params p;
    p.th_low = 50;  // ORBmatcher::TH_LOW
    p.nn_ratio = nnratio;
    p.check_orientation = checkOri ? 1 : 0;
    p.min_votes = minVotes;
    p.min_matches = minMatches;
    p.max_candidates = maxCandidates;
    std::vector<so_kf_candidate> out((size_t)(maxCandidates > 0 ? maxCandidates : 1));
    std::vector<int32_t> pairs(out.size() * (size_t)(kf.N > 0 ? kf.N : 1), -1);
    int32_t n_out = 0;
    check(so_kfstore_search(handle_, rec.data(), rec.size(), &p, out.data(), pairs.data(), &n_out, nullptr), "so_kfstore_search");
    std::vector<OverlapCandidate> res((size_t)n_out);
    for (int c = 0; c < n_out; c++) {
        OverlapCandidate& o = res[(size_t)c];
        o.mnClientId = out[(size_t)c].agent_id;
        o.mnId = (unsigned long)out[(size_t)c].keyframe_id;
        o.slot = out[(size_t)c].slot;
        o.votes = out[(size_t)c].votes;
        o.nmatches = out[(size_t)c].n_matches;
        o.vpMatches12.assign(pairs.begin() + (size_t)c * (size_t)kf.N, pairs.begin() + (size_t)(c + 1) * (size_t)kf.N);
    }
    return res;
}

}  // namespace ORB_SLAM2
