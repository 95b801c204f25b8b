// AgentMediator.h — drop-in for the cross-agent candidate search of ORB_SLAM2::AgentMediator on libswarmorb.so:
// CheckOverlapCandidates (code/include/AgentMediator.h, code/src/AgentMediator.cc:140-202: every new keyframe of a peer is
// looked up in every other agent's keyframe database) and the matching loop at the head of GetSim3 (:204-262:
// ORBmatcher(0.75, true).SearchByBoW(pCurrentKF, pKF, vvpMapPointMatches[i]), `nmatches < 20` discards the candidate).
// The reference walks KeyFrame* / KeyFrameDatabase* / ORBVocabulary; the adapter takes flattened keyframes (what a
// keyframe record carries) and returns, per surviving candidate, the (agent, keyframe) it names and vpMatches12 as
// keypoint indices - what the Sim3Solver that follows (:264-266) is constructed from.  Sim3 RANSAC, map merging and
// everything after stay in the reference.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/swarmorb.h"

namespace ORB_SLAM2 {

struct KeyFrameView {              // the fields of a KeyFrame the search reads
    int mnClientId = 0;            // origin agent (GetOriginMapId(), AgentMediator.cc:186)
    unsigned long mnId = 0;
    double mTimeStamp = 0.0;
    const float* Tcw = nullptr;    // 12 floats [R|t] (may be null: zeros)
    const float* K = nullptr;      // fx fy cx cy (may be null)
    int N = 0;
    const float* xy = nullptr;         // mvKeysUn[i].pt, N x 2
    const float* angle = nullptr;      // mvKeysUn[i].angle
    const int32_t* octave = nullptr;   // mvKeysUn[i].octave
    const uint8_t* descriptors = nullptr;  // mDescriptors, N x 32
    const int32_t* mapPointId = nullptr;   // mvpMapPoints[i] ? (isBad() ? -1 : mnId) : -1
};

struct OverlapCandidate {
    int mnClientId = 0;
    unsigned long mnId = 0;
    int slot = -1;          // where the keyframe sits in the store (KeyFrameStore::read)
    int votes = 0;          // detection score (phase 1)
    int nmatches = 0;       // SearchByBoW's return value
    std::vector<int> vpMatches12;  // per keypoint of the query: matched keypoint of the candidate or -1
};

class AgentMediator {
public:
    // capacity: keyframes the store keeps (a ring: the oldest are overwritten); maxKeypoints: nFeatures + 3 * nLevels
    AgentMediator(int capacityKeyFrames, int maxKeypoints, int device = 0);
    ~AgentMediator();
    AgentMediator(const AgentMediator&) = delete;
    AgentMediator& operator=(const AgentMediator&) = delete;

    // what `mpMap->AddKeyFrame(kf)` + the peer's `mpKeyFrameDatabase->add(kf)` amount to for the search (:170-176, :200)
    int AddKeyFrame(const KeyFrameView& kf);
    int KeyFramesInStore() const;
    // CheckOverlapCandidates' per-keyframe body + GetSim3's matching loop: candidates of OTHER agents with at least
    // minMatches SearchByBoW pairs, best detection score first
    std::vector<OverlapCandidate> CheckOverlapCandidates(const KeyFrameView& kf, float nnratio = 0.75f, bool checkOri = true,
                                                         int minVotes = 20, int minMatches = 20, int maxCandidates = 16);

private:
    std::vector<uint8_t> pack(const KeyFrameView& kf) const;
    so_kfstore* handle_ = nullptr;
    int max_keypoints_ = 0;
};

}  // namespace ORB_SLAM2
