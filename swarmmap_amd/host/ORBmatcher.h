// ORBmatcher.h — drop-in for the tracking-thread part of ORB_SLAM2::ORBmatcher (code/include/ORBmatcher.h:41-83).
// The reference methods take Frame / MapPoint objects; without that object graph in this repository the adapter
// takes flattened views of exactly the fields those methods read (see INTEGRATION.md for the three-line glue
// that fills them from a real Frame).  Constants, constructor arguments and return values are the reference's.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/swarmorb.h"

namespace ORB_SLAM2 {

struct MapPointViews {          // one entry per MapPoint* of vpMapPoints
    std::vector<uint8_t> in_view;   // pMP->mbTrackInView && !pMP->isBad()
    std::vector<float> proj_x, proj_y, view_cos;  // mTrackProjX, mTrackProjY, mTrackViewCos
    std::vector<int32_t> pred_level;               // mnTrackScaleLevel
    std::vector<uint8_t> desc;                     // GetDescriptor(), 32 B each
    std::vector<uint8_t> has_obs;                  // Observations() > 0
};

struct LastFrameViews {         // one entry per keypoint i of LastFrame
    std::vector<uint8_t> valid;     // mvpMapPoints[i] && !mvbOutlier[i] && invzc >= 0 && (u,v) inside the bounds
    std::vector<float> u, v, angle; // projection with CurrentFrame.mTcw; mvKeysUn[i].angle
    std::vector<int32_t> octave;    // mvKeys[i].octave
    std::vector<uint8_t> desc, has_obs;
};

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // code/src/ORBmatcher.cc:37-39

    // The reference constructs an ORBmatcher on the stack at every call site (Tracking.cc:470,623,715,998,1153 ...):
    // construction must be free.  All ORBmatcher objects of a thread share one device context (pinned staging, HBM
    // buffers, stream), created on the thread's first use of a device and kept until ReleaseThreadContext().
    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0);
    ~ORBmatcher();
    static void ReleaseThreadContext();
    // The next search of this thread looks at the same Frame as its previous one (TrackWithMotionModel's
    // SearchByProjection followed by SearchLocalPoints' on mCurrentFrame): the candidate upload is skipped.
    void SameFrameAsPreviousSearch();
    ORBmatcher(const ORBmatcher&) = delete;
    ORBmatcher& operator=(const ORBmatcher&) = delete;

    // Computes the Hamming distance between two ORB descriptors (32 bytes each) — host, like the reference
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b);

    // SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th)
    // kp_to_mp[k] = index into vpMapPoints to store in F.mvpMapPoints[k], or -1.  Returns nmatches.
    int SearchByProjection(const so_frame_view& F, const MapPointViews& vpMapPoints, float th,
                           std::vector<int32_t>& kp_to_mp);
    // SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
    int SearchByProjection(const so_frame_view& CurrentFrame, const LastFrameViews& LastFrame, float th,
                           std::vector<int32_t>& kp_to_last);
    // SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize)
    int SearchForInitialization(const so_frame_view& F1, const so_frame_view& F2, std::vector<float>& vbPrevMatched,
                                std::vector<int32_t>& vnMatches12, int windowSize = 10);

    // SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches) [variant 0] / (KeyFrame*, KeyFrame*, vpMatches12) [1]
    // descN/angleN/validN: per feature of each side; match_of_2 = vpMapPointMatches as indices into set 1,
    // match_of_1 = vpMatches12 as indices into set 2.
    int SearchByBoW(int variant, int n1, const uint8_t* desc1, const float* angle1, const uint8_t* valid1,
                    const so_featvec& vFeatVec1, int n2, const uint8_t* desc2, const float* angle2,
                    const uint8_t* valid2, const so_featvec& vFeatVec2, std::vector<int32_t>& match_of_2,
                    std::vector<int32_t>& match_of_1);
    // Fuse / SearchBySim3 core: best keypoint of pKF inside the projection window of every map point
    void SearchWindowBest(const so_frame_view& KF, int nq, const uint8_t* valid, const float* u, const float* v,
                          const float* radius, const int32_t* pred_level, const uint8_t* desc, bool chi2_gate,
                          const float* inv_sigma2, std::vector<int32_t>& best_idx, std::vector<int32_t>& best_dist);

protected:
    float mfNNratio;
    bool mbCheckOrientation;
    so_matcher* handle_ = nullptr;
};

}  // namespace ORB_SLAM2
