// ORBmatcher.h — drop-in for the tracking-thread part of ORB_SLAM2::ORBmatcher (code/include/ORBmatcher.h:41-83).
// The reference methods take Frame / MapPoint objects; without that object graph in this repository the adapter
// takes flattened views of exactly the fields those methods read (see INTEGRATION.md for the three-line glue
// that fills them from a real Frame).  Constants, constructor arguments and return values are the reference's.
#pragma once
#include <cstdint>
#include <utility>
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
    // ---- LocalMapping / loop closing / relocalisation routines.  The projections (Rcw * X + tcw, PredictScale,
    //      IsInImage, the depth / viewing-angle gates) and every object-graph side effect stay with the caller, as in
    //      INTEGRATION.md; what the reference then does per map point - GetFeaturesInArea, DescriptorDistance, best /
    //      greedy selection - is what these run on the GPU. ----
    struct WindowQueries {              // one entry per projected map point
        std::vector<uint8_t> valid;         // passed the caller's gates
        std::vector<float> u, v, radius;    // projection and th * mvScaleFactors[nPredictedLevel]
        std::vector<int32_t> pred_level;    // nPredictedLevel (Fuse / SearchBySim3: octave in [pred - 1, pred])
        std::vector<int32_t> min_level, max_level;  // greedy searches: explicit octave range
        std::vector<uint8_t> desc;          // pMP->GetDescriptor(), 32 B each
        std::vector<float> angle;           // greedy search with orientation check: the map point's keypoint angle
        int size() const { return (int)u.size(); }
    };
    struct KeyFrameFeatures {           // the per-feature fields of a KeyFrame the vocabulary-node routines read
        int n = 0;
        const float *x = nullptr, *y = nullptr, *angle = nullptr;  // mvKeysUn
        const int32_t* octave = nullptr;
        const uint8_t* desc = nullptr;   // mDescriptors
        const uint8_t* free_ = nullptr;  // !GetMapPoint(i)
    };

    // SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, vMatchedPairs, bOnlyStereo = false) —
    // code/src/ORBmatcher.cc:599-749.  (ex, ey) = the epipole in image 2 (:607-613).
    int SearchForTriangulation(const KeyFrameFeatures& kf1, const so_featvec& vFeatVec1, const KeyFrameFeatures& kf2,
                               const so_featvec& vFeatVec2, const float F12[9], float ex, float ey,
                               const std::vector<float>& mvScaleFactors2, const std::vector<float>& mvLevelSigma2_2,
                               std::vector<std::pair<size_t, size_t>>& vMatchedPairs);
    // Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th) — :751-891 (chi2 5.99 reprojection gate) and
    // Fuse(KeyFrame* pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) — :893-1009 (no gate): best keypoint per map
    // point; returns how many have bestDist <= TH_LOW (the ones the caller AddObservation()s / Replace()s).
    int Fuse(const so_frame_view& pKF, const WindowQueries& vpMapPoints, const std::vector<float>& mvInvLevelSigma2,
             std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist);
    int Fuse(const so_frame_view& pKF, const WindowQueries& vpPointsScw, std::vector<int32_t>& bestIdx,
             std::vector<int32_t>& bestDist);
    // SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) — :1011-1221: map points of 1 projected into 2 and of 2
    // into 1 (already-matched ones arrive as valid = 0), best keypoint <= TH_HIGH each way, kept where both agree.
    // vnMatch12[i1] = i2 or -1.  Returns nFound.
    int SearchBySim3(const so_frame_view& pKF1, const so_frame_view& pKF2, const WindowQueries& points1_in_2,
                     const WindowQueries& points2_in_1, std::vector<int32_t>& vnMatch12);
    // SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) — :264-373: sequential, a keypoint taken
    // by an earlier point or bound on entry (pKF.excluded) is skipped, TH_LOW, no orientation check
    int SearchByProjection(const so_frame_view& pKF, const WindowQueries& vpPointsScw, std::vector<int32_t>& kp_to_point);
    // SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) — :1356-1473
    int SearchByProjection(const so_frame_view& CurrentFrame, const WindowQueries& vpKFPoints, int ORBdist,
                           std::vector<int32_t>& kp_to_point);
    // ---- the same five routines with the projection on the device (so_fuse, so_fuse_sim3, so_search_by_sim3,
    //      so_search_by_projection_sim3, so_search_by_projection_keyframe): the caller hands over what the reference's
    //      loops read from each MapPoint and from the keyframe; projection, gates, PredictScale, window search and the
    //      routine's thresholds run behind the C ABI.  Map side effects stay with the caller. ----
    struct MapPointFields {                     // one entry per element of vpMapPoints
        std::vector<float> Xw, normal;          // GetWorldPos() (GetGlobalPos() where the routine says so), GetNormal()
        std::vector<float> max_dist, min_dist;  // mfMaxDistance, mfMinDistance
        std::vector<uint8_t> desc;              // GetDescriptor(), 32 B each
        std::vector<uint8_t> valid;             // pMP && !pMP->isBad() && the routine's own exclusions
        std::vector<float> angle;               // SearchByProjection(Frame, KeyFrame): pKF->mvKeysUn[i].angle
        int size() const { return (int)max_dist.size(); }
        so_mappoint_view view() const {
            so_mappoint_view v{};
            v.n = size(); v.Xw = Xw.data(); v.normal = normal.empty() ? nullptr : normal.data();
            v.max_dist = max_dist.data(); v.min_dist = min_dist.data(); v.desc = desc.data();
            v.valid = valid.empty() ? nullptr : valid.data();
            return v;
        }
    };
    struct Calibration {                        // of the target KeyFrame / Frame
        float fx, fy, cx, cy;
        float mfLogScaleFactor;
        std::vector<float> mvInvLevelSigma2;    // Fuse(pKF, vpMapPoints) only
    };
    struct Pose { float m[12]; };               // [Rcw | tcw], row-major 3 x 4
    struct Sim3 { float m[12]; };               // rows 0-2 of Scw
    // Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th) — :751-891.  bestIdx[i] = keypoint of pKF to fuse map
    // point i with (bestDist <= TH_LOW) or -1; returns their number.
    int Fuse(const so_frame_view& pKF, const Calibration& cal, const Pose& Tcw, const MapPointFields& vpMapPoints, float th,
             std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist);
    // Fuse(KeyFrame* pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) — :893-1009
    int Fuse(const so_frame_view& pKF, const Calibration& cal, const Sim3& Scw, const MapPointFields& vpPoints, float th,
             std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist);
    // SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) — :1011-1221.  vnMatch12[i1] = i2 or -1; returns nFound.
    int SearchBySim3(const so_frame_view& pKF1, const Calibration& cal1, const Pose& T1w, const so_frame_view& pKF2,
                     const Calibration& cal2, const Pose& T2w, const MapPointFields& vpMapPoints1,
                     const MapPointFields& vpMapPoints2, float s12, const float R12[9], const float t12[3], float th,
                     std::vector<int32_t>& vnMatch12);
    // SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, int th) — :264-373 (pKF.excluded = vpMatched[k])
    int SearchByProjection(const so_frame_view& pKF, const Calibration& cal, const Sim3& Scw, const MapPointFields& vpPoints,
                           int th, std::vector<int32_t>& kp_to_point);
    // SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist, bGlobal) — :1356-1473
    int SearchByProjection(const so_frame_view& CurrentFrame, const Calibration& cal, const Pose& Tcw,
                           const MapPointFields& vpKFPoints, float th, int ORBdist, std::vector<int32_t>& kp_to_point);

    // MapPoint::ComputeDistinctiveDescriptors (code/src/MapPoint.cc:323-392) for a batch: point p owns descriptors
    // [offsets[p], offsets[p + 1]); best[p] = index inside its own list
    std::vector<int32_t> ComputeDistinctiveDescriptors(const std::vector<int32_t>& offsets,
                                                       const std::vector<uint8_t>& descriptors);

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
