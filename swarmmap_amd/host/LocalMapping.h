// LocalMapping.h — the two data-parallel loops of ORB_SLAM2::LocalMapping / MapPoint that sit between the matcher and
// local BA, over flattened views (the object graph is not in this repository): the per-match body of
// LocalMapping::CreateNewMapPoints (code/src/LocalMapping.cc:263-420) and MapPoint::UpdateNormalAndDepth
// (code/src/MapPoint.cc:413-465) for a batch of points.  Thin wrappers over so_triangulate_matches /
// so_update_normal_and_depth; INTEGRATION.md 3b' shows where they go inside LocalMapping.cc.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/swarmorb.h"

namespace ORB_SLAM2 {

struct TriangulationKeyFrame {      // what CreateNewMapPoints reads of a KeyFrame
    float Tcw[12];                  // [GetRotation() | GetTranslation()]
    float fx, fy, cx, cy;           // (invfx = 1.0f / fx as Frame.cc:266 sets it)
    std::vector<float> mvScaleFactors, mvLevelSigma2;
};

struct TriangulationMatches {       // vMatchedIndices of all neighbours, flattened
    std::vector<int32_t> neighbour; // index into the neighbour list
    std::vector<float> xy1, xy2;    // mvKeysUn[idx].pt of the current keyframe / of the neighbour, 2 floats each
    std::vector<int32_t> octave1, octave2;
    int size() const { return (int)neighbour.size(); }
};

class LocalMappingOps {
public:
    explicit LocalMappingOps(int device = 0);
    ~LocalMappingOps();
    LocalMappingOps(const LocalMappingOps&) = delete;
    LocalMappingOps& operator=(const LocalMappingOps&) = delete;
    // CreateNewMapPoints: ok[k] = 1 where the reference would create a MapPoint from match k, x3D its position.
    // ratioFactor = 1.5f * mpCurrentKeyFrame->mfScaleFactor.  Returns the number of new points.
    int TriangulateMatches(const TriangulationKeyFrame& current, const std::vector<TriangulationKeyFrame>& neighbours,
                           float ratioFactor, const TriangulationMatches& matches, std::vector<uint8_t>& ok, std::vector<float>& x3D);
    // TriangulateMatches + the new points' mNormalVector / mfMaxDistance / mfMinDistance (what CreateNewMapPoints gets from
    // pMP->UpdateNormalAndDepth() right after the two AddObservation calls, LocalMapping.cc:403-412) in the same launch:
    // normal (3 per match), maxDistance, minDistance are set where ok[k] = 1.  Returns the number of new points.
    int CreateNewPoints(const TriangulationKeyFrame& current, const std::vector<TriangulationKeyFrame>& neighbours, float ratioFactor,
                        const TriangulationMatches& matches, std::vector<uint8_t>& ok, std::vector<float>& x3D,
                        std::vector<float>& normal, std::vector<float>& maxDistance, std::vector<float>& minDistance);
    // MapPoint::UpdateNormalAndDepth for points p = 0..n-1: observers' camera centres obsOw[offsets[p] .. offsets[p + 1]),
    // reference keyframe's centre / level scale / last level scale per point; normal, maxDistance, minDistance in / out.
    void UpdateNormalAndDepth(const std::vector<int32_t>& offsets, const std::vector<float>& obsOw, const std::vector<float>& Xw,
                              const std::vector<float>& refOw, const std::vector<float>& refLevelScale,
                              const std::vector<float>& refLastScale, std::vector<float>& normal, std::vector<float>& maxDistance,
                              std::vector<float>& minDistance);

private:
    so_matcher* handle_ = nullptr;
};

}  // namespace ORB_SLAM2
