// Frame.h — drop-in for the per-frame post-processing members of ORB_SLAM2::Frame (code/include/Frame.h,
// code/src/Frame.cc:277-292, 316-375, 454-514) and for MapPoint::ComputeDistinctiveDescriptors
// (code/src/MapPoint.cc:323-392) on libswarmorb.so.  Same names and argument meaning; the object-graph parts
// (mvKeysUn, mGrid, MapPoint track fields) are plain arrays the caller copies into its own members.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/swarmorb.h"
#include "swarmorb_types.h"

namespace ORB_SLAM2 {

// One object per camera and tracking thread, kept for the whole sequence (it owns the device context and the state
// the reference keeps in Frame's static members: calibration, image bounds, mbInitialComputations) - not one per image.
class Frame {
public:
    // K = (fx, fy, cx, cy), distCoef = (k1, k2, p1, p2[, k3]) as read from the settings file (Tracking.cc:60-84)
    Frame(const float K[4], const std::vector<float>& distCoef, int device = 0);
    ~Frame();
    Frame(const Frame&) = delete;
    Frame& operator=(const Frame&) = delete;

    // What the Frame constructor does after ExtractORB (Frame.cc:183-192): UndistortKeyPoints(),
    // ComputeImageBounds(imGray) on the first frame (mbInitialComputations), AssignFeaturesToGrid().
    // mvKeysUn comes back as a copy of mvKeys with undistorted pt; mGrid as CSR lists (cell = x * 48 + y).
    void UndistortAndAssign(const std::vector<swarmorb::KeyPoint>& mvKeys, int cols, int rows,
                            std::vector<swarmorb::KeyPoint>& mvKeysUn);
    const std::vector<int32_t>& GridStart() const { return cell_start_; }   // FRAME_GRID_COLS*ROWS + 1
    const std::vector<int32_t>& GridItems() const { return cell_items_; }   // keypoint indices, cell by cell
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;                    // static members in the reference

    // bool isInFrustum(MapPoint* pMP, float viewingCosLimit) for every candidate of Tracking::SearchLocalPoints:
    // returns mbTrackInView per point and fills the track fields where it is true.
    struct TrackFields {
        std::vector<uint8_t> mbTrackInView;
        std::vector<float> mTrackProjX, mTrackProjY, mTrackViewCos;
        std::vector<int32_t> mnTrackScaleLevel;
    };
    void isInFrustum(const float Tcw[12], const std::vector<float>& worldPos, const std::vector<float>& normal,
                     const std::vector<float>& mfMaxDistance, const std::vector<float>& mfMinDistance,
                     float viewingCosLimit, float mfLogScaleFactor, int mnScaleLevels, TrackFields& out);

private:
    so_camera cam_{};
    so_frame_ctx* handle_ = nullptr;
    bool mbInitialComputations = true;
    std::vector<int32_t> cell_of_, cell_start_, cell_items_;
};

// MapPoint::ComputeDistinctiveDescriptors for a batch of map points: descriptors of point p =
// rows [offsets[p], offsets[p+1]) ; returns, per point, the row (within its own list) to clone into mDescriptor.
class DistinctiveDescriptors {
public:
    explicit DistinctiveDescriptors(int device = 0);
    ~DistinctiveDescriptors();
    std::vector<int32_t> Compute(const std::vector<int32_t>& offsets, const std::vector<uint8_t>& descriptors);

private:
    so_matcher* handle_ = nullptr;
};

}  // namespace ORB_SLAM2
