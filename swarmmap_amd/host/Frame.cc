// Frame.cc — see Frame.h.
#include "Frame.h"

#include <stdexcept>
#include <string>

namespace ORB_SLAM2 {

static void check(int status, const char* what) {
    if (status != SO_OK)
        throw std::runtime_error(std::string(what) + ": " + so_status_string(status) + " (" + so_last_error() + ")");
}

Frame::Frame(const float K[4], const std::vector<float>& d, int device) {
    cam_.fx = K[0]; cam_.fy = K[1]; cam_.cx = K[2]; cam_.cy = K[3];
    cam_.k1 = d.size() > 0 ? d[0] : 0.f; cam_.k2 = d.size() > 1 ? d[1] : 0.f;
    cam_.p1 = d.size() > 2 ? d[2] : 0.f; cam_.p2 = d.size() > 3 ? d[3] : 0.f;
    cam_.k3 = d.size() > 4 ? d[4] : 0.f;
    check(so_frame_create(device, &handle_), "so_frame_create");
}
Frame::~Frame() { so_frame_destroy(handle_); }

void Frame::UndistortAndAssign(const std::vector<swarmorb::KeyPoint>& keys, int cols, int rows,
                               std::vector<swarmorb::KeyPoint>& keysUn) {
    const int n = (int)keys.size();
    std::vector<float> xy(2 * (size_t)n), un(2 * (size_t)n);
    for (int i = 0; i < n; i++) { xy[2 * i] = keys[i].pt.x; xy[2 * i + 1] = keys[i].pt.y; }
    float b[4] = {mnMinX, mnMaxX, mnMinY, mnMaxY};
    cell_of_.assign((size_t)n, -1);
    cell_start_.assign(64 * 48 + 1, 0);
    cell_items_.assign((size_t)(n > 0 ? n : 1), 0);
    int32_t inside = 0;
    check(so_frame_prepare(handle_, &cam_, cols, rows, mbInitialComputations ? 1 : 0, n, xy.data(), un.data(), b,
                           cell_of_.data(), cell_start_.data(), cell_items_.data(), &inside), "so_frame_prepare");
    mbInitialComputations = false;
    mnMinX = b[0]; mnMaxX = b[1]; mnMinY = b[2]; mnMaxY = b[3];
    cell_items_.resize((size_t)inside);
    keysUn = keys;  // cv::KeyPoint kp = mvKeys[i]; kp.pt = undistorted
    for (int i = 0; i < n; i++) { keysUn[i].pt.x = un[2 * i]; keysUn[i].pt.y = un[2 * i + 1]; }
}

void Frame::isInFrustum(const float Tcw[12], const std::vector<float>& P, const std::vector<float>& N,
                        const std::vector<float>& maxD, const std::vector<float>& minD, float viewingCosLimit,
                        float logScaleFactor, int nLevels, TrackFields& o) {
    const int n = (int)maxD.size();
    o.mbTrackInView.assign((size_t)n, 0);
    o.mTrackProjX.resize((size_t)n); o.mTrackProjY.resize((size_t)n); o.mTrackViewCos.resize((size_t)n);
    o.mnTrackScaleLevel.resize((size_t)n);
    const float b[4] = {mnMinX, mnMaxX, mnMinY, mnMaxY};
    check(so_frame_is_in_frustum(handle_, &cam_, b, Tcw, n, P.data(), N.data(), maxD.data(), minD.data(), viewingCosLimit,
                                 logScaleFactor, nLevels, o.mbTrackInView.data(), o.mTrackProjX.data(),
                                 o.mTrackProjY.data(), o.mTrackViewCos.data(), o.mnTrackScaleLevel.data()),
          "so_frame_is_in_frustum");
}

DistinctiveDescriptors::DistinctiveDescriptors(int device) { check(so_matcher_create(device, &handle_), "so_matcher_create"); }
DistinctiveDescriptors::~DistinctiveDescriptors() { so_matcher_destroy(handle_); }

std::vector<int32_t> DistinctiveDescriptors::Compute(const std::vector<int32_t>& off, const std::vector<uint8_t>& desc) {
    const int n = (int)off.size() - 1;
    std::vector<int32_t> best((size_t)(n > 0 ? n : 0));
    if (n > 0) check(so_distinctive_descriptors(handle_, n, off.data(), desc.data(), best.data(), nullptr),
                     "so_distinctive_descriptors");
    return best;
}

}  // namespace ORB_SLAM2
