// ORBextractor.h — drop-in for ORB_SLAM2::ORBextractor (code/include/ORBextractor.h:49-127) on libswarmorb.so.
// Same constructor arguments, call operator and getters; the CUDA/OpenCV-CUDA members are replaced by one
// so_extractor handle (per instance, like the reference's per-instance streams and device buffers).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/swarmorb.h"
#include "swarmorb_types.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0);
    ~ORBextractor();
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image.  Mask is ignored, as in the reference.
    void operator()(const swarmorb::ImageView& image, const swarmorb::ImageView& mask,
                    std::vector<swarmorb::KeyPoint>& keypoints, swarmorb::Descriptors& descriptors);
#ifdef SWARMORB_WITH_OPENCV
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                    cv::OutputArray descriptors);
#endif

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels, iniThFAST, minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    so_extractor* handle_ = nullptr;
    std::vector<so_keypoint> kp_buf_;
    std::vector<uint8_t> desc_buf_;
};

}  // namespace ORB_SLAM2
