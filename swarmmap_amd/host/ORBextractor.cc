// ORBextractor.cc — see ORBextractor.h.  The reference aborts the process on CUDA errors (checkCudaErrors);
// this adapter throws std::runtime_error with the library's status text instead.
#include "ORBextractor.h"

#include <cassert>
#include <cstring>

namespace ORB_SLAM2 {

static void check(int status, const char* what) {
    if (status != SO_OK)
        throw std::runtime_error(std::string(what) + ": " + so_status_string(status) + " (" + so_last_error() + ")");
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST, int device)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST) {
    so_extractor_config cfg{_nfeatures, _scaleFactor, _nlevels, _iniThFAST, _minThFAST, device};
    check(so_extractor_create(&cfg, &handle_), "so_extractor_create");
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    check(so_extractor_tables(handle_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                              mvInvLevelSigma2.data(), mnFeaturesPerLevel.data()),
          "so_extractor_tables");
    const int cap = so_extractor_capacity(handle_);
    kp_buf_.resize((size_t)cap);
    desc_buf_.resize((size_t)cap * 32);
}

ORBextractor::~ORBextractor() { so_extractor_destroy(handle_); }

void ORBextractor::operator()(const swarmorb::ImageView& image, const swarmorb::ImageView&,
                              std::vector<swarmorb::KeyPoint>& _keypoints, swarmorb::Descriptors& _descriptors) {
    if (image.empty()) return;  // ORBextractor.cc:750-751
    int n = 0;
    check(so_extractor_run(handle_, image.data, image.cols, image.rows, image.step, kp_buf_.data(), desc_buf_.data(),
                           (int)kp_buf_.size(), &n),
          "so_extractor_run");
    if (n == 0) {
        _descriptors.release();  // ORBextractor.cc:769-770
    } else {
        _descriptors.create(n);
        std::memcpy(_descriptors.data.data(), desc_buf_.data(), (size_t)n * 32);
    }
    _keypoints.clear();
    _keypoints.resize((size_t)n);
    static_assert(sizeof(swarmorb::KeyPoint) == sizeof(so_keypoint), "layout");
    if (n) std::memcpy(static_cast<void*>(_keypoints.data()), kp_buf_.data(), (size_t)n * sizeof(so_keypoint));
}

#ifdef SWARMORB_WITH_OPENCV
void ORBextractor::operator()(cv::InputArray _image, cv::InputArray, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    int n = 0;
    check(so_extractor_run(handle_, image.data, image.cols, image.rows, (int)image.step, kp_buf_.data(),
                           desc_buf_.data(), (int)kp_buf_.size(), &n),
          "so_extractor_run");
    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        std::memcpy(_descriptors.getMat().data, desc_buf_.data(), (size_t)n * 32);
    }
    _keypoints.resize((size_t)n);
    if (n) std::memcpy(static_cast<void*>(_keypoints.data()), kp_buf_.data(), (size_t)n * sizeof(so_keypoint));
}
#endif

}  // namespace ORB_SLAM2
