// swarmorb_types.h — the few OpenCV types the reference's hot-path signatures mention, for builds without OpenCV.
// With -DSWARMORB_WITH_OPENCV the adapters use cv::KeyPoint / cv::Mat directly (same memory layout).
#pragma once
#include <cstdint>
#include <vector>

#ifdef SWARMORB_WITH_OPENCV
#include <opencv2/core/core.hpp>
namespace swarmorb {
using KeyPoint = cv::KeyPoint;
}
#else
namespace swarmorb {
// cv::KeyPoint: {Point2f pt; float size, angle, response; int octave, class_id} = 28 bytes
struct Point2f {
    float x, y;
};
struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "must match cv::KeyPoint / so_keypoint");
}  // namespace swarmorb
#endif

namespace swarmorb {
// CV_8UC1 image view (cv::Mat::data / cols / rows / step)
struct ImageView {
    const uint8_t* data = nullptr;
    int cols = 0, rows = 0, step = 0;
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};
// N x 32 CV_8U descriptor matrix
struct Descriptors {
    std::vector<uint8_t> data;
    int rows = 0;
    void create(int n) {
        rows = n;
        data.resize((size_t)n * 32);
    }
    void release() {
        rows = 0;
        data.clear();
    }
    uint8_t* ptr(int r) { return data.data() + (size_t)r * 32; }
    const uint8_t* ptr(int r) const { return data.data() + (size_t)r * 32; }
};
}  // namespace swarmorb
