// COMPILE-ONLY stand-in used by tests/test_cpp_adapters_gpu.py to type-check the SWARMORB_WITH_OPENCV overload of
// ORB_SLAM2::ORBextractor::operator() in an image without OpenCV.  This is NOT OpenCV and is never linked into
// anything: it declares the handful of members that overload touches (cv::Mat::data/cols/rows/step/type,
// InputArray::getMat/empty, OutputArray::create/getMat/release, cv::KeyPoint's 28-byte layout, CV_8U/CV_8UC1),
// with the signatures OpenCV 3.4 / 4.x publish.
#pragma once
#include <cstddef>
#include <cstdint>
#define CV_8U 0
#define CV_8UC1 0
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct Mat {
    unsigned char* data = nullptr;
    int cols = 0, rows = 0;
    size_t step = 0;
    int type() const { return CV_8UC1; }
    bool empty() const { return data == nullptr; }
};
struct _InputArray {
    Mat m;
    Mat getMat() const { return m; }
    bool empty() const { return m.empty(); }
};
struct _OutputArray : _InputArray {
    void create(int, int, int) const {}
    void release() const {}
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
