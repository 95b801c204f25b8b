// COMPILE-ONLY stand-in for the OpenCV declarations the reference's headers and the swarmorb glue mention.  NOT OpenCV:
// declarations without definitions, usable with -fsyntax-only and nothing else (tests/test_glue_typecheck.py).
#pragma once
#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <list>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32F 5
#define CV_64F 6
#define CV_32FC1 5
namespace cv {
template <typename T> struct Point_ { T x, y; Point_(); Point_(T, T); };
typedef Point_<float> Point2f;
typedef Point_<int> Point;
typedef Point_<int> Point2i;
typedef Point_<double> Point2d;
template <typename T> struct Point3_ { T x, y, z; Point3_(); Point3_(T, T, T); };
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
template <typename T> struct Size_ { T width, height; Size_(); Size_(T, T); };
typedef Size_<int> Size;
template <typename T> struct Rect_ { T x, y, width, height; Rect_(); Rect_(T, T, T, T); };
typedef Rect_<int> Rect;
template <typename T, int N> struct Vec { T val[N]; T& operator[](int); const T& operator[](int) const; };
typedef Vec<float, 3> Vec3f;
struct Scalar { double val[4]; Scalar(); Scalar(double, double = 0, double = 0, double = 0); };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; KeyPoint(); KeyPoint(float, float, float, float = -1, float = 0, int = 0, int = -1); };
struct Range { int start, end; Range(); Range(int, int); static Range all(); };
struct MatExpr;
struct Mat {
    unsigned char* data;
    int cols, rows, flags, dims;
    struct Step { size_t operator[](int) const; operator size_t() const; } step;
    Mat(); Mat(int, int, int); Mat(int, int, int, void*, size_t = 0); Mat(int, int, int, const Scalar&); Mat(Size, int);
    Mat(const Mat&); Mat(const MatExpr&);
    template <typename T> explicit Mat(const std::vector<T>&, bool = false);
    template <typename T> Mat(const Point3_<T>&, bool = true);
    Mat& operator=(const Mat&); Mat& operator=(const MatExpr&); Mat& operator=(const Scalar&);
    ~Mat();
    int type() const; int depth() const; int channels() const; bool empty() const; size_t total() const; size_t elemSize() const;
    bool isContinuous() const;
    Size size() const;
    Mat clone() const; void copyTo(Mat&) const; void copyTo(Mat&, const Mat&) const; void convertTo(Mat&, int, double = 1, double = 0) const;
    void create(int, int, int); void release();
    Mat row(int) const; Mat col(int) const; Mat rowRange(int, int) const; Mat colRange(int, int) const;
    Mat operator()(const Rect&) const; Mat operator()(Range, Range) const;
    Mat reshape(int, int = 0) const;
    MatExpr t() const; MatExpr inv(int = 0) const; MatExpr mul(const Mat&, double = 1) const;
    double dot(const Mat&) const; Mat cross(const Mat&) const;
    template <typename T> T& at(int); template <typename T> const T& at(int) const;
    template <typename T> T& at(int, int); template <typename T> const T& at(int, int) const;
    template <typename T> T* ptr(int = 0); template <typename T> const T* ptr(int = 0) const;
    unsigned char* ptr(int = 0); const unsigned char* ptr(int = 0) const;
    static MatExpr zeros(int, int, int); static MatExpr eye(int, int, int); static MatExpr ones(int, int, int);
    static MatExpr zeros(Size, int);
    void push_back(const Mat&);
    Mat& setTo(const Scalar&);
};
struct MatExpr { operator Mat() const; MatExpr t() const; MatExpr inv(int = 0) const; Mat row(int) const; Mat col(int) const;
                 Mat rowRange(int, int) const; Mat colRange(int, int) const; template <typename T> T& at(int, int = 0); };
template <typename T> struct Mat_ : Mat { Mat_(); Mat_(int, int); Mat_(const Mat&); T& operator()(int, int); T& operator()(int);
                                          Mat_& operator<<(T); Mat_& operator,(T); };
MatExpr operator*(const Mat&, const Mat&); MatExpr operator*(const MatExpr&, const Mat&); MatExpr operator*(const Mat&, const MatExpr&);
MatExpr operator*(const MatExpr&, const MatExpr&);
MatExpr operator*(double, const Mat&); MatExpr operator*(const Mat&, double); MatExpr operator*(double, const MatExpr&); MatExpr operator*(const MatExpr&, double);
MatExpr operator+(const Mat&, const Mat&); MatExpr operator+(const MatExpr&, const Mat&); MatExpr operator+(const Mat&, const MatExpr&);
MatExpr operator+(const MatExpr&, const MatExpr&);
MatExpr operator-(const Mat&, const Mat&); MatExpr operator-(const MatExpr&, const Mat&); MatExpr operator-(const Mat&, const MatExpr&);
MatExpr operator-(const MatExpr&, const MatExpr&);
MatExpr operator-(const Mat&); MatExpr operator-(const MatExpr&);
MatExpr operator/(const Mat&, double); MatExpr operator/(const MatExpr&, double);
double norm(const Mat&, int = 4); double norm(const MatExpr&, int = 4); double determinant(const Mat&);
struct _InputArray { _InputArray(); _InputArray(const Mat&); _InputArray(const MatExpr&); template <typename T> _InputArray(const std::vector<T>&);
                     Mat getMat(int = -1) const; bool empty() const; int type() const; };
struct _OutputArray : _InputArray { _OutputArray(); _OutputArray(Mat&); template <typename T> _OutputArray(std::vector<T>&);
                                    void create(int, int, int) const; void release() const; };
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _OutputArray& InputOutputArray;
InputArray noArray();
template <typename T> struct Ptr { T* operator->() const; T& operator*() const; Ptr(); Ptr(T*); operator bool() const; T* get() const; };
struct FileNodeIterator;
struct FileNode { enum { SEQ = 5, MAP = 6 }; operator float() const; operator int() const; operator double() const; operator std::string() const; bool empty() const;
                  FileNode operator[](const std::string&) const; FileNode operator[](const char*) const; FileNode operator[](int) const; int type() const; size_t size() const;
                  FileNodeIterator begin() const; FileNodeIterator end() const; };
struct FileNodeIterator { FileNode operator*() const; FileNodeIterator& operator++(); bool operator!=(const FileNodeIterator&) const; };
struct FileStorage { enum { READ = 0, WRITE = 1 }; FileStorage(); FileStorage(const std::string&, int); bool isOpened() const; FileNode operator[](const std::string&) const;
                     FileNode operator[](const char*) const; void release(); bool open(const std::string&, int); };
template <class T> FileStorage& operator<<(FileStorage&, const T&);
template <class T> void operator>>(const FileNode&, T&);
struct SVD { enum { MODIFY_A = 1, FULL_UV = 4 }; static void compute(InputArray, OutputArray, OutputArray, OutputArray, int = 0); Mat u, w, vt; SVD(); SVD(InputArray, int = 0); };
void undistortPoints(InputArray, OutputArray, InputArray, InputArray, InputArray = noArray(), InputArray = noArray());
void hconcat(InputArray, InputArray, OutputArray); void vconcat(InputArray, InputArray, OutputArray);
struct RNG { RNG(); RNG(uint64_t); int uniform(int, int); float uniform(float, float); double gaussian(double); };
namespace cuda {
struct Stream { Stream(); void waitForCompletion(); static Stream& Null(); };
struct GpuMat { GpuMat(); GpuMat(int, int, int); GpuMat(const GpuMat&); int rows, cols; size_t step; unsigned char* data; void upload(InputArray); void upload(InputArray, Stream&);
                void download(OutputArray) const; void download(OutputArray, Stream&) const; GpuMat operator()(Rect) const; GpuMat rowRange(int, int) const; GpuMat colRange(int, int) const;
                bool empty() const; Size size() const; int type() const; void create(int, int, int); void release(); template <typename T> T* ptr(int = 0); };
struct Filter { virtual void apply(InputArray, OutputArray, Stream& = Stream::Null()) = 0; virtual ~Filter(); };
template <typename T> struct PtrStepSz { T* data; size_t step; int cols, rows; PtrStepSz(); PtrStepSz(const GpuMat&); };
typedef PtrStepSz<unsigned char> PtrStepSzb;
typedef PtrStepSz<int> PtrStepSzi;
template <typename T> struct PtrStep { T* data; size_t step; PtrStep(); PtrStep(const GpuMat&); };
typedef PtrStep<unsigned char> PtrStepb;
typedef PtrStep<int> PtrStepi;
}  // namespace cuda
}  // namespace cv
