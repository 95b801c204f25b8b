// frame_device.h — device side of the Frame post-processing between extractor and matcher:
// UndistortKeyPoints + ComputeImageBounds + AssignFeaturesToGrid in one launch, isInFrustum in another.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace so {

struct FrameCam {
    float fx, fy, cx, cy, k1, k2, p1, p2, k3;
};

constexpr int kFrameGridCols = 64, kFrameGridRows = 48;  // FRAME_GRID_COLS / ROWS, code/include/Frame.h:37-38
constexpr int kFrameMaxKeypoints = 16384;                // sort capacity of the grid kernel (LDS keys)

struct FramePrepareArgs {
    FrameCam cam;
    int width, height, n;
    int do_bounds;     // 1: ComputeImageBounds into bounds[4]; 0: bounds[] is an input; 2: bounds_value is (and is copied to bounds[])
    int do_undistort;  // 1: xy -> xy_un; 0: xy_un is an input
    int do_grid;       // 1: cell_of / cell_start / cell_items / n_inside
    const float* xy;   // 2n
    float* xy_un;      // 2n
    float* bounds;     // 4: mnMinX, mnMaxX, mnMinY, mnMaxY
    float bounds_value[4];  // do_bounds == 2
    int32_t* cell_of;     // n
    int32_t* cell_start;  // 64*48+1
    int32_t* cell_items;  // n
    int32_t* n_inside;
    // ---- device-resident frame (dframe.cpp): keypoints come straight from the extractor's HBM outputs and the
    //      matcher's candidate layout is produced here; all null / zero on the host-array path ----
    const void* ex_meta;       // SelectedKp[n]: (x, y, level, score) in level coordinates; xy = (x, y) * scale[level]
    const int32_t* ex_total;   // number of keypoints (device word); n above is then the capacity
    const uint8_t* ex_desc;    // n x 32
    float scale[8];            // mvScaleFactor
    int8_t* octave;            // by keypoint index
    uint4* desc_by_index;      // the frame's own copy of the descriptors (the extractor's buffer is reused by the next frame)
    const float* ex_angle;     // n: the extractor's keypoint angles (may be null)
    float* angle_by_index;     // the frame's own copy (the device-side resolve of TrackWithMotionModel's rotation check reads it)
    float* xy_un_host;         // host-mapped mirror of xy_un (may be null)
    float2* s_xy;              // candidates in grid-traversal order (cell x, cell y, index): position = tie-break rank
    int8_t* s_octave;
    uint4* s_desc;             // 2 per candidate
    int32_t* perm_host;        // host-mapped mirror of cell_items (position -> keypoint index)
    int32_t* col_start;        // 65: first position of every grid column
    int32_t* header_host;      // host-mapped {n, n_inside, 0, 0} + bounds[4] as float bits
};
void launch_frame_prepare(const FramePrepareArgs& a, hipStream_t s);
struct ExtractBatchMember;
// d_args[n * rot] in device memory: member b's launch is row (b * rot + d_members[b].prep_sel), skipped when d_members[b].skip
void launch_frame_prepare_batch(const FramePrepareArgs* d_args, int rot, const ExtractBatchMember* d_members, int n, hipStream_t s);

struct FrameFrustumArgs {
    FrameCam cam;
    float bounds[4];
    float Tcw[12];
    int n;
    const float* Xw;      // 3n
    const float* normal;  // 3n
    const float* max_dist;
    const float* min_dist;
    float viewing_cos_limit, log_scale_factor;
    int n_scale_levels;
    uint8_t* in_view;
    float* proj_x;
    float* proj_y;
    float* view_cos;
    int32_t* pred_level;
};
void launch_frame_frustum(const FrameFrustumArgs& a, hipStream_t s);
// so_map_write_positions: Xw[slots[i]] = X[i] (slots / X may live in pinned host memory)
void launch_map_scatter_positions(float* d_Xw, const int32_t* slots, const float* X, int n, hipStream_t s);
void launch_map_scatter_rows(float* d_Xw, float* d_normal, float* d_max, float* d_min, const int32_t* slots, const float* X,
                             const float* N, const float* mx, const float* mn, int n, hipStream_t s);
// so_map_write: rows first .. first + n - 1 <- the given arrays (pinned host memory, 4-byte aligned; null = column untouched)
void launch_map_write_range(float* d_Xw, float* d_normal, float* d_max, float* d_min, uint8_t* d_desc, const float* X, const float* N,
                            const float* mx, const float* mn, const uint8_t* D, int first, int n, hipStream_t s);

}  // namespace so
