// dframe_internal.h — the device-resident Frame and map-point table, shared by dframe.cpp (which owns them) and
// matcher.cpp (whose tracking searches read them in place).
//
// HBM layout of one so_dframe (capacity = the extractor's output capacity, everything allocated once):
//   by keypoint index   xy_un float2 | octave i8 | desc 32 B           (mvKeysUn, mvKeys[i].octave, mDescriptors)
//   grid                cell_start i32[64*48+1] | cell_items i32        (mGrid, cell = x * 48 + y)
//   matcher layout      s_xy float2 | s_octave i8 | s_desc 2 x uint4 | col_start i32[65]
//                       = the same keypoints in grid-traversal order (cell x, cell y, index): array position is the
//                       reference's tie-break rank, a GetFeaturesInArea window is one contiguous range of positions
//   host-mapped mirrors xy_un | cell_items (position -> index) | header {n, n_inside, bounds}
// so_map: SoA table indexed by map slot: Xw float3 | normal float3 | max_dist | min_dist | desc 32 B.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <stdint.h>

#include <vector>

#include "../../include/swarmorb.h"
#include "frame_device.h"

struct so_dframe {
    so_extractor* ex = nullptr;
    int device = 0;
    so::FrameCam cam{};
    int capacity = 0;
    int nlevels = 0;
    float scale[8] = {0};
    bool allocated = false;
    bool in_flight = false;
    bool waited = false;    // so_dframe_wait has run for the frame in flight: the device side is complete
    bool launched = false;  // the frame in flight has a prepare kernel behind it (false for an empty image)
    bool ready = false;    // the device side is complete: n / n_inside / bounds / position map valid, searches may be submitted
    bool mirrors = false;  // collected: the host copies of octave / angle (read by the searches' resolve) are valid
    uint64_t generation = 0;  // bumped by every submit (the matcher's "same frame as before" check)
    so::FramePrepareArgs prep{};   // the prepare launch as captured at the end of the extractor's frame graph
    uint64_t prep_revision = 0;    // bumped when `prep` changes
    // device
    uint8_t* d_block = nullptr;
    float2* d_xy_un = nullptr;
    int8_t* d_octave = nullptr;
    uint8_t* d_desc = nullptr;
    float* d_angle = nullptr;   // by keypoint index
    int32_t* d_cell_start = nullptr;
    int32_t* d_cell_items = nullptr;
    float2* d_s_xy = nullptr;
    int8_t* d_s_octave = nullptr;
    uint4* d_s_desc = nullptr;
    int32_t* d_col_start = nullptr;
    int32_t* d_n_inside = nullptr;
    float* d_bounds = nullptr;
    // host-mapped mirrors (written by the prepare kernel)
    uint8_t* h_block = nullptr;
    uint8_t* h_block_dev = nullptr;
    float* h_xy_un = nullptr;
    int32_t* h_perm = nullptr;
    int32_t* h_header = nullptr;  // n, n_inside, -, -, bounds[4]
    // host copies filled at collect
    int n = 0, n_inside = 0;
    float bounds[4] = {0, 0, 0, 0};
    bool bounds_known = false;  // a frame of this handle has completed: bounds[] holds ComputeImageBounds' result
    std::vector<int32_t> octave;
    std::vector<float> angle;
};

struct so_map {
    int device = 0;
    std::atomic<int> size{0};  // (read by a local-mapping thread's searches while the tracking thread appends)
    int capacity = 0;
    // The tables move when they grow.  A search that reads them from ANOTHER thread than the one that writes the map
    // (so_fuse_kframe_map in a local-mapping thread) holds this shared from taking the pointers until its kernels are
    // done; reserve() takes it exclusively for a reallocation (rare: the capacity doubles).
    std::shared_timed_mutex grow_mu;
    float* d_Xw = nullptr;      // 3 per slot
    float* d_normal = nullptr;  // 3 per slot
    float* d_max = nullptr;
    float* d_min = nullptr;
    uint8_t* d_desc = nullptr;  // 32 per slot
    void* h_stage = nullptr;    // pinned
    size_t h_stage_cap = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;    // false after so_map_share_stream: the writes go out on a matcher's stream
};
