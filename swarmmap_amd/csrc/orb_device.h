// orb_device.h — device-side data layout shared by the ORB front-end kernels and their host driver.
//
// HBM layout (one so_extractor = one agent's front-end context; everything is allocated once at the
// first frame and stays resident):
//   level l image     u8   h_l rows x pitch_l bytes, pitch_l = round_up(w_l + 64, 64), base 256-B aligned
//                          (un-blurred; the 19-px reflect-101 border the reference materialises is never
//                          read by FAST / angle / BRIEF — the blur applies reflect-101 by index instead)
//   level l score map u8   (nty_l*32 + 2) rows x spitch_l bytes; pixel ROI(3+c, 3+r) lives at
//                          [(r+1)*spitch + 4 + c]; a zero frame (1 row top/bottom, 4 cols left, >=1 right)
//                          lets the low-threshold pass read neighbours without bounds checks
//   level l tile flag u8   nty_l x ntx_l  "tile kept a corner at the high threshold"
//   level l keep bitmap u32 (nty_l*32) rows x ntx_l words, bit c of word (row, tx) = ROI pixel
//                          (3 + 32*tx + c, 3 + row) survived NMS
//   candidates        8 B records {i16 x, i16 y, u16 score, u16 level}, levels concatenated, raster order
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace so {

constexpr int kMaxLevels = 8;
constexpr int kFastBorder = 16;  // ROI origin inside a level (EDGE_THRESHOLD-3, ORBextractor.cc:695)
constexpr int kTile = 32;        // tileCalcKeypoints_kernel tile (Fast_gpu.cu:373-374)
constexpr int kFastCap = 10000;  // per-level candidate cap (Fast.hpp:32)
constexpr int kScoreXOff = 4;    // left zero frame of the score map (keeps dword stores aligned)

struct LevelDesc {
    uint8_t* img;       // level image
    uint8_t* score;     // score map (see layout above)
    uint8_t* tileflag;  // nty*ntx
    uint32_t* bitmap;   // (nty*32)*ntx
    int w, h, pitch;
    int spitch;
    int ntx, nty;
    int tile_base;  // first global tile index of this level
    int row_base;   // first global bitmap row of this level
};

struct PyramidParams {
    LevelDesc lv[kMaxLevels];
    int nlevels;
    int total_tiles;
    int total_rows;
    int th_high, th_low;
};

struct Candidate {  // 8 bytes
    int16_t x, y;   // ROI-relative, like GpuFast's kpLoc (Fast_gpu.cu:303,309)
    uint16_t score;
    uint16_t level;
};

struct CandidateHeader {  // written to host-mapped memory by the compaction kernel
    int32_t count[kMaxLevels];   // per level, capped at kFastCap
    int32_t offset[kMaxLevels];  // start of the level inside the record array
    int32_t total;
    int32_t uncapped_total;
    int32_t pad[2];
};

struct SelectedKp {  // host -> device after the quadtree: level coordinates (ROI + 16)
    int16_t x, y;
    uint16_t level;
    uint16_t score;  // FAST response (integer valued)
};

// launchers (orb_kernels.hip)
// host_src: width x height bytes, tightly packed, device-visible (pinned / registered host memory)
void launch_ingest(const uint8_t* host_src, int w, int h, const LevelDesc& level0, hipStream_t s);
void launch_resize(const LevelDesc& src, const LevelDesc& dst, hipStream_t s);
// levels first_level+1 .. n-1 from level first_level in one launch (bit-identical to chained launch_resize); false if this configuration does
// not fit the kernel's LDS boxes - the caller then chains launch_resize
bool launch_pyramid_fused(const PyramidParams& p, int first_level, hipStream_t s);
void launch_fast_score(const PyramidParams& p, hipStream_t s);
void launch_fast_low_count(const PyramidParams& p, int32_t* d_rowcount, hipStream_t s);
// records go to `cands` (device memory for the device quadtree, or host-mapped memory for the host quadtree);
// the header is written to both hdr_a and hdr_b (device copy + host-mapped copy; either may be null)
void launch_emit(const PyramidParams& p, const int32_t* d_rowcount, Candidate* cands, CandidateHeader* hdr_a,
                 CandidateHeader* hdr_b, int cand_capacity, hipStream_t s);
void launch_quadtree(const PyramidParams& p, const int* n_target, int sel_stride, const Candidate* d_cands,
                     const CandidateHeader* d_hdr, SelectedKp* d_sel, int32_t* d_count, hipStream_t s);
// HBM-resident copies of the frame's outputs (read by the device-resident frame, dframe.cpp); members may be null
struct DescribeDeviceOut {
    uint8_t* desc;     // capacity x 32
    float* angle;      // capacity
    SelectedKp* meta;  // capacity
    int32_t* total;
};
void launch_describe_qt(const PyramidParams& p, const SelectedKp* d_qt_sel, const int32_t* d_qt_count, int qt_stride,
                        int capacity, uint8_t* desc, float* angle, SelectedKp* meta, int32_t* total,
                        const DescribeDeviceOut& dev, hipStream_t s);
void launch_describe(const PyramidParams& p, const SelectedKp* d_sel, int n, uint8_t* d_desc, float* d_angle,
                     hipStream_t s);

}  // namespace so
