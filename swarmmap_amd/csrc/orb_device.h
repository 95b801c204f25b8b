// orb_device.h — device-side data layout shared by the ORB front-end kernels and their host driver.
//
// HBM layout (one so_extractor = one agent's front-end context; everything is allocated once at the
// first frame and stays resident):
//   level l image     u8   h_l rows x pitch_l bytes, pitch_l = round_up(w_l + 64, 64), base 256-B aligned
//                          (un-blurred; the 19-px reflect-101 border the reference materialises is never
//                          read by FAST / angle / BRIEF — the blur applies reflect-101 by index instead)
//   level l score map u8   TILE-MAJOR: one 1152-byte block per 32 x 32 FAST tile, block (ty, tx) at
//                          (ty * ntx + tx) * 1152: [ring 128 B][body 32 x 32 B].  The ring holds the tile's border
//                          pixels apart from the body (top row 32 B | bottom row 32 B | left column rows 1..30 |
//                          right column rows 1..30): it is ALL a neighbour's low-threshold pass reads of a tile, and
//                          ALL a tile that kept a high-threshold corner has to write besides the scores of its kept
//                          pixels - one 128-byte line instead of a partial store into each of the tile's 32 rows
//                          (the pitched map of rounds 1-2 dirtied every line of every tile: 3.9x the algorithmic
//                          bytes).  Body pixel (row, col) of the tile = ROI pixel (3 + 32 tx + col, 3 + 32 ty + row).
//   level l tile flag u8   nty_l x ntx_l  "tile kept a corner at the high threshold"
//   level l keep bitmap u32 tile-major: 32 words per tile (one 128-byte line), word `row` of tile (ty, tx) at
//                          ((ty * ntx + tx) * 32 + row); bit c of it = ROI pixel
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
constexpr int kScoreRing = 128;                            // ring part of a tile's score block (124 bytes used)
constexpr int kScoreBlock = kScoreRing + kTile * kTile;    // 1152 bytes per tile

struct LevelDesc {
    uint8_t* img;       // level image
    uint8_t* score;     // score map (see layout above)
    uint8_t* tileflag;  // nty*ntx
    uint32_t* bitmap;   // nty*ntx tiles x 32 words
    int w, h, pitch;
    int spitch;
    int ntx, nty;
    int tile_base;  // first global tile index of this level
    int row_base;   // first global bitmap row of this level
};

// position of border pixel (row, col) of a tile inside its ring (row or col is 0 or 31)
__host__ __device__ inline int score_ring_index(int row, int col) {
    if (row == 0) return col;
    if (row == kTile - 1) return kTile + col;
    return col == 0 ? 2 * kTile + (row - 1) : 2 * kTile + (kTile - 2) + (row - 1);
}

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

struct QtLevelArgs {
    int n_target[kMaxLevels];  // mnFeaturesPerLevel
    int sel_stride;            // slots per level in the output
};

// HBM-resident copies of the frame's outputs (read by the device-resident frame, dframe.cpp); members may be null
struct DescribeDeviceOut {
    uint8_t* desc;     // capacity x 32
    float* angle;      // capacity
    SelectedKp* meta;  // capacity
    int32_t* total;
};

// One member of an extraction batch (so_extractor_group, verdict item 9 of round 2): several agents' frames go through
// ONE chain of launches - every kernel of the chain gets one more grid dimension, the member, and reads that member's
// parameters from a device array of these records instead of from its kernel arguments.
struct ExtractBatchMember {
    PyramidParams P;
    QtLevelArgs qt;
    SelectedKp* qt_sel;   // the device quadtree's survivors, nlevels x qt.sel_stride
    int32_t* qt_count;
    struct {
        uint8_t* desc;    // host-mapped results, as launch_describe_qt's arguments
        float* angle;
        SelectedKp* meta;
        int32_t* total;
        DescribeDeviceOut dev;
    } out;
    // != 0: this member sits the current chain out (so_dframe_group_submit with a null image for it: an agent whose next frame is
    // extracted already).  Written by the chain's first kernel (the ingest) from the image table, read by every kernel behind it.
    int skip;
    int prep_sel;  // which of the member's Frame-constructor launches rides at the chain's end (frame_prepare_batch_kernel)
};

// launchers (orb_kernels.hip)
// host_src: width x height bytes, tightly packed, device-visible (pinned / registered host memory)
void launch_ingest(const uint8_t* host_src, int w, int h, const LevelDesc& level0, hipStream_t s);
void launch_resize(const LevelDesc& src, const LevelDesc& dst, hipStream_t s);
// levels first_level+1 .. n-1 from level first_level in one launch (bit-identical to chained launch_resize); false if this configuration does
// not fit the kernel's LDS boxes - the caller then chains launch_resize
bool launch_pyramid_fused(const PyramidParams& p, int first_level, hipStream_t s);
void launch_fast_score(const PyramidParams& p, hipStream_t s);
void launch_fast_low(const PyramidParams& p, hipStream_t s);
// The candidate list (host-quadtree path, so_extractor_get_candidates): d_rowcount is scratch (total_rows ints);
// records go to `cands` (device or host-mapped memory), the header to both hdr_a and hdr_b (either may be null)
void launch_emit(const PyramidParams& p, int32_t* d_rowcount, Candidate* cands, CandidateHeader* hdr_a,
                 CandidateHeader* hdr_b, int cand_capacity, hipStream_t s);
// DistributeOctTree of every level straight off the keep bitmap and the score map
void launch_quadtree(const PyramidParams& p, const int* n_target, int sel_stride, SelectedKp* d_sel, int32_t* d_count,
                     hipStream_t s);
void launch_describe_qt(const PyramidParams& p, const SelectedKp* d_qt_sel, const int32_t* d_qt_count, int qt_stride,
                        int capacity, uint8_t* desc, float* angle, SelectedKp* meta, int32_t* total,
                        const DescribeDeviceOut& dev, hipStream_t s);
void launch_describe(const PyramidParams& p, const SelectedKp* d_sel, int n, uint8_t* d_desc, float* d_angle,
                     hipStream_t s);
// The whole chain (ingest .. describe) for n members with pyramids of the same shape as `first`: d_srcs[n] = device-visible
// pointers to tightly packed w x h images (an array the device can read, e.g. host-mapped; a null entry: the member sits the
// chain out), d_sel[n] (may be null): copied to d_members[i].prep_sel by the first kernel; d_members[n] in device memory.
void launch_extract_batch(const ExtractBatchMember* d_members, const ExtractBatchMember& first, int n, const uint8_t* const* d_srcs,
                          const int32_t* d_sel, int w, int h, bool rows16, int capacity, hipStream_t s);
void launch_quadtree_batch(const ExtractBatchMember* d_members, int n_members, int nlevels, hipStream_t s);

}  // namespace so
