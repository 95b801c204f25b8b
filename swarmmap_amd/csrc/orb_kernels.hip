// orb_kernels.hip — gfx950 kernels of the ORB front-end (pyramid, FAST-9/16 + NMS, candidate compaction,
// fused orientation + Gaussian + rBRIEF).  Written for CDNA4: 64-lane waves, LDS-staged tiles, ballot
// compaction, all pyramid levels batched into one launch per stage.
//
// Built with -ffp-contract=off: the float stages (bilinear resize, separable Gaussian, BRIEF rotation) are
// defined without implicit FMA so results are bit-identical to the CPU oracle; explicit __builtin_fmaf is
// used where a fused op is part of the definition (polynomial atan2 / sincos).
//
// Reference behaviour followed (files under /root/reference/code):
//   src/cuda/Fast_gpu.cu:63-341   FAST ring layout, corner score, 32x32 tiles with threshold fallback, NMS
//   src/cuda/Fast_gpu.cu:402-470  intensity-centroid angle, addBorder
//   src/cuda/Orb_gpu.cu:63-100    steered BRIEF
//   src/ORBextractor.cc:821-855   pyramid (level l from level l-1), Gaussian 7x7 sigma 2
#include "orb_device.h"

#include <algorithm>
#include <cmath>

namespace so {

// ------------------------------------------------------------------------------------------------
// pyramid: INTER_LINEAR in the OpenCV-CUDA convention (src = dst * (1/f), floor, 4 float taps, rn)
// one thread = 4 horizontally adjacent destination pixels = one aligned dword store
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void resize_body(const uint8_t* __restrict__ src, int sw, int sh, int spitch,
                                            uint8_t* __restrict__ dst, int dw, int dh, int dpitch, float fx, float fy) {
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= dw || y >= dh) return;
    const float src_y = (float)y * fy;
    const int y1 = (int)floorf(src_y);
    const int y2 = y1 + 1;
    const int y2r = min(y2, sh - 1);
    const float wy2 = (float)y2 - src_y;
    const float wy1 = src_y - (float)y1;
    const uint8_t* r1 = src + (size_t)y1 * spitch;
    const uint8_t* r2 = src + (size_t)y2r * spitch;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x4 + k;
        const float src_x = (float)x * fx;
        int x1 = (int)floorf(src_x);
        x1 = min(x1, sw - 1);  // only reachable for the padding lanes x >= dw
        const int x2 = x1 + 1;
        const int x2r = min(x2, sw - 1);
        const float wx2 = (float)x2 - src_x;
        const float wx1 = src_x - (float)x1;
        float out = (float)r1[x1] * (wx2 * wy2);
        out = out + (float)r1[x2r] * (wx1 * wy2);
        out = out + (float)r2[x1] * (wx2 * wy1);
        out = out + (float)r2[x2r] * (wx1 * wy1);
        int v = (int)__builtin_rintf(out);
        v = min(max(v, 0), 255);
        packed |= (uint32_t)v << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)y * dpitch + x4) = packed;
}

__global__ __launch_bounds__(256) void resize_kernel(const uint8_t* __restrict__ src, int sw, int sh, int spitch,
                                                      uint8_t* __restrict__ dst, int dw, int dh, int dpitch,
                                                      float fx, float fy) {
    resize_body(src, sw, sh, spitch, dst, dw, dh, dpitch, fx, fy);
}
// several members' pyramids in one launch (so_extractor_group): blockIdx.z = member
__global__ __launch_bounds__(256) void resize_batch_kernel(const ExtractBatchMember* __restrict__ M, int level, float fx, float fy) {
    if (M[blockIdx.z].skip) return;
    const LevelDesc& S = M[blockIdx.z].P.lv[level - 1];
    const LevelDesc& D = M[blockIdx.z].P.lv[level];
    resize_body(S.img, S.w, S.h, S.pitch, D.img, D.w, D.h, D.pitch, fx, fy);
}


void launch_resize(const LevelDesc& s, const LevelDesc& d, hipStream_t st) {
    const float fx = (float)(1.0 / ((double)d.w / (double)s.w));
    const float fy = (float)(1.0 / ((double)d.h / (double)s.h));
    dim3 block(64, 4);
    dim3 grid((d.w + 255) / 256, (d.h + 3) / 4);
    hipLaunchKernelGGL(resize_kernel, grid, block, 0, st, s.img, s.w, s.h, s.pitch, d.img, d.w, d.h, d.pitch, fx,
                       fy);
}

// ------------------------------------------------------------------------------------------------
// Whole pyramid in ONE launch (ORBextractor::ComputePyramid, code/src/ORBextractor.cc:837-853: level l is resized from
// level l-1, so seven dependent launches of ~4.5 us each used to be 33 us per frame for 1.8 MB of traffic).
// Every level is cut into the same ntx x nty grid of tiles; workgroup (i, j) OWNS tile (i, j) of every level and writes
// exactly that to HBM.  To get there it stages, top level first, the bounding box each level must provide -
// need[l] = own[l] united with the bilinear footprint of need[l+1] - loads need[0] of the level-0 image into LDS and
// walks down the levels in LDS, ping-ponging between two buffers; pixels outside own[l] are recomputed by the
// neighbours that own them (the halo is ~1.3x the owned area at level 0).  Same operations in the same order as
// resize_kernel: bit-identical levels.
// ------------------------------------------------------------------------------------------------
constexpr int kPyrTile = 16;   // tile edge at the coarsest level
constexpr int kPyrBuf = 96;    // LDS region edge (pixels) the kernel can hold per level

struct PyramidFusedArgs {
    float fx[kMaxLevels], fy[kMaxLevels];  // level l from l-1: source coordinate = destination coordinate * f[l]
    int ntx, nty;
    int first;  // levels first+1 .. n-1 are produced from level `first` (already in HBM)
};

__device__ __forceinline__ void pyr_footprint(int a0, int a1, float f, int smax, int& s0, int& s1) {
    // source columns (rows) the destination span [a0, a1) reads: floor(a f) .. min(floor((a1-1) f) + 1, smax)
    s0 = min((int)floorf((float)a0 * f), smax);
    s1 = min(min((int)floorf((float)(a1 - 1) * f), smax) + 1, smax) + 1;
}

__global__ __launch_bounds__(256) void pyramid_fused_kernel(PyramidParams P, PyramidFusedArgs A) {
    __shared__ uint8_t buf[2][kPyrBuf * kPyrBuf];
    __shared__ int s_box[kMaxLevels][4];  // need[l]: x0, y0, x1, y1 (exclusive)
    const int tid = threadIdx.x;
    const int ti = blockIdx.x % A.ntx, tj = blockIdx.x / A.ntx;
    const int nl = P.nlevels;
    if (tid == 0) {
        int nx0 = 0, ny0 = 0, nx1 = 0, ny1 = 0;
        for (int l = nl - 1; l >= A.first; l--) {
            const LevelDesc& L = P.lv[l];
            int x0 = (int)((long long)ti * L.w / A.ntx), x1 = (int)((long long)(ti + 1) * L.w / A.ntx);
            int y0 = (int)((long long)tj * L.h / A.nty), y1 = (int)((long long)(tj + 1) * L.h / A.nty);
            if (l == A.first) x1 = x0, y1 = y0;  // the source level is only read: its box is the footprint alone
            if (l < nl - 1 && nx1 > nx0 && ny1 > ny0) {  // footprint of the coarser level's box in this one
                int fx0, fx1, fy0, fy1;
                pyr_footprint(nx0, nx1, A.fx[l + 1], L.w - 1, fx0, fx1);
                pyr_footprint(ny0, ny1, A.fy[l + 1], L.h - 1, fy0, fy1);
                if (x1 > x0 && y1 > y0) {
                    x0 = min(x0, fx0); x1 = max(x1, fx1); y0 = min(y0, fy0); y1 = max(y1, fy1);
                } else {
                    x0 = fx0; x1 = fx1; y0 = fy0; y1 = fy1;
                }
            }
            s_box[l][0] = x0; s_box[l][1] = y0; s_box[l][2] = x1; s_box[l][3] = y1;
            nx0 = x0; ny0 = y0; nx1 = x1; ny1 = y1;
        }
    }
    __syncthreads();
    {   // the needed box of the source level into LDS
        const LevelDesc& L = P.lv[A.first];
        const int x0 = s_box[A.first][0], y0 = s_box[A.first][1], bw = s_box[A.first][2] - x0, bh = s_box[A.first][3] - y0;
        for (int i = tid; i < bw * bh; i += 256) {
            const int r = i / bw, c = i - r * bw;
            buf[A.first & 1][r * kPyrBuf + c] = L.img[(size_t)(y0 + r) * L.pitch + x0 + c];
        }
    }
    __syncthreads();
    for (int l = A.first + 1; l < nl; l++) {
        const LevelDesc& S = P.lv[l - 1];
        const LevelDesc& D = P.lv[l];
        const uint8_t* src = buf[(l - 1) & 1];
        uint8_t* dst = buf[l & 1];
        const int sx0 = s_box[l - 1][0], sy0 = s_box[l - 1][1];
        const int dx0 = s_box[l][0], dy0 = s_box[l][1], bw = s_box[l][2] - dx0, bh = s_box[l][3] - dy0;
        const int ox0 = (int)((long long)ti * D.w / A.ntx), ox1 = (int)((long long)(ti + 1) * D.w / A.ntx);
        const int oy0 = (int)((long long)tj * D.h / A.nty), oy1 = (int)((long long)(tj + 1) * D.h / A.nty);
        const float fx = A.fx[l], fy = A.fy[l];
        for (int i = tid; i < bw * bh; i += 256) {
            const int r = i / bw, c = i - r * bw;
            const int x = dx0 + c, y = dy0 + r;
            const float src_y = (float)y * fy;
            const int y1 = (int)floorf(src_y);
            const int y2 = y1 + 1;
            const int y2r = min(y2, S.h - 1);
            const float wy2 = (float)y2 - src_y;
            const float wy1 = src_y - (float)y1;
            const float src_x = (float)x * fx;
            int x1 = (int)floorf(src_x);
            x1 = min(x1, S.w - 1);
            const int x2 = x1 + 1;
            const int x2r = min(x2, S.w - 1);
            const float wx2 = (float)x2 - src_x;
            const float wx1 = src_x - (float)x1;
            const uint8_t* r1 = src + (y1 - sy0) * kPyrBuf - sx0;
            const uint8_t* r2 = src + (y2r - sy0) * kPyrBuf - sx0;
            float out = (float)r1[x1] * (wx2 * wy2);
            out = out + (float)r1[x2r] * (wx1 * wy2);
            out = out + (float)r2[x1] * (wx2 * wy1);
            out = out + (float)r2[x2r] * (wx1 * wy1);
            int v = (int)__builtin_rintf(out);
            v = min(max(v, 0), 255);
            dst[r * kPyrBuf + c] = (uint8_t)v;
            if (x >= ox0 && x < ox1 && y >= oy0 && y < oy1) D.img[(size_t)y * D.pitch + x] = (uint8_t)v;
        }
        __syncthreads();
    }
}

// false: the boxes of this configuration do not fit the kernel's LDS regions (the caller chains resize launches)
bool launch_pyramid_fused(const PyramidParams& P, int first, hipStream_t st) {
    const int nl = P.nlevels;
    if (nl - first < 2) return true;
    PyramidFusedArgs A{};
    A.first = first;
    for (int l = 1; l < nl; l++) {
        A.fx[l] = (float)(1.0 / ((double)P.lv[l].w / (double)P.lv[l - 1].w));
        A.fy[l] = (float)(1.0 / ((double)P.lv[l].h / (double)P.lv[l - 1].h));
    }
    const LevelDesc& top = P.lv[nl - 1];
    A.ntx = (top.w + kPyrTile - 1) / kPyrTile;
    A.nty = (top.h + kPyrTile - 1) / kPyrTile;
    // largest box over all tiles and levels, with the kernel's own arithmetic (worst tile = any: sizes differ by 1)
    for (int tj = 0; tj < A.nty; tj += (A.nty > 1 ? A.nty - 1 : 1))
        for (int ti = 0; ti < A.ntx; ti += (A.ntx > 1 ? A.ntx - 1 : 1)) {
            int nx0 = 0, ny0 = 0, nx1 = 0, ny1 = 0;
            for (int l = nl - 1; l >= first; l--) {
                const LevelDesc& L = P.lv[l];
                int x0 = (int)((long long)ti * L.w / A.ntx), x1 = (int)((long long)(ti + 1) * L.w / A.ntx);
                int y0 = (int)((long long)tj * L.h / A.nty), y1 = (int)((long long)(tj + 1) * L.h / A.nty);
                if (l < nl - 1) {
                    const int fx0 = std::min((int)floorf((float)nx0 * A.fx[l + 1]), L.w - 1);
                    const int fx1 = std::min(std::min((int)floorf((float)(nx1 - 1) * A.fx[l + 1]), L.w - 1) + 1, L.w - 1) + 1;
                    const int fy0 = std::min((int)floorf((float)ny0 * A.fy[l + 1]), L.h - 1);
                    const int fy1 = std::min(std::min((int)floorf((float)(ny1 - 1) * A.fy[l + 1]), L.h - 1) + 1, L.h - 1) + 1;
                    x0 = std::min(x0, fx0); x1 = std::max(x1, fx1); y0 = std::min(y0, fy0); y1 = std::max(y1, fy1);
                }
                if (x1 - x0 + 2 > kPyrBuf || y1 - y0 + 2 > kPyrBuf) return false;
                nx0 = x0; ny0 = y0; nx1 = x1; ny1 = y1;
            }
        }
    hipLaunchKernelGGL(pyramid_fused_kernel, dim3(A.ntx * A.nty), dim3(256), 0, st, P, A);
    return true;
}

// ------------------------------------------------------------------------------------------------
// image ingest: a tightly packed host image (pinned memory, read in place over PCIe) -> the pitched level-0 image.
// One launch instead of a linear DMA into a landing buffer plus a pitched device-to-device copy (two runtime calls of
// ~8 us each on the tracking thread); a pitched host -> device DMA is split into one copy per row when the width is
// not a multiple of four bytes (1241-px KITTI rows: 376 copies, 2.2 ms).  Every thread produces one aligned dword of
// the destination from two aligned dwords of the source (v_alignbyte), whatever the row offset's alignment.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ingest_body(const uint8_t* __restrict__ src, int w, int h, uint8_t* __restrict__ dst, int pitch) {
    // (a wave = 64 consecutive dwords of one row: threadIdx.y selects the row)
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const bool in = x4 < w && y < h;
    const size_t off = (size_t)(in ? y : 0) * w + (in ? x4 : 0);  // byte offset of the first of the four pixels
    const size_t a = off & ~(size_t)3;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    const size_t last = ((size_t)w * h + 3) / 4 - 1;  // last readable dword of the source (rounded-up allocation is not assumed)
    const uint32_t lo = in ? s32[a >> 2] : 0u;
    // The second source dword of a lane is the first one of its right neighbour: every dword crosses PCIe ONCE (it was
    // read twice, as `lo` here and as `hi` there - twice the traffic of the one transfer on this path that is bound by
    // the link).  Rows that start dword-aligned need no second dword at all; the last lane of a wave fetches its own.
    const uint32_t shift = (uint32_t)(off & 3);
    uint32_t hi = (uint32_t)__shfl_down((int)lo, 1);
    if (shift != 0 && (threadIdx.x == 63 || x4 + 4 >= w)) hi = (in && (a >> 2) < last) ? s32[(a >> 2) + 1] : 0u;
    if (!in) return;
    const uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, shift);
    *reinterpret_cast<uint32_t*>(dst + (size_t)y * pitch + x4) = v;  // bytes beyond w land in the row's padding
}

__global__ __launch_bounds__(256) void ingest_kernel(const uint8_t* __restrict__ src, int w, int h, uint8_t* __restrict__ dst,
                                                      int pitch) {
    ingest_body(src, w, h, dst, pitch);
}
// (a null image: the member sits this chain out - the flag every later kernel of the chain tests is set here)
__device__ __forceinline__ bool ingest_member_skips(const uint8_t* src, const int32_t* sel, const ExtractBatchMember* M) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && threadIdx.y == 0) {
        ExtractBatchMember& m = const_cast<ExtractBatchMember*>(M)[blockIdx.z];
        m.skip = src == nullptr ? 1 : 0;
        m.prep_sel = sel ? sel[blockIdx.z] : 0;
    }
    return src == nullptr;
}
__global__ __launch_bounds__(256) void ingest_batch_kernel(const uint8_t* const* __restrict__ srcs, const int32_t* __restrict__ sel,
                                                            const ExtractBatchMember* M, int w, int h) {
    const uint8_t* src = srcs[blockIdx.z];
    if (ingest_member_skips(src, sel, M)) return;
    const LevelDesc& L0 = M[blockIdx.z].P.lv[0];
    ingest_body(src, w, h, L0.img, L0.pitch);
}


// rows that are a multiple of 16 bytes wide from a 16-byte aligned source (752-px EuRoC rows): one aligned 16-byte read
// and one aligned 16-byte store per thread - a quarter of the requests on the link
__device__ __forceinline__ void ingest16_body(const uint4* __restrict__ src, int w16, int h, uint8_t* __restrict__ dst, int pitch) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w16 || y >= h) return;
    *reinterpret_cast<uint4*>(dst + (size_t)y * pitch + 16 * (size_t)x) = src[(size_t)y * w16 + x];
}

__global__ __launch_bounds__(256) void ingest16_kernel(const uint4* __restrict__ src, int w16, int h, uint8_t* __restrict__ dst,
                                                        int pitch) {
    ingest16_body(src, w16, h, dst, pitch);
}
__global__ __launch_bounds__(256) void ingest16_batch_kernel(const uint8_t* const* __restrict__ srcs, const int32_t* __restrict__ sel,
                                                              const ExtractBatchMember* M, int w16, int h) {
    const uint8_t* src = srcs[blockIdx.z];
    if (ingest_member_skips(src, sel, M)) return;
    const LevelDesc& L0 = M[blockIdx.z].P.lv[0];
    ingest16_body(reinterpret_cast<const uint4*>(src), w16, h, L0.img, L0.pitch);
}


void launch_ingest(const uint8_t* host_src, int w, int h, const LevelDesc& level0, hipStream_t s) {
    if ((w & 15) == 0 && (reinterpret_cast<uintptr_t>(host_src) & 15) == 0 && (level0.pitch & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(level0.img) & 15) == 0) {
        dim3 block16(64, 4);
        dim3 grid16((w / 16 + 63) / 64, (h + 3) / 4);
        hipLaunchKernelGGL(ingest16_kernel, grid16, block16, 0, s, reinterpret_cast<const uint4*>(host_src), w / 16, h, level0.img,
                           level0.pitch);
        return;
    }
    dim3 block(64, 4);
    dim3 grid((w + 255) / 256, (h + 3) / 4);
    hipLaunchKernelGGL(ingest_kernel, grid, block, 0, s, host_src, w, h, level0.img, level0.pitch);
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(a, max(b, c)); }

// Largest threshold t at which the pixel is still a FAST-9 corner (== cornerScore's binary search,
// Fast_gpu.cu:195-218), from the 16 ring differences d[k] = ring[k] - centre:
//   bright run: all 9 consecutive d > t  <=>  t < min over the run   -> max over runs of (min) - 1
//   dark   run: all 9 consecutive d < -t <=>  t < min over run of -d -> -(min over runs of (max)) - 1
__device__ __forceinline__ int fast_corner_score(const int (&d)[16]) {
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        mn3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        mx3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    int best_b = -1000, best_d = 1000;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        best_b = max(best_b, min3i(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]));
        best_d = min(best_d, max3i(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]));
    }
    return max(best_b, -best_d) - 1;
}

__device__ __forceinline__ int find_level_by_tile(const PyramidParams& P, int tile) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; i++)
        if (i < P.nlevels && tile >= P.lv[i].tile_base) l = i;
    return l;
}

constexpr int kImgRows = 40, kImgPitchW = 12;  // 40 x 48-byte LDS image tile (dword granularity)
constexpr int kScRows = 34, kScPitch = 36;     // 34 x 34 scores (tile + 1-px halo)

// One workgroup (4 waves) per 32x32 tile of one pyramid level; all levels in one launch.
// Stage the 40x40 neighbourhood in LDS with aligned dword loads, score the 34x34 halo'd tile, NMS at the
// high threshold inside LDS (halo scores are recomputed here, which removes the reference's cross-block
// race on scoreMat), write the u8 score tile + tile flag + keep-bitmap words.
__device__ __forceinline__ void fast_score_body(const PyramidParams& P) {
    __shared__ uint32_t simg32[kImgRows * kImgPitchW];
    __shared__ uint8_t ssc[kScRows * kScPitch];
    const uint8_t* simg = reinterpret_cast<const uint8_t*>(simg32);

    const int tid = threadIdx.x;
    // XCD-aware block -> tile map.  Workgroup b runs on XCD b % 8 and each XCD has a private L2; a tile shares its
    // 128-byte image lines with the tiles left and right of it and its 8 halo rows with the tiles above and below.  So
    // every XCD takes one CONTIGUOUS run of the raster-ordered tiles (an eighth of them, 5-6 bands of level 0) and
    // walks it in order: lines are shared inside one L2, and only the rows at a run's two ends are fetched twice.
    // (FETCH_SIZE: a plain b -> tile map read 4.3x the image bytes, groups of four tiles per XCD 2.3x.)
    const int chunk = (P.total_tiles + 7) >> 3;
    const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile >= P.total_tiles) return;  // the grid is 8 * chunk workgroups
    const int lvl = find_level_by_tile(P, tile);
    const LevelDesc& L = P.lv[lvl];
    const int t = tile - L.tile_base;
    const int ty = t / L.ntx, tx = t - ty * L.ntx;
    const int x0 = 19 + kTile * tx, y0 = 19 + kTile * ty;  // tile origin in level coordinates
    const int ax = 12 + kTile * tx;                         // (x0 - 4) rounded down to a dword

    for (int i = tid; i < kImgRows * kImgPitchW; i += 256) {
        const int r = i / kImgPitchW, dc = i - r * kImgPitchW;
        const int gy = min(y0 - 4 + r, L.h - 1);
        simg32[i] = *reinterpret_cast<const uint32_t*>(L.img + (size_t)gy * L.pitch + ax + 4 * dc);
    }
    __syncthreads();

    const int th_low = P.th_low, th_high = P.th_high;
    // Pass 1 - the four compass points of the ring: a run of nine out of sixteen contains one point of every opposite
    // pair, so a pixel whose pairs (0, 8) and (4, 12) cannot both supply a point brighter than v + th_low (or both a
    // darker one) has no score >= th_low: 87 % of the pixels of the benchmark stream stop here with score 0.  The others
    // are compacted into a list (ballot + one LDS atomic per wave and round) and pass 2 runs the closed-form score on
    // dense lanes.  (Scoring every pixel made this kernel ALU-bound as soon as several frames shared a launch.)
    __shared__ uint16_t s_list[kScRows * kScRows];
    __shared__ int s_nlist;
    if (tid == 0) s_nlist = 0;
    __syncthreads();
    constexpr int PW = kImgPitchW * 4;
    for (int p0 = 0; p0 < kScRows * kScRows; p0 += 256) {
        const int p = p0 + tid;
        bool cand = false;
        if (p < kScRows * kScRows) {
            const int r = p / kScRows, c = p - r * kScRows;
            const int x = x0 - 1 + c, y = y0 - 1 + r;
            if (x >= 19 && x < L.w - 19 && y >= 19 && y < L.h - 19) {
                const uint8_t* ctr = simg + (r + 3) * PW + (c + 6);
                const int v = ctr[0];
                const int d0 = ctr[3 * PW] - v, d8 = ctr[-3 * PW] - v, d4 = ctr[3] - v, d12 = ctr[-3] - v;
                const bool bright = (d0 > th_low || d8 > th_low) && (d4 > th_low || d12 > th_low);
                const bool dark = (d0 < -th_low || d8 < -th_low) && (d4 < -th_low || d12 < -th_low);
                cand = bright || dark;
            }
            if (!cand) ssc[r * kScPitch + c] = 0;
        }
        const unsigned long long mask = __ballot(cand);
        int base = 0;
        if ((tid & 63) == 0 && mask) base = atomicAdd(&s_nlist, __popcll(mask));
        base = __shfl(base, 0);
        if (cand) s_list[base + __popcll(mask & ((1ull << (tid & 63)) - 1ull))] = (uint16_t)p;
    }
    __syncthreads();
    const int nlist = s_nlist;
    for (int i = tid; i < nlist; i += 256) {
        const int p = s_list[i];
        const int r = p / kScRows, c = p - r * kScRows;
        const uint8_t* ctr = simg + (r + 3) * PW + (c + 6);
        const int v = ctr[0];
        int d[16];
        d[0] = ctr[3 * PW + 0] - v;   d[1] = ctr[3 * PW + 1] - v;   d[2] = ctr[2 * PW + 2] - v;
        d[3] = ctr[1 * PW + 3] - v;   d[4] = ctr[3] - v;            d[5] = ctr[-1 * PW + 3] - v;
        d[6] = ctr[-2 * PW + 2] - v;  d[7] = ctr[-3 * PW + 1] - v;  d[8] = ctr[-3 * PW + 0] - v;
        d[9] = ctr[-3 * PW - 1] - v;  d[10] = ctr[-2 * PW - 2] - v; d[11] = ctr[-1 * PW - 3] - v;
        d[12] = ctr[-3] - v;          d[13] = ctr[1 * PW - 3] - v;  d[14] = ctr[2 * PW - 2] - v;
        d[15] = ctr[3 * PW - 1] - v;
        const int s = fast_corner_score(d);
        ssc[r * kScPitch + c] = (uint8_t)(s >= th_low ? s : 0);
    }
    __syncthreads();

    // NMS at the high threshold: each wave covers two tile rows per round -> one ballot = two bitmap words
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long keep[4];
    int any = 0, my_score[4];
    bool my_kp[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int row = 8 * k + 2 * wave + (lane >> 5), col = lane & 31;
        const uint8_t* q = ssc + (row + 1) * kScPitch + (col + 1);
        const int s = q[0];
        const int m = max(max(max((int)q[-kScPitch - 1], (int)q[-kScPitch]), max((int)q[-kScPitch + 1], (int)q[-1])),
                          max(max((int)q[1], (int)q[kScPitch - 1]), max((int)q[kScPitch], (int)q[kScPitch + 1])));
        const bool kp = (s >= th_high) && (s > m);
        my_kp[k] = kp;
        my_score[k] = s;
        keep[k] = __ballot(kp);
        any |= (keep[k] != 0ull);
    }
    const int has1 = __syncthreads_or(any);

    // u8 score tile -> its block of the tile-major score map.  Later stages read of a tile (a) its border pixels - the
    // halo of a neighbour's low-threshold pass -, (b) the scores of its kept pixels (candidate records) and, only if the
    // tile is empty, (c) everything (its own low-threshold pass).  So every tile writes its RING (one 128-byte line), an
    // empty tile its whole body (1 KB, contiguous), a tile with high-threshold corners just the kept pixels.
    uint8_t* blk = L.score + (size_t)t * kScoreBlock;
    if (tid < 31) {  // ring as 31 dwords: 8 top, 8 bottom, 15 of the two columns
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int bidx = 4 * tid + k;
            int row, col;
            if (bidx < kTile) { row = 0; col = bidx; }
            else if (bidx < 2 * kTile) { row = kTile - 1; col = bidx - kTile; }
            else if (bidx < 2 * kTile + (kTile - 2)) { row = bidx - 2 * kTile + 1; col = 0; }
            else { row = bidx - (2 * kTile + kTile - 2) + 1; col = kTile - 1; }
            v |= (uint32_t)ssc[(row + 1) * kScPitch + (col + 1)] << (8 * k);
        }
        reinterpret_cast<uint32_t*>(blk)[tid] = v;
    }
    if (!has1) {
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const uint8_t* q = ssc + (row + 1) * kScPitch + (c4 + 1);
        const uint32_t v = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        reinterpret_cast<uint32_t*>(blk + kScoreRing)[tid] = v;  // body (row, c4 .. c4 + 3)
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (my_kp[k]) {
                const int row = 8 * k + 2 * wave + (lane >> 5), col = lane & 31;
                blk[kScoreRing + row * kTile + col] = (uint8_t)my_score[k];
            }
    }
    if (tid == 0) L.tileflag[t] = (uint8_t)has1;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = 8 * k + 2 * wave;
            uint32_t* b = L.bitmap + (size_t)t * kTile + row;  // tile-major: a tile's 32 words are one 128-byte line
            b[0] = has1 ? (uint32_t)keep[k] : 0u;
            b[1] = has1 ? (uint32_t)(keep[k] >> 32) : 0u;
        }
    }
}

__global__ __launch_bounds__(256) void fast_score_kernel(PyramidParams P) { fast_score_body(P); }
__global__ __launch_bounds__(256) void fast_score_batch_kernel(const ExtractBatchMember* __restrict__ M) {
    if (M[blockIdx.y].skip) return;
    fast_score_body(M[blockIdx.y].P);
}


// ------------------------------------------------------------------------------------------------
// Low-threshold pass and candidate emission.
//   fast_low_kernel   one WAVE per tile (four tiles per workgroup, all levels in one grid): an empty tile gets its
//                     low-threshold pass (Fast_gpu.cu:317-339) with the deterministic neighbour rule of the oracle (a
//                     neighbour in a non-empty tile competes with its high-threshold score, one in an empty tile with
//                     its low-threshold score) and writes its 32 keep words.  Rounds 1-2 ran this per band, a wave
//                     walking up to six empty tiles one after the other: 15 us on 72 workgroups.
// The device quadtree (quadtree_kernel.hip) reads the keep bitmap and the score map itself.  The candidate LIST the
// reference hands to DistributeOctTree only exists on the host-quadtree path and for so_extractor_get_candidates:
//   rowcount_kernel   per band (one row of tiles of one level, 72 workgroups for 752x480): keypoints per pixel row
//   emit_kernel       per band: every band scans rowcount[] (a few thousand ints, L2-resident) to find where its rows
//                     start in the (level, y, x) raster order, applies the per-level cap of 10000, and writes its
//                     8-byte records.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_level_by_band(const PyramidParams& P, int band) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; i++)
        if (i < P.nlevels && band * kTile >= P.lv[i].row_base) l = i;
    return l;
}

constexpr int kMaxTilesX = 128;  // tiles per band (images up to 4k pixels wide)
constexpr int kMaxRows = 8192;   // bitmap rows over all levels (sum of level heights; ~2300 for 752x480)

constexpr int kLowPitch = 40;  // bytes per row of the per-wave score window (10 aligned dwords)

__device__ __forceinline__ void fast_low_body(const PyramidParams& P) {
    __shared__ uint32_t ssc32[4][kScRows * (kLowPitch / 4)];  // one 34-row score window per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // groups of four tiles, one contiguous run of groups per XCD (same reasoning as fast_score_kernel: the halo comes out
    // of the neighbours' rings, which the XCD's L2 holds when the neighbours are its own)
    const int ngroups = (P.total_tiles + 3) >> 2, gchunk = (ngroups + 7) >> 3;
    const int tile = (((int)(blockIdx.x & 7) * gchunk + (int)(blockIdx.x >> 3)) << 2) + wave;
    if (tile >= P.total_tiles) return;  // wave-uniform; no workgroup barrier below
    const int lvl = find_level_by_tile(P, tile);
    const LevelDesc& L = P.lv[lvl];
    const int t = tile - L.tile_base;
    const int ty = t / L.ntx, tx = t - ty * L.ntx;
    const int th_high = P.th_high;
    uint32_t* gb = L.bitmap + (size_t)t * kTile;
    if (L.tileflag[t]) return;  // kept corners at the high threshold: its words are final
    uint32_t* sc32 = ssc32[wave];
    uint8_t* sc = reinterpret_cast<uint8_t*>(sc32);
    // window pixel (r, c) of the 34 x 34 neighbourhood sits at byte (c + 3) of row r (40-byte rows): the tile's body
    // (1 KB, contiguous) lands dword by dword, the 132 halo pixels come out of the eight neighbours' rings
    {
        const uint32_t* body = reinterpret_cast<const uint32_t*>(L.score + (size_t)t * kScoreBlock + kScoreRing);
#pragma unroll
        for (int j = 0; j < kTile * (kTile / 4) / 64; j++) {
            const int i = lane + 64 * j;
            const int row = i >> 3, d4 = i & 7;
            sc32[(row + 1) * (kLowPitch / 4) + d4 + 1] = body[i];
        }
    }
    // halo pixels belong to neighbour tiles: one that kept high-threshold corners competes with its
    // high-threshold score only (132 halo pixels: rows 0 / 33 and columns 0 / 33); outside the grid: 0
    for (int i = lane; i < 4 * kScRows; i += 64) {
        const int side = i / kScRows, k = i - side * kScRows;
        const int r = side == 0 ? 0 : (side == 1 ? kScRows - 1 : k);
        const int c = side == 2 ? 0 : (side == 3 ? kScRows - 1 : k);
        const int tyq = ty + (r == 0 ? -1 : (r == kScRows - 1 ? 1 : 0));
        const int txq = tx + (c == 0 ? -1 : (c == kScRows - 1 ? 1 : 0));
        int v = 0;
        if (tyq >= 0 && tyq < L.nty && txq >= 0 && txq < L.ntx) {
            // the pixel's place inside its own tile: one past the edge wraps to the far edge of the neighbour
            const int prow = r == 0 ? kTile - 1 : (r == kScRows - 1 ? 0 : r - 1);
            const int pcol = c == 0 ? kTile - 1 : (c == kScRows - 1 ? 0 : c - 1);
            v = L.score[(size_t)(tyq * L.ntx + txq) * kScoreBlock + score_ring_index(prow, pcol)];
            if (L.tileflag[tyq * L.ntx + txq] && v < th_high) v = 0;
        }
        sc[r * kLowPitch + c + 3] = (uint8_t)v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t mine = 0;  // lane r < 32 ends up with keep word r
#pragma unroll 4
    for (int k = 0; k < 16; k++) {  // two tile rows per ballot
        const int row = 2 * k + (lane >> 5), col = lane & 31;
        const uint8_t* q = sc + (row + 1) * kLowPitch + (col + 1) + 3;
        const int s = q[0];
        const int m = max(max(max((int)q[-kLowPitch - 1], (int)q[-kLowPitch]), max((int)q[-kLowPitch + 1], (int)q[-1])),
                          max(max((int)q[1], (int)q[kLowPitch - 1]), max((int)q[kLowPitch], (int)q[kLowPitch + 1])));
        const unsigned long long keep = __ballot(s > 0 && s > m);
        if (lane == 2 * k) mine = (uint32_t)keep;
        if (lane == 2 * k + 1) mine = (uint32_t)(keep >> 32);
    }
    if (lane < kTile) gb[lane] = mine;  // one 128-byte line
}

__global__ __launch_bounds__(256) void fast_low_kernel(PyramidParams P) { fast_low_body(P); }
__global__ __launch_bounds__(256) void fast_low_batch_kernel(const ExtractBatchMember* __restrict__ M) {
    if (M[blockIdx.y].skip) return;
    fast_low_body(M[blockIdx.y].P);
}


__global__ __launch_bounds__(256) void rowcount_kernel(PyramidParams P, int32_t* __restrict__ rowcount) {
    const int tid = threadIdx.x;
    const int lvl = find_level_by_band(P, blockIdx.x);
    const LevelDesc& L = P.lv[lvl];
    const int ty = blockIdx.x - L.row_base / kTile;
    const int row = tid >> 3, part = tid & 7;  // 8 threads per pixel row
    int cnt = 0;
    for (int w = part; w < L.ntx; w += 8) cnt += __popc(L.bitmap[(size_t)(ty * L.ntx + w) * kTile + row]);
    cnt += __shfl_xor(cnt, 1);
    cnt += __shfl_xor(cnt, 2);
    cnt += __shfl_xor(cnt, 4);
    if (part == 0) rowcount[L.row_base + ty * kTile + row] = cnt;
}

__global__ __launch_bounds__(256) void emit_kernel(PyramidParams P, const int32_t* __restrict__ rowcount,
                                                    Candidate* __restrict__ h_cands,
                                                    CandidateHeader* __restrict__ h_header,
                                                    CandidateHeader* __restrict__ h_header2, int cand_capacity) {
    __shared__ int s_wave[4];
    __shared__ int s_lvl_start[kMaxLevels + 1];
    __shared__ int s_lvl_out[kMaxLevels + 1];
    __shared__ int s_row_start[kTile];
    __shared__ int s_rc[kMaxRows];
    __shared__ uint32_t sbm[kTile][kMaxTilesX];   // the band's words
    __shared__ int s_rank[kTile][kMaxTilesX];      // rank (inside the level) of each word's first keypoint
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int R = P.total_rows;
    const int lvl = find_level_by_band(P, blockIdx.x);
    const LevelDesc& L = P.lv[lvl];
    const int my_row0 = blockIdx.x * kTile;  // global bitmap row of this band's first row
    const int ty = blockIdx.x - L.row_base / kTile;

    for (int g = tid; g < R; g += 256) s_rc[g] = rowcount[g];
    for (int i = tid; i < kTile * L.ntx; i += 256) {
        const int tx = i >> 5, row = i & 31;
        sbm[row][tx] = L.bitmap[(size_t)(ty * L.ntx) * kTile + i];
    }
    __syncthreads();
    // exclusive prefix over all rows (thread t owns a contiguous chunk), redundantly in every band
    const int rpt = (R + 255) / 256;
    const int g0 = min(tid * rpt, R), g1 = min(g0 + rpt, R);
    int mine = 0;
    for (int g = g0; g < g1; g++) mine += s_rc[g];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int i = 0; i < wave; i++) wave_base += s_wave[i];
    const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    {
        int run = incl - mine + wave_base;
        for (int g = g0; g < g1; g++) {
#pragma unroll
            for (int l = 0; l < kMaxLevels; l++)
                if (l < P.nlevels && P.lv[l].nty > 0 && g == P.lv[l].row_base) s_lvl_start[l] = run;
            if (g >= my_row0 && g < my_row0 + kTile) s_row_start[g - my_row0] = run;
            run += s_rc[g];
        }
    }
    if (tid == 0) s_lvl_start[P.nlevels] = total;
    __syncthreads();
    if (tid == 0) {
        for (int l = P.nlevels - 1; l >= 0; l--)  // a level too small to hold a FAST tile owns no rows
            if (P.lv[l].nty == 0) s_lvl_start[l] = s_lvl_start[l + 1];
        int off = 0;
        for (int l = 0; l < P.nlevels; l++) {
            s_lvl_out[l] = off;
            off += min(s_lvl_start[l + 1] - s_lvl_start[l], kFastCap);
        }
        s_lvl_out[P.nlevels] = off;
        if (blockIdx.x == 0) {
            for (int which = 0; which < 2; which++) {
                CandidateHeader* hd = which == 0 ? h_header : h_header2;
                if (!hd) continue;
                for (int l = 0; l < kMaxLevels; l++) {
                    hd->count[l] = l < P.nlevels ? min(s_lvl_start[l + 1] - s_lvl_start[l], kFastCap) : 0;
                    hd->offset[l] = l < P.nlevels ? s_lvl_out[l] : off;
                }
                hd->total = off;
                hd->uncapped_total = total;
            }
        }
    }
    __syncthreads();

    // per-word ranks: one wave per row (8 rows per wave), all in LDS
    const int lvl_start = s_lvl_start[lvl], lvl_out = s_lvl_out[lvl];
    for (int rr = wave; rr < kTile; rr += 4) {
        int base = s_row_start[rr] - lvl_start;
        for (int w0 = 0; w0 < L.ntx; w0 += 64) {
            const int w = w0 + lane;
            const int c = w < L.ntx ? __popc(sbm[rr][w]) : 0;
            int incl2 = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl2, off);
                if (lane >= off) incl2 += v;
            }
            if (w < L.ntx) s_rank[rr][w] = base + incl2 - c;
            base += __shfl(incl2, 63);
        }
    }
    __syncthreads();
    // emission: every thread takes words round-robin, four at a time: the 32 score bytes behind each non-empty word
    // (one body row of a tile) are fetched as two 16-byte loads, all eight in flight before the first record is built
    // (a byte load per kept pixel chained one memory round trip per keypoint)
    const int nwords = kTile * L.ntx;
    for (int i0 = tid; i0 < nwords; i0 += 4 * 256) {
        uint32_t word[4];
        uint4 lo[4], hi[4];
        int rr4[4], w4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + 256 * u;
            word[u] = 0;
            if (i < nwords) {
                rr4[u] = i / L.ntx;
                w4[u] = i - rr4[u] * L.ntx;
                word[u] = sbm[rr4[u]][w4[u]];
            }
            if (word[u]) {  // body row rr of tile (ty, w)
                const uint4* srow = reinterpret_cast<const uint4*>(L.score + (size_t)(ty * L.ntx + w4[u]) * kScoreBlock +
                                                                    kScoreRing + rr4[u] * kTile);
                lo[u] = srow[0];
                hi[u] = srow[1];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!word[u]) continue;
            const uint32_t d[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
            int rank = s_rank[rr4[u]][w4[u]];
            const int y = 3 + ty * kTile + rr4[u], xb = 3 + 32 * w4[u];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint32_t sub = (word[u] >> (4 * j)) & 15u;
                while (sub) {
                    const int b = __ffs(sub) - 1;
                    sub &= sub - 1;
                    if (rank < kFastCap) {
                        const int o = lvl_out + rank;
                        if (o < cand_capacity) {
                            Candidate cd;
                            cd.x = (int16_t)(xb + 4 * j + b);
                            cd.y = (int16_t)y;
                            cd.score = (uint16_t)((d[j] >> (8 * b)) & 255u);
                            cd.level = (uint16_t)lvl;
                            h_cands[o] = cd;
                        }
                    }
                    rank++;
                }
            }
        }
    }
}

void launch_fast_score(const PyramidParams& p, hipStream_t s) {
    hipLaunchKernelGGL(fast_score_kernel, dim3(8 * ((p.total_tiles + 7) / 8)), dim3(256), 0, s, p);
}
void launch_fast_low(const PyramidParams& p, hipStream_t s) {
    const int ngroups = (p.total_tiles + 3) / 4;
    hipLaunchKernelGGL(fast_low_kernel, dim3(8 * ((ngroups + 7) / 8)), dim3(256), 0, s, p);
}
void launch_emit(const PyramidParams& p, int32_t* d_rowcount, Candidate* cands, CandidateHeader* hdr_a,
                 CandidateHeader* hdr_b, int cand_capacity, hipStream_t s) {
    hipLaunchKernelGGL(rowcount_kernel, dim3(p.total_rows / kTile), dim3(256), 0, s, p, d_rowcount);
    hipLaunchKernelGGL(emit_kernel, dim3(p.total_rows / kTile), dim3(256), 0, s, p, d_rowcount, cands, hdr_a, hdr_b,
                       cand_capacity);
}

// ------------------------------------------------------------------------------------------------
// describe: orientation + 7x7 Gaussian + steered BRIEF fused, one wave per keypoint.
// The 43x43 source patch (radius 18 samples + 3 blur taps, reflect-101 at the image edge) is staged in LDS
// once; the blur is evaluated only on the 37x37 window BRIEF can reach, so the full-frame blur pass and its
// HBM round trip disappear.  256 tests = 4 ballots of 64 lanes -> the four 64-bit descriptor words.
// ------------------------------------------------------------------------------------------------
__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "brief_pattern.inc"
};
__device__ __forceinline__ int reflect101(int p, int n) {
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p;
}

// deterministic atan2 / sincos (same polynomials as the oracle; only +,*,/,fma -> bit-identical)
__device__ __forceinline__ float det_atan2f(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (mx == 0.f) return 0.f;
    const float a = mn / mx;
    const float s = a * a;
    float p = 0.0028340641874819994f;
    p = __builtin_fmaf(p, s, -0.016005029901862144f);
    p = __builtin_fmaf(p, s, 0.042587608098983765f);
    p = __builtin_fmaf(p, s, -0.07495445758104324f);
    p = __builtin_fmaf(p, s, 0.10636754333972931f);
    p = __builtin_fmaf(p, s, -0.14202570915222168f);
    p = __builtin_fmaf(p, s, 0.19992484152317047f);
    p = __builtin_fmaf(p, s, -0.3333306610584259f);
    p = __builtin_fmaf(p, s, 1.0f);
    float r = a * p;
    if (ay > ax) r = 0x1.921fb6p+0f - r;
    if (x < 0.f) r = 0x1.921fb6p+1f - r;
    if (y < 0.f) r = -r;
    return r;
}

__device__ __forceinline__ void det_sincosf(float a, float& sn, float& cs) {
    const float k = __builtin_rintf(a * 0x1.45f306p-1f);
    float r = __builtin_fmaf(-k, 0x1.921fb6p+0f, a);
    r = __builtin_fmaf(-k, -0x1.777a5cp-25f, r);
    const float s = r * r;
    float ps = 2.716587005124893e-06f;
    ps = __builtin_fmaf(ps, s, -0.0001983911934075877f);
    ps = __builtin_fmaf(ps, s, 0.008333328180015087f);
    ps = __builtin_fmaf(ps, s, -0.1666666716337204f);
    ps = __builtin_fmaf(ps, s, 1.0f);
    const float sinr = r * ps;
    float pc = 2.4371513063670136e-05f;
    pc = __builtin_fmaf(pc, s, -0.001388652715831995f);
    pc = __builtin_fmaf(pc, s, 0.04166661202907562f);
    pc = __builtin_fmaf(pc, s, -0.5f);
    pc = __builtin_fmaf(pc, s, 1.0f);
    const float cosr = pc;
    const int q = ((int)k) & 3;
    sn = (q == 0) ? sinr : (q == 1) ? cosr : (q == 2) ? -sinr : -cosr;
    cs = (q == 0) ? cosr : (q == 1) ? -sinr : (q == 2) ? -cosr : sinr;
}

constexpr int kPatch = 43, kPatchPitch = 48;  // source patch (12 dwords per row: an aligned dword run covers any 43 bytes)
constexpr int kBlur = 37, kBlurPitch = 40;    // blurred window
#ifndef SO_DESC_THREADS
#define SO_DESC_THREADS 64
#endif
constexpr int kDescThreads = SO_DESC_THREADS;  // waves per keypoint x 64 (measured on MI355X, 1000 keypoints: one wave
                                               // 16.2 us, four waves 18.0 us - the barriers and the per-wave set-up cost
                                               // more than the shorter loops save)
constexpr int kDescWaves = kDescThreads / 64;
static_assert(kDescThreads == 64 || kDescThreads == 128 || kDescThreads == 256, "descriptor words are dealt to whole waves");

// radius of the orientation disc at row |v| (ORBextractor.cc umax, HALF_PATCH_SIZE 15): 4 bits per row
__device__ __forceinline__ int orb_umax(int av) { return (int)((0x3689ABCDDEEEFFFFull >> (4 * av)) & 15ull); }

__device__ __forceinline__ void describe_body(const PyramidParams& P, const SelectedKp kp, const int id,
                                              uint8_t* __restrict__ desc, float* __restrict__ angle_out,
                                              uint32_t* patch32, float* rowp, uint8_t* blur, int* s_mom,
                                              uint8_t* __restrict__ desc2 = nullptr,
                                              float* __restrict__ angle2 = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const LevelDesc& L = P.lv[kp.level];
    const int x = kp.x, y = kp.y;
    // this lane's test pairs (one dword each: x0, y0, x1, y1), fetched now - nothing below waits for them until the end
    uint32_t pat[4 / kDescWaves];
#pragma unroll
    for (int q = 0; q < 4 / kDescWaves; q++) pat[q] = reinterpret_cast<const uint32_t*>(c_pattern)[lane + 64 * (wave + kDescWaves * q)];
    constexpr float g0 = 0x1.1f5f62p-4f, g1 = 0x1.0c70fcp-3f, g2 = 0x1.869472p-3f, g3 = 0x1.ba95c0p-3f;
    constexpr float gauss7[7] = {g0, g1, g2, g3, g2, g1, g0};

    // 43 x 43 source patch.  Inside the image (all but keypoints within 21 px of a border) every row is an aligned run
    // of 12 dwords - 516 loads per keypoint instead of 1849 byte loads; `sh` is where the patch's first column sits in it.
    int sh = 0;
    if (x >= 21 && y >= 21 && x + 21 < L.w && y + 21 < L.h) {
        const int ax = (x - 21) & ~3;
        sh = (x - 21) & 3;
        const uint8_t* src = L.img + (size_t)(y - 21) * L.pitch + ax;
        for (int i = tid; i < kPatch * (kPatchPitch / 4); i += kDescThreads) {
            const int r = i / (kPatchPitch / 4), d = i - r * (kPatchPitch / 4);
            patch32[i] = *reinterpret_cast<const uint32_t*>(src + (size_t)r * L.pitch + 4 * d);
        }
    } else {
        uint8_t* pb = reinterpret_cast<uint8_t*>(patch32);
        for (int i = tid; i < kPatch * kPatch; i += kDescThreads) {
            const int r = i / kPatch, c = i - r * kPatch;
            const int gy = reflect101(y - 21 + r, L.h), gx = reflect101(x - 21 + c, L.w);
            pb[r * kPatchPitch + c] = L.img[(size_t)gy * L.pitch + gx];
        }
    }
    __syncthreads();
    const uint8_t* patch = reinterpret_cast<const uint8_t*>(patch32) + sh;

    // intensity centroid over the radius-15 disc (749 px), integer moments
    int m10 = 0, m01 = 0;
    for (int i = tid; i < 31 * 31; i += kDescThreads) {
        const int r = i / 31, c = i - r * 31;
        const int v = r - 15, u = c - 15;
        if (abs(u) <= orb_umax(abs(v))) {
            const int val = patch[(21 + v) * kPatchPitch + 21 + u];
            m10 += u * val;
            m01 += v * val;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m10 += __shfl_xor(m10, off);
        m01 += __shfl_xor(m01, off);
    }
    if (lane == 0) {
        s_mom[2 * wave] = m10;
        s_mom[2 * wave + 1] = m01;
    }

    // separable Gaussian: row pass (float), column pass, round-half-even.  A lane produces FOUR neighbouring outputs from
    // the ten source values they share (row pass: four aligned dwords of the patch row, shifted into place; column
    // pass: ten rows of one column) - 21 instructions per output instead of 43 / 30 with one output per lane and seven
    // byte (or float) reads each; every output is still the same chain of seven multiply-adds in the same order.
    constexpr int kGroups = (kBlur + 3) / 4;  // 10 groups of four along a row / a column
    for (int t = tid; t < kPatch * kGroups; t += kDescThreads) {
        const int r = t / kGroups, g = t - r * kGroups;
        const uint32_t* q = patch32 + r * (kPatchPitch / 4) + g;
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
        const uint32_t a0 = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)sh), a1 = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)sh),
                       a2 = __builtin_amdgcn_alignbyte(d3, d2, (uint32_t)sh);
        const float b[10] = {(float)(a0 & 255u), (float)((a0 >> 8) & 255u), (float)((a0 >> 16) & 255u), (float)(a0 >> 24),
                             (float)(a1 & 255u), (float)((a1 >> 8) & 255u), (float)((a1 >> 16) & 255u), (float)(a1 >> 24),
                             (float)(a2 & 255u), (float)((a2 >> 8) & 255u)};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 7; k++) sum = sum + b[j + k] * gauss7[k];
            if (4 * g + j < kBlur) rowp[r * kBlur + 4 * g + j] = sum;
        }
    }
    __syncthreads();
    if (kDescWaves > 1) {
        m10 = m01 = 0;
#pragma unroll
        for (int w = 0; w < kDescWaves; w++) {
            m10 += s_mom[2 * w];
            m01 += s_mom[2 * w + 1];
        }
    }
    float kp_dir = det_atan2f((float)m01, (float)m10);
    kp_dir += (float)(kp_dir < 0) * 0x1.921fb6p+2f;
    kp_dir *= 0x1.ca5dcp+5f;
    for (int t = tid; t < kGroups * kBlur; t += kDescThreads) {
        const int g = t / kBlur, c = t - g * kBlur;
        float v[10];
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = rowp[min(4 * g + k, kPatch - 1) * kBlur + c];  // (rows past the patch feed no valid output)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 7; k++) sum = sum + v[j + k] * gauss7[k];
            const int o = (int)__builtin_rintf(sum);
            if (4 * g + j < kBlur) blur[(4 * g + j) * kBlurPitch + c] = (uint8_t)min(max(o, 0), 255);
        }
    }
    __syncthreads();

    float a, b;  // a = cos, b = sin  (Orb_gpu.cu:77-79)
    det_sincosf(kp_dir * 0x1.1df46ap-6f, b, a);
#pragma unroll
    for (int q = 0; q < 4 / kDescWaves; q++) {
        const int wq = wave + kDescWaves * q;  // descriptor bits 64 wq .. 64 wq + 63
        const float p0x = (float)(int8_t)(pat[q] & 255u), p0y = (float)(int8_t)((pat[q] >> 8) & 255u);
        const float p1x = (float)(int8_t)((pat[q] >> 16) & 255u), p1y = (float)(int8_t)(pat[q] >> 24);
        const int r0 = (int)__builtin_rintf(p0x * b + p0y * a), c0 = (int)__builtin_rintf(p0x * a - p0y * b);
        const int r1 = (int)__builtin_rintf(p1x * b + p1y * a), c1 = (int)__builtin_rintf(p1x * a - p1y * b);
        const int t0 = blur[(18 + r0) * kBlurPitch + 18 + c0];
        const int t1 = blur[(18 + r1) * kBlurPitch + 18 + c1];
        const unsigned long long word = __ballot(t0 < t1);
        if (lane == 0) {
            reinterpret_cast<unsigned long long*>(desc + (size_t)id * 32)[wq] = word;
            if (desc2) reinterpret_cast<unsigned long long*>(desc2 + (size_t)id * 32)[wq] = word;  // HBM-resident copy (dframe.cpp)
        }
    }
    if (tid == 0) {
        angle_out[id] = kp_dir;
        if (angle2) angle2[id] = kp_dir;
    }
}

__global__ __launch_bounds__(kDescThreads) void describe_kernel(PyramidParams P, const SelectedKp* __restrict__ sel, int n,
                                                                 uint8_t* __restrict__ desc, float* __restrict__ angle_out) {
    __shared__ uint32_t patch[kPatch * (kPatchPitch / 4) + 4];
    __shared__ float rowp[kPatch * kBlur];
    __shared__ uint8_t blur[kBlur * kBlurPitch];
    __shared__ int s_mom[8];
    const int id = blockIdx.x;
    if (id >= n) return;
    describe_body(P, sel[id], id, desc, angle_out, patch, rowp, blur, s_mom);
}

// Same, fed by the device quadtree: block b finds its (level, k) from the per-level survivor counts, describes
// qt_sel[level][k] and writes descriptor / angle / keypoint record at the compact output index, straight into
// host-mapped memory (the caller's copy) and into HBM (`dev`: what the device-resident frame reads).  Launched
// with the capacity (rounded up to a multiple of eight) as grid; surplus blocks exit.
__device__ __forceinline__ void describe_qt_body(const PyramidParams& P, const SelectedKp* __restrict__ qt_sel,
                                                 const int32_t* __restrict__ qt_count, int qt_stride, uint8_t* __restrict__ desc,
                                                 float* __restrict__ angle_out, SelectedKp* __restrict__ meta_out,
                                                 int32_t* __restrict__ total_out, const DescribeDeviceOut& dev, int capacity) {
    __shared__ uint32_t patch[kPatch * (kPatchPitch / 4) + 4];
    __shared__ float rowp[kPatch * kBlur];
    __shared__ uint8_t blur[kBlur * kBlurPitch];
    __shared__ int s_mom[8];
    int cnt[kMaxLevels], total = 0;
#pragma unroll
    for (int l = 0; l < kMaxLevels; l++) {
        cnt[l] = l < P.nlevels ? qt_count[l] : 0;
        total += cnt[l];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *total_out = total;
        if (dev.total) *dev.total = total;
    }
    // Which keypoint this workgroup describes: workgroups are dealt to the eight XCDs round-robin (blockIdx.x % 8) and every
    // XCD has its own L2, so with id = blockIdx.x each XCD pulled patches from all over all eight levels - the 1.2 MB
    // pyramid crossed the fabric ~4.5 times per frame (FETCH_SIZE 5.3 MB, rounds 2-4).  XCD x takes the x-th eighth of the
    // keypoints instead: ids are ordered by level and, inside a level, by quadtree node, so an eighth is a compact region
    // of one or two levels (measured: DESIGN.md 3).  The output slot is the keypoint's id, as before.
    const int n_desc = min(total, capacity);  // (the output arrays hold `capacity` records; the quadtree's bound is below it)
    const int chunk = (n_desc + 7) >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    int id = xcd * chunk + j, lvl = -1, base = 0;
    if (j >= chunk || id >= n_desc) return;
#pragma unroll
    for (int l = 0; l < kMaxLevels; l++) {
        if (lvl < 0 && id < base + cnt[l]) lvl = l;
        if (lvl < 0) base += cnt[l];
    }
    if (lvl < 0) return;
    const SelectedKp kp = qt_sel[(size_t)lvl * qt_stride + (id - base)];
    if (threadIdx.x == 0) {
        meta_out[id] = kp;
        if (dev.meta) dev.meta[id] = kp;
    }
    describe_body(P, kp, id, desc, angle_out, patch, rowp, blur, s_mom, dev.desc, dev.angle);
}

__global__ __launch_bounds__(kDescThreads) void describe_qt_kernel(PyramidParams P, const SelectedKp* __restrict__ qt_sel,
                                                                    const int32_t* __restrict__ qt_count, int qt_stride,
                                                                    uint8_t* __restrict__ desc, float* __restrict__ angle_out,
                                                                    SelectedKp* __restrict__ meta_out,
                                                                    int32_t* __restrict__ total_out, DescribeDeviceOut dev, int capacity) {
    describe_qt_body(P, qt_sel, qt_count, qt_stride, desc, angle_out, meta_out, total_out, dev, capacity);
}
__global__ __launch_bounds__(kDescThreads) void describe_qt_batch_kernel(const ExtractBatchMember* __restrict__ M, int capacity) {
    const ExtractBatchMember& m = M[blockIdx.y];
    if (m.skip) return;
    describe_qt_body(m.P, m.qt_sel, m.qt_count, m.qt.sel_stride, m.out.desc, m.out.angle, m.out.meta, m.out.total, m.out.dev, capacity);
}


void launch_describe_qt(const PyramidParams& p, const SelectedKp* d_qt_sel, const int32_t* d_qt_count, int qt_stride,
                        int capacity, uint8_t* desc, float* angle, SelectedKp* meta, int32_t* total,
                        const DescribeDeviceOut& dev, hipStream_t s) {
    // (a multiple of eight workgroups: describe_qt_body deals the ids to the XCDs in eighths)
    hipLaunchKernelGGL(describe_qt_kernel, dim3(8 * ((capacity + 7) / 8)), dim3(kDescThreads), 0, s, p, d_qt_sel, d_qt_count, qt_stride, desc,
                       angle, meta, total, dev, capacity);
}

void launch_extract_batch(const ExtractBatchMember* d_members, const ExtractBatchMember& first, int n, const uint8_t* const* d_srcs,
                          const int32_t* d_sel, int w, int h, bool rows16, int capacity, hipStream_t s) {
    const PyramidParams& p = first.P;
    if (rows16)
        hipLaunchKernelGGL(ingest16_batch_kernel, dim3((w / 16 + 63) / 64, (h + 3) / 4, n), dim3(64, 4), 0, s, d_srcs, d_sel, d_members, w / 16, h);
    else
        hipLaunchKernelGGL(ingest_batch_kernel, dim3((w + 255) / 256, (h + 3) / 4, n), dim3(64, 4), 0, s, d_srcs, d_sel, d_members, w, h);
    for (int l = 1; l < p.nlevels; l++) {
        const LevelDesc& S = p.lv[l - 1];
        const LevelDesc& D = p.lv[l];
        const float fx = (float)(1.0 / ((double)D.w / (double)S.w)), fy = (float)(1.0 / ((double)D.h / (double)S.h));
        hipLaunchKernelGGL(resize_batch_kernel, dim3((D.w + 255) / 256, (D.h + 3) / 4, n), dim3(64, 4), 0, s, d_members, l, fx, fy);
    }
    hipLaunchKernelGGL(fast_score_batch_kernel, dim3(8 * ((p.total_tiles + 7) / 8), n), dim3(256), 0, s, d_members);
    const int ngroups = (p.total_tiles + 3) / 4;
    hipLaunchKernelGGL(fast_low_batch_kernel, dim3(8 * ((ngroups + 7) / 8), n), dim3(256), 0, s, d_members);
    launch_quadtree_batch(d_members, n, p.nlevels, s);
    hipLaunchKernelGGL(describe_qt_batch_kernel, dim3(8 * ((capacity + 7) / 8), n), dim3(kDescThreads), 0, s, d_members, capacity);
}

void launch_describe(const PyramidParams& p, const SelectedKp* d_sel, int n, uint8_t* d_desc, float* d_angle,
                     hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(describe_kernel, dim3(n), dim3(kDescThreads), 0, s, p, d_sel, n, d_desc, d_angle);
}

}  // namespace so

// ------------------------------------------------------------------------------------------------
// PMC calibration helper (tools/pmc_calibrate.py): streams `n_dwords` aligned dwords with the same access
// shape as fast_score_kernel's tile staging (one 4-byte load per lane, consecutive lanes -> consecutive
// dwords) and stores one word per workgroup, so FETCH_SIZE / WRITE_SIZE can be compared with known byte counts.
// ------------------------------------------------------------------------------------------------
namespace so {
__global__ __launch_bounds__(256) void calib_read_dwords_kernel(const uint32_t* __restrict__ src, size_t n_dwords,
                                                                 uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_dwords; i += (size_t)gridDim.x * 256) acc ^= src[i];
    acc ^= __shfl_xor((int)acc, 32);
    if ((threadIdx.x & 63) == 0) atomicXor(&sink[blockIdx.x & 1023], acc);
}
}  // namespace so

extern "C" int so_debug_stream_read(const void* d_src, unsigned long long n_bytes, void* d_sink_4k) {
    const size_t n = (size_t)(n_bytes / 4);
    hipLaunchKernelGGL(so::calib_read_dwords_kernel, dim3(4096), dim3(256), 0, 0, (const uint32_t*)d_src, n,
                       (uint32_t*)d_sink_4k);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 3;
}
