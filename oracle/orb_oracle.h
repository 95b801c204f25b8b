/*
 * orb_oracle.h — CPU restatement (parity oracle) of SwarmMap's ORB front-end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the product
 * (swarmmap_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: the reference (/root/reference) cannot be compiled in this image (needs CUDA,
 * OpenCV-CUDA, Eigen) and ships no tests or golden vectors.  The integer stages are pinned by
 * known-answer tests derived from the reference's own constants (tests/test_oracle_kat.py):
 *   - FAST-9/16 predicate == the reference's 8129-byte lookup table (code/src/cuda/Fast_gpu.cu:58)
 *   - rBRIEF pattern sha256 (code/src/ORBextractor.cc:80-338), umax table, level sizes,
 *     features-per-level (code/src/ORBextractor.cc:340-405)
 * The arithmetic of cv::cuda::resize / createGaussianFilter / fast-math atan2f,sinf,cosf lives in
 * un-vendored third-party code: for those stages this oracle DEFINES the convention (documented at
 * each function) and parity there is "unpinned" (SURVEY.md 8c).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Convention switch (tools/convention_sensitivity.py only; 0 everywhere else): each bit swaps ONE of the conventions this oracle
 * defines for arithmetic the reference keeps in un-vendored libraries for its plausible alternative. */
enum {
    ORC_CONV_RESIZE_HALF_PIXEL = 1,  /* bilinear resize samples at (dst + 0.5) / f - 0.5 instead of dst / f */
    ORC_CONV_GAUSS_FIXED8 = 2,       /* 7x7 Gaussian in 8-bit fixed point instead of float */
    ORC_CONV_TRIG_LIBM = 4,          /* libm atan2f / sinf / cosf instead of the shared polynomials */
    ORC_CONV_QUAT_LARGEST = 8,       /* Quaterniond(R): Shepperd's largest-of-four pivot instead of Eigen's trace-first branches */
    ORC_CONV_LDLT_REVERSED = 16,     /* reduced system: square-root-free LDL^T in reversed elimination order instead of Cholesky */
    ORC_CONV_UNDISTORT_20 = 32       /* cv::undistortPoints: 20 fixed-point iterations instead of 5 */
};
void orc_set_convention(int flags);
int orc_get_convention(void);

/* Same 28-byte layout as cv::KeyPoint. */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
} orc_config;

#define ORC_MAX_LEVELS 16
#define ORC_FAST_CAP 10000 /* GpuFast maxKeypoints default, include/cuda/Fast.hpp:32 */

typedef struct {
    float scale[ORC_MAX_LEVELS];
    float inv_scale[ORC_MAX_LEVELS];
    float sigma2[ORC_MAX_LEVELS];
    float inv_sigma2[ORC_MAX_LEVELS];
    int32_t features_per_level[ORC_MAX_LEVELS];
    int32_t umax[16];
} orc_tables;

/* ORBextractor::ORBextractor, code/src/ORBextractor.cc:340-405 */
void orc_make_tables(const orc_config* cfg, orc_tables* t);
/* Level size, code/src/ORBextractor.cc:824-825,840-841 */
void orc_level_size(int w, int h, float inv_scale, int* lw, int* lh);

/* E1: INTER_LINEAR resize in the OpenCV-CUDA convention (src = dst * scale, no half-pixel shift). */
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                       int dstride);
/* E1: BORDER_REFLECT_101 copyMakeBorder (19 px in the reference). */
void orc_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int border,
                           int dstride);
/* E6: 7x7 sigma=2 separable Gaussian, BORDER_REFLECT_101, float accumulate, round-half-even. */
void orc_gaussian7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);

/* E2: FAST predicate / score on a 16-pixel ring (bit k layout of SURVEY.md A.2). */
int orc_fast_is_corner_masks(int mask_dark, int mask_bright); /* run >= 9 test */
int orc_fast_table_lookup(const uint8_t* table8129, int mask); /* reference's c_table predicate */
int orc_fast_score(const uint8_t* img, int stride, int x, int y, int th); /* 0 if not a corner at th */

/* E2+E3: deterministic FAST+NMS over the ROI = level[16:h-16,16:w-16]; ROI-relative coordinates,
 * raster (y,x) order, truncated to cap.  Returns count. */
int orc_fast_detect(const uint8_t* level, int w, int h, int stride, int th_high, int th_low,
                    int16_t* xs, int16_t* ys, uint8_t* scores, int cap);

/* E4: DistributeOctTree; inputs ROI-relative; writes indices of the kept candidates in output order. */
int orc_distribute_octree(const int16_t* xs, const int16_t* ys, const uint8_t* scores, int n, int roi_w,
                          int roi_h, int n_target, int32_t* out_idx, int out_cap);

/* E5: intensity-centroid angle in degrees [0,360) at level coords (x,y) of the un-blurred level. */
float orc_ic_angle(const uint8_t* level, int stride, int x, int y, const int32_t* umax);
/* E7: 32-byte steered BRIEF from the blurred level. */
void orc_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32);

/* deterministic float helpers (the build's definition of atan2f / sincosf; |err| < 2e-7 rad) */
float orc_atan2f(float y, float x);
void orc_sincosf_deg(float deg, float* s, float* c);

/* E8: the whole operator().  kps/desc capacity must be >= nfeatures + 2*nlevels (octree overshoot). */
int orc_extract(const orc_config* cfg, const uint8_t* img, int w, int h, int stride, orc_keypoint* kps,
                uint8_t* desc, int cap);

/* Like orc_extract but also returns the per-level FAST candidates (before the octree) for stage-wise
 * parity tests: cand_* hold level-concatenated arrays, cand_count[l] per level. */
int orc_extract_debug(const orc_config* cfg, const uint8_t* img, int w, int h, int stride,
                      orc_keypoint* kps, uint8_t* desc, int cap, int16_t* cand_x, int16_t* cand_y,
                      uint8_t* cand_score, int32_t* cand_count, int cand_cap_per_level,
                      uint8_t* pyr_out /* concatenated un-blurred levels, tight stride, may be NULL */);

#ifdef __cplusplus
}
#endif
#endif
