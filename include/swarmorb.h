/*
 * swarmorb.h — C ABI of libswarmorb.so, the MI355X (gfx950) implementation of SwarmMap's per-frame
 * ORB front-end, Hamming matcher and bundle-adjustment hot path.
 *
 * Plain C types only (pointers + sizes); no torch / OpenCV / Eigen types cross this boundary.
 * Each entry point cites the reference interface (file:line under /root/reference) it replaces.
 * The reference-side bindings a maintainer would add are shown in INTEGRATION.md; C++ adapter
 * classes with the reference's own signatures live in swarmmap_amd/host/.
 *
 * Threading: a handle is NOT re-entrant (like the reference's extractor, which owns CUDA streams and
 * fixed-size device buffers, code/src/cuda/Fast_gpu.cu:342-369); different handles may be used
 * concurrently from different threads (N agents in one process).
 */
#ifndef SWARMORB_H
#define SWARMORB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the reference exit()s on CUDA errors via checkCudaErrors; we return) ---- */
enum {
    SO_OK = 0,
    SO_ERR_INVALID_ARG = 1,
    SO_ERR_NO_DEVICE = 2,    /* no HIP device / HIP runtime error at init */
    SO_ERR_HIP = 3,          /* HIP runtime error during a call (see so_last_error) */
    SO_ERR_CAPACITY = 4,     /* caller buffer too small */
    SO_ERR_SIZE_CHANGED = 5, /* image size differs from the first frame (reference: "WILL BREAK") */
    SO_ERR_NUMERIC = 6       /* BA: linear solve failed in every LM trial */
};
const char* so_status_string(int status);
/* last HIP error text on the calling thread ("" if none) */
const char* so_last_error(void);
/* number of visible HIP devices (0 when there is no GPU); never throws */
int so_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * ORB extractor  — replaces ORB_SLAM2::ORBextractor (code/include/ORBextractor.h:49-127,
 * code/src/ORBextractor.cc:340-855) incl. cuda::GpuFast / IC_Angle / GpuOrb
 * (code/src/cuda/Fast_gpu.cu, Orb_gpu.cu) and the cv::cuda resize/border/blur calls.
 * ---------------------------------------------------------------------------------------------- */
/* identical 28-byte layout to cv::KeyPoint {pt.x, pt.y, size, angle, response, octave, class_id} */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} so_keypoint;

typedef struct {
    int32_t nfeatures;   /* ORBextractor ctor args, code/include/ORBextractor.h:54 */
    float scale_factor;
    int32_t nlevels;     /* <= SO_MAX_LEVELS */
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t device;      /* HIP device ordinal (one agent per GPU) */
} so_extractor_config;

#define SO_MAX_LEVELS 8
#define SO_FAST_CAP 10000 /* GpuFast maxKeypoints, code/include/cuda/Fast.hpp:32 */

typedef struct so_extractor so_extractor;

/* ORBextractor::ORBextractor (code/src/ORBextractor.cc:340-405) */
int so_extractor_create(const so_extractor_config* cfg, so_extractor** out);
void so_extractor_destroy(so_extractor* ex);

/* Output capacity the caller must provide: nfeatures + 3*nlevels (the quadtree may overshoot its
 * per-level quota by up to 3, code/src/ORBextractor.cc:656-661). */
int so_extractor_capacity(const so_extractor* ex);

/* ORBextractor::operator() (code/src/ORBextractor.cc:746-819): CV_8UC1 host image in, keypoints
 * (level-0 pixel units, levels concatenated 0..n-1) and 32-byte descriptors out.  An empty image
 * (NULL / w<=0 / h<=0) returns SO_OK with *n_out = 0 and outputs untouched. */
int so_extractor_run(so_extractor* ex, const uint8_t* image, int width, int height, int stride,
                     so_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out);
/* Same, image already resident in device memory (HBM) on the extractor's device. */
int so_extractor_run_device(so_extractor* ex, const uint8_t* d_image, int width, int height, int stride,
                            so_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out);

/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares / GetInverseScaleSigmaSquares
 * (code/include/ORBextractor.h:64-87) + mnFeaturesPerLevel; each array has nlevels entries. */
int so_extractor_tables(const so_extractor* ex, float* scale, float* inv_scale, float* sigma2,
                        float* inv_sigma2, int32_t* features_per_level);

/* Stage outputs of the LAST run, for stage-wise parity tests (not used by the SLAM threads):
 *  - level image `level` (un-blurred), tightly packed w*h bytes (mvImagePyramid, ORBextractor.h:90)
 *  - FAST candidates of `level` before the quadtree (ROI-relative x,y; integer score), raster order
 *    (GpuFast::joinDetectAsync output, code/src/cuda/Fast_gpu.cu:379-388). */
int so_extractor_level_size(const so_extractor* ex, int level, int* w, int* h);
int so_extractor_get_level(so_extractor* ex, int level, uint8_t* out, int out_bytes);
int so_extractor_get_candidates(so_extractor* ex, int level, int16_t* xs, int16_t* ys, uint8_t* scores,
                                int capacity, int* n_out);

/* Per-stage GPU timing with HIP events on the extractor's own stream (off by default).
 * Stages: 0 upload+pyramid, 1 FAST score+NMS (high threshold), 2 FAST low-threshold pass,
 * 3 candidate compaction, 4 angle+blur+rBRIEF, 5 whole run wall time on the host (ms). */
#define SO_EXTRACTOR_N_STAGES 6
int so_extractor_set_profiling(so_extractor* ex, int enabled);
int so_extractor_get_profile(so_extractor* ex, float* ms_per_stage /* [SO_EXTRACTOR_N_STAGES] */);

#ifdef __cplusplus
}
#endif
#endif /* SWARMORB_H */
