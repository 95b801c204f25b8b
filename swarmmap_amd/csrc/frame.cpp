// frame.cpp — C ABI (include/swarmorb.h) of the Frame post-processing between extractor and matcher.  Inputs are
// packed into one pinned staging block and moved with one copy; the kernels write their results straight into
// host-mapped memory; one launch and one sync per call.  Runs on the calling thread's matcher stream.
#include <hip/hip_runtime.h>

#include <cstring>

#include "frame_device.h"
#include "so_common.h"

using namespace so;

struct so_frame_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void* h_in = nullptr;   // pinned
    size_t h_in_cap = 0;
    void* d_in = nullptr;
    size_t d_in_cap = 0;
    void* h_out = nullptr;  // host-mapped
    void* h_out_dev = nullptr;
    size_t h_out_cap = 0;
};

namespace {

int ensure(so_frame_ctx* f, size_t in_bytes, size_t out_bytes) {
    if (in_bytes > f->h_in_cap) {
        if (f->h_in) SO_HIP(hipHostFree(f->h_in));
        if (f->d_in) SO_HIP(hipFree(f->d_in));
        f->h_in = f->d_in = nullptr;
        f->h_in_cap = f->d_in_cap = 0;
        const size_t want = in_bytes + in_bytes / 2 + 4096;
        SO_HIP(hipHostMalloc(&f->h_in, want, hipHostMallocDefault));
        SO_HIP(hipMalloc(&f->d_in, want));
        f->h_in_cap = f->d_in_cap = want;
    }
    if (out_bytes > f->h_out_cap) {
        if (f->h_out) SO_HIP(hipHostFree(f->h_out));
        f->h_out = f->h_out_dev = nullptr;
        f->h_out_cap = 0;
        const size_t want = out_bytes + out_bytes / 2 + 4096;
        SO_HIP(hipHostMalloc(&f->h_out, want, hipHostMallocMapped));
        SO_HIP(hipHostGetDevicePointer(&f->h_out_dev, f->h_out, 0));
        f->h_out_cap = want;
    }
    return SO_OK;
}

FrameCam to_cam(const so_camera* c) { return FrameCam{c->fx, c->fy, c->cx, c->cy, c->k1, c->k2, c->p1, c->p2, c->k3}; }

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int so_frame_create(int device, so_frame_ctx** out) {
    if (!out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(device));
    so_frame_ctx* f = new so_frame_ctx();
    f->device = device;
    const hipError_t e = tracking_stream(device, 1, &f->stream);
    if (e != hipSuccess) {
        delete f;
        return hip_fail(e, "frame init", __FILE__, __LINE__);
    }
    *out = f;
    return SO_OK;
}

void so_frame_destroy(so_frame_ctx* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    if (f->h_in) (void)hipHostFree(f->h_in);
    if (f->d_in) (void)hipFree(f->d_in);
    if (f->h_out) (void)hipHostFree(f->h_out);
    delete f;
}

// Frame ctor steps between ExtractORB and the first matcher call (Frame.cc:183-192, 230-274): UndistortKeyPoints,
// ComputeImageBounds (when compute_bounds), AssignFeaturesToGrid
int so_frame_prepare(so_frame_ctx* f, const so_camera* cam, int32_t width, int32_t height, int compute_bounds,
                     int32_t n, const float* xy, float* xy_un, float* bounds4, int32_t* cell_of, int32_t* cell_start,
                     int32_t* cell_items, int32_t* n_inside) {
    if (!f || !cam || n < 0 || !bounds4 || (n > 0 && (!xy || !xy_un))) return SO_ERR_INVALID_ARG;
    if (n > kFrameMaxKeypoints) {
        last_error_ref() = "so_frame_prepare handles at most 16384 keypoints";
        return SO_ERR_CAPACITY;
    }
    const bool grid = cell_of && cell_start && cell_items && n_inside;
    SO_HIP(hipSetDevice(f->device));
    constexpr int ncell = kFrameGridCols * kFrameGridRows;
    const size_t in_bytes = up256(sizeof(float) * 2 * (size_t)n) + 256;
    const size_t o_un = 0, o_b = up256(sizeof(float) * 2 * (size_t)n), o_cell = o_b + 256,
                 o_start = o_cell + up256(sizeof(int32_t) * (size_t)n), o_items = o_start + up256(sizeof(int32_t) * (ncell + 1)),
                 o_inside = o_items + up256(sizeof(int32_t) * (size_t)n), out_bytes = o_inside + 256;
    int rc = ensure(f, in_bytes, out_bytes);
    if (rc) return rc;
    if (n > 0) memcpy(f->h_in, xy, sizeof(float) * 2 * (size_t)n);
    uint8_t* ho = (uint8_t*)f->h_out;
    uint8_t* od = (uint8_t*)f->h_out_dev;
    if (!compute_bounds) memcpy(ho + o_b, bounds4, 16);
    if (n > 0) SO_HIP(hipMemcpyAsync(f->d_in, f->h_in, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, f->stream));
    FramePrepareArgs a{};
    a.cam = to_cam(cam);
    a.width = width;
    a.height = height;
    a.n = n;
    a.do_bounds = compute_bounds ? 1 : 0;
    a.do_undistort = 1;
    a.do_grid = grid ? 1 : 0;
    a.xy = (const float*)f->d_in;
    a.xy_un = (float*)(od + o_un);
    a.bounds = (float*)(od + o_b);
    a.cell_of = (int32_t*)(od + o_cell);
    a.cell_start = (int32_t*)(od + o_start);
    a.cell_items = (int32_t*)(od + o_items);
    a.n_inside = (int32_t*)(od + o_inside);
    launch_frame_prepare(a, f->stream);
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(f->stream));
    if (n > 0) memcpy(xy_un, ho + o_un, sizeof(float) * 2 * (size_t)n);
    memcpy(bounds4, ho + o_b, 16);
    if (grid) {
        if (n > 0) memcpy(cell_of, ho + o_cell, sizeof(int32_t) * (size_t)n);
        memcpy(cell_start, ho + o_start, sizeof(int32_t) * (ncell + 1));
        *n_inside = *(const int32_t*)(ho + o_inside);
        if (*n_inside > 0) memcpy(cell_items, ho + o_items, sizeof(int32_t) * (size_t)*n_inside);
    }
    return SO_OK;
}

// Frame::isInFrustum over a batch of map points (Tracking::SearchLocalPoints, Tracking.cc:985-996)
int so_frame_is_in_frustum(so_frame_ctx* f, const so_camera* cam, const float* bounds4, const float* Tcw12, int32_t n,
                           const float* Xw, const float* normal, const float* max_dist, const float* min_dist,
                           float viewing_cos_limit, float log_scale_factor, int32_t n_scale_levels, uint8_t* in_view,
                           float* proj_x, float* proj_y, float* view_cos, int32_t* pred_level) {
    if (!f || !cam || !bounds4 || !Tcw12 || n < 0) return SO_ERR_INVALID_ARG;
    if (n == 0) return SO_OK;
    if (!Xw || !normal || !max_dist || !min_dist || !in_view || !proj_x || !proj_y || !view_cos || !pred_level)
        return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(f->device));
    const size_t sn = (size_t)n;
    const size_t i_X = 0, i_N = up256(12 * sn), i_max = i_N + up256(12 * sn), i_min = i_max + up256(4 * sn),
                 in_bytes = i_min + up256(4 * sn);
    // results: the four float / int arrays are pre-loaded with the caller's values (isInFrustum leaves the track
    // fields of a rejected point untouched)
    const size_t o_view = 0, o_px = up256(sn), o_py = o_px + up256(4 * sn), o_vc = o_py + up256(4 * sn),
                 o_lvl = o_vc + up256(4 * sn), out_bytes = o_lvl + up256(4 * sn);
    int rc = ensure(f, in_bytes, out_bytes);
    if (rc) return rc;
    uint8_t* hi = (uint8_t*)f->h_in;
    memcpy(hi + i_X, Xw, 12 * sn);
    memcpy(hi + i_N, normal, 12 * sn);
    memcpy(hi + i_max, max_dist, 4 * sn);
    memcpy(hi + i_min, min_dist, 4 * sn);
    uint8_t* ho = (uint8_t*)f->h_out;
    uint8_t* od = (uint8_t*)f->h_out_dev;
    memcpy(ho + o_px, proj_x, 4 * sn);
    memcpy(ho + o_py, proj_y, 4 * sn);
    memcpy(ho + o_vc, view_cos, 4 * sn);
    memcpy(ho + o_lvl, pred_level, 4 * sn);
    SO_HIP(hipMemcpyAsync(f->d_in, f->h_in, in_bytes, hipMemcpyHostToDevice, f->stream));
    FrameFrustumArgs a;
    a.cam = to_cam(cam);
    memcpy(a.bounds, bounds4, 16);
    memcpy(a.Tcw, Tcw12, 48);
    a.n = n;
    const uint8_t* di = (const uint8_t*)f->d_in;
    a.Xw = (const float*)(di + i_X);
    a.normal = (const float*)(di + i_N);
    a.max_dist = (const float*)(di + i_max);
    a.min_dist = (const float*)(di + i_min);
    a.viewing_cos_limit = viewing_cos_limit;
    a.log_scale_factor = log_scale_factor;
    a.n_scale_levels = n_scale_levels;
    a.in_view = od + o_view;
    a.proj_x = (float*)(od + o_px);
    a.proj_y = (float*)(od + o_py);
    a.view_cos = (float*)(od + o_vc);
    a.pred_level = (int32_t*)(od + o_lvl);
    launch_frame_frustum(a, f->stream);
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(f->stream));
    memcpy(in_view, ho + o_view, sn);
    memcpy(proj_x, ho + o_px, 4 * sn);
    memcpy(proj_y, ho + o_py, 4 * sn);
    memcpy(view_cos, ho + o_vc, 4 * sn);
    memcpy(pred_level, ho + o_lvl, 4 * sn);
    return SO_OK;
}

}  // extern "C"
