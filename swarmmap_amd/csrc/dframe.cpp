// dframe.cpp — C ABI of the device-resident Frame and of the device-resident map-point table (include/swarmorb.h).
//
// so_dframe = what ORB_SLAM2::Frame's constructor produces (code/src/Frame.cc:218-275: ExtractORB ->
// UndistortKeyPoints -> ComputeImageBounds -> AssignFeaturesToGrid), kept in HBM: the extractor's frame and ONE more
// kernel on the same stream turn the image into undistorted keypoints, the 64x48 grid and the matcher's candidate
// layout without a host round trip; the host receives mirrors (keypoints, descriptors, undistorted positions)
// through host-mapped memory for the parts of SLAM that stay on the CPU.  Two frames alternate per agent: frame t+1
// is extracted into one while the matcher reads the other.
// so_map = the MapPoint fields the per-frame operators read (mWorldPos, mNormalVector, mfMaxDistance, mfMinDistance,
// mDescriptor; code/include/MapPoint.h), indexed by a slot the caller assigns; written at keyframe rate, read by
// the tracking searches (matcher.cpp) every frame.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "dframe_internal.h"
#include "extractor_internal.h"
#include "so_common.h"

using namespace so;

namespace {

size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

int allocate(so_dframe* f, int capacity) {
    constexpr int ncell = kFrameGridCols * kFrameGridRows;
    const size_t c = (size_t)capacity;
    size_t o = 0;
    const size_t o_xy = o; o += up256(8 * c);
    const size_t o_oct = o; o += up256(c);
    const size_t o_desc = o; o += up256(32 * c);
    const size_t o_cs = o; o += up256(4 * (ncell + 1));
    const size_t o_ci = o; o += up256(4 * c);
    const size_t o_sxy = o; o += up256(8 * c);
    const size_t o_soct = o; o += up256(c);
    const size_t o_sdesc = o; o += up256(32 * c);
    const size_t o_col = o; o += up256(4 * (kFrameGridCols + 1));
    const size_t o_ni = o; o += 256;
    const size_t o_b = o; o += 256;
    const size_t o_ang = o; o += up256(4 * c);
    SO_HIP(hipMalloc((void**)&f->d_block, o));
    SO_HIP(so::memset_sync(f->d_block, 0, o));
    uint8_t* d = f->d_block;
    f->d_xy_un = (float2*)(d + o_xy);
    f->d_octave = (int8_t*)(d + o_oct);
    f->d_desc = d + o_desc;
    f->d_cell_start = (int32_t*)(d + o_cs);
    f->d_cell_items = (int32_t*)(d + o_ci);
    f->d_s_xy = (float2*)(d + o_sxy);
    f->d_s_octave = (int8_t*)(d + o_soct);
    f->d_s_desc = (uint4*)(d + o_sdesc);
    f->d_col_start = (int32_t*)(d + o_col);
    f->d_n_inside = (int32_t*)(d + o_ni);
    f->d_bounds = (float*)(d + o_b);
    f->d_angle = (float*)(d + o_ang);
    const size_t h_xy = 0, h_perm = up256(8 * c), h_hdr = h_perm + up256(4 * c), h_total = h_hdr + 256;
    SO_HIP(hipHostMalloc((void**)&f->h_block, h_total, hipHostMallocMapped));
    SO_HIP(hipHostGetDevicePointer((void**)&f->h_block_dev, f->h_block, 0));
    memset(f->h_block, 0, h_total);
    f->h_xy_un = (float*)(f->h_block + h_xy);
    f->h_perm = (int32_t*)(f->h_block + h_perm);
    f->h_header = (int32_t*)(f->h_block + h_hdr);
    f->capacity = capacity;
    f->octave.resize(c);
    f->angle.resize(c);
    f->allocated = true;
    return SO_OK;
}

int submit_body(so_dframe* f, const uint8_t* image, bool on_device, int w, int h, int stride);

// An error behind the extractor's own submission (capacity, allocation, the prepare launch) must not leave the handle
// with a frame "in flight" that nothing will ever collect: the extractor's job is drained and the handle is free again.
int submit_impl(so_dframe* f, const uint8_t* image, bool on_device, int w, int h, int stride) {
    const int rc = submit_body(f, image, on_device, w, h, stride);
    if (rc != SO_OK && f && f->in_flight) {
        const std::string why = last_error_ref();
        int n = 0;
        const int cap = so_extractor_capacity(f->ex);
        std::vector<so_keypoint> kp((size_t)(cap > 0 ? cap : 1));
        std::vector<uint8_t> de((size_t)(cap > 0 ? cap : 1) * 32);
        (void)so_extractor_collect(f->ex, kp.data(), de.data(), cap, &n);  // waits for the frame and frees the extractor
        f->in_flight = false;
        f->launched = false;
        last_error_ref() = why;
    }
    return rc;
}

// Sizes the handle on its first frame and fills in the Frame constructor's launch (Frame.cc:230-274 as one kernel behind
// the extractor's frame, same stream, no host sync in between) for a w x h image.
int prepare_args(so_dframe* f, int w, int h, FramePrepareArgs* out) {
    ExtractorDeviceView V;
    int rc;
    if ((rc = extractor_device_view(f->ex, &V))) return rc;
    SO_HIP(hipSetDevice(V.device));
    if (!f->allocated) {
        if (V.capacity > kFrameMaxKeypoints) {
            last_error_ref() = "so_dframe handles at most 16384 keypoints per frame";
            return SO_ERR_CAPACITY;
        }
        f->device = V.device;
        f->nlevels = V.nlevels;
        for (int l = 0; l < 8; l++) f->scale[l] = V.scale[l];
        if ((rc = allocate(f, V.capacity))) return rc;
    }
    FramePrepareArgs a{};
    a.cam = f->cam;
    a.width = w;
    a.height = h;
    a.n = f->capacity;
    // ComputeImageBounds depends on the camera only (the reference runs it once, Frame.cc:236-247 mbInitialComputations):
    // after this handle's first frame the kernel gets the values instead of undistorting the four corners again - four
    // lanes walking five Newton steps in FP64 in front of everything else, ~2 us of an 10 us kernel
    a.do_bounds = f->bounds_known ? 2 : 1;
    if (f->bounds_known) memcpy(a.bounds_value, f->bounds, 16);
    a.do_undistort = 1;
    a.do_grid = 1;
    a.xy_un = reinterpret_cast<float*>(f->d_xy_un);
    a.bounds = f->d_bounds;
    a.cell_of = nullptr;
    a.cell_start = f->d_cell_start;
    a.cell_items = f->d_cell_items;
    a.n_inside = f->d_n_inside;
    a.ex_meta = V.meta;
    a.ex_total = V.total;
    a.ex_desc = V.desc;
    for (int l = 0; l < 8; l++) a.scale[l] = V.scale[l];
    a.octave = f->d_octave;
    a.desc_by_index = reinterpret_cast<uint4*>(f->d_desc);
    a.ex_angle = V.angle;
    a.angle_by_index = f->d_angle;
    a.xy_un_host = reinterpret_cast<float*>(f->h_block_dev + ((uint8_t*)f->h_xy_un - f->h_block));
    a.s_xy = f->d_s_xy;
    a.s_octave = f->d_s_octave;
    a.s_desc = f->d_s_desc;
    a.perm_host = reinterpret_cast<int32_t*>(f->h_block_dev + ((uint8_t*)f->h_perm - f->h_block));
    a.col_start = f->d_col_start;
    a.header_host = reinterpret_cast<int32_t*>(f->h_block_dev + ((uint8_t*)f->h_header - f->h_block));
    memcpy(out, &a, sizeof(a));
    return SO_OK;
}

int submit_body(so_dframe* f, const uint8_t* image, bool on_device, int w, int h, int stride) {
    if (!f) return SO_ERR_INVALID_ARG;
    if (f->in_flight) {
        last_error_ref() = "so_dframe_submit: the previous frame of this handle has not been collected";
        return SO_ERR_INVALID_ARG;
    }
    f->ready = false;
    f->waited = false;
    f->mirrors = false;
    // from the second frame on the prepare launch below rides at the end of the extractor's frame graph
    if (f->allocated && f->prep_revision)
        extractor_set_graph_tail(f->ex, f, f->prep_revision,
                                 [](void* ctx, hipStream_t s) { launch_frame_prepare(static_cast<so_dframe*>(ctx)->prep, s); });
    int rc = on_device ? so_extractor_submit_device(f->ex, image, w, h, stride)
                       : so_extractor_submit(f->ex, image, w, h, stride);
    if (rc) return rc;
    f->generation++;
    f->in_flight = true;
    f->launched = false;
    if (!image || w <= 0 || h <= 0) return SO_OK;  // empty frame: collect hands out n = 0
    FramePrepareArgs a;
    if ((rc = prepare_args(f, w, h, &a))) return rc;
    if (memcmp(&a, &f->prep, sizeof(a)) != 0) {
        memcpy(&f->prep, &a, sizeof(a));
        f->prep_revision++;
    }
    ExtractorDeviceView V;
    if ((rc = extractor_device_view(f->ex, &V))) return rc;
    if (!extractor_tail_launched(f->ex)) launch_frame_prepare(a, V.stream);  // first frame / no graph / profiling
    SO_HIP(hipGetLastError());
    f->launched = true;
    return SO_OK;
}

}  // namespace

namespace {

template <typename T>
int grow(T** p, size_t old_elems, size_t new_elems, hipStream_t s) {
    T* q = nullptr;
    SO_HIP(hipMalloc((void**)&q, sizeof(T) * new_elems));
    if (*p && old_elems) SO_HIP(hipMemcpyAsync(q, *p, sizeof(T) * old_elems, hipMemcpyDeviceToDevice, s));
    SO_HIP(hipStreamSynchronize(s));
    if (*p) SO_HIP(hipFree(*p));
    *p = q;
    return SO_OK;
}

int reserve(so_map* m, int slots) {
    if (slots <= m->capacity) return SO_OK;
    // 288 GB of HBM: the table grows geometrically and is never shrunk (a 10^6-point map is 64 MB)
    size_t cap = m->capacity ? (size_t)m->capacity : 65536;
    while (cap < (size_t)slots) cap *= 2;
    const size_t old = (size_t)m->size.load();
    std::unique_lock<std::shared_timed_mutex> moving(m->grow_mu);  // no other thread's search is reading the old tables
    int rc;
    if ((rc = grow(&m->d_Xw, 3 * old, 3 * cap, m->stream))) return rc;
    if ((rc = grow(&m->d_normal, 3 * old, 3 * cap, m->stream))) return rc;
    if ((rc = grow(&m->d_max, old, cap, m->stream))) return rc;
    if ((rc = grow(&m->d_min, old, cap, m->stream))) return rc;
    if ((rc = grow(&m->d_desc, 32 * old, 32 * cap, m->stream))) return rc;
    m->capacity = (int)cap;
    return SO_OK;
}

int stage(so_map* m, size_t bytes) {
    if (bytes <= m->h_stage_cap) return SO_OK;
    if (m->h_stage) SO_HIP(hipHostFree(m->h_stage));
    m->h_stage = nullptr;
    m->h_stage_cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    SO_HIP(hipHostMalloc(&m->h_stage, want, hipHostMallocDefault));
    m->h_stage_cap = want;
    return SO_OK;
}

}  // namespace

extern "C" {

int so_dframe_create(so_extractor* ex, const so_camera* cam, so_dframe** out) {
    if (!ex || !cam || !out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    so_dframe* f = new so_dframe();
    f->ex = ex;
    f->cam = FrameCam{cam->fx, cam->fy, cam->cx, cam->cy, cam->k1, cam->k2, cam->p1, cam->p2, cam->k3};
    *out = f;
    return SO_OK;
}

void so_dframe_destroy(so_dframe* f) {
    if (!f) return;
    extractor_release_graph_tail(f->ex, f);
    if (f->allocated) {
        (void)hipSetDevice(f->device);
        ExtractorDeviceView V;
        if (extractor_device_view(f->ex, &V) == SO_OK && V.stream) (void)hipStreamSynchronize(V.stream);
        if (f->d_block) (void)hipFree(f->d_block);
        if (f->h_block) (void)hipHostFree(f->h_block);
    }
    delete f;
}

int so_dframe_group_submit(so_extractor_group* g, so_dframe* const* frames, const uint8_t* const* images, int width,
                           int height, int stride) {
    if (!g || !frames || !images || width <= 0 || height <= 0) return SO_ERR_INVALID_ARG;
    const int n = extractor_group_size(g);
    int n_in = 0;
    for (int i = 0; i < n; i++) {
        if (!images[i]) continue;  // member i sits this chain out (frames[i] is not looked at)
        n_in++;
        so_dframe* f = frames[i];
        if (!f || f->ex != extractor_group_member(g, i)) {
            last_error_ref() = "so_dframe_group_submit: frame i must be built on member i of the group";
            return SO_ERR_INVALID_ARG;
        }
        if (f->in_flight) {
            last_error_ref() = "so_dframe_submit: the previous frame of this handle has not been collected";
            return SO_ERR_INVALID_ARG;
        }
    }
    if (n_in == 0) return SO_ERR_INVALID_ARG;
    int rc = extractor_group_prepare(g, width, height);  // the members' buffers exist from here on
    if (rc) return rc;
    std::vector<FramePrepareArgs> preps((size_t)n);
    for (int i = 0; i < n; i++)
        if (images[i] && (rc = prepare_args(frames[i], width, height, &preps[(size_t)i]))) return rc;
    if ((rc = extractor_group_submit(g, images, width, height, stride, preps.data()))) return rc;
    for (int i = 0; i < n; i++) {
        if (!images[i]) continue;
        so_dframe* f = frames[i];
        f->ready = false;
        f->waited = false;
        f->mirrors = false;
        f->generation++;
        f->in_flight = true;
        f->launched = true;
    }
    return SO_OK;
}

int so_dframe_submit(so_dframe* f, const uint8_t* image, int width, int height, int stride) {
    return submit_impl(f, image, false, width, height, stride);
}

int so_dframe_submit_device(so_dframe* f, const uint8_t* d_image, int width, int height, int stride) {
    return submit_impl(f, d_image, true, width, height, stride);
}

// first half of collect: wait for the device side, read the header (count, bounds, position -> index map are valid)
int so_dframe_wait(so_dframe* f, int* n_out, float* bounds4) {
    if (!f || !n_out) return SO_ERR_INVALID_ARG;
    *n_out = 0;
    if (!f->in_flight) {
        last_error_ref() = "so_dframe_wait without a submitted frame";
        return SO_ERR_INVALID_ARG;
    }
    if (f->waited) {
        *n_out = f->n;
        if (bounds4) memcpy(bounds4, f->bounds, 16);
        return SO_OK;
    }
    int n = 0;
    const int rc = so_extractor_wait(f->ex, &n);  // waits for the extractor's stream (the prepare kernel is behind it)
    if (rc) return rc;
    if (!f->allocated || !f->launched) {  // an empty image: nothing ran behind the extractor
        f->n = f->n_inside = 0;
        f->ready = f->allocated;  // (bounds are those of the handle's earlier frames: they depend on the camera only)
        f->waited = true;
        return SO_OK;
    }
    // the prepare kernel was enqueued behind the extractor's frame on the same stream; the extractor's wait covers it
    // on every path but the host-quadtree one, which finishes inside submit
    ExtractorDeviceView V;
    if (extractor_device_view(f->ex, &V) == SO_OK) SO_HIP(hipStreamSynchronize(V.stream));
    if (f->h_header[0] != n) {
        last_error_ref() = "so_dframe_wait: device and host keypoint counts differ";
        return SO_ERR_HIP;
    }
    f->n = n;
    f->n_inside = f->h_header[1];
    memcpy(f->bounds, f->h_header + 4, 16);
    f->bounds_known = true;
    // octave / angle mirrors for the matcher's resolve come straight from the extractor's host-mapped results
    f->waited = true;
    f->ready = true;
    f->mirrors = false;
    *n_out = n;
    if (bounds4) memcpy(bounds4, f->bounds, 16);
    return SO_OK;
}

int so_dframe_collect(so_dframe* f, so_keypoint* keypoints, float* xy_un, uint8_t* descriptors, int capacity,
                      int* n_out, float* bounds4) {
    if (!f || !n_out || !keypoints || !descriptors) return SO_ERR_INVALID_ARG;
    *n_out = 0;
    if (!f->in_flight) {
        last_error_ref() = "so_dframe_collect without a submitted frame";
        return SO_ERR_INVALID_ARG;
    }
    int n = 0, rc;
    if ((rc = so_dframe_wait(f, &n, nullptr))) return rc;
    int n2 = 0;
    if ((rc = so_extractor_collect(f->ex, keypoints, descriptors, capacity, &n2))) return rc;  // copies only
    f->in_flight = false;
    f->waited = false;
    if (!f->allocated || !f->launched) return SO_OK;
    for (int i = 0; i < n; i++) {
        f->octave[(size_t)i] = keypoints[i].octave;
        f->angle[(size_t)i] = keypoints[i].angle;
    }
    f->mirrors = true;
    if (xy_un) memcpy(xy_un, f->h_xy_un, sizeof(float) * 2 * (size_t)n);
    if (bounds4) memcpy(bounds4, f->bounds, 16);
    *n_out = n;
    return SO_OK;
}

int so_dframe_device_view(const so_dframe* f, so_dframe_view* v) {
    if (!f || !v || !f->ready) return SO_ERR_INVALID_ARG;
    v->n = f->n;
    v->n_inside = f->n_inside;
    memcpy(v->bounds, f->bounds, 16);
    v->xy_un = reinterpret_cast<const float*>(f->d_xy_un);
    v->octave = f->d_octave;
    v->descriptors = f->d_desc;
    v->cell_start = f->d_cell_start;
    v->cell_items = f->d_cell_items;
    v->sorted_xy = reinterpret_cast<const float*>(f->d_s_xy);
    v->sorted_octave = f->d_s_octave;
    v->sorted_descriptors = reinterpret_cast<const uint8_t*>(f->d_s_desc);
    v->col_start = f->d_col_start;
    return SO_OK;
}

// Host copies of the grid (debug / parity tests): cell_start[64*48+1], cell_items[n_inside].
int so_dframe_get_grid(so_dframe* f, int32_t* cell_start, int32_t* cell_items, int32_t* n_inside) {
    if (!f || !f->ready || !cell_start || !cell_items || !n_inside) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(f->device));
    SO_HIP(so::memcpy_sync(cell_start, f->d_cell_start, sizeof(int32_t) * (kFrameGridCols * kFrameGridRows + 1),
                     hipMemcpyDeviceToHost));
    *n_inside = f->n_inside;
    if (f->n_inside > 0) memcpy(cell_items, f->h_perm, sizeof(int32_t) * (size_t)f->n_inside);
    return SO_OK;
}

// ------------------------------------------------------------------------------------------------
// map-point table
// ------------------------------------------------------------------------------------------------
int so_map_create(int device, so_map** out) {
    if (!out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(device));
    so_map* m = new so_map();
    m->device = device;
    const hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete m;
        return hip_fail(e, "map init", __FILE__, __LINE__);
    }
    *out = m;
    return SO_OK;
}

// The table's writes (so_map_write / _write_positions / _write_rows: a copy or one launch, synchronous) on a matcher's stream
// instead of a stream of the table's own: with several agents per GPU every busy stream beyond the runtime's hardware queues
// shares a queue with somebody's long chain of launches.  The matcher must outlive the table's last write.
int so_map_share_stream(so_map* m, const so_matcher* with) {
    if (!m || !with) return SO_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)(uintptr_t)so_matcher_stream_id(with);
    if (!s) return SO_ERR_INVALID_ARG;
    (void)hipSetDevice(m->device);
    std::unique_lock<std::shared_timed_mutex> lk(m->grow_mu);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->stream && m->owns_stream) (void)hipStreamDestroy(m->stream);
    m->stream = s;
    m->owns_stream = false;
    return SO_OK;
}

void so_map_destroy(so_map* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream && m->owns_stream) {
        (void)hipStreamSynchronize(m->stream);
        (void)hipStreamDestroy(m->stream);
    }
    for (void* p : {(void*)m->d_Xw, (void*)m->d_normal, (void*)m->d_max, (void*)m->d_min, (void*)m->d_desc})
        if (p) (void)hipFree(p);
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    delete m;
}

int so_map_size(const so_map* m) { return m ? m->size.load() : 0; }

int so_map_write(so_map* m, int32_t first, int32_t n, const float* Xw, const float* normal, const float* max_dist,
                 const float* min_dist, const uint8_t* desc) {
    if (!m || first < 0 || n < 0 || first > m->size) return SO_ERR_INVALID_ARG;
    if (n == 0) return SO_OK;
    const bool appending = first + n > m->size;
    if (appending && (!Xw || !normal || !max_dist || !min_dist || !desc)) return SO_ERR_INVALID_ARG;  // new slots are written whole
    SO_HIP(hipSetDevice(m->device));
    int rc;
    if ((rc = reserve(m, first + n))) return rc;
    const size_t sn = (size_t)n;
    if ((rc = stage(m, sn * 64))) return rc;
    uint8_t* h = (uint8_t*)m->h_stage;
    hipStream_t s = m->stream;
    // the five columns side by side in the pinned staging block; ONE kernel reads them in place (pinned memory is
    // device-visible) instead of five host-to-device copies
    size_t o = 0;
    auto put = [&](const void* src, size_t bytes) -> const uint8_t* {
        if (!src) return nullptr;
        memcpy(h + o, src, bytes);
        const uint8_t* at = h + o;
        o += bytes;  // (all sizes are multiples of four)
        return at;
    };
    const uint8_t* sX = put(Xw, 12 * sn);
    const uint8_t* sN = put(normal, 12 * sn);
    const uint8_t* sMx = put(max_dist, 4 * sn);
    const uint8_t* sMn = put(min_dist, 4 * sn);
    const uint8_t* sD = put(desc, 32 * sn);
    launch_map_write_range(m->d_Xw, m->d_normal, m->d_max, m->d_min, m->d_desc, (const float*)sX, (const float*)sN, (const float*)sMx,
                           (const float*)sMn, sD, first, n, s);
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(s));  // keyframe rate; afterwards every stream sees the new rows
    if (appending) m->size = first + n;
    return SO_OK;
}

int so_map_write_positions(so_map* m, int32_t n, const int32_t* slots, const float* Xw) {
    if (!m || n < 0 || (n > 0 && (!slots || !Xw))) return SO_ERR_INVALID_ARG;
    if (n == 0) return SO_OK;
    for (int i = 0; i < n; i++)
        if (slots[i] < 0 || slots[i] >= m->size) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    int rc;
    const size_t sn = (size_t)n;
    if ((rc = stage(m, sn * 16))) return rc;
    uint8_t* h = (uint8_t*)m->h_stage;
    memcpy(h, slots, 4 * sn);
    memcpy(h + 4 * sn, Xw, 12 * sn);
    // pinned memory is device-visible: the scatter kernel reads the staging block in place
    launch_map_scatter_positions(m->d_Xw, reinterpret_cast<const int32_t*>(h), reinterpret_cast<const float*>(h + 4 * sn), n,
                                 m->stream);
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(m->stream));
    return SO_OK;
}

int so_map_write_rows(so_map* m, int32_t n, const int32_t* slots, const float* Xw, const float* normal, const float* max_dist,
                      const float* min_dist) {
    if (!m || n < 0 || (n > 0 && !slots)) return SO_ERR_INVALID_ARG;
    if (n == 0 || (!Xw && !normal && !max_dist && !min_dist)) return SO_OK;
    for (int i = 0; i < n; i++)
        if (slots[i] < 0 || slots[i] >= m->size) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    int rc;
    const size_t sn = (size_t)n;
    if ((rc = stage(m, sn * 36))) return rc;
    uint8_t* h = (uint8_t*)m->h_stage;
    memcpy(h, slots, 4 * sn);
    size_t o = 4 * sn;
    auto put = [&](const float* src, size_t bytes) -> const float* {
        if (!src) return nullptr;
        memcpy(h + o, src, bytes);
        const float* p = reinterpret_cast<const float*>(h + o);
        o += bytes;
        return p;
    };
    const float* pX = put(Xw, 12 * sn);
    const float* pN = put(normal, 12 * sn);
    const float* pmx = put(max_dist, 4 * sn);
    const float* pmn = put(min_dist, 4 * sn);
    // pinned memory is device-visible: the scatter kernel reads the staging block in place
    launch_map_scatter_rows(m->d_Xw, m->d_normal, m->d_max, m->d_min, reinterpret_cast<const int32_t*>(h), pX, pN, pmx, pmn, n, m->stream);
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(m->stream));
    return SO_OK;
}

int so_map_read(so_map* m, int32_t first, int32_t n, float* Xw, uint8_t* desc) {
    if (!m || first < 0 || n < 0 || first + n > m->size) return SO_ERR_INVALID_ARG;
    if (n == 0) return SO_OK;
    SO_HIP(hipSetDevice(m->device));
    if (Xw) SO_HIP(so::memcpy_sync(Xw, m->d_Xw + 3 * (size_t)first, 12 * (size_t)n, hipMemcpyDeviceToHost));
    if (desc) SO_HIP(so::memcpy_sync(desc, m->d_desc + 32 * (size_t)first, 32 * (size_t)n, hipMemcpyDeviceToHost));
    return SO_OK;
}

}  // extern "C"
