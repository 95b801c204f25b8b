// extractor.cpp — host driver + C ABI of the ORB extractor (include/swarmorb.h).
//
// Replaces ORB_SLAM2::ORBextractor (code/src/ORBextractor.cc:340-855).  Per frame, on ONE HIP stream:
//   image copy -> [7 resize launches -> FAST score+NMS (all levels, 1 launch) -> low-threshold pass -> compaction
//   -> DistributeOctTree (1 launch, a workgroup per level) -> fused angle+blur+BRIEF writing keypoints and
//   descriptors straight into host-mapped memory] -> ONE host sync (the reference has 25).
// The bracketed chain is captured once as a hipGraph and replayed with one launch per frame; so_extractor_submit /
// _collect split the frame around the sync so that it runs under the caller's other work.  Level quotas above
// 1020 keypoints (or more than four root cells) take the reference's own placement of DistributeOctTree - on
// the host, between two syncs (quadtree.cpp) - with identical output.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "extractor_internal.h"
#include "frame_device.h"
#include "orb_device.h"
#include "quadtree.h"
#include "so_common.h"

using namespace so;

struct so_extractor {
    so_extractor_config cfg{};
    float scale[kMaxLevels]{}, inv_scale[kMaxLevels]{}, sigma2[kMaxLevels]{}, inv_sigma2[kMaxLevels]{};
    int features_per_level[kMaxLevels]{};

    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t ev[8]{};
    bool profiling = false;
    float prof_ms[SO_EXTRACTOR_N_STAGES]{};

    // sized at the first frame
    bool allocated = false;
    int width = 0, height = 0;
    PyramidParams P{};
    std::vector<void*> dev_allocs;
    int32_t* d_rowcount = nullptr;
    // device quadtree path (default): candidates never leave the GPU, one host sync per frame
    bool device_qt = false;
    bool cands_on_host = false;        // h_cands holds the last frame's candidates (debug API)
    bool pending_prof = false;         // the in-flight frame recorded its stage events
    // the frame's twelve launches as one hipGraph (captured on the first unprofiled frame of the one-sync path):
    // one runtime call per frame instead of twelve - the calls serialise on the runtime's lock when several agents
    // share a process
    // A consumer's launch can ride at the end of the graph (extractor_internal.h): one captured graph per (owner,
    // revision) - three device-resident frames take turns on one extractor in the tracking loop.
    struct FrameGraph {
        void* owner;
        uint64_t revision;
        hipGraph_t graph;
        hipGraphExec_t exec;
    };
    std::vector<FrameGraph> graphs;
    bool graph_failed = false;
    void* tail_owner = nullptr;
    uint64_t tail_revision = 0;
    ExtractorTailFn tail_fn = nullptr;
    bool tail_launched = false;
    int pending = 0;                   // 1: a submitted frame is in flight on the stream, 2: finished into pend_* below
    double t_begin = 0.0, t_enq = 0.0;
    std::vector<so_keypoint> pend_kps;  // submit on the host-quadtree path runs to completion into these
    std::vector<uint8_t> pend_desc;
    int pend_n = 0;
    Candidate* d_cands = nullptr;
    CandidateHeader* d_header = nullptr;
    SelectedKp* d_qt_sel = nullptr;    // [nlevels][qt_stride]
    int32_t* d_qt_count = nullptr;
    int qt_stride = 0;
    SelectedKp* h_meta = nullptr;      // host-mapped: (x, y, level, score) of every output keypoint
    SelectedKp* h_meta_dev = nullptr;
    int32_t* h_total = nullptr;        // host-mapped: number of output keypoints
    int32_t* h_total_dev = nullptr;
    Candidate* h_cands = nullptr;      // host-mapped
    Candidate* h_cands_dev = nullptr;  // device view of h_cands
    CandidateHeader* h_header = nullptr;
    CandidateHeader* h_header_dev = nullptr;
    int cand_capacity = 0;
    // phase 2 is zero-copy: the describe kernel reads the survivors from, and writes descriptors + angles to,
    // host-mapped memory (a few tens of KB per frame), so there is no H2D / D2H copy to enqueue and wait for
    SelectedKp* h_sel = nullptr;      // host-mapped
    SelectedKp* h_sel_dev = nullptr;  // device view
    uint8_t* h_desc = nullptr;        // host-mapped [cap*32 desc][cap*4 angle]
    uint8_t* h_desc_dev = nullptr;
    int out_capacity = 0;
    DescribeDeviceOut dev_out{};      // HBM-resident copies of desc / angle / meta / total (device-quadtree path)
    uint8_t* d_upload = nullptr;      // tightly packed landing buffer of host images (width x height)

    KeypointQuadtree qt;
    std::vector<int> picked;
    std::vector<int> sel_cand;  // candidate index of each selected keypoint
};

namespace {

int cv_round(float v) { return (int)lrintf(v); }

// ORBextractor::ORBextractor, code/src/ORBextractor.cc:340-378
void make_tables(so_extractor* ex) {
    const int nl = ex->cfg.nlevels;
    const double sf = (double)ex->cfg.scale_factor;
    ex->scale[0] = 1.0f;
    ex->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        ex->scale[i] = (float)((double)ex->scale[i - 1] * sf);
        ex->sigma2[i] = ex->scale[i] * ex->scale[i];
    }
    for (int i = 0; i < nl; i++) {
        ex->inv_scale[i] = 1.0f / ex->scale[i];
        ex->inv_sigma2[i] = 1.0f / ex->sigma2[i];
    }
    const float factor = (float)(1.0 / sf);
    float desired =
        (float)((double)((float)ex->cfg.nfeatures * (1.0f - factor)) / (1.0 - std::pow((double)factor, (double)nl)));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        ex->features_per_level[l] = cv_round(desired);
        sum += ex->features_per_level[l];
        desired *= factor;
    }
    ex->features_per_level[nl - 1] = std::max(ex->cfg.nfeatures - sum, 0);
}

template <typename T>
int dev_alloc(so_extractor* ex, T** p, size_t bytes, bool zero) {
    void* q = nullptr;
    SO_HIP(hipMalloc(&q, bytes));
    ex->dev_allocs.push_back(q);
    if (zero) SO_HIP(hipMemsetAsync(q, 0, bytes, ex->stream));
    *p = reinterpret_cast<T*>(q);
    return SO_OK;
}

int round_up(int v, int a) { return (v + a - 1) / a * a; }

// ComputePyramid's first-frame allocation, code/src/ORBextractor.cc:822-837 (no 19-px border: see orb_device.h)
int allocate(so_extractor* ex, int w, int h) {
    PyramidParams& P = ex->P;
    P.nlevels = ex->cfg.nlevels;
    P.th_high = ex->cfg.ini_th_fast;
    P.th_low = ex->cfg.min_th_fast;
    int tile_base = 0, row_base = 0;
    for (int l = 0; l < P.nlevels; l++) {
        LevelDesc& L = P.lv[l];
        L.w = cv_round((float)w * ex->inv_scale[l]);
        L.h = cv_round((float)h * ex->inv_scale[l]);
        if (L.w < 1 || L.h < 1) {
            last_error_ref() = "image too small for the requested number of pyramid levels";
            return SO_ERR_INVALID_ARG;
        }
        L.pitch = round_up(L.w + 64, 64);
        const int rw = L.w - 2 * kFastBorder, rh = L.h - 2 * kFastBorder;
        L.ntx = rw >= 7 ? (rw - 6 + kTile - 1) / kTile : 0;
        L.nty = rh >= 7 ? (rh - 6 + kTile - 1) / kTile : 0;
        if (L.ntx == 0 || L.nty == 0) L.ntx = L.nty = 0;
        if (L.ntx > 128) {
            last_error_ref() = "image wider than 4128 pixels is not supported";
            return SO_ERR_INVALID_ARG;
        }
        L.spitch = 0;  // (the score map is tile-major since round 3: no pitch)
        L.tile_base = tile_base;
        L.row_base = row_base;
        tile_base += L.ntx * L.nty;
        row_base += L.nty * kTile;
        // +4 rows of slack: FAST tiles at the bottom edge clamp rows, the slack only guards dword over-reads
        int rc = dev_alloc(ex, &L.img, (size_t)L.pitch * (L.h + 4), true);
        if (rc) return rc;
        rc = dev_alloc(ex, &L.score, (size_t)kScoreBlock * L.ntx * L.nty + 256, true);
        if (rc) return rc;
        rc = dev_alloc(ex, &L.tileflag, (size_t)L.ntx * L.nty + 64, true);
        if (rc) return rc;
        rc = dev_alloc(ex, &L.bitmap, sizeof(uint32_t) * ((size_t)L.ntx * L.nty * kTile + 64), true);
        if (rc) return rc;
    }
    P.total_tiles = tile_base;
    P.total_rows = row_base;
    if (P.total_rows > 8192) {
        last_error_ref() = "image too tall for the candidate compaction (sum of level heights > 8192)";
        return SO_ERR_INVALID_ARG;
    }

    ex->cand_capacity = P.nlevels * kFastCap;
    int rc = dev_alloc(ex, &ex->d_rowcount, sizeof(int32_t) * ((size_t)P.total_rows + 64), true);
    if (rc) return rc;
    SO_HIP(hipHostMalloc((void**)&ex->h_cands, sizeof(Candidate) * (size_t)ex->cand_capacity + 64, hipHostMallocMapped));
    SO_HIP(hipHostGetDevicePointer((void**)&ex->h_cands_dev, ex->h_cands, 0));
    SO_HIP(hipHostMalloc((void**)&ex->h_header, sizeof(CandidateHeader), hipHostMallocMapped));
    SO_HIP(hipHostGetDevicePointer((void**)&ex->h_header_dev, ex->h_header, 0));
    memset(ex->h_header, 0, sizeof(CandidateHeader));

    ex->out_capacity = so_extractor_capacity(ex);
    SO_HIP(hipHostMalloc((void**)&ex->h_sel, sizeof(SelectedKp) * (size_t)ex->out_capacity, hipHostMallocMapped));
    SO_HIP(hipHostGetDevicePointer((void**)&ex->h_sel_dev, ex->h_sel, 0));
    SO_HIP(hipHostMalloc((void**)&ex->h_desc, (size_t)ex->out_capacity * 36, hipHostMallocMapped));
    SO_HIP(hipHostGetDevicePointer((void**)&ex->h_desc_dev, ex->h_desc, 0));
    // DistributeOctTree on the device needs the whole tree of a level in LDS: N <= 1020 nodes, <= 4 root nodes
    ex->device_qt = getenv("SWARMORB_HOST_QUADTREE") == nullptr;
    int max_n = 0;
    for (int l = 0; l < P.nlevels; l++) {
        max_n = std::max(max_n, ex->features_per_level[l]);
        const LevelDesc& L = P.lv[l];
        const int W = L.w - 2 * kFastBorder, H = L.h - 2 * kFastBorder;
        if (L.ntx > 0 && (H <= 0 || (int)std::round((float)W / (float)H) > 4)) ex->device_qt = false;
    }
    if (max_n > 1020) ex->device_qt = false;
    ex->qt_stride = max_n + 4;
    rc = dev_alloc(ex, &ex->d_cands, sizeof(Candidate) * (size_t)ex->cand_capacity + 64, true);
    if (rc) return rc;
    rc = dev_alloc(ex, &ex->d_header, sizeof(CandidateHeader), true);
    if (rc) return rc;
    rc = dev_alloc(ex, &ex->d_qt_sel, sizeof(SelectedKp) * (size_t)ex->qt_stride * kMaxLevels, true);
    if (rc) return rc;
    rc = dev_alloc(ex, &ex->d_qt_count, sizeof(int32_t) * kMaxLevels, true);
    if (rc) return rc;
    SO_HIP(hipHostMalloc((void**)&ex->h_meta, sizeof(SelectedKp) * (size_t)ex->out_capacity, hipHostMallocMapped));
    SO_HIP(hipHostGetDevicePointer((void**)&ex->h_meta_dev, ex->h_meta, 0));
    SO_HIP(hipHostMalloc((void**)&ex->h_total, 64, hipHostMallocMapped));
    SO_HIP(hipHostGetDevicePointer((void**)&ex->h_total_dev, ex->h_total, 0));
    *ex->h_total = 0;
    if (ex->device_qt) {
        rc = dev_alloc(ex, &ex->dev_out.desc, (size_t)ex->out_capacity * 32, true);
        if (rc) return rc;
        rc = dev_alloc(ex, &ex->dev_out.angle, sizeof(float) * (size_t)ex->out_capacity, true);
        if (rc) return rc;
        rc = dev_alloc(ex, &ex->dev_out.meta, sizeof(SelectedKp) * (size_t)ex->out_capacity, true);
        if (rc) return rc;
        rc = dev_alloc(ex, &ex->dev_out.total, 64, true);
        if (rc) return rc;
    }
    rc = dev_alloc(ex, &ex->d_upload, (size_t)w * h + 256, false);
    if (rc) return rc;
    SO_HIP(hipStreamSynchronize(ex->stream));
    ex->width = w;
    ex->height = h;
    ex->allocated = true;
    return SO_OK;
}

// forget the captured frame graphs of one tail owner (nullptr: the graph without a tail)
void drop_frame_graphs(so_extractor* ex, void* owner) {
    for (size_t i = 0; i < ex->graphs.size();) {
        if (ex->graphs[i].owner == owner) {
            (void)hipGraphExecDestroy(ex->graphs[i].exec);
            (void)hipGraphDestroy(ex->graphs[i].graph);
            ex->graphs.erase(ex->graphs.begin() + (long)i);
        } else {
            i++;
        }
    }
}

// ComputePyramid's resizes: one fused launch, or the chained per-level launches (SWARMORB_CHAINED_PYRAMID, or a
// configuration the fused kernel cannot hold)
void launch_pyramid(const PyramidParams& P, hipStream_t s) {
    // Levels 1 .. first take one launch each; the ones behind them can come out of ONE fused launch
    // (pyramid_fused_kernel, SWARMORB_PYRAMID_FUSE_FROM=first).  Measured on MI355X (752x480, pyramid stage incl. the
    // 8 us image copy): all seven levels fused 34.5 us, fused from level 1 / 2 / 3: 30.4 / 29.1 / 28.2 us, seven chained
    // launches inside the frame's hipGraph 25.3 us - a 2.5 us launch of a wide grid beats a workgroup that walks the
    // levels one after the other, so the chain stays the default and the fused kernel an option.
    static const int first_env = getenv("SWARMORB_PYRAMID_FUSE_FROM") ? atoi(getenv("SWARMORB_PYRAMID_FUSE_FROM")) : 99;
    const int first = std::min(std::max(first_env, 0), P.nlevels - 1);
    for (int l = 1; l <= first; l++) launch_resize(P.lv[l - 1], P.lv[l], s);
    if (launch_pyramid_fused(P, first, s)) return;
    for (int l = first + 1; l < P.nlevels; l++) launch_resize(P.lv[l - 1], P.lv[l], s);
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Second half of a frame on the one-sync path: wait for the stream, assemble cv::KeyPoint-compatible records.
int collect_impl(so_extractor* ex, so_keypoint* kps, uint8_t* desc, int capacity, int* n_out) {
    if (!ex || !n_out || !kps || !desc) return SO_ERR_INVALID_ARG;
    *n_out = 0;
    if (ex->pending == 0) {
        last_error_ref() = "so_extractor_collect without a submitted frame";
        return SO_ERR_INVALID_ARG;
    }
    if (capacity < ex->out_capacity) return SO_ERR_CAPACITY;
    if (ex->pending == 2) {
        ex->pending = 0;
        memcpy(kps, ex->pend_kps.data(), sizeof(so_keypoint) * (size_t)ex->pend_n);
        memcpy(desc, ex->pend_desc.data(), (size_t)ex->pend_n * 32);
        *n_out = ex->pend_n;
        return SO_OK;
    }
    SO_HIP(hipSetDevice(ex->cfg.device));
    const bool prof = ex->pending_prof;
    if (ex->pending == 1) SO_HIP(hipStreamSynchronize(ex->stream));  // the only sync of the frame (unless so_extractor_wait did it)
    ex->pending = 0;
    const double t_synced = now_ms();
    ex->cands_on_host = false;
    const int n = std::min(*ex->h_total, ex->out_capacity);
    memcpy(desc, ex->h_desc, (size_t)n * 32);
    const float* angles = reinterpret_cast<const float*>(ex->h_desc + (size_t)ex->out_capacity * 32);
    for (int i = 0; i < n; i++) {
        const SelectedKp& sk = ex->h_meta[i];
        so_keypoint& o = kps[i];
        const int l = sk.level;
        o.x = (float)sk.x;
        o.y = (float)sk.y;
        if (l != 0) {  // ORBextractor.cc:808-814
            o.x *= ex->scale[l];
            o.y *= ex->scale[l];
        }
        o.size = (float)(int)(31.0f * ex->scale[l]);
        o.angle = angles[i];
        o.response = (float)sk.score;
        o.octave = l;
        o.class_id = -1;
    }
    *n_out = n;
    if (prof) {
        float ms = 0.f;
        for (int i = 0; i < SO_EXTRACTOR_N_STAGES; i++) ex->prof_ms[i] = 0.f;
        (void)hipEventElapsedTime(&ms, ex->ev[0], ex->ev[1]); ex->prof_ms[0] = ms;
        (void)hipEventElapsedTime(&ms, ex->ev[1], ex->ev[2]); ex->prof_ms[1] = ms;
        (void)hipEventElapsedTime(&ms, ex->ev[2], ex->ev[3]); ex->prof_ms[2] = ms;
        (void)hipEventElapsedTime(&ms, ex->ev[3], ex->ev[4]); ex->prof_ms[3] = ms;
        (void)hipEventElapsedTime(&ms, ex->ev[4], ex->ev[5]); ex->prof_ms[8] = ms;  // quadtree, on the GPU
        (void)hipEventElapsedTime(&ms, ex->ev[5], ex->ev[6]); ex->prof_ms[4] = ms;
        ex->prof_ms[6] = (float)(ex->t_enq - ex->t_begin);
        ex->prof_ms[7] = (float)(t_synced - ex->t_enq);  // includes whatever the caller did between submit and collect
        ex->prof_ms[10] = (float)(now_ms() - t_synced);
        ex->prof_ms[5] = (float)(now_ms() - ex->t_begin);  // submit to collect, whatever the caller did in between
    }
    return SO_OK;
}

int run_impl(so_extractor* ex, const uint8_t* image, bool on_device, int w, int h, int stride, so_keypoint* kps,
             uint8_t* desc, int capacity, int* n_out, bool submit_only = false) {
    if (!ex) return SO_ERR_INVALID_ARG;
    // the tail applies to ONE submit (a direct so_extractor_submit behind a device-resident frame must not run that
    // frame's prepare launch over the data it holds): it is consumed before ANY early return - a submit refused
    // because the extractor is still busy must not leave the frame's launch armed for whoever submits next
    void* const tail_owner = ex->tail_owner;
    const uint64_t tail_revision = ex->tail_revision;
    const ExtractorTailFn tail_fn = ex->tail_fn;
    ex->tail_owner = nullptr;
    ex->tail_revision = 0;
    ex->tail_fn = nullptr;
    if (!submit_only && !n_out) return SO_ERR_INVALID_ARG;
    if (ex->pending) {
        last_error_ref() = "a submitted frame has not been collected";
        return SO_ERR_INVALID_ARG;
    }
    ex->tail_launched = false;
    if (n_out) *n_out = 0;
    if (!image || w <= 0 || h <= 0) {  // ORBextractor.cc:750-751
        if (submit_only) {
            ex->pend_n = 0;
            ex->pending = 2;
        }
        return SO_OK;
    }
    if (stride < w || (!submit_only && (!kps || !desc))) return SO_ERR_INVALID_ARG;
    const double t_begin = now_ms();
    SO_HIP(hipSetDevice(ex->cfg.device));
    if (!ex->allocated) {
        int rc = allocate(ex, w, h);
        if (rc) return rc;
    } else if (w != ex->width || h != ex->height) {
        last_error_ref() = "image size changed between frames";
        return SO_ERR_SIZE_CHANGED;
    }
    if (submit_only && !(ex->device_qt && ex->P.total_tiles > 0)) {
        // host-quadtree contexts have a mid-frame sync: run the frame to completion now, hand it out at collect
        ex->pend_kps.resize((size_t)ex->out_capacity);
        ex->pend_desc.resize((size_t)ex->out_capacity * 32);
        int n = 0;
        const int rc = run_impl(ex, image, on_device, w, h, stride, ex->pend_kps.data(), ex->pend_desc.data(),
                                ex->out_capacity, &n, false);
        if (rc) return rc;
        ex->pend_n = n;
        ex->pending = 2;
        return SO_OK;
    }
    if (!submit_only && capacity < ex->out_capacity) return SO_ERR_CAPACITY;
    PyramidParams& P = ex->P;
    hipStream_t s = ex->stream;
    const bool prof = ex->profiling;

    if (prof) SO_HIP(hipEventRecord(ex->ev[0], s));
    // ComputePyramid, code/src/ORBextractor.cc:837-853
    // host images the device can read in place (pinned by the caller: hipHostMalloc / hipHostRegister / a framework's
    // pinned allocator) skip the DMA engine altogether; pageable ones take the copy path below
    static const bool no_ingest = getenv("SWARMORB_NO_INGEST_KERNEL") != nullptr;
    const uint8_t* visible = nullptr;
    if (!on_device && stride == w && !no_ingest) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, image) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer)
            visible = static_cast<const uint8_t*>(attr.devicePointer);
        else
            (void)hipGetLastError();  // pageable memory: not an error
    }
    if (visible) {
        launch_ingest(visible, w, h, P.lv[0], s);
    } else if (!on_device && stride == w) {
        // a pitched host -> device copy is split into one DMA per row when the width is not a multiple of four bytes
        // (1241-px KITTI rows: 376 copies, 2.2 ms); one linear upload into a packed landing buffer followed by a
        // device-side repack into the pitched level-0 image costs two enqueues instead
        SO_HIP(hipMemcpyAsync(ex->d_upload, image, (size_t)w * h, hipMemcpyHostToDevice, s));
        SO_HIP(hipMemcpy2DAsync(P.lv[0].img, (size_t)P.lv[0].pitch, ex->d_upload, (size_t)w, (size_t)w, (size_t)h,
                                hipMemcpyDeviceToDevice, s));
    } else {
        SO_HIP(hipMemcpy2DAsync(P.lv[0].img, (size_t)P.lv[0].pitch, image, (size_t)stride, (size_t)w, (size_t)h,
                                on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    }
    static const bool no_graph = getenv("SWARMORB_NO_GRAPH") != nullptr;
    if (!prof && !no_graph && ex->device_qt && P.total_tiles > 0 && !ex->graph_failed) {
        float* angle_dev = reinterpret_cast<float*>(ex->h_desc_dev + (size_t)ex->out_capacity * 32);
        hipGraphExec_t exec = nullptr;
        for (const auto& g : ex->graphs)
            if (g.owner == tail_owner && g.revision == tail_revision) exec = g.exec;
        if (!exec) {  // every launch argument is fixed once the context is sized: capture the chain once per tail
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                launch_pyramid(P, s);
                launch_fast_score(P, s);
                launch_fast_low(P, s);
                launch_quadtree(P, ex->features_per_level, ex->qt_stride, ex->d_qt_sel, ex->d_qt_count, s);
                launch_describe_qt(P, ex->d_qt_sel, ex->d_qt_count, ex->qt_stride, ex->out_capacity, ex->h_desc_dev,
                                   angle_dev, ex->h_meta_dev, ex->h_total_dev, ex->dev_out, s);
                if (tail_fn) tail_fn(tail_owner, s);
                e = hipStreamEndCapture(s, &graph);
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (e != hipSuccess) {  // run this and later frames launch by launch
                (void)hipGetLastError();
                if (graph) (void)hipGraphDestroy(graph);
                ex->graph_failed = true;
                exec = nullptr;
            } else {
                drop_frame_graphs(ex, tail_owner);  // an older revision of the same owner
                if (ex->graphs.size() >= 8) drop_frame_graphs(ex, ex->graphs.front().owner);
                ex->graphs.push_back({tail_owner, tail_revision, graph, exec});
            }
        }
        if (exec) {
            SO_HIP(hipGraphLaunch(exec, s));
            ex->tail_launched = tail_fn != nullptr;
            ex->t_begin = t_begin;
            ex->t_enq = now_ms();
            ex->pending = 1;
            ex->pending_prof = false;
            if (submit_only) return SO_OK;
            return collect_impl(ex, kps, desc, capacity, n_out);
        }
    }
    launch_pyramid(P, s);
    if (prof) SO_HIP(hipEventRecord(ex->ev[1], s));
    // ComputeKeyPointsOctTree, code/src/ORBextractor.cc:691-744 (all levels batched)
    if (P.total_tiles > 0) {
        launch_fast_score(P, s);
        if (prof) SO_HIP(hipEventRecord(ex->ev[2], s));
        launch_fast_low(P, s);
        if (prof) SO_HIP(hipEventRecord(ex->ev[3], s));
        if (ex->device_qt) {
            // one-sync path: candidates -> device quadtree -> fused describe writing straight to host-mapped memory
            // candidates -> device quadtree (reads the keep bitmap itself: no candidate list) -> fused describe
            if (prof) SO_HIP(hipEventRecord(ex->ev[4], s));
            launch_quadtree(P, ex->features_per_level, ex->qt_stride, ex->d_qt_sel, ex->d_qt_count, s);
            if (prof) SO_HIP(hipEventRecord(ex->ev[5], s));
            float* angle_dev = reinterpret_cast<float*>(ex->h_desc_dev + (size_t)ex->out_capacity * 32);
            launch_describe_qt(P, ex->d_qt_sel, ex->d_qt_count, ex->qt_stride, ex->out_capacity, ex->h_desc_dev,
                               angle_dev, ex->h_meta_dev, ex->h_total_dev, ex->dev_out, s);
            if (prof) SO_HIP(hipEventRecord(ex->ev[6], s));
            SO_HIP(hipGetLastError());
            ex->t_begin = t_begin;
            ex->t_enq = now_ms();
            ex->pending = 1;
            ex->pending_prof = prof;
            if (submit_only) return SO_OK;
            return collect_impl(ex, kps, desc, capacity, n_out);
        }
        launch_emit(P, ex->d_rowcount, ex->h_cands_dev, ex->h_header_dev, nullptr, ex->cand_capacity, s);
        ex->cands_on_host = true;
        if (prof) SO_HIP(hipEventRecord(ex->ev[4], s));
        SO_HIP(hipGetLastError());
        const double t_enq = now_ms();
        SO_HIP(hipStreamSynchronize(s));  // sync #1: candidates + header are in host memory now
        if (prof) {
            ex->prof_ms[6] = (float)(t_enq - t_begin);
            ex->prof_ms[7] = (float)(now_ms() - t_enq);
        }
    } else {
        memset(ex->h_header, 0, sizeof(CandidateHeader));
        SO_HIP(hipStreamSynchronize(s));
    }

    // DistributeOctTree per level (host), code/src/ORBextractor.cc:725-727
    const double t_tree = now_ms();
    const CandidateHeader& H = *ex->h_header;
    int n = 0;
    ex->sel_cand.clear();
    for (int l = 0; l < P.nlevels; l++) {
        const Candidate* c = ex->h_cands + H.offset[l];
        const LevelDesc& L = P.lv[l];
        ex->qt.distribute(c, H.count[l], L.w - 2 * kFastBorder, L.h - 2 * kFastBorder, ex->features_per_level[l],
                          ex->picked);
        for (int idx : ex->picked) {
            if (n >= ex->out_capacity) break;
            const Candidate& k = c[idx];
            ex->h_sel[n].x = (int16_t)(k.x + kFastBorder);  // addBorder_kernel, Fast_gpu.cu:461-470
            ex->h_sel[n].y = (int16_t)(k.y + kFastBorder);
            ex->h_sel[n].level = (uint16_t)l;
            ex->h_sel[n].score = k.score;
            ex->sel_cand.push_back(H.offset[l] + idx);
            n++;
        }
    }
    const double t_phase2 = now_ms();
    double t_asm = t_phase2;
    *ex->h_total = n;  // device consumers of this path read the count through the host-mapped word
    if (n > 0) {
        if (prof) SO_HIP(hipEventRecord(ex->ev[5], s));
        float* angle_dev = reinterpret_cast<float*>(ex->h_desc_dev + (size_t)ex->out_capacity * 32);
        launch_describe(P, ex->h_sel_dev, n, ex->h_desc_dev, angle_dev, s);
        if (prof) SO_HIP(hipEventRecord(ex->ev[6], s));
        SO_HIP(hipGetLastError());
        SO_HIP(hipStreamSynchronize(s));  // sync #2: descriptors + angles are in host memory now
        t_asm = now_ms();
        memcpy(desc, ex->h_desc, (size_t)n * 32);
        const float* angles = reinterpret_cast<const float*>(ex->h_desc + (size_t)ex->out_capacity * 32);
        for (int i = 0; i < n; i++) {
            const SelectedKp& sk = ex->h_sel[i];
            const Candidate& k = ex->h_cands[ex->sel_cand[(size_t)i]];
            so_keypoint& o = kps[i];
            const int l = sk.level;
            o.x = (float)sk.x;
            o.y = (float)sk.y;
            if (l != 0) {  // ORBextractor.cc:808-814
                o.x *= ex->scale[l];
                o.y *= ex->scale[l];
            }
            o.size = (float)(int)(31.0f * ex->scale[l]);  // PATCH_SIZE*scale passed through an int parameter
            o.angle = angles[i];
            o.response = (float)k.score;
            o.octave = l;
            o.class_id = -1;
        }
    }
    *n_out = n;
    if (prof) {
        float ms = 0.f;
        for (int i = 0; i < 6; i++) ex->prof_ms[i] = 0.f;
        ex->prof_ms[8] = (float)(t_phase2 - t_tree);
        ex->prof_ms[9] = (float)(t_asm - t_phase2);
        ex->prof_ms[10] = (float)(now_ms() - t_asm);
        if (P.total_tiles > 0) {
            (void)hipEventElapsedTime(&ms, ex->ev[0], ex->ev[1]); ex->prof_ms[0] = ms;
            (void)hipEventElapsedTime(&ms, ex->ev[1], ex->ev[2]); ex->prof_ms[1] = ms;
            (void)hipEventElapsedTime(&ms, ex->ev[2], ex->ev[3]); ex->prof_ms[2] = ms;
            (void)hipEventElapsedTime(&ms, ex->ev[3], ex->ev[4]); ex->prof_ms[3] = ms;
        }
        if (n > 0) {
            (void)hipEventElapsedTime(&ms, ex->ev[5], ex->ev[6]); ex->prof_ms[4] = ms;
        }
        ex->prof_ms[5] = (float)(now_ms() - t_begin);
    }
    return SO_OK;
}

}  // namespace

namespace so {
int extractor_device_view(so_extractor* ex, ExtractorDeviceView* out) {
    if (!ex || !out) return SO_ERR_INVALID_ARG;
    out->stream = ex->stream;
    out->device = ex->cfg.device;
    out->capacity = so_extractor_capacity(ex);
    out->nlevels = ex->cfg.nlevels;
    for (int l = 0; l < kMaxLevels; l++) out->scale[l] = l < ex->cfg.nlevels ? ex->scale[l] : 0.f;
    out->meta = nullptr;
    out->angle = nullptr;
    out->desc = nullptr;
    out->total = nullptr;
    if (!ex->allocated) return SO_OK;
    if (ex->device_qt && ex->P.total_tiles > 0) {
        out->meta = ex->dev_out.meta;
        out->angle = ex->dev_out.angle;
        out->desc = ex->dev_out.desc;
        out->total = ex->dev_out.total;
    } else {  // host quadtree: survivors, descriptors and angles live in host-mapped memory
        out->meta = ex->h_sel_dev;
        out->angle = reinterpret_cast<const float*>(ex->h_desc_dev + (size_t)ex->out_capacity * 32);
        out->desc = ex->h_desc_dev;
        out->total = ex->h_total_dev;
    }
    return SO_OK;
}

void extractor_set_graph_tail(so_extractor* ex, void* owner, uint64_t revision, ExtractorTailFn fn) {
    if (!ex) return;
    ex->tail_owner = owner;
    ex->tail_revision = owner ? revision : 0;
    ex->tail_fn = owner ? fn : nullptr;
}

void extractor_release_graph_tail(so_extractor* ex, void* owner) {
    if (!ex || !owner) return;
    if (ex->tail_owner == owner) extractor_set_graph_tail(ex, nullptr, 0, nullptr);
    (void)hipSetDevice(ex->cfg.device);
    if (ex->stream) (void)hipStreamSynchronize(ex->stream);
    drop_frame_graphs(ex, owner);
}

bool extractor_tail_launched(const so_extractor* ex) { return ex && ex->tail_launched; }
}  // namespace so

// ---- several extractors, one chain of launches (include/swarmorb.h: so_extractor_group) -----------------------
struct so_extractor_group {
    std::vector<so_extractor*> m;
    int device = 0;
    hipStream_t stream = nullptr;          // the first member's
    ExtractBatchMember* d_members = nullptr;
    std::vector<ExtractBatchMember> h_members;
    // Host-mapped, rewritten per submit and read by the chain's first kernel only (which leaves what the later kernels need in
    // d_members): this frame's images as the device sees them - null: the member sits this chain out - and which of the member's
    // Frame-constructor launches rides at the end of the chain.  One captured chain therefore serves every set of members and
    // every combination of the frames they rotate.
    const uint8_t** h_srcs = nullptr;
    const uint8_t** h_srcs_dev = nullptr;
    int32_t* h_sel = nullptr;
    int32_t* h_sel_dev = nullptr;
    // The Frame-constructor launches (so_dframe_group_submit: the tracking loop rotates three device-resident frames per agent):
    // kRot rows per member in HBM, a row per frame handle the member has been submitted with (least recently used row replaced).
    static constexpr int kRot = 4;
    FramePrepareArgs* d_prep = nullptr;    // members x kRot
    std::vector<FramePrepareArgs> h_prep;  // what the rows hold
    std::vector<unsigned long long> prep_use;  // 0: row never written
    struct Slot {                          // index: rows16 * 2 + with_prep
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
    };
    std::vector<Slot> slots;
    unsigned long long use_clock = 0;
    bool graph_failed = false;
    hipEvent_t done = nullptr;             // behind the last chain: h_srcs / h_sel / a prep row may be rewritten once it has fired
    bool launched = false;
    int width = 0, height = 0;
};

namespace {

void slot_drop_graph(so_extractor_group::Slot& sl) {
    if (sl.exec) (void)hipGraphExecDestroy(sl.exec);
    if (sl.graph) (void)hipGraphDestroy(sl.graph);
    sl.exec = nullptr;
    sl.graph = nullptr;
}
void group_drop_graph(so_extractor_group* g) {
    for (auto& sl : g->slots) slot_drop_graph(sl);
}

bool same_config(const so_extractor_config& a, const so_extractor_config& b) {
    return a.nfeatures == b.nfeatures && a.scale_factor == b.scale_factor && a.nlevels == b.nlevels &&
           a.ini_th_fast == b.ini_th_fast && a.min_th_fast == b.min_th_fast && a.device == b.device;
}

}  // namespace

namespace so {

int extractor_group_size(const so_extractor_group* g) { return g ? (int)g->m.size() : 0; }
so_extractor* extractor_group_member(const so_extractor_group* g, int i) { return g->m[(size_t)i]; }

// Sizes every member for w x h frames (what the first submit of a lone extractor does): the device-resident frames of a
// group need the members' capacities and output buffers before the first batched launch.
int extractor_group_prepare(so_extractor_group* g, int w, int h) {
    if (!g || w <= 0 || h <= 0) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(g->device));
    bool fresh = g->h_members.empty();
    for (so_extractor* ex : g->m) {
        if (!ex->allocated) {
            const int rc = allocate(ex, w, h);
            if (rc) return rc;
            fresh = true;
        } else if (ex->width != w || ex->height != h) {
            last_error_ref() = "image size changed between frames";
            return SO_ERR_SIZE_CHANGED;
        }
        if (!(ex->device_qt && ex->P.total_tiles > 0)) {
            last_error_ref() = "so_extractor_group: every member must run the device quadtree (no SWARMORB_HOST_QUADTREE, quotas <= 1020 "
                               "per level) on images large enough for FAST tiles";
            return SO_ERR_INVALID_ARG;
        }
    }
    if (!fresh) return SO_OK;
    g->width = w;
    g->height = h;
    g->h_members.assign(g->m.size(), ExtractBatchMember{});
    for (size_t i = 0; i < g->m.size(); i++) {
        so_extractor* ex = g->m[i];
        ExtractBatchMember& M = g->h_members[i];
        M.P = ex->P;
        for (int l = 0; l < kMaxLevels; l++) M.qt.n_target[l] = l < ex->P.nlevels ? ex->features_per_level[l] : 0;
        M.qt.sel_stride = ex->qt_stride;
        M.qt_sel = ex->d_qt_sel;
        M.qt_count = ex->d_qt_count;
        M.out.desc = ex->h_desc_dev;
        M.out.angle = reinterpret_cast<float*>(ex->h_desc_dev + (size_t)ex->out_capacity * 32);
        M.out.meta = ex->h_meta_dev;
        M.out.total = ex->h_total_dev;
        M.out.dev = ex->dev_out;
    }
    SO_HIP(so::memcpy_sync(g->d_members, g->h_members.data(), sizeof(ExtractBatchMember) * g->m.size(), hipMemcpyHostToDevice));
    group_drop_graph(g);
    return SO_OK;
}

// so_extractor_group_submit; preps (one per member, or null): the members' frame_prepare launches ride at the end of the chain.
// images[i] == null: member i sits this chain out (nothing of it is read, written or expected back)
int extractor_group_submit(so_extractor_group* g, const uint8_t* const* images, int w, int h, int stride, const FramePrepareArgs* preps) {
    if (!g || !images || w <= 0 || h <= 0) return SO_ERR_INVALID_ARG;
    if (stride != w) {
        last_error_ref() = "so_extractor_group_submit: images must be tightly packed (stride == width)";
        return SO_ERR_INVALID_ARG;
    }
    const double t_begin = now_ms();
    const int n = (int)g->m.size();
    int n_in = 0;
    for (int i = 0; i < n; i++) {
        if (!images[i]) continue;
        n_in++;
        if (g->m[(size_t)i]->pending) {
            last_error_ref() = "a submitted frame has not been collected";
            return SO_ERR_INVALID_ARG;
        }
    }
    if (n_in == 0) return SO_ERR_INVALID_ARG;
    int rc = extractor_group_prepare(g, w, h);
    if (rc) return rc;
    bool rows16 = (w & 15) == 0;
    std::vector<const uint8_t*> srcs((size_t)n, nullptr);
    for (int i = 0; i < n; i++) {
        if (!images[i]) continue;
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, images[i]) != hipSuccess || !attr.devicePointer) {
            (void)hipGetLastError();
            last_error_ref() = "so_extractor_group_submit: every image must be device-visible (pinned host memory or device memory)";
            return SO_ERR_INVALID_ARG;
        }
        srcs[(size_t)i] = static_cast<const uint8_t*>(attr.devicePointer);
        const LevelDesc& L0 = g->m[(size_t)i]->P.lv[0];
        rows16 = rows16 && (reinterpret_cast<uintptr_t>(srcs[(size_t)i]) & 15) == 0 && (L0.pitch & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(L0.img) & 15) == 0;
    }
    hipStream_t s = g->stream;
    // the chain before this one has read its rows of the host-mapped tables (it went out a whole frame ago: this does not wait)
    if (g->launched) SO_HIP(hipEventSynchronize(g->done));
    const bool with_prep = preps != nullptr;
    constexpr int kRot = so_extractor_group::kRot;
    for (int i = 0; i < n; i++) {
        g->h_srcs[i] = srcs[(size_t)i];
        g->h_sel[i] = 0;
        if (!with_prep || !srcs[(size_t)i]) continue;
        // the member's row that holds this frame handle's launch (written once per handle: the arguments do not change)
        int row = -1, lru = 0;
        for (int k = 0; k < kRot; k++) {
            const size_t q = (size_t)i * kRot + (size_t)k;
            if (g->prep_use[q] && memcmp(&g->h_prep[q], &preps[i], sizeof(FramePrepareArgs)) == 0) row = k;
            if (g->prep_use[q] < g->prep_use[(size_t)i * kRot + (size_t)lru]) lru = k;
        }
        if (row < 0) {
            row = lru;
            const size_t q = (size_t)i * kRot + (size_t)row;
            g->h_prep[q] = preps[i];
            SO_HIP(so::memcpy_sync(g->d_prep + q, &preps[i], sizeof(FramePrepareArgs), hipMemcpyHostToDevice));
        }
        g->prep_use[(size_t)i * kRot + (size_t)row] = ++g->use_clock;
        g->h_sel[i] = row;
    }
    if (g->slots.empty()) g->slots.resize(4);
    so_extractor_group::Slot* sl = &g->slots[(rows16 ? 2 : 0) + (with_prep ? 1 : 0)];
    const int capacity = g->m[0]->out_capacity;
    auto chain = [&]() {
        launch_extract_batch(g->d_members, g->h_members[0], n, g->h_srcs_dev, g->h_sel_dev, w, h, rows16, capacity, s);
        if (with_prep) launch_frame_prepare_batch(g->d_prep, kRot, g->d_members, n, s);
    };
    static const bool no_graph = getenv("SWARMORB_NO_GRAPH") != nullptr;
    if (!sl->exec && !no_graph && !g->graph_failed) {
        hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            chain();
            e = hipStreamEndCapture(s, &sl->graph);
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&sl->exec, sl->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            slot_drop_graph(*sl);
            g->graph_failed = true;
        }
    }
    if (sl->exec) SO_HIP(hipGraphLaunch(sl->exec, s));
    else chain();
    SO_HIP(hipGetLastError());
    SO_HIP(hipEventRecord(g->done, s));
    g->launched = true;
    for (int i = 0; i < n; i++) {  // a member's collect waits on the member's own stream
        so_extractor* ex = g->m[(size_t)i];
        if (srcs[(size_t)i] && ex->stream != s) SO_HIP(hipStreamWaitEvent(ex->stream, g->done, 0));
    }
    const double t_enq = now_ms();
    for (int i = 0; i < n; i++) {
        so_extractor* ex = g->m[(size_t)i];
        if (!srcs[(size_t)i]) continue;
        ex->tail_owner = nullptr;
        ex->tail_revision = 0;
        ex->tail_fn = nullptr;
        ex->tail_launched = with_prep;
        ex->t_begin = t_begin;
        ex->t_enq = t_enq;
        ex->pending = 1;
        ex->pending_prof = false;
        ex->cands_on_host = false;
    }
    return SO_OK;
}

}  // namespace so

extern "C" {

int so_extractor_create(const so_extractor_config* cfg, so_extractor** out) {
    if (!cfg || !out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->nlevels < 1 || cfg->nlevels > SO_MAX_LEVELS || cfg->nfeatures < 1 || !(cfg->scale_factor > 1.0f) ||
        cfg->ini_th_fast < cfg->min_th_fast || cfg->min_th_fast < 1 || cfg->ini_th_fast > 254) {
        last_error_ref() = "bad extractor configuration";
        return SO_ERR_INVALID_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(cfg->device));
    so_extractor* ex = new so_extractor();
    ex->cfg = *cfg;
    make_tables(ex);
    hipError_t e = context_stream(cfg->device, 0, &ex->stream, &ex->owns_stream);
    if (e != hipSuccess) {
        delete ex;
        return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
    }
    for (auto& v : ex->ev) {
        e = hipEventCreate(&v);
        if (e != hipSuccess) {
            delete ex;
            return hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
        }
    }
    *out = ex;
    return SO_OK;
}

void so_extractor_destroy(so_extractor* ex) {
    if (!ex) return;
    (void)hipSetDevice(ex->cfg.device);
    if (ex->stream) (void)hipStreamSynchronize(ex->stream);
    for (void* p : ex->dev_allocs) (void)hipFree(p);
    if (ex->h_cands) (void)hipHostFree(ex->h_cands);
    if (ex->h_header) (void)hipHostFree(ex->h_header);
    if (ex->h_sel) (void)hipHostFree(ex->h_sel);
    if (ex->h_desc) (void)hipHostFree(ex->h_desc);
    if (ex->h_meta) (void)hipHostFree(ex->h_meta);
    if (ex->h_total) (void)hipHostFree(ex->h_total);
    for (auto& g : ex->graphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    for (auto& v : ex->ev)
        if (v) (void)hipEventDestroy(v);
    if (ex->owns_stream && ex->stream) (void)hipStreamDestroy(ex->stream);
    delete ex;
}

int so_extractor_capacity(const so_extractor* ex) {
    if (!ex) return 0;
    return ex->cfg.nfeatures + 3 * ex->cfg.nlevels;
}

int so_extractor_submit(so_extractor* ex, const uint8_t* image, int width, int height, int stride) {
    return run_impl(ex, image, false, width, height, stride, nullptr, nullptr, 0, nullptr, true);
}

int so_extractor_submit_device(so_extractor* ex, const uint8_t* d_image, int width, int height, int stride) {
    return run_impl(ex, d_image, true, width, height, stride, nullptr, nullptr, 0, nullptr, true);
}

int so_extractor_collect(so_extractor* ex, so_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out) {
    return collect_impl(ex, keypoints, descriptors, capacity, n_out);
}

int so_extractor_group_create(so_extractor* const* members, int n, so_extractor_group** out) {
    if (!members || n <= 0 || n > SO_EXTRACTOR_GROUP_MAX || !out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    for (int i = 0; i < n; i++) {
        if (!members[i] || !same_config(members[i]->cfg, members[0]->cfg)) {
            last_error_ref() = "so_extractor_group_create: the members must share one configuration and one device";
            return SO_ERR_INVALID_ARG;
        }
        for (int j = 0; j < i; j++)
            if (members[j] == members[i]) return SO_ERR_INVALID_ARG;
    }
    so_extractor_group* g = new so_extractor_group();
    g->m.assign(members, members + n);
    g->device = members[0]->cfg.device;
    g->stream = members[0]->stream;
    auto fail = [&](hipError_t e) {
        last_error_ref() = std::string("so_extractor_group_create: ") + hipGetErrorString(e);
        so_extractor_group_destroy(g);
        return SO_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(g->device)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&g->d_members, sizeof(ExtractBatchMember) * (size_t)n)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&g->d_prep, sizeof(FramePrepareArgs) * (size_t)n * so_extractor_group::kRot)) != hipSuccess) return fail(e);
    g->h_prep.resize((size_t)n * so_extractor_group::kRot);
    g->prep_use.assign((size_t)n * so_extractor_group::kRot, 0);
    if ((e = hipHostMalloc((void**)&g->h_srcs, sizeof(void*) * (size_t)n, hipHostMallocMapped)) != hipSuccess) return fail(e);
    if ((e = hipHostGetDevicePointer((void**)&g->h_srcs_dev, g->h_srcs, 0)) != hipSuccess) return fail(e);
    if ((e = hipHostMalloc((void**)&g->h_sel, sizeof(int32_t) * (size_t)n, hipHostMallocMapped)) != hipSuccess) return fail(e);
    if ((e = hipHostGetDevicePointer((void**)&g->h_sel_dev, g->h_sel, 0)) != hipSuccess) return fail(e);
    if ((e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming)) != hipSuccess) return fail(e);
    *out = g;
    return SO_OK;
}

void so_extractor_group_destroy(so_extractor_group* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    group_drop_graph(g);
    if (g->d_members) (void)hipFree(g->d_members);
    if (g->d_prep) (void)hipFree(g->d_prep);
    if (g->h_srcs) (void)hipHostFree(g->h_srcs);
    if (g->h_sel) (void)hipHostFree(g->h_sel);
    if (g->done) (void)hipEventDestroy(g->done);
    delete g;
}

int so_extractor_group_submit(so_extractor_group* g, const uint8_t* const* images, int width, int height, int stride) {
    return extractor_group_submit(g, images, width, height, stride, nullptr);
}

int so_extractor_wait(so_extractor* ex, int* n_out) {
    if (!ex || !n_out) return SO_ERR_INVALID_ARG;
    *n_out = 0;
    if (ex->pending == 0) {
        last_error_ref() = "so_extractor_wait without a submitted frame";
        return SO_ERR_INVALID_ARG;
    }
    if (ex->pending == 2) {  // finished inside submit (host-quadtree contexts, empty images)
        *n_out = ex->pend_n;
        return SO_OK;
    }
    SO_HIP(hipSetDevice(ex->cfg.device));
    if (ex->pending == 1) SO_HIP(hipStreamSynchronize(ex->stream));
    ex->pending = 3;  // synchronised, results still in the context's host-mapped buffers
    *n_out = std::min(*ex->h_total, ex->out_capacity);
    return SO_OK;
}

int so_extractor_quadtree_on_device(const so_extractor* ex) { return (ex && ex->allocated && ex->device_qt) ? 1 : 0; }

int so_extractor_run(so_extractor* ex, const uint8_t* image, int width, int height, int stride,
                     so_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out) {
    return run_impl(ex, image, false, width, height, stride, keypoints, descriptors, capacity, n_out);
}

int so_extractor_run_device(so_extractor* ex, const uint8_t* d_image, int width, int height, int stride,
                            so_keypoint* keypoints, uint8_t* descriptors, int capacity, int* n_out) {
    return run_impl(ex, d_image, true, width, height, stride, keypoints, descriptors, capacity, n_out);
}

int so_extractor_tables(const so_extractor* ex, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                        int32_t* features_per_level) {
    if (!ex) return SO_ERR_INVALID_ARG;
    for (int l = 0; l < ex->cfg.nlevels; l++) {
        if (scale) scale[l] = ex->scale[l];
        if (inv_scale) inv_scale[l] = ex->inv_scale[l];
        if (sigma2) sigma2[l] = ex->sigma2[l];
        if (inv_sigma2) inv_sigma2[l] = ex->inv_sigma2[l];
        if (features_per_level) features_per_level[l] = ex->features_per_level[l];
    }
    return SO_OK;
}

int so_extractor_level_size(const so_extractor* ex, int level, int* w, int* h) {
    if (!ex || !ex->allocated || level < 0 || level >= ex->cfg.nlevels || !w || !h) return SO_ERR_INVALID_ARG;
    *w = ex->P.lv[level].w;
    *h = ex->P.lv[level].h;
    return SO_OK;
}

int so_extractor_get_level(so_extractor* ex, int level, uint8_t* out, int out_bytes) {
    if (!ex || !ex->allocated || level < 0 || level >= ex->cfg.nlevels || !out) return SO_ERR_INVALID_ARG;
    const LevelDesc& L = ex->P.lv[level];
    if (out_bytes < L.w * L.h) return SO_ERR_CAPACITY;
    SO_HIP(hipSetDevice(ex->cfg.device));
    SO_HIP(hipMemcpy2DAsync(out, (size_t)L.w, L.img, (size_t)L.pitch, (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost,
                            ex->stream));
    SO_HIP(hipStreamSynchronize(ex->stream));
    return SO_OK;
}

int so_extractor_get_candidates(so_extractor* ex, int level, int16_t* xs, int16_t* ys, uint8_t* scores, int capacity,
                                int* n_out) {
    if (!ex || !ex->allocated || level < 0 || level >= ex->cfg.nlevels || !n_out) return SO_ERR_INVALID_ARG;
    if (!ex->cands_on_host) {
        // device-quadtree path: the frame never built the candidate list (the quadtree reads the keep bitmap); the
        // bitmap and the score map of the last frame are still resident, so the list is produced on demand
        if (ex->pending) {
            last_error_ref() = "so_extractor_get_candidates: a submitted frame has not been collected";
            return SO_ERR_INVALID_ARG;
        }
        SO_HIP(hipSetDevice(ex->cfg.device));
        if (ex->P.total_tiles > 0) {
            launch_emit(ex->P, ex->d_rowcount, ex->h_cands_dev, ex->h_header_dev, nullptr, ex->cand_capacity, ex->stream);
            SO_HIP(hipGetLastError());
        }
        SO_HIP(hipStreamSynchronize(ex->stream));
        ex->cands_on_host = true;
    }
    const CandidateHeader& H = *ex->h_header;
    const int n = H.count[level];
    *n_out = n;
    if (capacity < n) return SO_ERR_CAPACITY;
    const Candidate* c = ex->h_cands + H.offset[level];
    for (int i = 0; i < n; i++) {
        if (xs) xs[i] = c[i].x;
        if (ys) ys[i] = c[i].y;
        if (scores) scores[i] = (uint8_t)c[i].score;
    }
    return SO_OK;
}

int so_extractor_set_profiling(so_extractor* ex, int enabled) {
    if (!ex) return SO_ERR_INVALID_ARG;
    ex->profiling = enabled != 0;
    return SO_OK;
}

int so_extractor_get_profile(so_extractor* ex, float* ms) {
    if (!ex || !ms) return SO_ERR_INVALID_ARG;
    for (int i = 0; i < SO_EXTRACTOR_N_STAGES; i++) ms[i] = ex->prof_ms[i];
    return SO_OK;
}

}  // extern "C"
