// mock_swarmorb.cc — TEST INFRASTRUCTURE: a CPU stand-in for the 76 entry points of libswarmorb.so that the replay harness
// (swarmmap_amd/host/replay.cc + closedloop.cc) calls, so that the harness's own host code - two threads per agent, the
// hand-over between them, the map model, the window gather, the packets, so_fleet_run's lockstep - can run under
// -fsanitize=address,undefined and -fsanitize=thread IN THIS CONTAINER (no GPU; sanitizers never run on the GPU box).
// `make -C swarmmap_amd/csrc host-asan host-tsan` builds tests/cpp/replay_sanitize.cc against it.
//
// It is NOT an implementation of the operators (that is the HIP library, checked against oracle/ by the -m gpu tests): it
// models a static camera over a plane with a fixed lattice of keypoints, so that every search returns plausible, mutually
// consistent indices (keypoint k of a frame is keypoint k of every frame and of every keyframe; a map point projects onto the
// lattice cell it was created from), poses come back unchanged and every solver "converges" at once.  What matters is that
// the harness around it takes its real code paths: new points, fusions, cullings, windows, write-backs, packets.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <vector>

#include "swarmorb.h"

namespace {
constexpr int kCols = 36, kRows = 25, kN = kCols * kRows;  // the lattice: 900 keypoints
constexpr float kStepX = 20.f, kStepY = 18.f, kX0 = 20.f, kY0 = 20.f, kPlaneZ = 2.0f;
thread_local std::string g_err;

inline void lattice_xy(int k, float* x, float* y) {
    *x = kX0 + kStepX * (float)(k % kCols);
    *y = kY0 + kStepY * (float)(k / kCols);
}
inline int lattice_of(float x, float y) {
    const int cx = (int)std::lround((x - kX0) / kStepX), cy = (int)std::lround((y - kY0) / kStepY);
    if (cx < 0 || cx >= kCols || cy < 0 || cy >= kRows) return -1;
    return cy * kCols + cx;
}
inline void descriptor_of(int k, uint8_t* d) {
    uint32_t s = 2654435761u * (uint32_t)(k + 1);
    for (int i = 0; i < 32; i++) {
        s = s * 1664525u + 1013904223u;
        d[i] = (uint8_t)(s >> 24);
    }
}
}  // namespace

struct so_extractor { so_extractor_config cfg; };
struct so_extractor_group { int n; };
struct so_dframe { so_camera cam; int n = 0; bool ready = false; };
struct so_matcher {
    const so_dframe* cur = nullptr;
    const so_map* map = nullptr;
    std::vector<int32_t> slots, kp_slot, last_slots;
    std::vector<uint8_t> skip, excluded;
    int mode = 0, n_local = 0, first_slot = 0;
    float T[12];
    bool stage = false, again = false;
    const so_matcher* linked_to = nullptr;  // so_track_stage_local_map_submit_after: the bindings come from that matcher's stage
    std::vector<int32_t> edges;
    bool batching = false;
};
struct so_kframe { int n; };
struct so_map {
    std::mutex mu;
    std::vector<float> X;
    float fx = 458.f, fy = 457.f, cx = 367.f, cy = 248.f;
};
struct so_ba { float T[12]; int n = 0; };
struct so_ba_group { std::atomic<int> members{0}; };
struct so_track_group { int pending = 0; };

static int project(const so_map* m, int slot) {  // lattice cell a map point projects to (static camera at the origin), -1 none
    if (slot < 0 || (size_t)(3 * slot + 2) >= m->X.size()) return -1;
    const float* P = &m->X[3 * (size_t)slot];
    if (!(P[2] > 0.f)) return -1;
    return lattice_of(m->fx * P[0] / P[2] + m->cx, m->fy * P[1] / P[2] + m->cy);
}

extern "C" {
const char* so_last_error(void) { return g_err.c_str(); }
int so_device_host_cpus(int, int, char* cpulist, int capacity) {
    if (cpulist && capacity > 0) cpulist[0] = 0;
    return 1;  // (not SO_OK: "no placement information", the harness leaves its threads where they are)
}
// ---- extractor / device frame ----
int so_extractor_create(const so_extractor_config* cfg, so_extractor** out) { *out = new so_extractor{*cfg}; return SO_OK; }
void so_extractor_destroy(so_extractor* ex) { delete ex; }
int so_extractor_capacity(const so_extractor*) { return kN + 64; }
int so_extractor_set_profiling(so_extractor*, int) { return SO_OK; }
int so_extractor_get_profile(so_extractor*, float* ms) { for (int i = 0; i < SO_EXTRACTOR_N_STAGES; i++) ms[i] = 0.f; return SO_OK; }
int so_extractor_tables(const so_extractor* ex, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* fpl) {
    float s = 1.f;
    for (int l = 0; l < ex->cfg.nlevels; l++) {
        if (scale) scale[l] = s;
        if (inv_scale) inv_scale[l] = 1.f / s;
        if (sigma2) sigma2[l] = s * s;
        if (inv_sigma2) inv_sigma2[l] = 1.f / (s * s);
        if (fpl) fpl[l] = kN / ex->cfg.nlevels;
        s *= ex->cfg.scale_factor;
    }
    return SO_OK;
}
int so_extractor_group_create(so_extractor* const*, int n, so_extractor_group** out) { *out = new so_extractor_group{n}; return SO_OK; }
void so_extractor_group_destroy(so_extractor_group* g) { delete g; }
int so_dframe_create(so_extractor*, const so_camera* cam, so_dframe** out) { *out = new so_dframe(); (*out)->cam = *cam; return SO_OK; }
void so_dframe_destroy(so_dframe* f) { delete f; }
int so_dframe_submit(so_dframe* f, const uint8_t* image, int w, int h, int) {
    volatile uint8_t touch = image[0] ^ image[(size_t)w * h - 1];  // (the image must be readable)
    (void)touch;
    f->n = kN;
    f->ready = true;
    return SO_OK;
}
int so_dframe_submit_device(so_dframe* f, const uint8_t* img, int w, int h, int s) { return so_dframe_submit(f, img, w, h, s); }
int so_dframe_group_submit(so_extractor_group* g, so_dframe* const* frames, const uint8_t* const* images, int w, int h, int s) {
    for (int i = 0; i < g->n; i++)
        if (images[i]) so_dframe_submit(frames[i], images[i], w, h, s);  // (null: the member sits the chain out)
    return SO_OK;
}
static void bounds_of(float* b) { if (b) { b[0] = 0.f; b[1] = 752.f; b[2] = 0.f; b[3] = 480.f; } }
int so_dframe_wait(so_dframe* f, int* n_out, float* bounds4) {
    if (n_out) *n_out = f->n;
    bounds_of(bounds4);
    return SO_OK;
}
int so_dframe_collect(so_dframe* f, so_keypoint* kps, float* xy_un, uint8_t* desc, int capacity, int* n_out, float* bounds4) {
    if (capacity < f->n) return SO_ERR_CAPACITY;
    for (int k = 0; k < f->n; k++) {
        float x, y;
        lattice_xy(k, &x, &y);
        if (kps) kps[k] = so_keypoint{x, y, 31.f, (float)((k * 37) % 360), 50.f, k % 3, -1};
        if (xy_un) {
            xy_un[2 * k] = x;
            xy_un[2 * k + 1] = y;
        }
        if (desc) descriptor_of(k, desc + 32 * (size_t)k);
    }
    if (n_out) *n_out = f->n;
    bounds_of(bounds4);
    return SO_OK;
}
// ---- map table ----
int so_map_create(int, so_map** out) { *out = new so_map(); return SO_OK; }
void so_map_destroy(so_map* m) { delete m; }
int so_map_write(so_map* m, int32_t first, int32_t n, const float* Xw, const float*, const float*, const float*, const uint8_t*) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->X.size() < 3 * (size_t)(first + n)) m->X.resize(3 * (size_t)(first + n), 0.f);
    if (Xw) memcpy(&m->X[3 * (size_t)first], Xw, sizeof(float) * 3 * (size_t)n);
    return SO_OK;
}
int so_map_write_rows(so_map* m, int32_t n, const int32_t* slots, const float* Xw, const float*, const float*, const float*) {
    std::lock_guard<std::mutex> lk(m->mu);
    for (int i = 0; i < n; i++) {
        if (slots[i] < 0 || 3 * (size_t)slots[i] + 2 >= m->X.size()) return SO_ERR_INVALID_ARG;
        if (Xw) memcpy(&m->X[3 * (size_t)slots[i]], Xw + 3 * (size_t)i, 12);
    }
    return SO_OK;
}
// ---- matcher ----
int so_matcher_create(int, so_matcher** out) { *out = new so_matcher(); return SO_OK; }
void so_matcher_destroy(so_matcher* m) { delete m; }
int so_matcher_last_kernel_ms(so_matcher*, float* ms) { *ms = 0.01f; return SO_OK; }
int so_matcher_last_stats(so_matcher*, double* s4) { s4[0] = s4[1] = 0.001; s4[2] = 1; s4[3] = 0; return SO_OK; }
int so_matcher_reserve(so_matcher*, int32_t) { return SO_OK; }
int so_matcher_set_profiling(so_matcher*, int) { return SO_OK; }
int so_matcher_private_stream(so_matcher*) { return SO_OK; }
int so_matcher_share_stream(so_matcher*, const so_matcher*) { return SO_OK; }
int so_map_share_stream(so_map*, const so_matcher*) { return SO_OK; }
uint64_t so_matcher_stream_id(const so_matcher*) { return 1; }
int so_matcher_set_track_group(so_matcher*, so_track_group*) { return SO_OK; }
int so_matcher_batch_begin(so_matcher* m) { m->batching = true; return SO_OK; }
int so_matcher_batch_end(so_matcher* m) { m->batching = false; return SO_OK; }
int so_matcher_batch_abort(so_matcher* m) { m->batching = false; return SO_OK; }
int so_track_group_create(int, so_track_group** out) { *out = new so_track_group(); return SO_OK; }
void so_track_group_destroy(so_track_group* g) { delete g; }
int so_track_group_pending(so_track_group*) { return 0; }
int so_track_group_launch(so_track_group*) { return SO_OK; }
int so_track_group_last_kernel_ms(so_track_group*, float* a, float* b) { if (a) *a = 0.01f; if (b) *b = 0.05f; return SO_OK; }

// keypoint k of the current frame is keypoint k of the last one
int so_track_search_last_frame(so_matcher*, const so_dframe* cur, const uint8_t* excluded, const so_dframe* last, const so_map*, const float*,
                               const int32_t* last_slot, const uint8_t*, float, int, int32_t* k2l, int32_t* nm) {
    int n = 0;
    for (int k = 0; k < cur->n; k++) {
        const bool ok = k < last->n && last_slot[k] >= 0 && !(excluded && excluded[k]) && (k % 11) != 0 && !(last_slot[k] < kN && (k % 7) == 3);
        k2l[k] = ok ? k : -1;
        n += ok;
    }
    *nm = n;
    return SO_OK;
}
int so_track_search_last_frame_submit(so_matcher* m, const so_dframe* cur, const uint8_t*, const so_dframe* last, const so_map*, const float*,
                                      const int32_t* last_slot, float) {
    m->cur = cur;
    m->slots.assign(last_slot, last_slot + last->n);
    m->mode = 2;
    return SO_OK;
}
int so_track_search_last_frame_wait(so_matcher* m, const uint8_t*, int, int32_t* k2l, int32_t* nm) {
    int n = 0;
    for (int k = 0; k < m->cur->n; k++) {
        const bool ok = k < (int)m->slots.size() && m->slots[(size_t)k] >= 0 && (k % 11) != 0 && !(m->slots[(size_t)k] < kN && (k % 7) == 3);
        k2l[k] = ok ? k : -1;
        n += ok;
    }
    *nm = n;
    m->mode = 0;
    return SO_OK;
}
// a local point lands on the lattice cell it projects to; the first one to get there takes the keypoint
static void local_search(const so_dframe* cur, const uint8_t* excluded, const so_map* map, int n_local, const int32_t* local_slot, int first_slot,
                         const uint8_t* skip, uint8_t* in_view, int32_t* k2m, int32_t* nm) {
    for (int k = 0; k < cur->n; k++) k2m[k] = -1;
    int n = 0;
    std::lock_guard<std::mutex> lk(const_cast<so_map*>(map)->mu);
    for (int i = 0; i < n_local; i++) {
        if (in_view) in_view[i] = 0;
        if (skip && skip[i]) continue;
        const int slot = local_slot ? local_slot[i] : first_slot + i;
        const int cell = project(map, slot);
        if (cell < 0 || cell >= cur->n) continue;
        if (in_view) in_view[i] = 1;
        // a seventh of the INITIAL map's points is never found again: their keypoints stay free, local mapping triangulates new
        // points there (which are found), and the initial ones are culled by their found ratio
        if (slot < kN && (cell % 7) == 3) continue;
        if ((excluded && excluded[cell]) || k2m[cell] >= 0) continue;
        k2m[cell] = i;
        n++;
    }
    *nm = n;
}
int so_track_search_local_map(so_matcher*, const so_dframe* cur, const uint8_t* excluded, const so_map* map, const float*, int32_t n_local,
                              const int32_t* local_slot, int32_t first_slot, const uint8_t* skip, const uint8_t*, float, float, float, float,
                              uint8_t* in_view, int32_t* k2m, int32_t* nm) {
    local_search(cur, excluded, map, n_local, local_slot, first_slot, skip, in_view, k2m, nm);
    return SO_OK;
}
int so_track_search_local_map_submit(so_matcher* m, const so_dframe* cur, const uint8_t* excluded, const so_map* map, const float*, int32_t n_local,
                                     const int32_t* local_slot, int32_t first_slot, const uint8_t* skip, float, float, float, float) {
    m->cur = cur; m->map = map; m->n_local = n_local; m->first_slot = first_slot; m->mode = 3;
    if (local_slot) m->slots.assign(local_slot, local_slot + n_local); else m->slots.clear();
    if (skip) m->skip.assign(skip, skip + n_local); else m->skip.clear();
    if (excluded) m->excluded.assign(excluded, excluded + cur->n); else m->excluded.clear();
    return SO_OK;
}
int so_track_search_local_map_wait(so_matcher* m, const uint8_t*, uint8_t* in_view, int32_t* k2m, int32_t* nm) {
    local_search(m->cur, m->excluded.empty() ? nullptr : m->excluded.data(), m->map, m->n_local, m->slots.empty() ? nullptr : m->slots.data(),
                 m->first_slot, m->skip.empty() ? nullptr : m->skip.data(), in_view, k2m, nm);
    m->mode = 0;
    return SO_OK;
}
// stages: the same searches with the pose (unchanged) and the edge list behind them
int so_track_stage_last_frame_submit(so_matcher* m, const so_dframe* cur, const so_dframe* last, const so_map* map, const float* T, const int32_t* last_slot,
                                     float, int, const float*, const float*) {
    m->cur = cur; m->map = map; m->stage = true; m->again = false; m->mode = 2;
    m->slots.assign(last_slot, last_slot + last->n);
    memcpy(m->T, T, 48);
    return SO_OK;
}
int so_track_stage_local_map_submit(so_matcher* m, const so_dframe* cur, const int32_t* kp_slot, int, const so_map* map, const float* T, int32_t n_local,
                                    const int32_t* local_slot, int32_t first_slot, const uint8_t* skip, float, float, float, float, const float*, const float*) {
    m->cur = cur; m->map = map; m->stage = true; m->again = false; m->mode = 3; m->n_local = n_local; m->first_slot = first_slot;
    m->kp_slot.assign(kp_slot, kp_slot + cur->n);
    if (local_slot) m->slots.assign(local_slot, local_slot + n_local); else m->slots.clear();
    if (skip) m->skip.assign(skip, skip + n_local); else m->skip.clear();
    memcpy(m->T, T, 48);
    return SO_OK;
}
int so_track_stage_local_map_submit_after(so_matcher* m, so_matcher* first, const so_dframe* cur, const so_map* map, int32_t n_local, const int32_t* local_slot,
                                          int32_t first_slot, const uint8_t* skip, float, float, float, float, const float*, const float*) {
    if (!first->stage || first->mode != 2) return SO_ERR_INVALID_ARG;
    m->cur = cur; m->map = map; m->stage = true; m->again = false; m->mode = 3; m->n_local = n_local; m->first_slot = first_slot;
    m->linked_to = first;
    m->kp_slot.assign((size_t)cur->n, -1);
    if (local_slot) m->slots.assign(local_slot, local_slot + n_local); else m->slots.clear();
    if (skip) m->skip.assign(skip, skip + n_local); else m->skip.clear();
    memcpy(m->T, first->T, 48);
    return SO_OK;
}
int so_track_stage_set_start_pose(so_matcher* m, const float* T) { memcpy(m->T, T, 48); return SO_OK; }
int so_track_stage_pose_again_submit(so_matcher* m, const float* T) {
    if (m->edges.empty()) return SO_ERR_INVALID_ARG;
    m->again = true;
    memcpy(m->T, T, 48);
    return SO_OK;
}
int so_track_stage_wait(so_matcher* m, int32_t* k2q, int32_t* nm, uint8_t* in_view, int32_t* n_edges, int32_t* edge_kp, uint8_t* edge_outlier,
                        float* T_out, int32_t* n_inliers, int32_t* info2) {
    const int n = m->cur->n;
    if (!m->again) {
        m->edges.clear();
        if (m->mode == 2) {
            m->last_slots = m->slots;
            int c = 0;
            for (int k = 0; k < n; k++) {
                const bool ok = k < (int)m->slots.size() && m->slots[(size_t)k] >= 0 && (k % 11) != 0 && !(m->slots[(size_t)k] < kN && (k % 7) == 3);
                k2q[k] = ok ? k : -1;
                c += ok;
                if (ok) m->edges.push_back(k);
            }
            *nm = c;
        } else {
            if (m->linked_to) {  // the bindings the first stage left: its matches minus its pose's outliers; already-bound local points are skipped
                const so_matcher* f = m->linked_to;
                for (int e : f->edges)
                    if ((e % 97) != 5 && e < (int)f->last_slots.size()) m->kp_slot[(size_t)e] = f->last_slots[(size_t)e];
                if (m->skip.empty()) m->skip.assign((size_t)m->n_local, 0);
                for (int i = 0; i < m->n_local; i++) {
                    const int slot = m->slots.empty() ? m->first_slot + i : m->slots[(size_t)i];
                    for (int k = 0; k < n; k++)
                        if (m->kp_slot[(size_t)k] == slot) { m->skip[(size_t)i] = 1; break; }
                }
                m->linked_to = nullptr;
            }
            std::vector<uint8_t> excl((size_t)n);
            for (int k = 0; k < n; k++) excl[(size_t)k] = m->kp_slot[(size_t)k] >= 0;
            local_search(m->cur, excl.data(), m->map, m->n_local, m->slots.empty() ? nullptr : m->slots.data(), m->first_slot,
                         m->skip.empty() ? nullptr : m->skip.data(), in_view, k2q, nm);
            for (int k = 0; k < n; k++)
                if (m->kp_slot[(size_t)k] >= 0 || k2q[k] >= 0) m->edges.push_back(k);
        }
    }
    *n_edges = (int)m->edges.size();
    int inl = 0;
    for (size_t e = 0; e < m->edges.size(); e++) {
        edge_kp[e] = m->edges[e];
        edge_outlier[e] = (m->edges[e] % 97) == 5 ? 1 : 0;  // a few outliers
        inl += !edge_outlier[e];
    }
    memcpy(T_out, m->T, 48);
    T_out[3] += 0.002f;  // the "optimised" pose drifts sideways: keyframes get a baseline, CreateNewMapPoints has neighbours to search
    *n_inliers = inl;
    if (info2) { info2[0] = 10; info2[1] = 12; }
    m->mode = 0;
    m->again = false;
    return SO_OK;
}
int so_track_stage_last_pose_kernel_ms(so_matcher*, float* ms) { *ms = 0.05f; return SO_OK; }
// ---- keyframes, triangulation, fusion ----
int so_hamming_top2(so_matcher*, const uint8_t* A, int32_t na, const uint8_t*, int32_t nb, int32_t* bi, int32_t* bd, int32_t* sd) {
    for (int i = 0; i < na; i++) {
        bi[i] = A[32 * (size_t)i] % nb;
        if (bd) bd[i] = 20;
        if (sd) sd[i] = 40;
    }
    return SO_OK;
}
int so_kframe_create(so_matcher*, const so_frame_view* KF, const so_featvec*, const float*, so_kframe** out) { *out = new so_kframe{KF->n}; return SO_OK; }
void so_kframe_destroy(so_kframe* k) { delete k; }
int so_search_for_triangulation_kframes(so_matcher*, const so_kframe* kf1, const uint8_t* free1, int32_t nn, const so_tri_neighbour* nb, int) {
    for (int j = 0; j < nn; j++) {
        int c = 0;
        for (int i = 0; i < kf1->n; i++) {
            const bool ok = free1[i] && i < nb[j].kf2->n && nb[j].free2[i] && ((i + j) % 4) == 0;  // a quarter of the free pairs per neighbour
            nb[j].matches12[i] = ok ? i : -1;
            c += ok;
        }
        *nb[j].nmatches = c;
    }
    return SO_OK;
}
int so_search_for_triangulation_kframe(so_matcher*, int32_t n1, const float*, const float*, const float*, const uint8_t*, const uint8_t* free1, const so_featvec*,
                                       const so_kframe* kf2, const uint8_t* free2, const float*, float, float, int, int32_t* m12, int32_t* nm) {
    int c = 0;
    for (int i = 0; i < n1; i++) {
        const bool ok = free1[i] && i < kf2->n && free2[i] && (i % 4) == 0;
        m12[i] = ok ? i : -1;
        c += ok;
    }
    *nm = c;
    return SO_OK;
}
int so_search_for_triangulation(so_matcher*, int32_t n1, const float*, const float*, const float*, const uint8_t*, const uint8_t* free1, const so_featvec*, int32_t n2,
                                const float*, const float*, const int32_t*, const float*, const uint8_t*, const uint8_t* free2, const so_featvec*, const float*, float,
                                float, const float*, const float*, int32_t, int, int32_t* m12, int32_t* nm) {
    int c = 0;
    for (int i = 0; i < n1; i++) {
        const bool ok = free1[i] && i < n2 && free2[i] && (i % 4) == 0;
        m12[i] = ok ? i : -1;
        c += ok;
    }
    *nm = c;
    return SO_OK;
}
static void backproject(const so_tri_keyframe* kf, const float* xy, float* X) {  // onto the plane, camera at the origin
    X[0] = (xy[0] - kf->cx) * kf->invfx * kPlaneZ;
    X[1] = (xy[1] - kf->cy) * kf->invfy * kPlaneZ;
    X[2] = kPlaneZ;
}
int so_triangulate_new_points(so_matcher*, const so_tri_keyframe* kf1, int32_t, const so_tri_keyframe*, float, int32_t n, const int32_t*, const float* xy1,
                              const int32_t*, const float*, const int32_t*, uint8_t* ok, float* x3D, float* normal, float* max_dist, float* min_dist) {
    for (int q = 0; q < n; q++) {
        ok[q] = (q % 5) != 0;
        backproject(kf1, xy1 + 2 * (size_t)q, x3D + 3 * (size_t)q);
        const float* X = x3D + 3 * (size_t)q;
        const float d = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
        if (normal) { normal[3 * (size_t)q] = X[0] / d; normal[3 * (size_t)q + 1] = X[1] / d; normal[3 * (size_t)q + 2] = X[2] / d; }
        if (max_dist) max_dist[q] = 1.5f * d;
        if (min_dist) min_dist[q] = 0.4f * d;
    }
    return SO_OK;
}
int so_triangulate_matches(so_matcher* m, const so_tri_keyframe* kf1, int32_t a, const so_tri_keyframe* kf2, float r, int32_t n, const int32_t* of, const float* xy1,
                           const int32_t* o1, const float* xy2, const int32_t* o2, uint8_t* ok, float* x3D) {
    return so_triangulate_new_points(m, kf1, a, kf2, r, n, of, xy1, o1, xy2, o2, ok, x3D, nullptr, nullptr, nullptr);
}
int so_fuse_kframe_map(so_matcher*, const so_kframe* KF, const so_camera*, const float*, float, const float*, const so_map* map, int32_t n, const int32_t* slots,
                       const uint8_t* valid, float, int32_t* best_idx, int32_t* best_dist, int32_t* n_fused, const so_window_queries*) {
    int c = 0;
    std::lock_guard<std::mutex> lk(const_cast<so_map*>(map)->mu);
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
        if (!valid[i]) continue;
        const int cell = project(map, slots[i]);
        if (cell < 0 || cell >= KF->n || (cell % 3) == 1) continue;
        best_idx[i] = cell;
        best_dist[i] = 15;
        c++;
    }
    if (n_fused) *n_fused = c;
    return SO_OK;
}
int so_fuse_kframe(so_matcher*, const so_kframe*, const so_camera*, const float*, float, const float*, const so_mappoint_view*, float, int32_t*, int32_t*, int32_t* nf,
                   const so_window_queries*) { if (nf) *nf = 0; return SO_OK; }
int so_fuse(so_matcher*, const so_frame_view*, const so_camera*, const float*, float, const float*, const so_mappoint_view*, float, int32_t*, int32_t*, int32_t* nf,
            const so_window_queries*) { if (nf) *nf = 0; return SO_OK; }
static void und_one(const float* X, const float* O, float* normal, float* mx, float* mn) {
    const float v[3] = {X[0] - O[0], X[1] - O[1], X[2] - O[2]};
    const float d = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-6f;
    for (int q = 0; q < 3; q++) normal[q] = v[q] / d;
    *mx = 1.5f * d;
    *mn = 0.4f * d;
}
int so_update_normal_and_depth(so_matcher*, int32_t n, const int32_t*, const float*, const float* Xw, const float* ref_Ow, const float*, const float*, float* normal,
                               float* mx, float* mn) {
    for (int i = 0; i < n; i++) und_one(Xw + 3 * (size_t)i, ref_Ow + 3 * (size_t)i, normal + 3 * (size_t)i, mx + i, mn + i);
    return SO_OK;
}
int so_update_normal_and_depth_indexed(so_matcher*, int32_t n, const int32_t* off, const int32_t* obs_kf, int32_t n_kf, const float* kf_Ow, const float* Xw,
                                       const int32_t* ref_kf, const float*, const float*, float* normal, float* mx, float* mn) {
    for (int i = 0; i < n; i++) {
        for (int k = off[i]; k < off[i + 1]; k++)
            if (obs_kf[k] < 0 || obs_kf[k] >= n_kf) return SO_ERR_INVALID_ARG;  // (every observer must be a row of the table)
        if (ref_kf[i] < 0 || ref_kf[i] >= n_kf) return SO_ERR_INVALID_ARG;
        und_one(Xw + 3 * (size_t)i, kf_Ow + 3 * (size_t)ref_kf[i], normal + 3 * (size_t)i, mx + i, mn + i);
    }
    return SO_OK;
}
// ---- optimisers: everything converged already ----
int so_ba_create(int, so_ba** out) { *out = new so_ba(); return SO_OK; }
void so_ba_destroy(so_ba* b) { delete b; }
int so_ba_group_create(int, double, so_ba_group** out) { *out = new so_ba_group(); return SO_OK; }
void so_ba_group_destroy(so_ba_group* g) { delete g; }  // (the mock's solvers hold no reference to it)
int so_ba_set_group(so_ba*, so_ba_group*) { return SO_OK; }
void so_ba_options_local(so_ba_options* o) { o->its_stage1 = 5; o->its_stage2 = 10; o->robust = 1; o->huber_delta = 2.4477f; o->chi2_threshold = 5.991f; }
int so_bundle_adjust_set_solve_timing(so_ba*, int) { return SO_OK; }
int so_bundle_adjust(so_ba*, const so_ba_problem* p, const so_ba_options*, const volatile uint8_t* stop, float* T_out, float* X_out, uint8_t* outl, double* chi2,
                     so_ba_info* info) {
    static const int nap_us = getenv("MOCK_BA_SLEEP_US") ? atoi(getenv("MOCK_BA_SLEEP_US")) : 0;  // a job longer than five ticks: agents sit ticks out
    if (nap_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(nap_us));
    for (int e = 0; e < p->n_edges; e++) {  // (every index the harness gathered must be in range)
        if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses || p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points) return SO_ERR_INVALID_ARG;
        if (outl) outl[e] = (e % 211) == 7;
        if (chi2) chi2[e] = 1.0;
    }
    memcpy(T_out, p->Tcw, sizeof(float) * 12 * (size_t)p->n_poses);
    memcpy(X_out, p->Xw, sizeof(float) * 3 * (size_t)p->n_points);
    if (info) {
        memset(info, 0, sizeof(*info));
        info->iterations_stage1 = 5; info->iterations_stage2 = 3; info->lm_trials = 8;
        info->aborted = stop && __atomic_load_n(const_cast<const uint8_t*>(stop), __ATOMIC_RELAXED) ? 1 : 0;
        info->gpu_ms = 0.5f;
        int nf = 0;
        for (int i = 0; i < p->n_poses; i++) nf += !p->fixed[i];
        info->n_free_keyframes = nf;
    }
    return SO_OK;
}
int so_pose_optimization_set_timing(so_ba*, int) { return SO_OK; }
int so_pose_optimization_last_kernel_ms(so_ba*, float* ms) { *ms = 0.05f; return SO_OK; }
int so_pose_optimization_submit(so_ba* b, const float* T, const float*, int32_t n, const float*, const float*, const float*) { memcpy(b->T, T, 48); b->n = n; return SO_OK; }
int so_pose_optimization_wait(so_ba* b, float* T_out, uint8_t* outlier, int32_t* n_inliers, int32_t* info) {
    memcpy(T_out, b->T, 48);
    T_out[3] += 0.002f;
    int inl = 0;
    for (int i = 0; i < b->n; i++) { outlier[i] = (i % 97) == 5; inl += !outlier[i]; }
    *n_inliers = inl;
    if (info) { info[0] = 10; info[1] = 12; }
    return SO_OK;
}
int so_pose_optimization_batch(so_ba* b, int32_t n, const so_pose_problem* P) {
    for (int i = 0; i < n; i++) {
        so_pose_optimization_submit(b, P[i].Tcw12, P[i].intr, P[i].n, P[i].Xw, P[i].obs, P[i].inv_sigma2);
        so_pose_optimization_wait(b, P[i].Tcw_out12, P[i].outlier, P[i].n_inliers, P[i].info);
    }
    return SO_OK;
}
}  // extern "C"
