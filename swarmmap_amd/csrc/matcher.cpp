// matcher.cpp — host driver + C ABI of the Hamming matcher (include/swarmorb.h).
//
// Replaces the tracking-thread routines of ORB_SLAM2::ORBmatcher (code/src/ORBmatcher.cc).  Per call:
//   flatten the frame into grid-traversal order and the queries into ONE pinned staging block -> one H2D copy ->
//   ONE top-K launch for all queries, K-lists written straight into host-mapped memory -> the reference's
//   order-dependent resolve on the host.  When a query's K-list is exhausted by keypoints that
//   earlier queries took (possible only if more than K candidates were in its window) that single query is
//   re-evaluated on the GPU with the current "taken" gate, so distances never come from the CPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "ba_device.h"
#include "dframe_internal.h"
#include "match_device.h"
#include "so_common.h"

using namespace so;

namespace {

constexpr int kGridCols = so::kMatchGridCols, kGridRows = so::kMatchGridRows;
constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // code/src/ORBmatcher.cc:37-39

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SO_OK;
        if (p) SO_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 256;
        SO_HIP(hipMalloc(&p, want));
        cap = want;
        return SO_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SO_OK;
        if (p) SO_HIP(hipHostFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 256;
        SO_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return SO_OK;
    }
    // grow, keeping the first `keep` bytes
    int ensure_keep(size_t bytes, size_t keep) {
        if (bytes <= cap) return SO_OK;
        void* np_ = nullptr;
        const size_t want = bytes + bytes / 2 + 256;
        SO_HIP(hipHostMalloc(&np_, want, hipHostMallocDefault));
        if (p && keep) memcpy(np_, p, keep < cap ? keep : cap);
        if (p) SO_HIP(hipHostFree(p));
        p = np_;
        cap = want;
        return SO_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Host memory the kernel writes directly (results are small and read once by the host).
struct MappedBuf {
    void* p = nullptr;
    void* dev = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SO_OK;
        if (p) SO_HIP(hipHostFree(p));
        p = dev = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 256;
        SO_HIP(hipHostMalloc(&p, want, hipHostMallocMapped));
        SO_HIP(hipHostGetDevicePointer(&dev, p, 0));
        cap = want;
        return SO_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = dev = nullptr;
        cap = 0;
    }
};

struct View {
    void* p = nullptr;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// A keyframe's matcher-side data resident in HBM (so_kframe_create): its keypoints in GetFeaturesInArea order and,
// when a feature vector came with it, in vocabulary-node order.  Immutable after creation.
struct so_kframe {
    int device = 0, n = 0;
    uint8_t* d = nullptr;
    size_t d_cap = 0;
    // grid layout
    size_t g_oct = 0, g_desc = 0, g_cols = 0, g_end = 0;
    int n_grid = 0;
    std::vector<int> perm_grid;
    float min_x = 0.f, max_x = 0.f, min_y = 0.f, max_y = 0.f, grid_inv_w = 0.f, grid_inv_h = 0.f, grid_min_y = 0.f;
    // node layout (behind the grid layout)
    bool has_nodes = false;
    size_t n_xy = 0, n_oct = 0, n_desc = 0;
    int n_node = 0;
    std::vector<int> perm_node;
    std::vector<int32_t> node_id, node_off;  // offsets relative to the first node's first feature
    std::vector<float> angle;                // mvKeysUn[i].angle by keypoint index
    float scale[8] = {0}, sigma2[8] = {0};
    int nlevels = 0;
};

struct so_track_group;
struct LinkRec;
static void link_rec_free(LinkRec* r);  // (defined behind the type, further down)

struct so_matcher {
    std::vector<int> scratch_rot_item, scratch_rot_b;  // rotation-histogram bookkeeping of the resolve loops
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float last_ms = 0.f;
    bool profile = true;  // event-timed kernels (so_matcher_last_kernel_ms)
    // host view of the last call: [0] ms enqueueing copies + launches, [1] ms blocked in stream syncs,
    // [2] kernel launches (1 + exact re-runs), [3] bytes staged host -> device
    double stat[4] = {0, 0, 0, 0};

    // One pinned staging buffer and its device twin per call: [xy | octave | desc | limit | queries | qdesc],
    // moved with a single H2D copy; the K-lists come back through host-mapped memory the kernel writes.
    PinBuf h_in;
    DevBuf d_in;
    MappedBuf h_out;
    size_t off_oct = 0, off_desc = 0, off_limit = 0, off_cols = 0, frame_end = 0, off_q = 0, off_qdesc = 0;
    size_t dirty_from = SIZE_MAX;  // staging bytes from here on are newer than the device copy
    View h_q, h_qdesc, h_keys, h_count;
    DevBuf d_A, d_B, d_res;
    PinBuf h_res;
    // exact re-run of ONE query (rerun_single): its own tiny staging, so the batch's queries and K-lists stay valid
    PinBuf h_rq;
    DevBuf d_rq;
    MappedBuf h_rout;

    float inv_sigma2[8] = {0}, sigma2[8] = {0}, scale[8] = {0}, ex = 0.f, ey = 0.f;  // gates of the current candidates
    int n_cand = 0;           // keypoints that are inside the grid (PosInGrid true)
    bool has_limit = false;
    bool has_cols = false;    // candidates are in grid-traversal order and the column table is staged
    int resident_n = -1;      // F->n of the grid-ordered frame resident in the staging / device blocks (-1: none)
    bool reuse_next = false;  // so_matcher_reuse_frame: the next call's frame is the resident one
    // identity of the resident frame (so_matcher_reuse_frame only applies when the next view is this very frame)
    const void *res_desc = nullptr, *res_x = nullptr, *res_y = nullptr;
    // candidates read in place from a device-resident frame (so_dframe) instead of the staged upload
    const so_dframe* src = nullptr;
    size_t off_slot = 0, off_skip = 0, track_end = 0;
    // a tracking search between its submit and its wait (so_track_search_*_submit / _wait)
    struct PendingTrack {
        int mode = 0;  // 0: none, 2: last-frame search, 3: local-map search
        bool empty = true;  // nothing was launched (no queries / no keypoints)
        TrackQuerySrc T{};
        int nq = 0;
        size_t c8_off = 0, view_off = 0;
        const so_dframe* cur = nullptr;
        const so_dframe* last = nullptr;
        const uint8_t* cur_excluded = nullptr;  // caller memory: must stay valid until the wait
        float nn_ratio = 0.f;
        // the map whose tables this search reads: held shared (so_map::grow_mu) from the submit to the end of the wait, so
        // that another thread's append (a local-mapping thread adding triangulated points) cannot move them meanwhile
        const so_map* held_map = nullptr;
        std::chrono::steady_clock::time_point t_launched;
    } pend;
    float min_x = 0.f, min_y = 0.f, grid_inv_w = 0.f, grid_inv_h = 0.f;
    float grid_min_y = 0.f, grid_min_x = 0.f;  // origin the resident candidates' cells were assigned with
    std::vector<int> perm;     // rank -> keypoint index
    // ---- a tracking stage resolved on the device (so_track_stage_*): search -> track_resolve_kernel -> pose_opt_chain_kernel
    DevBuf d_chain;     // [K-lists | counts | edge keypoints | edge slots | head]
    MappedBuf h_chain;  // [kp_to_q | edge keypoints | head | pose 64 | info 16 | outlier flags | bindings on entry]
    uint32_t* keys_dev_override = nullptr;  // where the next tracking search writes its K-lists instead of h_out
    uint8_t* cnt8_dev_override = nullptr;   // ... and its one-byte candidate counts
    int32_t* slot_out_override = nullptr;   // ... and, per query, the map slot it searched with
    DevBuf d_kpslot;                        // the frame's bindings behind the last stage, by keypoint index (read by the next stage)
    const so_dframe* kpslot_frame = nullptr;  // the frame they belong to (null: not valid)
    uint64_t kpslot_generation = 0;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    int chain_seq = 0;
    int chain_range[2] = {0, 0};  // PoseOptimization variant (0: <= 1024 edges, 1: more) each stage kind needed last time
    struct ChainPending {
        bool active = false;
        int kind = 0;  // 0: last-frame stage, 1: local-map stage, 2: the pose again over the same edges
        int n_kp = 0, nq = 0, seq = 0;
        bool events = false, valid_edges = false;
        bool grouped = false;  // launched by the matcher's so_track_group: no per-member events
        bool linked = false;   // so_track_stage_local_map_submit_after
        size_t h_k2q = 0, h_ekp = 0, h_head = 0, h_pose = 0, h_info = 0, h_outl = 0, h_slot_in = 0;
        float Tcw_in[12];
        const so_map* map = nullptr;  // the table the pose kernel reads its points from
        const so_dframe* frame = nullptr;
        bool holds_map = false;       // pose_again: held shared until the wait (a stage holds it through its search)
    } chain;
    so::PoseOptArgs chain_pose;  // the last stage's PoseOptimization launch (pose_again: another start pose, same edges)
    // ---- member of a so_track_group: the launches of its tracking stages are RECORDED (while group_recording, i.e. inside
    //      so_track_stage_*_submit) and go out with the other members' as three launches (so_track_group_launch)
    so_track_group* group = nullptr;
    bool group_recording = false;
    // ---- the local-map stage LINKED behind another matcher's last-frame stage (so_track_stage_local_map_submit_after): its three
    //      launches are recorded into `link_rec` and go out as one-row tables that track_link_kernel patches on the device
    bool link_recording = false;
    struct LinkRec* link_rec = nullptr;
    PinBuf h_link;
    DevBuf d_link;
    std::vector<int> cell_count;

    // ---- so_matcher_batch_begin / _end: independent calls staged side by side, launched together ----
    struct BatchJob {
        size_t base = 0;         // of the job's staged block inside hb_in / db_in
        size_t q_off = 0;        // of its MatchQuery records inside db_q when the projection writes them (else in the block)
        size_t off_oct = 0, off_desc = 0, off_limit = 0, off_cols = 0, off_q = 0, off_qdesc = 0;
        size_t keys_off = 0, cnt_off = 0, qw_off = 0;  // inside hb_out
        int n_cand = 0, nq = 0, K = 1;
        bool has_limit = false, has_cols = false, project = false;
        float inv_sigma2[8], sigma2[8], scale[8], ex = 0.f, ey = 0.f, min_x = 0.f, min_y = 0.f, grid_inv_w = 0.f, grid_inv_h = 0.f,
              grid_min_y = 0.f;
        so::ProjectSrc S{};
        size_t off_xw = 0, off_nrm = 0, off_maxd = 0, off_mind = 0, off_valid = 0;
        size_t mp_base = 0;      // block that holds the map points' fields and descriptors (an earlier job's when shared)
        const so_map* map = nullptr;  // the fields and descriptors are rows of this device-resident table instead:
        size_t off_slot = 0;          // the staged block holds [slots | valid]
        std::vector<int> perm;
        const so_kframe* ext = nullptr;  // candidates read from an HBM-resident keyframe instead of the staged block
        int ext_layout = 0;              // 0: grid order, 1: vocabulary-node order
        // project == 2 in the device table: SearchForTriangulation queries generated on the device from resident keyframe 1
        const so_kframe* tri_kf1 = nullptr;
        size_t off_tfree1 = 0, off_tnode = 0, off_trange = 0;  // inside the job's block
        float tri_F[9] = {0};
        // query descriptors by index: row qidx[i] of a descriptor table staged ONCE per batch (a keyframe's twenty
        // SearchForTriangulation calls query with subsets of the same keypoints' descriptors)
        bool q_indexed = false;
        size_t off_qidx = 0;     // of the indices inside the job's block
        size_t qtab_abs = 0;     // of the table inside hb_in / db_in (behind kBatchTableBytes)
        std::function<int(const uint32_t* keys, const int32_t* cnt, const so::MatchQueryW* qw, const std::vector<int>& perm)> resolve;
    };
    bool batching = false;
    std::vector<BatchJob> jobs;
    PinBuf hb_in;
    DevBuf db_in, db_q;
    MappedBuf hb_out;
    size_t hb_used = 0, dq_used = 0, out_used = 0;
    // The map points of the last projected job staged in this batch: a keyframe's Fuse calls hand the SAME arrays to every
    // neighbour's search (only `valid` differs), so a job whose arrays are those - same pointers AND same bytes - refers
    // to that block and stages its flags only (20 of a keyframe's 21 Fuse calls: 1.4 MB less to copy twice on the host
    // and once over PCIe).
    struct MpShare {
        const void *desc = nullptr, *xw = nullptr, *nrm = nullptr, *maxd = nullptr, *mind = nullptr;
        int n = -1;
        size_t base = 0, off_qdesc = 0, off_xw = 0, off_nrm = 0, off_maxd = 0, off_mind = 0;
    } mp_share;
    // The keypoint descriptors of the keyframe whose SearchForTriangulation calls this batch holds: staged once, by the
    // first of them (same pointer, same count AND same bytes as what the next call passes, else that call stages its own)
    struct TriShare {
        const uint8_t* desc = nullptr;
        int n = -1;
        size_t abs = 0;  // of the staged table inside hb_in (behind kBatchTableBytes)
    } tri_share;
};

namespace {

// Frame::PosInGrid, code/src/Frame.cc:433-443
inline bool pos_in_grid(const so_frame_view* F, int i, int& px, int& py) {
    const float ox = F->has_grid_origin ? F->grid_min_x : F->min_x;  // the origin the cells were assigned with
    const float oy = F->has_grid_origin ? F->grid_min_y : F->min_y;
    px = (int)roundf((F->x[i] - ox) * F->grid_inv_w);
    py = (int)roundf((F->y[i] - oy) * F->grid_inv_h);
    return !(px < 0 || px >= kGridCols || py < 0 || py >= kGridRows);
}

// Upload candidates in a given scan order (perm[position] = keypoint index); positions are the tie-break ranks.
int upload_ordered(so_matcher* m, int n, const float* x, const float* y, const int32_t* octave, const uint8_t* desc,
                   const uint8_t* excluded, const int32_t* limit_by_idx) {
    const int nc = (int)m->perm.size();
    if (nc > 65535) {
        last_error_ref() = "matcher supports at most 65535 candidates per call";
        return SO_ERR_INVALID_ARG;
    }
    (void)n;
    m->src = nullptr;
    m->n_cand = nc;
    const bool want_limit = (excluded != nullptr) || (limit_by_idx != nullptr);
    m->has_limit = want_limit;
    int rc;
    m->off_oct = align256(sizeof(float2) * (size_t)nc);
    m->off_desc = align256(m->off_oct + (size_t)nc);
    m->off_limit = align256(m->off_desc + (size_t)nc * 32);
    m->off_cols = align256(m->off_limit + sizeof(int32_t) * (size_t)nc);
    m->frame_end = align256(m->off_cols + sizeof(int32_t) * (kGridCols + 1));
    m->has_cols = false;
    m->resident_n = -1;
    if ((rc = m->h_in.ensure_keep(m->frame_end + 256, 0))) return rc;
    uint8_t* base = (uint8_t*)m->h_in.p;
    float2* hxy = (float2*)base;
    int8_t* hoct = (int8_t*)(base + m->off_oct);
    uint8_t* hdesc = base + m->off_desc;
    for (int r = 0; r < nc; r++) {
        const int i = m->perm[(size_t)r];
        hxy[r] = make_float2(x ? x[i] : 0.f, y ? y[i] : 0.f);
        hoct[r] = (int8_t)(octave ? octave[i] : 0);
        memcpy(hdesc + (size_t)r * 32, desc + (size_t)i * 32, 32);
    }
    if (want_limit) {
        int32_t* hl = (int32_t*)(base + m->off_limit);
        for (int r = 0; r < nc; r++) {
            const int i = m->perm[(size_t)r];
            int32_t lim = limit_by_idx ? limit_by_idx[i] : INT_MAX;
            if (excluded && excluded[i]) lim = 0;
            hl[r] = lim;
        }
    }
    m->dirty_from = 0;
    m->h_q.p = m->h_qdesc.p = nullptr;
    return SO_OK;
}

// Order the frame's keypoints the way GetFeaturesInArea visits them (cell x outer, cell y inner, insertion
// order inside a cell = keypoint index; code/src/Frame.cc:277-292,401-427) and upload the SoA.
// `reuse_requested`: so_matcher_reuse_frame preceded this call (the flag is consumed at the top of every public entry
// point, take_reuse below); it is honoured only if the view is the very frame the handle holds.
int upload_frame(so_matcher* m, const so_frame_view* F, const int32_t* limit_by_idx, bool reuse_requested) {
    const int n = F->n;
    const bool reuse = reuse_requested && m->src == nullptr && m->resident_n == n && m->has_cols &&
                       m->res_desc == F->desc && m->res_x == F->x && m->res_y == F->y && m->min_x == F->min_x &&
                       m->min_y == F->min_y && m->grid_inv_w == F->grid_inv_w && m->grid_inv_h == F->grid_inv_h &&
                       m->grid_min_x == (F->has_grid_origin ? F->grid_min_x : F->min_x) &&
                       m->grid_min_y == (F->has_grid_origin ? F->grid_min_y : F->min_y);
    if (reuse) {  // same frame as the previous call on this handle: only the eligibility gate is re-read
        const bool want_limit = (F->excluded != nullptr) || (limit_by_idx != nullptr);
        if (want_limit) {
            int32_t* hl = (int32_t*)((uint8_t*)m->h_in.p + m->off_limit);
            for (int r = 0; r < m->n_cand; r++) {
                const int i = m->perm[(size_t)r];
                int32_t lim = limit_by_idx ? limit_by_idx[i] : INT_MAX;
                if (F->excluded && F->excluded[i]) lim = 0;
                hl[r] = lim;
            }
            m->dirty_from = std::min(m->dirty_from, m->off_limit);
        }
        m->has_limit = want_limit;
        m->h_q.p = m->h_qdesc.p = nullptr;
        return SO_OK;
    }
    m->cell_count.assign((size_t)kGridCols * kGridRows + 1, 0);
    std::vector<int>& cc = m->cell_count;
    std::vector<int> cell((size_t)n, -1);
    for (int i = 0; i < n; i++) {
        int px, py;
        if (pos_in_grid(F, i, px, py)) {
            cell[(size_t)i] = px * kGridRows + py;
            cc[(size_t)cell[(size_t)i] + 1]++;
        }
    }
    for (size_t c = 1; c < cc.size(); c++) cc[c] += cc[c - 1];
    const int nc = cc.back();
    m->perm.assign((size_t)nc, 0);
    {
        std::vector<int> fill(cc.begin(), cc.end() - 1);
        for (int i = 0; i < n; i++)
            if (cell[(size_t)i] >= 0) m->perm[(size_t)fill[(size_t)cell[(size_t)i]]++] = i;
    }
    const int rc = upload_ordered(m, n, F->x, F->y, F->octave, F->desc, F->excluded, limit_by_idx);
    if (rc) return rc;
    int32_t* cols = (int32_t*)((uint8_t*)m->h_in.p + m->off_cols);
    for (int px = 0; px <= kGridCols; px++) cols[px] = cc[(size_t)px * kGridRows];
    m->has_cols = true;
    m->resident_n = n;
    m->res_desc = F->desc; m->res_x = F->x; m->res_y = F->y;
    m->min_x = F->min_x; m->min_y = F->min_y; m->grid_inv_w = F->grid_inv_w; m->grid_inv_h = F->grid_inv_h;
    m->grid_min_x = F->has_grid_origin ? F->grid_min_x : F->min_x;
    m->grid_min_y = F->has_grid_origin ? F->grid_min_y : F->min_y;
    return SO_OK;
}

// One-shot flag of so_matcher_reuse_frame: read and cleared before anything can return early.
inline bool take_reuse(so_matcher* m) {
    const bool r = m->reuse_next;
    m->reuse_next = false;
    return r;
}

// Candidates = a device-resident frame: nothing is sorted or uploaded, the handle only mirrors the position ->
// keypoint map and stages the optional eligibility gate (by position) at the start of its staging block.
int use_dframe(so_matcher* m, const so_dframe* f, const uint8_t* excluded) {
    if (!f || !f->ready) {
        last_error_ref() = "device-resident frame has not been collected";
        return SO_ERR_INVALID_ARG;
    }
    if (f->device != m->device) {
        last_error_ref() = "device-resident frame lives on another device";
        return SO_ERR_INVALID_ARG;
    }
    const int nc = f->n_inside;
    m->src = f;
    m->n_cand = nc;
    m->perm.assign(f->h_perm, f->h_perm + nc);
    m->has_cols = true;
    m->resident_n = -1;
    m->min_x = f->bounds[0];
    m->min_y = f->bounds[2];
    m->grid_min_x = f->bounds[0];
    m->grid_min_y = f->bounds[2];
    m->grid_inv_w = (float)kGridCols / (f->bounds[1] - f->bounds[0]);  // Frame.cc:259-260
    m->grid_inv_h = (float)kGridRows / (f->bounds[3] - f->bounds[2]);
    m->off_oct = m->off_desc = m->off_cols = 0;
    m->off_limit = 0;
    m->frame_end = align256(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1));
    int rc;
    if ((rc = m->h_in.ensure_keep(m->frame_end + 256, 0))) return rc;
    m->has_limit = excluded != nullptr;
    if (excluded) {
        int32_t* hl = (int32_t*)m->h_in.p;
        for (int r = 0; r < nc; r++) hl[r] = excluded[m->perm[(size_t)r]] ? 0 : INT_MAX;
        m->dirty_from = 0;
    } else {
        m->dirty_from = m->frame_end;
    }
    m->h_q.p = m->h_qdesc.p = nullptr;
    return SO_OK;
}

inline MatchQuery expand_query(const MatchQueryW& c) {
    MatchQuery q;
    memset(&q, 0, sizeof(q));
    q.u = c.u; q.v = c.v; q.r = c.r;
    q.min_level = c.min_level; q.max_level = c.max_level;
    q.active = c.active;
    q.max_dist = 256;
    return q;
}

inline int8_t level8(int l) { return (int8_t)(l < -128 ? -128 : (l > 127 ? 127 : l)); }

inline void init_query(MatchQuery& q) {
    memset(&q, 0, sizeof(q));
    q.max_dist = 256;
}

int upload_limit_only(so_matcher* m, const std::vector<int32_t>& limit_by_idx) {
    const int nc = m->n_cand;
    int32_t* hl = (int32_t*)((uint8_t*)m->h_in.p + m->off_limit);
    for (int r = 0; r < nc; r++) hl[r] = limit_by_idx[(size_t)m->perm[(size_t)r]];
    m->dirty_from = std::min(m->dirty_from, m->off_limit);
    m->has_limit = true;
    return SO_OK;
}

MatchFrameDev frame_dev(const so_matcher* m) {
    MatchFrameDev F;
    const uint8_t* base = (const uint8_t*)m->d_in.p;
    F.xy = (const float2*)base;
    F.octave = (const int8_t*)(base + m->off_oct);
    F.desc = (const uint4*)(base + m->off_desc);
    if (m->src) {  // the frame's own HBM arrays (dframe_internal.h), already in grid-traversal order
        F.xy = m->src->d_s_xy;
        F.octave = m->src->d_s_octave;
        F.desc = m->src->d_s_desc;
    }
    F.limit = m->has_limit ? (const int32_t*)(base + m->off_limit) : nullptr;
    F.n = m->n_cand;
    for (int l = 0; l < 8; l++) {
        F.inv_sigma2[l] = m->inv_sigma2[l];
        F.sigma2[l] = m->sigma2[l];
        F.scale[l] = m->scale[l];
    }
    F.ex = m->ex;
    F.ey = m->ey;
    F.col_start = m->has_cols ? (const int32_t*)(base + m->off_cols) : nullptr;
    if (m->src) F.col_start = m->src->d_col_start;
    F.min_x = m->min_x; F.min_y = m->min_y; F.grid_inv_w = m->grid_inv_w; F.grid_inv_h = m->grid_inv_h;
    F.grid_min_y = m->grid_min_y;
    return F;
}

// queries must already be in m->h_q / m->h_qdesc (ensure_queries).  Results land in m->h_keys / m->h_count.
int run_topk(so_matcher* m, int nq, int K, bool compact = false) {
    if (nq <= 0) return SO_OK;
    if (m->batching) {
        last_error_ref() = "this call cannot be part of a matcher batch (so_matcher_batch_begin): only so_search_for_triangulation, "
                           "so_fuse and so_fuse_sim3 can";
        return SO_ERR_INVALID_ARG;
    }
    int rc;
    const size_t total = m->off_qdesc + (size_t)nq * 32;
    if (m->d_in.cap < total) {
        if ((rc = m->d_in.ensure(m->h_in.cap))) return rc;
        m->dirty_from = 0;
    }
    const size_t keys_bytes = align256(sizeof(uint32_t) * (size_t)nq * K);
    if ((rc = m->h_out.ensure(keys_bytes + sizeof(int32_t) * (size_t)nq))) return rc;
    hipStream_t s = m->stream;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t from = std::min(m->dirty_from, m->off_q);
    {   // staged inputs go up with a copy kernel on this queue (an SDMA copy costs ~10 us more per call, match_kernels.hip)
        const size_t f16 = from & ~(size_t)15;
        launch_stage_in((uint8_t*)m->d_in.p + f16, (const uint8_t*)m->h_in.p + f16, total - f16, s);
    }
    m->dirty_from = SIZE_MAX;
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_topk_window(frame_dev(m), (const uint8_t*)m->d_in.p + m->off_q, compact,
                       (const uint4*)((const uint8_t*)m->d_in.p + m->off_qdesc), nq, K, (uint32_t*)m->h_out.dev,
                       (int32_t*)((uint8_t*)m->h_out.dev + keys_bytes), s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    const auto t1 = std::chrono::steady_clock::now();
    SO_HIP(hipStreamSynchronize(s));
    const auto t2 = std::chrono::steady_clock::now();
    m->stat[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    m->stat[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
    m->stat[2] += 1.0;
    m->stat[3] += (double)(total - from);
    m->h_keys.p = m->h_out.p;
    m->h_count.p = (uint8_t*)m->h_out.p + keys_bytes;
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    return SO_OK;
}

// Reserve the query part of the staging buffer behind the frame part (which is kept).
int ensure_queries(so_matcher* m, int nq, size_t record = sizeof(MatchQuery)) {
    const size_t n = (size_t)(nq > 0 ? nq : 1);
    m->off_q = m->frame_end;
    m->off_qdesc = align256(m->off_q + record * n);
    int rc = m->h_in.ensure_keep(m->off_qdesc + n * 32 + 256, m->frame_end);
    if (rc) return rc;
    m->h_q.p = (uint8_t*)m->h_in.p + m->off_q;
    m->h_qdesc.p = (uint8_t*)m->h_in.p + m->off_qdesc;
    return SO_OK;
}

constexpr size_t kBatchMaxJobs = kBatchMaxJobsDev;
constexpr size_t kBatchGridBytes = ((sizeof(BatchGridDev) + 255) / 256) * 256;  // the grid table sits in front of the job table
constexpr size_t kBatchTableBytes = kBatchGridBytes + ((sizeof(BatchJobDev) * kBatchMaxJobs + 255) / 256) * 256;

int batch_flush(so_matcher* m);

// The call whose inputs sit in m->h_in [0, staged_end) (frame part, then queries / descriptors / map points at the
// offsets recorded in m) joins the batch: its block is copied behind the others, its outputs are deferred to `resolve`.
int batch_defer(so_matcher* m, size_t staged_end, int nq, int K, const ProjectSrc* S, const size_t* mp_offsets5,
                std::function<int(const uint32_t*, const int32_t*, const MatchQueryW*, const std::vector<int>&)> resolve,
                const so_kframe* ext = nullptr, int ext_layout = 0, const so_mappoint_view* mp = nullptr, bool mp_shared = false,
                const so_map* map = nullptr, size_t off_slot = 0) {
    int rc;
    if (m->jobs.size() == kBatchMaxJobs && (rc = batch_flush(m))) return rc;
    so_matcher::BatchJob J;
    J.ext = ext;
    J.ext_layout = ext_layout;
    J.base = m->hb_used;
    J.off_oct = m->off_oct; J.off_desc = m->off_desc; J.off_limit = m->off_limit; J.off_cols = m->off_cols;
    J.off_q = m->off_q; J.off_qdesc = m->off_qdesc;
    J.n_cand = m->n_cand; J.has_limit = m->has_limit; J.has_cols = m->has_cols;
    memcpy(J.inv_sigma2, m->inv_sigma2, sizeof(J.inv_sigma2));
    memcpy(J.sigma2, m->sigma2, sizeof(J.sigma2));
    memcpy(J.scale, m->scale, sizeof(J.scale));
    J.ex = m->ex; J.ey = m->ey;
    J.min_x = m->min_x; J.min_y = m->min_y; J.grid_inv_w = m->grid_inv_w; J.grid_inv_h = m->grid_inv_h; J.grid_min_y = m->grid_min_y;
    J.nq = nq; J.K = K;
    J.project = S != nullptr;
    if (S) {
        J.S = *S;
        J.off_xw = mp_offsets5[0]; J.off_nrm = mp_offsets5[1]; J.off_maxd = mp_offsets5[2]; J.off_mind = mp_offsets5[3];
        J.off_valid = mp_offsets5[4];
        J.map = map;
        J.off_slot = off_slot;
        J.mp_base = mp_shared ? m->mp_share.base : J.base;
        if (mp_shared) {
            J.off_qdesc = m->mp_share.off_qdesc;
        } else if (mp) {
            so_matcher::MpShare& sh = m->mp_share;
            sh.desc = mp->desc; sh.xw = mp->Xw; sh.nrm = mp->normal; sh.maxd = mp->max_dist; sh.mind = mp->min_dist;
            sh.n = nq;
            sh.base = J.base;
            sh.off_qdesc = m->off_qdesc; sh.off_xw = mp_offsets5[0]; sh.off_nrm = mp_offsets5[1]; sh.off_maxd = mp_offsets5[2];
            sh.off_mind = mp_offsets5[3];
        }
        J.q_off = m->dq_used;
        m->dq_used += align256(sizeof(MatchQuery) * (size_t)nq);
    }
    if (!ext) J.perm = m->perm;  // (a resident keyframe carries its own position -> keypoint maps)
    J.resolve = std::move(resolve);
    const size_t block = align256(staged_end);
    if ((rc = m->hb_in.ensure_keep(kBatchTableBytes + J.base + block + 256, kBatchTableBytes + J.base))) return rc;
    memcpy((uint8_t*)m->hb_in.p + kBatchTableBytes + J.base, m->h_in.p, staged_end);
    m->hb_used += block;
    J.keys_off = m->out_used;
    m->out_used += align256(sizeof(uint32_t) * (size_t)nq * K);
    J.cnt_off = m->out_used;
    m->out_used += align256(sizeof(int32_t) * (size_t)nq);
    J.qw_off = m->out_used;
    if (S) m->out_used += align256(sizeof(MatchQueryW) * (size_t)nq);
    m->jobs.push_back(std::move(J));
    // whatever the handle holds of a frame is not what the next plain call may reuse
    m->resident_n = -1;
    m->dirty_from = 0;
    return SO_OK;
}

// Launches everything deferred so far (one staging copy, one projection launch, one search launch), waits, resolves.
int batch_flush(so_matcher* m) {
    const int nj = (int)m->jobs.size();
    if (nj == 0) return SO_OK;
    int rc;
    const size_t total = kBatchTableBytes + m->hb_used;
    if ((rc = m->db_in.ensure(total))) return rc;
    if ((rc = m->db_q.ensure(m->dq_used + 256))) return rc;
    if ((rc = m->hb_out.ensure(m->out_used + 256))) return rc;
    // maps read by this batch: their tables stay where they are until the kernels are done
    std::vector<const so_map*> maps;
    for (const so_matcher::BatchJob& J : m->jobs)
        if (J.map && std::find(maps.begin(), maps.end(), J.map) == maps.end()) maps.push_back(J.map);
    struct MapLocks {
        std::vector<const so_map*>& v;
        size_t held = 0;
        explicit MapLocks(std::vector<const so_map*>& maps_) : v(maps_) {
            for (const so_map* mp : v) {
                const_cast<so_map*>(mp)->grow_mu.lock_shared();
                held++;
            }
        }
        void release() {
            for (size_t i = 0; i < held; i++) const_cast<so_map*>(v[i])->grow_mu.unlock_shared();
            held = 0;
        }
        ~MapLocks() { release(); }
    } map_locks(maps);
    BatchGridDev* grid = (BatchGridDev*)m->hb_in.p;
    BatchJobDev* tab = (BatchJobDev*)((uint8_t*)m->hb_in.p + kBatchGridBytes);
    uint8_t* dbase = (uint8_t*)m->db_in.p + kBatchTableBytes;
    int proj_blocks = 0, topk_blocks = 0;
    memset(grid, 0, sizeof(*grid));
    grid->n_jobs = nj;
    for (int j = 0; j < nj; j++) {
        const so_matcher::BatchJob& J = m->jobs[(size_t)j];
        BatchJobDev D;
        memset(&D, 0, sizeof(D));
        const uint8_t* b = dbase + J.base;
        D.F.xy = (const float2*)b;
        D.F.octave = (const int8_t*)(b + J.off_oct);
        D.F.desc = (const uint4*)(b + J.off_desc);
        D.F.limit = J.has_limit ? (const int32_t*)(b + J.off_limit) : nullptr;
        D.F.n = J.n_cand;
        memcpy(D.F.inv_sigma2, J.inv_sigma2, sizeof(J.inv_sigma2));
        memcpy(D.F.sigma2, J.sigma2, sizeof(J.sigma2));
        memcpy(D.F.scale, J.scale, sizeof(J.scale));
        D.F.ex = J.ex; D.F.ey = J.ey;
        D.F.col_start = J.has_cols ? (const int32_t*)(b + J.off_cols) : nullptr;
        if (J.ext) {  // candidates of an HBM-resident keyframe; the staged block holds the gate, queries, descriptors
            const so_kframe* k = J.ext;
            if (J.ext_layout == 0) {
                D.F.xy = (const float2*)k->d;
                D.F.octave = (const int8_t*)(k->d + k->g_oct);
                D.F.desc = (const uint4*)(k->d + k->g_desc);
                D.F.col_start = (const int32_t*)(k->d + k->g_cols);
            } else {
                D.F.xy = (const float2*)(k->d + k->n_xy);
                D.F.octave = (const int8_t*)(k->d + k->n_oct);
                D.F.desc = (const uint4*)(k->d + k->n_desc);
                D.F.col_start = nullptr;
            }
            D.F.limit = J.has_limit ? (const int32_t*)b : nullptr;  // the gate is the first thing in the block
        }
        D.F.min_x = J.min_x; D.F.min_y = J.min_y; D.F.grid_inv_w = J.grid_inv_w; D.F.grid_inv_h = J.grid_inv_h;
        D.F.grid_min_y = J.grid_min_y;
        const uint8_t* mb = dbase + J.mp_base;
        D.qdesc = (const uint4*)((J.project ? mb : b) + J.off_qdesc);
        D.qslot = nullptr;
        D.keys = (uint32_t*)((uint8_t*)m->hb_out.dev + J.keys_off);
        D.count = (int32_t*)((uint8_t*)m->hb_out.dev + J.cnt_off);
        D.nq = J.nq; D.K = J.K; D.project = J.project ? 1 : 0;
        if (J.project) {
            D.S = J.S;
            D.S.Xw = (const float*)(mb + J.off_xw);
            D.S.normal = (const float*)(mb + J.off_nrm);
            D.S.max_dist = (const float*)(mb + J.off_maxd);
            D.S.min_dist = (const float*)(mb + J.off_mind);
            D.S.valid = b + J.off_valid;
            D.S.slot = nullptr;
            D.S.n_rows = 0;
            if (J.map) {  // (its tables cannot move before the wait below: batch_flush holds the map's grow lock shared)
                D.S.Xw = J.map->d_Xw;
                D.S.normal = J.map->d_normal;
                D.S.max_dist = J.map->d_max;
                D.S.min_dist = J.map->d_min;
                D.S.slot = (const int32_t*)(b + J.off_slot);
                D.S.n_rows = J.map->size.load();
                D.qdesc = (const uint4*)J.map->d_desc;
                D.qslot = D.S.slot;
            }
            D.S.n = J.nq;
            D.q = (MatchQuery*)((uint8_t*)m->db_q.p + J.q_off);
            D.qw = (MatchQueryW*)((uint8_t*)m->hb_out.dev + J.qw_off);
        } else if (J.tri_kf1) {
            const so_kframe* k1 = J.tri_kf1;
            D.project = 2;
            D.q = (MatchQuery*)((uint8_t*)m->db_q.p + J.q_off);
            D.qdesc = (const uint4*)(k1->d + k1->n_desc);  // query p's descriptor = row p of keyframe 1's node-ordered table
            D.qslot = nullptr;
            D.t_xy1 = (const float2*)(k1->d + k1->n_xy);
            D.t_free1 = b + J.off_tfree1;
            D.t_node = (const uint16_t*)(b + J.off_tnode);
            D.t_range = (const int2*)(b + J.off_trange);
            memcpy(D.t_F, J.tri_F, sizeof(D.t_F));
        } else {
            D.q = (MatchQuery*)(dbase + J.base + J.off_q);
            if (J.q_indexed) {
                D.qdesc = (const uint4*)(dbase + J.qtab_abs);
                D.qslot = (const int32_t*)(b + J.off_qidx);
            }
        }
        grid->first_proj[j] = proj_blocks;
        grid->first_topk[j] = topk_blocks;
        if (J.project || J.tri_kf1) proj_blocks += (J.nq + 255) / 256;
        topk_blocks += (J.nq + 3) / 4;
        tab[j] = D;
    }
    for (int j = nj; j <= (int)kBatchMaxJobs; j++) {
        grid->first_proj[j] = proj_blocks;
        grid->first_topk[j] = topk_blocks;
    }
    hipStream_t s = m->stream;
    const auto t0 = std::chrono::steady_clock::now();
    launch_stage_in(m->db_in.p, m->hb_in.p, total, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_batch((const BatchGridDev*)m->db_in.p, (const BatchJobDev*)((const uint8_t*)m->db_in.p + kBatchGridBytes), nj, proj_blocks, topk_blocks, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    const auto t1 = std::chrono::steady_clock::now();
    SO_HIP(hipStreamSynchronize(s));
    map_locks.release();
    const auto t2 = std::chrono::steady_clock::now();
    m->stat[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    m->stat[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
    m->stat[2] += 2.0;
    m->stat[3] += (double)total;
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    rc = SO_OK;
    for (int j = 0; j < nj; j++) {
        const so_matcher::BatchJob& J = m->jobs[(size_t)j];
        const uint8_t* ob = (const uint8_t*)m->hb_out.p;
        const std::vector<int>& perm = J.ext ? (J.ext_layout == 0 ? J.ext->perm_grid : J.ext->perm_node) : J.perm;
        const int r1 = J.resolve((const uint32_t*)(ob + J.keys_off), (const int32_t*)(ob + J.cnt_off),
                                 (const MatchQueryW*)(ob + J.qw_off), perm);
        if (r1 && !rc) rc = r1;
    }
    m->jobs.clear();
    m->hb_used = m->dq_used = m->out_used = 0;
    m->mp_share.n = -1;  // (its block went with the flush)
    m->tri_share.n = -1;
    return rc;
}

struct Entry {
    int idx, dist;
};

// Exact top-K of ONE query under a dynamic per-keypoint gate (rare path, see file header).  Uses its own staging:
// the batch's queries / K-lists (host-mapped, read in place by the callers) are not touched.
int rerun_single(so_matcher* m, const MatchQuery& q, const uint8_t* qdesc, const std::vector<int32_t>& limit_by_idx,
                 int K, Entry* out, int* n_found) {
    int rc = upload_limit_only(m, limit_by_idx);
    if (rc) return rc;
    constexpr size_t kQ = 256;  // query struct, then its descriptor at +128
    if ((rc = m->h_rq.ensure_keep(kQ, 0))) return rc;
    if ((rc = m->d_rq.ensure(kQ))) return rc;
    if ((rc = m->h_rout.ensure(512))) return rc;
    memcpy(m->h_rq.p, &q, sizeof(MatchQuery));
    memcpy((uint8_t*)m->h_rq.p + 128, qdesc, 32);
    hipStream_t s = m->stream;
    const auto t0 = std::chrono::steady_clock::now();
    // the limit gate sits at the end of the frame part of the staging block: send it (and whatever else is newer)
    if (m->dirty_from < m->frame_end) {
        const size_t f16 = m->dirty_from & ~(size_t)15;
        launch_stage_in((uint8_t*)m->d_in.p + f16, (const uint8_t*)m->h_in.p + f16, m->frame_end - f16, s);
    }
    m->dirty_from = SIZE_MAX;
    launch_stage_in(m->d_rq.p, m->h_rq.p, kQ, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_topk_window(frame_dev(m), m->d_rq.p, false, (const uint4*)((const uint8_t*)m->d_rq.p + 128), 1, K,
                       (uint32_t*)m->h_rout.dev, (int32_t*)((uint8_t*)m->h_rout.dev + 256), s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    const auto t1 = std::chrono::steady_clock::now();
    SO_HIP(hipStreamSynchronize(s));
    const auto t2 = std::chrono::steady_clock::now();
    m->stat[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    m->stat[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
    m->stat[2] += 1.0;
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    const uint32_t* keys = (const uint32_t*)m->h_rout.p;
    *n_found = 0;
    for (int k = 0; k < K; k++) {
        if (keys[k] == 0xFFFFFFFFu) break;
        out[*n_found].dist = (int)(keys[k] >> 16);
        out[*n_found].idx = m->perm[(size_t)(keys[k] & 0xFFFFu)];
        (*n_found)++;
    }
    return SO_OK;
}

// ORBmatcher::ComputeThreeMaxima, code/src/ORBmatcher.cc:1475-1506 (on bin populations)
void three_maxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

inline int rot_bin(float a1, float a2) {  // ORBmatcher.cc:1319-1325
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * (1.0f / HISTO_LENGTH));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

bool frame_ok(const so_frame_view* F) {
    return F && F->n >= 0 && (F->n == 0 || (F->x && F->y && F->octave && F->desc));
}

}  // namespace

extern "C" {

int so_matcher_create(int device, so_matcher** out) {
    if (!out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(device));
    so_matcher* m = new so_matcher();
    m->device = device;
    hipError_t e = context_stream(device, 1, &m->stream, &m->owns_stream);
    if (e == hipSuccess) e = hipEventCreate(&m->e0);
    if (e == hipSuccess) e = hipEventCreate(&m->e1);
    if (e != hipSuccess) {
        delete m;
        return hip_fail(e, "matcher init", __FILE__, __LINE__);
    }
    m->profile = getenv("SWARMORB_NO_EVENTS") == nullptr;
    *out = m;
    return SO_OK;
}

void so_matcher_destroy(so_matcher* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->pend.held_map) {  // a tracking search was submitted and never waited for
        const_cast<so_map*>(m->pend.held_map)->grow_mu.unlock_shared();
        m->pend.held_map = nullptr;
    }
    if (m->chain.holds_map && m->chain.map) {  // ... or a stage's repeated pose
        const_cast<so_map*>(m->chain.map)->grow_mu.unlock_shared();
        m->chain.holds_map = false;
    }
    for (DevBuf* b : {&m->d_in, &m->d_A, &m->d_B, &m->d_res, &m->d_link}) b->release();
    m->h_link.release();
    link_rec_free(m->link_rec);
    for (PinBuf* b : {&m->h_in, &m->h_res, &m->h_rq}) b->release();
    m->d_rq.release();
    m->h_out.release();
    m->h_rout.release();
    m->hb_in.release(); m->db_in.release(); m->db_q.release(); m->hb_out.release();
    m->d_chain.release();
    m->d_kpslot.release();
    m->h_chain.release();
    if (m->pe0) (void)hipEventDestroy(m->pe0);
    if (m->pe1) (void)hipEventDestroy(m->pe1);
    if (m->e0) (void)hipEventDestroy(m->e0);
    if (m->e1) (void)hipEventDestroy(m->e1);
    if (m->owns_stream && m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int so_matcher_last_kernel_ms(so_matcher* m, float* ms) {
    if (!m || !ms) return SO_ERR_INVALID_ARG;
    *ms = m->last_ms;
    return SO_OK;
}

int so_matcher_set_profiling(so_matcher* m, int enabled) {
    if (!m) return SO_ERR_INVALID_ARG;
    static const bool env_no_events = getenv("SWARMORB_NO_EVENTS") != nullptr;
    m->profile = enabled != 0 && !env_no_events;
    return SO_OK;
}

int so_matcher_reuse_frame(so_matcher* m) {
    if (!m) return SO_ERR_INVALID_ARG;
    m->reuse_next = true;
    return SO_OK;
}

// Staging for tracking searches of up to n_queries map points (K-lists, in-view flags, slots, gates) allocated now:
// pinned allocations cost 0.1-0.3 ms each, and a local map that grows keyframe by keyframe would otherwise pay one
// every time it outgrows the 1.5x slack (the first seconds of a sequence: +25-50 us per frame).
static int chain_reserve(so_matcher* m, int nq);  // (the tracking stages' device / host-mapped blocks: defined with them, below)

int so_matcher_reserve(so_matcher* m, int32_t n_queries) {
    if (!m || n_queries < 0 || m->pend.mode != 0) return SO_ERR_INVALID_ARG;  // (not while a submitted search owns the buffers)
    SO_HIP(hipSetDevice(m->device));
    constexpr int K = 8;
    const size_t nq = (size_t)n_queries;
    const size_t keys_bytes = align256(sizeof(uint32_t) * nq * K);
    const size_t view_off = align256(keys_bytes + sizeof(int32_t) * nq), c8_off = align256(view_off + nq);
    int rc;
    if ((rc = m->h_out.ensure(c8_off + nq))) return rc;
    // slots + skip bytes behind the frame block (whatever its size turns out to be: 256 KB covers 4096 keypoints)
    const size_t in_bytes = align256((size_t)256 * 1024 + sizeof(int32_t) * nq) + align256(nq) + 512;
    if (in_bytes > m->h_in.cap && (rc = m->h_in.ensure_keep(in_bytes, m->h_in.cap))) return rc;
    return chain_reserve(m, n_queries);
}

int so_matcher_last_stats(so_matcher* m, double* stats4) {
    if (!m || !stats4) return SO_ERR_INVALID_ARG;
    for (int i = 0; i < 4; i++) stats4[i] = m->stat[i];
    return SO_OK;
}

int so_matcher_topk(so_matcher* m, const so_frame_view* F, const int32_t* limit, int32_t nq, const float* u,
                    const float* v, const float* r, const int32_t* min_level, const int32_t* max_level,
                    const uint8_t* active, const uint8_t* qdesc, int32_t K, int32_t* out_idx, int32_t* out_dist,
                    int32_t* out_count) {
    if (!m || !frame_ok(F) || nq < 0 || K < 1 || K > 64) return SO_ERR_INVALID_ARG;
    if (nq > 0 && (!u || !v || !r || !min_level || !max_level || !qdesc || !out_idx || !out_dist || !out_count))
        return SO_ERR_INVALID_ARG;
    const bool reuse = take_reuse(m);
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    int rc = upload_frame(m, F, limit, reuse);
    if (rc) return rc;
    if ((rc = ensure_queries(m, nq))) return rc;
    MatchQuery* hq = (MatchQuery*)m->h_q.p;
    for (int i = 0; i < nq; i++) {
        init_query(hq[i]);
        hq[i].u = u[i];
        hq[i].v = v[i];
        hq[i].r = r[i];
        hq[i].min_level = min_level[i];
        hq[i].max_level = max_level[i];
        hq[i].active = active ? (active[i] != 0) : 1;
    }
    if (nq > 0) memcpy(m->h_qdesc.p, qdesc, (size_t)nq * 32);
    if ((rc = run_topk(m, nq, K))) return rc;
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;
    const int32_t* cnt = (const int32_t*)m->h_count.p;
    for (int i = 0; i < nq; i++) {
        out_count[i] = cnt[i];
        for (int k = 0; k < K; k++) {
            const uint32_t key = keys[(size_t)i * K + k];
            if (key == 0xFFFFFFFFu) {
                out_idx[(size_t)i * K + k] = -1;
                out_dist[(size_t)i * K + k] = 256;
            } else {
                out_idx[(size_t)i * K + k] = m->perm[(size_t)(key & 0xFFFFu)];
                out_dist[(size_t)i * K + k] = (int32_t)(key >> 16);
            }
        }
    }
    return SO_OK;
}

}  // extern "C"

namespace {

// The host-side fields of a device-resident frame as a frame view (x / y stay null: the searches below never read
// positions on the host).
so_frame_view view_of(const so_dframe* f, const uint8_t* excluded) {
    so_frame_view v{};
    v.n = f->n;
    v.octave = f->octave.data();
    v.angle = f->angle.data();
    v.excluded = excluded;
    v.min_x = f->bounds[0]; v.max_x = f->bounds[1]; v.min_y = f->bounds[2]; v.max_y = f->bounds[3];
    v.scale_factors = f->scale;
    v.nlevels = f->nlevels;
    return v;
}

// M1 — ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), code/src/ORBmatcher.cc:44-121.
// src != null: the candidates are the device-resident frame's (F then only carries its host-side fields).
int m1_search(so_matcher* m, const so_frame_view* F, const so_dframe* src, bool reuse, int32_t n_mp,
              const uint8_t* in_view, const float* proj_x, const float* proj_y, const float* view_cos,
              const int32_t* pred_level, const uint8_t* mp_desc, const uint8_t* mp_has_obs, float th, float nn_ratio,
              int32_t* kp_to_mp, int32_t* nmatches) {
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    *nmatches = 0;
    for (int k = 0; k < F->n; k++) kp_to_mp[k] = -1;
    if (n_mp == 0 || F->n == 0) return SO_OK;
    constexpr int K = 8;
    int rc = src ? use_dframe(m, src, F->excluded) : upload_frame(m, F, nullptr, reuse);
    if (rc) return rc;
    if ((rc = ensure_queries(m, n_mp, sizeof(MatchQueryW)))) return rc;
    MatchQueryW* hq = (MatchQueryW*)m->h_q.p;
    const bool bFactor = th != 1.0f;
    for (int i = 0; i < n_mp; i++) {
        MatchQueryW& q = hq[i];
        const int lvl = pred_level[i];
        const bool level_ok = lvl >= 0 && lvl < F->nlevels;
        q.active = (in_view[i] != 0 && level_ok) ? 1 : 0;
        float r = view_cos[i] > 0.998f ? 2.5f : 4.0f;  // RadiusByViewingCos, :123-128
        if (bFactor) r *= th;
        q.u = proj_x[i];
        q.v = proj_y[i];
        q.r = q.active ? r * F->scale_factors[lvl] : 0.f;
        q.min_level = level8(lvl - 1);
        q.max_level = level8(lvl);
        q.pad = 0;
    }
    memcpy(m->h_qdesc.p, mp_desc, (size_t)n_mp * 32);
    if ((rc = run_topk(m, n_mp, K, true))) return rc;
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;  // host-mapped, read in place (re-runs use their own staging)
    const int32_t* cnt = (const int32_t*)m->h_count.p;
    const MatchQueryW* queries = hq;
    std::vector<int32_t> gate;
    int nm = 0;
    for (int i = 0; i < n_mp; i++) {
        if (!queries[(size_t)i].active || cnt[(size_t)i] == 0) continue;
        Entry e[2];
        int found = 0, walked = 0;
        for (; walked < K && found < 2; walked++) {
            const uint32_t key = keys[(size_t)i * K + walked];
            if (key == 0xFFFFFFFFu) break;
            const int idx = m->perm[(size_t)(key & 0xFFFFu)];
            // F.mvpMapPoints[idx] bound earlier in this call to a point with observations (:83-85)
            if (kp_to_mp[idx] >= 0 && mp_has_obs[kp_to_mp[idx]]) continue;
            e[found].idx = idx;
            e[found].dist = (int)(key >> 16);
            found++;
        }
        if (found < 2 && walked == K && cnt[(size_t)i] > K) {  // list exhausted by taken keypoints: exact re-run
            gate.assign((size_t)F->n, INT_MAX);
            for (int k = 0; k < F->n; k++)
                if ((F->excluded && F->excluded[k]) || (kp_to_mp[k] >= 0 && mp_has_obs[kp_to_mp[k]])) gate[(size_t)k] = 0;
            if ((rc = rerun_single(m, expand_query(queries[(size_t)i]), mp_desc + (size_t)i * 32, gate, 2, e, &found))) return rc;
        }
        if (found == 0) continue;
        const int bestDist = e[0].dist, bestIdx = e[0].idx, bestLevel = F->octave[bestIdx];
        const int bestDist2 = found > 1 ? e[1].dist : 256;
        const int bestLevel2 = found > 1 ? F->octave[e[1].idx] : -1;
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
            kp_to_mp[bestIdx] = i;
            nm++;
        }
    }
    *nmatches = nm;
    return SO_OK;
}

// M2 — ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), code/src/ORBmatcher.cc:1223-1354
int m2_search(so_matcher* m, const so_frame_view* cur, const so_dframe* src, bool reuse, int32_t n_last,
              const uint8_t* valid, const float* u, const float* v, const int32_t* last_octave, const float* last_angle,
              const uint8_t* mp_desc, const uint8_t* mp_has_obs, float th, int check_orientation, int32_t* kp_to_last,
              int32_t* nmatches) {
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    *nmatches = 0;
    for (int k = 0; k < cur->n; k++) kp_to_last[k] = -1;
    if (n_last == 0 || cur->n == 0) return SO_OK;
    constexpr int K = 8;  // deep enough that a list exhausted by already-bound keypoints (exact re-run) is rare
    int rc = src ? use_dframe(m, src, cur->excluded) : upload_frame(m, cur, nullptr, reuse);
    if (rc) return rc;
    if ((rc = ensure_queries(m, n_last, sizeof(MatchQueryW)))) return rc;
    MatchQueryW* hq = (MatchQueryW*)m->h_q.p;
    for (int i = 0; i < n_last; i++) {
        MatchQueryW& q = hq[i];
        const int oct = last_octave[i];
        q.active = (valid[i] != 0 && oct >= 0 && oct < cur->nlevels) ? 1 : 0;
        q.u = u[i];
        q.v = v[i];
        q.r = q.active ? th * cur->scale_factors[oct] : 0.f;  // :1276
        q.min_level = level8(oct - 1);                         // :1285
        q.max_level = level8(oct + 1);
        q.pad = 0;
    }
    memcpy(m->h_qdesc.p, mp_desc, (size_t)n_last * 32);
    if ((rc = run_topk(m, n_last, K, true))) return rc;
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;  // host-mapped, read in place (re-runs use their own staging)
    const int32_t* cnt = (const int32_t*)m->h_count.p;
    const MatchQueryW* queries = hq;
    std::vector<int32_t> gate;
    std::vector<int>&rot_item = m->scratch_rot_item, &rot_b = m->scratch_rot_b;  // (capacity kept from call to call)
    rot_item.clear();
    rot_b.clear();
    int hist[HISTO_LENGTH] = {0};
    int nm = 0;
    for (int i = 0; i < n_last; i++) {
        if (!queries[(size_t)i].active || cnt[(size_t)i] == 0) continue;
        Entry e[1];
        int found = 0, walked = 0;
        for (; walked < K && found < 1; walked++) {
            const uint32_t key = keys[(size_t)i * K + walked];
            if (key == 0xFFFFFFFFu) break;
            const int idx = m->perm[(size_t)(key & 0xFFFFu)];
            if (kp_to_last[idx] >= 0 && mp_has_obs[kp_to_last[idx]]) continue;
            e[0].idx = idx;
            e[0].dist = (int)(key >> 16);
            found++;
        }
        if (found < 1 && walked == K && cnt[(size_t)i] > K) {
            gate.assign((size_t)cur->n, INT_MAX);
            for (int k = 0; k < cur->n; k++)
                if ((cur->excluded && cur->excluded[k]) || (kp_to_last[k] >= 0 && mp_has_obs[kp_to_last[k]]))
                    gate[(size_t)k] = 0;
            if ((rc = rerun_single(m, expand_query(queries[(size_t)i]), mp_desc + (size_t)i * 32, gate, 1, e, &found))) return rc;
        }
        if (found == 0) continue;
        if (e[0].dist <= TH_HIGH) {
            kp_to_last[e[0].idx] = i;
            nm++;
            if (check_orientation) {
                const int b = rot_bin(last_angle[i], cur->angle[e[0].idx]);
                rot_item.push_back(e[0].idx);
                rot_b.push_back(b);
                hist[b]++;
            }
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(hist, HISTO_LENGTH, i1, i2, i3);
        for (size_t j = 0; j < rot_item.size(); j++)
            if (rot_b[j] != i1 && rot_b[j] != i2 && rot_b[j] != i3) {
                kp_to_last[rot_item[j]] = -1;
                nm--;
            }
    }
    *nmatches = nm;
    return SO_OK;
}

}  // namespace

extern "C" {

int so_search_by_projection_mappoints(so_matcher* m, const so_frame_view* F, int32_t n_mp, const uint8_t* in_view,
                                      const float* proj_x, const float* proj_y, const float* view_cos,
                                      const int32_t* pred_level, const uint8_t* mp_desc, const uint8_t* mp_has_obs,
                                      float th, float nn_ratio, int32_t* kp_to_mp, int32_t* nmatches) {
    if (!m) return SO_ERR_INVALID_ARG;
    const bool reuse = take_reuse(m);
    if (!frame_ok(F) || n_mp < 0 || !kp_to_mp || !nmatches || !F->scale_factors) return SO_ERR_INVALID_ARG;
    if (n_mp > 0 && (!in_view || !proj_x || !proj_y || !view_cos || !pred_level || !mp_desc || !mp_has_obs))
        return SO_ERR_INVALID_ARG;
    return m1_search(m, F, nullptr, reuse, n_mp, in_view, proj_x, proj_y, view_cos, pred_level, mp_desc, mp_has_obs, th,
                     nn_ratio, kp_to_mp, nmatches);
}

int so_search_by_projection_mappoints_dframe(so_matcher* m, const so_dframe* F, const uint8_t* excluded, int32_t n_mp,
                                             const uint8_t* in_view, const float* proj_x, const float* proj_y,
                                             const float* view_cos, const int32_t* pred_level, const uint8_t* mp_desc,
                                             const uint8_t* mp_has_obs, float th, float nn_ratio, int32_t* kp_to_mp,
                                             int32_t* nmatches) {
    if (!m) return SO_ERR_INVALID_ARG;
    (void)take_reuse(m);
    if (!F || !F->ready || !F->mirrors || n_mp < 0 || !kp_to_mp || !nmatches) return SO_ERR_INVALID_ARG;
    if (n_mp > 0 && (!in_view || !proj_x || !proj_y || !view_cos || !pred_level || !mp_desc || !mp_has_obs))
        return SO_ERR_INVALID_ARG;
    const so_frame_view v = view_of(F, excluded);
    return m1_search(m, &v, F, false, n_mp, in_view, proj_x, proj_y, view_cos, pred_level, mp_desc, mp_has_obs, th,
                     nn_ratio, kp_to_mp, nmatches);
}

int so_search_by_projection_lastframe(so_matcher* m, const so_frame_view* cur, int32_t n_last, const uint8_t* valid,
                                      const float* u, const float* v, const int32_t* last_octave,
                                      const float* last_angle, const uint8_t* mp_desc, const uint8_t* mp_has_obs,
                                      float th, int check_orientation, int32_t* kp_to_last, int32_t* nmatches) {
    if (!m) return SO_ERR_INVALID_ARG;
    const bool reuse = take_reuse(m);
    if (!frame_ok(cur) || n_last < 0 || !kp_to_last || !nmatches || !cur->scale_factors) return SO_ERR_INVALID_ARG;
    if (n_last > 0 && (!valid || !u || !v || !last_octave || !mp_desc || !mp_has_obs)) return SO_ERR_INVALID_ARG;
    if (check_orientation && n_last > 0 && (!last_angle || !cur->angle)) return SO_ERR_INVALID_ARG;
    return m2_search(m, cur, nullptr, reuse, n_last, valid, u, v, last_octave, last_angle, mp_desc, mp_has_obs, th,
                     check_orientation, kp_to_last, nmatches);
}

int so_search_by_projection_lastframe_dframe(so_matcher* m, const so_dframe* cur, const uint8_t* excluded,
                                             int32_t n_last, const uint8_t* valid, const float* u, const float* v,
                                             const int32_t* last_octave, const float* last_angle,
                                             const uint8_t* mp_desc, const uint8_t* mp_has_obs, float th,
                                             int check_orientation, int32_t* kp_to_last, int32_t* nmatches) {
    if (!m) return SO_ERR_INVALID_ARG;
    (void)take_reuse(m);
    if (!cur || !cur->ready || !cur->mirrors || n_last < 0 || !kp_to_last || !nmatches) return SO_ERR_INVALID_ARG;
    if (n_last > 0 && (!valid || !u || !v || !last_octave || !mp_desc || !mp_has_obs)) return SO_ERR_INVALID_ARG;
    if (check_orientation && n_last > 0 && !last_angle) return SO_ERR_INVALID_ARG;
    const so_frame_view vw = view_of(cur, excluded);
    return m2_search(m, &vw, cur, false, n_last, valid, u, v, last_octave, last_angle, mp_desc, mp_has_obs, th,
                     check_orientation, kp_to_last, nmatches);
}

// M4 — ORBmatcher::SearchForInitialization, code/src/ORBmatcher.cc:375-479
int so_search_for_initialization(so_matcher* m, const so_frame_view* F1, const so_frame_view* F2,
                                 float* prev_matched, int window, float nn_ratio, int check_orientation,
                                 int32_t* matches12, int32_t* nmatches) {
    if (!m || !frame_ok(F1) || !frame_ok(F2) || !matches12 || !nmatches || (F1->n > 0 && !prev_matched))
        return SO_ERR_INVALID_ARG;
    if (check_orientation && ((F1->n > 0 && !F1->angle) || (F2->n > 0 && !F2->angle))) return SO_ERR_INVALID_ARG;
    const bool reuse = take_reuse(m);
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    *nmatches = 0;
    const int n1 = F1->n, n2 = F2->n;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return SO_OK;
    constexpr int K = 8;
    so_frame_view F2v = *F2;
    F2v.excluded = nullptr;  // SearchForInitialization never looks at mvpMapPoints
    int rc = upload_frame(m, &F2v, nullptr, reuse);
    if (rc) return rc;
    if ((rc = ensure_queries(m, n1))) return rc;
    MatchQuery* hq = (MatchQuery*)m->h_q.p;
    for (int i = 0; i < n1; i++) {
        MatchQuery& q = hq[i];
        init_query(q);
        const int level1 = F1->octave[i];
        q.active = !(level1 > 0);  // :393-395
        q.u = prev_matched[2 * i];
        q.v = prev_matched[2 * i + 1];
        q.r = (float)window;
        q.min_level = level1;
        q.max_level = level1;
    }
    memcpy(m->h_qdesc.p, F1->desc, (size_t)n1 * 32);
    if ((rc = run_topk(m, n1, K))) return rc;
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;  // host-mapped, read in place (re-runs use their own staging)
    const int32_t* cnt = (const int32_t*)m->h_count.p;
    const MatchQuery* queries = hq;
    std::vector<int32_t> matched_dist((size_t)n2, INT_MAX);
    std::vector<int32_t> matches21((size_t)n2, -1);
    std::vector<int>&rot_item = m->scratch_rot_item, &rot_b = m->scratch_rot_b;  // (capacity kept from call to call)
    rot_item.clear();
    rot_b.clear();
    int hist[HISTO_LENGTH] = {0};
    int nm = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        if (!queries[(size_t)i1].active || cnt[(size_t)i1] == 0) continue;
        Entry e[2];
        int found = 0, walked = 0;
        for (; walked < K && found < 2; walked++) {
            const uint32_t key = keys[(size_t)i1 * K + walked];
            if (key == 0xFFFFFFFFu) break;
            const int i2 = m->perm[(size_t)(key & 0xFFFFu)];
            const int dist = (int)(key >> 16);
            if (matched_dist[(size_t)i2] <= dist) continue;  // :413-414
            e[found].idx = i2;
            e[found].dist = dist;
            found++;
        }
        if (found < 2 && walked == K && cnt[(size_t)i1] > K) {
            if ((rc = rerun_single(m, queries[(size_t)i1], F1->desc + (size_t)i1 * 32, matched_dist, 2, e, &found)))
                return rc;
        }
        if (found == 0) continue;
        const int bestDist = e[0].dist, bestIdx2 = e[0].idx;
        const int bestDist2 = found > 1 ? e[1].dist : INT_MAX;
        if (bestDist <= TH_LOW) {
            if ((float)bestDist < (float)bestDist2 * nn_ratio) {
                if (matches21[(size_t)bestIdx2] >= 0) {
                    matches12[matches21[(size_t)bestIdx2]] = -1;
                    nm--;
                }
                matches12[i1] = bestIdx2;
                matches21[(size_t)bestIdx2] = i1;
                matched_dist[(size_t)bestIdx2] = bestDist;
                nm++;
                if (check_orientation) {
                    const int b = rot_bin(F1->angle[i1], F2->angle[bestIdx2]);
                    rot_item.push_back(i1);
                    rot_b.push_back(b);
                    hist[b]++;
                }
            }
        }
    }
    if (check_orientation) {
        int a, b, c;
        three_maxima(hist, HISTO_LENGTH, a, b, c);
        for (size_t j = 0; j < rot_item.size(); j++) {
            if (rot_b[j] == a || rot_b[j] == b || rot_b[j] == c) continue;
            if (matches12[rot_item[j]] >= 0) {
                matches12[rot_item[j]] = -1;
                nm--;
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)  // :472-475
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = F2->x[matches12[i1]];
            prev_matched[2 * i1 + 1] = F2->y[matches12[i1]];
        }
    *nmatches = nm;
    return SO_OK;
}

static int top2_common(so_matcher* m, const uint4* dA, int na, const uint4* dB, int nb, int32_t* best_idx,
                       int32_t* best_dist, int32_t* second_dist) {
    if (nb >= (1 << 20)) return SO_ERR_INVALID_ARG;
    int rc;
    if ((rc = m->d_res.ensure(sizeof(int32_t) * 3 * (size_t)na))) return rc;
    if ((rc = m->h_res.ensure(sizeof(int32_t) * 3 * (size_t)na))) return rc;
    int32_t* dr = (int32_t*)m->d_res.p;
    hipStream_t s = m->stream;
    SO_HIP(hipEventRecord(m->e0, s));
    launch_hamming_top2(dA, na, dB, nb, dr, dr + na, dr + 2 * (size_t)na, s);
    SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    SO_HIP(hipMemcpyAsync(m->h_res.p, dr, sizeof(int32_t) * 3 * (size_t)na, hipMemcpyDeviceToHost, s));
    SO_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms = ms;
    const int32_t* hr = (const int32_t*)m->h_res.p;
    memcpy(best_idx, hr, sizeof(int32_t) * (size_t)na);
    memcpy(best_dist, hr + na, sizeof(int32_t) * (size_t)na);
    memcpy(second_dist, hr + 2 * (size_t)na, sizeof(int32_t) * (size_t)na);
    return SO_OK;
}

int so_hamming_top2(so_matcher* m, const uint8_t* A, int32_t na, const uint8_t* B, int32_t nb, int32_t* best_idx,
                    int32_t* best_dist, int32_t* second_dist) {
    if (!m || na < 0 || nb < 0 || (na > 0 && (!A || !best_idx || !best_dist || !second_dist)) || (nb > 0 && !B))
        return SO_ERR_INVALID_ARG;
    (void)take_reuse(m);
    if (na == 0) return SO_OK;
    SO_HIP(hipSetDevice(m->device));
    if (m->batching) {
        last_error_ref() = "so_hamming_top2 cannot be part of a matcher batch";
        return SO_ERR_INVALID_ARG;
    }
    if (nb >= (1 << 20)) return SO_ERR_INVALID_ARG;
    // like every other call: inputs through the pinned staging block and the copy kernel on the handle's own queue, results
    // written by the kernel into host-mapped memory (two pageable hipMemcpyAsync + a device-to-host copy cost ~60 us more
    // per call - it is on the local-mapping thread's path once per keyframe, as the vocabulary stand-in)
    int rc;
    const size_t o_b = align256((size_t)na * 32), end = align256(o_b + (size_t)(nb > 0 ? nb : 1) * 32);
    if ((rc = m->h_in.ensure_keep(end + 256, 0))) return rc;
    if ((rc = m->d_in.ensure(end + 256))) return rc;
    const size_t rb = align256(sizeof(int32_t) * (size_t)na);
    if ((rc = m->h_out.ensure(3 * rb))) return rc;
    memcpy(m->h_in.p, A, (size_t)na * 32);
    if (nb > 0) memcpy((uint8_t*)m->h_in.p + o_b, B, (size_t)nb * 32);
    m->resident_n = -1;
    m->dirty_from = 0;
    m->src = nullptr;
    hipStream_t s = m->stream;
    uint8_t* db = (uint8_t*)m->d_in.p;
    launch_stage_in(db, m->h_in.p, end, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_hamming_top2((const uint4*)db, na, (const uint4*)(db + o_b), nb, (int32_t*)m->h_out.dev, (int32_t*)((uint8_t*)m->h_out.dev + rb),
                        (int32_t*)((uint8_t*)m->h_out.dev + 2 * rb), s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    m->last_ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms = ms;
    memcpy(best_idx, m->h_out.p, sizeof(int32_t) * (size_t)na);
    memcpy(best_dist, (const uint8_t*)m->h_out.p + rb, sizeof(int32_t) * (size_t)na);
    memcpy(second_dist, (const uint8_t*)m->h_out.p + 2 * rb, sizeof(int32_t) * (size_t)na);
    return SO_OK;
}

// MapPoint::ComputeDistinctiveDescriptors for a batch of map points (code/src/MapPoint.cc:323-392)
int so_distinctive_descriptors(so_matcher* m, int32_t n_points, const int32_t* offsets, const uint8_t* descriptors,
                               int32_t* best_idx, int32_t* best_median) {
    if (!m || n_points < 0 || (n_points > 0 && (!offsets || !best_idx))) return SO_ERR_INVALID_ARG;
    (void)take_reuse(m);
    if (n_points == 0) return SO_OK;
    const int total = offsets[n_points];
    if (offsets[0] != 0 || total < 0 || (total > 0 && !descriptors)) return SO_ERR_INVALID_ARG;
    for (int p = 0; p < n_points; p++) {
        const int N = offsets[p + 1] - offsets[p];
        if (N < 0) return SO_ERR_INVALID_ARG;
        if (N > kDistinctiveMaxObs) {
            last_error_ref() = "a map point has more than 512 observations";
            return SO_ERR_CAPACITY;
        }
    }
    SO_HIP(hipSetDevice(m->device));
    int rc;
    const size_t off_bytes = (sizeof(int32_t) * ((size_t)n_points + 1) + 255) & ~(size_t)255;
    const size_t in_bytes = off_bytes + (size_t)(total > 0 ? total : 1) * 32;
    if ((rc = m->h_in.ensure_keep(in_bytes, 0))) return rc;
    if (m->d_in.cap < in_bytes && (rc = m->d_in.ensure(m->h_in.cap))) return rc;
    m->dirty_from = 0;  // the staging block no longer holds a candidate frame
    m->resident_n = -1;
    m->src = nullptr;
    if ((rc = m->h_out.ensure(sizeof(int32_t) * 2 * (size_t)n_points))) return rc;
    memcpy(m->h_in.p, offsets, sizeof(int32_t) * ((size_t)n_points + 1));
    if (total > 0) memcpy((uint8_t*)m->h_in.p + off_bytes, descriptors, (size_t)total * 32);
    hipStream_t s = m->stream;
    SO_HIP(hipMemcpyAsync(m->d_in.p, m->h_in.p, in_bytes, hipMemcpyHostToDevice, s));
    int32_t* out = (int32_t*)m->h_out.dev;
    launch_distinctive_desc((const uint4*)((const uint8_t*)m->d_in.p + off_bytes), (const int32_t*)m->d_in.p, n_points,
                            out, out + n_points, s);
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(s));
    const int32_t* ho = (const int32_t*)m->h_out.p;
    memcpy(best_idx, ho, sizeof(int32_t) * (size_t)n_points);
    if (best_median) memcpy(best_median, ho + n_points, sizeof(int32_t) * (size_t)n_points);
    return SO_OK;
}

int so_hamming_top2_device(so_matcher* m, const uint8_t* d_A, int32_t na, const uint8_t* d_B, int32_t nb,
                           int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
    if (!m || na < 0 || nb < 0 || (na > 0 && (!d_A || !best_idx || !best_dist || !second_dist)) || (nb > 0 && !d_B))
        return SO_ERR_INVALID_ARG;
    (void)take_reuse(m);
    if (na == 0) return SO_OK;
    SO_HIP(hipSetDevice(m->device));
    return top2_common(m, (const uint4*)d_A, na, (const uint4*)d_B, nb, best_idx, best_dist, second_dist);
}

}  // extern "C"

// =====================================================================================================
// M3, M5, M6, M7 — the remaining ORBmatcher routines
// =====================================================================================================
namespace {

bool featvec_ok(const so_featvec* fv, int n) {
    if (!fv || fv->n_nodes < 0) return false;
    if (fv->n_nodes == 0) return true;
    if (!fv->node_id || !fv->off || !fv->idx) return false;
    for (int k = 0; k < fv->n_nodes; k++) {
        if (fv->off[k] > fv->off[k + 1]) return false;
        if (k > 0 && fv->node_id[k] <= fv->node_id[k - 1]) return false;
    }
    for (int a = fv->off[0]; a < fv->off[fv->n_nodes]; a++)
        if (fv->idx[a] < 0 || fv->idx[a] >= n) return false;
    return true;
}

int fv_lower_bound(const so_featvec* fv, int from, int id) {
    int k = from;
    while (k < fv->n_nodes && fv->node_id[k] < id) k++;
    return k;
}

struct NodeJoin {  // the merge-join of two feature vectors (ORBmatcher.cc:166-240): common nodes in order
    std::vector<int> k1, k2;
};

NodeJoin join_nodes(const so_featvec* fv1, const so_featvec* fv2) {
    NodeJoin j;
    int k1 = 0, k2 = 0;
    while (k1 < fv1->n_nodes && k2 < fv2->n_nodes) {
        if (fv1->node_id[k1] == fv2->node_id[k2]) {
            j.k1.push_back(k1++);
            j.k2.push_back(k2++);
        } else if (fv1->node_id[k1] < fv2->node_id[k2]) {
            k1 = fv_lower_bound(fv1, k1, fv2->node_id[k2]);
        } else {
            k2 = fv_lower_bound(fv2, k2, fv1->node_id[k1]);
        }
    }
    return j;
}

// candidates = the features of set 2 in feature-vector order; returns position offset of each node of fv2
int upload_featvec_candidates(so_matcher* m, int n2, const float* x2, const float* y2, const int32_t* octave2,
                              const uint8_t* desc2, const so_featvec* fv2, const uint8_t* excluded) {
    const int total = fv2->n_nodes > 0 ? fv2->off[fv2->n_nodes] - fv2->off[0] : 0;
    m->perm.resize((size_t)total);
    for (int a = 0; a < total; a++) m->perm[(size_t)a] = fv2->idx[fv2->off[0] + a];
    return upload_ordered(m, n2, x2, y2, octave2, desc2, excluded, nullptr);
}

void apply_rot_hist(const int* hist, const std::vector<int>& items, const std::vector<int>& bins, int32_t* target,
                    int& nm) {
    int a, b, c;
    three_maxima(hist, HISTO_LENGTH, a, b, c);
    for (size_t j = 0; j < items.size(); j++) {
        if (bins[j] == a || bins[j] == b || bins[j] == c) continue;
        target[items[j]] = -1;
        nm--;
    }
}

}  // namespace

extern "C" {

int so_search_by_bow(so_matcher* m, int variant, int32_t n1, const uint8_t* desc1, const float* angle1,
                     const uint8_t* valid1, const so_featvec* fv1, int32_t n2, const uint8_t* desc2,
                     const float* angle2, const uint8_t* valid2, const so_featvec* fv2, float nn_ratio,
                     int check_orientation, int32_t* match_of_2, int32_t* match_of_1, int32_t* nmatches) {
    if (!m || !nmatches || n1 < 0 || n2 < 0 || (variant != 0 && variant != 1)) return SO_ERR_INVALID_ARG;
    if ((n1 > 0 && (!desc1 || !valid1)) || (n2 > 0 && !desc2) || (variant == 1 && n2 > 0 && !valid2))
        return SO_ERR_INVALID_ARG;
    if (check_orientation && ((n1 > 0 && !angle1) || (n2 > 0 && !angle2))) return SO_ERR_INVALID_ARG;
    if (!featvec_ok(fv1, n1) || !featvec_ok(fv2, n2)) return SO_ERR_INVALID_ARG;
    (void)take_reuse(m);
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    *nmatches = 0;
    std::vector<int32_t> m2((size_t)n2, -1), m1((size_t)n1, -1);
    auto finish = [&]() {
        if (match_of_2) memcpy(match_of_2, m2.data(), sizeof(int32_t) * (size_t)n2);
        if (match_of_1) memcpy(match_of_1, m1.data(), sizeof(int32_t) * (size_t)n1);
    };
    const NodeJoin J = join_nodes(fv1, fv2);
    if (J.k1.empty() || n1 == 0 || n2 == 0) {
        finish();
        return SO_OK;
    }
    constexpr int K = 8;
    // variant 1: targets without a good map point never compete (static gate)
    std::vector<uint8_t> excl;
    if (variant == 1) {
        excl.resize((size_t)n2);
        for (int i = 0; i < n2; i++) excl[(size_t)i] = valid2[i] ? 0 : 1;
    }
    int rc = upload_featvec_candidates(m, n2, nullptr, nullptr, nullptr, desc2, fv2, variant == 1 ? excl.data() : nullptr);
    if (rc) return rc;
    // queries in the reference's visiting order: common nodes ascending, features of set 1 in node order
    std::vector<int> q_idx1;
    std::vector<MatchQuery> queries;
    const int base2 = fv2->off[0];
    for (size_t j = 0; j < J.k1.size(); j++) {
        const int k1 = J.k1[j], k2 = J.k2[j];
        for (int a = fv1->off[k1]; a < fv1->off[k1 + 1]; a++) {
            const int idx1 = fv1->idx[a];
            if (!valid1[idx1]) continue;
            MatchQuery q;
            init_query(q);
            q.active = 1;
            q.flags = kQRange;
            q.c_begin = fv2->off[k2] - base2;
            q.c_end = fv2->off[k2 + 1] - base2;
            queries.push_back(q);
            q_idx1.push_back(idx1);
        }
    }
    const int nq = (int)queries.size();
    if (nq == 0) {
        finish();
        return SO_OK;
    }
    if ((rc = ensure_queries(m, nq))) return rc;
    memcpy(m->h_q.p, queries.data(), sizeof(MatchQuery) * (size_t)nq);
    uint8_t* hd = (uint8_t*)m->h_qdesc.p;
    for (int i = 0; i < nq; i++) memcpy(hd + (size_t)i * 32, desc1 + (size_t)q_idx1[(size_t)i] * 32, 32);
    if ((rc = run_topk(m, nq, K))) return rc;
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;  // host-mapped, read in place (re-runs use their own staging)
    const int32_t* cnt = (const int32_t*)m->h_count.p;
    std::vector<int32_t> gate;
    std::vector<int>&rot_item = m->scratch_rot_item, &rot_b = m->scratch_rot_b;  // (capacity kept from call to call)
    rot_item.clear();
    rot_b.clear();
    int hist[HISTO_LENGTH] = {0};
    int nm = 0;
    for (int i = 0; i < nq; i++) {
        if (cnt[(size_t)i] == 0) continue;
        const int idx1 = q_idx1[(size_t)i];
        Entry e[2];
        int found = 0, walked = 0;
        for (; walked < K && found < 2; walked++) {
            const uint32_t key = keys[(size_t)i * K + walked];
            if (key == 0xFFFFFFFFu) break;
            const int idx2 = m->perm[(size_t)(key & 0xFFFFu)];
            if (m2[(size_t)idx2] >= 0) continue;  // vpMapPointMatches[realIdxF] / vbMatched2[idx2]
            e[found].idx = idx2;
            e[found].dist = (int)(key >> 16);
            found++;
        }
        if (found < 2 && walked == K && cnt[(size_t)i] > K) {
            gate.assign((size_t)n2, INT_MAX);
            for (int k = 0; k < n2; k++)
                if (m2[(size_t)k] >= 0 || (variant == 1 && !valid2[k])) gate[(size_t)k] = 0;
            if ((rc = rerun_single(m, queries[(size_t)i], desc1 + (size_t)idx1 * 32, gate, 2, e, &found))) return rc;
        }
        if (found == 0) continue;
        const int bestDist1 = e[0].dist, bestIdx2 = e[0].idx, bestDist2 = found > 1 ? e[1].dist : 256;
        const bool pass = variant == 0 ? (bestDist1 <= TH_LOW) : (bestDist1 < TH_LOW);
        if (pass && (float)bestDist1 < nn_ratio * (float)bestDist2) {
            m2[(size_t)bestIdx2] = idx1;
            m1[(size_t)idx1] = bestIdx2;
            if (check_orientation) {
                const int b = rot_bin(angle1[idx1], angle2[bestIdx2]);
                rot_item.push_back(variant == 0 ? bestIdx2 : idx1);
                rot_b.push_back(b);
                hist[b]++;
            }
            nm++;
        }
    }
    if (check_orientation) {
        if (variant == 0) {
            int a, b, c;
            three_maxima(hist, HISTO_LENGTH, a, b, c);
            for (size_t j = 0; j < rot_item.size(); j++) {
                if (rot_b[j] == a || rot_b[j] == b || rot_b[j] == c) continue;
                const int t = rot_item[j];
                if (m2[(size_t)t] >= 0) m1[(size_t)m2[(size_t)t]] = -1;
                m2[(size_t)t] = -1;
                nm--;
            }
        } else {  // vbMatched2 is internal to the reference; here both arrays are returned and must agree
            int a, b, c;
            three_maxima(hist, HISTO_LENGTH, a, b, c);
            for (size_t j = 0; j < rot_item.size(); j++) {
                if (rot_b[j] == a || rot_b[j] == b || rot_b[j] == c) continue;
                const int i1 = rot_item[j];
                if (m1[(size_t)i1] >= 0) m2[(size_t)m1[(size_t)i1]] = -1;
                m1[(size_t)i1] = -1;
                nm--;
            }
        }
    }
    finish();
    *nmatches = nm;
    return SO_OK;
}

int so_search_for_triangulation(so_matcher* m, int32_t n1, const float* x1, const float* y1, const float* angle1,
                                const uint8_t* desc1, const uint8_t* free1, const so_featvec* fv1, int32_t n2,
                                const float* x2, const float* y2, const int32_t* octave2, const float* angle2,
                                const uint8_t* desc2, const uint8_t* free2, const so_featvec* fv2, const float* F12,
                                float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                int32_t nlevels2, int check_orientation, int32_t* matches12, int32_t* nmatches) {
    if (!m || !nmatches || !matches12 || n1 < 0 || n2 < 0 || !F12 || !scale_factors2 || !level_sigma2_2 ||
        nlevels2 < 1 || nlevels2 > 8)
        return SO_ERR_INVALID_ARG;
    if ((n1 > 0 && (!x1 || !y1 || !desc1 || !free1)) || (n2 > 0 && (!x2 || !y2 || !octave2 || !desc2 || !free2)))
        return SO_ERR_INVALID_ARG;
    if (check_orientation && ((n1 > 0 && !angle1) || (n2 > 0 && !angle2))) return SO_ERR_INVALID_ARG;
    if (!featvec_ok(fv1, n1) || !featvec_ok(fv2, n2)) return SO_ERR_INVALID_ARG;
    (void)take_reuse(m);
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    const NodeJoin J = join_nodes(fv1, fv2);
    if (J.k1.empty() || n1 == 0 || n2 == 0) return SO_OK;
    std::vector<uint8_t> excl((size_t)n2);
    for (int i = 0; i < n2; i++) excl[(size_t)i] = free2[i] ? 0 : 1;  // "|| pMP2" (:662); vbMatched2 is never set
    for (int l = 0; l < 8; l++) {
        m->scale[l] = l < nlevels2 ? scale_factors2[l] : 0.f;
        m->sigma2[l] = l < nlevels2 ? level_sigma2_2[l] : 0.f;
    }
    m->ex = ex;
    m->ey = ey;
    int rc = upload_featvec_candidates(m, n2, x2, y2, octave2, desc2, fv2, excl.data());
    if (rc) return rc;
    std::vector<int> q_idx1;
    std::vector<MatchQuery> queries;
    const int base2 = fv2->off[0];
    for (size_t j = 0; j < J.k1.size(); j++) {
        const int k1 = J.k1[j], k2 = J.k2[j];
        for (int a = fv1->off[k1]; a < fv1->off[k1 + 1]; a++) {
            const int idx1 = fv1->idx[a];
            if (!free1[idx1]) continue;  // "If there is already a MapPoint skip" (:641-643)
            MatchQuery q;
            init_query(q);
            q.active = 1;
            q.flags = kQRange | kQEpipolar | kQPreferLast;
            q.max_dist = TH_LOW;  // "dist > TH_LOW || dist > bestDist -> continue" with bestDist starting at TH_LOW
            q.c_begin = fv2->off[k2] - base2;
            q.c_end = fv2->off[k2 + 1] - base2;
            q.la = x1[idx1] * F12[0] + y1[idx1] * F12[3] + F12[6];  // CheckDistEpipolarLine :134-136
            q.lb = x1[idx1] * F12[1] + y1[idx1] * F12[4] + F12[7];
            q.lc = x1[idx1] * F12[2] + y1[idx1] * F12[5] + F12[8];
            queries.push_back(q);
            q_idx1.push_back(idx1);
        }
    }
    const int nq = (int)queries.size();
    if (nq == 0) return SO_OK;
    if ((rc = ensure_queries(m, nq))) return rc;
    memcpy(m->h_q.p, queries.data(), sizeof(MatchQuery) * (size_t)nq);
    uint8_t* hd = (uint8_t*)m->h_qdesc.p;
    for (int i = 0; i < nq; i++) memcpy(hd + (size_t)i * 32, desc1 + (size_t)q_idx1[(size_t)i] * 32, 32);
    // the part after the launch: bindings + rotation histogram (ORBmatcher.cc:708-735); deferred inside a batch
    auto resolve = [m, nq, q_idx1 = std::move(q_idx1), angle1, angle2, check_orientation, matches12, nmatches](
                       const uint32_t* keys, const int32_t*, const MatchQueryW*, const std::vector<int>& perm) -> int {
        std::vector<int>&rot_item = m->scratch_rot_item, &rot_b = m->scratch_rot_b;  // (capacity kept from call to call)
        rot_item.clear();
        rot_b.clear();
        int hist[HISTO_LENGTH] = {0};
        int nm = 0;
        for (int i = 0; i < nq; i++) {
            if (keys[i] == 0xFFFFFFFFu) continue;
            const int idx1 = q_idx1[(size_t)i];
            const int idx2 = perm[(size_t)(0xFFFF - (keys[i] & 0xFFFFu))];  // prefer-last keys store 0xFFFF - position
            matches12[idx1] = idx2;
            nm++;
            if (check_orientation) {
                const int b = rot_bin(angle1[idx1], angle2[idx2]);
                rot_item.push_back(idx1);
                rot_b.push_back(b);
                hist[b]++;
            }
        }
        if (check_orientation) apply_rot_hist(hist, rot_item, rot_b, matches12, nm);
        *nmatches = nm;
        return SO_OK;
    };
    if (m->batching) return batch_defer(m, m->off_qdesc + (size_t)nq * 32, nq, 1, nullptr, nullptr, std::move(resolve));
    if ((rc = run_topk(m, nq, 1))) return rc;
    return resolve((const uint32_t*)m->h_keys.p, nullptr, nullptr, m->perm);
}

}  // extern "C"

namespace {

// The sequential part of the greedy window searches (ORBmatcher.cc:345-368, :1423-1452) over the K-lists of a finished
// launch (m->h_keys / m->h_count): a keypoint bound on entry never appears in a list (limit gate), one taken by an
// earlier query of this call is skipped here; a query whose list is exhausted that way is re-run exactly.
int resolve_greedy(so_matcher* m, const so_frame_view* F, int nq, const MatchQuery* queries, const uint8_t* qdesc,
                   const float* q_angle, int32_t max_dist, int check_orientation, int K, int32_t* kp_to_query,
                   int32_t* nmatches) {
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;  // host-mapped, read in place (re-runs use their own staging)
    const int32_t* cnt = (const int32_t*)m->h_count.p;
    std::vector<int32_t> gate;
    std::vector<int>&rot_item = m->scratch_rot_item, &rot_b = m->scratch_rot_b;  // (capacity kept from call to call)
    rot_item.clear();
    rot_b.clear();
    int hist[HISTO_LENGTH] = {0};
    int nm = 0, rc;
    for (int i = 0; i < nq; i++) {
        if (!queries[(size_t)i].active || cnt[(size_t)i] == 0) continue;
        Entry e[1];
        int found = 0, walked = 0;
        for (; walked < K && found < 1; walked++) {
            const uint32_t key = keys[(size_t)i * K + walked];
            if (key == 0xFFFFFFFFu) break;
            const int idx = m->perm[(size_t)(key & 0xFFFFu)];
            if (kp_to_query[idx] >= 0) continue;
            e[0].idx = idx;
            e[0].dist = (int)(key >> 16);
            found++;
        }
        if (found < 1 && walked == K && cnt[(size_t)i] > K) {
            gate.assign((size_t)F->n, INT_MAX);
            for (int k = 0; k < F->n; k++)
                if ((F->excluded && F->excluded[k]) || kp_to_query[k] >= 0) gate[(size_t)k] = 0;
            if ((rc = rerun_single(m, queries[(size_t)i], qdesc + (size_t)i * 32, gate, 1, e, &found))) return rc;
        }
        if (found == 0) continue;
        if (e[0].dist <= max_dist) {
            kp_to_query[e[0].idx] = i;
            nm++;
            if (check_orientation) {
                const int b = rot_bin(q_angle[i], F->angle[e[0].idx]);
                rot_item.push_back(e[0].idx);
                rot_b.push_back(b);
                hist[b]++;
            }
        }
    }
    if (check_orientation) apply_rot_hist(hist, rot_item, rot_b, kp_to_query, nm);
    *nmatches = nm;
    return SO_OK;
}

}  // namespace

extern "C" {

int so_search_window_best(so_matcher* m, const so_frame_view* KF, int32_t nq, const uint8_t* valid, const float* u,
                          const float* v, const float* radius, const int32_t* pred_level, const uint8_t* qdesc,
                          int chi2_gate, const float* inv_sigma2, int32_t* best_idx, int32_t* best_dist) {
    if (!m || !frame_ok(KF) || nq < 0) return SO_ERR_INVALID_ARG;
    if (nq > 0 && (!valid || !u || !v || !radius || !pred_level || !qdesc || !best_idx || !best_dist))
        return SO_ERR_INVALID_ARG;
    if (chi2_gate && (!inv_sigma2 || KF->nlevels > 8)) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    const bool reuse = take_reuse(m);
    for (int i = 0; i < nq; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
    }
    if (nq == 0 || KF->n == 0) return SO_OK;
    for (int l = 0; l < 8; l++) m->inv_sigma2[l] = (chi2_gate && l < KF->nlevels) ? inv_sigma2[l] : 0.f;
    so_frame_view view = *KF;
    view.excluded = nullptr;  // Fuse / SearchBySim3 look at every keypoint of the keyframe
    int rc = upload_frame(m, &view, nullptr, reuse);
    if (rc) return rc;
    if ((rc = ensure_queries(m, nq))) return rc;
    MatchQuery* hq = (MatchQuery*)m->h_q.p;
    for (int i = 0; i < nq; i++) {
        MatchQuery& q = hq[i];
        init_query(q);
        q.active = valid[i] != 0;
        q.u = u[i];
        q.v = v[i];
        q.r = radius[i];
        q.min_level = pred_level[i] - 1;  // "kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel" (:838)
        q.max_level = pred_level[i];
        if (pred_level[i] < 0) q.active = 0;
        if (chi2_gate) q.flags |= kQChi2Gate;
    }
    memcpy(m->h_qdesc.p, qdesc, (size_t)nq * 32);
    if ((rc = run_topk(m, nq, 1))) return rc;
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;
    for (int i = 0; i < nq; i++)
        if (keys[i] != 0xFFFFFFFFu) {
            best_idx[i] = m->perm[(size_t)(keys[i] & 0xFFFFu)];
            best_dist[i] = (int32_t)(keys[i] >> 16);
        }
    return SO_OK;
}

int so_search_window_greedy(so_matcher* m, const so_frame_view* F, int32_t nq, const uint8_t* valid, const float* u,
                            const float* v, const float* radius, const int32_t* min_level, const int32_t* max_level,
                            const uint8_t* qdesc, const float* q_angle, int32_t max_dist, int check_orientation,
                            int32_t* kp_to_query, int32_t* nmatches) {
    if (!m || !frame_ok(F) || nq < 0 || !kp_to_query || !nmatches) return SO_ERR_INVALID_ARG;
    if (nq > 0 && (!valid || !u || !v || !radius || !min_level || !max_level || !qdesc)) return SO_ERR_INVALID_ARG;
    if (check_orientation && nq > 0 && (!q_angle || !F->angle)) return SO_ERR_INVALID_ARG;
    const bool reuse = take_reuse(m);
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    *nmatches = 0;
    for (int k = 0; k < F->n; k++) kp_to_query[k] = -1;
    if (nq == 0 || F->n == 0) return SO_OK;
    constexpr int K = 4;
    int rc = upload_frame(m, F, nullptr, reuse);
    if (rc) return rc;
    if ((rc = ensure_queries(m, nq))) return rc;
    MatchQuery* hq = (MatchQuery*)m->h_q.p;
    for (int i = 0; i < nq; i++) {
        MatchQuery& q = hq[i];
        init_query(q);
        q.active = valid[i] != 0;
        q.u = u[i];
        q.v = v[i];
        q.r = radius[i];
        q.min_level = min_level[i];
        q.max_level = max_level[i];
    }
    memcpy(m->h_qdesc.p, qdesc, (size_t)nq * 32);
    if ((rc = run_topk(m, nq, K))) return rc;
    return resolve_greedy(m, F, nq, hq, qdesc, q_angle, max_dist, check_orientation, K, kp_to_query, nmatches);
}

}  // extern "C"

// =====================================================================================================
// Keyframe-side map-point searches with the projection on the device: Fuse x2, SearchBySim3,
// SearchByProjection(KeyFrame, Scw) and SearchByProjection(Frame, KeyFrame).  Per call: the target's candidates and
// the map points' fields go up in one staging block; project_queries_kernel (thread per map point) writes the window
// queries in HBM, topk_window_kernel (wave per query) reads them from there - no host hop between the two - and the
// host receives the K-lists plus a compact copy of the queries (for the sequential resolve and the parity tests).
// =====================================================================================================
namespace {

// cv::Mat conventions of oracle/project_oracle.h, evaluated once per call on the host.
void sim3_decompose(const float* S, float* A12, float* Ow) {  // ORBmatcher.cc:272-277 = :901-906
    const double d = (double)S[0] * (double)S[0] + (double)S[1] * (double)S[1] + (double)S[2] * (double)S[2];
    const float scw = (float)std::sqrt(d);
    const float inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) A12[4 * r + c] = S[4 * r + c] * inv;  // Rcw = sRcw / scw, tcw = Scw.col(3) / scw
    for (int j = 0; j < 3; j++) {  // Ow = -Rcw.t() * tcw
        const double s = (double)A12[0 + j] * (double)A12[3] + (double)A12[4 + j] * (double)A12[7] + (double)A12[8 + j] * (double)A12[11];
        Ow[j] = (float)(-s);
    }
}

void camera_center(const float* T, float* Ow) {  // KeyFrame::SetPose / Frame::UpdatePoseMatrices: Ow = -Rcw.t() * tcw
    for (int j = 0; j < 3; j++) {
        const double s = (double)T[0 + j] * (double)T[3] + (double)T[4 + j] * (double)T[7] + (double)T[8 + j] * (double)T[11];
        Ow[j] = (float)(-s);
    }
}

// sR12 = s12 * R12, sR21 = (1.0 / s12) * R12.t(), t21 = -sR21 * t12 (ORBmatcher.cc:1027-1029) as [sR | t] rows
void sim3_relative(float s12, const float* R12, const float* t12, float* B12_12, float* B21_12) {
    const float inv = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) {
            B12_12[4 * r + c] = R12[3 * r + c] * s12;
            B21_12[4 * r + c] = R12[3 * c + r] * inv;
        }
        B12_12[4 * r + 3] = t12[r];
    }
    for (int r = 0; r < 3; r++) {
        const double s = (double)B21_12[4 * r] * (double)t12[0] + (double)B21_12[4 * r + 1] * (double)t12[1] +
                         (double)B21_12[4 * r + 2] * (double)t12[2];
        B21_12[4 * r + 3] = (float)(-s);
    }
}

bool mappoints_ok(const so_mappoint_view* mp, bool need_normal) {
    if (!mp || mp->n < 0) return false;
    if (mp->n == 0) return true;
    return mp->Xw && mp->max_dist && mp->min_dist && mp->desc && (!need_normal || mp->normal);
}

void fill_target(ProjectSrc& S, const so_frame_view* F, const so_camera* cam, float log_scale_factor, float th) {
    S.fx = cam->fx; S.fy = cam->fy; S.cx = cam->cx; S.cy = cam->cy;
    S.bounds[0] = F->min_x; S.bounds[1] = F->max_x; S.bounds[2] = F->min_y; S.bounds[3] = F->max_y;
    for (int l = 0; l < 8; l++) S.scale[l] = l < F->nlevels ? F->scale_factors[l] : 0.f;
    S.nlevels = F->nlevels;
    S.log_scale_factor = log_scale_factor;
    S.th = th;
}

bool target_ok(const so_frame_view* F, const so_camera* cam) {
    return frame_ok(F) && cam && F->scale_factors && F->nlevels >= 1 && F->nlevels <= 8;
}

// Stages the map points behind the resident candidates, runs projection + window search (K-lists of length K) and
// waits.  Afterwards m->h_keys / m->h_count hold the lists and `qw` points at the compact queries (host-mapped).
struct ProjStage {
    size_t off_qdesc, off_xw, off_nrm, off_maxd, off_mind, off_valid, staged_end;
    bool shared = false;  // fields and descriptors are those of m->mp_share's block; only `valid` was staged
};

// host half: the map points' fields into the staging block behind the candidates
int stage_projected(so_matcher* m, const so_mappoint_view* mp, ProjStage& P) {
    const int n = mp->n;
    P.shared = false;
    if (m->batching && m->jobs.size() == kBatchMaxJobs) {  // make room now: a flush takes the shared block with it
        const int frc = batch_flush(m);
        if (frc) return frc;
    }
    const so_matcher::MpShare& sh = m->mp_share;
    if (m->batching && sh.n == n && sh.desc == mp->desc && sh.xw == mp->Xw && sh.nrm == mp->normal && sh.maxd == mp->max_dist &&
        sh.mind == mp->min_dist) {
        const uint8_t* sb = (const uint8_t*)m->hb_in.p + kBatchTableBytes + sh.base;
        const size_t sn = (size_t)n;
        if (!memcmp(sb + sh.off_qdesc, mp->desc, 32 * sn) && !memcmp(sb + sh.off_xw, mp->Xw, 12 * sn) &&
            (!mp->normal || !memcmp(sb + sh.off_nrm, mp->normal, 12 * sn)) && !memcmp(sb + sh.off_maxd, mp->max_dist, 4 * sn) &&
            !memcmp(sb + sh.off_mind, mp->min_dist, 4 * sn)) {
            P.shared = true;
            P.off_qdesc = sh.off_qdesc; P.off_xw = sh.off_xw; P.off_nrm = sh.off_nrm; P.off_maxd = sh.off_maxd; P.off_mind = sh.off_mind;
            P.off_valid = m->frame_end;
            P.staged_end = align256(P.off_valid + sn);
            int rc;
            if ((rc = m->h_in.ensure_keep(P.staged_end + 256, m->frame_end))) return rc;
            uint8_t* hb = (uint8_t*)m->h_in.p;
            if (mp->valid) memcpy(hb + P.off_valid, mp->valid, sn);
            else memset(hb + P.off_valid, 1, sn);
            m->off_qdesc = P.off_qdesc;
            m->off_q = P.staged_end;
            m->h_q.p = nullptr;
            m->h_qdesc.p = const_cast<uint8_t*>(sb) + sh.off_qdesc;
            return SO_OK;
        }
    }
    P.off_qdesc = m->frame_end;
    P.off_xw = align256(P.off_qdesc + 32 * (size_t)n);
    P.off_nrm = align256(P.off_xw + 12 * (size_t)n);
    P.off_maxd = align256(P.off_nrm + 12 * (size_t)n);
    P.off_mind = align256(P.off_maxd + 4 * (size_t)n);
    P.off_valid = align256(P.off_mind + 4 * (size_t)n);
    P.staged_end = align256(P.off_valid + (size_t)n);
    int rc;
    if ((rc = m->h_in.ensure_keep(P.staged_end + 256, m->frame_end))) return rc;
    uint8_t* hb = (uint8_t*)m->h_in.p;
    memcpy(hb + P.off_qdesc, mp->desc, 32 * (size_t)n);
    memcpy(hb + P.off_xw, mp->Xw, 12 * (size_t)n);
    if (mp->normal) memcpy(hb + P.off_nrm, mp->normal, 12 * (size_t)n);
    memcpy(hb + P.off_maxd, mp->max_dist, 4 * (size_t)n);
    memcpy(hb + P.off_mind, mp->min_dist, 4 * (size_t)n);
    if (mp->valid) memcpy(hb + P.off_valid, mp->valid, (size_t)n);
    else memset(hb + P.off_valid, 1, (size_t)n);
    m->off_qdesc = P.off_qdesc;
    m->off_q = P.staged_end;
    m->h_q.p = nullptr;
    m->h_qdesc.p = hb + P.off_qdesc;
    return SO_OK;
}

int run_projected(so_matcher* m, const so_mappoint_view* mp, ProjectSrc& S, int K, const MatchQueryW** qw) {
    const int n = mp->n;
    if (m->batching) {
        last_error_ref() = "this call cannot be part of a matcher batch";
        return SO_ERR_INVALID_ARG;
    }
    ProjStage P;
    int rc;
    if ((rc = stage_projected(m, mp, P))) return rc;
    const size_t off_qdesc = P.off_qdesc, off_xw = P.off_xw, off_nrm = P.off_nrm, off_maxd = P.off_maxd, off_mind = P.off_mind,
                 off_valid = P.off_valid, staged_end = P.staged_end;
    const size_t off_q = staged_end;  // device only: written by project_queries_kernel
    const size_t total_dev = off_q + sizeof(MatchQuery) * (size_t)n;
    uint8_t* hb = (uint8_t*)m->h_in.p;
    if (m->d_in.cap < total_dev) {
        if ((rc = m->d_in.ensure(std::max(m->h_in.cap, total_dev)))) return rc;
        m->dirty_from = 0;
    }
    const size_t keys_bytes = align256(sizeof(uint32_t) * (size_t)n * K);
    const size_t cnt_bytes = align256(sizeof(int32_t) * (size_t)n);
    if ((rc = m->h_out.ensure(keys_bytes + cnt_bytes + sizeof(MatchQueryW) * (size_t)n))) return rc;
    uint8_t* db = (uint8_t*)m->d_in.p;
    S.Xw = (const float*)(db + off_xw);
    S.normal = (const float*)(db + off_nrm);
    S.max_dist = (const float*)(db + off_maxd);
    S.min_dist = (const float*)(db + off_mind);
    S.valid = db + off_valid;
    S.n = n;
    hipStream_t s = m->stream;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t from = std::min(m->dirty_from, off_qdesc) & ~(size_t)15;
    launch_stage_in(db + from, hb + from, staged_end - from, s);
    m->dirty_from = SIZE_MAX;
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_project_queries(S, (MatchQuery*)(db + off_q), (MatchQueryW*)((uint8_t*)m->h_out.dev + keys_bytes + cnt_bytes), s);
    launch_topk_window(frame_dev(m), db + off_q, false, (const uint4*)(db + off_qdesc), n, K, (uint32_t*)m->h_out.dev,
                       (int32_t*)((uint8_t*)m->h_out.dev + keys_bytes), s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    const auto t1 = std::chrono::steady_clock::now();
    SO_HIP(hipStreamSynchronize(s));
    const auto t2 = std::chrono::steady_clock::now();
    m->stat[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    m->stat[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
    m->stat[2] += 2.0;
    m->stat[3] += (double)(staged_end - from);
    m->h_keys.p = m->h_out.p;
    m->h_count.p = (uint8_t*)m->h_out.p + keys_bytes;
    *qw = (const MatchQueryW*)((const uint8_t*)m->h_out.p + keys_bytes + cnt_bytes);
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    return SO_OK;
}

void export_queries(const so_window_queries* out, const MatchQueryW* qw, int n) {
    if (!out) return;
    for (int i = 0; i < n; i++) {
        if (out->active) out->active[i] = qw[i].active;
        if (out->u) out->u[i] = qw[i].u;
        if (out->v) out->v[i] = qw[i].v;
        if (out->radius) out->radius[i] = qw[i].r;
        if (out->level) out->level[i] = qw[i].active ? qw[i].min_level + 1 : 0;
    }
}

void begin_call(so_matcher* m) {
    if (m->batching) return;  // a batch's statistics cover all of its flushes: they are reset by so_matcher_batch_begin only
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
}

// Best keypoint per projected map point (K = 1): Fuse x2 and each direction of SearchBySim3.
int projected_best(so_matcher* m, const so_frame_view* KF, const so_mappoint_view* mp, ProjectSrc& S, bool chi2_gate,
                   const float* inv_sigma2, bool reuse, int32_t* best_idx, int32_t* best_dist,
                   const so_window_queries* queries_out, std::function<void()> post = nullptr) {
    const int n = mp->n;
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
    }
    if (n == 0) {
        if (post) post();
        return SO_OK;
    }
    for (int l = 0; l < 8; l++) m->inv_sigma2[l] = (chi2_gate && l < KF->nlevels) ? inv_sigma2[l] : 0.f;
    so_frame_view view = *KF;
    view.excluded = nullptr;  // Fuse / SearchBySim3 look at every keypoint of the keyframe
    int rc = upload_frame(m, &view, nullptr, reuse);
    if (rc) return rc;
    S.level_above = 0;
    S.qflags = chi2_gate ? (uint32_t)kQChi2Gate : 0u;
    S.q_max_dist = 256;
    const int n_kf = KF->n;
    so_window_queries qout{};
    const bool want_q = queries_out != nullptr;
    if (want_q) qout = *queries_out;
    auto resolve = [n, n_kf, want_q, qout, best_idx, best_dist, post](const uint32_t* keys, const int32_t*, const MatchQueryW* qw,
                                                                     const std::vector<int>& perm) -> int {
        export_queries(want_q ? &qout : nullptr, qw, n);
        if (n_kf > 0)
            for (int i = 0; i < n; i++)
                if (keys[i] != 0xFFFFFFFFu) {
                    best_idx[i] = perm[(size_t)(keys[i] & 0xFFFFu)];
                    best_dist[i] = (int32_t)(keys[i] >> 16);
                }
        if (post) post();
        return SO_OK;
    };
    if (m->batching) {
        ProjStage P;
        if ((rc = stage_projected(m, mp, P))) return rc;
        const size_t offs[5] = {P.off_xw, P.off_nrm, P.off_maxd, P.off_mind, P.off_valid};
        return batch_defer(m, P.staged_end, n, 1, &S, offs, std::move(resolve), nullptr, 0, mp, P.shared);
    }
    const MatchQueryW* qw = nullptr;
    if ((rc = run_projected(m, mp, S, 1, &qw))) return rc;
    return resolve((const uint32_t*)m->h_keys.p, nullptr, qw, m->perm);
}

// Sequential greedy binding per projected map point: the two keyframe-side SearchByProjection overloads.
int projected_greedy(so_matcher* m, const so_frame_view* F, const so_mappoint_view* mp, ProjectSrc& S, int level_above,
                     const float* q_angle, int max_dist, int check_orientation, bool reuse, int32_t* kp_to_point,
                     int32_t* nmatches, const so_window_queries* queries_out) {
    constexpr int K = 4;
    const int n = mp->n;
    *nmatches = 0;
    for (int k = 0; k < F->n; k++) kp_to_point[k] = -1;
    if (n == 0) return SO_OK;
    int rc = upload_frame(m, F, nullptr, reuse);
    if (rc) return rc;
    S.level_above = level_above;
    S.qflags = 0;
    S.q_max_dist = 256;
    const MatchQueryW* qw = nullptr;
    if ((rc = run_projected(m, mp, S, K, &qw))) return rc;
    export_queries(queries_out, qw, n);
    if (F->n == 0) return SO_OK;
    std::vector<MatchQuery> hq((size_t)n);
    for (int i = 0; i < n; i++) hq[(size_t)i] = expand_query(qw[i]);
    return resolve_greedy(m, F, n, hq.data(), mp->desc, q_angle, max_dist, check_orientation, K, kp_to_point, nmatches);
}

int keep_within(int n, int32_t* best_idx, const int32_t* best_dist, int th) {
    int kept = 0;
    for (int i = 0; i < n; i++) {
        if (best_idx[i] >= 0 && best_dist[i] <= th) kept++;
        else best_idx[i] = -1;
    }
    return kept;
}

}  // namespace

extern "C" {

int so_fuse(so_matcher* m, const so_frame_view* KF, const so_camera* cam, const float* Tcw12, float log_scale_factor,
            const float* inv_level_sigma2, const so_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist,
            int32_t* n_fused, const so_window_queries* queries_out) {
    if (!m || !target_ok(KF, cam) || !Tcw12 || !inv_level_sigma2 || !mappoints_ok(mp, true) || !n_fused)
        return SO_ERR_INVALID_ARG;
    if (mp->n > 0 && (!best_idx || !best_dist)) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    begin_call(m);
    const bool reuse = take_reuse(m);
    *n_fused = 0;
    ProjectSrc S{};
    memcpy(S.A, Tcw12, sizeof(S.A));
    camera_center(Tcw12, S.Ow);  // pKF->GetCameraCenter()
    S.flags = kPAngleGate;
    fill_target(S, KF, cam, log_scale_factor, th);
    const int n = mp->n;
    return projected_best(m, KF, mp, S, true, inv_level_sigma2, reuse, best_idx, best_dist, queries_out,
                          [n, best_idx, best_dist, n_fused] { *n_fused = keep_within(n, best_idx, best_dist, TH_LOW); });  // :873
}

int so_fuse_sim3(so_matcher* m, const so_frame_view* KF, const so_camera* cam, const float* Scw12, float log_scale_factor,
                 const so_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist, int32_t* n_fused,
                 const so_window_queries* queries_out) {
    if (!m || !target_ok(KF, cam) || !Scw12 || !mappoints_ok(mp, true) || !n_fused) return SO_ERR_INVALID_ARG;
    if (mp->n > 0 && (!best_idx || !best_dist)) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    begin_call(m);
    const bool reuse = take_reuse(m);
    *n_fused = 0;
    ProjectSrc S{};
    sim3_decompose(Scw12, S.A, S.Ow);
    S.flags = kPAngleGate;
    fill_target(S, KF, cam, log_scale_factor, th);
    const int n = mp->n;
    return projected_best(m, KF, mp, S, false, nullptr, reuse, best_idx, best_dist, queries_out,
                          [n, best_idx, best_dist, n_fused] { *n_fused = keep_within(n, best_idx, best_dist, TH_LOW); });  // :995
}

int so_search_by_sim3(so_matcher* m, const so_frame_view* KF1, const so_frame_view* KF2, const so_camera* cam,
                      const float* T1w12, const float* T2w12, float s12, const float* R12, const float* t12,
                      float log_scale_factor1, float log_scale_factor2, const so_mappoint_view* mp1,
                      const so_mappoint_view* mp2, float th, int32_t* match12, int32_t* n_found,
                      const so_window_queries* queries1_out, const so_window_queries* queries2_out) {
    if (!m || !target_ok(KF1, cam) || !target_ok(KF2, cam) || !T1w12 || !T2w12 || !R12 || !t12 ||
        !mappoints_ok(mp1, false) || !mappoints_ok(mp2, false) || !n_found)
        return SO_ERR_INVALID_ARG;
    if (mp1->n > 0 && !match12) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    begin_call(m);
    (void)take_reuse(m);
    *n_found = 0;
    if (m->batching) {
        last_error_ref() = "so_search_by_sim3 cannot be part of a matcher batch (its second pass depends on nothing, but the "
                           "agreement step needs both): call it outside";
        return SO_ERR_INVALID_ARG;
    }
    float B12[12], B21[12];
    sim3_relative(s12, R12, t12, B12, B21);
    const int N1 = mp1->n, N2 = mp2->n;
    std::vector<int32_t> vnMatch1((size_t)N1, -1), vnMatch2((size_t)N2, -1), dist((size_t)std::max(N1, N2) + 1);
    {   // Transform from KF1 to KF2 and search (:1054-1127)
        ProjectSrc S{};
        memcpy(S.A, T1w12, sizeof(S.A));
        memcpy(S.B, B21, sizeof(S.B));
        S.flags = kPChain;
        fill_target(S, KF2, cam, log_scale_factor2, th);
        const int rc = projected_best(m, KF2, mp1, S, false, nullptr, false, vnMatch1.data(), dist.data(), queries1_out);
        if (rc) return rc;
        keep_within(N1, vnMatch1.data(), dist.data(), TH_HIGH);
    }
    {   // Transform from KF2 to KF1 and search (:1130-1203)
        ProjectSrc S{};
        memcpy(S.A, T2w12, sizeof(S.A));
        memcpy(S.B, B12, sizeof(S.B));
        S.flags = kPChain;
        fill_target(S, KF1, cam, log_scale_factor1, th);
        const int rc = projected_best(m, KF1, mp2, S, false, nullptr, false, vnMatch2.data(), dist.data(), queries2_out);
        if (rc) return rc;
        keep_within(N2, vnMatch2.data(), dist.data(), TH_HIGH);
    }
    int nFound = 0;  // Check agreement (:1205-1218)
    for (int i1 = 0; i1 < N1; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[(size_t)i1];
        if (idx2 >= 0 && idx2 < N2 && vnMatch2[(size_t)idx2] == i1) {
            match12[i1] = idx2;
            nFound++;
        }
    }
    *n_found = nFound;
    return SO_OK;
}

int so_search_by_projection_sim3(so_matcher* m, const so_frame_view* KF, const so_camera* cam, const float* Scw12,
                                 float log_scale_factor, const so_mappoint_view* mp, int th, int32_t* kp_to_point,
                                 int32_t* nmatches, const so_window_queries* queries_out) {
    if (!m || !target_ok(KF, cam) || !Scw12 || !mappoints_ok(mp, true) || !kp_to_point || !nmatches)
        return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    begin_call(m);
    const bool reuse = take_reuse(m);
    ProjectSrc S{};
    sim3_decompose(Scw12, S.A, S.Ow);
    S.flags = kPAngleGate;
    fill_target(S, KF, cam, log_scale_factor, (float)th);
    return projected_greedy(m, KF, mp, S, 0, nullptr, TH_LOW, 0, reuse, kp_to_point, nmatches, queries_out);
}

int so_search_by_projection_keyframe(so_matcher* m, const so_frame_view* F, const so_camera* cam, const float* Tcw12,
                                     float log_scale_factor, const so_mappoint_view* mp, const float* mp_angle, float th,
                                     int32_t orb_dist, int check_orientation, int32_t* kp_to_point, int32_t* nmatches,
                                     const so_window_queries* queries_out) {
    if (!m || !target_ok(F, cam) || !Tcw12 || !mappoints_ok(mp, false) || !kp_to_point || !nmatches)
        return SO_ERR_INVALID_ARG;
    if (check_orientation && mp->n > 0 && (!mp_angle || !F->angle)) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    begin_call(m);
    const bool reuse = take_reuse(m);
    ProjectSrc S{};
    memcpy(S.A, Tcw12, sizeof(S.A));
    camera_center(Tcw12, S.Ow);  // Ow = -Rcw.t() * tcw, :1362
    S.flags = kPFrameForm;
    fill_target(S, F, cam, log_scale_factor, th);
    return projected_greedy(m, F, mp, S, 1, mp_angle, orb_dist, check_orientation, reuse, kp_to_point, nmatches,
                            queries_out);
}

// ---- HBM-resident keyframes -------------------------------------------------------------------------------------
}  // extern "C"

namespace {
// Device blocks of destroyed keyframes are kept for the next ones (a keyframe leaves the covisibility ring as another
// enters it; hipMalloc / hipFree cost 50-100 us each and synchronise the device).
struct KfBlock {
    uint8_t* p;
    size_t cap;
    int device;
};
std::mutex g_kf_pool_mu;
std::vector<KfBlock> g_kf_pool;
constexpr size_t kKfPoolMax = 48;

uint8_t* kf_block_take(int device, size_t bytes, size_t* cap_out) {
    std::lock_guard<std::mutex> lk(g_kf_pool_mu);
    for (size_t i = 0; i < g_kf_pool.size(); i++)
        if (g_kf_pool[i].device == device && g_kf_pool[i].cap >= bytes && g_kf_pool[i].cap <= 2 * bytes + 65536) {
            uint8_t* p = g_kf_pool[i].p;
            *cap_out = g_kf_pool[i].cap;
            g_kf_pool.erase(g_kf_pool.begin() + (long)i);
            return p;
        }
    return nullptr;
}

bool kf_block_give(int device, uint8_t* p, size_t cap) {
    std::lock_guard<std::mutex> lk(g_kf_pool_mu);
    if (g_kf_pool.size() >= kKfPoolMax) return false;
    g_kf_pool.push_back(KfBlock{p, cap, device});
    return true;
}
}  // namespace

extern "C" {

int so_kframe_create(so_matcher* m, const so_frame_view* KF, const so_featvec* fv, const float* level_sigma2, so_kframe** out) {
    if (!m || !out || !frame_ok(KF) || !KF->scale_factors || KF->nlevels < 1 || KF->nlevels > 8 || m->batching) return SO_ERR_INVALID_ARG;
    if (fv && (!featvec_ok(fv, KF->n) || !level_sigma2)) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    SO_HIP(hipSetDevice(m->device));
    (void)take_reuse(m);
    so_kframe* k = new so_kframe();
    k->device = m->device;
    k->n = KF->n;
    so_frame_view view = *KF;
    view.excluded = nullptr;
    int rc = upload_frame(m, &view, nullptr, false);  // the grid layout, staged in m->h_in [0, frame_end)
    if (rc) { delete k; return rc; }
    k->g_oct = m->off_oct; k->g_desc = m->off_desc; k->g_cols = m->off_cols; k->g_end = m->frame_end;
    k->n_grid = m->n_cand;
    k->perm_grid = m->perm;
    k->min_x = KF->min_x; k->max_x = KF->max_x; k->min_y = KF->min_y; k->max_y = KF->max_y;
    k->grid_inv_w = KF->grid_inv_w; k->grid_inv_h = KF->grid_inv_h; k->grid_min_y = m->grid_min_y;
    k->nlevels = KF->nlevels;
    for (int l = 0; l < 8; l++) {
        k->scale[l] = l < KF->nlevels ? KF->scale_factors[l] : 0.f;
        k->sigma2[l] = (level_sigma2 && l < KF->nlevels) ? level_sigma2[l] : 0.f;
    }
    if (KF->angle) k->angle.assign(KF->angle, KF->angle + KF->n);
    // (no null-stream copies: another thread of the process may be capturing a graph; uploads go through the handle's stream)
    size_t total = k->g_end;
    if (fv) {
        const size_t nn = (size_t)(fv->n_nodes > 0 ? fv->off[fv->n_nodes] - fv->off[0] : 0);
        total += align256(align256(align256(sizeof(float2) * nn) + nn) + 32 * nn) + 256;
    }
    hipError_t e = hipSuccess;
    k->d = kf_block_take(m->device, total + 256, &k->d_cap);
    if (!k->d) {
        k->d_cap = total + 256 + (total >> 3);  // a little slack: the next keyframe has a few more keypoints inside the grid
        e = hipMalloc((void**)&k->d, k->d_cap);
    }
    // the grid layout waits in a second pinned block while the staging block takes the node layout; both go up with the
    // copy kernel on the handle's queue, one wait at the end
    if (e == hipSuccess && m->h_res.ensure(k->g_end + 256) != SO_OK) e = hipErrorOutOfMemory;
    if (e == hipSuccess) {
        memcpy(m->h_res.p, m->h_in.p, k->g_end);
        launch_stage_in(k->d, m->h_res.p, k->g_end, m->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !fv) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(m->stream);
        if (k->d) (void)hipFree(k->d);
        delete k;
        m->resident_n = -1;
        m->dirty_from = 0;
        return hip_fail(e, "so_kframe_create", __FILE__, __LINE__);
    }
    if (fv) {
        rc = upload_featvec_candidates(m, KF->n, KF->x, KF->y, KF->octave, KF->desc, fv, nullptr);
        if (rc) {  // the copy kernel that reads m->h_res may still be running: wait, then give the block back
            (void)hipStreamSynchronize(m->stream);
            if (k->d && !kf_block_give(k->device, k->d, k->d_cap)) (void)hipFree(k->d);
            delete k;
            m->resident_n = -1;
            m->dirty_from = 0;
            return rc;
        }
        k->has_nodes = true;
        k->n_xy = k->g_end;
        k->n_oct = k->g_end + m->off_oct;
        k->n_desc = k->g_end + m->off_desc;
        k->n_node = m->n_cand;
        k->perm_node = m->perm;
        k->node_id.assign(fv->node_id, fv->node_id + fv->n_nodes);
        k->node_off.resize((size_t)fv->n_nodes + 1);
        for (int a = 0; a <= fv->n_nodes; a++) k->node_off[(size_t)a] = fv->off[a] - fv->off[0];
        const size_t node_bytes = align256(m->off_desc + 32 * (size_t)k->n_node);
        if (k->g_end + node_bytes > k->d_cap) e = hipErrorInvalidValue;
        if (e == hipSuccess && node_bytes > 0) {
            launch_stage_in(k->d + k->g_end, m->h_in.p, node_bytes, m->stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    }
    m->resident_n = -1;
    m->dirty_from = 0;
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(m->stream);
        if (k->d) (void)hipFree(k->d);
        delete k;
        return hip_fail(e, "so_kframe_create", __FILE__, __LINE__);
    }
    *out = k;
    return SO_OK;
}

void so_kframe_destroy(so_kframe* k) {
    if (!k) return;
    if (k->d && !kf_block_give(k->device, k->d, k->d_cap)) {
        (void)hipSetDevice(k->device);
        (void)hipFree(k->d);
    }
    delete k;
}

}  // extern "C"

namespace {
// Runs `call` as a one-job batch when the handle is not inside one.
template <typename F>
int as_batch(so_matcher* m, F&& call) {
    if (m->batching) return call();
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    m->batching = true;
    m->mp_share.n = -1;
    m->tri_share.n = -1;
    m->jobs.clear();
    m->hb_used = m->dq_used = m->out_used = 0;
    int rc = call();
    const int rc2 = batch_flush(m);
    m->batching = false;
    m->jobs.clear();
    m->src = nullptr;
    m->resident_n = -1;
    m->dirty_from = 0;
    return rc ? rc : rc2;
}

// what a resident keyframe contributes to the handle's per-call frame state
void adopt_kframe(so_matcher* m, const so_kframe* k, int layout) {
    m->src = nullptr;
    m->n_cand = layout == 0 ? k->n_grid : k->n_node;
    m->has_cols = layout == 0;
    m->min_x = k->min_x; m->min_y = k->min_y; m->grid_inv_w = k->grid_inv_w; m->grid_inv_h = k->grid_inv_h;
    m->grid_min_y = k->grid_min_y;
    m->off_oct = m->off_desc = m->off_cols = 0;
    m->resident_n = -1;
}
}  // namespace

extern "C" {

int so_fuse_kframe(so_matcher* m, const so_kframe* KF, const so_camera* cam, const float* Tcw12, float log_scale_factor,
                   const float* inv_level_sigma2, const so_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist,
                   int32_t* n_fused, const so_window_queries* queries_out) {
    if (!m || !KF || !cam || !Tcw12 || !inv_level_sigma2 || !mappoints_ok(mp, true) || !n_fused) return SO_ERR_INVALID_ARG;
    if (mp->n > 0 && (!best_idx || !best_dist)) return SO_ERR_INVALID_ARG;
    if (KF->device != m->device) {
        last_error_ref() = "resident keyframe lives on another device";
        return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(m->device));
    (void)take_reuse(m);
    *n_fused = 0;
    const int n = mp->n;
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
    }
    if (n == 0) return SO_OK;
    return as_batch(m, [&]() -> int {
        ProjectSrc S{};
        memcpy(S.A, Tcw12, sizeof(S.A));
        camera_center(Tcw12, S.Ow);
        S.flags = kPAngleGate;
        S.fx = cam->fx; S.fy = cam->fy; S.cx = cam->cx; S.cy = cam->cy;
        S.bounds[0] = KF->min_x; S.bounds[1] = KF->max_x; S.bounds[2] = KF->min_y; S.bounds[3] = KF->max_y;
        for (int l = 0; l < 8; l++) S.scale[l] = KF->scale[l];
        S.nlevels = KF->nlevels;
        S.log_scale_factor = log_scale_factor;
        S.th = th;
        S.level_above = 0;
        S.qflags = kQChi2Gate;
        S.q_max_dist = 256;
        adopt_kframe(m, KF, 0);
        m->has_limit = false;
        for (int l = 0; l < 8; l++) m->inv_sigma2[l] = l < KF->nlevels ? inv_level_sigma2[l] : 0.f;
        m->frame_end = 0;  // nothing of the frame is staged
        ProjStage P;
        int rc = stage_projected(m, mp, P);
        if (rc) return rc;
        so_window_queries qout{};
        const bool want_q = queries_out != nullptr;
        if (want_q) qout = *queries_out;
        const int n_kf = KF->n_grid;
        auto resolve = [n, n_kf, want_q, qout, best_idx, best_dist, n_fused](const uint32_t* keys, const int32_t*, const MatchQueryW* qw,
                                                                             const std::vector<int>& perm) -> int {
            export_queries(want_q ? &qout : nullptr, qw, n);
            if (n_kf > 0)
                for (int i = 0; i < n; i++)
                    if (keys[i] != 0xFFFFFFFFu) {
                        best_idx[i] = perm[(size_t)(keys[i] & 0xFFFFu)];
                        best_dist[i] = (int32_t)(keys[i] >> 16);
                    }
            *n_fused = keep_within(n, best_idx, best_dist, TH_LOW);
            return SO_OK;
        };
        const size_t offs[5] = {P.off_xw, P.off_nrm, P.off_maxd, P.off_mind, P.off_valid};
        return batch_defer(m, P.staged_end, n, 1, &S, offs, std::move(resolve), KF, 0, mp, P.shared);
    });
}

int so_fuse_kframe_map(so_matcher* m, const so_kframe* KF, const so_camera* cam, const float* Tcw12, float log_scale_factor,
                       const float* inv_level_sigma2, const so_map* map, int32_t n, const int32_t* slots, const uint8_t* valid,
                       float th, int32_t* best_idx, int32_t* best_dist, int32_t* n_fused, const so_window_queries* queries_out) {
    if (!m || !KF || !cam || !Tcw12 || !inv_level_sigma2 || !map || n < 0 || !n_fused) return SO_ERR_INVALID_ARG;
    if (n > 0 && (!slots || !best_idx || !best_dist)) return SO_ERR_INVALID_ARG;
    if (KF->device != m->device || map->device != m->device) {
        last_error_ref() = "resident keyframe / map lives on another device";
        return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(m->device));
    (void)take_reuse(m);
    *n_fused = 0;
    for (int i = 0; i < n; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
    }
    if (n == 0) return SO_OK;
    if (map->size.load() == 0) {  // an empty table has no rows to read (and no storage): every point is inactive
        if (queries_out)
            for (int i = 0; i < n; i++) {
                if (queries_out->active) queries_out->active[i] = 0;
                if (queries_out->u) queries_out->u[i] = 0.f;
                if (queries_out->v) queries_out->v[i] = 0.f;
                if (queries_out->radius) queries_out->radius[i] = 0.f;
                if (queries_out->level) queries_out->level[i] = 0;
            }
        return SO_OK;
    }
    return as_batch(m, [&]() -> int {
        ProjectSrc S{};
        memcpy(S.A, Tcw12, sizeof(S.A));
        camera_center(Tcw12, S.Ow);
        S.flags = kPAngleGate;
        S.fx = cam->fx; S.fy = cam->fy; S.cx = cam->cx; S.cy = cam->cy;
        S.bounds[0] = KF->min_x; S.bounds[1] = KF->max_x; S.bounds[2] = KF->min_y; S.bounds[3] = KF->max_y;
        for (int l = 0; l < 8; l++) S.scale[l] = KF->scale[l];
        S.nlevels = KF->nlevels;
        S.log_scale_factor = log_scale_factor;
        S.th = th;
        S.level_above = 0;
        S.qflags = kQChi2Gate;
        S.q_max_dist = 256;
        adopt_kframe(m, KF, 0);
        m->has_limit = false;
        for (int l = 0; l < 8; l++) m->inv_sigma2[l] = l < KF->nlevels ? inv_level_sigma2[l] : 0.f;
        m->frame_end = 0;  // nothing of the frame is staged; the block is [slots | valid]
        int rc;
        if (m->jobs.size() == kBatchMaxJobs && (rc = batch_flush(m))) return rc;
        const size_t sn = (size_t)n, off_valid = align256(4 * sn), staged_end = align256(off_valid + sn);
        if ((rc = m->h_in.ensure_keep(staged_end + 256, 0))) return rc;
        uint8_t* hb = (uint8_t*)m->h_in.p;
        memcpy(hb, slots, 4 * sn);
        if (valid) memcpy(hb + off_valid, valid, sn);
        else memset(hb + off_valid, 1, sn);
        m->off_qdesc = 0;  // (the descriptors are the map's)
        m->off_q = staged_end;
        m->h_q.p = nullptr;
        m->h_qdesc.p = nullptr;
        so_window_queries qout{};
        const bool want_q = queries_out != nullptr;
        if (want_q) qout = *queries_out;
        const int n_kf = KF->n_grid;
        auto resolve = [n, n_kf, want_q, qout, best_idx, best_dist, n_fused](const uint32_t* keys, const int32_t*, const MatchQueryW* qw,
                                                                             const std::vector<int>& perm) -> int {
            export_queries(want_q ? &qout : nullptr, qw, n);
            if (n_kf > 0)
                for (int i = 0; i < n; i++)
                    if (keys[i] != 0xFFFFFFFFu) {
                        best_idx[i] = perm[(size_t)(keys[i] & 0xFFFFu)];
                        best_dist[i] = (int32_t)(keys[i] >> 16);
                    }
            *n_fused = keep_within(n, best_idx, best_dist, TH_LOW);
            return SO_OK;
        };
        const size_t offs[5] = {0, 0, 0, 0, off_valid};
        return batch_defer(m, staged_end, n, 1, &S, offs, std::move(resolve), KF, 0, nullptr, false, map, 0);
    });
}

int so_search_for_triangulation_kframe(so_matcher* m, int32_t n1, const float* x1, const float* y1, const float* angle1,
                                       const uint8_t* desc1, const uint8_t* free1, const so_featvec* fv1, const so_kframe* kf2,
                                       const uint8_t* free2, const float* F12, float ex, float ey, int check_orientation,
                                       int32_t* matches12, int32_t* nmatches) {
    if (!m || !kf2 || !kf2->has_nodes || !nmatches || !matches12 || n1 < 0 || !F12) return SO_ERR_INVALID_ARG;
    if ((n1 > 0 && (!x1 || !y1 || !desc1 || !free1)) || (kf2->n > 0 && !free2)) return SO_ERR_INVALID_ARG;
    if (check_orientation && ((n1 > 0 && !angle1) || (kf2->n > 0 && kf2->angle.empty()))) return SO_ERR_INVALID_ARG;
    if (!featvec_ok(fv1, n1)) return SO_ERR_INVALID_ARG;
    if (kf2->device != m->device) {
        last_error_ref() = "resident keyframe lives on another device";
        return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(m->device));
    (void)take_reuse(m);
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    const so_featvec fv2{(int32_t)kf2->node_id.size(), kf2->node_id.data(), kf2->node_off.data(), nullptr};
    // (join_nodes only reads node ids)
    NodeJoin J;
    {
        int k1 = 0, k2 = 0;
        while (k1 < fv1->n_nodes && k2 < fv2.n_nodes) {
            if (fv1->node_id[k1] == fv2.node_id[k2]) {
                J.k1.push_back(k1++);
                J.k2.push_back(k2++);
            } else if (fv1->node_id[k1] < fv2.node_id[k2]) {
                k1 = fv_lower_bound(fv1, k1, fv2.node_id[k2]);
            } else {
                k2 = fv_lower_bound(&fv2, k2, fv1->node_id[k1]);
            }
        }
    }
    if (J.k1.empty() || n1 == 0 || kf2->n_node == 0) return SO_OK;
    return as_batch(m, [&]() -> int {
        adopt_kframe(m, kf2, 1);
        for (int l = 0; l < 8; l++) {
            m->scale[l] = kf2->scale[l];
            m->sigma2[l] = kf2->sigma2[l];
        }
        m->ex = ex;
        m->ey = ey;
        // staged block: [gate by position | queries | query descriptors]
        const int nc = kf2->n_node;
        m->has_limit = true;
        m->off_limit = 0;
        m->frame_end = align256(sizeof(int32_t) * (size_t)nc);
        int rc;
        if ((rc = m->h_in.ensure_keep(m->frame_end + 256, 0))) return rc;
        {
            int32_t* hl = (int32_t*)m->h_in.p;
            for (int r = 0; r < nc; r++) hl[r] = free2[kf2->perm_node[(size_t)r]] ? INT_MAX : 0;  // "|| pMP2" (:662)
        }
        // the queries are counted first and then written where they are staged (no intermediate vector, no second copy)
        int nq = 0;
        for (size_t j = 0; j < J.k1.size(); j++) {
            const int k1 = J.k1[j];
            for (int a = fv1->off[k1]; a < fv1->off[k1 + 1]; a++) nq += free1[fv1->idx[a]] ? 1 : 0;
        }
        if (nq == 0) return SO_OK;
        if ((rc = ensure_queries(m, nq))) return rc;
        std::vector<int> q_idx1((size_t)nq);
        {
            MatchQuery* hq = (MatchQuery*)m->h_q.p;
            int at = 0;
            for (size_t j = 0; j < J.k1.size(); j++) {
                const int k1 = J.k1[j], k2 = J.k2[j];
                const int cb = kf2->node_off[(size_t)k2], ce = kf2->node_off[(size_t)k2 + 1];
                for (int a = fv1->off[k1]; a < fv1->off[k1 + 1]; a++) {
                    const int idx1 = fv1->idx[a];
                    if (!free1[idx1]) continue;
                    MatchQuery& q = hq[at];
                    init_query(q);
                    q.active = 1;
                    q.flags = kQRange | kQEpipolar | kQPreferLast;
                    q.max_dist = TH_LOW;
                    q.c_begin = cb;
                    q.c_end = ce;
                    q.la = x1[idx1] * F12[0] + y1[idx1] * F12[3] + F12[6];
                    q.lb = x1[idx1] * F12[1] + y1[idx1] * F12[4] + F12[7];
                    q.lc = x1[idx1] * F12[2] + y1[idx1] * F12[5] + F12[8];
                    q_idx1[(size_t)at++] = idx1;
                }
            }
        }
        // Query descriptors: 4-byte indices into the keyframe's descriptor table, which the first such call of a batch stages
        // (32 n1 bytes, once) - a keyframe's twenty calls used to stage ~570 x 32 bytes each, twice.
        size_t staged_end = 0, off_qidx = 0;
        off_qidx = m->off_qdesc;
        {
            int32_t* hi = (int32_t*)m->h_qdesc.p;  // (ensure_queries left room for nq x 32 bytes here)
            for (int i = 0; i < nq; i++) hi[i] = q_idx1[(size_t)i];
        }
        staged_end = align256(off_qidx + 4 * (size_t)nq);
        so_matcher::TriShare& sh = m->tri_share;
        // (a full batch is flushed by batch_defer before this job joins it: whatever was shared goes with the flush)
        const bool same = m->jobs.size() < kBatchMaxJobs && sh.n == n1 && sh.desc == desc1 &&
                          memcmp((const uint8_t*)m->hb_in.p + kBatchTableBytes + sh.abs, desc1, 32 * (size_t)n1) == 0;
        size_t tab_in_block = 0;
        if (!same) {  // the table rides at the end of this job's block
            tab_in_block = staged_end;
            if ((rc = m->h_in.ensure_keep(staged_end + 32 * (size_t)n1 + 256, staged_end))) return rc;
            memcpy((uint8_t*)m->h_in.p + staged_end, desc1, 32 * (size_t)n1);
            staged_end += 32 * (size_t)n1;
        }
        const float* angle2 = kf2->angle.data();
        auto resolve = [m, nq, q_idx1 = std::move(q_idx1), angle1, angle2, check_orientation, matches12, nmatches](
                           const uint32_t* keys, const int32_t*, const MatchQueryW*, const std::vector<int>& perm) -> int {
            std::vector<int>&rot_item = m->scratch_rot_item, &rot_b = m->scratch_rot_b;
            rot_item.clear();
            rot_b.clear();
            int hist[HISTO_LENGTH] = {0};
            int nm = 0;
            for (int i = 0; i < nq; i++) {
                if (keys[i] == 0xFFFFFFFFu) continue;
                const int idx1 = q_idx1[(size_t)i];
                const int idx2 = perm[(size_t)(0xFFFF - (keys[i] & 0xFFFFu))];
                matches12[idx1] = idx2;
                nm++;
                if (check_orientation) {
                    const int b = rot_bin(angle1[idx1], angle2[idx2]);
                    rot_item.push_back(idx1);
                    rot_b.push_back(b);
                    hist[b]++;
                }
            }
            if (check_orientation) apply_rot_hist(hist, rot_item, rot_b, matches12, nm);
            *nmatches = nm;
            return SO_OK;
        };
        rc = batch_defer(m, staged_end, nq, 1, nullptr, nullptr, std::move(resolve), kf2, 1);
        if (rc == SO_OK) {
            so_matcher::BatchJob& J = m->jobs.back();
            J.q_indexed = true;
            J.off_qidx = off_qidx;
            if (!same) {
                sh.desc = desc1;
                sh.n = n1;
                sh.abs = J.base + tab_in_block;
            }
            J.qtab_abs = sh.abs;
        }
        return rc;
    });
}

// SearchForTriangulation of ONE resident keyframe against several resident neighbours with the queries built on the device
// (so_search_for_triangulation_kframes): per neighbour the host stages keyframe 2's gate, keyframe 1's free flags / node
// index by position and the node join (~6 KB instead of ~40 KB of query records), the projection launch of the batch
// writes the MatchQuery records into HBM, the search launch reads keyframe 1's descriptors where they are resident.
int so_search_for_triangulation_kframes(so_matcher* m, const so_kframe* kf1, const uint8_t* free1, int32_t n_neighbours,
                                        const so_tri_neighbour* nb, int check_orientation) {
    if (!m || !kf1 || !kf1->has_nodes || n_neighbours < 0 || (n_neighbours > 0 && !nb) || (kf1->n > 0 && !free1)) return SO_ERR_INVALID_ARG;
    if (kf1->device != m->device) {
        last_error_ref() = "resident keyframe lives on another device";
        return SO_ERR_INVALID_ARG;
    }
    if (check_orientation && kf1->n > 0 && kf1->angle.empty()) return SO_ERR_INVALID_ARG;
    for (int j = 0; j < n_neighbours; j++) {
        const so_tri_neighbour& N = nb[j];
        if (!N.kf2 || !N.kf2->has_nodes || N.kf2->device != m->device || !N.matches12 || !N.nmatches || (N.kf2->n > 0 && !N.free2)) return SO_ERR_INVALID_ARG;
        if (check_orientation && N.kf2->n > 0 && N.kf2->angle.empty()) return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(m->device));
    (void)take_reuse(m);
    const int n1p = kf1->n_node, nn1 = (int)kf1->node_id.size();
    if (n1p > 65535 || nn1 > 65535) return SO_ERR_CAPACITY;
    // keyframe 1 by position of its node order: free flag and node index (the same for every neighbour)
    std::vector<uint8_t> free1p((size_t)std::max(n1p, 1));
    std::vector<uint16_t> node_of((size_t)std::max(n1p, 1));
    for (int a = 0; a < nn1; a++)
        for (int p = kf1->node_off[(size_t)a]; p < kf1->node_off[(size_t)a + 1]; p++) node_of[(size_t)p] = (uint16_t)a;
    for (int p = 0; p < n1p; p++) free1p[(size_t)p] = free1[kf1->perm_node[(size_t)p]] ? 1 : 0;
    return as_batch(m, [&]() -> int {
        int rc;
        for (int j = 0; j < n_neighbours; j++) {
            const so_tri_neighbour& N = nb[j];
            const so_kframe* kf2 = N.kf2;
            *N.nmatches = 0;
            for (int i = 0; i < kf1->n; i++) N.matches12[i] = -1;
            if (n1p == 0 || kf2->n_node == 0) continue;
            adopt_kframe(m, kf2, 1);
            for (int l = 0; l < 8; l++) {
                m->scale[l] = kf2->scale[l];
                m->sigma2[l] = kf2->sigma2[l];
            }
            m->ex = N.ex;
            m->ey = N.ey;
            // staged block: [gate of keyframe 2 by position | free1 by position | node of position | range per node]
            const int nc = kf2->n_node;
            m->has_limit = true;
            m->off_limit = 0;
            m->frame_end = align256(sizeof(int32_t) * (size_t)nc);
            const size_t o_free = m->frame_end, o_node = align256(o_free + (size_t)n1p), o_range = align256(o_node + 2 * (size_t)n1p),
                         end = align256(o_range + 8 * (size_t)std::max(nn1, 1));
            if ((rc = m->h_in.ensure_keep(end + 256, 0))) return rc;
            uint8_t* hb = (uint8_t*)m->h_in.p;
            {
                int32_t* hl = (int32_t*)hb;
                for (int r = 0; r < nc; r++) hl[r] = N.free2[kf2->perm_node[(size_t)r]] ? INT_MAX : 0;  // "|| pMP2" (:662)
            }
            memcpy(hb + o_free, free1p.data(), (size_t)n1p);
            memcpy(hb + o_node, node_of.data(), 2 * (size_t)n1p);
            int32_t* hr = (int32_t*)(hb + o_range);
            bool any = false;
            {   // merge join of the two node lists (both ascending by node id)
                size_t b2 = 0;
                const size_t nn2 = kf2->node_id.size();
                for (int a = 0; a < nn1; a++) {
                    while (b2 < nn2 && kf2->node_id[b2] < kf1->node_id[(size_t)a]) b2++;
                    if (b2 < nn2 && kf2->node_id[b2] == kf1->node_id[(size_t)a]) {
                        hr[2 * a] = kf2->node_off[b2];
                        hr[2 * a + 1] = kf2->node_off[b2 + 1];
                        any = true;
                    } else {
                        hr[2 * a] = hr[2 * a + 1] = 0;
                    }
                }
            }
            if (!any) continue;
            m->off_q = m->off_qdesc = end;  // (no query records or descriptors in the block)
            const float* angle1 = kf1->angle.data();
            const float* angle2 = kf2->angle.data();
            int32_t* matches12 = N.matches12;
            int32_t* nmatches = N.nmatches;
            auto resolve = [m, n1p, kf1, angle1, angle2, check_orientation, matches12, nmatches](
                               const uint32_t* keys, const int32_t*, const MatchQueryW*, const std::vector<int>& perm) -> int {
                std::vector<int>&rot_item = m->scratch_rot_item, &rot_b = m->scratch_rot_b;
                rot_item.clear();
                rot_b.clear();
                int hist[HISTO_LENGTH] = {0};
                int nm = 0;
                for (int p = 0; p < n1p; p++) {  // (node order = the order of the reference's two nested loops)
                    if (keys[p] == 0xFFFFFFFFu) continue;
                    const int idx1 = kf1->perm_node[(size_t)p];
                    const int idx2 = perm[(size_t)(0xFFFF - (keys[p] & 0xFFFFu))];
                    matches12[idx1] = idx2;
                    nm++;
                    if (check_orientation) {
                        const int b = rot_bin(angle1[idx1], angle2[idx2]);
                        rot_item.push_back(idx1);
                        rot_b.push_back(b);
                        hist[b]++;
                    }
                }
                if (check_orientation) apply_rot_hist(hist, rot_item, rot_b, matches12, nm);
                *nmatches = nm;
                return SO_OK;
            };
            if ((rc = batch_defer(m, end, n1p, 1, nullptr, nullptr, std::move(resolve), kf2, 1))) return rc;
            so_matcher::BatchJob& J = m->jobs.back();
            J.tri_kf1 = kf1;
            J.off_tfree1 = o_free;
            J.off_tnode = o_node;
            J.off_trange = o_range;
            memcpy(J.tri_F, N.F12, sizeof(J.tri_F));
            J.q_off = m->dq_used;  // the generated records live in db_q, like a projected job's
            m->dq_used += align256(sizeof(MatchQuery) * (size_t)n1p);
        }
        return SO_OK;
    });
}

// ---- the local-mapping thread's per-point loops between the matcher and local BA --------------------------------
static int triangulate_impl(so_matcher* m, const so_tri_keyframe* kf1, int32_t n_kf2, const so_tri_keyframe* kf2, float ratio_factor,
                            int32_t n, const int32_t* kf2_of_match, const float* xy1, const int32_t* octave1, const float* xy2,
                            const int32_t* octave2, uint8_t* ok, float* x3D, float* normal, float* max_dist, float* min_dist) {
    if (!m || !kf1 || n_kf2 < 0 || n < 0 || (n_kf2 > 0 && !kf2)) return SO_ERR_INVALID_ARG;
    if (n > 0 && (!kf2_of_match || !xy1 || !octave1 || !xy2 || !octave2 || !ok || !x3D || n_kf2 == 0)) return SO_ERR_INVALID_ARG;
    const bool with_fields = normal != nullptr;
    if (with_fields && n > 0 && (!max_dist || !min_dist)) return SO_ERR_INVALID_ARG;
    if (m->batching) {
        last_error_ref() = "so_triangulate_matches / so_triangulate_new_points cannot be part of a matcher batch";
        return SO_ERR_INVALID_ARG;
    }
    auto kf_ok = [](const so_tri_keyframe& k) { return k.scale_factors && k.level_sigma2 && k.nlevels >= 1 && k.nlevels <= 8; };
    if (!kf_ok(*kf1)) return SO_ERR_INVALID_ARG;
    for (int j = 0; j < n_kf2; j++)
        if (!kf_ok(kf2[j])) return SO_ERR_INVALID_ARG;
    for (int k = 0; k < n; k++)
        if (kf2_of_match[k] < 0 || kf2_of_match[k] >= n_kf2 || octave1[k] < 0 || octave1[k] >= kf1->nlevels || octave2[k] < 0 ||
            octave2[k] >= kf2[kf2_of_match[k]].nlevels)
            return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    (void)take_reuse(m);
    begin_call(m);
    if (n == 0) return SO_OK;
    auto fill = [](TriKeyframeDev& D, const so_tri_keyframe& k) {
        memcpy(D.Tcw, k.Tcw, sizeof(D.Tcw));
        camera_center(k.Tcw, D.Ow);
        D.fx = k.fx; D.fy = k.fy; D.cx = k.cx; D.cy = k.cy; D.invfx = k.invfx; D.invfy = k.invfy;
        for (int l = 0; l < 8; l++) {
            D.scale[l] = l < k.nlevels ? k.scale_factors[l] : 1.f;
            D.sigma2[l] = l < k.nlevels ? k.level_sigma2[l] : 1.f;
        }
    };
    const size_t o_kf = 0, o_of = align256(sizeof(TriKeyframeDev) * (size_t)n_kf2), o_xy1 = align256(o_of + 4 * (size_t)n),
                 o_xy2 = align256(o_xy1 + 8 * (size_t)n), o_o1 = align256(o_xy2 + 8 * (size_t)n), o_o2 = align256(o_o1 + 4 * (size_t)n),
                 end = align256(o_o2 + 4 * (size_t)n);
    int rc;
    if ((rc = m->h_in.ensure_keep(end + 256, 0))) return rc;
    if ((rc = m->d_in.ensure(end + 256))) return rc;
    const size_t ok_bytes = align256((size_t)n), x_bytes = align256(12 * (size_t)n), f_bytes = align256(4 * (size_t)n);
    if ((rc = m->h_out.ensure(ok_bytes + x_bytes + (with_fields ? x_bytes + 2 * f_bytes : 0)))) return rc;
    uint8_t* hb = (uint8_t*)m->h_in.p;
    for (int j = 0; j < n_kf2; j++) fill(((TriKeyframeDev*)(hb + o_kf))[j], kf2[j]);
    memcpy(hb + o_of, kf2_of_match, 4 * (size_t)n);
    memcpy(hb + o_xy1, xy1, 8 * (size_t)n);
    memcpy(hb + o_xy2, xy2, 8 * (size_t)n);
    memcpy(hb + o_o1, octave1, 4 * (size_t)n);
    memcpy(hb + o_o2, octave2, 4 * (size_t)n);
    m->resident_n = -1;  // the staging block no longer holds a frame
    m->dirty_from = 0;
    m->src = nullptr;
    uint8_t* db = (uint8_t*)m->d_in.p;
    TriArgs A;
    fill(A.kf1, *kf1);
    A.kf2 = (const TriKeyframeDev*)(db + o_kf);
    A.kf2_of = (const int32_t*)(db + o_of);
    A.xy1 = (const float2*)(db + o_xy1);
    A.xy2 = (const float2*)(db + o_xy2);
    A.oct1 = (const int32_t*)(db + o_o1);
    A.oct2 = (const int32_t*)(db + o_o2);
    A.ok = (uint8_t*)m->h_out.dev;
    A.x3D = (float*)((uint8_t*)m->h_out.dev + ok_bytes);
    A.normal = with_fields ? (float*)((uint8_t*)m->h_out.dev + ok_bytes + x_bytes) : nullptr;
    A.max_dist = with_fields ? (float*)((uint8_t*)m->h_out.dev + ok_bytes + 2 * x_bytes) : nullptr;
    A.min_dist = with_fields ? (float*)((uint8_t*)m->h_out.dev + ok_bytes + 2 * x_bytes + f_bytes) : nullptr;
    A.last_scale = kf1->scale_factors[kf1->nlevels - 1];
    A.ratio_factor = ratio_factor;
    A.n = n;
    hipStream_t s = m->stream;
    const auto t0 = std::chrono::steady_clock::now();
    launch_stage_in(db, hb, end, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_triangulate(A, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    const auto t1 = std::chrono::steady_clock::now();
    SO_HIP(hipStreamSynchronize(s));
    const auto t2 = std::chrono::steady_clock::now();
    m->stat[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    m->stat[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
    m->stat[2] += 1.0;
    m->stat[3] += (double)end;
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    const uint8_t* hok = (const uint8_t*)m->h_out.p;
    const float* hx = (const float*)((const uint8_t*)m->h_out.p + ok_bytes);
    const float* hn = (const float*)((const uint8_t*)m->h_out.p + ok_bytes + x_bytes);
    const float* hmx = (const float*)((const uint8_t*)m->h_out.p + ok_bytes + 2 * x_bytes);
    const float* hmn = (const float*)((const uint8_t*)m->h_out.p + ok_bytes + 2 * x_bytes + f_bytes);
    for (int k = 0; k < n; k++) {
        ok[k] = hok[k];
        if (!hok[k]) continue;
        memcpy(x3D + 3 * (size_t)k, hx + 3 * (size_t)k, 12);
        if (with_fields) {
            memcpy(normal + 3 * (size_t)k, hn + 3 * (size_t)k, 12);
            max_dist[k] = hmx[k];
            min_dist[k] = hmn[k];
        }
    }
    return SO_OK;
}

int so_triangulate_matches(so_matcher* m, const so_tri_keyframe* kf1, int32_t n_kf2, const so_tri_keyframe* kf2, float ratio_factor,
                           int32_t n, const int32_t* kf2_of_match, const float* xy1, const int32_t* octave1, const float* xy2,
                           const int32_t* octave2, uint8_t* ok, float* x3D) {
    return triangulate_impl(m, kf1, n_kf2, kf2, ratio_factor, n, kf2_of_match, xy1, octave1, xy2, octave2, ok, x3D, nullptr, nullptr,
                            nullptr);
}

int so_triangulate_new_points(so_matcher* m, const so_tri_keyframe* kf1, int32_t n_kf2, const so_tri_keyframe* kf2, float ratio_factor,
                              int32_t n, const int32_t* kf2_of_match, const float* xy1, const int32_t* octave1, const float* xy2,
                              const int32_t* octave2, uint8_t* ok, float* x3D, float* normal, float* max_dist, float* min_dist) {
    if (!normal || !max_dist || !min_dist) return SO_ERR_INVALID_ARG;
    return triangulate_impl(m, kf1, n_kf2, kf2, ratio_factor, n, kf2_of_match, xy1, octave1, xy2, octave2, ok, x3D, normal, max_dist,
                            min_dist);
}

int so_update_normal_and_depth(so_matcher* m, int32_t n_points, const int32_t* offsets, const float* obs_Ow, const float* Xw,
                               const float* ref_Ow, const float* ref_level_scale, const float* ref_last_scale, float* normal,
                               float* max_dist, float* min_dist) {
    if (!m || n_points < 0) return SO_ERR_INVALID_ARG;
    if (n_points > 0 && (!offsets || !Xw || !ref_Ow || !ref_level_scale || !ref_last_scale || !normal || !max_dist || !min_dist))
        return SO_ERR_INVALID_ARG;
    if (m->batching) {
        last_error_ref() = "so_update_normal_and_depth cannot be part of a matcher batch";
        return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(m->device));
    (void)take_reuse(m);
    begin_call(m);
    if (n_points == 0) return SO_OK;
    const int n = n_points;
    for (int p = 0; p < n; p++)
        if (offsets[p] > offsets[p + 1] || offsets[p] < 0) return SO_ERR_INVALID_ARG;
    const size_t total = (size_t)offsets[n];
    if (total > 0 && !obs_Ow) return SO_ERR_INVALID_ARG;
    const size_t o_off = 0, o_ow = align256(4 * ((size_t)n + 1)), o_x = align256(o_ow + 12 * total), o_r = align256(o_x + 12 * (size_t)n),
                 o_ls = align256(o_r + 12 * (size_t)n), o_ll = align256(o_ls + 4 * (size_t)n), end = align256(o_ll + 4 * (size_t)n);
    int rc;
    if ((rc = m->h_in.ensure_keep(end + 256, 0))) return rc;
    if ((rc = m->d_in.ensure(end + 256))) return rc;
    const size_t nb = align256(12 * (size_t)n), sb = align256(4 * (size_t)n);
    if ((rc = m->h_out.ensure(nb + 2 * sb))) return rc;
    uint8_t* hb = (uint8_t*)m->h_in.p;
    memcpy(hb + o_off, offsets, 4 * ((size_t)n + 1));
    if (total) memcpy(hb + o_ow, obs_Ow, 12 * total);
    memcpy(hb + o_x, Xw, 12 * (size_t)n);
    memcpy(hb + o_r, ref_Ow, 12 * (size_t)n);
    memcpy(hb + o_ls, ref_level_scale, 4 * (size_t)n);
    memcpy(hb + o_ll, ref_last_scale, 4 * (size_t)n);
    m->resident_n = -1;
    m->dirty_from = 0;
    m->src = nullptr;
    // a point without observations keeps what the caller passed in (UpdateNormalAndDepth returns early): seed the outputs
    memcpy(m->h_out.p, normal, 12 * (size_t)n);
    memcpy((uint8_t*)m->h_out.p + nb, max_dist, 4 * (size_t)n);
    memcpy((uint8_t*)m->h_out.p + nb + sb, min_dist, 4 * (size_t)n);
    uint8_t* db = (uint8_t*)m->d_in.p;
    NormalDepthArgs A;
    A.off = (const int32_t*)(db + o_off);
    A.obs_Ow = (const float*)(db + o_ow);
    A.Xw = (const float*)(db + o_x);
    A.ref_Ow = (const float*)(db + o_r);
    A.ref_level_scale = (const float*)(db + o_ls);
    A.ref_last_scale = (const float*)(db + o_ll);
    A.normal = (float*)m->h_out.dev;
    A.max_dist = (float*)((uint8_t*)m->h_out.dev + nb);
    A.min_dist = (float*)((uint8_t*)m->h_out.dev + nb + sb);
    A.n = n;
    A.kf_Ow = nullptr;
    A.obs_kf = A.ref_kf = nullptr;
    hipStream_t s = m->stream;
    launch_stage_in(db, hb, end, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_normal_depth(A, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    memcpy(normal, m->h_out.p, 12 * (size_t)n);
    memcpy(max_dist, (const uint8_t*)m->h_out.p + nb, 4 * (size_t)n);
    memcpy(min_dist, (const uint8_t*)m->h_out.p + nb + sb, 4 * (size_t)n);
    return SO_OK;
}

int so_update_normal_and_depth_indexed(so_matcher* m, int32_t n_points, const int32_t* offsets, const int32_t* obs_kf, int32_t n_kf,
                                       const float* kf_Ow, const float* Xw, const int32_t* ref_kf, const float* ref_level_scale,
                                       const float* ref_last_scale, float* normal, float* max_dist, float* min_dist) {
    if (!m || n_points < 0 || n_kf < 0) return SO_ERR_INVALID_ARG;
    if (n_points > 0 && (!offsets || !Xw || !ref_kf || !kf_Ow || !ref_level_scale || !ref_last_scale || !normal || !max_dist || !min_dist))
        return SO_ERR_INVALID_ARG;
    if (m->batching) {
        last_error_ref() = "so_update_normal_and_depth_indexed cannot be part of a matcher batch";
        return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(m->device));
    (void)take_reuse(m);
    begin_call(m);
    if (n_points == 0) return SO_OK;
    const int n = n_points;
    for (int p = 0; p < n; p++)
        if (offsets[p] > offsets[p + 1] || offsets[p] < 0 || ref_kf[p] < 0 || ref_kf[p] >= n_kf) return SO_ERR_INVALID_ARG;
    const size_t total = (size_t)offsets[n];
    if (total > 0 && !obs_kf) return SO_ERR_INVALID_ARG;
    for (size_t k = 0; k < total; k++)
        if (obs_kf[k] < 0 || obs_kf[k] >= n_kf) return SO_ERR_INVALID_ARG;
    const size_t o_off = 0, o_ok = align256(4 * ((size_t)n + 1)), o_c = align256(o_ok + 4 * total), o_x = align256(o_c + 12 * (size_t)n_kf),
                 o_r = align256(o_x + 12 * (size_t)n), o_ls = align256(o_r + 4 * (size_t)n), o_ll = align256(o_ls + 4 * (size_t)n),
                 end = align256(o_ll + 4 * (size_t)n);
    int rc;
    if ((rc = m->h_in.ensure_keep(end + 256, 0))) return rc;
    if ((rc = m->d_in.ensure(end + 256))) return rc;
    const size_t nb = align256(12 * (size_t)n), sb = align256(4 * (size_t)n);
    if ((rc = m->h_out.ensure(nb + 2 * sb))) return rc;
    uint8_t* hb = (uint8_t*)m->h_in.p;
    memcpy(hb + o_off, offsets, 4 * ((size_t)n + 1));
    if (total) memcpy(hb + o_ok, obs_kf, 4 * total);
    memcpy(hb + o_c, kf_Ow, 12 * (size_t)n_kf);
    memcpy(hb + o_x, Xw, 12 * (size_t)n);
    memcpy(hb + o_r, ref_kf, 4 * (size_t)n);
    memcpy(hb + o_ls, ref_level_scale, 4 * (size_t)n);
    memcpy(hb + o_ll, ref_last_scale, 4 * (size_t)n);
    m->resident_n = -1;
    m->dirty_from = 0;
    m->src = nullptr;
    memcpy(m->h_out.p, normal, 12 * (size_t)n);
    memcpy((uint8_t*)m->h_out.p + nb, max_dist, 4 * (size_t)n);
    memcpy((uint8_t*)m->h_out.p + nb + sb, min_dist, 4 * (size_t)n);
    uint8_t* db = (uint8_t*)m->d_in.p;
    NormalDepthArgs A;
    A.off = (const int32_t*)(db + o_off);
    A.obs_Ow = nullptr;
    A.ref_Ow = nullptr;
    A.Xw = (const float*)(db + o_x);
    A.ref_level_scale = (const float*)(db + o_ls);
    A.ref_last_scale = (const float*)(db + o_ll);
    A.normal = (float*)m->h_out.dev;
    A.max_dist = (float*)((uint8_t*)m->h_out.dev + nb);
    A.min_dist = (float*)((uint8_t*)m->h_out.dev + nb + sb);
    A.n = n;
    A.kf_Ow = (const float*)(db + o_c);
    A.obs_kf = (const int32_t*)(db + o_ok);
    A.ref_kf = (const int32_t*)(db + o_r);
    hipStream_t s = m->stream;
    launch_stage_in(db, hb, end, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_normal_depth(A, s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    memcpy(normal, m->h_out.p, 12 * (size_t)n);
    memcpy(max_dist, (const uint8_t*)m->h_out.p + nb, 4 * (size_t)n);
    memcpy(min_dist, (const uint8_t*)m->h_out.p + nb + sb, 4 * (size_t)n);
    return SO_OK;
}

int so_matcher_batch_begin(so_matcher* m) {
    if (!m || m->batching) return SO_ERR_INVALID_ARG;
    if (m->pend.mode != 0) {
        last_error_ref() = "a tracking search is pending on this handle";
        return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    m->batching = true;
    m->mp_share.n = -1;
    m->tri_share.n = -1;
    m->jobs.clear();
    m->hb_used = m->dq_used = m->out_used = 0;
    return SO_OK;
}

namespace {
void batch_close(so_matcher* m) {
    m->batching = false;
    m->jobs.clear();
    m->hb_used = m->dq_used = m->out_used = 0;
    m->mp_share.n = -1;
    m->tri_share.n = -1;
    m->src = nullptr;
    m->resident_n = -1;
    m->dirty_from = 0;
}
}  // namespace

int so_matcher_batch_end(so_matcher* m) {
    if (!m || !m->batching) return SO_ERR_INVALID_ARG;
    int rc = SO_OK;
    const hipError_t e = hipSetDevice(m->device);
    if (e != hipSuccess) rc = so::hip_fail(e, "hipSetDevice", __FILE__, __LINE__);
    else rc = batch_flush(m);  // (statistics accumulate over automatic mid-batch flushes too: reset in batch_begin only)
    batch_close(m);
    return rc;
}

int so_matcher_batch_abort(so_matcher* m) {
    if (!m) return SO_ERR_INVALID_ARG;
    if (!m->batching) return SO_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);  // an automatic flush may still be running
    batch_close(m);
    return SO_OK;
}

}  // extern "C"

// =====================================================================================================
// Tracking searches on device-resident inputs (so_dframe + so_map): the per-frame path of
// Tracking::TrackWithMotionModel / TrackLocalMap without a host round trip of keypoints, descriptors or map points.
// Per call the host sends the pose, one map slot per query (4 B) and the optional gates; projections, frustum
// tests, window queries and the K-lists are produced by ONE launch; the order-dependent resolve stays here.
// =====================================================================================================
namespace {

struct TrackStage {  // staging layout of one tracking search behind the (optional) limit gate
    size_t off_slot, off_skip, end;
};

TrackQuerySrc track_src(const so_matcher* m, const so_dframe* cur, const so_map* map, const float* Tcw12, float th) {
    TrackQuerySrc T{};
    T.Xw = map->d_Xw;
    T.normal = map->d_normal;
    T.max_dist = map->d_max;
    T.min_dist = map->d_min;
    T.desc = reinterpret_cast<const uint4*>(map->d_desc);
    memcpy(T.Tcw, Tcw12, 48);
    T.fx = cur->cam.fx; T.fy = cur->cam.fy; T.cx = cur->cam.cx; T.cy = cur->cam.cy;
    memcpy(T.bounds, cur->bounds, 16);
    for (int l = 0; l < 8; l++) T.scale[l] = cur->scale[l];
    T.nlevels = cur->nlevels;
    T.th = th;
    T.n_slots = map->size;
    (void)m;
    return T;
}

// Gates of a tracking search.  Small problems (<= 4096 candidates, <= 16384 queries: every tracking frame) carry them
// as bit masks inside the kernel arguments and read the per-query map slots straight from pinned host memory: ONE
// launch per search, nothing staged.  Larger ones stage [limit | slots | skip] with a copy kernel first.
struct TrackGates {
    const int32_t* slots;   // per query (host), or null: slot_base + query index
    int slot_base;
    const uint8_t* skip;    // per query (host), may be null
    const uint8_t* excluded;  // per KEYPOINT INDEX (host), may be null
};

void set_bits_from_excluded(const so_matcher* m, const uint8_t* excluded_by_idx, const std::vector<int32_t>* limit_by_idx,
                            TrackQuerySrc& T) {
    memset(T.excl_bits, 0, sizeof(T.excl_bits));
    const int nc = m->n_cand;
    for (int r = 0; r < nc; r++) {
        const int i = m->perm[(size_t)r];
        const bool ex = limit_by_idx ? (*limit_by_idx)[(size_t)i] == 0 : (excluded_by_idx && excluded_by_idx[i]);
        if (ex) T.excl_bits[r >> 5] |= 1u << (r & 31);
    }
}

int finish_topk_track(so_matcher* m, int nq, int K);

// launch half: everything up to and including the kernel launch; finish_topk_track waits and maps the results
}  // namespace

// Tracking stages of SEVERAL agents as one chain of launches (include/swarmorb.h: so_track_group_*).  Each member's
// so_track_stage_*_submit stages its inputs as always and records what it would launch; so_track_group_launch copies the
// recorded argument rows into HBM and launches search / resolve / PoseOptimization ONCE each with the agent as a grid
// dimension.  Members share one stream (handles created on one thread without private streams).
struct so_track_group {
    int device = 0;
    struct Rec {
        so_matcher* m = nullptr;
        bool search = false, resolve = false, pose = false;
        int mode = 0, range = 0;
        so::TrackGroupJob job;
        so::TrackResolveArgs res;
        so::PoseOptArgs pose_args;
    };
    std::vector<Rec> recs;
    int n_recs = 0;              // rows recorded since the last launch (recs keeps its capacity)
    PinBuf h_tab[2];
    DevBuf d_tab[2];
    int flip = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr, pe0 = nullptr, pe1 = nullptr;
    bool timed_search = false, timed_pose = false;
    int launches = 0;
};

namespace {

}  // namespace
struct LinkRec : so_track_group::Rec {};
static void link_rec_free(LinkRec* r) { delete r; }
namespace {

so_track_group::Rec& group_rec(so_matcher* m) {
    if (m->link_recording) return *m->link_rec;
    so_track_group* g = m->group;
    for (int i = 0; i < g->n_recs; i++)
        if (g->recs[(size_t)i].m == m) return g->recs[(size_t)i];
    if ((int)g->recs.size() <= g->n_recs) g->recs.emplace_back();
    so_track_group::Rec& r = g->recs[(size_t)g->n_recs++];
    r.m = m;
    r.search = r.resolve = r.pose = false;
    r.mode = r.range = 0;
    return r;
}

int launch_topk_track_async(so_matcher* m, TrackQuerySrc& T, int mode, int nq, int K, const TrackGates& G) {
    int rc;
    const bool bits = m->n_cand <= kTrackMaxCandBits && nq <= kTrackMaxQueryBits;
    const size_t off_slot = m->frame_end;
    const size_t off_skip = align256(off_slot + sizeof(int32_t) * (size_t)nq);
    const size_t end = align256(off_skip + (size_t)nq);
    if ((rc = m->h_in.ensure_keep(end + 256, m->frame_end))) return rc;
    uint8_t* h = (uint8_t*)m->h_in.p;
    if (G.slots) memcpy(h + off_slot, G.slots, sizeof(int32_t) * (size_t)nq);
    m->off_slot = off_slot;
    m->off_skip = off_skip;
    m->track_end = end;
    const size_t keys_bytes = align256(sizeof(uint32_t) * (size_t)nq * K);
    if ((rc = m->h_out.ensure(keys_bytes + sizeof(int32_t) * (size_t)nq))) return rc;
    hipStream_t s = m->stream;
    const auto t0 = std::chrono::steady_clock::now();
    size_t staged = 0;
    T.slot_base = G.slot_base;
    T.keys_soa = 1;
    T.use_bits = bits ? 1 : 0;
    if (!bits && (m->group_recording || m->link_recording)) return SO_RETRY_ON_HOST;  // (gates too large for the argument block: the plain calls run it)
    if (bits) {
        set_bits_from_excluded(m, G.excluded, nullptr, T);
        memset(T.skip_bits, 0, sizeof(uint32_t) * (size_t)((nq + 31) / 32));
        if (G.skip)
            for (int i = 0; i < nq; i++) {
                if (G.skip[i]) {
                    T.skip_bits[i >> 5] |= 1u << (i & 31);
                } else {  // few queries are skipped: eight clear bytes at a time
                    uint64_t w8;
                    while (i + 8 < nq && (memcpy(&w8, G.skip + i + 1, 8), w8 == 0)) i += 8;
                }
            }
        T.slot = G.slots ? reinterpret_cast<const int32_t*>(h + off_slot) : nullptr;  // pinned: the kernel reads it in place
        T.skip = nullptr;
        m->has_limit = false;
    } else {
        if (G.skip) memcpy(h + off_skip, G.skip, (size_t)nq);
        if (m->d_in.cap < end) {
            if ((rc = m->d_in.ensure(m->h_in.cap))) return rc;
            m->dirty_from = 0;
        }
        // one contiguous span covers whatever is new: [limit gate (if rewritten) | slots | skip]
        size_t from = SIZE_MAX, to = 0;
        if (m->has_limit && m->dirty_from < m->frame_end) { from = m->dirty_from; to = m->frame_end; }
        if (G.slots) { from = std::min(from, off_slot); to = off_slot + sizeof(int32_t) * (size_t)nq; }
        if (G.skip) { from = std::min(from, off_skip); to = off_skip + (size_t)nq; }
        if (to > from) {
            const size_t f16 = from & ~(size_t)15;
            launch_stage_in((uint8_t*)m->d_in.p + f16, h + f16, to - f16, s);
            staged = to - from;
        }
        m->dirty_from = SIZE_MAX;
        const uint8_t* d = (const uint8_t*)m->d_in.p;
        T.slot = G.slots ? reinterpret_cast<const int32_t*>(d + off_slot) : nullptr;
        T.skip = G.skip ? d + off_skip : nullptr;
    }
    if (m->group_recording || m->link_recording) {  // a member of a so_track_group inside a stage submit (or a linked stage): the search goes out with the group's / the link's
        so_track_group::Rec& r = group_rec(m);
        r.job.F = frame_dev(m);
        r.job.T = T;
        r.job.keys = m->keys_dev_override ? m->keys_dev_override : (uint32_t*)m->h_out.dev;
        r.job.count = (int32_t*)((uint8_t*)m->h_out.dev + keys_bytes);
        r.job.nq = nq;
        r.job.K = K;
        r.search = true;
        r.mode = mode;
    } else {
        if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
        launch_topk_track(frame_dev(m), T, mode, 0, nq, K, m->keys_dev_override ? m->keys_dev_override : (uint32_t*)m->h_out.dev,
                          (int32_t*)((uint8_t*)m->h_out.dev + keys_bytes), s);
        if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
        SO_HIP(hipGetLastError());
    }
    const auto t1 = std::chrono::steady_clock::now();
    m->stat[0] += std::chrono::duration<double, std::milli>(t1 - t0).count();
    m->stat[2] += 1.0;
    m->stat[3] += (double)staged;
    return SO_OK;
}

int finish_topk_track(so_matcher* m, int nq, int K) {
    const size_t keys_bytes = align256(sizeof(uint32_t) * (size_t)nq * K);
    const auto t1 = std::chrono::steady_clock::now();
    SO_HIP(hipStreamSynchronize(m->stream));
    const auto t2 = std::chrono::steady_clock::now();
    m->stat[1] += std::chrono::duration<double, std::milli>(t2 - t1).count();
    m->h_keys.p = m->h_out.p;
    m->h_count.p = (uint8_t*)m->h_out.p + keys_bytes;
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    return SO_OK;
}

// Exact top-K of ONE tracking query under a dynamic per-keypoint gate (see rerun_single): the query is rebuilt on
// the device from the same inputs, only the gate is re-sent.
int rerun_track(so_matcher* m, TrackQuerySrc T, int mode, int qi, const std::vector<int32_t>& limit_by_idx, int K,
                Entry* out, int* n_found) {
    int rc;
    if ((rc = m->h_rout.ensure(512))) return rc;
    hipStream_t s = m->stream;
    if (T.use_bits) {
        set_bits_from_excluded(m, nullptr, &limit_by_idx, T);
    } else {
        if ((rc = upload_limit_only(m, limit_by_idx))) return rc;
        const size_t f16 = m->dirty_from & ~(size_t)15;
        if (m->dirty_from < m->frame_end)
            launch_stage_in((uint8_t*)m->d_in.p + f16, (const uint8_t*)m->h_in.p + f16, m->frame_end - f16, s);
        m->dirty_from = SIZE_MAX;
    }
    T.in_view_out = nullptr;
    T.count8_out = nullptr;
    T.keys_soa = 0;
    if (m->profile) SO_HIP(hipEventRecord(m->e0, s));
    launch_topk_track(frame_dev(m), T, mode, qi, 1, K, (uint32_t*)m->h_rout.dev, (int32_t*)((uint8_t*)m->h_rout.dev + 256), s);
    if (m->profile) SO_HIP(hipEventRecord(m->e1, s));
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(s));
    m->stat[2] += 1.0;
    float ms = 0.f;
    if (m->profile && hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    const uint32_t* keys = (const uint32_t*)m->h_rout.p;
    *n_found = 0;
    for (int k = 0; k < K; k++) {
        if (keys[k] == 0xFFFFFFFFu) break;
        out[*n_found].dist = (int)(keys[k] >> 16);
        out[*n_found].idx = m->perm[(size_t)(keys[k] & 0xFFFFu)];
        (*n_found)++;
    }
    return SO_OK;
}

bool track_args_ok(const so_matcher* m, const so_dframe* cur, const so_map* map, const float* Tcw12) {
    if (!m || !cur || !cur->ready || !map || !Tcw12) return false;
    if (map->device != m->device || cur->device != m->device) {
        last_error_ref() = "matcher, frame and map live on different devices";
        return false;
    }
    return true;
}

}  // namespace

extern "C" {

// TrackWithMotionModel's search — ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono = true),
// code/src/ORBmatcher.cc:1223-1354, whole function (projection :1242-1276 included).  submit = everything up to the
// launch; wait = stream sync + the order-dependent resolve.
int so_track_search_last_frame_submit(so_matcher* m, const so_dframe* cur, const uint8_t* cur_excluded, const so_dframe* last,
                                      const so_map* map, const float* Tcw12, const int32_t* last_slot, float th) {
    if (m) (void)take_reuse(m);
    if (!track_args_ok(m, cur, map, Tcw12) || !last || !last->ready) return SO_ERR_INVALID_ARG;
    if (m->pend.mode != 0) {
        last_error_ref() = "a tracking search of this matcher has been submitted and not waited for";
        return SO_ERR_INVALID_ARG;
    }
    const int n_last = last->n;
    if (n_last > 0 && !last_slot) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    so_matcher::PendingTrack& P = m->pend;
    P = so_matcher::PendingTrack{};
    P.mode = 2;
    P.cur = cur;
    P.last = last;
    P.cur_excluded = cur_excluded;
    P.nq = n_last;
    if (n_last == 0 || cur->n == 0) return SO_OK;
    constexpr int K = 8;
    int rc = use_dframe(m, cur, cur_excluded);
    if (rc) { P.mode = 0; return rc; }
    const_cast<so_map*>(map)->grow_mu.lock_shared();
    P.held_map = map;
    struct Release {  // (error paths below: nothing stays in flight)
        so_matcher::PendingTrack& P;
        ~Release() {
            if (P.mode == 0 && P.held_map) {
                const_cast<so_map*>(P.held_map)->grow_mu.unlock_shared();
                P.held_map = nullptr;
            }
        }
    } release{P};
    P.T = track_src(m, cur, map, Tcw12, th);
    P.T.last_octave = last->d_octave;
    const TrackGates G{last_slot, 0, nullptr, cur_excluded};
    // per-query candidate counts as one byte each behind the K-lists: the only plane the resolve reads for every query
    const size_t keys_bytes2 = align256(sizeof(uint32_t) * (size_t)n_last * K);
    P.c8_off = align256(keys_bytes2 + sizeof(int32_t) * (size_t)n_last);
    if ((rc = m->h_out.ensure(P.c8_off + (size_t)n_last))) { P.mode = 0; return rc; }
    P.T.count8_out = m->cnt8_dev_override ? m->cnt8_dev_override : (uint8_t*)m->h_out.dev + P.c8_off;
    P.T.slot_out = m->slot_out_override;
    if ((rc = launch_topk_track_async(m, P.T, 2, n_last, K, G))) { P.mode = 0; return rc; }
    P.empty = false;
    return SO_OK;
}

int so_track_search_last_frame_wait(so_matcher* m, const uint8_t* slot_has_obs, int check_orientation, int32_t* kp_to_last,
                                    int32_t* nmatches) {
    if (!m || !kp_to_last || !nmatches) return SO_ERR_INVALID_ARG;
    if (m->pend.mode != 2) {
        last_error_ref() = "so_track_search_last_frame_wait without a submitted search";
        return SO_ERR_INVALID_ARG;
    }
    so_matcher::PendingTrack P = m->pend;
    m->pend.mode = 0;
    m->pend.held_map = nullptr;
    struct Unhold {
        const so_map* mp;
        ~Unhold() { if (mp) const_cast<so_map*>(mp)->grow_mu.unlock_shared(); }
    } unhold{P.held_map};
    const so_dframe* cur = P.cur;
    const so_dframe* last = P.last;
    const int n_last = P.nq;
    if (!cur->mirrors || !last->mirrors) {  // the resolve reads the frames' host copies (octave, angle)
        (void)hipStreamSynchronize(m->stream);
        last_error_ref() = "so_track_search_last_frame_wait: collect both frames (so_dframe_collect) before waiting for the search";
        return SO_ERR_INVALID_ARG;
    }
    *nmatches = 0;
    for (int k = 0; k < cur->n; k++) kp_to_last[k] = -1;
    if (P.empty) return SO_OK;
    constexpr int K = 8;
    SO_HIP(hipSetDevice(m->device));
    int rc = finish_topk_track(m, n_last, K);
    if (rc) return rc;
    const uint8_t* cur_excluded = P.cur_excluded;
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;
    const uint8_t* cnt = (const uint8_t*)m->h_out.p + P.c8_off;
    auto has_obs = [&](int i) { return !slot_has_obs || slot_has_obs[i]; };
    std::vector<int32_t> gate;
    std::vector<int>&rot_item = m->scratch_rot_item, &rot_b = m->scratch_rot_b;  // (capacity kept from call to call)
    rot_item.clear();
    rot_b.clear();
    int hist[HISTO_LENGTH] = {0};
    int nm = 0;
    for (int i = 0; i < n_last; i++) {
        if (cnt[(size_t)i] == 0) continue;
        Entry e[1];
        int found = 0, walked = 0;
        for (; walked < K && found < 1; walked++) {
            const uint32_t key = keys[(size_t)walked * n_last + i];  // [k][query]
            if (key == 0xFFFFFFFFu) break;
            const int idx = m->perm[(size_t)(key & 0xFFFFu)];
            if (kp_to_last[idx] >= 0 && has_obs(kp_to_last[idx])) continue;
            e[0].idx = idx;
            e[0].dist = (int)(key >> 16);
            found++;
        }
        if (found < 1 && walked == K && cnt[(size_t)i] > K) {
            gate.assign((size_t)cur->n, INT_MAX);
            for (int k = 0; k < cur->n; k++)
                if ((cur_excluded && cur_excluded[k]) || (kp_to_last[k] >= 0 && has_obs(kp_to_last[k]))) gate[(size_t)k] = 0;
            if ((rc = rerun_track(m, P.T, 2, i, gate, 1, e, &found))) return rc;
        }
        if (found == 0) continue;
        if (e[0].dist <= TH_HIGH) {
            kp_to_last[e[0].idx] = i;
            nm++;
            if (check_orientation) {
                const int b = rot_bin(last->angle[(size_t)i], cur->angle[(size_t)e[0].idx]);
                rot_item.push_back(e[0].idx);
                rot_b.push_back(b);
                hist[b]++;
            }
        }
    }
    if (check_orientation) {
        int i1, i2, i3;
        three_maxima(hist, HISTO_LENGTH, i1, i2, i3);
        for (size_t j = 0; j < rot_item.size(); j++)
            if (rot_b[j] != i1 && rot_b[j] != i2 && rot_b[j] != i3) {
                kp_to_last[rot_item[j]] = -1;
                nm--;
            }
    }
    *nmatches = nm;
    return SO_OK;
}

int so_track_search_last_frame(so_matcher* m, const so_dframe* cur, const uint8_t* cur_excluded, const so_dframe* last,
                               const so_map* map, const float* Tcw12, const int32_t* last_slot,
                               const uint8_t* slot_has_obs, float th, int check_orientation, int32_t* kp_to_last,
                               int32_t* nmatches) {
    if (!kp_to_last || !nmatches) return SO_ERR_INVALID_ARG;
    const int rc = so_track_search_last_frame_submit(m, cur, cur_excluded, last, map, Tcw12, last_slot, th);
    if (rc) return rc;
    return so_track_search_last_frame_wait(m, slot_has_obs, check_orientation, kp_to_last, nmatches);
}

// TrackLocalMap's search — Tracking::SearchLocalPoints (code/src/Tracking.cc:964-1007): Frame::isInFrustum(pMP,
// cos_limit) (code/src/Frame.cc:316-375) for every local map point that is not already matched in this frame, then
// ORBmatcher::SearchByProjection(F, vpMapPoints, th) (code/src/ORBmatcher.cc:44-121)
int so_track_search_local_map_submit(so_matcher* m, const so_dframe* cur, const uint8_t* cur_excluded, const so_map* map,
                                     const float* Tcw12, int32_t n_local, const int32_t* local_slot, int32_t first_slot,
                                     const uint8_t* skip, float th, float nn_ratio, float viewing_cos_limit,
                                     float log_scale_factor) {
    if (m) (void)take_reuse(m);
    if (!track_args_ok(m, cur, map, Tcw12) || n_local < 0) return SO_ERR_INVALID_ARG;
    if (m->pend.mode != 0) {
        last_error_ref() = "a tracking search of this matcher has been submitted and not waited for";
        return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(m->device));
    m->last_ms = 0.f;
    m->stat[0] = m->stat[1] = m->stat[2] = m->stat[3] = 0.0;
    so_matcher::PendingTrack& P = m->pend;
    P = so_matcher::PendingTrack{};
    P.mode = 3;
    P.cur = cur;
    P.cur_excluded = cur_excluded;
    P.nq = n_local;
    P.nn_ratio = nn_ratio;
    if (n_local == 0) return SO_OK;
    constexpr int K = 8;
    int rc = use_dframe(m, cur, cur_excluded);
    if (rc) { P.mode = 0; return rc; }
    const_cast<so_map*>(map)->grow_mu.lock_shared();
    P.held_map = map;
    struct Release {
        so_matcher::PendingTrack& P;
        ~Release() {
            if (P.mode == 0 && P.held_map) {
                const_cast<so_map*>(P.held_map)->grow_mu.unlock_shared();
                P.held_map = nullptr;
            }
        }
    } release{P};
    P.T = track_src(m, cur, map, Tcw12, th);
    P.T.cos_limit = viewing_cos_limit;
    P.T.log_scale_factor = log_scale_factor;
    {   // smallest integer d with nn_ratio * d >= TH_HIGH in the float arithmetic of the ratio test (ORBmatcher.cc:112)
        int d = TH_HIGH;
        while (d < 256 && (float)TH_HIGH > nn_ratio * (float)d) d++;
        P.T.second_best_bound = d;
    }
    // mbTrackInView of every query and the one-byte candidate counts come back through host-mapped memory behind the K-lists
    const size_t keys_bytes = align256(sizeof(uint32_t) * (size_t)n_local * K);
    P.view_off = align256(keys_bytes + sizeof(int32_t) * (size_t)n_local);
    P.c8_off = align256(P.view_off + (size_t)n_local);
    if ((rc = m->h_out.ensure(P.c8_off + (size_t)n_local))) { P.mode = 0; return rc; }
    P.T.in_view_out = (uint8_t*)m->h_out.dev + P.view_off;
    P.T.count8_out = m->cnt8_dev_override ? m->cnt8_dev_override : (uint8_t*)m->h_out.dev + P.c8_off;
    P.T.slot_out = m->slot_out_override;
    const TrackGates G{local_slot, local_slot ? 0 : first_slot, skip, cur_excluded};
    if ((rc = launch_topk_track_async(m, P.T, 3, n_local, K, G))) { P.mode = 0; return rc; }
    P.empty = false;
    return SO_OK;
}

int so_track_search_local_map_wait(so_matcher* m, const uint8_t* slot_has_obs, uint8_t* in_view, int32_t* kp_to_local,
                                   int32_t* nmatches) {
    if (!m || !kp_to_local || !nmatches) return SO_ERR_INVALID_ARG;
    if (m->pend.mode != 3) {
        last_error_ref() = "so_track_search_local_map_wait without a submitted search";
        return SO_ERR_INVALID_ARG;
    }
    so_matcher::PendingTrack P = m->pend;
    m->pend.mode = 0;
    m->pend.held_map = nullptr;
    struct Unhold {
        const so_map* mp;
        ~Unhold() { if (mp) const_cast<so_map*>(mp)->grow_mu.unlock_shared(); }
    } unhold{P.held_map};
    const so_dframe* cur = P.cur;
    const int n_local = P.nq;
    if (!cur->mirrors) {
        (void)hipStreamSynchronize(m->stream);
        last_error_ref() = "so_track_search_local_map_wait: collect the frame (so_dframe_collect) before waiting for the search";
        return SO_ERR_INVALID_ARG;
    }
    *nmatches = 0;
    for (int k = 0; k < cur->n; k++) kp_to_local[k] = -1;
    if (in_view && n_local > 0) memset(in_view, 0, (size_t)n_local);
    if (P.empty) return SO_OK;
    constexpr int K = 8;
    SO_HIP(hipSetDevice(m->device));
    int rc = finish_topk_track(m, n_local, K);
    if (rc) return rc;
    const uint8_t* cur_excluded = P.cur_excluded;
    const float nn_ratio = P.nn_ratio;
    const uint8_t* view = (const uint8_t*)m->h_out.p + P.view_off;
    if (in_view) memcpy(in_view, view, (size_t)n_local);
    if (cur->n == 0) return SO_OK;
    const uint32_t* keys = (const uint32_t*)m->h_keys.p;
    const uint8_t* cnt = (const uint8_t*)m->h_out.p + P.c8_off;  // 0 for a point that is not in view
    auto has_obs = [&](int i) { return !slot_has_obs || slot_has_obs[i]; };
    std::vector<int32_t> gate;
    int nm = 0;
    for (int i = 0; i < n_local; i++) {
        if (cnt[(size_t)i] == 0) {
            // most of a local map is out of view: skip eight empty counts at a time
            uint64_t w8;
            while (i + 8 < n_local && (memcpy(&w8, cnt + i + 1, 8), w8 == 0)) i += 8;
            continue;
        }
        Entry e[2];
        int found = 0, walked = 0;
        for (; walked < K && found < 2; walked++) {
            const uint32_t key = keys[(size_t)walked * n_local + i];  // [k][query]
            if (key == 0xFFFFFFFFu) break;
            const int idx = m->perm[(size_t)(key & 0xFFFFu)];
            if (kp_to_local[idx] >= 0 && has_obs(kp_to_local[idx])) continue;  // ORBmatcher.cc:83-85
            e[found].idx = idx;
            e[found].dist = (int)(key >> 16);
            found++;
        }
        if (found < 2 && walked == K && cnt[(size_t)i] > K) {
            gate.assign((size_t)cur->n, INT_MAX);
            for (int k = 0; k < cur->n; k++)
                if ((cur_excluded && cur_excluded[k]) || (kp_to_local[k] >= 0 && has_obs(kp_to_local[k]))) gate[(size_t)k] = 0;
            if ((rc = rerun_track(m, P.T, 3, i, gate, 2, e, &found))) return rc;
        }
        if (found == 0) continue;
        const int bestDist = e[0].dist, bestIdx = e[0].idx, bestLevel = cur->octave[(size_t)bestIdx];
        const int bestDist2 = found > 1 ? e[1].dist : 256;
        const int bestLevel2 = found > 1 ? cur->octave[(size_t)e[1].idx] : -1;
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
            kp_to_local[bestIdx] = i;
            nm++;
        }
    }
    *nmatches = nm;
    return SO_OK;
}

int so_track_search_local_map(so_matcher* m, const so_dframe* cur, const uint8_t* cur_excluded, const so_map* map,
                              const float* Tcw12, int32_t n_local, const int32_t* local_slot, int32_t first_slot,
                              const uint8_t* skip, const uint8_t* slot_has_obs, float th, float nn_ratio,
                              float viewing_cos_limit, float log_scale_factor, uint8_t* in_view, int32_t* kp_to_local,
                              int32_t* nmatches) {
    if (!kp_to_local || !nmatches) return SO_ERR_INVALID_ARG;
    const int rc = so_track_search_local_map_submit(m, cur, cur_excluded, map, Tcw12, n_local, local_slot, first_slot, skip,
                                                    th, nn_ratio, viewing_cos_limit, log_scale_factor);
    if (rc) return rc;
    return so_track_search_local_map_wait(m, slot_has_obs, in_view, kp_to_local, nmatches);
}

}  // extern "C"

// =====================================================================================================
// A tracking stage without a host hop (include/swarmorb.h: so_track_stage_*): the plain search's launch, then the
// order-dependent resolve on the device (match_kernels.hip: track_resolve_kernel) and the PoseOptimization kernel over
// the edges it lists (ba_kernels.hip: pose_opt_chain_kernel), all on the matcher's stream; the host waits for the pose's
// completion word.
// =====================================================================================================
namespace {

struct ChainOffsets {
    size_t d_keys, d_cnt8, d_qslot, d_ekp, d_eslot, d_head, d_total;
    size_t h_k2q, h_ekp, h_head, h_pose, h_info, h_outl, h_slot_in, h_total;
};

ChainOffsets chain_offsets(int nq, int K, int nk) {
    ChainOffsets o{};
    size_t d = 0;
    o.d_keys = d; d += align256(sizeof(uint32_t) * (size_t)nq * (size_t)K);
    o.d_cnt8 = d; d += align256((size_t)nq);
    o.d_qslot = d; d += align256(sizeof(int32_t) * (size_t)nq);
    o.d_ekp = d; d += align256(sizeof(int32_t) * (size_t)nk);
    o.d_eslot = d; d += align256(sizeof(int32_t) * (size_t)nk);
    o.d_head = d; d += 256;
    o.d_total = d;
    size_t h = 0;
    o.h_k2q = h; h += align256(sizeof(int32_t) * (size_t)nk);
    o.h_ekp = h; h += align256(sizeof(int32_t) * (size_t)nk);
    o.h_head = h; h += 256;
    o.h_pose = h; h += 256;
    o.h_info = h; h += 256;
    o.h_outl = h; h += align256((size_t)nk + 16);
    o.h_slot_in = h; h += align256(sizeof(int32_t) * (size_t)nk);
    o.h_total = h;
    return o;
}

// resolve + pose behind a search that has just been launched (m->pend holds it)
int chain_launch(so_matcher* m, int kind, const ChainOffsets& O, const so_dframe* cur, const so_dframe* last, const so_map* map,
                 const float* Tcw12, int check_orientation, const int32_t* kp_slot_in_dev, const float* intr4,
                 const float* level_inv_sigma2) {
    so_matcher::PendingTrack& P = m->pend;
    uint8_t* d = (uint8_t*)m->d_chain.p;
    uint8_t* hd = (uint8_t*)m->h_chain.dev;
    hipStream_t s = m->stream;
    so::TrackResolveArgs R{};
    R.keys = (const uint32_t*)(d + O.d_keys);
    R.cnt8 = d + O.d_cnt8;
    R.nq = P.nq;
    R.K = 8;
    R.mode = P.mode;
    R.nn_ratio = P.nn_ratio;
    R.n_cand = m->n_cand;
    R.n_kp = cur->n;
    R.s_octave = cur->d_s_octave;
    R.cell_items = cur->d_cell_items;
    R.check_orientation = check_orientation;
    R.q_angle = last ? last->d_angle : nullptr;
    R.cur_angle = cur->d_angle;
    R.q_slot = (const int32_t*)(d + O.d_qslot);  // written by the search (TrackQuerySrc::slot_out)
    R.slot_base = 0;
    R.kp_slot_in = kp_slot_in_dev;
    R.kp_slot_out = (int32_t*)m->d_kpslot.p;
    R.kp_to_q = (int32_t*)(hd + O.h_k2q);
    R.e_kp = (int32_t*)(d + O.d_ekp);
    R.e_slot = (int32_t*)(d + O.d_eslot);
    R.e_kp_host = (int32_t*)(hd + O.h_ekp);
    R.head = (int32_t*)(d + O.d_head);
    R.head_host = (int32_t*)(hd + O.h_head);
    if (m->group_recording || m->link_recording) {
        so_track_group::Rec& r = group_rec(m);
        r.res = R;
        r.resolve = true;
    } else {
        so::launch_track_resolve(R, s);
    }
    so::PoseOptArgs& A = m->chain_pose;
    A = so::PoseOptArgs{};
    A.Xw = nullptr; A.obs = nullptr; A.inv_sigma2 = nullptr;
    for (int k = 0; k < 4; k++) A.K[k] = (double)intr4[k];
    so::pose_from_Tcw12(Tcw12, A.init);
    A.n = 0;
    A.err = nullptr;
    A.outlier = hd + O.h_outl;
    A.pose_out = reinterpret_cast<so::BaPose*>(hd + O.h_pose);
    A.info = reinterpret_cast<int*>(hd + O.h_info);
    A.trace = nullptr;
    A.e_kp = R.e_kp;
    A.e_slot = R.e_slot;
    A.head = R.head;
    A.map_Xw = map->d_Xw;
    A.kp_xy_un = cur->d_xy_un;
    A.kp_octave = cur->d_octave;
    for (int l = 0; l < 8; l++) A.lvl_inv_sigma2[l] = l < cur->nlevels ? level_inv_sigma2[l] : 0.f;
    // TrackWithMotionModel: the outliers of its pose lose their map point (Tracking.cc:745-760) - also in the bindings
    // the next stage reads from the device
    A.kp_slot_clear = kind == 0 ? (int32_t*)m->d_kpslot.p : nullptr;
    so_matcher::ChainPending& C = m->chain;
    C = so_matcher::ChainPending{};
    C.active = true;
    C.kind = kind;
    C.n_kp = cur->n;
    C.nq = P.nq;
    C.h_k2q = O.h_k2q; C.h_ekp = O.h_ekp; C.h_head = O.h_head; C.h_pose = O.h_pose; C.h_info = O.h_info; C.h_outl = O.h_outl;
    C.h_slot_in = O.h_slot_in;
    memcpy(C.Tcw_in, Tcw12, 48);
    C.map = map;
    C.frame = cur;
    m->kpslot_frame = nullptr;  // (valid again when this stage has come back complete)
    if (++m->chain_seq == 0) m->chain_seq = 1;
    C.seq = m->chain_seq;
    A.done_seq = C.seq;
    reinterpret_cast<volatile int*>((uint8_t*)m->h_chain.p + O.h_info)[3] = 0;
    memset((uint8_t*)m->h_chain.p + O.h_head, 0, 64);
    std::atomic_thread_fence(std::memory_order_release);
    C.grouped = m->group_recording || m->link_recording;
    C.events = m->profile && !C.grouped;
    if (C.grouped) {
        so_track_group::Rec& r = group_rec(m);
        r.pose_args = A;
        r.pose = true;
        r.range = m->chain_range[kind];
        return SO_OK;
    }
    if (C.events) {
        if (!m->pe0) SO_HIP(hipEventCreate(&m->pe0));
        if (!m->pe1) SO_HIP(hipEventCreate(&m->pe1));
        SO_HIP(hipEventRecord(m->pe0, s));
    }
    so::launch_pose_opt_chain(A, m->chain_range[kind], s);
    if (C.events) SO_HIP(hipEventRecord(m->pe1, s));
    SO_HIP(hipGetLastError());
    return SO_OK;
}

}  // namespace

static int chain_reserve(so_matcher* m, int nq) {
    if (m->chain.active) return SO_OK;
    const ChainOffsets O = chain_offsets(nq, 8, (int)so::kResolveMaxCand);
    int rc;
    if ((rc = m->d_chain.ensure(O.d_total)) || (rc = m->h_chain.ensure(O.h_total))) return rc;
    const void* before = m->d_kpslot.p;
    if ((rc = m->d_kpslot.ensure(sizeof(int32_t) * (size_t)so::kResolveMaxCand))) return rc;
    if (m->d_kpslot.p != before) m->kpslot_frame = nullptr;
    return SO_OK;
}

namespace {

int chain_prepare(so_matcher* m, int nq, int nk, ChainOffsets* O) {
    *O = chain_offsets(nq, 8, nk);
    int rc;
    if ((rc = m->d_chain.ensure(O->d_total))) return rc;
    if ((rc = m->h_chain.ensure(O->h_total))) return rc;
    const void* kpslot_before = m->d_kpslot.p;
    if ((rc = m->d_kpslot.ensure(sizeof(int32_t) * (size_t)std::max(nk, (int)so::kResolveMaxCand)))) return rc;
    if (m->d_kpslot.p != kpslot_before) m->kpslot_frame = nullptr;  // (a new block: the last stage's bindings are gone)
    m->keys_dev_override = (uint32_t*)((uint8_t*)m->d_chain.p + O->d_keys);
    m->cnt8_dev_override = (uint8_t*)m->d_chain.p + O->d_cnt8;
    m->slot_out_override = (int32_t*)((uint8_t*)m->d_chain.p + O->d_qslot);
    return SO_OK;
}

// Scope of a stage submit by a member of a so_track_group: launches are recorded while it lives; drop() takes the member's
// row out again when the stage is handed back (the caller then runs the plain calls, which launch at once).
struct GroupRecording {
    so_matcher* m;
    explicit GroupRecording(so_matcher* m_) : m(m_) { m->group_recording = m->group != nullptr; }
    ~GroupRecording() { m->group_recording = false; }
    int drop(int rc) {
        if (m->group) {
            so_track_group* g = m->group;
            for (int i = 0; i < g->n_recs; i++)
                if (g->recs[(size_t)i].m == m) {
                    for (int j = i; j + 1 < g->n_recs; j++) std::swap(g->recs[(size_t)j], g->recs[(size_t)j + 1]);
                    g->n_recs--;
                    break;
                }
        }
        m->group_recording = false;
        return rc;
    }
};

void chain_drop_search(so_matcher* m) {  // a submitted search that will not be waited for by so_track_search_*_wait
    if (m->pend.held_map) const_cast<so_map*>(m->pend.held_map)->grow_mu.unlock_shared();
    m->pend.held_map = nullptr;
    m->pend.mode = 0;
}

}  // namespace

extern "C" {

int so_track_stage_last_frame_submit(so_matcher* m, const so_dframe* cur, const so_dframe* last, const so_map* map,
                                     const float* Tcw12, const int32_t* last_slot, float th, int check_orientation,
                                     const float* intr4, const float* level_inv_sigma2) {
    if (!m || !cur || !last || !map || !Tcw12 || !intr4 || !level_inv_sigma2 || m->chain.active) return SO_ERR_INVALID_ARG;
    m->chain.valid_edges = false;  // (whatever happens below: the edge list of an earlier stage is not this frame's)
    if (!cur->ready || !last->ready || last->n <= 0 || cur->n <= 0) return SO_RETRY_ON_HOST;  // nothing to chain: the plain calls handle it
    SO_HIP(hipSetDevice(m->device));
    ChainOffsets O;
    int rc = chain_prepare(m, last->n, cur->n, &O);
    GroupRecording recording(m);  // (a member of a so_track_group: what follows records its launches instead of issuing them)
    if (rc == SO_OK) rc = so_track_search_last_frame_submit(m, cur, nullptr, last, map, Tcw12, last_slot, th);
    m->keys_dev_override = nullptr;
    m->cnt8_dev_override = nullptr;  m->slot_out_override = nullptr;
    if (rc != SO_OK) return recording.drop(rc);
    if (m->pend.empty) {  // nothing was launched
        chain_drop_search(m);
        return recording.drop(SO_RETRY_ON_HOST);
    }
    rc = chain_launch(m, 0, O, cur, last, map, Tcw12, check_orientation, nullptr, intr4, level_inv_sigma2);
    if (rc != SO_OK) {
        recording.drop(rc);
        (void)hipStreamSynchronize(m->stream);
        chain_drop_search(m);
        m->chain.active = false;
    }
    return rc;
}

}  // extern "C"

// kp_slot_dev != null (linked stage): the bindings on entry are read from THAT device array (another matcher's d_kpslot), the
// host array kp_slot only says "nothing is known here" (all -1: the excluded set is formed on the device)
static int stage_local_map_submit(so_matcher* m, const so_dframe* cur, const int32_t* kp_slot, int kp_slot_is_last_stage,
                                  const int32_t* kp_slot_dev, const so_map* map, const float* Tcw12, int32_t n_local,
                                  const int32_t* local_slot, int32_t first_slot, const uint8_t* skip, float th, float nn_ratio,
                                  float viewing_cos_limit, float log_scale_factor, const float* intr4, const float* level_inv_sigma2) {
    if (!m || !cur || !kp_slot || !map || !Tcw12 || !intr4 || !level_inv_sigma2 || n_local < 0 || m->chain.active)
        return SO_ERR_INVALID_ARG;
    m->chain.valid_edges = false;
    if (!cur->ready || n_local <= 0 || cur->n <= 0) return SO_RETRY_ON_HOST;
    SO_HIP(hipSetDevice(m->device));
    ChainOffsets O;
    int rc = chain_prepare(m, n_local, cur->n, &O);
    if (rc != SO_OK) {
        m->keys_dev_override = nullptr;
        m->cnt8_dev_override = nullptr;  m->slot_out_override = nullptr;
        return rc;
    }
    // the bindings on entry: the search's excluded set, and edges of the pose problem.  The resolve kernel reads them from
    // the host-mapped copy - or, when they are what the last stage left on the device (its matches, minus its pose's
    // outliers), from there: no read across PCIe inside the chain
    const bool on_device = kp_slot_dev != nullptr || (kp_slot_is_last_stage && m->kpslot_frame == cur && m->kpslot_generation == cur->generation);
    int32_t* slot_in = (int32_t*)((uint8_t*)m->h_chain.p + O.h_slot_in);
    if (!on_device) memcpy(slot_in, kp_slot, sizeof(int32_t) * (size_t)cur->n);
    static thread_local std::vector<uint8_t> excluded;
    excluded.resize((size_t)cur->n);
    for (int k = 0; k < cur->n; k++) excluded[(size_t)k] = kp_slot[k] >= 0 ? 1 : 0;
    GroupRecording recording(m);
    rc = so_track_search_local_map_submit(m, cur, excluded.data(), map, Tcw12, n_local, local_slot, first_slot, skip, th, nn_ratio,
                                          viewing_cos_limit, log_scale_factor);
    m->keys_dev_override = nullptr;
    m->cnt8_dev_override = nullptr;  m->slot_out_override = nullptr;
    if (rc != SO_OK) return recording.drop(rc);
    if (m->pend.empty) {
        chain_drop_search(m);
        return recording.drop(SO_RETRY_ON_HOST);
    }
    rc = chain_launch(m, 1, O, cur, nullptr, map, Tcw12, 0,
                      kp_slot_dev ? kp_slot_dev : on_device ? (const int32_t*)m->d_kpslot.p : (const int32_t*)((uint8_t*)m->h_chain.dev + O.h_slot_in), intr4,
                      level_inv_sigma2);
    if (rc != SO_OK) {
        recording.drop(rc);
        (void)hipStreamSynchronize(m->stream);
        chain_drop_search(m);
        m->chain.active = false;
    }
    return rc;
}

extern "C" {

int so_track_stage_local_map_submit(so_matcher* m, const so_dframe* cur, const int32_t* kp_slot, int kp_slot_is_last_stage,
                                    const so_map* map, const float* Tcw12, int32_t n_local, const int32_t* local_slot,
                                    int32_t first_slot, const uint8_t* skip, float th, float nn_ratio, float viewing_cos_limit,
                                    float log_scale_factor, const float* intr4, const float* level_inv_sigma2) {
    return stage_local_map_submit(m, cur, kp_slot, kp_slot_is_last_stage, nullptr, map, Tcw12, n_local, local_slot, first_slot, skip, th, nn_ratio,
                                  viewing_cos_limit, log_scale_factor, intr4, level_inv_sigma2);
}

// TrackLocalMap's stage enqueued BEHIND TrackWithMotionModel's (`first`: another matcher of the same stream whose last-frame
// stage for `cur` is in flight) without waiting for it: include/swarmorb.h.
int so_track_stage_local_map_submit_after(so_matcher* m, so_matcher* first, const so_dframe* cur, const so_map* map, int32_t n_local,
                                          const int32_t* local_slot, int32_t first_slot, const uint8_t* skip_static, float th, float nn_ratio,
                                          float viewing_cos_limit, float log_scale_factor, const float* intr4, const float* level_inv_sigma2) {
    if (!m || !first || m == first || !cur || !map || !intr4 || !level_inv_sigma2 || m->chain.active || m->group || first->group)
        return SO_ERR_INVALID_ARG;
    if (!first->chain.active || first->chain.kind != 0 || first->chain.frame != cur || first->stream != m->stream || first->device != m->device) {
        last_error_ref() = "so_track_stage_local_map_submit_after: `first` must have the last-frame stage of this frame in flight on the same stream";
        return SO_ERR_INVALID_ARG;
    }
    if (cur->n > so::kResolveMaxCand) return SO_RETRY_ON_HOST;
    static thread_local std::vector<int32_t> none;
    none.assign((size_t)cur->n, -1);
    if (!m->link_rec) m->link_rec = new LinkRec();
    so_track_group::Rec& R = *m->link_rec;
    R.search = R.resolve = R.pose = false;
    R.m = m;
    m->link_recording = true;
    const int rc = stage_local_map_submit(m, cur, none.data(), 0, (const int32_t*)first->d_kpslot.p, map, first->chain.Tcw_in, n_local, local_slot,
                                          first_slot, skip_static, th, nn_ratio, viewing_cos_limit, log_scale_factor, intr4, level_inv_sigma2);
    m->link_recording = false;
    if (rc != SO_OK) return rc;
    if (!R.search || !R.resolve || !R.pose) {  // (cannot happen: SO_OK means the three launches were recorded)
        (void)hipStreamSynchronize(m->stream);
        chain_drop_search(m);
        m->chain.active = false;
        return SO_ERR_HIP;
    }
    // the three one-row tables -> HBM; the link kernel patches pose, gates and start pose in place; the grouped kernels read them
    const size_t off_pose = 0, off_res = align256(sizeof(so::PoseOptArgs)), off_job = align256(off_res + sizeof(so::TrackResolveArgs));
    const size_t total = align256(off_job + sizeof(so::TrackGroupJob));
    int rc2;
    if ((rc2 = m->h_link.ensure(total)) || (rc2 = m->d_link.ensure(total))) {
        (void)hipStreamSynchronize(m->stream);
        chain_drop_search(m);
        m->chain.active = false;
        return rc2;
    }
    uint8_t* h = (uint8_t*)m->h_link.p;
    uint8_t* d = (uint8_t*)m->d_link.p;
    memcpy(h + off_pose, &R.pose_args, sizeof(so::PoseOptArgs));
    memcpy(h + off_res, &R.res, sizeof(so::TrackResolveArgs));
    memcpy(h + off_job, &R.job, sizeof(so::TrackGroupJob));
    std::atomic_thread_fence(std::memory_order_release);
    hipStream_t s = m->stream;
    launch_stage_in(d, h, total, s);
    so::TrackLinkArgs L{};
    L.pose1 = reinterpret_cast<const double*>(first->chain_pose.pose_out);
    L.kp_slot = (const int32_t*)first->d_kpslot.p;
    L.cell_items = cur->d_cell_items;
    L.n_cand = m->n_cand;
    L.n_kp = cur->n;
    L.local_slot = R.job.T.slot;
    L.first_slot = R.job.T.slot_base;
    L.n_local = n_local;
    L.job = reinterpret_cast<so::TrackGroupJob*>(d + off_job);
    L.pose2_init = reinterpret_cast<double*>(d + off_pose + offsetof(so::PoseOptArgs, init));
    so::launch_track_link(L, s);
    so::launch_topk_track_group(reinterpret_cast<const so::TrackGroupJob*>(d + off_job), 1, 3, R.job.nq, s);
    so::launch_track_resolve_group(reinterpret_cast<const so::TrackResolveArgs*>(d + off_res), 1, R.job.nq, s);
    so::launch_pose_opt_chain_group(reinterpret_cast<const so::PoseOptArgs*>(d + off_pose), 1, 1 << (R.range ? 1 : 0), s);
    SO_HIP(hipGetLastError());
    m->chain.linked = true;
    return SO_OK;
}

// The start pose of a linked stage as the host knows it once the first stage has been waited for (what so_track_stage_wait hands
// back when the stage has fewer than three edges: Optimizer.cc:358-359 leaves the frame's pose alone).
int so_track_stage_set_start_pose(so_matcher* m, const float* Tcw12) {
    if (!m || !Tcw12 || !m->chain.active) return SO_ERR_INVALID_ARG;
    memcpy(m->chain.Tcw_in, Tcw12, 48);
    return SO_OK;
}

int so_track_stage_pose_again_submit(so_matcher* m, const float* Tcw12) {
    if (!m || !Tcw12 || m->chain.active || !m->chain.valid_edges) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    so_matcher::ChainPending& C = m->chain;
    so::PoseOptArgs& A = m->chain_pose;
    so::pose_from_Tcw12(Tcw12, A.init);
    A.kp_slot_clear = nullptr;
    memcpy(C.Tcw_in, Tcw12, 48);
    if (++m->chain_seq == 0) m->chain_seq = 1;
    C.seq = m->chain_seq;
    A.done_seq = C.seq;
    reinterpret_cast<volatile int*>((uint8_t*)m->h_chain.p + C.h_info)[3] = 0;
    std::atomic_thread_fence(std::memory_order_release);
    const int range_kind = C.kind == 2 ? 1 : C.kind;  // (the edges are the last stage's)
    C.kind = 2;
    C.active = true;
    // the kernel reads the map table in place: no reallocation by another thread's append until the wait
    const_cast<so_map*>(C.map)->grow_mu.lock_shared();
    C.holds_map = true;
    A.map_Xw = C.map->d_Xw;
    C.grouped = m->group != nullptr;
    C.events = m->profile && !C.grouped;
    hipStream_t s = m->stream;
    if (C.grouped) {
        m->group_recording = true;
        so_track_group::Rec& r = group_rec(m);
        m->group_recording = false;
        r.pose_args = A;
        r.pose = true;
        r.range = m->chain_range[range_kind];
        return SO_OK;
    }
    if (C.events) {
        if (!m->pe0) SO_HIP(hipEventCreate(&m->pe0));
        if (!m->pe1) SO_HIP(hipEventCreate(&m->pe1));
        SO_HIP(hipEventRecord(m->pe0, s));
    }
    so::launch_pose_opt_chain(A, m->chain_range[range_kind], s);
    if (C.events) SO_HIP(hipEventRecord(m->pe1, s));
    SO_HIP(hipGetLastError());
    return SO_OK;
}

int so_track_stage_wait(so_matcher* m, int32_t* kp_to_q, int32_t* nmatches, uint8_t* in_view, int32_t* n_edges, int32_t* edge_kp,
                        uint8_t* edge_outlier, float* Tcw_out12, int32_t* n_inliers, int32_t* info2) {
    if (!m || !m->chain.active || !n_edges || !edge_kp || !edge_outlier || !Tcw_out12 || !n_inliers) return SO_ERR_INVALID_ARG;
    so_matcher::ChainPending& C = m->chain;
    const bool again = C.kind == 2;
    if (!again && (!kp_to_q || !nmatches)) return SO_ERR_INVALID_ARG;
    C.active = false;
    const uint8_t* h = (const uint8_t*)m->h_chain.p;
    const volatile int* done = reinterpret_cast<const volatile int*>(h + C.h_info) + 3;
    const auto t1 = std::chrono::steady_clock::now();
    int rc = SO_OK;
    for (unsigned long it = 1; *done != C.seq; it++) {
        if ((it & 0x1fff) == 0) {  // ~0.2 ms of polling: let the stream's completion wake us instead
            const hipError_t q = hipStreamSynchronize(m->stream);
            if (q != hipSuccess || *done != C.seq) {
                last_error_ref() = q != hipSuccess ? std::string("tracking stage: ") + hipGetErrorString(q)
                                                   : std::string("tracking stage: the pose kernel finished without publishing its results");
                rc = SO_ERR_HIP;
            }
            break;
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    m->stat[1] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    so_matcher::PendingTrack P = m->pend;
    if (!again) chain_drop_search(m);  // the search and everything behind it are done: the map may grow again
    if (C.holds_map) {
        const_cast<so_map*>(C.map)->grow_mu.unlock_shared();
        C.holds_map = false;
    }
    if (rc != SO_OK) {
        C.valid_edges = false;
        return rc;
    }
    if (!again && m->profile && !C.grouped) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->e0, m->e1) == hipSuccess) m->last_ms += ms;
    }
    int head[4], inf[4];
    memcpy(head, h + C.h_head, 16);
    memcpy(inf, h + C.h_info, 16);
    static const bool stage_debug = getenv("SWARMORB_STAGE_DEBUG") != nullptr;
    if (stage_debug && !again) {
        const int* hh = (const int*)(h + C.h_head);
        fprintf(stderr, "[stage %d] edges %d matches %d fallback %d rounds %d active %d | ticks (10 ns): loads %d rounds %d rotation %d by-keypoint %d out %d\n",
                C.kind, hh[0], hh[1], hh[2], hh[3], hh[4], hh[8], hh[9], hh[10], hh[11], hh[12]);
    }
    const int kind = again ? 1 : C.kind;
    if (!again) m->chain_range[kind] = head[0] > 1024 ? 1 : 0;  // what the next call of this stage launches
    *n_edges = 0;
    *n_inliers = 0;
    if (info2) info2[0] = info2[1] = 0;
    if (!again) {
        *nmatches = 0;
        if (in_view && P.mode == 3) memcpy(in_view, (const uint8_t*)m->h_out.p + P.view_off, (size_t)P.nq);
    }
    if (head[2] != 0 || inf[0] == -1) {  // the resolve gave up, or more edges than the launched variant holds
        if (!again)
            for (int k = 0; k < C.n_kp; k++) kp_to_q[k] = -1;
        C.valid_edges = false;
        return SO_RETRY_ON_HOST;
    }
    C.valid_edges = true;
    if (!again) {
        memcpy(kp_to_q, h + C.h_k2q, sizeof(int32_t) * (size_t)C.n_kp);
        *nmatches = head[1];
        m->kpslot_frame = C.frame;  // d_kpslot holds this frame's bindings behind this stage
        m->kpslot_generation = C.frame->generation;
    }
    const int ne = head[0];
    *n_edges = ne;
    memcpy(edge_kp, h + C.h_ekp, sizeof(int32_t) * (size_t)ne);
    if (inf[0] == -2) {  // fewer than three edges: PoseOptimization returns 0 and touches nothing (Optimizer.cc:358-359)
        memset(edge_outlier, 0, (size_t)ne);
        memcpy(Tcw_out12, C.Tcw_in, 48);
        return SO_OK;
    }
    memcpy(edge_outlier, h + C.h_outl, (size_t)ne);
    so::BaPose Pz;
    memcpy(&Pz, h + C.h_pose, sizeof(Pz));
    so::pose_to_Tcw12(Pz, Tcw_out12);
    *n_inliers = ne - inf[0];
    if (info2) {
        info2[0] = inf[1];
        info2[1] = inf[2];
    }
    return SO_OK;
}

int so_track_stage_invalidate(so_matcher* m) {
    if (!m) return SO_ERR_INVALID_ARG;
    m->kpslot_frame = nullptr;
    return SO_OK;
}

int so_track_stage_last_rounds(so_matcher* m, int32_t* rounds, int32_t* active_queries) {
    if (!m || !rounds || !m->h_chain.p || m->chain.active) return SO_ERR_INVALID_ARG;
    const int32_t* head = (const int32_t*)((const uint8_t*)m->h_chain.p + m->chain.h_head);
    *rounds = head[3];
    if (active_queries) *active_queries = head[4];
    if (getenv("SWARMORB_STAGE_DEBUG"))
        fprintf(stderr, "[stage] edges %d matches %d fallback %d rounds %d active %d | ticks (10 ns): loads %d rounds %d rotation %d by-keypoint %d out %d\n",
                head[0], head[1], head[2], head[3], head[4], head[8], head[9], head[10], head[11], head[12]);
    return SO_OK;
}

int so_track_group_create(int device, so_track_group** out) {
    if (!out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    so_track_group* g = new so_track_group();
    g->device = device;
    *out = g;
    return SO_OK;
}

void so_track_group_destroy(so_track_group* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (int i = 0; i < 2; i++) {
        g->h_tab[i].release();
        g->d_tab[i].release();
    }
    for (hipEvent_t e : {g->e0, g->e1, g->pe0, g->pe1})
        if (e) (void)hipEventDestroy(e);
    delete g;
}

int so_matcher_set_track_group(so_matcher* m, so_track_group* g) {
    if (!m || m->chain.active || m->pend.mode != 0) return SO_ERR_INVALID_ARG;
    if (g && g->device != m->device) return SO_ERR_INVALID_ARG;
    if (m->group && m->group != g) {  // leaving a group: a row recorded and never launched goes with it
        so_track_group* old = m->group;
        for (int i = 0; i < old->n_recs; i++)
            if (old->recs[(size_t)i].m == m) {
                for (int j = i; j + 1 < old->n_recs; j++) std::swap(old->recs[(size_t)j], old->recs[(size_t)j + 1]);
                old->n_recs--;
                break;
            }
    }
    m->group = g;
    return SO_OK;
}

int so_track_group_pending(so_track_group* g) { return g ? g->n_recs : 0; }

uint64_t so_matcher_stream_id(const so_matcher* m) { return m ? (uint64_t)(uintptr_t)m->stream : 0; }

static int matcher_idle(so_matcher* m) { return m && !m->chain.active && m->pend.mode == 0 && !m->batching; }

int so_matcher_private_stream(so_matcher* m) {
    if (!matcher_idle(m)) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    if (m->stream) SO_HIP(hipStreamSynchronize(m->stream));
    hipStream_t s = nullptr;
    SO_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (m->owns_stream && m->stream) (void)hipStreamDestroy(m->stream);
    m->stream = s;
    m->owns_stream = true;
    return SO_OK;
}

int so_matcher_share_stream(so_matcher* m, const so_matcher* other) {
    if (!matcher_idle(m) || !other || other->device != m->device || other == m) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(m->device));
    if (m->stream) SO_HIP(hipStreamSynchronize(m->stream));
    if (m->owns_stream && m->stream) (void)hipStreamDestroy(m->stream);
    m->stream = other->stream;
    m->owns_stream = false;
    return SO_OK;
}

int so_track_group_launch(so_track_group* g) {
    if (!g) return SO_ERR_INVALID_ARG;
    const int n = g->n_recs;
    if (n == 0) return SO_OK;
    g->n_recs = 0;
    const so_track_group::Rec& r0 = g->recs[0];
    hipStream_t s = r0.m->stream;
    bool want_events = false;
    int max_nq = 0, range_mask = 0;
    for (int i = 0; i < n; i++) {
        const so_track_group::Rec& r = g->recs[(size_t)i];
        // one stream (the members' handles come from one thread), one kind of stage per launch
        if (r.m->stream != s || r.search != r0.search || r.resolve != r0.resolve || r.pose != r0.pose || (r.search && r.mode != r0.mode)) {
            last_error_ref() = "so_track_group_launch: the members' recorded stages differ in kind, or their matchers do not share a stream "
                               "(create the handles on one thread, without so_runtime_private_streams)";
            // the members' waits must not spin for a launch that never comes
            (void)hipStreamSynchronize(s);
            for (int j = 0; j < n; j++) {
                so_matcher* m = g->recs[(size_t)j].m;
                if (!g->recs[(size_t)j].search) continue;
                chain_drop_search(m);
            }
            for (int j = 0; j < n; j++) {
                so_matcher* m = g->recs[(size_t)j].m;
                if (m->chain.holds_map) {
                    const_cast<so_map*>(m->chain.map)->grow_mu.unlock_shared();
                    m->chain.holds_map = false;
                }
                m->chain.active = false;
                m->chain.valid_edges = false;
            }
            return SO_ERR_INVALID_ARG;
        }
        want_events = want_events || r.m->profile;
        if (r.search) max_nq = std::max(max_nq, r.job.nq);
        range_mask |= 1 << (r.range ? 1 : 0);
    }
    SO_HIP(hipSetDevice(g->device));
    // rows -> one pinned block -> HBM (one copy launch): [pose rows | resolve rows | search rows]
    const size_t off_pose = 0;
    const size_t off_res = align256(off_pose + sizeof(so::PoseOptArgs) * (size_t)n);
    const size_t off_job = align256(off_res + (r0.resolve ? sizeof(so::TrackResolveArgs) * (size_t)n : 0));
    const size_t total = align256(off_job + (r0.search ? sizeof(so::TrackGroupJob) * (size_t)n : 0));
    g->flip ^= 1;
    PinBuf& H = g->h_tab[g->flip];
    DevBuf& D = g->d_tab[g->flip];
    int rc;
    if ((rc = H.ensure(total)) || (rc = D.ensure(total))) return rc;
    uint8_t* h = (uint8_t*)H.p;
    for (int i = 0; i < n; i++) {
        const so_track_group::Rec& r = g->recs[(size_t)i];
        if (r.pose) memcpy(h + off_pose + sizeof(so::PoseOptArgs) * (size_t)i, &r.pose_args, sizeof(so::PoseOptArgs));
        if (r.resolve) memcpy(h + off_res + sizeof(so::TrackResolveArgs) * (size_t)i, &r.res, sizeof(so::TrackResolveArgs));
        if (r.search) memcpy(h + off_job + sizeof(so::TrackGroupJob) * (size_t)i, &r.job, sizeof(so::TrackGroupJob));
    }
    std::atomic_thread_fence(std::memory_order_release);
    launch_stage_in(D.p, h, total, s);
    const uint8_t* d = (const uint8_t*)D.p;
    g->timed_search = g->timed_pose = false;
    if (want_events) {
        for (hipEvent_t* e : {&g->e0, &g->e1, &g->pe0, &g->pe1})
            if (!*e) SO_HIP(hipEventCreate(e));
    }
    if (r0.search) {
        if (want_events) SO_HIP(hipEventRecord(g->e0, s));
        so::launch_topk_track_group((const so::TrackGroupJob*)(d + off_job), n, r0.mode, max_nq, s);
        if (want_events) SO_HIP(hipEventRecord(g->e1, s));
        g->timed_search = want_events;
    }
    if (r0.resolve) so::launch_track_resolve_group((const so::TrackResolveArgs*)(d + off_res), n, max_nq, s);
    if (r0.pose) {
        if (want_events) SO_HIP(hipEventRecord(g->pe0, s));
        so::launch_pose_opt_chain_group((const so::PoseOptArgs*)(d + off_pose), n, range_mask, s);
        if (want_events) SO_HIP(hipEventRecord(g->pe1, s));
        g->timed_pose = want_events;
    }
    SO_HIP(hipGetLastError());
    g->launches++;
    return SO_OK;
}

int so_track_group_last_kernel_ms(so_track_group* g, float* search_ms, float* pose_ms) {
    if (!g) return SO_ERR_INVALID_ARG;
    if (search_ms) *search_ms = 0.f;
    if (pose_ms) *pose_ms = 0.f;
    if (search_ms && g->timed_search && hipEventSynchronize(g->e1) == hipSuccess) (void)hipEventElapsedTime(search_ms, g->e0, g->e1);
    if (pose_ms && g->timed_pose && hipEventSynchronize(g->pe1) == hipSuccess) (void)hipEventElapsedTime(pose_ms, g->pe0, g->pe1);
    return SO_OK;
}

int so_track_stage_last_pose_kernel_ms(so_matcher* m, float* ms) {
    if (!m || !ms) return SO_ERR_INVALID_ARG;
    *ms = 0.f;
    // (the wait returns on the kernel's completion word, which precedes the stop event's own completion)
    if (m->pe0 && m->pe1 && m->chain.events && !m->chain.active && hipEventSynchronize(m->pe1) == hipSuccess)
        (void)hipEventElapsedTime(ms, m->pe0, m->pe1);
    return SO_OK;
}

}  // extern "C"
