// ba.cpp — host driver + C ABI of the bundle-adjustment solver (include/swarmorb.h).
//
// Replaces Optimizer::LocalBundleAdjustment / BundleAdjustment (code/src/Optimizer.cc:42-237,436-740) from
// "build g2o graph" to "recover optimized data", and Optimizer::PoseOptimization (:239-434), on flattened problems.
// The host sorts the edges by landmark, stages the problem as one pinned block (one copy in, one result block out)
// and enqueues; the Levenberg-Marquardt control flow (code/Thirdparty/g2o/g2o/core/
// optimization_algorithm_levenberg.cpp:61-164, sparse_optimizer.cpp:354-419) runs ON THE DEVICE (BaLm state +
// ba_trial_decide_kernel): one host wait per optimize() call, three per local window.  Estimates live in a
// current/trial buffer pair, so g2o's push/pop/discardTop is an index flip.  Windows up to 43 free keyframes use
// the single-workgroup solvers of ba_kernels.hip, larger maps the blocked solver of ba_dense.hip with pair-list
// Schur gathers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <climits>
#include <cstdlib>
#include <thread>
#include <cstring>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <numeric>
#include <vector>

#include "ba_device.h"
#include "pose_convert.h"
#include "so_common.h"

using namespace so;

namespace {

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SO_OK;
        if (p) SO_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        SO_HIP(hipMalloc(&p, want));
        cap = want;
        return SO_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

constexpr int kMaxReducedDim = 96 * kDenseMaxPanels;  // reduced camera system: 6 * n_free <= 49152 (8192 keyframes; S is
                                                     // stored densely - 19 GB of the 288 GB - but only its block skyline is computed on)

// dense_flow_kernel's workgroups wait for each other, so every one of a launch must be resident (one per CU: 152 KB of
// LDS).  The tiles of all solver contexts of this process that are inside so_bundle_adjust at the same time have to fit
// the budget (one counter for the process: contexts on different devices share it, which only errs on the safe side);
// a context that does not get its share solves with the multi-launch path for that call.
std::atomic<int> g_flow_tiles{0};

// CUs the dataflow launches of this process may occupy together: what the device has (a partitioned MI355X exposes 32
// or 64 of its 256 per device) minus a reserve for the tracking threads' kernels; 0 when a CU cannot hold a workgroup
int flow_resident_budget(int device) {
    static std::atomic<int> cached[64];
    if (device < 0 || device >= 64) return 0;
    int v = cached[device].load();
    if (v != 0) return v > 0 ? v : 0;
    // what the runtime's occupancy calculator says the device can hold of these kernels (0: a workgroup does not fit a CU)
    int budget = -1;
    const int capacity = dense_flow_resident_capacity(device);
    if (capacity > 0) budget = std::max(capacity - 32, 0);  // 32 CUs stay with the tracking threads (quadtree alone wants 8 whole CUs)
    if (const char* e = getenv("SWARMORB_FLOW_BUDGET")) budget = atoi(e);  // diagnostic: claim more (or less) than the device has
    if (budget <= 0) budget = -1;
    cached[device].store(budget);
    return budget > 0 ? budget : 0;
}

// Host staging on a few threads: fn(part, n_parts) works on its share.  The workers belong to the solver context and
// are started by the first problem large enough to want them (from ~30 k observations - a 40-keyframe window - the
// staging is a sixth of the call); handing a job over costs a few microseconds.
class StagingPool {
public:
    ~StagingPool() { stop(); }
    template <typename F>
    void run(int n_parts, F fn) {
        if (n_parts <= 1) {
            fn(0, 1);
            return;
        }
        start(n_parts - 1);
        {
            std::unique_lock<std::mutex> lk(mu_);
            job_ = [&fn, n_parts](int t) { fn(t, n_parts); };
            n_parts_ = n_parts;
            pending_ = n_parts - 1;
            generation_++;
        }
        cv_.notify_all();
        fn(0, n_parts);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    void stop() {
        {
            std::unique_lock<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
        workers_.clear();
        quit_ = false;
    }

private:
    void start(int n) {
        while ((int)workers_.size() < n) {
            const int id = (int)workers_.size() + 1;
            workers_.emplace_back([this, id] {
                unsigned long seen = 0;
                for (;;) {
                    std::function<void(int)> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [&] { return quit_ || (generation_ != seen && id < n_parts_); });
                        if (quit_) return;
                        seen = generation_;
                        job = job_;
                    }
                    job(id);
                    std::unique_lock<std::mutex> lk(mu_);
                    if (--pending_ == 0) done_.notify_all();
                }
            });
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::function<void(int)> job_;
    unsigned long generation_ = 0;
    int n_parts_ = 0, pending_ = 0;
    bool quit_ = false;
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- the SE3Quat conversions at the map boundary (Converter.cc:37-47,49-74; se3quat.h:58-60): pose_convert.h, shared with the device ----
void pose_from_Tcw(const float* T, BaPose& P) {
    so::pose_from_T12_hd(T, P.q, P.t);
    P.pad = 0;
}

void pose_to_Tcw(const BaPose& P, float* T) { so::pose_to_T12_hd(P.q, P.t, T); }

}  // namespace

namespace so {  // (for matcher.cpp's tracking stages, which launch the indexed PoseOptimization kernel themselves)
void pose_from_Tcw12(const float* T, BaPose& P) { pose_from_Tcw(T, P); }
void pose_to_Tcw12(const BaPose& P, float* T) { pose_to_Tcw(P, T); }
}  // namespace so

struct so_ba_group;

struct so_ba {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;  // around the PoseOptimization kernel (its own pair: another thread may be in so_bundle_adjust)
    float pose_kernel_ms = 0.f;
    bool pose_ms_pending = false;  // pe0 / pe1 hold a finished measurement not yet read
    int pose_seq = 0;              // completion stamp of the last PoseOptimization launch (host spins on it)
    struct PosePending {           // so_pose_optimization_submit -> _wait
        bool active = false, launched = false, events = false;
        int n = 0, done_seq = 0;
        hipStream_t stream = nullptr;
        uint8_t* hout = nullptr;
        double* trace = nullptr;
    } pose_pending;
    bool pose_timing = true;       // HIP events around the PoseOptimization kernel (so_pose_optimization_set_timing)
    bool solve_timing = false;     // HIP events around every reduced-system solve (so_bundle_adjust_set_solve_timing)
    static constexpr int kSolveEvents = 32;  // the first trials of a call are event-timed around the solve kernel
    hipEvent_t ev_solve[2 * kSolveEvents] = {nullptr};
    float solve_ms = 0.f;
    int n_solves = 0;
    hipStream_t dense_side = nullptr;       // blocked dense solver: side stream + events of its look-ahead
    std::vector<hipEvent_t> dense_events;
    DensePlan plan;                         // blocked solver: tiles of every trailing update for the current structure
    StagingPool staging;                    // host threads of the problem staging (started on demand)
    unsigned flow_epoch = 0;                // single-launch dataflow solve: stamp of the last solve (never reset)
    unsigned* h_flow_abort = nullptr;       // host-mapped: stamped by a dataflow workgroup whose wait outlasted its budget
    unsigned* h_flow_abort_dev = nullptr;
    bool flow_broken = false;               // a dataflow solve of this context timed out: chain of launches from now on
    BaResidentSync* d_rsync = nullptr;      // the resident trial loop's meeting words (ba_kernels.hip: ba_lm_resident_kernel), lazily
    unsigned resident_epoch = 0;            // one per resident launch of this context (20 bits)
    long long resident_launches = 0;        // stages of this context that ran resident
    int flow_timeouts = 0;
    int flow_reserved = 0;                  // tiles this context holds of the process-wide residency budget
    std::vector<int> tile_first;
    int stage2_hint = 0;                    // trials the second stage of the last call needed (0: unknown, all are enqueued ahead)
    // ---- member of a so_ba_group: the LM chain of a local window is recorded and goes out merged with the other members' ----
    so_ba_group* group = nullptr;
    BaRecorder rec;
    hipEvent_t grp_uploaded = nullptr;      // on the member's own stream: its problem is in HBM
    hipStream_t grp_stream = nullptr;       // the group stream its round was issued on (top-ups and polls go there)
    bool grp_launched = false;              // (under the group's mutex) the round this member submitted to has been issued
    int grp_event_slot = 0;                 // which of the group's event pairs bracket that round's chain
    int done_seq = 0;                       // tags the completion words of a call (h_abort + 16)
    bool leftover_launches = false;         // the last call returned on its early completion word: launches may still be draining
    hipEvent_t e1a = nullptr;               // behind the early epilogue
    int linear_solver = 0;                  // so_ba_set_linear_solver: 0 direct (block-skyline Cholesky), 1 block-Jacobi PCG (ba_pcg.hip)
    BaPcgHost pcg;                          // its tolerance / iteration cap, workspace pointers and counters
    int pcg_nnz_blocks = 0;                 // nonzero 6 x 6 blocks of S in the last PCG problem
    Buf d_pcg, d_pcg_idx, d_pcg_val;
    BaLm* h_lm = nullptr;        // host-mapped copy of the LM state, written by the decision kernels
    BaLm* h_lm_dev = nullptr;
    uint8_t* h_abort = nullptr;  // host-mapped forceStopFlag the decision kernel polls
    uint8_t* h_abort_dev = nullptr;

    // d_in: the problem as one block (see Layout in so_bundle_adjust), staged in pinned h_in and moved with one
    // copy; d_out / h_out: the result block coming back the same way; the rest is device-only working storage
    Buf d_in, d_out, d_pose1, d_pt1, d_err, d_chi2, d_tab, d_Hpp, d_bp, d_Hll, d_bl, d_W, d_Dinv, d_db, d_BDinv, d_S,
        d_bs, d_xl, d_partial, d_po, d_lm, d_dense_ws, d_dense_x, d_pr_off, d_pr_cur, d_pr, d_big, d_scan_tmp, d_plan, d_flow, d_flow_big;
    void* h_in = nullptr;
    size_t h_in_cap = 0;
    void* h_out = nullptr;
    size_t h_out_cap = 0;
    void* h_plan = nullptr;  // pinned staging of the blocked solver's tile_first + tile lists
    size_t h_plan_cap = 0;
    uint8_t* h_po = nullptr;      // pinned + host-mapped staging for PoseOptimization
    uint8_t* h_po_dev = nullptr;
    size_t h_po_cap = 0;
    std::vector<Buf*> all() {
        return {&d_in, &d_out, &d_pose1, &d_pt1, &d_err, &d_chi2, &d_tab, &d_Hpp, &d_bp, &d_Hll, &d_bl, &d_W, &d_Dinv,
                &d_db, &d_BDinv, &d_S, &d_bs, &d_xl, &d_partial, &d_po, &d_lm, &d_dense_ws, &d_dense_x, &d_pr_off, &d_pr_cur, &d_pr, &d_big, &d_scan_tmp, &d_plan,
                &d_flow, &d_flow_big, &d_pcg, &d_pcg_idx, &d_pcg_val};
    }
};

// Local bundle adjustments of several agents (one so_ba each, each on its own local-mapping thread) as ONE chain of
// launches: include/swarmorb.h, so_ba_group_*.  A member's so_bundle_adjust stages and uploads its window on its own stream,
// records the launches of its LM chain (ba_device.h: BaRecorder) and hands the list in; the first member to arrive waits a
// short window for the others, merges the lists phase by phase - launches of the same kind become one launch with the member
// as blockIdx.y - and issues everything on the group's stream.  Every member then waits for its own completion word.
struct so_ba_group {
    int device = 0;
    hipStream_t stream = nullptr;   // rounds are issued on `streams` in turn (stream = streams[0]): a round whose members arrived
    static constexpr int kMaxStreams = 4;  // late does not queue behind the chain of the round before
    hipStream_t streams[kMaxStreams] = {};
    int n_streams = 1, next_stream = 0;
    int n_members = 0;          // members registered (so_ba_set_group)
    double window_us = 250.0;   // how long the first arrival of a round waits for the rest
    std::mutex mu;
    std::condition_variable cv;
    std::vector<so_ba*> waiting;  // this round's arrivals (under mu)
    bool launching = false;       // a leader is issuing a round (the table blocks are its own until it is done)
    static constexpr int kRowSlots = 4;
    void* h_rows = nullptr;       // pinned staging of the BaDev table, kRowSlots blocks used in turn
    void* d_rows = nullptr;
    size_t rows_cap = 0;          // rows per block
    int flip = 0;
    hipEvent_t row_copied[kRowSlots] = {};  // behind the LAST launch of the round that used block i: both its pinned and its device half are free again
    bool row_used[kRowSlots] = {};
    static constexpr int kEventPairs = 4;
    hipEvent_t ev[2 * kEventPairs] = {};
    int ev_next = 0;
    // statistics
    long long rounds = 0, members_total = 0, grouped_launches = 0, solo_launches = 0, rows_launched = 0;
    bool released = false;  // so_ba_group_destroy was called: the last member to leave frees the group
};

namespace {

// Issue one round: the recorded chains of `mem` merged phase by phase on the group's stream.
int ba_group_issue(so_ba_group* g, const std::vector<so_ba*>& mem) {
    SO_HIP(hipSetDevice(g->device));
    hipStream_t gs = g->streams[g->next_stream];
    g->next_stream = (g->next_stream + 1) % g->n_streams;
    const int M = (int)mem.size();
    for (so_ba* b : mem) b->grp_stream = gs;
    // stage 2: enqueue what the members' last calls needed + 2 (a stage that wants more tops itself up after its wait)
    int ahead2 = 1;
    bool any_hint_unknown = false;
    for (so_ba* b : mem) {
        if (b->stage2_hint <= 0) any_hint_unknown = true;
        ahead2 = std::max(ahead2, b->stage2_hint + 2);
    }
    static const bool no_hint = getenv("SWARMORB_BA_NO_STAGE2_HINT") != nullptr;
    const int last_phase2 = (any_hint_unknown || no_hint) ? kBaPhaseEpilogue - 1 : kBaPhaseStage2 + ahead2;
    // the BaDev table: the distinct argument blocks of every member (its stage-1 and stage-2 views)
    std::vector<std::vector<int>> row_of((size_t)M);
    std::vector<BaDev> rows;
    for (int m = 0; m < M; m++) {
        const std::vector<BaLaunchRec>& L = mem[(size_t)m]->rec.list;
        row_of[(size_t)m].assign(L.size(), -1);
        const size_t first_row = rows.size();
        for (size_t i = 0; i < L.size(); i++) {
            if (L[i].kind == kBaKSolo || L[i].kind == kBaKSignal) continue;
            int found = -1;
            for (size_t q = first_row; q < rows.size() && found < 0; q++)
                if (memcmp(&rows[q], &L[i].d, sizeof(BaDev)) == 0) found = (int)q;
            if (found < 0) {
                rows.push_back(L[i].d);
                found = (int)rows.size() - 1;
            }
            row_of[(size_t)m][i] = found;
        }
    }
    if (rows.size() > g->rows_cap) {
        SO_HIP(hipStreamSynchronize(gs));  // (an earlier round may still read the old blocks)
        for (int i = 0; i < g->n_streams; i++) SO_HIP(hipStreamSynchronize(g->streams[i]));
        if (g->h_rows) (void)hipHostFree(g->h_rows);
        if (g->d_rows) (void)hipFree(g->d_rows);
        g->h_rows = g->d_rows = nullptr;
        g->rows_cap = 0;
        const size_t cap = rows.size() + 16;
        SO_HIP(hipHostMalloc(&g->h_rows, so_ba_group::kRowSlots * cap * sizeof(BaDev), hipHostMallocDefault));
        SO_HIP(hipMalloc(&g->d_rows, so_ba_group::kRowSlots * cap * sizeof(BaDev)));
        g->rows_cap = cap;
        for (bool& u : g->row_used) u = false;
    }
    // Rounds are issued without waiting for the ones before (members of different rounds overlap on the GPU's queue): a
    // pinned block is rewritten only when the copy that read it has run, a device block only behind the kernels that read it
    // (same stream).
    g->flip = (g->flip + 1) % so_ba_group::kRowSlots;
    if (g->row_used[g->flip]) SO_HIP(hipEventSynchronize(g->row_copied[g->flip]));
    BaDev* h_rows = (BaDev*)g->h_rows + (size_t)g->flip * g->rows_cap;
    BaDev* d_rows = (BaDev*)g->d_rows + (size_t)g->flip * g->rows_cap;
    memcpy(h_rows, rows.data(), rows.size() * sizeof(BaDev));
    const int slot = g->ev_next;
    g->ev_next = (g->ev_next + 1) % so_ba_group::kEventPairs;
    for (so_ba* b : mem) {
        SO_HIP(hipStreamWaitEvent(gs, b->grp_uploaded, 0));  // the member's problem is in HBM before its chain starts
        b->grp_event_slot = slot;
    }
    SO_HIP(hipMemcpyAsync(d_rows, h_rows, rows.size() * sizeof(BaDev), hipMemcpyHostToDevice, gs));
    SO_HIP(hipEventRecord(g->ev[2 * slot], gs));
    // merge: phases in ascending order; inside a phase the members' launches are aligned by their position in it
    std::vector<size_t> pos((size_t)M, 0);
    long long n_grouped = 0, n_solo = 0, n_rows = 0;
    for (;;) {
        int phase = INT_MAX;
        for (int m = 0; m < M; m++) {
            const std::vector<BaLaunchRec>& L = mem[(size_t)m]->rec.list;
            if (pos[(size_t)m] < L.size()) phase = std::min(phase, L[pos[(size_t)m]].phase);
        }
        if (phase == INT_MAX) break;
        const bool skip = phase > last_phase2 && phase < kBaPhaseEpilogue;  // stage-2 trials beyond what the members' last calls needed
        for (int step = 0;; step++) {
            // the launches at position `step` of this phase, member by member
            int kinds_present = 0;
            bool any = false;
            for (int m = 0; m < M; m++) {
                const std::vector<BaLaunchRec>& L = mem[(size_t)m]->rec.list;
                const size_t i = pos[(size_t)m] + (size_t)step;
                if (i < L.size() && L[i].phase == phase) {
                    any = true;
                    kinds_present |= 1 << L[i].kind;
                }
            }
            if (!any) {
                for (int m = 0; m < M; m++) {  // the phase is over for everybody: move on
                    const std::vector<BaLaunchRec>& L = mem[(size_t)m]->rec.list;
                    while (pos[(size_t)m] < L.size() && L[pos[(size_t)m]].phase == phase) pos[(size_t)m]++;
                }
                break;
            }
            if (skip) continue;
            for (int kind = 0; kind < kBaKCount; kind++) {
                if (!(kinds_present & (1 << kind))) continue;
                BaGroupArgs A{};
                int max_grid = 1;
                size_t lds = 0;
                for (int m = 0; m < M; m++) {
                    const std::vector<BaLaunchRec>& L = mem[(size_t)m]->rec.list;
                    const size_t i = pos[(size_t)m] + (size_t)step;
                    if (i >= L.size() || L[i].phase != phase || L[i].kind != kind) continue;
                    const BaLaunchRec& R = L[i];
                    if (kind == kBaKSolo) {
                        R.solo(gs);
                        n_solo++;
                        continue;
                    }
                    const int k = A.n++;
                    A.grid[k] = R.grid;
                    A.row[k] = row_of[(size_t)m][i] < 0 ? 0 : row_of[(size_t)m][i];
                    A.i0[k] = R.i0; A.i1[k] = R.i1; A.i2[k] = R.i2;
                    A.f0[k] = R.f0;
                    A.p0[k] = R.p0; A.p1[k] = R.p1; A.p2[k] = R.p2; A.p3[k] = R.p3;
                    max_grid = std::max(max_grid, R.grid);
                    lds = std::max(lds, R.lds);
                    if (A.n == kBaGroupMax) {  // (more members than an argument block holds: another launch)
                        launch_ba_group(kind, d_rows, A, max_grid, lds, gs);
                        n_grouped++;
                        n_rows += A.n;
                        A = BaGroupArgs{};
                        max_grid = 1;
                        lds = 0;
                    }
                }
                if (kind != kBaKSolo && A.n > 0) {
                    launch_ba_group(kind, d_rows, A, max_grid, lds, gs);
                    n_grouped++;
                    n_rows += A.n;
                }
            }
        }
    }
    SO_HIP(hipEventRecord(g->ev[2 * slot + 1], gs));
    SO_HIP(hipEventRecord(g->row_copied[g->flip], gs));
    g->row_used[g->flip] = true;
    SO_HIP(hipGetLastError());
    {   // (so_ba_group_stats reads these under the mutex)
        std::lock_guard<std::mutex> lk(g->mu);
        g->rounds++;
        g->members_total += M;
        g->grouped_launches += n_grouped;
        g->solo_launches += n_solo;
        g->rows_launched += n_rows;
    }
    return SO_OK;
}

// A member hands its recorded chain in and returns once the round it belongs to has been issued.
int ba_group_submit(so_ba* b) {
    so_ba_group* g = b->group;
    std::unique_lock<std::mutex> lk(g->mu);
    b->grp_launched = false;
    g->waiting.push_back(b);
    if (g->waiting.size() > 1) {  // somebody is collecting this round already
        g->cv.notify_all();
        g->cv.wait(lk, [b] { return b->grp_launched; });
        return b->rec.list.empty() ? SO_OK : SO_ERR_HIP;  // (the leader empties the lists it has issued; a failed round leaves them)
    }
    // first of a round: wait for the other members, but not for long
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds((long long)g->window_us);
    g->cv.wait_until(lk, deadline, [g] { return (int)g->waiting.size() >= g->n_members; });
    g->cv.wait(lk, [g] { return !g->launching; });  // (the round before is still being issued)
    std::vector<so_ba*> mem;
    mem.swap(g->waiting);
    g->launching = true;
    lk.unlock();
    const int rc = ba_group_issue(g, mem);
    lk.lock();
    g->launching = false;
    for (so_ba* m : mem) {
        if (rc == SO_OK) m->rec.list.clear();
        m->grp_launched = true;
    }
    lk.unlock();
    g->cv.notify_all();
    return rc;
}

int ensure_pinned(void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return SO_OK;
    if (*p) SO_HIP(hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    SO_HIP(hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return SO_OK;
}

struct Layout {  // offsets into a staging block, 256-byte aligned
    size_t total = 0;
    size_t add(size_t bytes) {
        const size_t at = total;
        total = (total + bytes + 255) & ~(size_t)255;
        return at;
    }
};

struct Run {
    so_ba* b;
    BaDev d{};
    int n_free = 0, n_active_edges = 0;
    int nb_err = 1, nb_upd = 1;
    int blocks_enqueued = 0;  // trial blocks of this call so far (indexes the solve-event pool)
    int n_reordered = 0;  // keyframes moved to the separator block at the end of the elimination order
    const volatile uint8_t* stop = nullptr;
    bool terminate() const { return stop && *stop; }
    bool returned_early = false;  // the results were taken behind the early epilogue (its stop event is e1a)
};

// internal status of so_bundle_adjust's first attempt: a workgroup of a single-launch (dataflow) solve waited longer than
// its budget for another one - the launch was not fully resident (another process on the GPU, a CU mask, ...).  The call
// is repeated on the chain-of-launches path; never returned to the caller.
constexpr int kErrFlowTimeout = 1000;

// Windows whose trials run as resident launches (ba_lm_resident_kernel), per device: their workgroups hold a CU's LDS each for the
// length of a stage, so only so many may run at once (the next caller takes the chain of launches; nobody waits).
// SWARMORB_BA_RESIDENT=1 switches the path on, SWARMORB_BA_RESIDENT_WGS=g the workgroups per window (default 32),
// SWARMORB_BA_RESIDENT_MAX=n the windows per device (default 6: 192 of 256 CUs at 32 workgroups).
std::atomic<int> g_resident_in_flight[64];
struct ResidentSlot {
    int device = -1;
    bool take(int dev, int limit) {
        if (dev < 0 || dev >= 64) return false;
        if (g_resident_in_flight[dev].fetch_add(1) >= limit) {
            g_resident_in_flight[dev].fetch_sub(1);
            return false;
        }
        device = dev;
        return true;
    }
    ~ResidentSlot() {
        if (device >= 0) g_resident_in_flight[device].fetch_sub(1);
    }
};

// Wait for the stream; with a forceStopFlag, poll it meanwhile and forward it to the device.
int wait_stream(Run& r) {
    hipStream_t s = r.b->stream;
    if (!r.stop) {
        SO_HIP(hipStreamSynchronize(s));
        return SO_OK;
    }
    // The flag is mirrored, not latched (a stop request withdrawn between two optimize() calls does not abort the next
    // one), and the poll backs off: the local-mapping thread must not burn a core next to the tracking thread, which
    // is itself bound by host enqueue latency.  20 us per nap: well below one LM trial of a window (~90 us).
    int spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return SO_OK;
        if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
        *r.b->h_abort = *r.stop ? 1 : 0;
        if (++spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg, driven from the device: the host
// enqueues the prologue (errors, linearisation, lambda init) and `iterations` trials, waits once and only
// enqueues more if trials were rejected (each rejected trial needs one more than the iteration count).
//
// What follows the stage rides behind its trials instead of waiting for the host to see the stage end (a stream
// synchronise + the next launches idled the GPU ~35 us at every stage boundary and ~45 us at the end of the call):
// `after_trials` is enqueued behind every batch of trials.  It is either the NEXT stage's prologue, gated on the device
// (kBaGateIdle: it takes effect only if this stage is over when it runs - then `*next_begun` is set and this stage's
// results are the ones ba_stage_begin_kernel saved), or the call's epilogue (finish + copy out), which is harmless to
// run early and is simply run again behind the extra trials.
struct StageChain {
    std::function<void()> after_trials;  // may be empty
    bool starts_next_stage = false;
};

int optimize(Run& r, int iterations, bool prologue, const StageChain& chain, int* done_out, double* chi_out, bool* next_begun) {
    so_ba* b = r.b;
    hipStream_t s = b->stream;
    *done_out = 0;
    if (next_begun) *next_begun = false;
    if (r.n_free + (r.n_active_edges > 0 ? 1 : 0) == 0) return SO_OK;  // 0 vertices to optimize
    const uint8_t* abort_dev = r.stop ? b->h_abort_dev : nullptr;
    if (prologue) {
        launch_ba_errors(r.d, 0, kBaGateNone, r.nb_err, s);
        launch_ba_build(r.d, kBaGateNone, s);
        launch_ba_stage_begin(r.d, r.nb_err, iterations, b->h_lm_dev, kBaGateNone, nullptr, s);
        SO_HIP(hipGetLastError());
        b->h_lm->stages_begun++;  // (the host copy trails the device until the next wait)
    }
    const int my_stage = b->h_lm->stages_begun;
    const int trials_before = b->h_lm->trials, first_block = r.blocks_enqueued;  // h_lm: state after the last wait
    int budget = iterations, rc;
    BaLm lm;
    for (;;) {
        for (int i = 0; i < budget; i++) {
            const int k = r.blocks_enqueued++;
            const bool timed = b->solve_timing && k < so_ba::kSolveEvents;
            launch_ba_trial(r.d, r.nb_err, r.nb_upd, abort_dev, b->h_lm_dev,
                            timed ? b->ev_solve[2 * k] : nullptr, timed ? b->ev_solve[2 * k + 1] : nullptr, s);
        }
        if (chain.after_trials) chain.after_trials();
        SO_HIP(hipGetLastError());
        if ((rc = wait_stream(r))) return rc;
        if (*(volatile unsigned*)b->h_flow_abort != 0) return kErrFlowTimeout;  // a dataflow solve gave up waiting
        memcpy(&lm, b->h_lm, sizeof(lm));
        if (lm.stages_begun != my_stage) {  // the chained prologue ran: this stage is over and the next one has begun
            if (next_begun) *next_begun = true;
            break;
        }
        if (!lm.active) break;
        budget = std::max(1, lm.iterations - lm.it);
    }
    const bool chained = lm.stages_begun != my_stage;
    const int real = lm.trials - trials_before;  // the first `real` blocks of this stage ran, the rest returned at once
    for (int k = first_block; b->solve_timing && k < first_block + real && k < so_ba::kSolveEvents; k++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, b->ev_solve[2 * k], b->ev_solve[2 * k + 1]) == hipSuccess) {
            b->solve_ms += ms;
            b->n_solves++;
        }
    }
    *done_out = chained ? lm.prev_done : lm.done;
    *chi_out = chained ? lm.prev_chi_out : lm.chi_out;
    return SO_OK;
}

}  // namespace

extern "C" {

int so_ba_create(int device, so_ba** out) {
    if (!out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(device));
    so_ba* b = new so_ba();
    b->device = device;
    // the solver's stream is created by the first so_bundle_adjust: a handle used only for PoseOptimization (which runs on
    // the tracking thread's stream) must not take one of the few hardware queues the runtime multiplexes streams onto
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipEventCreate(&b->e0);
    if (e == hipSuccess) e = hipEventCreate(&b->e1);
    if (e == hipSuccess) e = hipEventCreate(&b->e1a);
    if (e == hipSuccess) e = hipEventCreate(&b->pe0);
    if (e == hipSuccess) e = hipEventCreate(&b->pe1);
    for (hipEvent_t& ev : b->ev_solve)
        if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_lm, sizeof(BaLm), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&b->h_lm_dev, b->h_lm, 0);
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_flow_abort, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&b->h_flow_abort_dev, b->h_flow_abort, 0);
    if (e == hipSuccess) *b->h_flow_abort = 0;
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_abort, 64, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&b->h_abort_dev, b->h_abort, 0);
    if (e != hipSuccess) {
        delete b;
        return hip_fail(e, "ba init", __FILE__, __LINE__);
    }
    *out = b;
    return SO_OK;
}

void so_ba_destroy(so_ba* b) {
    if (!b) return;
    if (b->group) so_ba_set_group(b, nullptr);
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (Buf* q : b->all()) q->release();
    if (b->dense_side) (void)hipStreamSynchronize(b->dense_side);
    for (hipEvent_t e : b->dense_events)
        if (e) (void)hipEventDestroy(e);
    if (b->dense_side) (void)hipStreamDestroy(b->dense_side);
    if (b->h_lm) (void)hipHostFree(b->h_lm);
    if (b->h_flow_abort) (void)hipHostFree(b->h_flow_abort);
    if (b->d_rsync) (void)hipFree(b->d_rsync);
    if (b->h_in) (void)hipHostFree(b->h_in);
    if (b->h_out) (void)hipHostFree(b->h_out);
    if (b->h_plan) (void)hipHostFree(b->h_plan);
    if (b->h_abort) (void)hipHostFree(b->h_abort);
    for (hipEvent_t ev : b->ev_solve)
        if (ev) (void)hipEventDestroy(ev);
    if (b->h_po) (void)hipHostFree(b->h_po);
    if (b->e0) (void)hipEventDestroy(b->e0);
    if (b->e1) (void)hipEventDestroy(b->e1);
    if (b->e1a) (void)hipEventDestroy(b->e1a);
    if (b->grp_uploaded) (void)hipEventDestroy(b->grp_uploaded);
    if (b->pe0) (void)hipEventDestroy(b->pe0);
    if (b->pe1) (void)hipEventDestroy(b->pe1);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int so_ba_group_create(int device, double window_us, so_ba_group** out) {
    if (!out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(device));
    so_ba_group* g = new so_ba_group();
    g->device = device;
    if (window_us > 0.0) g->window_us = window_us;
    g->n_streams = getenv("SWARMORB_BA_GROUP_STREAMS") ? std::min((int)so_ba_group::kMaxStreams, std::max(1, atoi(getenv("SWARMORB_BA_GROUP_STREAMS")))) : 1;
    hipError_t e = hipSuccess;
    for (int i = 0; i < g->n_streams && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&g->streams[i], hipStreamNonBlocking);
    g->stream = g->streams[0];
    for (hipEvent_t& ev : g->ev)
        if (e == hipSuccess) e = hipEventCreate(&ev);
    for (hipEvent_t& ev : g->row_copied)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        delete g;
        return hip_fail(e, "ba group init", __FILE__, __LINE__);
    }
    *out = g;
    return SO_OK;
}

static void ba_group_free(so_ba_group* g);

void so_ba_group_destroy(so_ba_group* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->released = true;
        if (g->n_members > 0) return;  // members (other threads' solver contexts) still refer to it: the last one to leave frees it
    }
    ba_group_free(g);
}

static void ba_group_free(so_ba_group* g) {
    (void)hipSetDevice(g->device);
    for (int i = 0; i < g->n_streams; i++)
        if (g->streams[i]) (void)hipStreamSynchronize(g->streams[i]);
    for (hipEvent_t ev : g->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : g->row_copied)
        if (ev) (void)hipEventDestroy(ev);
    if (g->h_rows) (void)hipHostFree(g->h_rows);
    if (g->d_rows) (void)hipFree(g->d_rows);
    for (int i = 0; i < g->n_streams; i++)
        if (g->streams[i]) (void)hipStreamDestroy(g->streams[i]);
    delete g;
}

int so_ba_set_group(so_ba* b, so_ba_group* g) {
    if (!b || (g && g->device != b->device)) return SO_ERR_INVALID_ARG;
    if (b->group == g) return SO_OK;
    if (b->group) {
        bool last;
        {
            std::lock_guard<std::mutex> lk(b->group->mu);
            last = --b->group->n_members == 0 && b->group->released;
        }
        if (last) ba_group_free(b->group);
    }
    b->group = g;
    if (g) {
        std::lock_guard<std::mutex> lk(g->mu);
        g->n_members++;
    }
    return SO_OK;
}

int so_ba_resident_stages(const so_ba* b, long long* n_out) {
    if (!b || !n_out) return SO_ERR_INVALID_ARG;
    *n_out = b->resident_launches;
    return SO_OK;
}

int so_ba_group_stats(so_ba_group* g, double* out8) {
    if (!g || !out8) return SO_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    out8[0] = (double)g->rounds; out8[1] = (double)g->members_total; out8[2] = (double)g->grouped_launches;
    out8[3] = (double)g->solo_launches; out8[4] = (double)g->rows_launched; out8[5] = (double)g->n_members; out8[6] = g->window_us; out8[7] = 0.0;
    return SO_OK;
}

void so_ba_options_local(so_ba_options* o) {
    if (!o) return;
    o->its_stage1 = 5;
    o->its_stage2 = 10;
    o->robust = 1;
    o->huber_delta = std::sqrt(5.991f);  // const float thHuberMono = sqrt(5.991), Optimizer.cc:547
    o->chi2_threshold = 5.991f;
}

void so_ba_options_global(so_ba_options* o, int32_t n_iterations, int32_t robust) {
    if (!o) return;
    o->its_stage1 = n_iterations;
    o->its_stage2 = 0;
    o->robust = robust;
    o->huber_delta = std::sqrt(5.99f);  // const float thHuber2D = sqrt(5.99), Optimizer.cc:91
    o->chi2_threshold = 5.991f;
}

// Diagnostic: keep `workgroups` workgroups of `lds_bytes` LDS each busy-waiting for `milliseconds` on a stream of their
// own (asynchronous: returns once the launch is queued; the memory behind it lives as long as the process).
int so_runtime_occupy(int device, int workgroups, int lds_bytes, int milliseconds) {
    static hipStream_t stream[64] = {nullptr};
    static unsigned* sink[64] = {nullptr};
    if (device < 0 || device >= 64) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(device));
    if (!stream[device]) {
        SO_HIP(hipStreamCreateWithFlags(&stream[device], hipStreamNonBlocking));
        SO_HIP(hipMalloc((void**)&sink[device], 64));
    }
    return launch_occupy(workgroups, lds_bytes, milliseconds, sink[device], stream[device]) ? SO_OK : SO_ERR_INVALID_ARG;
}

static int bundle_adjust_once(so_ba* b, const so_ba_problem* p, const so_ba_options* opt, const volatile uint8_t* stop,
                              float* Tcw_out, float* Xw_out, uint8_t* edge_outlier, double* edge_chi2, so_ba_info* info);

int so_bundle_adjust(so_ba* b, const so_ba_problem* p, const so_ba_options* opt, const volatile uint8_t* stop,
                     float* Tcw_out, float* Xw_out, uint8_t* edge_outlier, double* edge_chi2, so_ba_info* info) {
    int rc = bundle_adjust_once(b, p, opt, stop, Tcw_out, Xw_out, edge_outlier, edge_chi2, info);
    if (rc == kErrFlowTimeout) {
        // The single-launch solve was not fully resident and its workgroups gave up waiting for each other (the kernels
        // have run to their end: the stream is drained).  Nothing has been written to the caller's arrays; the inputs
        // are untouched.  This context solves with the chain of launches from now on, starting with this very call.
        b->flow_broken = true;
        b->flow_timeouts++;
        fprintf(stderr, "[swarmorb] bundle adjustment: the single-launch solve timed out waiting for a workgroup that was "
                        "not resident (is another process using the GPU?); repeating the call on the multi-launch path, "
                        "which this solver context keeps from now on (SWARMORB_DENSE_NO_FLOW=1 selects it up front)\n");
        rc = bundle_adjust_once(b, p, opt, stop, Tcw_out, Xw_out, edge_outlier, edge_chi2, info);
        if (rc == kErrFlowTimeout) {  // cannot happen: the second attempt launches no dataflow kernel
            last_error_ref() = "bundle adjustment: dataflow solve timed out twice";
            rc = SO_ERR_HIP;
        }
    }
    return rc;
}

static int bundle_adjust_once(so_ba* b, const so_ba_problem* p, const so_ba_options* opt, const volatile uint8_t* stop,
                              float* Tcw_out, float* Xw_out, uint8_t* edge_outlier, double* edge_chi2, so_ba_info* info) {
    if (!b || !p || !opt || !Tcw_out || !Xw_out) return SO_ERR_INVALID_ARG;
    if (p->n_poses < 0 || p->n_points < 0 || p->n_edges < 0) return SO_ERR_INVALID_ARG;
    if ((p->n_poses > 0 && (!p->Tcw || !p->fixed || !p->intr)) || (p->n_points > 0 && !p->Xw) ||
        (p->n_edges > 0 && (!p->edge_pose || !p->edge_point || !p->obs || !p->inv_sigma2)))
        return SO_ERR_INVALID_ARG;
    const int nP = p->n_poses, nL = p->n_points, nE = p->n_edges;
    for (int e = 0; e < nE; e++)
        if (p->edge_pose[e] < 0 || p->edge_pose[e] >= nP || p->edge_point[e] < 0 || p->edge_point[e] >= nL) {
            last_error_ref() = "edge references a vertex out of range";
            return SO_ERR_INVALID_ARG;
        }
    const double t_begin = now_ms();
    SO_HIP(hipSetDevice(b->device));
    if (b->leftover_launches) {  // the last call returned on its early completion word: let what was queued behind it drain
        SO_HIP(hipStreamSynchronize(b->stream));
        b->leftover_launches = false;
    }
    b->solve_ms = 0.f;
    b->n_solves = 0;
    *(volatile unsigned*)b->h_flow_abort = 0;
    so_ba_info inf;
    memset(&inf, 0, sizeof(inf));
    Run r;
    r.b = b;
    r.stop = stop;
    struct FlowLease {  // the context's share of the residency budget goes back on every way out
        so_ba* b;
        ~FlowLease() {
            if (b->flow_reserved > 0) {
                // the CUs go back only when this context's launches are over: a no-op on the normal way out (the results
                // have been copied back), a real wait on an error return
                if (b->stream) (void)hipStreamSynchronize(b->stream);
                g_flow_tiles.fetch_sub(b->flow_reserved);
            }
            b->flow_reserved = 0;
        }
    } flow_lease{b};

    auto finish_untouched = [&]() {  // Optimizer.cc:631-633: return before optimising
        for (int i = 0; i < nP; i++) {
            BaPose P;
            pose_from_Tcw(p->Tcw + 12 * (size_t)i, P);  // toSE3Quat and back, like the reference's recovery loop
            pose_to_Tcw(P, Tcw_out + 12 * (size_t)i);
        }
        for (size_t i = 0; i < (size_t)nL * 3; i++) Xw_out[i] = (float)(double)p->Xw[i];
        if (edge_outlier) memset(edge_outlier, 0, (size_t)nE);
        if (edge_chi2) for (int e = 0; e < nE; e++) edge_chi2[e] = 0.0;
        inf.wall_ms = (float)(now_ms() - t_begin);
        if (info) *info = inf;
    };
    if (r.terminate()) {
        inf.aborted = 1;
        finish_untouched();
        return SO_OK;
    }
    if (nE == 0) {
        finish_untouched();
        return SO_OK;
    }

    // SparseOptimizer::initializeOptimization(0) + buildIndexMapping (sparse_optimizer.cpp:166-270): every edge is
    // at level 0, a keyframe gets a hessian index if it is not fixed and has an edge
    const double tS0 = now_ms();
    std::vector<uint8_t> touched((size_t)nP, 0);
    for (int e = 0; e < nE; e++) touched[(size_t)p->edge_pose[e]] = 1;
    int nf = 0;
    for (int i = 0; i < nP; i++) nf += (touched[(size_t)i] && !p->fixed[i]) ? 1 : 0;
    if (6 * nf > kMaxReducedDim) {
        last_error_ref() = "reduced camera system too large for the blocked solver (more than 8192 free keyframes)";
        return SO_ERR_CAPACITY;
    }
    r.n_free = nf;
    r.n_active_edges = nE;

    // ---- the problem as ONE staging block (pinned), mirrored by ONE device block ----
    Layout L;
    const size_t o_pose = L.add(sizeof(BaPose) * (size_t)nP), o_pt = L.add(sizeof(double) * 3 * (size_t)nL),
                 o_intr = L.add(sizeof(double) * 4 * (size_t)nP), o_obs = L.add(sizeof(double) * 2 * (size_t)nE),
                 o_w = L.add(sizeof(double) * (size_t)nE), o_epose = L.add(sizeof(int) * (size_t)nE),
                 o_ept = L.add(sizeof(int) * (size_t)nE), o_ptoff = L.add(sizeof(int) * ((size_t)nL + 1)),
                 o_hidx = L.add(sizeof(int) * (size_t)nP), o_free = L.add(sizeof(int) * (size_t)std::max(nf, 1)),
                 o_poseoff = L.add(sizeof(int) * ((size_t)nf + 1)), o_pedges = L.add(sizeof(int) * (size_t)nE),
                 o_pepoint = L.add(sizeof(int) * (size_t)nE), o_eact = L.add((size_t)nE),
                 o_ptact = L.add((size_t)std::max(nL, 1));
    int rc;
    if ((rc = ensure_pinned(&b->h_in, &b->h_in_cap, L.total))) return rc;
    if ((rc = b->d_in.ensure(L.total))) return rc;
    uint8_t* hb = (uint8_t*)b->h_in;
    BaPose* h_pose = (BaPose*)(hb + o_pose);
    double* h_pt = (double*)(hb + o_pt);
    double* h_intr = (double*)(hb + o_intr);
    double* h_obs = (double*)(hb + o_obs);
    double* h_w = (double*)(hb + o_w);
    int* h_epose = (int*)(hb + o_epose);
    int* h_ept = (int*)(hb + o_ept);
    int* h_ptoff = (int*)(hb + o_ptoff);
    int* h_hidx = (int*)(hb + o_hidx);
    int* h_free = (int*)(hb + o_free);
    int* h_poseoff = (int*)(hb + o_poseoff);
    int* h_pedges = (int*)(hb + o_pedges);
    int* h_pepoint = (int*)(hb + o_pepoint);
    uint8_t* h_eact = hb + o_eact;
    uint8_t* h_ptact = hb + o_ptact;

    const double tS1 = now_ms();
    for (int i = 0; i < nP; i++) pose_from_Tcw(p->Tcw + 12 * (size_t)i, h_pose[i]);  // toSE3Quat
    for (size_t i = 0; i < (size_t)nL * 3; i++) h_pt[i] = (double)p->Xw[i];          // toVector3d
    for (size_t i = 0; i < (size_t)nP * 4; i++) h_intr[i] = (double)p->intr[i];
    const double tS2 = now_ms();
    // stable counting sort of the edges by landmark: a landmark's observations become contiguous
    for (int i = 0; i <= nL; i++) h_ptoff[i] = 0;
    for (int e = 0; e < nE; e++) h_ptoff[p->edge_point[e] + 1]++;
    for (int i = 0; i < nL; i++) h_ptoff[i + 1] += h_ptoff[i];
    std::vector<int> perm((size_t)nE);  // sorted position -> original edge index
    {
        bool sorted = true;  // (a caller that walks its map point by point hands the edges over in landmark order already)
        for (int e = 1; e < nE && sorted; e++) sorted = p->edge_point[e - 1] <= p->edge_point[e];
        if (sorted) {
            for (int e = 0; e < nE; e++) perm[(size_t)e] = e;
        } else {
            std::vector<int> fill(h_ptoff, h_ptoff + nL);
            for (int e = 0; e < nE; e++) perm[(size_t)fill[(size_t)p->edge_point[e]]++] = e;
        }
    }
    const double tS3 = now_ms();
    for (int i = 0, h = 0; i < nP; i++) {
        h_hidx[i] = -1;
        if (touched[(size_t)i] && !p->fixed[i]) {
            h_hidx[i] = h;
            h_free[h++] = i;
        }
    }
    // Large maps: elimination order.  g2o numbers the free keyframes by mnId and leaves the ordering of the reduced
    // system to the sparse solver (AMD inside SimplicialLDLT, linear_solver_eigen.h:94-124); here the order decides the
    // block skyline.  A merged multi-agent map in natural order (agent by agent) is banded per agent - until a landmark
    // seen from two distant places (trajectories crossing, a loop) stretches the row envelope of every later keyframe
    // that sees it across everything in between, and the factorisation becomes one chain over all panels.  Moving those
    // keyframes to the end (a separator block, the last rows of an arrowhead) leaves every agent's band on its own and
    // lets the dataflow solve run their chains side by side: of every landmark whose observers span more than kLinkSpan
    // keyframes, the observers beyond that span from the first one go to the back; both parts keep their natural order.
    // A map in which everything sees everything is unaffected (its keyframes all move, i.e. none does).
    static const bool no_reorder = getenv("SWARMORB_BA_NO_REORDER") != nullptr;
    constexpr int kReorderMinFree = 9 * 16, kLinkSpan = 48;
    if (nf >= kReorderMinFree && !no_reorder) {
        std::vector<uint8_t> back((size_t)nf, 0);
        std::vector<int> eh;  // hessian index of the keyframe of every (landmark-sorted) edge, filled for the full walks
        auto eh_at = [&](int k) { return eh.empty() ? h_hidx[p->edge_pose[perm[(size_t)k]]] : eh[(size_t)k]; };
        const int rparts = nE >= 150000 ? 4 : nE >= 30000 ? 3 : 1;  // threads of the full walks (each with its own flags, merged after)
        auto mark_far = [&](int stride) {  // every stride-th landmark
            const int np = stride == 1 ? rparts : 1;
            std::vector<std::vector<uint8_t>> mine((size_t)np, std::vector<uint8_t>(np > 1 ? (size_t)nf : 0, 0));
            b->staging.run(np, [&](int t, int) {
                uint8_t* flag = np > 1 ? mine[(size_t)t].data() : back.data();
                const int l0 = (int)((long long)nL * t / np), l1 = (int)((long long)nL * (t + 1) / np);
                for (int l = l0; l < l1; l += stride) {
                    int lo = INT_MAX;
                    for (int k = h_ptoff[l]; k < h_ptoff[l + 1]; k++) {
                        const int h = eh_at(k);
                        if (h >= 0 && h < lo) lo = h;
                    }
                    if (lo == INT_MAX) continue;
                    for (int k = h_ptoff[l]; k < h_ptoff[l + 1]; k++) {
                        const int h = eh_at(k);
                        if (h >= 0 && h - lo > kLinkSpan) flag[(size_t)h] = 1;
                    }
                }
            });
            int n = 0;
            for (int h = 0; h < nf; h++) {
                for (int t = 0; t < (np > 1 ? np : 0); t++) back[(size_t)h] |= mine[(size_t)t][(size_t)h];
                n += back[(size_t)h];
            }
            return n;
        };
        // a sample of the landmarks first: a map in which everything sees everything (most keyframes far from some
        // co-observer) keeps its natural order and is spared the full walks over the observations
        int n_far = mark_far(16);
        const bool dense_map = 10 * n_far >= 7 * nf;
        if (dense_map) std::fill(back.begin(), back.end(), 0);
        else {
            eh.resize((size_t)nE);
            b->staging.run(rparts, [&](int t, int np) {
                const int k0 = (int)((long long)nE * t / np), k1 = (int)((long long)nE * (t + 1) / np);
                for (int k = k0; k < k1; k++) eh[(size_t)k] = h_hidx[p->edge_pose[perm[(size_t)k]]];
            });
            n_far = mark_far(1);
        }
        // The bands that remain must also start on tile boundaries (16 keyframes): a tile that holds the end of one
        // agent's band and the start of the next couples their chains again.  Where no landmark bridges two neighbours
        // of the remaining order (a band ends), the last keyframes of the band - fewer than 16 - follow to the back so
        // that the next band starts a new tile.
        if (!dense_map && n_far > 0 && 10 * n_far < 7 * nf) {
            std::vector<int> pos((size_t)nf, -1);  // position in the remaining order
            int ni = 0;
            for (int h = 0; h < nf; h++)
                if (!back[(size_t)h]) pos[(size_t)h] = ni++;
            std::vector<int> bridge((size_t)ni + 2, 0);  // difference array: a landmark seen from positions lo..hi bridges lo+1..hi
            std::vector<std::vector<int>> bparts((size_t)rparts, std::vector<int>((size_t)ni + 2, 0));
            b->staging.run(rparts, [&](int t, int np) {
                std::vector<int>& br = bparts[(size_t)t];
                const int l0 = (int)((long long)nL * t / np), l1 = (int)((long long)nL * (t + 1) / np);
                for (int l = l0; l < l1; l++) {
                    int lo = INT_MAX, hi = -1;
                    for (int k = h_ptoff[l]; k < h_ptoff[l + 1]; k++) {
                        const int h = eh_at(k);
                        if (h < 0 || back[(size_t)h]) continue;
                        lo = std::min(lo, pos[(size_t)h]);
                        hi = std::max(hi, pos[(size_t)h]);
                    }
                    if (hi > lo) {
                        br[(size_t)lo + 1]++;
                        br[(size_t)hi + 1]--;
                    }
                }
            });
            for (int t = 0; t < rparts; t++)
                for (int q = 0; q < ni + 2; q++) bridge[(size_t)q] += bparts[(size_t)t][(size_t)q];
            std::vector<int> at((size_t)ni, 0);  // keyframe (natural hidx) at each remaining position
            for (int h = 0; h < nf; h++)
                if (pos[(size_t)h] >= 0) at[(size_t)pos[(size_t)h]] = h;
            int run = 0, kept = 0;  // bridges open at the current position; keyframes kept so far
            for (int q = 0; q < ni; q++) {
                run += bridge[(size_t)q];
                if (q > 0 && run == 0) {  // nothing connects position q to anything before it: a band ended at q - 1
                    const int cut = kept % 16;
                    for (int c = 1; c <= cut; c++) back[(size_t)at[(size_t)(q - c)]] = 1;  // (cut < band length or the band goes whole)
                    kept -= cut;
                }
                kept++;
            }
        }
        if (!dense_map && 10 * n_far >= 7 * nf) std::fill(back.begin(), back.end(), 0);
        int n_back = 0;
        for (int h = 0; h < nf; h++) n_back += back[(size_t)h];
        if (n_back > 0 && n_back < nf) {
            std::vector<int> old_free(h_free, h_free + nf);
            int h = 0;
            for (int pass = 0; pass < 2; pass++)
                for (int o = 0; o < nf; o++)
                    if ((int)back[(size_t)o] == pass) {
                        const int i = old_free[(size_t)o];
                        h_hidx[i] = h;
                        h_free[h++] = i;
                    }
        }
        r.n_reordered = n_back;
    }
    const double tS4 = now_ms();
    // the observations in landmark order + the per-keyframe lists; big maps on `parts` threads, each owning a range of
    // landmarks (so a landmark's observations - and the duplicate check - stay with one thread)
    // (SWARMORB_BA_PARTS overrides.  On a 21 k-edge window, LBA-M's size, alone on the box this part of the staging is
    //  0.133 ms on one thread, 0.094 on two, 0.075 on four - but in the bench, next to a tracking thread and the matcher
    //  job, four threads LOSE: windows 1.40-1.62 ms against 1.44-1.46, the keyframe's matcher job 0.90-0.98 ms against
    //  0.73-0.82, 2.06 k frames/s against 2.22 k over three runs each: the wake-ups cost the neighbours more than the
    //  shorter fill returns.  Threads from 30 k observations on, as before.)
    static const int parts_env = getenv("SWARMORB_BA_PARTS") ? atoi(getenv("SWARMORB_BA_PARTS")) : 0;
    const int parts = parts_env > 0 ? std::min(parts_env, 16) : nE >= 150000 ? 4 : nE >= 30000 ? 3 : 1;
    std::vector<int> part_lm((size_t)parts + 1, 0);  // landmark ranges with about equal numbers of observations
    for (int t = 1; t < parts; t++) {
        const int target = (int)((long long)nE * t / parts);
        part_lm[(size_t)t] = (int)(std::lower_bound(h_ptoff, h_ptoff + nL + 1, target) - h_ptoff);
        part_lm[(size_t)t] = std::min(std::max(part_lm[(size_t)t], part_lm[(size_t)t - 1]), nL);
    }
    part_lm[(size_t)parts] = nL;
    std::vector<std::vector<int>> part_cnt((size_t)parts, std::vector<int>((size_t)nf + 1, 0));  // observations per free keyframe and part
    std::atomic<int> dup{0};
    b->staging.run(parts, [&](int t, int) {
        std::vector<int> seen((size_t)nP, -1);  // the edge table holds one edge per (landmark, keyframe)
        std::vector<int>& cnt = part_cnt[(size_t)t];
        const int k0 = h_ptoff[part_lm[(size_t)t]], k1 = h_ptoff[part_lm[(size_t)t + 1]];
        for (int k = k0; k < k1; k++) {
            const int e = perm[(size_t)k], ip = p->edge_pose[e], il = p->edge_point[e];
            h_epose[k] = ip;
            h_ept[k] = il;
            h_obs[2 * (size_t)k] = (double)p->obs[2 * (size_t)e];
            h_obs[2 * (size_t)k + 1] = (double)p->obs[2 * (size_t)e + 1];
            h_w[k] = (double)p->inv_sigma2[e];
            h_eact[k] = 1;
            if (seen[(size_t)ip] == il) dup.store(1);
            seen[(size_t)ip] = il;
            if (h_hidx[ip] >= 0) cnt[(size_t)h_hidx[ip]]++;
        }
    });
    if (dup.load()) {
        last_error_ref() = "a landmark is observed twice by the same keyframe";
        return SO_ERR_INVALID_ARG;
    }
    h_poseoff[0] = 0;
    for (int i = 0; i < nf; i++) {
        int c = 0;
        for (int t = 0; t < parts; t++) c += part_cnt[(size_t)t][(size_t)i];
        h_poseoff[i + 1] = h_poseoff[i] + c;
    }
    // free keyframe -> its edges, ascending (= by landmark): part t starts behind what the parts before it hold
    for (int i = 0; i < nf; i++) {
        int at = h_poseoff[i];
        for (int t = 0; t < parts; t++) {
            const int c = part_cnt[(size_t)t][(size_t)i];
            part_cnt[(size_t)t][(size_t)i] = at;
            at += c;
        }
    }
    b->staging.run(parts, [&](int t, int) {
        std::vector<int>& fill = part_cnt[(size_t)t];
        const int k0 = h_ptoff[part_lm[(size_t)t]], k1 = h_ptoff[part_lm[(size_t)t + 1]];
        for (int k = k0; k < k1; k++) {
            const int h = h_hidx[h_epose[k]];
            if (h >= 0) {
                h_pepoint[fill[(size_t)h]] = h_ept[k];  // the edge's landmark next to it: one hop less in the Schur gather
                h_pedges[fill[(size_t)h]++] = k;
            }
        }
    });
    const double tS5 = now_ms();
    for (int i = 0; i < nL; i++) h_ptact[i] = h_ptoff[i + 1] > h_ptoff[i];
    static const bool trace = getenv("SWARMORB_BA_TRACE") != nullptr;
    const double t_staged = now_ms();

    if (!b->stream) SO_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    hipStream_t s = b->stream;
    // A member of a so_ba_group uploads on the group's stream (stream order puts the upload in front of its chain; one busy stream
    // fewer per agent: INTEGRATION.md 3e).  SWARMORB_BA_GROUP_OWN_UPLOAD=1: on the member's own stream, behind an event.
    static const bool own_upload = getenv("SWARMORB_BA_GROUP_OWN_UPLOAD") != nullptr;
    // (the same conditions as `grouped` below: a local window - at most 43 free keyframes, hence neither the blocked nor the
    //  pair-list path - on the single-enqueue path)
    const bool upload_on_group = b->group != nullptr && !own_upload && !b->solve_timing && getenv("SWARMORB_BA_NO_CHAIN") == nullptr &&
                                 nf <= kBaSmallSolverMaxFree && (nf + (nE > 0 ? 1 : 0)) != 0;
    if (upload_on_group) s = b->group->stream;
    SO_HIP(hipMemcpyAsync(b->d_in.p, b->h_in, L.total, hipMemcpyHostToDevice, s));
    const size_t sE = (size_t)nE, sL = (size_t)std::max(nL, 1), sF = (size_t)std::max(nf, 1), n = 6 * sF;
    if ((rc = b->d_pose1.ensure(sizeof(BaPose) * (size_t)std::max(nP, 1)))) return rc;
    if ((rc = b->d_pt1.ensure(sizeof(double) * 3 * sL))) return rc;
    if ((rc = b->d_err.ensure(sizeof(double) * 2 * sE))) return rc;
    if ((rc = b->d_chi2.ensure(sizeof(double) * sE))) return rc;
    // single-workgroup solvers up to 43 free keyframes, the blocked solver (S padded to a multiple of 96) beyond; from
    // kBaPairsMinFree free keyframes on the Schur gather walks per-block pair lists and needs no edge table
    const bool dense_path = nf > kBaSmallSolverMaxFree;
    static const int pairs_min_env = getenv("SWARMORB_BA_PAIRS_MIN") ? atoi(getenv("SWARMORB_BA_PAIRS_MIN")) : 0;
    const bool pairs_path = dense_path && nf >= (pairs_min_env > 0 ? pairs_min_env : kBaPairsMinFree);
    if (!pairs_path && (rc = b->d_tab.ensure(sizeof(int) * sL * sF))) return rc;
    if ((rc = b->d_Hll.ensure(sizeof(double) * 9 * sL))) return rc;
    if ((rc = b->d_bl.ensure(sizeof(double) * 3 * sL))) return rc;
    if ((rc = b->d_Dinv.ensure(sizeof(double) * 9 * sL))) return rc;
    if ((rc = b->d_db.ensure(sizeof(double) * 3 * sL))) return rc;
    if ((rc = b->d_xl.ensure(sizeof(double) * 3 * sL))) return rc;
    if ((rc = b->d_W.ensure(sizeof(double) * 18 * sE))) return rc;
    if ((rc = b->d_BDinv.ensure(sizeof(double) * 18 * sE))) return rc;
    if ((rc = b->d_partial.ensure(sizeof(double) * kBaPartialCount))) return rc;
    if ((rc = b->d_Hpp.ensure(sizeof(double) * 36 * sF))) return rc;
    if ((rc = b->d_bp.ensure(sizeof(double) * 6 * sF))) return rc;
    const size_t ldS = dense_path ? (n + 95) / 96 * 96 : n;
    if ((rc = b->d_S.ensure(sizeof(double) * ldS * ldS))) return rc;
    if ((rc = b->d_bs.ensure(sizeof(double) * ldS))) return rc;
    if (dense_path) {
        if (!b->dense_side) {
            int lo_pri = 0, hi_pri = 0;  // the side stream carries the serial chain: highest priority
            if (hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri) != hipSuccess) lo_pri = hi_pri = 0;
            SO_HIP(hipStreamCreateWithPriority(&b->dense_side, hipStreamNonBlocking, hi_pri));
            b->dense_events.assign(1 + 2 * kDenseMaxPanels, nullptr);
            for (hipEvent_t& e : b->dense_events) SO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if ((rc = b->d_dense_ws.ensure(sizeof(double) * (ldS / 96) * 96 * 96))) return rc;
        if ((rc = b->d_dense_x.ensure(sizeof(double) * ldS))) return rc;
        // Block skyline of S (LinearSolverEigen only ever touches the structural nonzeros, linear_solver_eigen.h:
        // 147-232; here: 96-row tiles in natural keyframe order).  Block (i1, i2) of S is nonzero iff a landmark is seen
        // by both keyframes, so row tile I starts at the smallest tile of any keyframe sharing a landmark with one of
        // I's keyframes; Cholesky fill stays inside that row envelope.
        const int T = (int)(ldS / 96);
        b->tile_first.resize((size_t)T);
        // (a local window of up to 128 keyframes - 8 tiles - is taken as dense: its keyframes are covisible by
        // construction and the walk over the observations costs more than the few tiles it could drop)
        constexpr int kSkylineMinTiles = 9;
        for (int I = 0; I < T; I++) b->tile_first[(size_t)I] = T >= kSkylineMinTiles ? I : 0;
        for (int l = 0; T >= kSkylineMinTiles && l < nL; l++) {
            int lo = INT_MAX;
            for (int k = h_ptoff[l]; k < h_ptoff[l + 1]; k++) {
                const int h = h_hidx[h_epose[k]];
                if (h >= 0 && h < lo) lo = h;
            }
            if (lo == INT_MAX) continue;
            const int tlo = lo / 16;  // 16 keyframes (96 rows) per tile
            for (int k = h_ptoff[l]; k < h_ptoff[l + 1]; k++) {
                const int h = h_hidx[h_epose[k]];
                if (h >= 0 && tlo < b->tile_first[(size_t)(h / 16)]) b->tile_first[(size_t)(h / 16)] = tlo;
            }
        }
        static const bool force_dense = getenv("SWARMORB_DENSE_FULL") != nullptr;  // A/B: ignore the structure
        if (force_dense) std::fill(b->tile_first.begin(), b->tile_first.end(), 0);
        // the dataflow solves need their workgroups resident: reserve the CUs first, the plan is built for that many
        {
            long long nnz = 0;
            for (int I = 0; I < T; I++) nnz += I - b->tile_first[(size_t)I] + 1;
            const int budget = flow_resident_budget(b->device);
            static const bool no_flow = getenv("SWARMORB_DENSE_NO_FLOW") != nullptr, no_flow_big = getenv("SWARMORB_DENSE_NO_FLOW_BIG") != nullptr;
            static const int flow_max = getenv("SWARMORB_DENSE_FLOW_MAX_TILES") ? atoi(getenv("SWARMORB_DENSE_FLOW_MAX_TILES")) : kFlowDefaultMaxTiles;
            const bool big = !(T <= 21 && nnz <= std::min(flow_max, dense_flow_max_tiles()));
            if (no_flow || b->flow_broken || (big && no_flow_big)) {
            } else if (!big) {  // a workgroup per tile: all or nothing
                const int want = (int)nnz;
                int cur = g_flow_tiles.load();
                while (cur + want <= budget && !g_flow_tiles.compare_exchange_weak(cur, cur + want)) {
                }
                if (cur + want <= budget) b->flow_reserved = want;
            } else {  // ticketed kernel: whatever is free, if that is worth it
                int cur = g_flow_tiles.load(), take = 0;
                do {
                    take = (int)std::min<long long>(nnz, budget - cur);
                } while (take >= 128 && !g_flow_tiles.compare_exchange_weak(cur, cur + take));
                if (take >= 128) b->flow_reserved = take;  // (the early tickets of build_dense_plan need room: see there)
            }
        }
        build_dense_plan(T, b->tile_first.data(), b->dense_side != nullptr, &b->plan, b->flow_reserved);
        const size_t first_bytes = (sizeof(int) * (size_t)T + 255) & ~(size_t)255;
        const size_t tiles_bytes = sizeof(int2) * std::max<size_t>(b->plan.tiles.size(), 1);
        if ((rc = b->d_plan.ensure(first_bytes + tiles_bytes))) return rc;
        // one copy from pinned memory, no wait: the call does not return before the stream has drained
        if ((rc = ensure_pinned(&b->h_plan, &b->h_plan_cap, first_bytes + tiles_bytes))) return rc;
        memcpy(b->h_plan, b->tile_first.data(), sizeof(int) * (size_t)T);
        if (!b->plan.tiles.empty()) memcpy((uint8_t*)b->h_plan + first_bytes, b->plan.tiles.data(), sizeof(int2) * b->plan.tiles.size());
        SO_HIP(hipMemcpyAsync(b->d_plan.p, b->h_plan, first_bytes + sizeof(int2) * b->plan.tiles.size(), hipMemcpyHostToDevice, s));
        if (b->flow_reserved > 0) {
            // small kernel: fixed flag block + the forward vectors; ticketed kernel: a flag per tile slot + 2 x 512 + counters
            const size_t nslots = b->plan.flow_big ? (size_t)T * (T + 1) / 2 : 0;
            const size_t bytes = b->plan.flow_big ? sizeof(unsigned) * (nslots + 2 * kDenseMaxPanels + 8)
                                                  : sizeof(unsigned) * kFlowFlagWords + sizeof(double) * 256 * 96;
            Buf& fb = b->plan.flow_big ? b->d_flow_big : b->d_flow;
            if (bytes > fb.cap) {  // stamps of earlier solves live here: a fresh block starts from zero
                if ((rc = fb.ensure(bytes))) return rc;
                SO_HIP(hipMemsetAsync(fb.p, 0, fb.cap, s));
            }
        }
    }
    const double t_dense_setup = now_ms();
    // pair lists of the large-map gather: capacity from the landmarks' observation counts (an upper bound: fixed
    // keyframes' observations are counted too)
    size_t pair_cap = 0, n_blk = (size_t)nf * ((size_t)nf + 1) / 2, big_cap = 0, scan_bytes = 0;
    if (pairs_path) {
        for (int i = 0; i < nL; i++) {
            const size_t k = (size_t)(h_ptoff[i + 1] - h_ptoff[i]);
            pair_cap += k * (k - (k > 0 ? 1 : 0)) / 2;
        }
        big_cap = (size_t)nf + std::min(n_blk - (size_t)nf, pair_cap / (kBaSmallBlockPairs + 1));
        scan_bytes = ba_pairs_scan_temp_bytes((int)n_blk);
        if ((rc = b->d_pr_off.ensure(sizeof(int) * (n_blk + 1)))) return rc;
        if ((rc = b->d_pr_cur.ensure(sizeof(int) * (n_blk + 1)))) return rc;
        if ((rc = b->d_pr.ensure(sizeof(int) * 6 * std::max<size_t>(pair_cap, 1)))) return rc;
        if ((rc = b->d_big.ensure(sizeof(int) * (big_cap + 1)))) return rc;
        if ((rc = b->d_scan_tmp.ensure(std::max<size_t>(scan_bytes, 16)))) return rc;
    }
    if ((rc = b->d_lm.ensure(sizeof(BaLm)))) return rc;
    Layout O;  // result block
    const size_t r_pose = O.add(sizeof(BaPose) * (size_t)nP), r_pt = O.add(sizeof(double) * 3 * (size_t)nL),
                 r_chi2 = O.add(sizeof(double) * sE), r_out = O.add(sE);
    if ((rc = ensure_pinned(&b->h_out, &b->h_out_cap, O.total))) return rc;
    if ((rc = b->d_out.ensure(O.total))) return rc;
    {
        BaClearList cl{};
        cl.item[cl.n++] = {b->d_err.p, sizeof(double) * 2 * sE, 0u};   // _error of a fresh edge
        cl.item[cl.n++] = {b->d_chi2.p, sizeof(double) * sE, 0u};
        cl.item[cl.n++] = {b->d_partial.p, sizeof(double) * kBaPartialCount, 0u};
        if (!pairs_path) cl.item[cl.n++] = {b->d_tab.p, sizeof(int) * sL * sF, 0xFFFFFFFFu};  // -1: keyframe does not observe the landmark
        static_assert(sizeof(BaLm) % 4 == 0, "cleared as 32-bit words");
        cl.item[cl.n++] = {b->d_lm.p, sizeof(BaLm), 0u};  // current estimate = buffer 0, no trials yet
        launch_ba_clear(cl, s);
    }
    memset(b->h_lm, 0, sizeof(BaLm));
    *b->h_abort = 0;

    BaDev& d = r.d;
    uint8_t* db = (uint8_t*)b->d_in.p;
    d.n_poses = nP;
    d.n_points = nL;
    d.n_edges = nE;
    d.n_free = nf;
    d.lm = b->d_lm.as<BaLm>();
    d.pose[0] = (BaPose*)(db + o_pose);
    d.pose[1] = b->d_pose1.as<BaPose>();
    d.pt[0] = (double*)(db + o_pt);
    d.pt[1] = b->d_pt1.as<double>();
    d.intr = (const double*)(db + o_intr);
    d.e_pose = (const int*)(db + o_epose);
    d.e_point = (const int*)(db + o_ept);
    d.e_obs = (const double*)(db + o_obs);
    d.e_w = (const double*)(db + o_w);
    d.e_active = db + o_eact;
    d.pt_active = db + o_ptact;
    d.pt_off = (const int*)(db + o_ptoff);
    d.pose_hidx = (const int*)(db + o_hidx);
    d.free_pose = (const int*)(db + o_free);
    d.pose_off = (const int*)(db + o_poseoff);
    d.pose_edges = (const int*)(db + o_pedges);
    d.pose_edge_point = (const int*)(db + o_pepoint);
    d.edge_tab = b->d_tab.as<int>();
    d.e_err = b->d_err.as<double>();
    d.e_chi2 = b->d_chi2.as<double>();
    d.Hpp = b->d_Hpp.as<double>();
    d.bp = b->d_bp.as<double>();
    d.Hll = b->d_Hll.as<double>();
    d.bl = b->d_bl.as<double>();
    d.W = b->d_W.as<double>();
    d.Dinv = b->d_Dinv.as<double>();
    d.db = b->d_db.as<double>();
    d.BDinv = b->d_BDinv.as<double>();
    d.S = b->d_S.as<double>();
    d.bs = b->d_bs.as<double>();
    d.ldS = (int)ldS;
    d.dense_ws = b->d_dense_ws.as<double>();
    d.dense_x = b->d_dense_x.as<double>();
    d.use_pairs = pairs_path ? 1 : 0;
    d.fold_prep = (!pairs_path && !dense_path) ? 1 : 0;  // measured: beyond 43 free keyframes the stored products win (r4_lba_sweep.txt)
    d.pr_off = b->d_pr_off.as<int>();
    d.pr_cur = b->d_pr_cur.as<int>();
    d.pr_l = b->d_pr.as<int>();
    d.pr_k1 = d.pr_l + std::max<size_t>(pair_cap, 1);
    d.pr_k2 = d.pr_k1 + std::max<size_t>(pair_cap, 1);
    d.ps_l = d.pr_k2 + std::max<size_t>(pair_cap, 1);
    d.ps_k1 = d.ps_l + std::max<size_t>(pair_cap, 1);
    d.ps_k2 = d.ps_k1 + std::max<size_t>(pair_cap, 1);
    d.big_list = b->d_big.as<int>();
    d.big_n = d.big_list ? d.big_list + big_cap : nullptr;
    d.big_cap = (int)big_cap;
    d.dense_side = dense_path ? b->dense_side : nullptr;
    d.dense_events = dense_path ? b->dense_events.data() : nullptr;
    d.tile_first = dense_path ? b->d_plan.as<int>() : nullptr;
    d.plan_tiles = dense_path ? reinterpret_cast<const int2*>((const uint8_t*)b->d_plan.p + ((sizeof(int) * (ldS / 96) + 255) & ~(size_t)255)) : nullptr;
    d.plan = dense_path ? &b->plan : nullptr;
    const bool flow = dense_path && b->flow_reserved > 0;
    d.flow_tiles = flow ? d.plan_tiles + b->plan.flow_first_tile : nullptr;
    const bool flow_big = flow && b->plan.flow_big;
    d.flow_flags = !flow ? nullptr : flow_big ? b->d_flow_big.as<unsigned>() : b->d_flow.as<unsigned>();
    d.flow_vec = flow && !flow_big ? reinterpret_cast<double*>(b->d_flow.as<unsigned>() + kFlowFlagWords) : nullptr;
    d.flow_epoch = &b->flow_epoch;
    d.flow_abort_host = b->h_flow_abort_dev;
    {
        static const double timeout_ms = getenv("SWARMORB_FLOW_TIMEOUT_MS") ? atof(getenv("SWARMORB_FLOW_TIMEOUT_MS")) : 500.0;
        d.flow_timeout_ticks = (unsigned long long)(timeout_ms * 1e5);  // wall_clock64() counts at 100 MHz
    }
    d.flow_nslots = flow_big ? (int)((ldS / 96) * (ldS / 96 + 1) / 2) : 0;
    d.flow_grid = flow_big ? b->flow_reserved : 0;
    d.xl = b->d_xl.as<double>();
    d.partial = b->d_partial.as<double>();
    d.robust = opt->robust;
    d.stage = 1;
    d.huber_delta = (double)opt->huber_delta;
    d.huber_dsqr = (float)((double)opt->huber_delta * (double)opt->huber_delta);  // RobustKernelHuber::setDelta
    r.nb_err = std::min(1024, std::max(1, (nE + 255) / 256));
    r.nb_upd = std::min(1024, std::max(1, (8 * nL + nP + 255) / 256));
    if (!pairs_path) launch_ba_edge_table(d, s);
    if (dense_path) launch_ba_dense_pad(d, s);
    if (pairs_path) launch_ba_build_pairs(d, b->d_scan_tmp.p, scan_bytes, s);
    d.use_pcg = 0;
    d.pcg_host = nullptr;
    b->pcg.iterations = 0;
    b->pcg.solves = 0;
    b->pcg.fault = 0;
    if (pairs_path && b->linear_solver == 1) {
        // block-Jacobi PCG instead of the blocked Cholesky: the 6 x 6 block structure of S from the pair lists (once per
        // problem: counts, scan, one look at the total, fill), then the workspace
        const size_t sNF = (size_t)nf, nA = sNF, nB = (sNF + 255) / 256;  // (partA: a partial per block row at most)
        Layout P;
        const size_t p_counts = P.add(sizeof(int) * (sNF + 1)), p_indptr = P.add(sizeof(int) * (sNF + 1)), p_Minv = P.add(sizeof(double) * 36 * sNF),
                     p_x = P.add(sizeof(double) * 6 * sNF), p_r = P.add(sizeof(double) * 6 * sNF), p_z = P.add(sizeof(double) * 6 * sNF),
                     p_p = P.add(sizeof(double) * 6 * sNF), p_Sp = P.add(sizeof(double) * 6 * sNF), p_A = P.add(sizeof(double) * nA),
                     p_B = P.add(sizeof(double) * 2 * nB), p_scal = P.add(sizeof(double) * 8), p_status = P.add(sizeof(unsigned long long));
        if ((rc = b->d_pcg.ensure(P.total))) return rc;
        uint8_t* pb = (uint8_t*)b->d_pcg.p;
        int* d_counts = (int*)(pb + p_counts);
        int* d_indptr = (int*)(pb + p_indptr);
        launch_ba_pcg_structure(d, d_counts, d_indptr, nullptr, s);
        int nnzb = 0;
        SO_HIP(hipMemcpyAsync(&nnzb, d_indptr + nf, sizeof(int), hipMemcpyDeviceToHost, s));
        SO_HIP(hipStreamSynchronize(s));
        if ((rc = b->d_pcg_idx.ensure(sizeof(int) * (size_t)std::max(nnzb, 1)))) return rc;
        launch_ba_pcg_structure(d, d_counts, d_indptr, b->d_pcg_idx.as<int>(), s);
        BaPcgDev& q = b->pcg.dev;
        q.indptr = d_indptr;
        q.indices = b->d_pcg_idx.as<int>();
        // the blocks' values side by side for the products (SWARMORB_PCG_DENSE_READ=1: read them where they lie in S, as before round 6)
        static const bool dense_read = getenv("SWARMORB_PCG_DENSE_READ") != nullptr;
        q.Sc = nullptr;
        if (!dense_read) {
            if ((rc = b->d_pcg_val.ensure(sizeof(double) * 36 * (size_t)std::max(nnzb, 1)))) return rc;
            q.Sc = b->d_pcg_val.as<double>();
        }
        q.Minv = (double*)(pb + p_Minv);
        q.x = (double*)(pb + p_x); q.r = (double*)(pb + p_r); q.z = (double*)(pb + p_z); q.p = (double*)(pb + p_p); q.Sp = (double*)(pb + p_Sp);
        q.partA = (double*)(pb + p_A); q.partB = (double*)(pb + p_B); q.scal = (double*)(pb + p_scal);
        q.status = (unsigned long long*)(pb + p_status);
        // (the status word shares the context's host-mapped block with the dataflow solves' abort word: bytes 8..15)
        q.status_host = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(b->h_flow_abort_dev) + 8);
        b->pcg.status_host = reinterpret_cast<volatile unsigned long long*>(reinterpret_cast<uint8_t*>(b->h_flow_abort) + 8);
        b->pcg_nnz_blocks = nnzb;
        // a workgroup per block row from 20 blocks per row on (measured: GBA-2, 516 per row, 7.5 -> 4.4 ms per solve; GBA-2r, 51 per
        // row, 13.0 -> 12.3; GBA-1r, 37 per row, 6.0 -> 5.4); SWARMORB_PCG_WIDE=0/1 forces one or the other
        static const int wide_env = getenv("SWARMORB_PCG_WIDE") ? atoi(getenv("SWARMORB_PCG_WIDE")) : -1;
        b->pcg.wide = wide_env >= 0 ? (wide_env != 0) : ((long long)nnzb >= 20ll * nf);
        d.use_pcg = 1;
        d.pcg_host = &b->pcg;
    }

    const double t_uploaded = now_ms();
    if (trace) fprintf(stderr, "[ba] %d free keyframes, %d moved to the separator block\n", nf, r.n_reordered);
    if (trace) fprintf(stderr, "[ba] staging: before %.3f touched+layout %.3f convert %.3f sort %.3f order %.3f fill %.3f lists %.3f\n",
                       tS0 - t_begin, tS1 - tS0, tS2 - tS1, tS3 - tS2, tS4 - tS3, tS5 - tS4, t_staged - tS5);
    if (trace) fprintf(stderr, "[ba] dense setup done at +%.3f, launches done at +%.3f\n", t_dense_setup - t_staged, t_uploaded - t_staged);
    // A member of a so_ba_group with a local window: the chain below is recorded, not launched, and goes out merged with the
    // other members' on the group's stream (ba_group_submit); this member's uploads above stay on its own stream.
    const bool grouped = b->group != nullptr && !b->solve_timing && getenv("SWARMORB_BA_NO_CHAIN") == nullptr && !dense_path &&
                         !pairs_path && (r.n_free + (r.n_active_edges > 0 ? 1 : 0)) != 0;
    struct RecorderScope {  // (whatever way this function is left, the thread's launches are real again)
        ~RecorderScope() { g_ba_recorder = nullptr; }
    } recorder_scope;
    if (grouped) {
        if (!b->grp_uploaded) SO_HIP(hipEventCreateWithFlags(&b->grp_uploaded, hipEventDisableTiming));
        SO_HIP(hipEventRecord(b->grp_uploaded, s));  // (on the group's stream already unless SWARMORB_BA_GROUP_OWN_UPLOAD: then the round waits for it)
        s = b->group->stream;  // (until the round is issued: nothing is launched meanwhile, the chain is recorded)
        b->rec.list.clear();
        b->rec.phase = 0;
        g_ba_recorder = &b->rec;
    }
    SO_HIP(hipEventRecord(b->e0, s));
    double chi = 0.0;
    int done = 0;
    // Optimizer.cc:682-739: outlier flags from the edges' stored errors and isDepthPositive(), optimised data back.
    // The finish kernel writes the result block straight into pinned host memory (0.65 MB for a 64-keyframe window): no
    // copy engine behind it - a device-to-host copy enqueued behind a long chain of kernels was seen to start hundreds of
    // microseconds after the kernel in front of it.
    uint8_t* ob = nullptr;
    SO_HIP(hipHostGetDevicePointer((void**)&ob, b->h_out, 0));
    auto epilogue = [&]() {
        b->rec.phase = kBaPhaseEpilogue;
        launch_ba_finish(r.d, (double)opt->chi2_threshold, (BaPose*)(ob + r_pose), (double*)(ob + r_pt), (double*)(ob + r_chi2),
                         ob + r_out, s);
        if (!g_ba_recorder) (void)hipEventRecord(b->e1, s);
    };
    // Completion words (single-enqueue path): the host spins on a word in host-mapped memory instead of polling the stream
    // (hipStreamQuery + 20 us naps saw the end of a window tens of microseconds late), and the call's results can be
    // picked up BEFORE the launches enqueued as a reserve have drained: [stage 2's trials as the last call needed them]
    // [finish + word 1] [two more trials] [finish + word 2].  Word 1 with the LM state "over": the reserve trials return
    // at once (they drain under the caller's write-back), the second finish rewrites the same bytes.
    if (++b->done_seq >= (1 << 20)) b->done_seq = 1;
    int* done_word_dev = reinterpret_cast<int*>(b->h_abort_dev + 16);
    volatile int* done_word = reinterpret_cast<volatile int*>(b->h_abort + 16);
    *done_word = 0;
    int phases = 0, early_phase = 0;
    auto signal = [&]() { launch_ba_signal(done_word_dev, (b->done_seq << 8) | ++phases, s); };
    auto wait_word = [&](int phase) -> int {
        const int want_seq = b->done_seq;
        for (unsigned long it = 1;; it++) {
            const int w = *done_word;
            if ((w >> 8) == want_seq && (w & 255) >= phase) break;
            if (r.stop) *b->h_abort = *r.stop ? 1 : 0;  // forceStopFlag forwarded to the decision kernels
            if ((it & 0xfff) == 0) {  // every ~0.1 ms: has the stream died or drained without the word?
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
                if (it > 0x40000) std::this_thread::sleep_for(std::chrono::microseconds(20));  // a global map: seconds, not a window
            }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        return SO_OK;
    };
    const bool two_stages = opt->its_stage2 > 0;
    const uint8_t* abort_dev = r.stop ? b->h_abort_dev : nullptr;
    BaDev d2 = r.d;
    d2.robust = 0;  // e->setRobustKernel(nullptr)
    d2.stage = 2;
    static const bool no_chain = getenv("SWARMORB_BA_NO_CHAIN") != nullptr;
    const bool have_work = r.n_free + (r.n_active_edges > 0 ? 1 : 0) != 0;
    double t_opt1 = now_ms();
    if (have_work && !b->solve_timing && !no_chain) {
        // The whole call as ONE enqueue in the usual case (no trial rejected, no stop request): stage 1's prologue and
        // trials, stage 2's prologue gated on "stage 1 is over" (kBaGateIdle), stage 2's trials - launches carry their
        // stage's tag and return at once unless lm->active equals it -, finish + copy out.  The host looks once, at the
        // end; a stream synchronise + the next launches idled the GPU ~35 us at the stage boundary and ~45 us at the end.
        // If a stage needs more trials than its iteration count (rejected trials), what was enqueued behind it has
        // returned at once; the host adds the missing trials and the rest of the chain again.
        // The stage's trials as ONE resident launch where the window allows it (ba_kernels.hip): it ends by itself when the stage
        // is over, so all of a stage's trials - rejected ones included - are "enqueued" at once and nothing is held in reserve.
        // (an experiment's switches, read per call so that the tests can flip them: NOTES G.10)
        const bool resident_env = getenv("SWARMORB_BA_RESIDENT") && atoi(getenv("SWARMORB_BA_RESIDENT")) != 0;
        const int resident_wgs = getenv("SWARMORB_BA_RESIDENT_WGS") ? atoi(getenv("SWARMORB_BA_RESIDENT_WGS")) : 32;
        const int resident_max = getenv("SWARMORB_BA_RESIDENT_MAX") ? atoi(getenv("SWARMORB_BA_RESIDENT_MAX")) : 6;
        ResidentSlot resident_slot;
        bool resident = resident_env && !grouped && !b->flow_broken && !r.d.use_pcg && !r.d.plan && resident_slot.take(b->device, resident_max);
        if (resident && !b->d_rsync) {
            if (hipMalloc((void**)&b->d_rsync, sizeof(BaResidentSync)) != hipSuccess || so::memset_sync(b->d_rsync, 0, sizeof(BaResidentSync)) != hipSuccess) {
                (void)hipGetLastError();
                resident = false;
            }
        }
        auto trials = [&](const BaDev& d, int n) {
            if (resident) {
                if (++b->resident_epoch >= (1u << 20)) b->resident_epoch = 1;
                b->rec.phase = (d.stage == 2 ? kBaPhaseStage2 : 0) + 1;
                if (launch_ba_trials_resident(d, r.nb_upd, kBaResidentMaxTrials, abort_dev, b->h_lm_dev, b->d_rsync, b->resident_epoch, resident_wgs, s)) {
                    r.blocks_enqueued += n;
                    b->resident_launches++;
                    return;
                }
                resident = false;  // (not a window the resident loop covers: the chain, for the rest of this call)
            }
            for (int i = 0; i < n; i++) {
                r.blocks_enqueued++;
                b->rec.phase = (d.stage == 2 ? kBaPhaseStage2 : 0) + 1 + i;  // (only read while the chain is being recorded)
                launch_ba_trial(d, r.nb_err, r.nb_upd, abort_dev, b->h_lm_dev, nullptr, nullptr, s);
            }
        };
        auto stage2_chained = [&]() {  // Optimizer.cc:644-656 without leaving the device
            b->rec.phase = kBaPhaseStage2;
            launch_ba_mark_outliers(d2, (double)opt->chi2_threshold, kBaGateIdle, s);
            launch_ba_errors(d2, 0, kBaGateIdle, r.nb_err, s);
            launch_ba_build(d2, kBaGateIdle, s);
            launch_ba_stage_begin(d2, r.nb_err, opt->its_stage2, b->h_lm_dev, kBaGateIdle, abort_dev, s);
            // Stage 2 of a window that is re-optimised keyframe after keyframe ends on g2o's convergence test after a few
            // iterations (levenberg.cpp:154-161); every trial enqueued beyond that is six launches that return at once, ~15 us
            // of queue time each.  Enqueue what the context's last call needed plus two; a stage that wants more is
            // topped up by the loop below (one host round trip).
            static const bool no_hint = getenv("SWARMORB_BA_NO_STAGE2_HINT") != nullptr;  // A/B: all iterations ahead, as before round 5
            // (a group member records all of them: the group enqueues what its members' last calls needed + 2, for everybody)
            const int ahead = (no_hint || grouped || resident) ? opt->its_stage2 : std::min(opt->its_stage2, b->stage2_hint > 0 ? b->stage2_hint : opt->its_stage2);
            trials(d2, ahead);
            early_phase = 0;
            static const bool no_early = getenv("SWARMORB_BA_NO_EARLY") != nullptr;  // A/B: the reserve in front of the only epilogue
            if (no_early) trials(d2, std::min(2, opt->its_stage2 - ahead));
            // (not with PCG: its solve keeps the host in a loop per enqueued trial, a reserve would block the return it is for)
            if (ahead < opt->its_stage2 && !no_early && !r.d.use_pcg) {  // results out as soon as the stage is over; two trials in reserve behind them
                launch_ba_finish(r.d, (double)opt->chi2_threshold, (BaPose*)(ob + r_pose), (double*)(ob + r_pt), (double*)(ob + r_chi2),
                                 ob + r_out, s);
                (void)hipEventRecord(b->e1a, s);
                signal();
                early_phase = phases;
                trials(d2, std::min(2, opt->its_stage2 - ahead));
            }
        };
        launch_ba_errors(r.d, 0, kBaGateNone, r.nb_err, s);
        launch_ba_build(r.d, kBaGateNone, s);
        launch_ba_stage_begin(r.d, r.nb_err, opt->its_stage1, b->h_lm_dev, kBaGateNone, nullptr, s);
        trials(r.d, opt->its_stage1);
        if (two_stages) stage2_chained();
        epilogue();
        signal();
        if (grouped) {  // hand the recorded chain in; back when the round it belongs to has been issued on the group's stream
            g_ba_recorder = nullptr;
            if ((rc = ba_group_submit(b))) return rc;
            s = b->grp_stream;  // (the round's stream: what this call still launches or polls goes there)
        }
        BaLm lm;
        bool stopped_between = false, returned_early = false;
        for (;;) {
            SO_HIP(hipGetLastError());
            if (early_phase > 0) {  // the results may be out before the reserve has drained
                if ((rc = wait_word(early_phase))) return rc;
                if (*(volatile unsigned*)b->h_flow_abort != 0) return kErrFlowTimeout;
                memcpy(&lm, b->h_lm, sizeof(lm));
                early_phase = 0;
                if (lm.active == 0 && lm.stages_begun >= 2) {
                    returned_early = true;
                    b->leftover_launches = true;
                    break;
                }
            }
            if ((rc = wait_word(phases))) return rc;
            if (*(volatile unsigned*)b->h_flow_abort != 0) return kErrFlowTimeout;  // a dataflow solve gave up waiting
            if (b->pcg.fault) {  // a PCG solve was abandoned by its host loop (ba_pcg.hip): the trial it fed is not a result
                last_error_ref() = b->pcg.fault == 1 ? "so_bundle_adjust: a PCG solve did not finish within 30 s" : "so_bundle_adjust: the stream failed during a PCG solve";
                return b->pcg.fault == 1 ? SO_ERR_TIMEOUT : SO_ERR_HIP;
            }
            memcpy(&lm, b->h_lm, sizeof(lm));
            if (lm.active == 1) {  // stage 1 wants more trials than were enqueued
                trials(r.d, std::max(1, lm.iterations - lm.it));
                if (two_stages) stage2_chained();
                epilogue();
                signal();
                continue;
            }
            if (lm.active == 2) {
                trials(d2, std::max(1, lm.iterations - lm.it));
                epilogue();
                signal();
                continue;
            }
            if (two_stages && lm.stages_begun == 1 && !stopped_between) {
                // stage 1 is over and stage 2 did not begin on the device: a stop request was up when its prologue ran
                if (r.terminate()) {
                    stopped_between = true;  // Optimizer.cc:641-643: bDoMore = false
                    break;
                }
                launch_ba_mark_outliers(d2, (double)opt->chi2_threshold, kBaGateNone, s);
                launch_ba_errors(d2, 0, kBaGateNone, r.nb_err, s);
                launch_ba_build(d2, kBaGateNone, s);
                launch_ba_stage_begin(d2, r.nb_err, opt->its_stage2, b->h_lm_dev, kBaGateNone, nullptr, s);
                trials(d2, opt->its_stage2);
                epilogue();
                signal();
                continue;
            }
            break;
        }
        r.returned_early = returned_early;
        const bool begun2 = lm.stages_begun >= 2;
        inf.iterations_stage1 = begun2 ? lm.prev_done : lm.done;
        inf.chi2_initial = begun2 ? lm.prev_chi_begin : lm.chi_begin;  // chi2 before optimising (information only)
        const double chi1 = begun2 ? lm.prev_chi_out : lm.chi_out;
        inf.chi2_final = inf.iterations_stage1 > 0 ? chi1 : inf.chi2_initial;
        if (begun2) {
            inf.iterations_stage2 = lm.done;
            if (lm.done > 0) inf.chi2_final = lm.chi_out;
            b->stage2_hint = std::max(1, (int)lm.done);  // (+ two in reserve behind the early epilogue)
        }
        if (stopped_between || r.terminate()) inf.aborted = 1;
        r.d.robust = begun2 ? 0 : r.d.robust;
        t_opt1 = now_ms();
    } else {
        // stage by stage (solve events on, SWARMORB_BA_NO_CHAIN, or nothing to optimise): the host sees every stage end
        StageChain none;
        if ((rc = optimize(r, opt->its_stage1, true, none, &done, &chi, nullptr))) return rc;  // optimizer.optimize(5)
        inf.iterations_stage1 = done;
        inf.chi2_initial = b->h_lm->chi_begin;  // chi2 before optimising (information only)
        inf.chi2_final = done > 0 ? chi : inf.chi2_initial;
        t_opt1 = now_ms();
        bool do_more = two_stages;
        if (r.terminate()) {
            do_more = false;
            inf.aborted = 1;
        }
        if (do_more) {
            launch_ba_mark_outliers(d2, (double)opt->chi2_threshold, kBaGateNone, s);
            r.d = d2;
            if ((rc = optimize(r, opt->its_stage2, true, none, &done, &chi, nullptr))) return rc;  // optimizer.optimize(10)
            inf.iterations_stage2 = done;
            if (done > 0) inf.chi2_final = chi;
            if (r.terminate()) inf.aborted = 1;
        }
        epilogue();
        SO_HIP(hipGetLastError());
        SO_HIP(hipStreamSynchronize(s));
    }
    const double t_opt2 = now_ms();
    const uint8_t* ho = (const uint8_t*)b->h_out;
    const BaPose* o_poses = (const BaPose*)(ho + r_pose);
    const double* o_pts = (const double*)(ho + r_pt);
    const double* o_chi2 = (const double*)(ho + r_chi2);
    const uint8_t* o_outl = ho + r_out;
    for (int k = 0; k < nE; k++) {
        const int e = perm[(size_t)k];
        if (edge_outlier) edge_outlier[e] = o_outl[k];
        if (edge_chi2) edge_chi2[e] = o_chi2[k];
        inf.n_outliers += o_outl[k];
    }
    for (int i = 0; i < nP; i++) pose_to_Tcw(o_poses[i], Tcw_out + 12 * (size_t)i);
    for (size_t i = 0; i < (size_t)nL * 3; i++) Xw_out[i] = (float)o_pts[i];
    inf.lambda_final = b->h_lm->lambda;
    inf.lm_trials = b->h_lm->trials;
    float ms = 0.f;
    if (grouped) {  // the span of the merged chain this window was part of
        so_ba_group* g = b->group;
        if (hipEventSynchronize(g->ev[2 * b->grp_event_slot + 1]) == hipSuccess &&
            hipEventElapsedTime(&ms, g->ev[2 * b->grp_event_slot], g->ev[2 * b->grp_event_slot + 1]) == hipSuccess)
            inf.gpu_ms = ms;
    } else if (hipEventElapsedTime(&ms, b->e0, r.returned_early ? b->e1a : b->e1) == hipSuccess) inf.gpu_ms = ms;
    inf.solve_ms = b->solve_ms;
    inf.n_solves = b->n_solves;
    if (r.d.plan) {
        inf.solve_gflop_structural = r.d.plan->flop_structural * 1e-9;
        inf.solve_gflop_dense = r.d.plan->flop_dense * 1e-9;
        inf.nnz_tiles = (double)r.d.plan->nnz_tiles;
        inf.solver_path = r.d.flow_tiles ? 2 : 1;
    }
    if (r.d.use_pcg) {
        inf.solver_path = 3;
        inf.pcg_iterations = (int32_t)std::min<long long>(b->pcg.iterations, INT32_MAX);
        inf.nnz_tiles = (double)b->pcg_nnz_blocks;  // (PCG: nonzero 6 x 6 blocks of S, not 96 x 96 tiles)
    }
    inf.n_free_keyframes = r.n_free;
    inf.flow_timeouts = b->flow_timeouts;
    inf.wall_ms = (float)(now_ms() - t_begin);
    if (trace)
        fprintf(stderr, "[ba] stage %.3f upload+alloc %.3f opt1 %.3f (%d it) opt2 %.3f (%d it) finish %.3f | trials %d blocks %d\n",
                t_staged - t_begin, t_uploaded - t_staged, t_opt1 - t_uploaded, inf.iterations_stage1,
                t_opt2 - t_opt1, inf.iterations_stage2, now_ms() - t_opt2, inf.lm_trials, r.blocks_enqueued);
    if (info) *info = inf;
    return SO_OK;
}

// Optimizer::PoseOptimization — code/src/Optimizer.cc:239-434.  Two halves (the tracking thread has host work that
// fits under the kernel: submitting the next frame, keyframe bookkeeping); so_pose_optimization = submit + wait.
int so_pose_optimization_submit(so_ba* b, const float* Tcw12, const float* intr, int32_t n, const float* Xw,
                                const float* obs, const float* inv_sigma2) {
    if (!b || !Tcw12 || !intr || n < 0 || b->pose_pending.active) return SO_ERR_INVALID_ARG;
    if (n > 0 && (!Xw || !obs || !inv_sigma2)) return SO_ERR_INVALID_ARG;
    so_ba::PosePending& Q = b->pose_pending;
    Q = so_ba::PosePending{};
    Q.active = true;
    Q.n = n;
    if (n < 3) return SO_OK;  // :358-359, nothing is touched
    SO_HIP(hipSetDevice(b->device));
    // PoseOptimization belongs to the tracking thread (Tracking.cc:743,779): it runs on that thread's matcher
    // stream, never on the solver's own stream where a local-mapping window may have ~100 launches queued
    hipStream_t s = nullptr;
    SO_HIP(tracking_stream(b->device, 1, &s));
    // one pinned staging block: [Xw 12n | obs 8n | w 4n] in, [pose 64 | info 16 | outlier n] out
    const size_t in_bytes = (size_t)n * 24, out_bytes = 64 + 16 + (size_t)n;
    const size_t need = in_bytes + out_bytes + 64;
    if (need > b->h_po_cap) {
        if (b->h_po) SO_HIP(hipHostFree(b->h_po));
        b->h_po = nullptr;
        b->h_po_cap = 0;
        SO_HIP(hipHostMalloc((void**)&b->h_po, need * 2, hipHostMallocMapped));
        SO_HIP(hipHostGetDevicePointer((void**)&b->h_po_dev, b->h_po, 0));
        b->h_po_cap = need * 2;
    }
    int rc;
    // device block: [inputs 24n | err 16n | pose 64 | info 16 | outlier n]
    const size_t off_err = ((size_t)n * 24 + 15) / 16 * 16, off_pose = off_err + (size_t)n * 16, off_info = off_pose + 64,
                 off_out = off_info + 16;
    const size_t off_trace = (off_out + (size_t)n + 15) / 16 * 16;
    if ((rc = b->d_po.ensure(off_trace + 256 * 32 + 64))) return rc;
    uint8_t* h = b->h_po;
    memcpy(h, Xw, (size_t)n * 12);
    memcpy(h + (size_t)n * 12, obs, (size_t)n * 8);
    memcpy(h + (size_t)n * 20, inv_sigma2, (size_t)n * 4);
    uint8_t* d = b->d_po.as<uint8_t>();
    // Frames of ordinary size (<= 3072 matched points) run the register-resident kernel, which reads its inputs once and
    // writes three small results: both go through host-mapped memory, so the call is one launch and no copies to
    // enqueue.  Larger problems re-read the edges every trial and therefore get a device copy.
    static const bool env_classic = getenv("SWARMORB_POSE_CLASSIC") != nullptr, env_trace = getenv("SWARMORB_POSE_TRACE") != nullptr;
    const bool zero_copy = n <= kPoseOptLdsMax && !env_classic;
    uint8_t* hout = h + in_bytes;
    PoseOptArgs a;
    if (zero_copy) {
        a.Xw = reinterpret_cast<const float*>(b->h_po_dev);
        a.obs = reinterpret_cast<const float*>(b->h_po_dev + (size_t)n * 12);
        a.inv_sigma2 = reinterpret_cast<const float*>(b->h_po_dev + (size_t)n * 20);
        a.pose_out = reinterpret_cast<BaPose*>(b->h_po_dev + in_bytes);
        a.info = reinterpret_cast<int*>(b->h_po_dev + in_bytes + 64);
        a.outlier = b->h_po_dev + in_bytes + 80;
    } else {
        SO_HIP(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
        a.Xw = reinterpret_cast<const float*>(d);
        a.obs = reinterpret_cast<const float*>(d + (size_t)n * 12);
        a.inv_sigma2 = reinterpret_cast<const float*>(d + (size_t)n * 20);
        a.pose_out = reinterpret_cast<BaPose*>(d + off_pose);
        a.info = reinterpret_cast<int*>(d + off_info);
        a.outlier = d + off_out;
    }
    for (int k = 0; k < 4; k++) a.K[k] = (double)intr[k];
    pose_from_Tcw(Tcw12, a.init);  // Converter::toSE3Quat(pFrame->mTcw)
    a.n = n;
    a.err = reinterpret_cast<double*>(d + off_err);
    a.trace = env_trace ? reinterpret_cast<double*>(d + off_trace) : nullptr;
    // zero-copy results: the kernel raises a completion word behind them and the host spins on it - the round trip ends
    // when the results are in host memory, not when the stream's completion signal has made its way back
    static const bool env_no_spin = getenv("SWARMORB_POSE_NO_SPIN") != nullptr;
    const bool spin = zero_copy && !env_no_spin;
    if (spin && ++b->pose_seq == 0) b->pose_seq = 1;
    a.done_seq = spin ? b->pose_seq : 0;
    // hout moves with n, so the completion word may land on bytes an earlier call left there (outlier flags, batch
    // arguments): clear it before the launch - pose_seq is never 0, so a stale word can never read as "done"
    if (spin) {
        *reinterpret_cast<volatile int*>(hout + 64 + 12) = 0;
        std::atomic_thread_fence(std::memory_order_release);
    }
    static const bool env_no_events = getenv("SWARMORB_NO_EVENTS") != nullptr;  // diagnostic: cost of the two event records
    const bool no_events = env_no_events || !b->pose_timing;
    if (!no_events) SO_HIP(hipEventRecord(b->pe0, s));
    launch_pose_opt(a, s);
    if (!no_events) SO_HIP(hipEventRecord(b->pe1, s));
    SO_HIP(hipGetLastError());
    if (!zero_copy) SO_HIP(hipMemcpyAsync(hout, d + off_pose, 64 + 16 + (size_t)n, hipMemcpyDeviceToHost, s));
    Q.launched = true;
    Q.stream = s;
    Q.hout = hout;
    Q.done_seq = a.done_seq;
    Q.events = !no_events;
    Q.trace = a.trace;
    return SO_OK;
}

int so_pose_optimization_wait(so_ba* b, float* Tcw_out12, uint8_t* outlier, int32_t* n_inliers, int32_t* info) {
    if (!b || !b->pose_pending.active || !Tcw_out12 || !n_inliers) return SO_ERR_INVALID_ARG;
    so_ba::PosePending Q = b->pose_pending;
    b->pose_pending.active = false;
    const int n = Q.n;
    if (n > 0 && !outlier) return SO_ERR_INVALID_ARG;
    *n_inliers = 0;
    if (info) info[0] = info[1] = 0;
    if (!Q.launched) return SO_OK;
    if (Q.done_seq) {
        const volatile int* done = reinterpret_cast<const volatile int*>(Q.hout + 64) + 3;
        // a kernel of 60-110 us: spin for about twice that, then hand the core back (several agents per GPU put every
        // tracking thread here at once, against the local-mapping threads) and let the stream's completion wake us
        for (unsigned long it = 1; *done != Q.done_seq; it++) {
            if ((it & 0x1fff) == 0) {  // every ~8 k polls (~0.2 ms): stop burning the core
                const hipError_t q = hipStreamSynchronize(Q.stream);
                if (q != hipSuccess || *done != Q.done_seq) {
                    last_error_ref() = q != hipSuccess ? std::string("PoseOptimization: ") + hipGetErrorString(q)
                                                       : std::string("PoseOptimization kernel finished without publishing its results");
                    return SO_ERR_HIP;
                }
                break;
            }
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        SO_HIP(hipStreamSynchronize(Q.stream));
    }
    b->pose_ms_pending = Q.events;  // the elapsed time is read (after the stop event) when somebody asks for it
    BaPose P;
    memcpy(&P, Q.hout, sizeof(BaPose));
    int inf[4];
    memcpy(inf, Q.hout + 64, 16);
    memcpy(outlier, Q.hout + 80, (size_t)n);
    if (Q.trace) {  // debugging aid: dump the LM trial log
        std::vector<double> tr(4 * 256);
        SO_HIP(so::memcpy_sync(tr.data(), Q.trace, sizeof(double) * tr.size(), hipMemcpyDeviceToHost));
        for (int k = 0; k < inf[2] && k < 256; k++)
            fprintf(stderr, "gpu trial %d lambda %.6e temp %.9e rho %.6e cur %.9e\n", k, tr[4 * k], tr[4 * k + 1], tr[4 * k + 2], tr[4 * k + 3]);
    }
    pose_to_Tcw(P, Tcw_out12);  // Converter::toCvMat(SE3quat_recov)
    *n_inliers = n - inf[0];
    if (info) {
        info[0] = inf[1];
        info[1] = inf[2];
    }
    return SO_OK;
}

int so_pose_optimization(so_ba* b, const float* Tcw12, const float* intr, int32_t n, const float* Xw, const float* obs,
                         const float* inv_sigma2, float* Tcw_out12, uint8_t* outlier, int32_t* n_inliers,
                         int32_t* info) {
    if (!b || !Tcw12 || !intr || n < 0 || !Tcw_out12 || !n_inliers) return SO_ERR_INVALID_ARG;
    if (n > 0 && (!Xw || !obs || !inv_sigma2 || !outlier)) return SO_ERR_INVALID_ARG;
    const int rc = so_pose_optimization_submit(b, Tcw12, intr, n, Xw, obs, inv_sigma2);
    if (rc != SO_OK) {
        b->pose_pending.active = false;
        return rc;
    }
    return so_pose_optimization_wait(b, Tcw_out12, outlier, n_inliers, info);
}

// Several independent PoseOptimization problems in ONE launch (a workgroup per problem): the frames of several agents
// driven in lockstep by one thread.  Inputs and results travel through host-mapped memory like the single call.
int so_pose_optimization_batch(so_ba* b, int32_t n_problems, const so_pose_problem* problems) {
    if (b && b->pose_pending.active) return SO_ERR_INVALID_ARG;  // a submitted single problem owns the staging block
    if (!b || n_problems < 0 || (n_problems > 0 && !problems)) return SO_ERR_INVALID_ARG;
    if (n_problems == 0) return SO_OK;
    int max_n = 0;
    size_t total_in = 0, total_out = 0;
    for (int p = 0; p < n_problems; p++) {
        const so_pose_problem& q = problems[p];
        if (!q.Tcw12 || !q.intr || q.n < 0 || !q.Tcw_out12 || !q.n_inliers) return SO_ERR_INVALID_ARG;
        if (q.n > 0 && (!q.Xw || !q.obs || !q.inv_sigma2 || !q.outlier)) return SO_ERR_INVALID_ARG;
        max_n = std::max(max_n, (int)q.n);
        total_in += ((size_t)q.n * 24 + 63) & ~(size_t)63;
        total_out += (80 + (size_t)q.n + 63) & ~(size_t)63;
    }
    if (max_n > 3072) {  // beyond the batched kernel: one call per problem
        for (int p = 0; p < n_problems; p++) {
            const so_pose_problem& q = problems[p];
            const int rc = so_pose_optimization(b, q.Tcw12, q.intr, q.n, q.Xw, q.obs, q.inv_sigma2, q.Tcw_out12, q.outlier,
                                                q.n_inliers, q.info);
            if (rc) return rc;
        }
        return SO_OK;
    }
    SO_HIP(hipSetDevice(b->device));
    hipStream_t s = nullptr;
    SO_HIP(tracking_stream(b->device, 1, &s));
    const size_t args_bytes = (sizeof(PoseOptArgs) * (size_t)n_problems + 255) & ~(size_t)255;
    const size_t need = args_bytes + total_in + total_out + 256;
    if (need > b->h_po_cap) {
        if (b->h_po) SO_HIP(hipHostFree(b->h_po));
        b->h_po = nullptr;
        b->h_po_cap = 0;
        SO_HIP(hipHostMalloc((void**)&b->h_po, need * 2, hipHostMallocMapped));
        SO_HIP(hipHostGetDevicePointer((void**)&b->h_po_dev, b->h_po, 0));
        b->h_po_cap = need * 2;
    }
    uint8_t* h = b->h_po;
    uint8_t* d = b->h_po_dev;
    PoseOptArgs* args = reinterpret_cast<PoseOptArgs*>(h);
    size_t off = args_bytes;
    std::vector<size_t> out_off((size_t)n_problems);
    for (int p = 0; p < n_problems; p++) {
        const so_pose_problem& q = problems[p];
        const size_t n = (size_t)q.n;
        PoseOptArgs& a = args[p];
        memset(&a, 0, sizeof(a));
        if (n) {
            memcpy(h + off, q.Xw, n * 12);
            memcpy(h + off + n * 12, q.obs, n * 8);
            memcpy(h + off + n * 20, q.inv_sigma2, n * 4);
        }
        a.Xw = reinterpret_cast<const float*>(d + off);
        a.obs = reinterpret_cast<const float*>(d + off + n * 12);
        a.inv_sigma2 = reinterpret_cast<const float*>(d + off + n * 20);
        off += (n * 24 + 63) & ~(size_t)63;
        out_off[(size_t)p] = off;
        a.pose_out = reinterpret_cast<BaPose*>(d + off);
        a.info = reinterpret_cast<int*>(d + off + 64);
        a.outlier = d + off + 80;
        off += (80 + n + 63) & ~(size_t)63;
        for (int k = 0; k < 4; k++) a.K[k] = (double)q.intr[k];
        pose_from_Tcw(q.Tcw12, a.init);
        a.n = q.n >= 3 ? q.n : 0;  // n < 3: "return 0" with untouched outputs (Optimizer.cc:358-359); the workgroup idles
        a.err = nullptr;
        a.trace = nullptr;
        a.done_seq = 0;
    }
    if (b->pose_timing) SO_HIP(hipEventRecord(b->pe0, s));
    if (!launch_pose_opt_batch(reinterpret_cast<const PoseOptArgs*>(d), n_problems, max_n, s)) return SO_ERR_INVALID_ARG;
    if (b->pose_timing) SO_HIP(hipEventRecord(b->pe1, s));
    SO_HIP(hipGetLastError());
    SO_HIP(hipStreamSynchronize(s));
    b->pose_ms_pending = b->pose_timing;
    for (int p = 0; p < n_problems; p++) {
        const so_pose_problem& q = problems[p];
        *q.n_inliers = 0;
        if (q.info) q.info[0] = q.info[1] = 0;
        if (q.n < 3) continue;
        const uint8_t* ho = h + out_off[(size_t)p];
        BaPose P;
        memcpy(&P, ho, sizeof(BaPose));
        int inf[4];
        memcpy(inf, ho + 64, 16);
        memcpy(q.outlier, ho + 80, (size_t)q.n);
        pose_to_Tcw(P, q.Tcw_out12);
        *q.n_inliers = q.n - inf[0];
        if (q.info) {
            q.info[0] = inf[1];
            q.info[1] = inf[2];
        }
    }
    return SO_OK;
}

int so_ba_set_linear_solver(so_ba* b, int solver, double rel_tolerance, int max_iterations) {
    if (!b || (solver != 0 && solver != 1)) return SO_ERR_INVALID_ARG;
    b->linear_solver = solver;
    if (rel_tolerance > 0.0) b->pcg.tol = rel_tolerance;
    if (max_iterations > 0) b->pcg.max_it = max_iterations;
    return SO_OK;
}

int so_bundle_adjust_set_solve_timing(so_ba* b, int enabled) {
    if (!b) return SO_ERR_INVALID_ARG;
    b->solve_timing = enabled != 0;
    return SO_OK;
}

int so_pose_optimization_set_timing(so_ba* b, int enabled) {
    if (!b) return SO_ERR_INVALID_ARG;
    b->pose_timing = enabled != 0;
    if (!b->pose_timing) {
        b->pose_ms_pending = false;
        b->pose_kernel_ms = 0.f;
    }
    return SO_OK;
}

int so_pose_optimization_last_kernel_ms(so_ba* b, float* ms) {
    if (!b || !ms) return SO_ERR_INVALID_ARG;
    if (b->pose_ms_pending) {
        (void)hipEventSynchronize(b->pe1);  // the call returned on the kernel's completion word, the event may trail it
        (void)hipEventElapsedTime(&b->pose_kernel_ms, b->pe0, b->pe1);
        b->pose_ms_pending = false;
    }
    *ms = b->pose_kernel_ms;
    return SO_OK;
}

}  // extern "C"
