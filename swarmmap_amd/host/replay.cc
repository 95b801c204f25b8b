// replay.cc — the per-frame replay of SURVEY 8d as a C++ host loop over the C ABI (include/swarmorb.h): what an
// agent's Tracking thread and LocalMapping thread do with the library, without an interpreter in the loop.
//   tracking thread (the caller of so_replay_run), per frame t:
//       collect frame t from the extractor, submit frame t+1            (ORBextractor::operator(), pipelined)
//       SearchByProjection(cur, last, th 15)                             (Tracking.cc:715, prepared projections)
//       SearchByProjection(cur, local map points, th 1)                  (Tracking.cc:998)
//       3 x Optimizer::PoseOptimization                                  (Tracking.cc:716,1002 + one retry)
//       every lba_every frames: hand a window to the local-mapping thread (at most two waiting: back-pressure)
//   local-mapping thread: Optimizer::LocalBundleAdjustment on each queued window, in order
// bench.py prepares the inputs (device images, projections, pose problems, the window), calls so_replay_run for
// the timed region and reads the accumulated statistics.  Built by csrc/Makefile into libswarmorb_replay.so with g++.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/swarmorb.h"

namespace {

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct StepInputs {  // the tracking thread's projections for one frame (inputs of M2 / M1)
    std::vector<uint8_t> last_valid, last_desc, last_has_obs;
    std::vector<float> last_u, last_v, last_angle;
    std::vector<int32_t> last_octave;
    std::vector<uint8_t> mp_in_view, mp_desc, mp_has_obs;
    std::vector<float> mp_x, mp_y, mp_cos;
    std::vector<int32_t> mp_level;
};

struct PoseCase {
    float Tcw[12], K[4];
    std::vector<float> Xw, obs, w;
};

struct BaWindow {
    std::vector<float> Tcw, intr, Xw, obs, w;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> epose, epoint;
};

}  // namespace

struct so_replay {
    int device = 0, width = 0, height = 0, lba_every = 5;
    so_extractor* ex = nullptr;
    so_matcher* matcher = nullptr;  // one per tracking thread: both searches of a frame share its candidate upload
    so_ba* tracker_opt = nullptr;
    so_ba* mapper_opt = nullptr;
    std::vector<const uint8_t*> frames;
    std::vector<StepInputs> steps;
    std::vector<PoseCase> poses;  // 3 per group
    BaWindow window;
    float scale_factors[8] = {0};
    int nlevels = 8;
    // extractor outputs
    std::vector<so_keypoint> kps;
    std::vector<uint8_t> desc;
    std::vector<float> x, y, angle;
    std::vector<int32_t> octave, kp_to;
    int n_kp = 0;
    bool in_flight = false;
    // local-mapping thread
    std::thread mapper;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> queue;  // 1 = timed window, 0 = warm-up window
    int running = 0;
    bool quit = false;
    std::string error;
    // statistics (timed steps only)
    double stat[32] = {0};
    std::vector<float> ba_Tcw, ba_Xw;
    std::vector<uint8_t> ba_out;
};

namespace {

enum {  // indices of so_replay::stat, mirrored in bench.py
    kSteps = 0, kExtractMs, kMatchMs, kPoseMs, kSubmitWaitMs, kKp, kM2, kM1, kMatchKernelMs, kPoseKernelMs, kPoseTrials,
    kPoseCalls, kLbaWindows, kLbaBusyMs, kLbaGpuMs, kLbaSolveMs, kLbaSolves, kStage0 /* 11 extractor stages */
};

void mapper_loop(so_replay* r) {
    (void)0;
    for (;;) {
        int timed;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv.wait(lk, [r] { return r->quit || !r->queue.empty(); });
            if (r->queue.empty()) return;
            timed = r->queue.front();
            r->running = 1;
        }
        const double t0 = now_ms();
        so_ba_problem p{};
        const BaWindow& w = r->window;
        p.n_poses = (int32_t)w.fixed.size();
        p.Tcw = w.Tcw.data();
        p.fixed = w.fixed.data();
        p.intr = w.intr.data();
        p.n_points = (int32_t)(w.Xw.size() / 3);
        p.Xw = w.Xw.data();
        p.n_edges = (int32_t)w.epose.size();
        p.edge_pose = w.epose.data();
        p.edge_point = w.epoint.data();
        p.obs = w.obs.data();
        p.inv_sigma2 = w.w.data();
        so_ba_options opt;
        so_ba_options_local(&opt);
        so_ba_info info{};
        const int rc = so_bundle_adjust(r->mapper_opt, &p, &opt, nullptr, r->ba_Tcw.data(), r->ba_Xw.data(),
                                        r->ba_out.data(), nullptr, &info);
        const double busy = now_ms() - t0;
        {
            std::lock_guard<std::mutex> lk(r->mu);
            if (rc != SO_OK && r->error.empty()) r->error = std::string("so_bundle_adjust: ") + so_last_error();
            if (timed) {
                r->stat[kLbaWindows] += 1;
                r->stat[kLbaBusyMs] += busy;
                r->stat[kLbaGpuMs] += info.gpu_ms;
                r->stat[kLbaSolveMs] += info.solve_ms;
                r->stat[kLbaSolves] += info.n_solves;
            }
            r->queue.pop_front();
            r->running = 0;
        }
        r->cv.notify_all();
    }
}

int fail(so_replay* r, const char* what) {
    r->error = std::string(what) + ": " + so_last_error();
    return SO_ERR_HIP;
}

}  // namespace

extern "C" {

int so_replay_create(int device, int width, int height, int nfeatures, int lba_every, so_replay** out) {
    if (!out) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    so_replay* r = new so_replay();
    r->device = device;
    r->width = width;
    r->height = height;
    r->lba_every = lba_every > 0 ? lba_every : 5;
    so_extractor_config cfg{nfeatures, 1.2f, 8, 20, 7, device};
    int rc = so_extractor_create(&cfg, &r->ex);
    if (rc == SO_OK) rc = so_matcher_create(device, &r->matcher);
    if (rc == SO_OK) rc = so_ba_create(device, &r->tracker_opt);
    if (rc == SO_OK) rc = so_ba_create(device, &r->mapper_opt);
    if (rc != SO_OK) {
        delete r;
        return rc;
    }
    float inv[8], s2[8], is2[8];
    int32_t npl[8];
    so_extractor_tables(r->ex, r->scale_factors, inv, s2, is2, npl);
    const int cap = so_extractor_capacity(r->ex);
    r->kps.resize((size_t)cap);
    r->desc.resize((size_t)cap * 32);
    r->x.resize((size_t)cap); r->y.resize((size_t)cap); r->angle.resize((size_t)cap);
    r->octave.resize((size_t)cap); r->kp_to.resize((size_t)cap);
    r->mapper = std::thread(mapper_loop, r);
    *out = r;
    return SO_OK;
}

void so_replay_destroy(so_replay* r) {
    if (!r) return;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->quit = true;
    }
    r->cv.notify_all();
    if (r->mapper.joinable()) r->mapper.join();
    if (r->in_flight) {
        int n = 0;
        (void)so_extractor_collect(r->ex, r->kps.data(), r->desc.data(), (int)r->kps.size(), &n);
    }
    so_extractor_destroy(r->ex);
    so_matcher_destroy(r->matcher);
    so_ba_destroy(r->tracker_opt);
    so_ba_destroy(r->mapper_opt);
    delete r;
}

const char* so_replay_error(so_replay* r) { return r ? r->error.c_str() : "null handle"; }

int so_replay_set_frames(so_replay* r, const uint64_t* device_pointers, int n) {
    if (!r || !device_pointers || n <= 0) return SO_ERR_INVALID_ARG;
    r->frames.clear();
    for (int i = 0; i < n; i++) r->frames.push_back(reinterpret_cast<const uint8_t*>(device_pointers[i]));
    return SO_OK;
}

int so_replay_set_step(so_replay* r, int t, int n_last, const uint8_t* valid, const float* u, const float* v,
                       const int32_t* octave, const float* angle, const uint8_t* desc, const uint8_t* has_obs, int n_mp,
                       const uint8_t* in_view, const float* px, const float* py, const float* view_cos,
                       const int32_t* level, const uint8_t* mp_desc, const uint8_t* mp_has_obs) {
    if (!r || t < 0) return SO_ERR_INVALID_ARG;
    if ((size_t)t >= r->steps.size()) r->steps.resize((size_t)t + 1);
    StepInputs& s = r->steps[(size_t)t];
    s.last_valid.assign(valid, valid + n_last);
    s.last_u.assign(u, u + n_last);
    s.last_v.assign(v, v + n_last);
    s.last_octave.assign(octave, octave + n_last);
    s.last_angle.assign(angle, angle + n_last);
    s.last_desc.assign(desc, desc + (size_t)n_last * 32);
    s.last_has_obs.assign(has_obs, has_obs + n_last);
    s.mp_in_view.assign(in_view, in_view + n_mp);
    s.mp_x.assign(px, px + n_mp);
    s.mp_y.assign(py, py + n_mp);
    s.mp_cos.assign(view_cos, view_cos + n_mp);
    s.mp_level.assign(level, level + n_mp);
    s.mp_desc.assign(mp_desc, mp_desc + (size_t)n_mp * 32);
    s.mp_has_obs.assign(mp_has_obs, mp_has_obs + n_mp);
    return SO_OK;
}

int so_replay_add_pose_case(so_replay* r, const float* Tcw12, const float* K4, int n, const float* Xw, const float* obs,
                            const float* inv_sigma2) {
    if (!r || n < 0) return SO_ERR_INVALID_ARG;
    PoseCase c;
    memcpy(c.Tcw, Tcw12, 48);
    memcpy(c.K, K4, 16);
    c.Xw.assign(Xw, Xw + 3 * (size_t)n);
    c.obs.assign(obs, obs + 2 * (size_t)n);
    c.w.assign(inv_sigma2, inv_sigma2 + n);
    r->poses.push_back(std::move(c));
    return SO_OK;
}

int so_replay_set_window(so_replay* r, const so_ba_problem* p) {
    if (!r || !p) return SO_ERR_INVALID_ARG;
    BaWindow& w = r->window;
    w.Tcw.assign(p->Tcw, p->Tcw + 12 * (size_t)p->n_poses);
    w.fixed.assign(p->fixed, p->fixed + p->n_poses);
    w.intr.assign(p->intr, p->intr + 4 * (size_t)p->n_poses);
    w.Xw.assign(p->Xw, p->Xw + 3 * (size_t)p->n_points);
    w.epose.assign(p->edge_pose, p->edge_pose + p->n_edges);
    w.epoint.assign(p->edge_point, p->edge_point + p->n_edges);
    w.obs.assign(p->obs, p->obs + 2 * (size_t)p->n_edges);
    w.w.assign(p->inv_sigma2, p->inv_sigma2 + p->n_edges);
    r->ba_Tcw.resize(w.Tcw.size());
    r->ba_Xw.resize(w.Xw.size());
    r->ba_out.resize(w.epose.size());
    return SO_OK;
}

// Resource allocation before any step is counted: one window through the local-mapping handle and one pose problem
// through the tracking handle size their device buffers (what a process does once at start-up, not per frame).
int so_replay_preallocate(so_replay* r) {
    if (!r || r->window.epose.empty() || r->poses.empty()) return SO_ERR_INVALID_ARG;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        r->queue.push_back(0);
    }
    r->cv.notify_all();
    const PoseCase& c = r->poses[0];
    std::vector<uint8_t> outl(c.w.size());
    float Tout[12];
    int32_t inl = 0;
    if (so_pose_optimization(r->tracker_opt, c.Tcw, c.K, (int)c.w.size(), c.Xw.data(), c.obs.data(), c.w.data(), Tout,
                             outl.data(), &inl, nullptr) != SO_OK)
        return fail(r, "so_pose_optimization");
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv.wait(lk, [r] { return r->queue.empty() && !r->running; });
    return r->error.empty() ? SO_OK : SO_ERR_HIP;
}

int so_replay_set_profiling(so_replay* r, int enabled) { return r ? so_extractor_set_profiling(r->ex, enabled) : SO_ERR_INVALID_ARG; }

// Runs frames [first_t, first_t + n_steps).  Frame first_t must be in flight (so_replay_prime) - the loop collects
// it, submits the next one and leaves that one in flight when it returns.
int so_replay_prime(so_replay* r, int t) {
    if (!r || r->frames.empty() || r->in_flight) return SO_ERR_INVALID_ARG;
    const int rc = so_extractor_submit_device(r->ex, r->frames[(size_t)t % r->frames.size()], r->width, r->height, r->width);
    if (rc != SO_OK) return fail(r, "so_extractor_submit_device");
    r->in_flight = true;
    return SO_OK;
}

int so_replay_run(so_replay* r, int first_t, int n_steps, int timed) {
    if (!r || !r->in_flight || r->frames.empty() || r->poses.size() < 3) return SO_ERR_INVALID_ARG;
    const int groups = (int)r->poses.size() / 3;
    std::vector<uint8_t> outl;
    for (int t = first_t; t < first_t + n_steps; t++) {
        if ((size_t)t >= r->steps.size()) return SO_ERR_INVALID_ARG;
        const double t0 = now_ms();
        int n = 0;
        if (so_extractor_collect(r->ex, r->kps.data(), r->desc.data(), (int)r->kps.size(), &n) != SO_OK)
            return fail(r, "so_extractor_collect");
        r->in_flight = false;
        if (so_extractor_submit_device(r->ex, r->frames[(size_t)(t + 1) % r->frames.size()], r->width, r->height,
                                       r->width) != SO_OK)
            return fail(r, "so_extractor_submit_device");
        r->in_flight = true;
        r->n_kp = n;
        for (int i = 0; i < n; i++) {
            const so_keypoint& k = r->kps[(size_t)i];
            r->x[(size_t)i] = k.x; r->y[(size_t)i] = k.y; r->angle[(size_t)i] = k.angle; r->octave[(size_t)i] = k.octave;
        }
        const double t1 = now_ms();
        so_frame_view F{};
        F.n = n; F.x = r->x.data(); F.y = r->y.data(); F.octave = r->octave.data(); F.angle = r->angle.data();
        F.desc = r->desc.data(); F.excluded = nullptr;
        F.min_x = 0.f; F.max_x = (float)r->width; F.min_y = 0.f; F.max_y = (float)r->height;
        F.grid_inv_w = 64.0f / (F.max_x - F.min_x);
        F.grid_inv_h = 48.0f / (F.max_y - F.min_y);
        F.scale_factors = r->scale_factors;
        F.nlevels = r->nlevels;
        const StepInputs& s = r->steps[(size_t)t];
        int32_t nm2 = 0, nm1 = 0;
        float k2 = 0.f, k1 = 0.f;
        if (so_search_by_projection_lastframe(r->matcher, &F, (int32_t)s.last_u.size(), s.last_valid.data(), s.last_u.data(),
                                              s.last_v.data(), s.last_octave.data(), s.last_angle.data(),
                                              s.last_desc.data(), s.last_has_obs.data(), 15.0f, 1, r->kp_to.data(),
                                              &nm2) != SO_OK)
            return fail(r, "so_search_by_projection_lastframe");
        so_matcher_last_kernel_ms(r->matcher, &k2);
        so_matcher_reuse_frame(r->matcher);  // the local-map search looks at the same frame (Tracking.cc:1014 then :1153)
        if (so_search_by_projection_mappoints(r->matcher, &F, (int32_t)s.mp_x.size(), s.mp_in_view.data(), s.mp_x.data(),
                                              s.mp_y.data(), s.mp_cos.data(), s.mp_level.data(), s.mp_desc.data(),
                                              s.mp_has_obs.data(), 1.0f, 0.8f, r->kp_to.data(), &nm1) != SO_OK)
            return fail(r, "so_search_by_projection_mappoints");
        so_matcher_last_kernel_ms(r->matcher, &k1);
        const double t2 = now_ms();
        double pose_kernel = 0.0, pose_trials = 0.0;
        for (int j = 0; j < 3; j++) {
            const PoseCase& c = r->poses[(size_t)(3 * (t % groups) + j)];
            const int np = (int)c.w.size();
            outl.resize((size_t)np);
            float Tout[12];
            int32_t inl = 0, info[2] = {0, 0};
            if (so_pose_optimization(r->tracker_opt, c.Tcw, c.K, np, c.Xw.data(), c.obs.data(), c.w.data(), Tout,
                                     outl.data(), &inl, info) != SO_OK)
                return fail(r, "so_pose_optimization");
            float ms = 0.f;
            so_pose_optimization_last_kernel_ms(r->tracker_opt, &ms);
            pose_kernel += ms;
            pose_trials += info[1];
        }
        const double t3 = now_ms();
        if (t % r->lba_every == 0) {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv.wait(lk, [r] { return r->queue.size() < 3; });  // the running window + two waiting
            r->queue.push_back(timed ? 1 : 0);
            lk.unlock();
            r->cv.notify_all();
        }
        const double t4 = now_ms();
        if (timed) {
            double* st = r->stat;
            st[kSteps] += 1; st[kExtractMs] += t1 - t0; st[kMatchMs] += t2 - t1; st[kPoseMs] += t3 - t2;
            st[kSubmitWaitMs] += t4 - t3; st[kKp] += n; st[kM2] += nm2; st[kM1] += nm1; st[kMatchKernelMs] += k1 + k2;
            st[kPoseKernelMs] += pose_kernel; st[kPoseTrials] += pose_trials; st[kPoseCalls] += 3;
            float prof[SO_EXTRACTOR_N_STAGES];
            if (so_extractor_get_profile(r->ex, prof) == SO_OK)
                for (int i = 0; i < SO_EXTRACTOR_N_STAGES; i++) st[kStage0 + i] += prof[i];
        }
        if (!r->error.empty()) return SO_ERR_HIP;
    }
    return SO_OK;
}

// Wait until the local-mapping thread has optimised every queued window.
int so_replay_drain(so_replay* r) {
    if (!r) return SO_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv.wait(lk, [r] { return r->queue.empty() && !r->running; });
    return r->error.empty() ? SO_OK : SO_ERR_HIP;
}

// Collects the frame left in flight by so_replay_run (end of a run; its extraction was part of the timed region).
int so_replay_finish(so_replay* r) {
    if (!r) return SO_ERR_INVALID_ARG;
    if (r->in_flight) {
        int n = 0;
        if (so_extractor_collect(r->ex, r->kps.data(), r->desc.data(), (int)r->kps.size(), &n) != SO_OK)
            return fail(r, "so_extractor_collect");
        r->in_flight = false;
        r->n_kp = n;
    }
    return SO_OK;
}

int so_replay_stats(so_replay* r, double* out32) {
    if (!r || !out32) return SO_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(r->mu);
    memcpy(out32, r->stat, sizeof(r->stat));
    return SO_OK;
}

// descriptors of the last collected frame (for the cross-agent exchange tick) and the extractor handle (candidates)
int so_replay_last_frame(so_replay* r, const uint8_t** desc, int* n) {
    if (!r || !desc || !n) return SO_ERR_INVALID_ARG;
    *desc = r->desc.data();
    *n = r->n_kp;
    return SO_OK;
}
so_extractor* so_replay_extractor(so_replay* r) { return r ? r->ex : nullptr; }
so_matcher* so_replay_matcher(so_replay* r) { return r ? r->matcher : nullptr; }

}  // extern "C"
