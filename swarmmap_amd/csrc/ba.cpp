// ba.cpp — host driver + C ABI of the bundle-adjustment solver (include/swarmorb.h).
//
// Replaces Optimizer::LocalBundleAdjustment / BundleAdjustment (code/src/Optimizer.cc:42-237,436-740) from
// "build g2o graph" to "recover optimized data", on a flattened problem.  The Levenberg-Marquardt control
// flow (code/Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-164, sparse_optimizer.cpp:354-419)
// runs on the host and reads back three scalars per trial (chi2, scale, solve-ok); all arithmetic on residuals,
// Jacobians, the Schur complement, the reduced solve and the manifold updates runs in the kernels of
// ba_kernels.hip.  Estimates live in a current/trial buffer pair, so g2o's push/pop/discardTop is a pointer swap.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "ba_device.h"
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

constexpr int kMaxReducedDim = 6144;  // dense reduced camera system: 6 * n_free <= this (301 MB of FP64)

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---- host copies of the SE3Quat conversions at the map boundary (Converter.cc:37-47,49-74; se3quat.h:58-60) ----
void quat_from_R(const double* R, double* q) {  // Eigen::Quaterniond(Matrix3d), published algorithm
    double t = R[0] + R[4] + R[8];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}

void pose_from_Tcw(const float* T, BaPose& P) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    quat_from_R(R, P.q);
    if (P.q[3] < 0) for (double& v : P.q) v = -v;  // normalizeRotation
    const double n = std::sqrt(P.q[0] * P.q[0] + P.q[1] * P.q[1] + P.q[2] * P.q[2] + P.q[3] * P.q[3]);
    for (double& v : P.q) v /= n;
    P.t[0] = T[3];
    P.t[1] = T[7];
    P.t[2] = T[11];
    P.pad = 0;
}

void pose_to_Tcw(const BaPose& P, float* T) {  // to_homogeneous_matrix cast to float
    const double* q = P.q;
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    T[0] = (float)(1 - (tyy + tzz)); T[1] = (float)(txy - twz);       T[2] = (float)(txz + twy);        T[3] = (float)P.t[0];
    T[4] = (float)(txy + twz);       T[5] = (float)(1 - (txx + tzz)); T[6] = (float)(tyz - twx);        T[7] = (float)P.t[1];
    T[8] = (float)(txz - twy);       T[9] = (float)(tyz + twx);       T[10] = (float)(1 - (txx + tyy)); T[11] = (float)P.t[2];
}

}  // namespace

struct so_ba {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    static constexpr int kSolveEvents = 32;  // the first trials of a call are event-timed around the solve kernel
    hipEvent_t ev_solve[2 * kSolveEvents] = {nullptr};
    float solve_ms = 0.f;
    int n_solves = 0;
    BaLm* h_lm = nullptr;        // host-mapped copy of the LM state, written by the decision kernels
    BaLm* h_lm_dev = nullptr;
    uint8_t* h_abort = nullptr;  // host-mapped forceStopFlag the decision kernel polls
    uint8_t* h_abort_dev = nullptr;

    Buf d_pose[2], d_pt[2], d_intr, d_epose, d_ept, d_obs, d_w, d_active, d_err, d_chi2, d_ptoff, d_ptact, d_hidx,
        d_freepose, d_poseoff, d_poseedges, d_blkoff, d_blki1, d_blki2, d_pk1, d_pk2, d_Hpp, d_bp, d_Hll, d_bl, d_W,
        d_Dinv, d_db, d_BDinv, d_S, d_bs, d_xl, d_partial, d_depth, d_po, d_lm;
    uint8_t* h_po = nullptr;      // pinned staging for PoseOptimization
    size_t h_po_cap = 0;
    std::vector<Buf*> all() {
        return {&d_pose[0], &d_pose[1], &d_pt[0], &d_pt[1], &d_intr, &d_epose, &d_ept, &d_obs, &d_w, &d_active, &d_err,
                &d_chi2, &d_ptoff, &d_ptact, &d_hidx, &d_freepose, &d_poseoff, &d_poseedges, &d_blkoff, &d_blki1,
                &d_blki2, &d_pk1, &d_pk2, &d_Hpp, &d_bp, &d_Hll, &d_bl, &d_W, &d_Dinv, &d_db, &d_BDinv, &d_S, &d_bs,
                &d_xl, &d_partial, &d_depth, &d_po, &d_lm};
    }
};

namespace {

template <typename T>
int upload(so_ba* b, Buf& buf, const std::vector<T>& v) {
    int rc = buf.ensure(sizeof(T) * std::max<size_t>(v.size(), 1));
    if (rc) return rc;
    if (!v.empty()) SO_HIP(hipMemcpyAsync(buf.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, b->stream));
    return SO_OK;
}

struct Problem {  // host-side working copy, edges sorted by point
    int n_poses = 0, n_points = 0, n_edges = 0;
    std::vector<int> perm;  // sorted position -> original edge index
    std::vector<int> e_pose, e_point;
    std::vector<int> pt_off;
    std::vector<uint8_t> fixed;
    std::vector<int> level;  // per sorted edge
};

struct Stage {
    int n_free = 0, n_blk = 0, n_active_edges = 0;
    std::vector<uint8_t> e_active, pt_active;
    std::vector<int> pose_hidx, free_pose, pose_off, pose_edges, blk_off, blk_i1, blk_i2, pair_k1, pair_k2;
};

// SparseOptimizer::initializeOptimization(0) + buildIndexMapping (sparse_optimizer.cpp:166-270) and the CSR
// lists the kernels reduce over.
void build_stage(const Problem& P, Stage& S) {
    S.e_active.assign((size_t)P.n_edges, 0);
    S.pt_active.assign((size_t)P.n_points, 0);
    std::vector<uint8_t> pose_touched((size_t)P.n_poses, 0);
    S.n_active_edges = 0;
    for (int e = 0; e < P.n_edges; e++)
        if (P.level[(size_t)e] == 0) {
            S.e_active[(size_t)e] = 1;
            S.pt_active[(size_t)P.e_point[(size_t)e]] = 1;
            pose_touched[(size_t)P.e_pose[(size_t)e]] = 1;
            S.n_active_edges++;
        }
    S.pose_hidx.assign((size_t)P.n_poses, -1);
    S.free_pose.clear();
    for (int i = 0; i < P.n_poses; i++)
        if (pose_touched[(size_t)i] && !P.fixed[(size_t)i]) {
            S.pose_hidx[(size_t)i] = (int)S.free_pose.size();
            S.free_pose.push_back(i);
        }
    S.n_free = (int)S.free_pose.size();
    // free pose -> active edges (ascending sorted-edge index)
    S.pose_off.assign((size_t)S.n_free + 1, 0);
    for (int e = 0; e < P.n_edges; e++)
        if (S.e_active[(size_t)e]) {
            const int h = S.pose_hidx[(size_t)P.e_pose[(size_t)e]];
            if (h >= 0) S.pose_off[(size_t)h + 1]++;
        }
    for (int i = 0; i < S.n_free; i++) S.pose_off[(size_t)i + 1] += S.pose_off[(size_t)i];
    S.pose_edges.assign((size_t)S.pose_off[(size_t)S.n_free], 0);
    {
        std::vector<int> fill(S.pose_off.begin(), S.pose_off.end() - 1);
        for (int e = 0; e < P.n_edges; e++)
            if (S.e_active[(size_t)e]) {
                const int h = S.pose_hidx[(size_t)P.e_pose[(size_t)e]];
                if (h >= 0) S.pose_edges[(size_t)fill[(size_t)h]++] = e;
            }
    }
    // upper blocks of the reduced camera system and the edge pairs that feed them
    const int nf = S.n_free;
    S.n_blk = nf * (nf + 1) / 2;
    S.blk_i1.resize((size_t)S.n_blk);
    S.blk_i2.resize((size_t)S.n_blk);
    {
        int g = 0;
        for (int i1 = 0; i1 < nf; i1++)
            for (int i2 = i1; i2 < nf; i2++, g++) {
                S.blk_i1[(size_t)g] = i1;
                S.blk_i2[(size_t)g] = i2;
            }
    }
    auto blk_id = [nf](int i1, int i2) { return i1 * nf - i1 * (i1 - 1) / 2 + (i2 - i1); };
    S.blk_off.assign((size_t)S.n_blk + 1, 0);
    std::vector<std::pair<int, int>> obs;  // (hessian index, edge) of one landmark
    for (int pass = 0; pass < 2; pass++) {
        std::vector<int> fill;
        if (pass == 1) {
            for (int g = 0; g < S.n_blk; g++) S.blk_off[(size_t)g + 1] += S.blk_off[(size_t)g];
            S.pair_k1.assign((size_t)S.blk_off[(size_t)S.n_blk], 0);
            S.pair_k2.assign((size_t)S.blk_off[(size_t)S.n_blk], 0);
            fill.assign(S.blk_off.begin(), S.blk_off.end() - 1);
        }
        for (int il = 0; il < P.n_points; il++) {
            if (!S.pt_active[(size_t)il]) continue;
            obs.clear();
            for (int e = P.pt_off[(size_t)il]; e < P.pt_off[(size_t)il + 1]; e++)
                if (S.e_active[(size_t)e]) {
                    const int h = S.pose_hidx[(size_t)P.e_pose[(size_t)e]];
                    if (h >= 0) obs.emplace_back(h, e);
                }
            std::sort(obs.begin(), obs.end());
            for (size_t a = 0; a < obs.size(); a++)
                for (size_t c = a; c < obs.size(); c++) {
                    const int g = blk_id(obs[a].first, obs[c].first);
                    if (pass == 0) {
                        S.blk_off[(size_t)g + 1]++;
                    } else {
                        const int o = fill[(size_t)g]++;
                        S.pair_k1[(size_t)o] = obs[a].second;
                        S.pair_k2[(size_t)o] = obs[c].second;
                    }
                }
        }
    }
}

struct Run {
    so_ba* b;
    BaDev d{};
    Problem P;
    Stage S;
    int nb_err = 1, nb_upd = 1;
    int blocks_enqueued = 0;  // trial blocks of this call so far (indexes the solve-event pool)
    const volatile uint8_t* stop = nullptr;
    bool terminate() const { return stop && *stop; }
};

int upload_stage(Run& r) {
    so_ba* b = r.b;
    Stage& S = r.S;
    int rc;
    if ((rc = upload(b, b->d_active, S.e_active))) return rc;
    if ((rc = upload(b, b->d_ptact, S.pt_active))) return rc;
    if ((rc = upload(b, b->d_hidx, S.pose_hidx))) return rc;
    if ((rc = upload(b, b->d_freepose, S.free_pose))) return rc;
    if ((rc = upload(b, b->d_poseoff, S.pose_off))) return rc;
    if ((rc = upload(b, b->d_poseedges, S.pose_edges))) return rc;
    if ((rc = upload(b, b->d_blkoff, S.blk_off))) return rc;
    if ((rc = upload(b, b->d_blki1, S.blk_i1))) return rc;
    if ((rc = upload(b, b->d_blki2, S.blk_i2))) return rc;
    if ((rc = upload(b, b->d_pk1, S.pair_k1))) return rc;
    if ((rc = upload(b, b->d_pk2, S.pair_k2))) return rc;
    const size_t nf = (size_t)std::max(S.n_free, 1), n = 6 * nf;
    if ((rc = b->d_Hpp.ensure(sizeof(double) * 36 * nf))) return rc;
    if ((rc = b->d_bp.ensure(sizeof(double) * 6 * nf))) return rc;
    if ((rc = b->d_S.ensure(sizeof(double) * n * n))) return rc;
    if ((rc = b->d_bs.ensure(sizeof(double) * n))) return rc;
    BaDev& d = r.d;
    d.n_free = S.n_free;
    d.e_active = b->d_active.as<uint8_t>();
    d.pt_active = b->d_ptact.as<uint8_t>();
    d.pose_hidx = b->d_hidx.as<int>();
    d.free_pose = b->d_freepose.as<int>();
    d.pose_off = b->d_poseoff.as<int>();
    d.pose_edges = b->d_poseedges.as<int>();
    d.blk_off = b->d_blkoff.as<int>();
    d.pair_k1 = b->d_pk1.as<int>();
    d.pair_k2 = b->d_pk2.as<int>();
    d.Hpp = b->d_Hpp.as<double>();
    d.bp = b->d_bp.as<double>();
    d.S = b->d_S.as<double>();
    d.bs = b->d_bs.as<double>();
    return SO_OK;
}

// Wait for the stream; with a forceStopFlag, poll it meanwhile and forward it to the device.
int wait_stream(Run& r) {
    hipStream_t s = r.b->stream;
    if (!r.stop) {
        SO_HIP(hipStreamSynchronize(s));
        return SO_OK;
    }
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return SO_OK;
        if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
        if (*r.stop) *r.b->h_abort = 1;
    }
}

// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg, driven from the device: the host
// enqueues the prologue (errors, linearisation, lambda init) and `iterations` trials, waits once and only
// enqueues more if trials were rejected (each rejected trial needs one more than the iteration count).
int optimize(Run& r, int iterations, int* done_out, double* chi_out) {
    so_ba* b = r.b;
    hipStream_t s = b->stream;
    *done_out = 0;
    if (r.S.n_free + (r.S.n_active_edges > 0 ? 1 : 0) == 0) return SO_OK;  // 0 vertices to optimize
    launch_ba_errors(r.d, 0, false, r.nb_err, s);
    launch_ba_build(r.d, false, s);
    launch_ba_maxdiag(r.d, s);
    launch_ba_stage_begin(r.d, r.nb_err, iterations, b->h_lm_dev, s);
    SO_HIP(hipGetLastError());
    const int trials_before = b->h_lm->trials, first_block = r.blocks_enqueued;  // h_lm: state after the last wait
    int budget = iterations, rc;
    for (;;) {
        for (int i = 0; i < budget; i++) {
            const int k = r.blocks_enqueued++;
            const bool timed = k < so_ba::kSolveEvents;
            launch_ba_trial(r.d, b->d_blki1.as<int>(), b->d_blki2.as<int>(), r.S.n_blk, r.nb_err, r.nb_upd,
                            r.stop ? b->h_abort_dev : nullptr, b->h_lm_dev, timed ? b->ev_solve[2 * k] : nullptr,
                            timed ? b->ev_solve[2 * k + 1] : nullptr, s);
        }
        SO_HIP(hipGetLastError());
        if ((rc = wait_stream(r))) return rc;
        BaLm lm;
        memcpy(&lm, b->h_lm, sizeof(lm));
        if (!lm.active) break;
        budget = std::max(1, lm.iterations - lm.it);
    }
    BaLm lm;
    memcpy(&lm, b->h_lm, sizeof(lm));
    const int real = lm.trials - trials_before;  // the first `real` blocks of this stage ran, the rest returned at once
    for (int k = first_block; k < first_block + real && k < so_ba::kSolveEvents; k++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, b->ev_solve[2 * k], b->ev_solve[2 * k + 1]) == hipSuccess) {
            b->solve_ms += ms;
            b->n_solves++;
        }
    }
    *done_out = lm.done;
    *chi_out = lm.chi_out;
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
    hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&b->e0);
    if (e == hipSuccess) e = hipEventCreate(&b->e1);
    for (hipEvent_t& ev : b->ev_solve)
        if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e == hipSuccess) e = hipHostMalloc((void**)&b->h_lm, sizeof(BaLm), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&b->h_lm_dev, b->h_lm, 0);
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
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (Buf* q : b->all()) q->release();
    if (b->h_lm) (void)hipHostFree(b->h_lm);
    if (b->h_abort) (void)hipHostFree(b->h_abort);
    for (hipEvent_t ev : b->ev_solve)
        if (ev) (void)hipEventDestroy(ev);
    if (b->h_po) (void)hipHostFree(b->h_po);
    if (b->e0) (void)hipEventDestroy(b->e0);
    if (b->e1) (void)hipEventDestroy(b->e1);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
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

int so_bundle_adjust(so_ba* b, const so_ba_problem* p, const so_ba_options* opt, const volatile uint8_t* stop,
                     float* Tcw_out, float* Xw_out, uint8_t* edge_outlier, double* edge_chi2, so_ba_info* info) {
    if (!b || !p || !opt || !Tcw_out || !Xw_out) return SO_ERR_INVALID_ARG;
    if (p->n_poses < 0 || p->n_points < 0 || p->n_edges < 0) return SO_ERR_INVALID_ARG;
    if ((p->n_poses > 0 && (!p->Tcw || !p->fixed || !p->intr)) || (p->n_points > 0 && !p->Xw) ||
        (p->n_edges > 0 && (!p->edge_pose || !p->edge_point || !p->obs || !p->inv_sigma2)))
        return SO_ERR_INVALID_ARG;
    for (int e = 0; e < p->n_edges; e++)
        if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses || p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points) {
            last_error_ref() = "edge references a vertex out of range";
            return SO_ERR_INVALID_ARG;
        }
    const double t_begin = now_ms();
    SO_HIP(hipSetDevice(b->device));
    b->solve_ms = 0.f;
    b->n_solves = 0;
    so_ba_info inf;
    memset(&inf, 0, sizeof(inf));
    Run r;
    r.b = b;
    r.stop = stop;
    Problem& P = r.P;
    P.n_poses = p->n_poses;
    P.n_points = p->n_points;
    P.n_edges = p->n_edges;
    P.fixed.assign(p->fixed, p->fixed + p->n_poses);

    std::vector<BaPose> h_pose((size_t)P.n_poses);
    for (int i = 0; i < P.n_poses; i++) pose_from_Tcw(p->Tcw + 12 * (size_t)i, h_pose[(size_t)i]);  // toSE3Quat
    std::vector<double> h_pt((size_t)P.n_points * 3);
    for (size_t i = 0; i < h_pt.size(); i++) h_pt[i] = (double)p->Xw[i];  // toVector3d

    auto finish_untouched = [&]() {  // Optimizer.cc:631-633: return before optimising
        for (int i = 0; i < P.n_poses; i++) pose_to_Tcw(h_pose[(size_t)i], Tcw_out + 12 * (size_t)i);
        for (size_t i = 0; i < h_pt.size(); i++) Xw_out[i] = (float)h_pt[i];
        if (edge_outlier) memset(edge_outlier, 0, (size_t)P.n_edges);
        if (edge_chi2) for (int e = 0; e < P.n_edges; e++) edge_chi2[e] = 0.0;
        inf.wall_ms = (float)(now_ms() - t_begin);
        if (info) *info = inf;
    };
    if (r.terminate()) {
        inf.aborted = 1;
        finish_untouched();
        return SO_OK;
    }
    if (P.n_edges == 0) {
        finish_untouched();
        return SO_OK;
    }

    // stable counting sort of the edges by landmark: a landmark's observations become contiguous
    P.pt_off.assign((size_t)P.n_points + 1, 0);
    for (int e = 0; e < P.n_edges; e++) P.pt_off[(size_t)p->edge_point[e] + 1]++;
    for (int i = 0; i < P.n_points; i++) P.pt_off[(size_t)i + 1] += P.pt_off[(size_t)i];
    P.perm.assign((size_t)P.n_edges, 0);
    {
        std::vector<int> fill(P.pt_off.begin(), P.pt_off.end() - 1);
        for (int e = 0; e < P.n_edges; e++) P.perm[(size_t)fill[(size_t)p->edge_point[e]]++] = e;
    }
    P.e_pose.resize((size_t)P.n_edges);
    P.e_point.resize((size_t)P.n_edges);
    std::vector<double> h_obs((size_t)P.n_edges * 2), h_w((size_t)P.n_edges);
    for (int k = 0; k < P.n_edges; k++) {
        const int e = P.perm[(size_t)k];
        P.e_pose[(size_t)k] = p->edge_pose[e];
        P.e_point[(size_t)k] = p->edge_point[e];
        h_obs[2 * (size_t)k] = (double)p->obs[2 * (size_t)e];
        h_obs[2 * (size_t)k + 1] = (double)p->obs[2 * (size_t)e + 1];
        h_w[(size_t)k] = (double)p->inv_sigma2[e];
    }
    P.level.assign((size_t)P.n_edges, 0);
    std::vector<double> h_intr((size_t)P.n_poses * 4);
    for (size_t i = 0; i < h_intr.size(); i++) h_intr[i] = (double)p->intr[i];

    const bool trace = getenv("SWARMORB_BA_TRACE") != nullptr;
    const double t_sorted = now_ms();
    build_stage(P, r.S);
    const double t_stage1 = now_ms();
    if (6 * r.S.n_free > kMaxReducedDim) {
        last_error_ref() = "reduced camera system too large for the dense solver (6*n_free > 6144)";
        return SO_ERR_CAPACITY;
    }

    int rc;
    hipStream_t s = b->stream;
    if ((rc = upload(b, b->d_pose[0], h_pose))) return rc;
    if ((rc = b->d_pose[1].ensure(sizeof(BaPose) * std::max<size_t>(h_pose.size(), 1)))) return rc;
    if ((rc = upload(b, b->d_pt[0], h_pt))) return rc;
    if ((rc = b->d_pt[1].ensure(sizeof(double) * std::max<size_t>(h_pt.size(), 1)))) return rc;
    if ((rc = upload(b, b->d_intr, h_intr))) return rc;
    if ((rc = upload(b, b->d_epose, P.e_pose))) return rc;
    if ((rc = upload(b, b->d_ept, P.e_point))) return rc;
    if ((rc = upload(b, b->d_obs, h_obs))) return rc;
    if ((rc = upload(b, b->d_w, h_w))) return rc;
    if ((rc = upload(b, b->d_ptoff, P.pt_off))) return rc;
    const size_t nE = (size_t)P.n_edges, nL = (size_t)std::max(P.n_points, 1);
    if ((rc = b->d_err.ensure(sizeof(double) * 2 * nE))) return rc;
    if ((rc = b->d_chi2.ensure(sizeof(double) * nE))) return rc;
    if ((rc = b->d_depth.ensure(sizeof(double) * nE))) return rc;
    if ((rc = b->d_Hll.ensure(sizeof(double) * 9 * nL))) return rc;
    if ((rc = b->d_bl.ensure(sizeof(double) * 3 * nL))) return rc;
    if ((rc = b->d_Dinv.ensure(sizeof(double) * 9 * nL))) return rc;
    if ((rc = b->d_db.ensure(sizeof(double) * 3 * nL))) return rc;
    if ((rc = b->d_xl.ensure(sizeof(double) * 3 * nL))) return rc;
    if ((rc = b->d_W.ensure(sizeof(double) * 18 * nE))) return rc;
    if ((rc = b->d_BDinv.ensure(sizeof(double) * 18 * nE))) return rc;
    if ((rc = b->d_partial.ensure(sizeof(double) * kBaPartialCount))) return rc;
    SO_HIP(hipMemsetAsync(b->d_err.p, 0, sizeof(double) * 2 * nE, s));   // _error of a fresh edge
    SO_HIP(hipMemsetAsync(b->d_chi2.p, 0, sizeof(double) * nE, s));
    SO_HIP(hipMemsetAsync(b->d_partial.p, 0, sizeof(double) * kBaPartialCount, s));

    BaDev& d = r.d;
    d.n_poses = P.n_poses;
    d.n_points = P.n_points;
    d.n_edges = P.n_edges;
    d.intr = b->d_intr.as<double>();
    d.e_pose = b->d_epose.as<int>();
    d.e_point = b->d_ept.as<int>();
    d.e_obs = b->d_obs.as<double>();
    d.e_w = b->d_w.as<double>();
    d.e_err = b->d_err.as<double>();
    d.e_chi2 = b->d_chi2.as<double>();
    d.pt_off = b->d_ptoff.as<int>();
    d.Hll = b->d_Hll.as<double>();
    d.bl = b->d_bl.as<double>();
    d.W = b->d_W.as<double>();
    d.Dinv = b->d_Dinv.as<double>();
    d.db = b->d_db.as<double>();
    d.BDinv = b->d_BDinv.as<double>();
    d.xl = b->d_xl.as<double>();
    d.partial = b->d_partial.as<double>();
    d.robust = opt->robust;
    d.huber_delta = (double)opt->huber_delta;
    d.huber_dsqr = (float)((double)opt->huber_delta * (double)opt->huber_delta);  // RobustKernelHuber::setDelta
    r.nb_err = std::min(1024, std::max(1, (P.n_edges + 255) / 256));
    r.nb_upd = std::min(1024, std::max(1, (8 * P.n_points + P.n_poses + 255) / 256));
    if ((rc = upload_stage(r))) return rc;

    if ((rc = b->d_lm.ensure(sizeof(BaLm)))) return rc;
    SO_HIP(hipMemsetAsync(b->d_lm.p, 0, sizeof(BaLm), s));  // current estimate = buffer 0, no trials yet
    memset(b->h_lm, 0, sizeof(BaLm));
    *b->h_abort = 0;
    d.lm = b->d_lm.as<BaLm>();
    d.pose[0] = b->d_pose[0].as<BaPose>();
    d.pose[1] = b->d_pose[1].as<BaPose>();
    d.pt[0] = b->d_pt[0].as<double>();
    d.pt[1] = b->d_pt[1].as<double>();

    const double t_uploaded = now_ms();
    SO_HIP(hipEventRecord(b->e0, s));
    double chi = 0.0;
    int done = 0;
    if ((rc = optimize(r, opt->its_stage1, &done, &chi))) return rc;  // optimizer.optimize(5)
    inf.iterations_stage1 = done;
    inf.chi2_initial = b->h_lm->chi_begin;  // chi2 before optimising (information only)
    inf.chi2_final = done > 0 ? chi : inf.chi2_initial;
    const double t_opt1 = now_ms();
    bool do_more = opt->its_stage2 > 0;
    if (r.terminate()) {
        do_more = false;
        inf.aborted = 1;
    }
    if (do_more) {
        // Optimizer.cc:644-656 without leaving the device: outlier edges drop to level 1, the robust kernel goes,
        // initializeOptimization(0) = the same CSR lists with the dropped edges skipped
        launch_ba_mark_outliers(r.d, (double)opt->chi2_threshold, s);
        r.d.robust = 0;  // e->setRobustKernel(nullptr)
        if ((rc = optimize(r, opt->its_stage2, &done, &chi))) return rc;  // optimizer.optimize(10)
        inf.iterations_stage2 = done;
        if (done > 0) inf.chi2_final = chi;
        if (r.terminate()) inf.aborted = 1;
    }
    const double t_opt2 = now_ms();
    // Optimizer.cc:682-739: outlier flags from the edges' stored errors, then recover the optimised data
    const int cur = b->h_lm->cur;
    std::vector<double> h_chi2(nE), h_depth(nE);
    launch_ba_depth(r.d, b->d_depth.as<double>(), s);  // isDepthPositive()
    SO_HIP(hipMemcpyAsync(h_chi2.data(), b->d_chi2.p, sizeof(double) * nE, hipMemcpyDeviceToHost, s));
    SO_HIP(hipMemcpyAsync(h_depth.data(), b->d_depth.p, sizeof(double) * nE, hipMemcpyDeviceToHost, s));
    SO_HIP(hipMemcpyAsync(h_pose.data(), b->d_pose[cur].p, sizeof(BaPose) * h_pose.size(), hipMemcpyDeviceToHost, s));
    if (!h_pt.empty())
        SO_HIP(hipMemcpyAsync(h_pt.data(), b->d_pt[cur].p, sizeof(double) * h_pt.size(), hipMemcpyDeviceToHost, s));
    SO_HIP(hipEventRecord(b->e1, s));
    SO_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < P.n_edges; k++) {
        const int e = P.perm[(size_t)k];
        const int out = (h_chi2[(size_t)k] > (double)opt->chi2_threshold || !(h_depth[(size_t)k] > 0.0)) ? 1 : 0;
        if (edge_outlier) edge_outlier[e] = (uint8_t)out;
        if (edge_chi2) edge_chi2[e] = h_chi2[(size_t)k];
        inf.n_outliers += out;
    }
    for (int i = 0; i < P.n_poses; i++) pose_to_Tcw(h_pose[(size_t)i], Tcw_out + 12 * (size_t)i);
    for (size_t i = 0; i < h_pt.size(); i++) Xw_out[i] = (float)h_pt[i];
    inf.lambda_final = b->h_lm->lambda;
    inf.lm_trials = b->h_lm->trials;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, b->e0, b->e1) == hipSuccess) inf.gpu_ms = ms;
    inf.solve_ms = b->solve_ms;
    inf.n_solves = b->n_solves;
    inf.wall_ms = (float)(now_ms() - t_begin);
    if (trace)
        fprintf(stderr, "[ba] sort %.3f lists %.3f upload %.3f opt1 %.3f (%d it) opt2 %.3f (%d it) finish %.3f | trials %d blocks %d\n",
                t_sorted - t_begin, t_stage1 - t_sorted, t_uploaded - t_stage1, t_opt1 - t_uploaded, inf.iterations_stage1,
                t_opt2 - t_opt1, inf.iterations_stage2, now_ms() - t_opt2, inf.lm_trials, r.blocks_enqueued);
    if (info) *info = inf;
    return SO_OK;
}

// Optimizer::PoseOptimization — code/src/Optimizer.cc:239-434
int so_pose_optimization(so_ba* b, const float* Tcw12, const float* intr, int32_t n, const float* Xw, const float* obs,
                         const float* inv_sigma2, float* Tcw_out12, uint8_t* outlier, int32_t* n_inliers,
                         int32_t* info) {
    if (!b || !Tcw12 || !intr || n < 0 || !Tcw_out12 || !n_inliers) return SO_ERR_INVALID_ARG;
    if (n > 0 && (!Xw || !obs || !inv_sigma2 || !outlier)) return SO_ERR_INVALID_ARG;
    *n_inliers = 0;
    if (info) info[0] = info[1] = 0;
    if (n < 3) return SO_OK;  // :344-345, nothing is touched
    SO_HIP(hipSetDevice(b->device));
    hipStream_t s = b->stream;
    // one pinned staging block: [Xw 12n | obs 8n | w 4n] in, [pose 64 | info 16 | outlier n] out
    const size_t in_bytes = (size_t)n * 24, out_bytes = 64 + 16 + (size_t)n;
    const size_t need = in_bytes + out_bytes + 64;
    if (need > b->h_po_cap) {
        if (b->h_po) SO_HIP(hipHostFree(b->h_po));
        b->h_po = nullptr;
        b->h_po_cap = 0;
        SO_HIP(hipHostMalloc((void**)&b->h_po, need * 2, hipHostMallocDefault));
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
    SO_HIP(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    PoseOptArgs a;
    a.Xw = reinterpret_cast<const float*>(d);
    a.obs = reinterpret_cast<const float*>(d + (size_t)n * 12);
    a.inv_sigma2 = reinterpret_cast<const float*>(d + (size_t)n * 20);
    for (int k = 0; k < 4; k++) a.K[k] = (double)intr[k];
    pose_from_Tcw(Tcw12, a.init);  // Converter::toSE3Quat(pFrame->mTcw)
    a.n = n;
    a.err = reinterpret_cast<double*>(d + off_err);
    a.pose_out = reinterpret_cast<BaPose*>(d + off_pose);
    a.info = reinterpret_cast<int*>(d + off_info);
    a.outlier = d + off_out;
    a.trace = getenv("SWARMORB_POSE_TRACE") ? reinterpret_cast<double*>(d + off_trace) : nullptr;
    launch_pose_opt(a, s);
    SO_HIP(hipGetLastError());
    uint8_t* hout = h + in_bytes;
    SO_HIP(hipMemcpyAsync(hout, d + off_pose, 64 + 16 + (size_t)n, hipMemcpyDeviceToHost, s));
    SO_HIP(hipStreamSynchronize(s));
    BaPose P;
    memcpy(&P, hout, sizeof(BaPose));
    int inf[4];
    memcpy(inf, hout + 64, 16);
    memcpy(outlier, hout + 80, (size_t)n);
    if (a.trace) {  // debugging aid: dump the LM trial log
        std::vector<double> tr(4 * 256);
        SO_HIP(hipMemcpy(tr.data(), a.trace, sizeof(double) * tr.size(), hipMemcpyDeviceToHost));
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

}  // extern "C"
