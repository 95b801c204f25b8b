// closedloop.cc — the closed tracking + local-mapping loop of the replay harness: what a keyframe's local-mapping job
// computes flows back into the map the next frames are tracked against (code/src/LocalMapping.cc:53-110 LocalMapping::Run;
// code/src/Optimizer.cc:436-560 the window, :713-739 the write-back; Tracking::UpdateLastFrame / CheckReplacedInLastFrame,
// code/src/Tracking.cc:603-614, 656-662).  swarmmap_amd/closedloop.py is the same logic in Python, statement by statement: its module
// text says what is kept of the reference and what the harness simplifies; tests compare the two chains frame by frame and
// keyframe by keyframe (tests/test_closedloop_gpu.py).
//
// Threads: the local-mapping thread owns the map model (points, keyframes, observations both ways) and issues every
// operator call of a keyframe - one so_matcher batch for CreateNewMapPoints' searches, one launch for the triangulation of
// all their matches, one batch for SearchInNeighbors' Fuse calls over the HBM-resident keyframes and the HBM-resident map
// table, so_bundle_adjust over the keyframe's own window, so_update_normal_and_depth for the points it moved; the
// tracking thread owns its view of the map (positions, bad / replaced flags, the local map's slots) and applies the
// job's packet between two frames (what Optimizer.cc:713 does under Map::mMutexMapUpdate).  New rows of the device map are
// appended by the local-mapping thread (tracking does not look beyond its own count until the packet arrives); rows local
// BA moved are rewritten by the tracking thread when it applies the packet.
#include "replay_internal.h"

namespace {

constexpr int kThCovisible = 15;  // KeyFrame::UpdateConnections th (code/src/KeyFrame.cc:512-525)

struct Lm {  // shorthand over the local-mapping side of ClosedLoop
    ClosedLoop& M;
    explicit Lm(ClosedLoop& m) : M(m) {}
    int n_points() const { return (int)M.bad.size(); }
    int resolve(int s) const {  // the live point a binding stands for (mpReplaced chain)
        while (s >= 0 && M.bad[(size_t)s]) s = M.repl[(size_t)s];
        return s;
    }
    bool in_kf(int s, int kf) const {
        for (const auto& o : M.obs[(size_t)s])
            if (o.first == kf) return true;
        return false;
    }
    void set_bad(int s) {  // MapPoint::SetBadFlag
        for (const auto& o : M.obs[(size_t)s]) M.kfs[(size_t)o.first]->mp[(size_t)o.second] = -1;
        M.obs[(size_t)s].clear();
        M.bad[(size_t)s] = 1;
        M.repl[(size_t)s] = -1;
        M.newly_bad.push_back(s);
    }
    void replace(int a, int b) {  // MapPoint::Replace: a is replaced by b
        if (a == b) return;
        std::vector<std::pair<int32_t, int32_t>> obs;
        obs.swap(M.obs[(size_t)a]);
        M.bad[(size_t)a] = 1;
        M.repl[(size_t)a] = b;
        M.newly_bad.push_back(a);
        for (const auto& o : obs) {
            if (!in_kf(b, o.first)) {
                M.kfs[(size_t)o.first]->mp[(size_t)o.second] = b;
                M.obs[(size_t)b].push_back(o);
            } else {
                M.kfs[(size_t)o.first]->mp[(size_t)o.second] = -1;
            }
        }
    }
    void erase_observation(int s, int kf, int idx) {  // KeyFrame::EraseMapPointMatch + MapPoint::EraseObservation
        M.kfs[(size_t)kf]->mp[(size_t)idx] = -1;
        auto& o = M.obs[(size_t)s];
        o.erase(std::remove(o.begin(), o.end(), std::make_pair((int32_t)kf, (int32_t)idx)), o.end());
        if (M.ref_kf[(size_t)s] == kf && !o.empty()) M.ref_kf[(size_t)s] = o[0].first;
        if (o.size() <= 2) set_bad(s);
    }
    int append(const float* X, const float* N, float mx, float mn, const uint8_t* D, int kf) {
        const int s = n_points();
        M.X.insert(M.X.end(), X, X + 3);
        M.N.insert(M.N.end(), N, N + 3);
        M.mx.push_back(mx);
        M.mn.push_back(mn);
        M.D.insert(M.D.end(), D, D + 32);
        M.bad.push_back(0);
        M.repl.push_back(-1);
        M.ref_kf.push_back(kf);
        M.first_kf.push_back(kf);
        M.obs.emplace_back();
        return s;
    }
    int next_stamp() {
        if (M.stamp.size() < M.bad.size()) M.stamp.resize(M.bad.size() + M.bad.size() / 2 + 1024, -1);
        return ++M.stamp_id;
    }
};

void centre(const float* T, double* Ow) {  // -R^T t of a float pose, in double
    for (int q = 0; q < 3; q++) Ow[q] = -((double)T[q] * T[3] + (double)T[4 + q] * T[7] + (double)T[8 + q] * T[11]);
}

// the gate of CreateNewMapPoints (LocalMapping.cc:228-241): baseline / median depth of the neighbour < 0.01, the median
// depth as KeyFrame::ComputeSceneMedianDepth(2) has it (element [(n - 1) / 2] of the sorted depths; -1 without points).  The median
// is at most the largest depth, so a baseline of 1 % of THAT passes without the selection (same decision, a fifth of the time)
bool baseline_too_short(const ClosedLoop& M, const KfSnap& k2, double baseline, std::vector<double>& z) {
    z.clear();
    const float* T = k2.T;
    double zmax = 0.0;
    for (int i = 0; i < k2.n; i++) {
        const int s = k2.mp[(size_t)i];
        if (s < 0) continue;
        const float* X = &M.X[3 * (size_t)s];
        const double d = (double)T[8] * (double)X[0] + (double)T[9] * (double)X[1] + (double)T[10] * (double)X[2] + (double)T[11];
        z.push_back(d);
        if (d > zmax) zmax = d;
    }
    if (z.empty()) return baseline / -1.0 < 0.01;
    if (zmax > 0.0 && baseline / zmax >= 0.01) {
        // median <= zmax; a non-positive median (most points behind the camera) fails the gate the reference's way - only
        // possible when at least half of the depths are <= 0
        size_t nonpos = 0;
        for (double d : z) nonpos += d <= 0.0 ? 1 : 0;
        if (2 * nonpos < z.size()) return false;
    }
    const size_t mid = (z.size() - 1) / 2;
    std::nth_element(z.begin(), z.begin() + (long)mid, z.end());
    return baseline / z[mid] < 0.01;
}

struct Window {
    std::vector<int32_t> kf;     // window keyframes, ascending id
    std::vector<uint8_t> fixed;
    std::vector<int32_t> pts;    // point slots, ascending
    std::vector<int32_t> e_kf, e_idx, e_pt;
    std::vector<float> Tcw, intr, Xw, obs, w;
    std::vector<int32_t> e_pose;
};

// The window Optimizer::LocalBundleAdjustment gathers for keyframe c (closedloop.local_window)
bool local_window(so_replay* r, ClosedLoop& M, const KfSnap& c, Window& W) {
    Lm L(M);
    const int k = c.id, nkf = (int)M.kfs.size();
    std::vector<int32_t> share((size_t)nkf, 0);
    for (int i = 0; i < c.n; i++) {
        const int s = c.mp[(size_t)i];
        if (s < 0) continue;
        for (const auto& o : M.obs[(size_t)s]) share[(size_t)o.first]++;
    }
    std::vector<int32_t> cand;
    for (int kf = 0; kf < nkf; kf++)
        if (kf != k && share[(size_t)kf] >= kThCovisible) cand.push_back(kf);
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return share[(size_t)a] != share[(size_t)b] ? share[(size_t)a] > share[(size_t)b] : a > b; });
    std::vector<int32_t> local(1, k);
    for (size_t i = 0; i < cand.size() && (int)local.size() < M.n_free; i++) local.push_back(cand[i]);
    std::vector<uint8_t> is_local((size_t)nkf, 0);
    for (int kf : local) is_local[(size_t)kf] = 1;
    const int id = L.next_stamp();
    std::vector<int32_t> pts;
    for (int kf : local) {
        const KfSnap& q = *M.kfs[(size_t)kf];
        for (int i = 0; i < q.n; i++) {
            const int s = q.mp[(size_t)i];
            if (s >= 0 && M.stamp[(size_t)s] != id) {
                M.stamp[(size_t)s] = id;
                pts.push_back(s);
            }
        }
    }
    std::sort(pts.begin(), pts.end());
    // ONE walk over the points' observation lists (a heap block per point: the cache misses of this function) into flat
    // arrays, counting the observers outside the local set on the way; everything after reads the flat arrays in order
    static thread_local std::vector<int32_t> t_kf, t_idx, t_first;
    size_t cap = 0;
    for (int s : pts) cap += M.obs[(size_t)s].size();
    t_kf.resize(cap); t_idx.resize(cap); t_first.resize(pts.size() + 1);
    std::vector<int32_t> count((size_t)nkf, 0);
    {
        size_t at = 0;
        for (size_t q = 0; q < pts.size(); q++) {
            t_first[q] = (int32_t)at;
            for (const auto& o : M.obs[(size_t)pts[q]]) {
                t_kf[at] = o.first;
                t_idx[at] = o.second;
                at++;
                if (!is_local[(size_t)o.first]) count[(size_t)o.first]++;
            }
        }
        t_first[pts.size()] = (int32_t)at;
    }
    std::vector<int32_t> fx;
    for (int kf = 0; kf < nkf; kf++)
        if (count[(size_t)kf] > 0) fx.push_back(kf);
    std::sort(fx.begin(), fx.end(), [&](int a, int b) { return count[(size_t)a] != count[(size_t)b] ? count[(size_t)a] > count[(size_t)b] : a > b; });
    if ((int)fx.size() > M.n_fixed) fx.resize((size_t)M.n_fixed);
    W.kf = local;
    W.kf.insert(W.kf.end(), fx.begin(), fx.end());
    std::sort(W.kf.begin(), W.kf.end());
    std::vector<int32_t> row((size_t)nkf, -1);
    for (size_t p = 0; p < W.kf.size(); p++) row[(size_t)W.kf[p]] = (int32_t)p;
    // the edges of a point are written as they are found (keypoint coordinates and weights straight from the keyframes'
    // arrays) and taken back if there are fewer than two
    W.fixed.resize(W.kf.size());
    for (size_t p = 0; p < W.kf.size(); p++) W.fixed[p] = (is_local[(size_t)W.kf[p]] && W.kf[p] != 0) ? 0 : 1;
    std::vector<const float*> kx((size_t)nkf, nullptr), ky((size_t)nkf, nullptr);
    std::vector<const int32_t*> ko((size_t)nkf, nullptr);
    for (int kf : W.kf) {
        const KfSnap& q = *M.kfs[(size_t)kf];
        kx[(size_t)kf] = q.x.data(); ky[(size_t)kf] = q.y.data(); ko[(size_t)kf] = q.octave.data();
    }
    W.pts.clear();
    W.e_kf.resize(cap); W.e_idx.resize(cap); W.e_pt.resize(cap); W.e_pose.resize(cap); W.obs.resize(2 * cap); W.w.resize(cap);
    size_t ne = 0;
    bool any_free = false;
    for (size_t q = 0; q < pts.size(); q++) {
        const size_t e0 = ne;
        bool fr = false;
        for (int32_t a = t_first[q]; a < t_first[q + 1]; a++) {
            const int kf = t_kf[(size_t)a], idx = t_idx[(size_t)a];
            const int p = row[(size_t)kf];
            if (p < 0) continue;
            W.e_kf[ne] = kf;
            W.e_idx[ne] = idx;
            W.e_pt[ne] = (int32_t)W.pts.size();
            W.e_pose[ne] = p;
            W.obs[2 * ne] = kx[(size_t)kf][idx];
            W.obs[2 * ne + 1] = ky[(size_t)kf][idx];
            W.w[ne] = r->inv_sigma2[ko[(size_t)kf][idx]];
            fr = fr || W.fixed[(size_t)p] == 0;
            ne++;
        }
        if (ne - e0 < 2) {
            ne = e0;
            continue;
        }
        any_free = any_free || fr;
        W.pts.push_back(pts[q]);
    }
    W.e_kf.resize(ne); W.e_idx.resize(ne); W.e_pt.resize(ne); W.e_pose.resize(ne); W.obs.resize(2 * ne); W.w.resize(ne);
    if (W.pts.size() < 10 || !any_free) return false;
    const size_t np = W.kf.size(), nl = W.pts.size();
    W.Tcw.resize(12 * np); W.intr.resize(4 * np); W.Xw.resize(3 * nl);
    for (size_t p = 0; p < np; p++) {
        memcpy(&W.Tcw[12 * p], M.kfs[(size_t)W.kf[p]]->T, 48);
        W.intr[4 * p] = r->cam.fx; W.intr[4 * p + 1] = r->cam.fy; W.intr[4 * p + 2] = r->cam.cx; W.intr[4 * p + 3] = r->cam.cy;
    }
    for (size_t q = 0; q < nl; q++) memcpy(&W.Xw[3 * q], &M.X[3 * (size_t)W.pts[q]], 12);
    return true;
}

}  // namespace

// KeyFrame::ComputeBoW stand-in (node = nearest of the vocabulary's centroid descriptors, lowest index on ties) and the
// keyframe's upload into HBM (so_kframe_create: both candidate layouts), on whichever thread's matcher handle `m` is
int cl_keyframe_featvec_upload(so_replay* r, so_matcher* m, KfSnap& c) {
    const int n = c.n, nv = (int)(r->vocab.size() / 32);
    std::vector<int32_t> node((size_t)n), bd((size_t)n), sd((size_t)n);
    if (n > 0 && so_hamming_top2(m, c.desc.data(), n, r->vocab.data(), nv, node.data(), bd.data(), sd.data()) != SO_OK) return SO_ERR_HIP;
    std::vector<int32_t> count((size_t)nv + 1, 0);
    for (int i = 0; i < n; i++) count[(size_t)node[(size_t)i] + 1]++;
    for (int v = 0; v < nv; v++) count[(size_t)v + 1] += count[(size_t)v];
    std::vector<int32_t> pos(count.begin(), count.end() - 1), by_node((size_t)n);
    for (int i = 0; i < n; i++) by_node[(size_t)pos[(size_t)node[(size_t)i]]++] = i;
    c.node_id.clear(); c.idx.clear();
    c.off.assign(1, 0);
    for (int v = 0; v < nv; v++)
        if (count[(size_t)v + 1] > count[(size_t)v]) {
            c.node_id.push_back(v);
            for (int a = count[(size_t)v]; a < count[(size_t)v + 1]; a++) c.idx.push_back(by_node[(size_t)a]);
            c.off.push_back((int32_t)c.idx.size());
        }
    const so_featvec fv{(int32_t)c.node_id.size(), c.node_id.data(), c.off.data(), c.idx.data()};
    float level_sigma2[8];
    for (int l = 0; l < 8; l++) level_sigma2[l] = r->scale[l] * r->scale[l];
    const so_frame_view V = keyframe_view(r, c);
    return so_kframe_create(m, &V, &fv, level_sigma2, &c.dev) == SO_OK ? SO_OK : SO_ERR_HIP;
}

// One keyframe through local mapping (closedloop.lm_job).  Local-mapping thread.
int cl_lm_job(so_replay* r, const std::shared_ptr<KfSnap>& c, bool timed, so_ba_info* info_out) {
    ClosedLoop& M = *r->cl;
    Lm L(M);
    so_matcher* m = r->mapper_matcher;
    const double t0 = now_ms();
    double st[40] = {0};
    float kms = 0.f;
    so_matcher_set_profiling(m, 1);
    const int k = (int)M.kfs.size(), n = c->n;
    c->id = k;
    M.kfs.push_back(c);
    if (k == 0) {  // the initial map travels with keyframe 0 (its keypoints back-projected by the tracking thread)
        for (int i = 0; i < n; i++)
            L.append(&c->mpX[3 * (size_t)i], &c->mpN[3 * (size_t)i], c->mpMax[(size_t)i], c->mpMin[(size_t)i], &c->mpDesc[32 * (size_t)i], 0);
    }
    const int n_before = L.n_points();
    M.newly_bad.clear();
    // ---- ProcessNewKeyFrame (LocalMapping.cc:134-172): the tracked bindings become observations
    for (int i = 0; i < n; i++) {
        int s = c->mp[(size_t)i];
        if (s < 0) continue;
        s = L.resolve(s);
        if (s < 0 || L.in_kf(s, k)) {
            c->mp[(size_t)i] = -1;
            continue;
        }
        c->mp[(size_t)i] = s;
        M.obs[(size_t)s].emplace_back(k, i);
    }
    // ---- MapPointCulling (:174-205): the found / visible ratio Tracking counted, then the observation rule
    {
        std::vector<int32_t> keep;
        for (int s : M.recent) {
            if (M.bad[(size_t)s]) continue;
            const int q = s - c->cnt_from;
            if (q >= 0 && q < (int)c->cnt_vis.size() && (float)c->cnt_found[(size_t)q] / (float)c->cnt_vis[(size_t)q] < 0.25f)
                L.set_bad(s);  // GetFoundRatio() < 0.25f (:177-178)
            else if (k - M.first_kf[(size_t)s] >= 2 && M.obs[(size_t)s].size() <= 2) L.set_bad(s);
            else if (k - M.first_kf[(size_t)s] >= 3) continue;
            else keep.push_back(s);
        }
        M.recent.swap(keep);
    }
    st[24] = now_ms() - t0;  // process + culling
    // ---- KeyFrame::ComputeBoW stand-in + upload of the keyframe: done by the tracking thread when it made the keyframe (under
    //      its last PoseOptimization kernel, where it only waits); here only for a keyframe that came without
    const double tn0 = now_ms();
    if (!c->dev && cl_keyframe_featvec_upload(r, m, *c) != SO_OK) return SO_ERR_HIP;
    const so_featvec fv1{(int32_t)c->node_id.size(), c->node_id.data(), c->off.data(), c->idx.data()};
    float level_sigma2[8];
    for (int l = 0; l < 8; l++) level_sigma2[l] = r->scale[l] * r->scale[l];
    st[kLmNodeMs] = now_ms() - tn0;
    const int r0 = std::max(0, k - r->lm_neighbours);
    const int nn = k - r0;
    auto ring = [&](int j) -> KfSnap& { return *M.kfs[(size_t)(r0 + j)]; };
    // ---- CreateNewMapPoints (:207-420): SearchForTriangulation against the neighbours that pass the baseline / median-depth
    //      gate (:228-241) as ONE batch, the matches of all of them triangulated in ONE launch, then new points in the
    //      reference's order (neighbour by neighbour, keypoint by keypoint; a keypoint takes the first point it gets)
    int64_t n_tri = 0, n_new = 0, n_fused = 0, n_back = 0;
    const double tc0 = now_ms();
    std::vector<std::vector<int32_t>> tri_m12((size_t)nn);
    std::vector<int32_t> tri_nm((size_t)nn, 0);
    std::vector<uint8_t> searched((size_t)nn, 0);
    std::vector<std::vector<uint8_t>> free2((size_t)nn);
    std::vector<uint8_t> free1((size_t)n);
    for (int i = 0; i < n; i++) free1[(size_t)i] = c->mp[(size_t)i] < 0 ? 1 : 0;
    {
        double Oc[3];
        centre(c->T, Oc);
        std::vector<double> z;
        // one call for all neighbours that pass the gate: both sides resident, the queries built on the device
        std::vector<so_tri_neighbour> nbs;
        for (int j = 0; j < nn; j++) {
            KfSnap& k2 = ring(j);
            double O2[3];
            centre(k2.T, O2);
            const double d0 = O2[0] - Oc[0], d1 = O2[1] - Oc[1], d2 = O2[2] - Oc[2];
            const double baseline = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            if (baseline_too_short(M, k2, baseline, z)) continue;
            so_tri_neighbour N;
            memset(&N, 0, sizeof(N));
            fundamental_and_epipole(r, c->T, k2.T, N.F12, &N.ex, &N.ey);
            free2[(size_t)j].resize((size_t)k2.n);
            for (int i = 0; i < k2.n; i++) free2[(size_t)j][(size_t)i] = k2.mp[(size_t)i] < 0 ? 1 : 0;
            tri_m12[(size_t)j].resize((size_t)n);
            N.kf2 = k2.dev;
            N.free2 = free2[(size_t)j].data();
            N.matches12 = tri_m12[(size_t)j].data();
            N.nmatches = &tri_nm[(size_t)j];
            nbs.push_back(N);
            searched[(size_t)j] = 1;
            st[kLmTriCalls] += 1;
        }
        if (!nbs.empty()) {
            const double tbe = now_ms();
            // mbCheckOrientation = false: CreateNewMapPoints' matcher is ORBmatcher(0.6, false), LocalMapping.cc:197 - no rotation
            // histogram, so a feature's match depends on nothing the other features do, and the searches of all neighbours against
            // one snapshot + creation in the reference's order below ARE the reference's neighbour-by-neighbour loop (:219-416;
            // swarmmap_amd/closedloop.py, tests/test_closedloop_oracle.py::test_batched_searches_equal_the_reference_interleaving)
            if (so_search_for_triangulation_kframes(m, c->dev, free1.data(), (int32_t)nbs.size(), nbs.data(), 0) != SO_OK) return SO_ERR_HIP;
            st[36] = now_ms() - tbe;  // staging of all neighbours + launch + wait + resolve
            double ms4[4] = {0};
            so_matcher_last_stats(m, ms4);
            st[kLmBatchEnqueueMs] += ms4[0];
            st[kLmBatchWaitMs] += ms4[1];
            so_matcher_last_kernel_ms(m, &kms);
            st[kLmBatchKernelMs] += kms;
        }
    }
    st[kLmStageTriMs] = now_ms() - tc0;
    {
        const double ta = now_ms();
        std::vector<int32_t>&of = r->lm_tof, &o1 = r->lm_to1, &o2 = r->lm_to2;
        std::vector<float>&p1 = r->lm_txy1, &p2 = r->lm_txy2;
        std::vector<int32_t> i1s, i2s;
        of.clear(); o1.clear(); o2.clear(); p1.clear(); p2.clear();
        for (int j = 0; j < nn; j++) {
            if (!searched[(size_t)j]) continue;
            const KfSnap& k2 = ring(j);
            const std::vector<int32_t>& m12 = tri_m12[(size_t)j];
            for (int i = 0; i < n; i++) {
                const int i2 = m12[(size_t)i];
                if (i2 < 0) continue;
                of.push_back(j);
                i1s.push_back(i); i2s.push_back(i2);
                p1.push_back(c->x[(size_t)i]); p1.push_back(c->y[(size_t)i]);
                o1.push_back(c->octave[(size_t)i]);
                p2.push_back(k2.x[(size_t)i2]); p2.push_back(k2.y[(size_t)i2]);
                o2.push_back(k2.octave[(size_t)i2]);
            }
        }
        const int nt = (int)of.size();
        if (nt > 0) {
            auto tri_kf = [r, ls = &level_sigma2[0]](const KfSnap& q) {
                so_tri_keyframe t;
                memset(&t, 0, sizeof(t));
                memcpy(t.Tcw, q.T, sizeof(t.Tcw));
                t.fx = r->cam.fx; t.fy = r->cam.fy; t.cx = r->cam.cx; t.cy = r->cam.cy;
                t.invfx = 1.0f / r->cam.fx; t.invfy = 1.0f / r->cam.fy;
                t.scale_factors = r->scale;
                t.level_sigma2 = ls;
                t.nlevels = r->nlevels;
                return t;
            };
            const so_tri_keyframe k1 = tri_kf(*c);
            std::vector<so_tri_keyframe> k2s;
            for (int j = 0; j < nn; j++) k2s.push_back(tri_kf(ring(j)));
            std::vector<uint8_t>& okv = r->lm_tok;
            std::vector<float>&X3 = r->lm_tX, &nrm = r->lm_nnrm, &mxd = r->lm_nmax, &mnd = r->lm_nmin;
            okv.assign((size_t)nt, 0);
            X3.assign(3 * (size_t)nt, 0.f); nrm.assign(3 * (size_t)nt, 0.f); mxd.assign((size_t)nt, 0.f); mnd.assign((size_t)nt, 0.f);
            const float ratio_factor = 1.5f * 1.2f;  // 1.5f * mpCurrentKeyFrame->mfScaleFactor, :214
            if (so_triangulate_new_points(m, &k1, (int32_t)k2s.size(), k2s.data(), ratio_factor, nt, of.data(), p1.data(), o1.data(), p2.data(),
                                          o2.data(), okv.data(), X3.data(), nrm.data(), mxd.data(), mnd.data()) != SO_OK)
                return SO_ERR_HIP;
            so_matcher_last_kernel_ms(m, &kms);
            st[kLmTriangKernelMs] = kms;
            for (int q = 0; q < nt; q++) {
                KfSnap& k2 = ring(of[(size_t)q]);
                const size_t i1 = (size_t)i1s[(size_t)q], i2 = (size_t)i2s[(size_t)q];
                // a keypoint bound against an earlier neighbour would not have been searched again (ORBmatcher.cc:638-641): its
                // later matches do not exist for the reference - neither as matches (the log's count) nor as points.  No test of
                // the neighbour's keypoint: two features matched to ONE keypoint of a neighbour both create their point, the second
                // AddMapPoint takes the neighbour's binding (LocalMapping.cc:403-416)
                if (c->mp[i1] >= 0) continue;
                n_tri++;
                if (!okv[(size_t)q]) continue;
                const int s = L.append(&X3[3 * (size_t)q], &nrm[3 * (size_t)q], mxd[(size_t)q], mnd[(size_t)q], &c->desc[32 * i1], k);
                M.obs[(size_t)s].emplace_back(k, (int32_t)i1);
                M.obs[(size_t)s].emplace_back(k2.id, (int32_t)i2);
                c->mp[i1] = s;
                k2.mp[i2] = s;
                M.recent.push_back(s);
                n_new++;
            }
            if (n_new > 0 && so_map_write(r->map, n_before, (int32_t)n_new, &M.X[3 * (size_t)n_before], &M.N[3 * (size_t)n_before],
                                          &M.mx[(size_t)n_before], &M.mn[(size_t)n_before], &M.D[32 * (size_t)n_before]) != SO_OK)
                return SO_ERR_HIP;
        }
        st[kLmTriangMs] = now_ms() - ta;
    }
    // ---- SearchInNeighbors (:423-498): Fuse into every neighbour and the neighbours' points into this keyframe as ONE batch
    //      over the resident keyframes and the resident map table; AddObservation / Replace in the reference's order
    if (nn > 0) {
        const double td0 = now_ms();
        std::vector<int32_t> cs(c->mp);
        std::vector<std::vector<int32_t>> best((size_t)nn + 1), dist((size_t)nn + 1);
        std::vector<std::vector<uint8_t>> valid((size_t)nn + 1);
        std::vector<int32_t> nf((size_t)nn + 1, 0);
        if (so_matcher_batch_begin(m) != SO_OK) return SO_ERR_HIP;
        int rc = SO_OK;
        for (int j = 0; j < nn && rc == SO_OK; j++) {
            KfSnap& k2 = ring(j);
            const int id = L.next_stamp();
            for (int i = 0; i < k2.n; i++)
                if (k2.mp[(size_t)i] >= 0) M.stamp[(size_t)k2.mp[(size_t)i]] = id;
            valid[(size_t)j].resize((size_t)n);
            for (int i = 0; i < n; i++) valid[(size_t)j][(size_t)i] = (cs[(size_t)i] >= 0 && M.stamp[(size_t)cs[(size_t)i]] != id) ? 1 : 0;
            best[(size_t)j].assign((size_t)n, -1); dist[(size_t)j].assign((size_t)n, 256);
            rc = so_fuse_kframe_map(m, k2.dev, &r->cam, k2.T, r->log_sf, r->inv_sigma2, r->map, n, cs.data(), valid[(size_t)j].data(), 3.0f,
                                    best[(size_t)j].data(), dist[(size_t)j].data(), &nf[(size_t)j], nullptr);
            st[kLmFuseCalls] += 1;
            st[kLmFusePoints] += n;
        }
        std::vector<int32_t>& cand = r->lm_cslot;
        cand.clear();
        if (rc == SO_OK) {
            const int job = L.next_stamp();
            for (int j = 0; j < nn; j++) {  // vpFuseCandidates, once each (mnFuseCandidateForKF)
                const KfSnap& k2 = ring(j);
                for (int i = 0; i < k2.n; i++) {
                    const int s = k2.mp[(size_t)i];
                    if (s < 0 || M.bad[(size_t)s] || M.stamp[(size_t)s] == job) continue;
                    M.stamp[(size_t)s] = job;
                    cand.push_back(s);
                }
            }
            const int cid = L.next_stamp();
            for (int i = 0; i < n; i++)
                if (c->mp[(size_t)i] >= 0) M.stamp[(size_t)c->mp[(size_t)i]] = cid;
            const size_t q = cand.size();
            valid[(size_t)nn].resize(q);
            for (size_t i = 0; i < q; i++) valid[(size_t)nn][i] = M.stamp[(size_t)cand[i]] != cid ? 1 : 0;
            best[(size_t)nn].assign(q, -1); dist[(size_t)nn].assign(q, 256);
            rc = so_fuse_kframe_map(m, c->dev, &r->cam, c->T, r->log_sf, r->inv_sigma2, r->map, (int32_t)q, cand.data(), valid[(size_t)nn].data(),
                                    3.0f, best[(size_t)nn].data(), dist[(size_t)nn].data(), &nf[(size_t)nn], nullptr);
            st[kLmFuseCalls] += 1;
            st[kLmFusePoints] += (double)q;
        }
        if (rc != SO_OK) {
            so_matcher_batch_abort(m);
            return SO_ERR_HIP;
        }
        st[kLmStageFuseMs] = now_ms() - td0;
        const double te0 = now_ms();
        if (so_matcher_batch_end(m) != SO_OK) return SO_ERR_HIP;
        {
            double ms4[4] = {0};
            so_matcher_last_stats(m, ms4);
            st[kLmBatchEnqueueMs] += ms4[0];
            st[kLmBatchWaitMs] += ms4[1];
            so_matcher_last_kernel_ms(m, &kms);
            st[25] = kms;  // the Fuse batch's kernels
        }
        st[kLmBatchEndMs] = now_ms() - te0;
        const double tf0 = now_ms();
        // ORBmatcher::Fuse's search reads nothing the loop over the targets changes; what changes is which points are still looked
        // at (isBad() || IsInKeyFrame(pKF), ORBmatcher.cc:778-784) and what sits at the keypoint found (:863-880): the batch searched
        // every pair valid on the snapshot (a superset: bad flags and observations only grow here), the walk below applies the
        // gates on the LIVE state in the reference's order = LocalMapping.cc:451-481 target by target
        auto apply_fuse = [&](KfSnap& target, const std::vector<int32_t>& slots, const std::vector<int32_t>& bst) {
            int64_t done = 0;
            for (size_t i = 0; i < bst.size(); i++) {
                if (bst[i] < 0) continue;
                const int p = slots[i];
                if (p < 0 || M.bad[(size_t)p] || L.in_kf(p, target.id)) continue;
                const size_t kp = (size_t)bst[i];
                const int q = target.mp[kp];
                if (q >= 0) {
                    if (!M.bad[(size_t)q]) {
                        if (M.obs[(size_t)q].size() > M.obs[(size_t)p].size()) L.replace(p, q);  // ORBmatcher.cc:873-878
                        else L.replace(q, p);
                    }
                } else {
                    M.obs[(size_t)p].emplace_back(target.id, (int32_t)kp);
                    target.mp[kp] = p;
                }
                done++;
            }
            return done;
        };
        for (int j = 0; j < nn; j++) n_fused += apply_fuse(ring(j), cs, best[(size_t)j]);
        n_back = apply_fuse(*c, cand, best[(size_t)nn]);
        st[26] = now_ms() - tf0;  // apply
    }
    st[kLmJobs] = 1;
    st[kLmWallMs] = now_ms() - t0;  // up to local BA
    st[kLmTriMatches] = (double)n_tri;
    st[kLmFused] = (double)(n_fused + n_back);
    st[kLmNewPoints] = (double)n_new;
    // ---- Optimizer::LocalBundleAdjustment over the keyframe's own window
    int64_t lba[5] = {0, 0, 0, 0, 0};  // edges, outliers, free, fixed, points
    LmPacket pk;
    so_ba_info info{};
    const double tl0 = now_ms();
    if ((int)M.kfs.size() > 2) {  // LocalMapping.cc:81
        static thread_local Window W;
        const double tg0 = now_ms();
        const bool have = local_window(r, M, *c, W);
        st[27] = now_ms() - tg0;  // gather
        if (have) {
            so_ba_problem p{};
            p.n_poses = (int32_t)W.kf.size(); p.Tcw = W.Tcw.data(); p.fixed = W.fixed.data(); p.intr = W.intr.data();
            p.n_points = (int32_t)W.pts.size(); p.Xw = W.Xw.data();
            p.n_edges = (int32_t)W.e_kf.size(); p.edge_pose = W.e_pose.data(); p.edge_point = W.e_pt.data();
            p.obs = W.obs.data(); p.inv_sigma2 = W.w.data();
            so_ba_options opt;
            so_ba_options_local(&opt);
            r->ba_Tcw.resize(W.Tcw.size()); r->ba_Xw.resize(W.Xw.size()); r->ba_out.resize(W.e_kf.size());
            so_bundle_adjust_set_solve_timing(r->mapper_opt, (r->lba_windows_run++ % 8) == 0);  // (event-timed solves take the stage-by-stage path: one window in eight)
            M.stop = 0;  // mbAbortBA = false (LocalMapping.cc:77)
            const double tb0 = now_ms();
            if (so_bundle_adjust(r->mapper_opt, &p, &opt, M.stop_flag(), r->ba_Tcw.data(), r->ba_Xw.data(), r->ba_out.data(), nullptr, &info) != SO_OK)
                return SO_ERR_HIP;
            st[28] = now_ms() - tb0;  // the solver call
            const double tw0 = now_ms();
            M.windows++;
            if (info.aborted) M.aborted++;
            for (size_t q = 0; q < W.kf.size(); q++)  // SetPose (Optimizer.cc:713-727)
                if (!W.fixed[q]) memcpy(M.kfs[(size_t)W.kf[q]]->T, &r->ba_Tcw[12 * q], 48);
            for (size_t q = 0; q < W.pts.size(); q++) memcpy(&M.X[3 * (size_t)W.pts[q]], &r->ba_Xw[3 * q], 12);  // SetWorldPos
            int64_t n_out = 0;
            for (size_t e = 0; e < W.e_kf.size(); e++) {  // EraseMapPointMatch / EraseObservation (:697-711)
                if (!r->ba_out[e]) continue;
                n_out++;
                const int s = W.pts[(size_t)W.e_pt[e]];
                if (M.bad[(size_t)s]) continue;
                const auto& o = M.obs[(size_t)s];
                if (std::find(o.begin(), o.end(), std::make_pair(W.e_kf[e], W.e_idx[e])) == o.end()) continue;
                L.erase_observation(s, W.e_kf[e], W.e_idx[e]);
            }
            lba[0] = (int64_t)W.e_kf.size(); lba[1] = n_out; lba[4] = (int64_t)W.pts.size();
            for (uint8_t f : W.fixed) lba[f ? 3 : 2]++;
            st[32] = now_ms() - tw0;  // SetPose / SetWorldPos / EraseObservation
            const double tu0 = now_ms();
            // UpdateNormalAndDepth of the window's points (:729-737)
            std::vector<int32_t> live;
            for (int s : W.pts)
                if (!M.bad[(size_t)s] && !M.obs[(size_t)s].empty()) live.push_back(s);
            if (!live.empty()) {
                // observers by index into the window's keyframes (so_update_normal_and_depth_indexed): 4 bytes per observation
                std::vector<float>&Xn = r->lm_nX, &ls = r->lm_nls, &ll = r->lm_nll, &cen = r->lm_nobs;
                std::vector<float>&nrm = r->lm_nnrm, &mxd = r->lm_nmax, &mnd = r->lm_nmin;
                std::vector<int32_t>&off = r->lm_noff, &okf = r->lm_to1, &rkf = r->lm_to2;
                const size_t nl = live.size(), nk = M.kfs.size();
                size_t total = 0;
                for (int s : live) total += M.obs[(size_t)s].size();
                off.resize(nl + 1); okf.resize(total); rkf.resize(nl);
                Xn.resize(3 * nl); ls.resize(nl); ll.assign(nl, r->scale[r->nlevels - 1]); nrm.resize(3 * nl); mxd.resize(nl); mnd.resize(nl);
                cen.resize(3 * nk);
                for (size_t kf = 0; kf < nk; kf++) {  // (every keyframe: an observer need not be part of the window)
                    double O[3];
                    centre(M.kfs[kf]->T, O);
                    for (int q = 0; q < 3; q++) cen[3 * kf + (size_t)q] = (float)O[q];
                }
                size_t at = 0;
                for (size_t q = 0; q < nl; q++) {
                    const size_t s = (size_t)live[q];
                    const auto& o = M.obs[s];
                    off[q] = (int32_t)at;
                    int rk = M.ref_kf[s], ri = -1;
                    for (const auto& e : o) {
                        okf[at++] = e.first;
                        if (e.first == rk && ri < 0) ri = e.second;
                    }
                    if (ri < 0) {  // the reference keyframe's observation is gone: the first observer takes over
                        rk = o[0].first;
                        ri = o[0].second;
                        M.ref_kf[s] = rk;
                    }
                    rkf[q] = rk;
                    ls[q] = r->scale[M.kfs[(size_t)rk]->octave[(size_t)ri]];
                    memcpy(&Xn[3 * q], &M.X[3 * s], 12);
                    memcpy(&nrm[3 * q], &M.N[3 * s], 12);
                    mxd[q] = M.mx[s];
                    mnd[q] = M.mn[s];
                }
                off[nl] = (int32_t)at;
                st[33] = now_ms() - tu0;  // the observers per point
                const double tv0 = now_ms();
                if (so_update_normal_and_depth_indexed(m, (int32_t)nl, off.data(), okf.data(), (int32_t)nk, cen.data(), Xn.data(), rkf.data(),
                                                       ls.data(), ll.data(), nrm.data(), mxd.data(), mnd.data()) != SO_OK)
                    return SO_ERR_HIP;
                st[34] = now_ms() - tv0;  // the call
                for (size_t q = 0; q < nl; q++) {
                    const size_t s = (size_t)live[q];
                    memcpy(&M.N[3 * s], &nrm[3 * q], 12);
                    M.mx[s] = mxd[q];
                    M.mn[s] = mnd[q];
                }
            }
            pk.moved = W.pts;
            st[29] = now_ms() - tw0;  // write-back
        }
    }
    st[30] = now_ms() - tl0;  // local BA incl. gather and write-back
    // ---- the packet
    const double tp0 = now_ms();
    pk.kf = k;
    pk.first_new = n_before;
    pk.n_points = L.n_points();
    pk.new_X.assign(M.X.begin() + 3 * (long)n_before, M.X.end());
    {
        const size_t nm = pk.moved.size();
        pk.moved_X.resize(3 * nm); pk.moved_N.resize(3 * nm); pk.moved_mx.resize(nm); pk.moved_mn.resize(nm);
        for (size_t q = 0; q < nm; q++) {
            const size_t s = (size_t)pk.moved[q];
            memcpy(&pk.moved_X[3 * q], &M.X[3 * s], 12);
            memcpy(&pk.moved_N[3 * q], &M.N[3 * s], 12);
            pk.moved_mx[q] = M.mx[s];
            pk.moved_mn[q] = M.mn[s];
        }
    }
    std::sort(M.newly_bad.begin(), M.newly_bad.end());
    pk.bad = M.newly_bad;
    for (int s : pk.bad) pk.bad_repl.push_back(M.repl[(size_t)s]);
    memcpy(pk.kf_T, c->T, 48);
    pk.recent_from = M.recent.empty() ? L.n_points() : *std::min_element(M.recent.begin(), M.recent.end());
    {
        const int id = L.next_stamp();
        const int first = std::max(0, (int)M.kfs.size() - std::max(1, r->local_keyframes));
        for (int kf = first; kf < (int)M.kfs.size(); kf++) {
            const KfSnap& q = *M.kfs[(size_t)kf];
            for (int i = 0; i < q.n; i++) {
                const int s = q.mp[(size_t)i];
                if (s >= 0 && M.stamp[(size_t)s] != id && !M.bad[(size_t)s]) {
                    M.stamp[(size_t)s] = id;
                    pk.local_slots.push_back(s);
                }
            }
        }
        std::sort(pk.local_slots.begin(), pk.local_slots.end());
    }
    // the keyframe that leaves the neighbour ring with this job is never searched again: its HBM block goes back to the pool
    // (the next keyframe's upload takes it instead of a hipMalloc)
    if (k >= r->lm_neighbours && M.kfs[(size_t)r0]->dev) {  // (the next job's ring starts at r0 + 1)
        so_kframe_destroy(M.kfs[(size_t)r0]->dev);
        M.kfs[(size_t)r0]->dev = nullptr;
    }
    const int64_t row[12] = {c->t, nn, n_tri, n_new, n_fused, n_back, lba[0], lba[1], lba[2], lba[3], lba[4], (int64_t)pk.bad.size()};
    st[35] = now_ms() - tp0;  // the packet
    st[31] = now_ms() - t0;   // the whole job
    if (info_out) *info_out = info;
    static const bool job_trace = getenv("SWARMORB_CL_TRACE") != nullptr;
    if (job_trace && timed)  // one line per job: when it began (ms, process clock) and where its time went
        fprintf(stderr, "[job] agent %p kf_t %d begin %.3f | process %.3f tri_stage %.3f (end %.3f) triangulate %.3f fuse_stage %.3f fuse_end %.3f apply %.3f "
                        "gather %.3f ba %.3f (gpu %.3f, trials %d) writeback %.3f (und %.3f) packet %.3f | job %.3f\n",
                (void*)r, c->t, t0, st[24], st[kLmStageTriMs], st[36], st[kLmTriangMs], st[kLmStageFuseMs], st[kLmBatchEndMs], st[26], st[27], st[28],
                (double)info.gpu_ms, (int)info.lm_trials, st[29], st[34], st[35], st[31]);
    {
        std::lock_guard<std::mutex> lk(r->mu);
        M.lm_log.insert(M.lm_log.end(), row, row + 12);
        if (timed)
            for (int i = 0; i < 40; i++) r->lm_stat[i] += st[i];
    }
    {
        std::lock_guard<std::mutex> lk(M.mu);
        M.outbox.push_back(std::move(pk));
    }
    M.cv.notify_all();
    return SO_OK;
}

// Tracking thread: frame t can start without waiting (no job out, its packet not due yet, or the packet is there; a failed job
// counts as ready - cl_frame_begin reports it).  so_fleet_run's elastic ticks ask before they take an agent into a tick.
bool cl_frame_ready(so_replay* r, int t) {
    if (!r->cl) return true;
    ClosedLoop& M = *r->cl;
    if (!M.job_pending || M.policy != 0 || t < M.apply_at) return true;
    std::lock_guard<std::mutex> lk(M.mu);
    return !M.outbox.empty() || M.failed;
}

// Tracking thread, before the frame's first search: what local mapping handed back arrives in the tracked map.
int cl_frame_begin(so_replay* r, int t) {
    ClosedLoop& M = *r->cl;
    if (!M.job_pending) return SO_OK;
    LmPacket pk;
    {
        std::unique_lock<std::mutex> lk(M.mu);
        if (M.policy == 0) {
            if (t < M.apply_at) return SO_OK;
            const double w0 = now_ms();
            // (deterministic schedule: wait for the job.  Spin first - the job is usually a few hundred microseconds from its
            //  end and a futex wake-up costs 30-60 us of the cycle -, sleep on the condition variable after 2 ms)
            while (M.outbox.empty() && !M.failed && now_ms() - w0 < 2.0) {
                lk.unlock();
                for (int i = 0; i < 64; i++) __builtin_ia32_pause();
                lk.lock();
            }
            M.cv.wait(lk, [&] { return !M.outbox.empty() || M.failed; });
            M.wait_ms += now_ms() - w0;
            M.t_packet = now_ms();
            if (M.outbox.empty()) return SO_ERR_HIP;
        } else if (M.outbox.empty()) {
            return SO_OK;
        }
        pk = std::move(M.outbox.front());
        M.outbox.pop_front();
    }
    M.job_pending = false;
    const double a0 = now_ms();
    const size_t n_old = r->mp_X.size() / 3;
    if ((int)n_old != pk.first_new) {
        r->error = "closed loop: the tracking side's map size and the packet disagree";
        return SO_ERR_HIP;
    }
    r->mp_X.insert(r->mp_X.end(), pk.new_X.begin(), pk.new_X.end());
    M.tv_bad.resize((size_t)pk.n_points, 0);
    M.tv_repl.resize((size_t)pk.n_points, -1);
    M.tv_vis.resize((size_t)pk.n_points, 1);    // a new MapPoint starts with mnVisible = mnFound = 1 (MapPoint.cc:38)
    M.tv_found.resize((size_t)pk.n_points, 1);
    if (!pk.moved.empty()) {
        for (size_t q = 0; q < pk.moved.size(); q++) memcpy(&r->mp_X[3 * (size_t)pk.moved[q]], &pk.moved_X[3 * q], 12);
        if (so_map_write_rows(r->map, (int32_t)pk.moved.size(), pk.moved.data(), pk.moved_X.data(), pk.moved_N.data(), pk.moved_mx.data(),
                              pk.moved_mn.data()) != SO_OK) {
            r->error = std::string("so_map_write_rows: ") + so_last_error();
            return SO_ERR_HIP;
        }
    }
    for (size_t q = 0; q < pk.bad.size(); q++) {
        M.tv_bad[(size_t)pk.bad[q]] = 1;
        M.tv_repl[(size_t)pk.bad[q]] = pk.bad_repl[q];
        if (pk.bad_repl[q] >= 0) {  // MapPoint::Replace hands its counters to the survivor (MapPoint.cc:280-281)
            M.tv_vis[(size_t)pk.bad_repl[q]] += M.tv_vis[(size_t)pk.bad[q]];
            M.tv_found[(size_t)pk.bad_repl[q]] += M.tv_found[(size_t)pk.bad[q]];
        }
    }
    M.recent_from = pk.recent_from;
    M.tv_local.swap(pk.local_slots);
    // Tracking::UpdateLastFrame (Tracking.cc:656-662): the last frame follows its reference keyframe
    M.T_ref = from_f12(pk.kf_T);
    r->T_last = mul(M.Tlr, M.T_ref);
    // Tracking::CheckReplacedInLastFrame (:940-955)
    so_replay::FrameHost& Lf = r->fh[r->cur];  // (called before the step flips `cur`: the frame tracked last)
    for (int i = 0; i < Lf.n; i++) {
        int s = Lf.kp_mp[(size_t)i];
        if (s >= 0 && M.tv_bad[(size_t)s]) {
            while (s >= 0 && M.tv_bad[(size_t)s]) s = M.tv_repl[(size_t)s];
            Lf.kp_mp[(size_t)i] = s;
        }
    }
    M.apply_ms += now_ms() - a0;
    return SO_OK;
}

// Tracking thread: a keyframe has been handed to local mapping (it is the reference keyframe from now on) ...
void cl_keyframe_queued(so_replay* r, int t, const std::shared_ptr<KfSnap>& snap) {
    ClosedLoop& M = *r->cl;
    M.T_ref = from_f12(snap->T);
    M.job_pending = true;
    M.apply_at = t + M.delay;
    M.last_kf_t = t;
    M.kf_t.push_back(t);
    M.n_kf++;
    if (M.t_packet > 0.0) {  // (diagnostics: the part of the cycle that is not the job)
        M.handover_ms += now_ms() - M.t_packet;
        M.n_handover++;
        M.t_packet = 0.0;
    }
}

// ... and the end of a frame: its pose relative to its reference keyframe (mlRelativeFramePoses)
void cl_frame_end(so_replay* r, int t) {
    ClosedLoop& M = *r->cl;
    (void)t;
    M.Tlr = mul(r->step.T, rigid_inverse_general(M.T_ref));
    M.ref_log.push_back(M.n_kf - 1);
    M.Tcr_log.insert(M.Tcr_log.end(), M.Tlr.a, M.Tlr.a + 16);
}
