// replay.cc — the per-frame replay of SURVEY 8d as a C++ host loop over the C ABI (include/swarmorb.h): what an
// agent's Tracking thread and LocalMapping thread do with the library, without an interpreter in the loop.
//
// Tracking thread (the caller of so_replay_run), per frame t — a chained, device-resident frame:
//     collect frame t  (keypoints / descriptors / undistorted points arrive through host-mapped memory)
//     submit frame t+1 (host image -> HBM upload + ExtractORB + UndistortKeyPoints + AssignFeaturesToGrid, async)
//     TrackWithMotionModel:  so_track_search_last_frame (projection + window search on the GPU)          Tracking.cc:714-768
//                            so_pose_optimization over the matches it found, outliers dropped             Optimizer.cc:239-434
//     TrackLocalMap:         so_track_search_local_map (isInFrustum + window search on the GPU)           Tracking.cc:770-807, 964-1007
//                            so_pose_optimization over all matches
//                            a third so_pose_optimization from the last frame's pose (TrackReferenceKeyFrame's
//                            fallback; SURVEY 8d counts three calls per frame), result unused
//     "keyframe":            unmatched keypoints become map points (so_map_write)
//     every lba_every frames: hand a window to the local-mapping thread (at most two waiting: back-pressure)
// Local-mapping thread, per queued keyframe (every lba_every-th tracked frame), in LocalMapping::Run's order
// (code/src/LocalMapping.cc:53-110): CreateNewMapPoints' matcher load - SearchForTriangulation of the new keyframe against
// each of the last <= 20 keyframes (:197-246) -, SearchInNeighbors' - Fuse of its map points into each of them and of
// theirs into it (:451-481) -, then Optimizer::LocalBundleAdjustment (so_bundle_adjust) on the queued window.  The
// matcher job works on snapshots of the tracked frames (keypoints, bindings, pose, the bound map points' fields) and
// needs a vocabulary stand-in (so_replay_set_vocabulary: the feature vector of a keyframe = its descriptors' nearest
// of ~100 centroid descriptors, found with so_hamming_top2 - DBoW2 itself is out of scope); its
// results are counted and logged, not fed back into the tracked map (like the window's).
//
// It is NOT the reference's Tracking state machine (out of scope): it is the shortest loop that chains every
// per-frame operator the way Tracking does, on the synthetic planar scene of swarmmap_amd/synth.py (map points come
// from back-projection onto the known plane).  swarmmap_amd/minitrack.py is the same loop in Python; tests compare
// this one, frame by frame, with that loop run over the CPU oracle.
// Built by csrc/Makefile into libswarmorb_replay.so with g++.
#include "replay_internal.h"


namespace {


// CreateNewMapPoints' and SearchInNeighbors' matcher load for the new keyframe `c` (see the file header).
int lm_matcher_job(so_replay* r, const std::shared_ptr<KfSnap>& c, bool timed) {
    so_matcher* m = r->mapper_matcher;
    const double t0 = now_ms();
    const int n = c->n, nv = (int)(r->vocab.size() / 32);
    double st[24] = {0};
    float kms = 0.f;
    so_matcher_set_profiling(m, 1);
    {   // KeyFrame::ComputeBoW's feature vector, stand-in: node = nearest centroid descriptor (lowest index on ties)
        std::vector<int32_t> node((size_t)n), bd((size_t)n), sd((size_t)n);
        if (n > 0 && so_hamming_top2(m, c->desc.data(), n, r->vocab.data(), nv, node.data(), bd.data(), sd.data()) != SO_OK) return SO_ERR_HIP;
        std::vector<int32_t> count((size_t)nv + 1, 0);
        for (int i = 0; i < n; i++) count[(size_t)node[(size_t)i] + 1]++;
        for (int v = 0; v < nv; v++) count[(size_t)v + 1] += count[(size_t)v];
        std::vector<int32_t> pos(count.begin(), count.end() - 1), by_node((size_t)n);
        for (int i = 0; i < n; i++) by_node[(size_t)pos[(size_t)node[(size_t)i]]++] = i;  // feature order inside a node
        c->off.assign(1, 0);
        for (int v = 0; v < nv; v++)
            if (count[(size_t)v + 1] > count[(size_t)v]) {
                c->node_id.push_back(v);
                for (int a = count[(size_t)v]; a < count[(size_t)v + 1]; a++) c->idx.push_back(by_node[(size_t)a]);
                c->off.push_back((int32_t)c->idx.size());
            }
    }
    const so_featvec fv1{(int32_t)c->node_id.size(), c->node_id.data(), c->off.data(), c->idx.data()};
    float level_sigma2[8];
    for (int l = 0; l < 8; l++) level_sigma2[l] = r->scale[l] * r->scale[l];
    const so_frame_view Vc = keyframe_view(r, *c);
    if (r->lm_resident && so_kframe_create(m, &Vc, &fv1, level_sigma2, &c->dev) != SO_OK) return SO_ERR_HIP;
    st[kLmNodeMs] = now_ms() - t0;
    std::vector<uint8_t> free1((size_t)n), free2;
    for (int i = 0; i < n; i++) free1[(size_t)i] = c->mp[(size_t)i] < 0 ? 1 : 0;
    int n_tri = 0, n_fused = 0, n_back = 0;
    // Every search of this keyframe is independent of the others here (nothing is written back into the map), so with
    // lm_batch they all go out as ONE batch: one staging copy, one projection launch, one search launch, one wait.
    // Outputs are held per call until the batch ends.
    const bool batch = r->lm_batch;
    const size_t nn = r->lm_ring.size();
    std::vector<std::vector<int32_t>> tri_m12(nn), fuse_best(nn + 1), fuse_dist(nn + 1);
    std::vector<int32_t> tri_nm(nn, 0), fuse_n(nn + 1, 0);
    std::vector<std::vector<uint8_t>> fuse_valid(nn);
    const double tb0 = now_ms();
    if (batch && so_matcher_batch_begin(m) != SO_OK) return SO_ERR_HIP;
    struct BatchGuard {  // an early return below leaves the handle usable (the deferred calls' output arrays die with this frame)
        so_matcher* m;
        bool armed;
        ~BatchGuard() { if (armed) so_matcher_batch_abort(m); }
    } guard{m, batch};
    // ---- CreateNewMapPoints: SearchForTriangulation(mpCurrentKeyFrame, pKF2, F12, vMatchedIndices, false) per neighbour
    size_t jn = 0;
    const double tc0 = now_ms();
    // both sides resident and inside a batch: ONE call for all neighbours, the queries built on the device (round 5)
    static const bool tri_multi = !(getenv("SWARMORB_LM_TRI_MULTI") && atoi(getenv("SWARMORB_LM_TRI_MULTI")) == 0);
    std::vector<std::vector<uint8_t>> free2s;
    bool multi = tri_multi && batch && c->dev && !r->lm_ring.empty();
    for (const auto& kf2 : r->lm_ring) multi = multi && kf2->dev;
    if (multi) {
        std::vector<so_tri_neighbour> nbs;
        free2s.resize(nn);
        for (const auto& kf2 : r->lm_ring) {
            so_tri_neighbour N;
            memset(&N, 0, sizeof(N));
            fundamental_and_epipole(r, c->T, kf2->T, N.F12, &N.ex, &N.ey);
            free2s[jn].resize((size_t)kf2->n);
            for (int i = 0; i < kf2->n; i++) free2s[jn][(size_t)i] = kf2->mp[(size_t)i] < 0 ? 1 : 0;
            tri_m12[jn].resize((size_t)n);
            N.kf2 = kf2->dev;
            N.free2 = free2s[jn].data();
            N.matches12 = tri_m12[jn].data();
            N.nmatches = &tri_nm[jn];
            nbs.push_back(N);
            st[kLmTriCalls] += 1;
            jn++;
        }
        const double ta = now_ms();
        if (so_search_for_triangulation_kframes(m, c->dev, free1.data(), (int32_t)nbs.size(), nbs.data(), 1) != SO_OK) return SO_ERR_HIP;
        st[kLmTriMs] += now_ms() - ta;
    }
    for (const auto& kf2 : r->lm_ring) {
        if (multi) break;
        float F12[9], ex, ey;
        fundamental_and_epipole(r, c->T, kf2->T, F12, &ex, &ey);
        free2.resize((size_t)kf2->n);
        for (int i = 0; i < kf2->n; i++) free2[(size_t)i] = kf2->mp[(size_t)i] < 0 ? 1 : 0;
        const so_featvec fv2{(int32_t)kf2->node_id.size(), kf2->node_id.data(), kf2->off.data(), kf2->idx.data()};
        tri_m12[jn].resize((size_t)n);
        const double ta = now_ms();
        const int trc = kf2->dev
            ? so_search_for_triangulation_kframe(m, n, c->x.data(), c->y.data(), c->angle.data(), c->desc.data(), free1.data(), &fv1,
                                                 kf2->dev, free2.data(), F12, ex, ey, 1, tri_m12[jn].data(), &tri_nm[jn])
            : so_search_for_triangulation(m, n, c->x.data(), c->y.data(), c->angle.data(), c->desc.data(), free1.data(), &fv1, kf2->n,
                                          kf2->x.data(), kf2->y.data(), kf2->octave.data(), kf2->angle.data(), kf2->desc.data(),
                                          free2.data(), &fv2, F12, ex, ey, r->scale, level_sigma2, r->nlevels, 1, tri_m12[jn].data(),
                                          &tri_nm[jn]);
        if (trc != SO_OK) return SO_ERR_HIP;
        st[kLmTriMs] += now_ms() - ta;
        if (!batch) {
            so_matcher_last_kernel_ms(m, &kms);
            st[kLmTriKernelMs] += kms;
        }
        st[kLmTriCalls] += 1;
        jn++;
    }
    st[kLmStageTriMs] = now_ms() - tc0;
    const double td0 = now_ms();
    // ---- SearchInNeighbors: matcher.Fuse(pKFi, vpMapPointMatches) per neighbour (LocalMapping.cc:451-457) ...
    auto grow = [&](size_t slots) {
        if (slots > r->lm_stamp.size()) { r->lm_stamp.resize(slots + slots / 2 + 1024, -1); r->lm_cstamp.resize(r->lm_stamp.size(), -1); }
    };
    int max_slot = -1;
    for (int i = 0; i < n; i++) max_slot = std::max(max_slot, c->mp[(size_t)i]);
    for (const auto& k2 : r->lm_ring)
        for (int i = 0; i < k2->n; i++) max_slot = std::max(max_slot, k2->mp[(size_t)i]);
    grow((size_t)(max_slot + 1));
    so_mappoint_view P;
    memset(&P, 0, sizeof(P));
    P.n = n; P.Xw = c->mpX.data(); P.normal = c->mpN.data(); P.max_dist = c->mpMax.data(); P.min_dist = c->mpMin.data();
    P.desc = c->mpDesc.data();
    jn = 0;
    for (const auto& k2 : r->lm_ring) {
        const int id = ++r->lm_stamp_id;
        for (int i = 0; i < k2->n; i++)
            if (k2->mp[(size_t)i] >= 0) r->lm_stamp[(size_t)k2->mp[(size_t)i]] = id;
        std::vector<uint8_t>& valid = fuse_valid[jn];
        valid.resize((size_t)n);
        for (int i = 0; i < n; i++)  // pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKFi), ORBmatcher.cc:770-774
            valid[(size_t)i] = (c->mp[(size_t)i] >= 0 && r->lm_stamp[(size_t)c->mp[(size_t)i]] != id) ? 1 : 0;
        P.valid = valid.data();
        const so_frame_view V2 = keyframe_view(r, *k2);
        fuse_best[jn].resize((size_t)n); fuse_dist[jn].resize((size_t)n);
        const double ta = now_ms();
        // (resident keyframe + resident map: the call stages the points' slots and flags, 5 bytes each; their positions,
        //  normals, distance ranges and descriptors are rows of the map table the tracking searches read as well)
        const int frc = (k2->dev && r->lm_from_map)
            ? so_fuse_kframe_map(m, k2->dev, &r->cam, k2->T, r->log_sf, r->inv_sigma2, r->map, n, c->mp.data(), valid.data(), 3.0f,
                                 fuse_best[jn].data(), fuse_dist[jn].data(), &fuse_n[jn], nullptr)
            : k2->dev
            ? so_fuse_kframe(m, k2->dev, &r->cam, k2->T, r->log_sf, r->inv_sigma2, &P, 3.0f, fuse_best[jn].data(), fuse_dist[jn].data(),
                             &fuse_n[jn], nullptr)
            : so_fuse(m, &V2, &r->cam, k2->T, r->log_sf, r->inv_sigma2, &P, 3.0f, fuse_best[jn].data(), fuse_dist[jn].data(), &fuse_n[jn],
                      nullptr);
        if (frc != SO_OK) return SO_ERR_HIP;
        st[kLmFuseMs] += now_ms() - ta;
        if (!batch) {
            so_matcher_last_kernel_ms(m, &kms);
            st[kLmFuseKernelMs] += kms;
        }
        st[kLmFuseCalls] += 1;
        st[kLmFusePoints] += n;
        jn++;
    }
    st[kLmStageFuseMs] = now_ms() - td0;
    const double te0 = now_ms();
    // ... then the neighbours' map points into the new keyframe: vpFuseCandidates, once each (:459-481)
    std::vector<float>&X = r->lm_cX, &N = r->lm_cN, &mx = r->lm_cmax, &mn = r->lm_cmin;  // (capacity kept from keyframe to keyframe)
    std::vector<uint8_t>&D = r->lm_cD, &ok = r->lm_cok;
    if (!r->lm_ring.empty()) {
        const int job = ++r->lm_stamp_id;
        const int cid = ++r->lm_stamp_id;
        for (int i = 0; i < n; i++)
            if (c->mp[(size_t)i] >= 0) r->lm_stamp[(size_t)c->mp[(size_t)i]] = cid;
        size_t cap = 0;
        for (const auto& k2 : r->lm_ring) cap += (size_t)k2->n;
        const bool from_map = c->dev && r->lm_from_map;
        std::vector<int32_t>& cslot = r->lm_cslot;
        if (from_map) cslot.resize(cap);
        else { X.resize(3 * cap); N.resize(3 * cap); mx.resize(cap); mn.resize(cap); D.resize(32 * cap); }
        ok.resize(cap);
        size_t q = 0;
        for (const auto& k2 : r->lm_ring)
            for (int i = 0; i < k2->n; i++) {
                const int slot = k2->mp[(size_t)i];
                if (slot < 0 || r->lm_cstamp[(size_t)slot] == job) continue;  // mnFuseCandidateForKF
                r->lm_cstamp[(size_t)slot] = job;
                if (from_map) {
                    cslot[q] = slot;
                } else {
                    memcpy(&X[3 * q], &k2->mpX[3 * (size_t)i], 12);
                    memcpy(&N[3 * q], &k2->mpN[3 * (size_t)i], 12);
                    mx[q] = k2->mpMax[(size_t)i];
                    mn[q] = k2->mpMin[(size_t)i];
                    memcpy(&D[32 * q], &k2->mpDesc[32 * (size_t)i], 32);
                }
                ok[q] = r->lm_stamp[(size_t)slot] != cid ? 1 : 0;  // !IsInKeyFrame(mpCurrentKeyFrame)
                q++;
            }
        so_mappoint_view Q;
        memset(&Q, 0, sizeof(Q));
        Q.n = (int32_t)q; Q.Xw = X.data(); Q.normal = N.data(); Q.max_dist = mx.data(); Q.min_dist = mn.data();
        Q.desc = D.data(); Q.valid = ok.data();
        fuse_best[nn].resize((size_t)Q.n); fuse_dist[nn].resize((size_t)Q.n);
        const double ta = now_ms();
        const int brc = from_map
            ? so_fuse_kframe_map(m, c->dev, &r->cam, c->T, r->log_sf, r->inv_sigma2, r->map, Q.n, cslot.data(), ok.data(), 3.0f,
                                 fuse_best[nn].data(), fuse_dist[nn].data(), &fuse_n[nn], nullptr)
            : c->dev
            ? so_fuse_kframe(m, c->dev, &r->cam, c->T, r->log_sf, r->inv_sigma2, &Q, 3.0f, fuse_best[nn].data(), fuse_dist[nn].data(),
                             &fuse_n[nn], nullptr)
            : so_fuse(m, &Vc, &r->cam, c->T, r->log_sf, r->inv_sigma2, &Q, 3.0f, fuse_best[nn].data(), fuse_dist[nn].data(), &fuse_n[nn],
                      nullptr);
        if (brc != SO_OK) return SO_ERR_HIP;
        st[kLmFuseMs] += now_ms() - ta;
        if (!batch) {
            so_matcher_last_kernel_ms(m, &kms);
            st[kLmFuseKernelMs] += kms;
        }
        st[kLmFuseCalls] += 1;
        st[kLmFusePoints] += Q.n;
    }
    st[kLmStageBackMs] = now_ms() - te0;
    if (batch) {
        const double ta = now_ms();
        guard.armed = false;  // (so_matcher_batch_end leaves the batch on every path)
        if (so_matcher_batch_end(m) != SO_OK) return SO_ERR_HIP;
        double ms4[4] = {0};
        so_matcher_last_stats(m, ms4);
        st[kLmBatchEnqueueMs] = ms4[0];
        st[kLmBatchWaitMs] = ms4[1];
        so_matcher_last_kernel_ms(m, &kms);
        st[kLmBatchKernelMs] = kms;
        st[kLmBatchEndMs] = now_ms() - ta;
        st[kLmBatchMs] = now_ms() - tb0;
    }
    for (size_t j = 0; j < nn; j++) {
        n_tri += tri_nm[j];
        n_fused += fuse_n[j];
    }
    n_back = fuse_n[nn];
    // ---- CreateNewMapPoints, the rest of its per-match body (LocalMapping.cc:263-420): the matches of ALL neighbours are
    //      triangulated and gated in one launch, then the new points get their normal and scale-invariance range
    //      (MapPoint::UpdateNormalAndDepth, :408: two observations each, the new keyframe is the reference) in another
    int n_new = 0;
    {
        const double ta = now_ms();
        std::vector<int32_t>&of = r->lm_tof, &o1 = r->lm_to1, &o2 = r->lm_to2;
        std::vector<float>&p1 = r->lm_txy1, &p2 = r->lm_txy2;
        of.clear(); o1.clear(); o2.clear(); p1.clear(); p2.clear();
        size_t j = 0;
        for (const auto& k2 : r->lm_ring) {
            const std::vector<int32_t>& m12 = tri_m12[j];
            for (int i = 0; i < n; i++) {
                const int i2 = m12[(size_t)i];
                if (i2 < 0) continue;
                of.push_back((int32_t)j);
                p1.push_back(c->x[(size_t)i]); p1.push_back(c->y[(size_t)i]);
                o1.push_back(c->octave[(size_t)i]);
                p2.push_back(k2->x[(size_t)i2]); p2.push_back(k2->y[(size_t)i2]);
                o2.push_back(k2->octave[(size_t)i2]);
            }
            j++;
        }
        const int nt = (int)of.size();
        if (nt > 0) {
            auto tri_kf = [r, level_sigma2 = &level_sigma2[0]](const KfSnap& k) {
                so_tri_keyframe t;
                memset(&t, 0, sizeof(t));
                memcpy(t.Tcw, k.T, sizeof(t.Tcw));
                t.fx = r->cam.fx; t.fy = r->cam.fy; t.cx = r->cam.cx; t.cy = r->cam.cy;
                t.invfx = 1.0f / r->cam.fx; t.invfy = 1.0f / r->cam.fy;  // Frame.cc:266
                t.scale_factors = r->scale;
                t.level_sigma2 = level_sigma2;
                t.nlevels = r->nlevels;
                return t;
            };
            const so_tri_keyframe k1 = tri_kf(*c);
            std::vector<so_tri_keyframe> k2s;
            for (const auto& k2 : r->lm_ring) k2s.push_back(tri_kf(*k2));
            std::vector<uint8_t>& okv = r->lm_tok;
            std::vector<float>& X3 = r->lm_tX;
            okv.assign((size_t)nt, 0);
            X3.resize(3 * (size_t)nt);
            const float ratio_factor = 1.5f * 1.2f;  // 1.5f * mpCurrentKeyFrame->mfScaleFactor, :214
            // triangulation, its gates, and the accepted points' normal / distance range (observations: this keyframe and
            // the neighbour; reference keyframe: this one) in ONE launch; SWARMORB_LM_TRI_FUSED=0: the two calls of before
            std::vector<float>&nrm = r->lm_nnrm, &mxd = r->lm_nmax, &mnd = r->lm_nmin;
            static const bool fused = !(getenv("SWARMORB_LM_TRI_FUSED") && atoi(getenv("SWARMORB_LM_TRI_FUSED")) == 0);
            if (fused) {
                nrm.assign(3 * (size_t)nt, 0.f); mxd.assign((size_t)nt, 0.f); mnd.assign((size_t)nt, 0.f);
                if (so_triangulate_new_points(m, &k1, (int32_t)k2s.size(), k2s.data(), ratio_factor, nt, of.data(), p1.data(), o1.data(),
                                              p2.data(), o2.data(), okv.data(), X3.data(), nrm.data(), mxd.data(), mnd.data()) != SO_OK)
                    return SO_ERR_HIP;
                so_matcher_last_kernel_ms(m, &kms);
                st[kLmTriangKernelMs] = kms;
                for (int k = 0; k < nt; k++) n_new += okv[(size_t)k] ? 1 : 0;
            } else {
                if (so_triangulate_matches(m, &k1, (int32_t)k2s.size(), k2s.data(), ratio_factor, nt, of.data(), p1.data(), o1.data(),
                                           p2.data(), o2.data(), okv.data(), X3.data()) != SO_OK)
                    return SO_ERR_HIP;
                so_matcher_last_kernel_ms(m, &kms);
                st[kLmTriangKernelMs] = kms;
                // the new points: observations = (new keyframe, neighbour), reference keyframe = the new one
                std::vector<float>&obs = r->lm_nobs, &Xn = r->lm_nX, &rO = r->lm_nref, &ls = r->lm_nls, &ll = r->lm_nll;
                std::vector<int32_t>& off = r->lm_noff;
                obs.clear(); Xn.clear(); rO.clear(); ls.clear(); ll.clear(); off.assign(1, 0);
                float Owc[3];
                auto centre = [](const float* T, float* Ow) {
                    for (int q = 0; q < 3; q++) Ow[q] = (float)(-((double)T[q] * T[3] + (double)T[4 + q] * T[7] + (double)T[8 + q] * T[11]));
                };
                centre(c->T, Owc);
                std::vector<float> Ow2(3 * k2s.size());
                size_t jj = 0;
                for (const auto& k2 : r->lm_ring) centre(k2->T, &Ow2[3 * jj++]);
                for (int k = 0; k < nt; k++) {
                    if (!okv[(size_t)k]) continue;
                    n_new++;
                    obs.insert(obs.end(), Owc, Owc + 3);
                    obs.insert(obs.end(), &Ow2[3 * (size_t)of[(size_t)k]], &Ow2[3 * (size_t)of[(size_t)k]] + 3);
                    off.push_back((int32_t)(obs.size() / 3));
                    Xn.insert(Xn.end(), &X3[3 * (size_t)k], &X3[3 * (size_t)k] + 3);
                    rO.insert(rO.end(), Owc, Owc + 3);
                    ls.push_back(r->scale[o1[(size_t)k]]);
                    ll.push_back(r->scale[r->nlevels - 1]);
                }
                if (n_new > 0) {
                    nrm.assign(3 * (size_t)n_new, 0.f); mxd.assign((size_t)n_new, 0.f); mnd.assign((size_t)n_new, 0.f);
                    if (so_update_normal_and_depth(m, n_new, off.data(), obs.data(), Xn.data(), rO.data(), ls.data(), ll.data(), nrm.data(),
                                                   mxd.data(), mnd.data()) != SO_OK)
                        return SO_ERR_HIP;
                }
            }
        }
        st[kLmTriangMs] = now_ms() - ta;
        st[kLmNewPoints] = n_new;
    }
    const int32_t row[6] = {c->t, (int32_t)r->lm_ring.size(), n_tri, n_fused, n_back, n_new};
    r->lm_ring.push_back(c);
    while ((int)r->lm_ring.size() > r->lm_neighbours) r->lm_ring.pop_front();
    st[kLmJobs] = 1; st[kLmWallMs] = now_ms() - t0; st[kLmTriMatches] = n_tri; st[kLmFused] = n_fused + n_back;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->lm_log.insert(r->lm_log.end(), row, row + 6);
        if (timed)
            for (int i = 0; i < 24; i++) r->lm_stat[i] += st[i];  // ([24..31]: the closed loop's job)
    }
    return SO_OK;
}

// The calling thread onto the agent's CPUs: one group of cores behind one L3 on the NUMA node next to its GPU.  The
// loop's two threads hand keyframes to each other and both touch pinned staging memory and ring doorbells all the
// time; left to the OS on a two-socket, 16-CCD box they cost 5 % of the frame rate and most of the run-to-run spread
// (include/swarmorb.h, so_device_host_cpus).  SWARMORB_NO_PIN=1 leaves placement to the OS.
// role >= 0 with SWARMORB_PIN_CORES=1 (A/B): the thread gets ONE cpu of the group - the role-th of its first half, i.e. a
// physical core of its own (the second half of a group are the SMT siblings) - instead of the whole group
void pin_to_device_node(const so_replay* r, int role = -1) {
    if (r->host_cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n_set = 0;
    std::vector<int> listed;
    const char* p = r->host_cpus.c_str();
    while (*p) {  // "a-b,c,d-e"
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            if (end == p + 1) break;
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0) {
                CPU_SET((int)c, &set);
                listed.push_back((int)c);
                n_set++;
            }
        if (*p == ',') p++;
    }
    // inside what the thread is allowed already (a harness that placed the whole process knows better than a guess by
    // creation order: bench.py pins every thread of the rank, the HIP runtime's included, to the agents' groups)
    cpu_set_t now, both;
    CPU_ZERO(&now);
    if (pthread_getaffinity_np(pthread_self(), sizeof(now), &now) == 0) {
        CPU_AND(&both, &set, &now);
        if (CPU_COUNT(&both) == 0) return;
        set = both;
    }
    static const bool cores = getenv("SWARMORB_PIN_CORES") && atoi(getenv("SWARMORB_PIN_CORES")) != 0;
    if (cores && role >= 0 && listed.size() >= 4) {
        const int half = (int)listed.size() / 2, cpu = listed[(size_t)(role % half)];
        if (CPU_ISSET(cpu, &set)) {
            CPU_ZERO(&set);
            CPU_SET(cpu, &set);
        }
    }
    if (n_set > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}

std::mutex g_pin_mu;
std::vector<uint8_t> g_pin_slots;  // slot -> taken by a live agent of this process
int take_pin_slot() {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (size_t i = 0; i < g_pin_slots.size(); i++)
        if (!g_pin_slots[i]) {
            g_pin_slots[i] = 1;
            return (int)i;
        }
    g_pin_slots.push_back(1);
    return (int)g_pin_slots.size() - 1;
}
void give_pin_slot(int slot) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (slot >= 0 && (size_t)slot < g_pin_slots.size()) g_pin_slots[(size_t)slot] = 0;
}

// The thread that calls so_replay_run / _run_live / so_fleet_run is the application's: it runs on the agent's CPUs for the
// duration of the call only.
struct CallerPin {
    cpu_set_t saved;
    bool have = false;
    explicit CallerPin(const so_replay* r) {
        CPU_ZERO(&saved);
        have = !r->host_cpus.empty() && pthread_getaffinity_np(pthread_self(), sizeof(saved), &saved) == 0;
        pin_to_device_node(r, 0);
    }
    ~CallerPin() {
        if (have) (void)pthread_setaffinity_np(pthread_self(), sizeof(saved), &saved);
    }
};

void mapper_loop(so_replay* r) {
    pin_to_device_node(r, 1);  // (the library's own thread: pinned for its lifetime)
    for (;;) {
        int timed;
        bool pre = false;
        std::shared_ptr<KfSnap> kf;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            if (r->cl) {  // closed loop: the next keyframe is at most a frame or two away - spin before sleeping (a futex
                          // wake-up is 30-60 us of a cycle the local-mapping thread bounds)
                const double w0 = now_ms();
                while (!r->quit && r->queue.empty() && now_ms() - w0 < 1.0) {
                    lk.unlock();
                    for (int i = 0; i < 64; i++) __builtin_ia32_pause();
                    lk.lock();
                }
            }
            r->cv.wait(lk, [r] { return r->quit || !r->queue.empty(); });
            if (r->queue.empty()) return;
            timed = r->queue.front().timed;
            kf = r->queue.front().kf;
            pre = r->queue.front().pre;
            r->running = 1;
        }
        const double t0 = now_ms();
        int lm_rc = SO_OK;
        if (r->cl && pre) {  // the keyframe-to-be's feature vector + upload, while the tracking thread still tracks the frame
            lm_rc = kf ? cl_keyframe_featvec_upload(r, r->mapper_matcher, *kf) : SO_OK;
            {
                std::lock_guard<std::mutex> lk(r->mu);
                if (lm_rc != SO_OK && r->error.empty()) r->error = std::string("keyframe feature vector / upload: ") + so_last_error();
                r->queue.pop_front();
                r->running = 0;
            }
            r->cv.notify_all();
            continue;
        }
        if (r->cl) {  // the closed loop: the keyframe's whole local-mapping job, results handed back to the tracking side
            so_ba_info info{};
            lm_rc = kf ? cl_lm_job(r, kf, timed != 0, &info) : SO_OK;
            const double busy = now_ms() - t0;
            {
                std::lock_guard<std::mutex> lk(r->mu);
                if (lm_rc != SO_OK && r->error.empty()) r->error = std::string("closed-loop local-mapping job: ") + so_last_error();
                if (timed) {
                    r->stat[kLbaWindows] += info.lm_trials > 0 ? 1 : 0;
                    r->stat[kLbaBusyMs] += busy;
                    r->stat[kLbaGpuMs] += info.gpu_ms;
                    r->stat[kLbaSolveMs] += info.solve_ms;
                    r->stat[kLbaSolves] += info.n_solves;
                    r->stat[kLbaTrials] += info.lm_trials;
                }
                r->queue.pop_front();
                r->running = 0;
            }
            r->cv.notify_all();
            if (lm_rc != SO_OK) {  // an error wakes a tracking thread that waits for the packet: the flag is set under the
                                   // mutex its predicate is evaluated under, so the wake-up cannot fall between test and sleep
                std::lock_guard<std::mutex> lk(r->cl->mu);
                r->cl->failed = true;
            }
            r->cl->cv.notify_all();
            continue;
        }
        if (kf) lm_rc = lm_matcher_job(r, kf, timed != 0);  // CreateNewMapPoints + SearchInNeighbors before the LBA
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
        // the solve kernel is event-timed on every fourth window only (two event records per LM trial idle the stream)
        so_bundle_adjust_set_solve_timing(r->mapper_opt, (r->lba_windows_run++ % 4) == 0);
        const int rc = so_bundle_adjust(r->mapper_opt, &p, &opt, nullptr, r->ba_Tcw.data(), r->ba_Xw.data(),
                                        r->ba_out.data(), nullptr, &info);
        const double busy = now_ms() - t0;
        {
            std::lock_guard<std::mutex> lk(r->mu);
            if (rc != SO_OK && r->error.empty()) r->error = std::string("so_bundle_adjust: ") + so_last_error();
            if (lm_rc != SO_OK && r->error.empty()) r->error = std::string("local-mapping matcher job: ") + so_last_error();
            if (timed) {
                r->stat[kLbaWindows] += 1;
                r->stat[kLbaBusyMs] += busy;
                r->stat[kLbaGpuMs] += info.gpu_ms;
                r->stat[kLbaSolveMs] += info.solve_ms;
                r->stat[kLbaSolves] += info.n_solves;
                r->stat[kLbaTrials] += info.lm_trials;
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

// New map points for the selected keypoints of the current frame: back-projection onto the plane z = plane_z of the
// first camera (swarmmap_amd/minitrack.py add_points), reference normal and scale-invariance distances as
// MapPoint::UpdateNormalAndDepth computes them (code/src/MapPoint.cc:395-433).
int add_points(so_replay* r, const M4& T, const so_replay::FrameHost& F, const std::vector<uint8_t>& sel, int n_sel) {
    const double fx = r->cam.fx, fy = r->cam.fy, cx = r->cam.cx, cy = r->cam.cy;
    const double* a = T.a;
    double Ow[3];
    for (int j = 0; j < 3; j++) Ow[j] = -(a[0 + j] * a[3] + a[4 + j] * a[7] + a[8 + j] * a[11]);
    r->new_X.resize((size_t)n_sel * 3); r->new_N.resize((size_t)n_sel * 3);
    r->new_max.resize((size_t)n_sel); r->new_min.resize((size_t)n_sel); r->new_desc.resize((size_t)n_sel * 32);
    int k = 0;
    for (int i = 0; i < F.n; i++) {
        if (!sel[(size_t)i]) continue;
        const double ray[3] = {((double)F.xy_un[2 * (size_t)i] - cx) / fx, ((double)F.xy_un[2 * (size_t)i + 1] - cy) / fy, 1.0};
        double dir[3];
        for (int j = 0; j < 3; j++) dir[j] = ray[0] * a[0 + j] + ray[1] * a[4 + j] + ray[2] * a[8 + j];  // R^T ray
        const double d = (r->plane_z - Ow[2]) / dir[2];
        double PO[3], X[3];
        for (int j = 0; j < 3; j++) {
            X[j] = Ow[j] + dir[j] * d;
            PO[j] = X[j] - Ow[j];
        }
        const double dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
        const double mx = dist * (double)r->scale[F.kps[(size_t)i].octave];
        for (int j = 0; j < 3; j++) {
            r->new_X[3 * (size_t)k + j] = (float)X[j];
            r->new_N[3 * (size_t)k + j] = (float)(PO[j] / dist);
        }
        r->new_max[(size_t)k] = (float)(1.2 * mx);
        r->new_min[(size_t)k] = (float)(0.8 * mx / (double)r->scale[r->nlevels - 1]);
        memcpy(&r->new_desc[32 * (size_t)k], &F.desc[32 * (size_t)i], 32);
        k++;
    }
    const int first = (int)(r->mp_X.size() / 3);
    if (so_map_write(r->map, first, n_sel, r->new_X.data(), r->new_N.data(), r->new_max.data(), r->new_min.data(),
                     r->new_desc.data()) != SO_OK)
        return -1;
    r->mp_X.insert(r->mp_X.end(), r->new_X.begin(), r->new_X.end());
    if (!r->vocab.empty()) {  // the local-mapping matcher job reads these through the keyframe snapshots
        r->mp_N.insert(r->mp_N.end(), r->new_N.begin(), r->new_N.end());
        r->mp_max.insert(r->mp_max.end(), r->new_max.begin(), r->new_max.end());
        r->mp_min.insert(r->mp_min.end(), r->new_min.begin(), r->new_min.end());
        r->mp_desc.insert(r->mp_desc.end(), r->new_desc.begin(), r->new_desc.begin() + 32 * (size_t)n_sel);
    }
    r->kf_first_slot.push_back(first);
    return first;
}

int submit_frame(so_replay* r, int t) {
    const int h = (r->submitted + 1) % 3;
    const uint8_t* img = r->frames[(size_t)t % r->frames.size()];
    const int rc = r->frames_on_device ? so_dframe_submit_device(r->fr[h], img, r->width, r->height, r->width)
                                       : so_dframe_submit(r->fr[h], img, r->width, r->height, r->width);
    if (rc != SO_OK) return fail(r, "so_dframe_submit");
    r->submitted = h;
    r->in_flight = true;
    return SO_OK;
}

}  // namespace

extern "C" {

// K4 = fx fy cx cy, dist5 = k1 k2 p1 p2 k3 (may be NULL = no distortion)
int so_replay_create(int device, int width, int height, int nfeatures, int lba_every, const float* K4, const float* dist5,
                     int keyframe_every, float keyframe_ratio, float plane_z, int local_keyframes, int third_pose,
                     so_replay** out) {
    if (!out || !K4) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    so_replay* r = new so_replay();
    r->device = device;
    r->width = width;
    r->height = height;
    r->lba_every = lba_every > 0 ? lba_every : 5;
    r->keyframe_every = keyframe_every > 0 ? keyframe_every : 8;
    r->keyframe_ratio = keyframe_ratio;
    r->plane_z = plane_z;
    r->local_keyframes = local_keyframes;
    r->third_pose = third_pose;
    r->cam.fx = K4[0]; r->cam.fy = K4[1]; r->cam.cx = K4[2]; r->cam.cy = K4[3];
    if (dist5) { r->cam.k1 = dist5[0]; r->cam.k2 = dist5[1]; r->cam.p1 = dist5[2]; r->cam.p2 = dist5[3]; r->cam.k3 = dist5[4]; }
    so_extractor_config cfg{nfeatures, 1.2f, 8, 20, 7, device};
    int rc = so_extractor_create(&cfg, &r->ex);
    if (rc == SO_OK) rc = so_dframe_create(r->ex, &r->cam, &r->fr[0]);
    if (rc == SO_OK) rc = so_dframe_create(r->ex, &r->cam, &r->fr[1]);
    if (rc == SO_OK) rc = so_dframe_create(r->ex, &r->cam, &r->fr[2]);
    if (rc == SO_OK) rc = so_matcher_create(device, &r->matcher);
    if (rc == SO_OK) rc = so_matcher_create(device, &r->matcher2);  // (same thread: same stream as `matcher`)
    if (rc == SO_OK) rc = so_map_create(device, &r->map);
    if (rc == SO_OK) rc = so_ba_create(device, &r->tracker_opt);
    if (rc == SO_OK) rc = so_ba_create(device, &r->mapper_opt);
    if (rc == SO_OK) rc = so_matcher_create(device, &r->mapper_matcher);
    if (rc != SO_OK) {
        delete r;
        return rc;
    }
    float inv[8], s2[8];
    int32_t npl[8];
    so_extractor_tables(r->ex, r->scale, inv, s2, r->inv_sigma2, npl);
    r->log_sf = (float)std::log((double)1.2f);  // log(mfScaleFactor), MapPoint.cc:478 (float scale factor)
    r->cap = so_extractor_capacity(r->ex);
    {   // sized once, like a tracker sizes its buffers for the local map it allows: 12 keyframes' worth of new points,
        // or 32 k map points when the whole map is searched
        const int reserve_q = local_keyframes > 0 ? (local_keyframes + 1) * r->cap : 32768;
        if (so_matcher_reserve(r->matcher, reserve_q) != SO_OK || so_matcher_reserve(r->matcher2, reserve_q) != SO_OK) {
            delete r;
            return SO_ERR_HIP;
        }
        r->skip.reserve((size_t)reserve_q);
        r->new_desc.reserve((size_t)reserve_q);
        r->last_slot.reserve((size_t)r->cap);
        r->excluded.reserve((size_t)r->cap);
        r->k2l.reserve((size_t)r->cap);
        r->k2m.reserve((size_t)r->cap);
    }
    for (auto& f : r->fh) {
        f.kps.resize((size_t)r->cap);
        f.xy_un.resize((size_t)r->cap * 2);
        f.desc.resize((size_t)r->cap * 32);
        f.kp_mp.resize((size_t)r->cap);
        f.outlier.resize((size_t)r->cap);
    }
    if (const char* e = getenv("SWARMORB_LM_BATCH")) r->lm_batch = atoi(e) != 0;
    if (const char* e = getenv("SWARMORB_LM_RESIDENT")) r->lm_resident = atoi(e) != 0;
    if (const char* e = getenv("SWARMORB_LM_MAP")) r->lm_from_map = atoi(e) != 0;
    if (!getenv("SWARMORB_NO_PIN")) {
        // one last-level-cache group of the device's NUMA node per agent: GPU d's first agent takes group d, the next
        // LIVE agents of this process the groups behind it (a destroyed agent's group is free again: replays that follow
        // each other in one process - bench configs, a test session - stay on the same groups)
        r->pin_slot = take_pin_slot();
        char cpus[512];
        const int base = getenv("SWARMORB_PIN_SLOT_BASE") ? atoi(getenv("SWARMORB_PIN_SLOT_BASE")) : 0;  // (several processes on one GPU)
        if (so_device_host_cpus(device, device + base + r->pin_slot, cpus, (int)sizeof(cpus)) == SO_OK) r->host_cpus = cpus;
    }
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
        auto& f = r->fh[0];
        (void)so_dframe_collect(r->fr[r->submitted], f.kps.data(), nullptr, f.desc.data(), r->cap, &n, nullptr);
    }
    so_extractor_group_destroy(r->fleet_group);
    so_track_group_destroy(r->fleet_track_group);
    so_ba_group_destroy(r->fleet_ba_group);  // (freed when the last agent's solver has left it)
    for (so_dframe* f : r->fr) so_dframe_destroy(f);
    so_extractor_destroy(r->ex);
    so_matcher_destroy(r->matcher);
    so_matcher_destroy(r->matcher2);
    so_map_destroy(r->map);
    so_ba_destroy(r->tracker_opt);
    so_ba_destroy(r->mapper_opt);
    so_matcher_destroy(r->mapper_matcher);
    give_pin_slot(r->pin_slot);
    delete r;
}

const char* so_replay_error(so_replay* r) { return r ? r->error.c_str() : "null handle"; }

// frames: n pointers to width x height u8 images, row stride = width; host memory (pinned for asynchronous uploads)
// or device memory
int so_replay_set_frames(so_replay* r, const uint64_t* pointers, int n, int on_device) {
    if (!r || !pointers || n <= 0) return SO_ERR_INVALID_ARG;
    r->frames.clear();
    for (int i = 0; i < n; i++) r->frames.push_back(reinterpret_cast<const uint8_t*>(pointers[i]));
    r->frames_on_device = on_device != 0;
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

// Resource allocation before any step is counted: one window through the local-mapping handle sizes its device
// buffers (what a process does once at start-up, not per frame).
int so_replay_preallocate(so_replay* r) {
    if (!r || r->window.epose.empty()) return SO_ERR_INVALID_ARG;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        r->queue.push_back(LmJob{});
    }
    r->cv.notify_all();
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv.wait(lk, [r] { return r->queue.empty() && !r->running; });
    return r->error.empty() ? SO_OK : SO_ERR_HIP;
}

// Vocabulary stand-in for the local-mapping matcher job: n centroid descriptors (32 B each).  Set before the first
// frame; without it the local-mapping thread only optimises its windows.
int so_replay_set_vocabulary(so_replay* r, const uint8_t* centroids, int n, int neighbours) {
    if (!r || !centroids || n <= 0 || r->n_tracked > 0) return SO_ERR_INVALID_ARG;
    r->vocab.assign(centroids, centroids + 32 * (size_t)n);
    if (neighbours > 0) r->lm_neighbours = neighbours;
    return SO_OK;
}
// statistics of the timed matcher jobs (indices: the kLm* enumeration above) and the log of every job:
// 6 ints each = frame index, neighbours, SearchForTriangulation matches, points fused into neighbours, fused back, new map points
int so_replay_lm_stats(so_replay* r, double* out16) {  // (40 doubles)
    if (!r || !out16) return SO_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(r->mu);
    memcpy(out16, r->lm_stat, sizeof(r->lm_stat));
    return SO_OK;
}
int so_replay_lm_log(so_replay* r, int32_t* out, int cap_rows) {
    if (!r) return SO_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(r->mu);
    const int rows = (int)(r->lm_log.size() / 6);
    if (out) memcpy(out, r->lm_log.data(), sizeof(int32_t) * 6 * (size_t)std::min(rows, cap_rows));
    return rows;
}

// The closed loop (closedloop.cc): every keyframe's local-mapping results - triangulated points, fused duplicates, the poses
// and points local BA moved over the keyframe's OWN window, the observations it rejected - flow back into the tracked map.
// Needs a vocabulary and local_keyframes > 0; set before the first frame.  kf_every: frames between keyframes; delay:
// frames between a keyframe and the arrival of its results in the tracked map (policy 0, the deterministic schedule:
// the tracking thread waits for the job if it is not done by then); policy 1: the reference's own policy - results arrive
// when they are ready, a keyframe is only made while local mapping is idle, a busy local mapper gets InterruptBA
// (Tracking.cc:880-890, LocalMapping.cc:581-583).  n_free / n_fixed: caps of the window's free / fixed keyframes.
// The tracking stages as one chain of launches each (so_track_stage_*: the default) or as the separate calls with the resolve
// and the pose-problem gather on the host: same results to the bit (tests/test_closedloop_gpu.py), different latency
// so_fleet_run tracks frame (tick + offset) of this agent at every tick: agents that were run on their own for `offset` frames first
// (so_replay_run(r, 0, offset)) are that many frames ahead of the fleet's clock, so their keyframes - one every kf_every frames
// of their OWN stream - fall on different ticks and their local-mapping jobs do not all start in the same instant.
int so_replay_set_fleet_offset(so_replay* r, int offset) {
    if (!r || offset < 0) return SO_ERR_INVALID_ARG;
    r->fleet_offset = offset;
    return SO_OK;
}

// Elastic ticks of the fleet this agent leads (agents[0] of so_fleet_run): ticks driven so far and the agents they took in
// total - slots / (ticks * agents) is the share of tick places that were filled.
int so_replay_fleet_ticks(so_replay* r, long long* ticks, long long* slots) {
    if (!r) return SO_ERR_INVALID_ARG;
    if (ticks) *ticks = r->fleet_ticks;
    if (slots) *slots = r->fleet_tick_slots;
    return SO_OK;
}

int so_replay_set_track_chain(so_replay* r, int on) {
    if (!r) return SO_ERR_INVALID_ARG;
    r->track_chain = on ? 1 : 0;
    return SO_OK;
}

int so_replay_set_closed_loop(so_replay* r, int kf_every, int delay, int n_free, int n_fixed, int policy) {
    if (!r || r->n_tracked > 0 || r->vocab.empty() || r->local_keyframes <= 0 || kf_every < 1 || delay < 1 || n_free < 1 || n_fixed < 0)
        return SO_ERR_INVALID_ARG;
    std::unique_ptr<ClosedLoop> cl(new ClosedLoop());
    cl->kf_every = kf_every; cl->delay = delay; cl->n_free = n_free; cl->n_fixed = n_fixed; cl->policy = policy ? 1 : 0;
    {   // the local-mapping thread exists since so_replay_create and looks at r->cl under this mutex (found by `make host-tsan`)
        std::lock_guard<std::mutex> lk(r->mu);
        r->cl = std::move(cl);
    }
    return SO_OK;
}
// counts[0..7]: local-mapping jobs, windows solved, windows aborted by the stop flag, InterruptBA calls, map slots, bad
// points, keyframes, microseconds the tracking thread spent applying packets; wait_ms: time the tracking thread waited for packets (deterministic schedule)
int so_replay_cl_counts(so_replay* r, int64_t* counts8, double* wait_ms) {
    if (!r || !r->cl || !counts8) return SO_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(r->mu);  // (call after so_replay_drain: the local-mapping side is read here)
    ClosedLoop& M = *r->cl;
    int64_t nbad = 0;
    for (uint8_t b : M.bad) nbad += b;
    const int64_t c[8] = {(int64_t)(M.lm_log.size() / 12), M.windows, M.aborted, M.interrupts, (int64_t)M.bad.size(), nbad, (int64_t)M.kfs.size(), (int64_t)(M.apply_ms * 1e3)};
    memcpy(counts8, c, sizeof(c));
    if (wait_ms) *wait_ms = M.wait_ms;
    return SO_OK;
}
// the local-mapping log: 12 int64 per job (swarmmap_amd/closedloop.py LM_LOG_COLUMNS); returns the number of rows
int so_replay_cl_log(so_replay* r, int64_t* out, int cap_rows) {
    if (!r || !r->cl) return SO_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(r->mu);
    const int rows = (int)(r->cl->lm_log.size() / 12);
    if (out) memcpy(out, r->cl->lm_log.data(), sizeof(int64_t) * 12 * (size_t)std::min(rows, cap_rows));
    return rows;
}
// after so_replay_drain: the keyframes' frame indices and final poses (12 floats each), and per tracked frame the id of its
// reference keyframe and its pose relative to it (16 doubles) - System::SaveTrajectoryTUM composes the two
// (code/src/System.cc:225-252)
int so_replay_cl_keyframes(so_replay* r, int32_t* kf_t, float* poses12, int cap) {
    if (!r || !r->cl) return SO_ERR_INVALID_ARG;
    const int n = (int)r->cl->kfs.size();
    for (int i = 0; i < std::min(n, cap); i++) {
        if (kf_t) kf_t[i] = r->cl->kfs[(size_t)i]->t;
        if (poses12) memcpy(poses12 + 12 * (size_t)i, r->cl->kfs[(size_t)i]->T, 48);
    }
    return n;
}
// after so_replay_drain: keyframe `kf`'s keypoint -> map slot bindings as local mapping left them (mvpMapPoints after
// ProcessNewKeyFrame / CreateNewMapPoints / SearchInNeighbors / local BA's EraseMapPointMatch); returns the keypoint count
int so_replay_cl_bindings(so_replay* r, int kf, int32_t* mp, int cap) {
    if (!r || !r->cl || kf < 0 || kf >= (int)r->cl->kfs.size()) return SO_ERR_INVALID_ARG;
    const KfSnap& k = *r->cl->kfs[(size_t)kf];
    if (mp) memcpy(mp, k.mp.data(), 4 * (size_t)std::min(k.n, cap));
    return k.n;
}
// ... and the map's bad flags + replaced-by slots (mbBad, mpReplaced); returns the number of slots
int so_replay_cl_points(so_replay* r, uint8_t* bad, int32_t* replaced_by, int cap) {
    if (!r || !r->cl) return SO_ERR_INVALID_ARG;
    const int n = (int)r->cl->bad.size();
    if (bad) memcpy(bad, r->cl->bad.data(), (size_t)std::min(n, cap));
    if (replaced_by) memcpy(replaced_by, r->cl->repl.data(), 4 * (size_t)std::min(n, cap));
    return n;
}
int so_replay_cl_frames(so_replay* r, int32_t* ref_kf, double* Tcr16, int cap) {
    if (!r || !r->cl) return SO_ERR_INVALID_ARG;
    const int n = (int)r->cl->ref_log.size();
    const int m = std::min(n, cap);
    if (ref_kf) memcpy(ref_kf, r->cl->ref_log.data(), 4 * (size_t)m);
    if (Tcr16) memcpy(Tcr16, r->cl->Tcr_log.data(), 8 * 16 * (size_t)m);
    return n;
}

int so_replay_set_profiling(so_replay* r, int enabled) { return r ? so_extractor_set_profiling(r->ex, enabled) : SO_ERR_INVALID_ARG; }

// Runs frames [first_t, first_t + n_steps).  Frame first_t must be in flight (so_replay_prime) - the loop collects
// it, submits the next one and leaves that one in flight when it returns.
int so_replay_prime(so_replay* r, int t) {
    if (!r || r->frames.empty() || r->in_flight) return SO_ERR_INVALID_ARG;
    return submit_frame(r, t);
}

}  // extern "C"

// ---- one tracked frame as stages.  so_replay_run chains them for one agent; so_fleet_run walks several agents through
//      them in lockstep (searches submitted for all agents before any is waited for, PoseOptimization of all agents in
//      one launch). ----
namespace {

int step_m2_submit(so_replay* r);
void step_m1_submit_linked(so_replay* r);

// Closed loop, deterministic schedule: `kf_every` frames after the last keyframe the frame being tracked WILL be one (local
// mapping has just handed its results over and is idle).  Its keypoints and descriptors are known as soon as the frame is
// collected: the local-mapping thread computes the feature vector and uploads the keyframe into HBM now, while this thread
// runs the frame's searches and PoseOptimization calls; bindings and pose follow when the frame is tracked (queue_keyframe).
void prepare_keyframe(so_replay* r, int t) {
    ClosedLoop& M = *r->cl;
    static const bool off = getenv("SWARMORB_CL_NO_PREPARE") != nullptr;
    r->pre_kf.reset();
    if (off || M.policy != 0 || M.job_pending || t - M.last_kf_t < M.kf_every) return;
    const so_replay::FrameHost& F = r->fh[r->cur];
    auto k = std::make_shared<KfSnap>();
    const int n = F.n;
    k->n = n;
    k->t = t;
    k->x.resize((size_t)n); k->y.resize((size_t)n); k->angle.resize((size_t)n); k->octave.resize((size_t)n);
    k->mp.assign((size_t)n, -1);
    k->desc.assign(F.desc.begin(), F.desc.begin() + 32 * (size_t)n);
    for (int i = 0; i < n; i++) {
        k->x[(size_t)i] = F.xy_un[2 * (size_t)i];
        k->y[(size_t)i] = F.xy_un[2 * (size_t)i + 1];
        k->angle[(size_t)i] = F.kps[(size_t)i].angle;
        k->octave[(size_t)i] = F.kps[(size_t)i].octave;
    }
    memcpy(k->bounds, r->bounds, sizeof(k->bounds));
    r->pre_kf = k;
    LmJob job;
    job.kf = k;
    job.pre = true;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->queue.push_back(job);
    }
    r->cv.notify_all();
}

// Frame constructor: collect frame t, put frame t+1 in flight.  The first frame of a run initialises the map.
int step_begin(so_replay* r, int t, bool submit_next = true) {
    so_replay::Step& S = r->step;
    S = so_replay::Step{};
    S.t = t;
    S.t0 = now_ms();
    // kernel times come from HIP events on every kEventEvery-th frame of the run (two event records per search and
    // per PoseOptimization call are ~11 us of a 0.43 ms frame when taken on every frame)
    S.timed_kernels = (t % kEventEvery) == 0;
    so_matcher_set_profiling(r->matcher, S.timed_kernels);
    so_matcher_set_profiling(r->matcher2, S.timed_kernels);
    so_pose_optimization_set_timing(r->tracker_opt, S.timed_kernels);
    S.hcur = r->submitted;
    if (r->cl) {  // what local mapping handed back arrives between two frames (Optimizer.cc:713 under Map::mMutexMapUpdate)
        const int rc = cl_frame_begin(r, t);
        if (rc) return rc;
    }
    r->cur ^= 1;
    so_replay::FrameHost& F = r->fh[r->cur];
    int n = 0;
    if (so_dframe_wait(r->fr[S.hcur], &n, r->bounds) != SO_OK) return fail(r, "so_dframe_wait");
    F.n = n;
    S.first = r->n_tracked == 0;
    // the frame is complete on the device: the motion-model search goes out now and runs under the host copies of the
    // keypoints and under the submission of the next frame
    if (!S.first && (S.m2_rc = step_m2_submit(r))) return S.m2_rc;
    S.m2_submitted = !S.first;
    if (so_dframe_collect(r->fr[S.hcur], F.kps.data(), F.xy_un.data(), F.desc.data(), r->cap, &n, r->bounds) != SO_OK)
        return fail(r, "so_dframe_collect");
    r->in_flight = false;
    S.first = r->n_tracked == 0;
    // the next frame goes to the extractor here - or, on the single-agent path, under this frame's first
    // PoseOptimization kernel, where the tracking thread would only be waiting (so_replay_run)
    S.next_submitted = submit_next || S.first;
    if (r->live) {  // a live camera has not produced frame t + 1 yet
        S.next_submitted = true;
    } else if (S.next_submitted) {
        const int rc = submit_frame(r, t + 1);
        if (rc) return rc;
    }
    if (r->cl && !S.first) prepare_keyframe(r, t);
    S.t1 = S.tm2 = S.tp1 = S.tm1 = S.tp2 = S.tp3 = S.tmap = now_ms();
    S.n_in = n;
    S.map_size_at_begin = (int)(r->mp_X.size() / 3);
    for (int i = 0; i < n; i++) F.kp_mp[(size_t)i] = -1;
    memset(F.outlier.data(), 0, (size_t)n);
    S.T = M4::eye();
    if (S.first) {  // every keypoint becomes a map point, the camera defines the world frame
        std::vector<uint8_t> all((size_t)n, 1);
        const int first = add_points(r, S.T, F, all, n);
        if (first < 0) return fail(r, "so_map_write");
        for (int i = 0; i < n; i++) F.kp_mp[(size_t)i] = first + i;
        r->kf_inliers = n;
        S.keyframe = 1;
        if (r->cl) {  // the initial map is the local map until the first keyframe's job reports
            r->cl->tv_bad.assign((size_t)n, 0);
            r->cl->tv_repl.assign((size_t)n, -1);
            r->cl->tv_vis.assign((size_t)n, 1);
            r->cl->tv_found.assign((size_t)n, 1);
            r->cl->recent_from = n;
            r->cl->tv_local.resize((size_t)n);
            for (int i = 0; i < n; i++) r->cl->tv_local[(size_t)i] = first + i;
        }
        S.tm2 = S.tp1 = S.tm1 = S.tp2 = S.tp3 = S.tmap = now_ms();
    }
    return SO_OK;
}

// TrackWithMotionModel's search (Tracking.cc:714-741)
// SWARMORB_TRACK_CHAIN=0: the tracking stages as separate calls with the resolve and the pose-problem gather on the host (rounds 2-4)
bool track_chain_on(const so_replay* r) {
    static const bool on = !(getenv("SWARMORB_TRACK_CHAIN") && atoi(getenv("SWARMORB_TRACK_CHAIN")) == 0);
    return r->track_chain >= 0 ? r->track_chain != 0 : on;
}

// TrackLocalMap's stage enqueued behind TrackWithMotionModel's (so_track_stage_local_map_submit_after): what the stage needs from
// the first one - pose, bindings, excluded keypoints, already-matched local points - is handed over on the device; the host's
// share is the list of local points and their bad flags, both known when the frame begins.  SWARMORB_TRACK_LINK=0: off.
void step_m1_submit_linked(so_replay* r) {
    static const bool off = getenv("SWARMORB_TRACK_LINK") && atoi(getenv("SWARMORB_TRACK_LINK")) == 0;
    so_replay::Step& S = r->step;
    if (off || r->lockstep) return;
    int rc;
    if (r->cl) {
        ClosedLoop& M = *r->cl;
        // Not on the frame that is about to become a keyframe under the deterministic schedule: there the local-mapping thread
        // prepares the keyframe (feature vector + upload, prepare_keyframe) while this frame is tracked, and its few kernels then
        // queue behind the back-to-back chain of both stages instead of slipping into the gap between them - the keyframe was
        // handed over 80 us later (0.374 against 0.294 ms after the packet, measured) on the one path that bounds the loop.
        if (M.policy == 0 && !M.job_pending && S.t - M.last_kf_t >= M.kf_every) return;
        const int nl = (int)M.tv_local.size();
        r->skip_static.resize((size_t)nl);
        for (int i = 0; i < nl; i++) r->skip_static[(size_t)i] = M.tv_bad[(size_t)M.tv_local[(size_t)i]];
        rc = so_track_stage_local_map_submit_after(r->matcher2, r->matcher, r->fr[S.hcur], r->map, nl, M.tv_local.data(), 0, r->skip_static.data(),
                                                   1.0f, 0.8f, 0.5f, r->log_sf, r->K4, r->inv_sigma2);
    } else {
        const int n_map = (int)(r->mp_X.size() / 3);
        int first = 0;
        if (r->local_keyframes > 0 && (int)r->kf_first_slot.size() > r->local_keyframes)
            first = r->kf_first_slot[r->kf_first_slot.size() - (size_t)r->local_keyframes];
        rc = so_track_stage_local_map_submit_after(r->matcher2, r->matcher, r->fr[S.hcur], r->map, n_map - first, nullptr, first, nullptr, 1.0f, 0.8f,
                                                   0.5f, r->log_sf, r->K4, r->inv_sigma2);
    }
    S.stage2_linked = rc == SO_OK;  // (anything else: nothing was enqueued, the stage goes out after the first one as before)
}

int step_m2_submit(so_replay* r) {
    so_replay::Step& S = r->step;
    so_replay::FrameHost& L = r->fh[r->cur ^ 1];
    const M4 T_pred = mul(r->velocity, r->T_last);
    to_f12(T_pred, S.Tp);
    r->last_slot.resize((size_t)L.n);
    for (int i = 0; i < L.n; i++)
        r->last_slot[(size_t)i] = (L.kp_mp[(size_t)i] >= 0 && !L.outlier[(size_t)i]) ? L.kp_mp[(size_t)i] : -1;
    if (track_chain_on(r) && (!r->lockstep || r->fleet_chain)) {
        // the whole stage as one chain of launches: search -> resolve on the device -> PoseOptimization; one wait (step_stage1_wait)
        r->K4[0] = r->cam.fx; r->K4[1] = r->cam.fy; r->K4[2] = r->cam.cx; r->K4[3] = r->cam.cy;
        const int rc = so_track_stage_last_frame_submit(r->matcher, r->fr[S.hcur], r->fr[(S.hcur + 2) % 3], r->map, S.Tp,
                                                        r->last_slot.data(), 15.0f, 1, r->K4, r->inv_sigma2);
        if (rc == SO_OK) {
            S.stage1_dev = true;
            step_m1_submit_linked(r);  // (TrackLocalMap's stage right behind it, when it can be: no host between the stages)
            return SO_OK;
        }
        if (rc != SO_RETRY_ON_HOST) return fail(r, "so_track_stage_last_frame_submit");
    }
    if (so_track_search_last_frame_submit(r->matcher, r->fr[S.hcur], nullptr, r->fr[(S.hcur + 2) % 3], r->map, S.Tp,
                                          r->last_slot.data(), 15.0f) != SO_OK)
        return fail(r, "so_track_search_last_frame_submit");
    return SO_OK;
}

int step_m2_wait(so_replay* r) {
    so_replay::Step& S = r->step;
    so_replay::FrameHost& F = r->fh[r->cur];
    so_replay::FrameHost& L = r->fh[r->cur ^ 1];
    const int n = F.n;
    r->k2l.resize((size_t)n);
    int32_t nm = 0;
    float kms = 0.f;
    double ms4[4];
    if (so_track_search_last_frame_wait(r->matcher, nullptr, 1, r->k2l.data(), &nm) != SO_OK)
        return fail(r, "so_track_search_last_frame_wait");
    so_matcher_last_kernel_ms(r->matcher, &kms);
    S.match_kernel += kms;
    so_matcher_last_stats(r->matcher, ms4);
    S.mstat[0] += ms4[0]; S.mstat[1] += ms4[1];
    S.reruns += ms4[2];
    if (nm < 20) {  // Tracking.cc:733-737: wider window
        S.wide_m2 += 1;
        if (so_track_search_last_frame(r->matcher, r->fr[S.hcur], nullptr, r->fr[(S.hcur + 2) % 3], r->map, S.Tp,
                                       r->last_slot.data(), nullptr, 30.0f, 1, r->k2l.data(), &nm) != SO_OK)
            return fail(r, "so_track_search_last_frame");
        so_matcher_last_kernel_ms(r->matcher, &kms);
        S.match_kernel += kms;
    }
    S.nm2 = nm;
    for (int k = 0; k < n; k++)
        if (r->k2l[(size_t)k] >= 0) F.kp_mp[(size_t)k] = L.kp_mp[(size_t)r->k2l[(size_t)k]];
    S.tm2 = now_ms();
    return SO_OK;
}

// Optimizer::PoseOptimization inputs of the current frame: the keypoints that have a map point, ascending index
// same_edges: the keypoint -> map point bindings have not changed since the last gather of this frame (the third call of
// a frame runs over the second one's edges from another start pose): the arrays are still right
void pose_gather(so_replay* r, const float* T_in12, float* T_out12, int32_t* n_inliers, int32_t* info2, so_pose_problem* q,
                 bool same_edges = false) {
    const so_replay::FrameHost& F = r->fh[r->cur];
    if (!same_edges) {
        r->idx.clear();
        for (int i = 0; i < F.n; i++)
            if (F.kp_mp[(size_t)i] >= 0) r->idx.push_back(i);
    }
    const int np = (int)r->idx.size();
    r->pX.resize((size_t)np * 3); r->pobs.resize((size_t)np * 2); r->pw.resize((size_t)np); r->pose_out.assign((size_t)np, 0);
    for (int k = 0; k < (same_edges ? 0 : np); k++) {
        const int i = r->idx[(size_t)k];
        const size_t s = (size_t)F.kp_mp[(size_t)i];
        memcpy(&r->pX[3 * (size_t)k], &r->mp_X[3 * s], 12);
        r->pobs[2 * (size_t)k] = F.xy_un[2 * (size_t)i];
        r->pobs[2 * (size_t)k + 1] = F.xy_un[2 * (size_t)i + 1];
        r->pw[(size_t)k] = r->inv_sigma2[F.kps[(size_t)i].octave];
    }
    r->K4[0] = r->cam.fx; r->K4[1] = r->cam.fy; r->K4[2] = r->cam.cx; r->K4[3] = r->cam.cy;
    memcpy(T_out12, T_in12, 48);
    *n_inliers = 0;
    info2[0] = info2[1] = 0;
    q->Tcw12 = T_in12; q->intr = r->K4; q->n = np; q->Xw = r->pX.data(); q->obs = r->pobs.data(); q->inv_sigma2 = r->pw.data();
    q->Tcw_out12 = T_out12; q->outlier = r->pose_out.data(); q->n_inliers = n_inliers; q->info = info2;
}

void pose_account(so_replay* r, const so_pose_problem& q, float kernel_ms) {
    so_replay::Step& S = r->step;
    S.pose_calls++;
    if (!S.timed_kernels) return;  // kernel time, trials and points are accumulated over the event-timed calls only
    S.pose_kernel += kernel_ms;
    S.pose_trials += q.info[1];
    S.pose_points += q.n;
    S.pose_timed_calls++;
}

// under_kernel (may be empty): host work of the tracking thread that does not depend on this call's result; it runs
// between the launch and the wait
template <typename F>
int pose_single(so_replay* r, const float* T_in12, float* T_out12, int32_t* n_inliers, F under_kernel, bool same_edges = false) {
    so_pose_problem q;
    int32_t info2[2];
    pose_gather(r, T_in12, T_out12, n_inliers, info2, &q, same_edges);
    if (so_pose_optimization_submit(r->tracker_opt, q.Tcw12, q.intr, q.n, q.Xw, q.obs, q.inv_sigma2) != SO_OK)
        return fail(r, "so_pose_optimization_submit");
    const int rc = under_kernel();
    if (so_pose_optimization_wait(r->tracker_opt, q.Tcw_out12, q.outlier, q.n_inliers, q.info) != SO_OK)
        return fail(r, "so_pose_optimization_wait");
    if (rc) return rc;
    float ms = 0.f;
    if (r->step.timed_kernels) so_pose_optimization_last_kernel_ms(r->tracker_opt, &ms);
    pose_account(r, q, ms);
    return SO_OK;
}
int pose_single(so_replay* r, const float* T_in12, float* T_out12, int32_t* n_inliers) {
    return pose_single(r, T_in12, T_out12, n_inliers, [] { return 0; });
}

// A stage that ran on the device: the matches, the edge list (= the keypoints with a map point, ascending: what pose_gather
// builds), the outlier flags and the pose come back together.  false: not finished on the device - the caller repeats the
// stage with the separate calls.
bool stage_collect(so_replay* r, std::vector<int32_t>& kp_to_q, int32_t* nm, uint8_t* in_view, float* T_out12, int32_t* n_inliers,
                   int* rc_out, so_matcher* sm = nullptr) {
    so_replay::Step& S = r->step;
    const bool linked = sm != nullptr && sm != r->matcher;
    if (!sm) sm = r->matcher;
    const so_replay::FrameHost& F = r->fh[r->cur];
    kp_to_q.resize((size_t)F.n);
    r->idx.resize((size_t)F.n);
    r->pose_out.resize((size_t)F.n);
    int32_t ne = 0, info2[2] = {0, 0};
    const int rc = so_track_stage_wait(sm, kp_to_q.data(), nm, in_view, &ne, r->idx.data(), r->pose_out.data(), T_out12,
                                       n_inliers, info2);
    *rc_out = SO_OK;
    if (rc == SO_RETRY_ON_HOST) return false;
    if (rc != SO_OK) {
        *rc_out = fail(r, "so_track_stage_wait");
        return false;
    }
    r->idx.resize((size_t)ne);
    r->pose_out.resize((size_t)ne);
    S.pose_calls++;
    if (S.timed_kernels && !linked) {  // (a linked stage carries no events: its pose kernel is timed on the frames that are not linked... see kEventEvery)
        float kms = 0.f;
        so_track_stage_last_pose_kernel_ms(sm, &kms);
        S.pose_kernel += kms;
        S.pose_trials += info2[1];
        S.pose_points += ne;
        S.pose_timed_calls++;
    }
    return true;
}

// TrackWithMotionModel on the device chain: the matches of so_track_search_last_frame + the first PoseOptimization
int step_stage1_wait(so_replay* r, int32_t* inl) {
    so_replay::Step& S = r->step;
    so_replay::FrameHost& F = r->fh[r->cur];
    so_replay::FrameHost& L = r->fh[r->cur ^ 1];
    int32_t nm = 0;
    int rc = SO_OK;
    const bool ok = stage_collect(r, r->k2l, &nm, nullptr, S.Ta, inl, &rc);
    if (rc) return rc;
    float kms = 0.f;
    double ms4[4];
    so_matcher_last_kernel_ms(r->matcher, &kms);
    S.match_kernel += kms;
    so_matcher_last_stats(r->matcher, ms4);
    S.mstat[0] += ms4[0]; S.mstat[1] += ms4[1];
    S.reruns += ms4[2];
    if ((!ok || nm < 20) && S.stage2_linked) {
        // the stage enqueued behind this one ran on what this one left: wait for it, drop its result
        std::vector<int32_t> k2((size_t)F.n), ek((size_t)F.n);
        std::vector<uint8_t> eo((size_t)F.n), vw((size_t)std::max(1, (int)r->skip_static.size() + (int)(r->mp_X.size() / 3)));
        int32_t nm2 = 0, ne2 = 0, inl2 = 0, info2[2];
        float T2[12];
        so_track_stage_set_start_pose(r->matcher2, S.Tp);
        const int rc2 = so_track_stage_wait(r->matcher2, k2.data(), &nm2, vw.data(), &ne2, ek.data(), eo.data(), T2, &inl2, info2);
        if (rc2 != SO_OK && rc2 != SO_RETRY_ON_HOST) return fail(r, "so_track_stage_wait (linked stage)");
        S.stage2_linked = false;
    }
    if (!ok || nm < 20) {  // not finished on the device, or Tracking.cc:733-737's wider window: the separate calls
        S.stage1_dev = false;
        const float th = ok ? 30.0f : 15.0f;
        if (ok) S.wide_m2 += 1;
        r->k2l.resize((size_t)F.n);
        if (so_track_search_last_frame(r->matcher, r->fr[S.hcur], nullptr, r->fr[(S.hcur + 2) % 3], r->map, S.Tp, r->last_slot.data(),
                                       nullptr, th, 1, r->k2l.data(), &nm) != SO_OK)
            return fail(r, "so_track_search_last_frame");
        so_matcher_last_kernel_ms(r->matcher, &kms);
        S.match_kernel += kms;
        if (!ok && nm < 20) {
            S.wide_m2 += 1;
            if (so_track_search_last_frame(r->matcher, r->fr[S.hcur], nullptr, r->fr[(S.hcur + 2) % 3], r->map, S.Tp,
                                           r->last_slot.data(), nullptr, 30.0f, 1, r->k2l.data(), &nm) != SO_OK)
                return fail(r, "so_track_search_last_frame");
        }
    }
    S.nm2 = nm;
    for (int k = 0; k < F.n; k++)
        if (r->k2l[(size_t)k] >= 0) F.kp_mp[(size_t)k] = L.kp_mp[(size_t)r->k2l[(size_t)k]];
    S.tm2 = now_ms();
    if (!S.stage1_dev) return pose_single(r, S.Tp, S.Ta, inl);
    return SO_OK;
}

void pose1_apply(so_replay* r) {  // Tracking.cc:745-760: outliers lose their map point
    so_replay::FrameHost& F = r->fh[r->cur];
    for (size_t k = 0; k < r->idx.size(); k++)
        if (r->pose_out[k]) F.kp_mp[(size_t)r->idx[k]] = -1;
    r->step.tp1 = now_ms();
}

// TrackLocalMap's search (Tracking.cc:770-807, 964-1007)
int step_m1_submit(so_replay* r) {
    so_replay::Step& S = r->step;
    so_replay::FrameHost& F = r->fh[r->cur];
    const int n = F.n;
    if (r->cl) {  // the closed loop's local map: the slots local mapping listed when it finished its last keyframe
        ClosedLoop& M = *r->cl;
        const int nl = (int)M.tv_local.size();
        S.first_slot = 0;
        S.n_local = nl;
        r->skip.resize((size_t)nl);
        for (int i = 0; i < nl; i++) r->skip[(size_t)i] = M.tv_bad[(size_t)M.tv_local[(size_t)i]];
        r->excluded.resize((size_t)n);
        // slot -> bound in this frame, as the reference marks it: mnLastFrameSeen = the frame's id (a table cleared per frame
        // would cost a memset of the whole map's size on every frame)
        if (M.tv_seen.size() < r->mp_X.size() / 3) M.tv_seen.resize(r->mp_X.size() / 3, -1);
        for (int k = 0; k < n; k++) {
            const int s = F.kp_mp[(size_t)k];
            r->excluded[(size_t)k] = s >= 0 ? 1 : 0;
            if (s >= 0) {
                M.tv_seen[(size_t)s] = S.t;
                M.tv_vis[(size_t)s]++;  // SearchLocalPoints: points already matched (Tracking.cc:966-975)
            }
        }
        for (int i = 0; i < nl; i++)
            if (M.tv_seen[(size_t)M.tv_local[(size_t)i]] == S.t) r->skip[(size_t)i] = 1;  // already matched: mbTrackInView = false (:966-978)
        if (S.stage2_linked) {  // in flight behind stage 1 already (step_m1_submit_linked): the arrays above are the host's own bookkeeping
            S.stage2_dev = true;  // and the inputs of the separate calls should the stage be handed back
            return SO_OK;
        }
        if (track_chain_on(r) && (!r->lockstep || r->fleet_chain)) {
            const int rc = so_track_stage_local_map_submit(r->matcher, r->fr[S.hcur], F.kp_mp.data(), S.stage1_dev ? 1 : 0, r->map, S.Ta, nl, M.tv_local.data(), 0,
                                                           r->skip.data(), 1.0f, 0.8f, 0.5f, r->log_sf, r->K4, r->inv_sigma2);
            if (rc == SO_OK) {
                S.stage2_dev = true;
                return SO_OK;
            }
            if (rc != SO_RETRY_ON_HOST) return fail(r, "so_track_stage_local_map_submit");
        }
        if (so_track_search_local_map_submit(r->matcher, r->fr[S.hcur], r->excluded.data(), r->map, S.Ta, nl, M.tv_local.data(), 0,
                                             r->skip.data(), 1.0f, 0.8f, 0.5f, r->log_sf) != SO_OK)
            return fail(r, "so_track_search_local_map_submit");
        return SO_OK;
    }
    const int n_map = (int)(r->mp_X.size() / 3);
    int first = 0;
    if (r->local_keyframes > 0 && (int)r->kf_first_slot.size() > r->local_keyframes)
        first = r->kf_first_slot[r->kf_first_slot.size() - (size_t)r->local_keyframes];
    S.first_slot = first;
    S.n_local = n_map - first;
    r->skip.assign((size_t)S.n_local, 0);
    r->excluded.resize((size_t)n);
    for (int k = 0; k < n; k++) {
        const int s = F.kp_mp[(size_t)k];
        r->excluded[(size_t)k] = s >= 0 ? 1 : 0;
        if (s >= first) r->skip[(size_t)(s - first)] = 1;  // already matched: mbTrackInView = false (:966-978)
    }
    if (S.stage2_linked) {
        S.stage2_dev = true;
        return SO_OK;
    }
    if (track_chain_on(r) && (!r->lockstep || r->fleet_chain)) {
        const int rc = so_track_stage_local_map_submit(r->matcher, r->fr[S.hcur], F.kp_mp.data(), S.stage1_dev ? 1 : 0, r->map, S.Ta, S.n_local, nullptr, first,
                                                       r->skip.data(), 1.0f, 0.8f, 0.5f, r->log_sf, r->K4, r->inv_sigma2);
        if (rc == SO_OK) {
            S.stage2_dev = true;
            return SO_OK;
        }
        if (rc != SO_RETRY_ON_HOST) return fail(r, "so_track_stage_local_map_submit");
    }
    if (so_track_search_local_map_submit(r->matcher, r->fr[S.hcur], r->excluded.data(), r->map, S.Ta, S.n_local, nullptr, first,
                                         r->skip.data(), 1.0f, 0.8f, 0.5f, r->log_sf) != SO_OK)
        return fail(r, "so_track_search_local_map_submit");
    return SO_OK;
}

int step_m1_wait(so_replay* r) {
    so_replay::Step& S = r->step;
    so_replay::FrameHost& F = r->fh[r->cur];
    const int n = F.n;
    r->k2m.resize((size_t)n);
    std::vector<uint8_t>& view = r->new_desc;  // scratch
    view.resize(std::max(view.size(), (size_t)S.n_local));
    int32_t nmm = 0;
    float kms = 0.f;
    double ms4[4];
    if (S.stage2_dev) {  // search, resolve and the second PoseOptimization came back together
        int rc = SO_OK;
        if (S.stage2_linked) so_track_stage_set_start_pose(r->matcher2, S.Ta);
        if (!stage_collect(r, r->k2m, &nmm, view.data(), S.Tb, &S.n_in, &rc, S.stage2_linked ? r->matcher2 : r->matcher)) {
            if (rc) return rc;
            S.stage2_dev = false;  // not finished on the device: the separate calls, from the search on
            if (so_track_search_local_map(r->matcher, r->fr[S.hcur], r->excluded.data(), r->map, S.Ta, S.n_local,
                                          r->cl ? r->cl->tv_local.data() : nullptr, r->cl ? 0 : S.first_slot, r->skip.data(), nullptr, 1.0f,
                                          0.8f, 0.5f, r->log_sf, view.data(), r->k2m.data(), &nmm) != SO_OK)
                return fail(r, "so_track_search_local_map");
        }
    } else if (so_track_search_local_map_wait(r->matcher, nullptr, view.data(), r->k2m.data(), &nmm) != SO_OK) {
        return fail(r, "so_track_search_local_map_wait");
    }
    so_matcher_last_kernel_ms(r->matcher, &kms);
    S.match_kernel += kms;
    so_matcher_last_stats(r->matcher, ms4);
    S.mstat[2] += ms4[0]; S.mstat[3] += ms4[1];
    S.reruns += ms4[2];
    S.nm1 = nmm;
    for (int i = 0; i < S.n_local; i++) S.n_view += view[(size_t)i];
    if (r->cl)
        for (int i = 0; i < S.n_local; i++)
            if (view[(size_t)i]) r->cl->tv_vis[(size_t)r->cl->tv_local[(size_t)i]]++;  // local points in the frustum (:990-993)
    for (int k = 0; k < n; k++)
        if (r->k2m[(size_t)k] >= 0)
            F.kp_mp[(size_t)k] = r->cl ? r->cl->tv_local[(size_t)r->k2m[(size_t)k]] : S.first_slot + r->k2m[(size_t)k];
    S.tm1 = now_ms();
    return SO_OK;
}

void pose2_apply(so_replay* r) {
    so_replay::Step& S = r->step;
    so_replay::FrameHost& F = r->fh[r->cur];
    for (size_t k = 0; k < r->idx.size(); k++)
        if (r->pose_out[k]) F.outlier[(size_t)r->idx[k]] = 1;
        else if (r->cl) r->cl->tv_found[(size_t)F.kp_mp[(size_t)r->idx[k]]]++;  // TrackLocalMap: IncreaseFound (Tracking.cc:783-786)
    S.T = from_f12(S.Tb);
    S.tp2 = now_ms();
}

std::shared_ptr<KfSnap> snapshot_keyframe(so_replay* r, int t);

// closed loop: snapshot the tracked frame and hand it to the local-mapping thread
void queue_keyframe(so_replay* r, int t) {
    std::shared_ptr<KfSnap> snap = snapshot_keyframe(r, t);  // (fills the prepared keyframe if there is one)
    // KeyFrame::ComputeBoW + the keyframe's upload: normally done already - the local-mapping thread, idle since the frame
    // began, got the keypoints then (prepare_keyframe) -; otherwise on this thread's matcher now (no search of the frame is
    // pending any more) rather than inside the job: the local-mapping thread bounds the loop
    static const bool lm_does_it = getenv("SWARMORB_CL_FEATVEC_ON_LM") != nullptr;
    if (!r->pre_kf && !lm_does_it && r->step.kf_under_pose && cl_keyframe_featvec_upload(r, r->matcher, *snap) != SO_OK)
        r->error = std::string("keyframe feature vector / upload: ") + so_last_error();
    r->pre_kf.reset();
    {   // Tracking's counters of the recently added points, as they stand now (MapPointCulling reads GetFoundRatio())
        ClosedLoop& M = *r->cl;
        snap->cnt_from = std::min(M.recent_from, (int)M.tv_vis.size());
        snap->cnt_vis.assign(M.tv_vis.begin() + snap->cnt_from, M.tv_vis.end());
        snap->cnt_found.assign(M.tv_found.begin() + snap->cnt_from, M.tv_found.end());
    }
    LmJob job;
    job.timed = r->step_timed;
    job.kf = snap;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->queue.push_back(job);
    }
    r->cv.notify_all();
    cl_keyframe_queued(r, t, snap);
    r->step.kf_queued = true;
}

// "keyframe": unmatched keypoints become map points; then the motion model
int step_keyframe(so_replay* r) {
    so_replay::Step& S = r->step;
    so_replay::FrameHost& F = r->fh[r->cur];
    const int n = F.n;
    if (r->cl) {
        // Tracking::NeedNewKeyFrame, monocular (Tracking.cc:810-892): enough frames since the last keyframe or too few
        // inliers against it - and local mapping idle; under the reference's policy a busy local mapper gets InterruptBA
        // instead (:893-903) and the keyframe waits; the deterministic schedule never asks while a job is out
        ClosedLoop& M = *r->cl;
        const int t = S.t, since = t - M.last_kf_t;
        const bool weak = (double)S.n_in < r->keyframe_ratio * (double)r->kf_inliers;
        bool need;
        if (M.policy == 0) need = !M.job_pending && (since >= M.kf_every || (since >= M.delay && weak));
        else {
            need = since >= M.kf_every || weak;
            if (need && M.job_pending) {
                bool done;
                {
                    std::lock_guard<std::mutex> lk(M.mu);
                    done = !M.outbox.empty();
                }
                if (!done) {  // mpLocalMapper->InterruptBA()
                    M.stop = 1;
                    M.interrupts++;
                }
                need = false;
            }
        }
        if (need) {
            r->kf_inliers = S.n_in > 1 ? S.n_in : 1;
            S.keyframe = 1;
            S.kf_under_pose = true;
            // the keyframe goes to local mapping NOW - this runs under the frame's last PoseOptimization kernel, whose
            // result nothing uses - not at the end of the step: its job starts ~80 us earlier
            queue_keyframe(r, t);
        }
        r->velocity = mul(S.T, rigid_inverse_general(r->T_last));
        S.tmap = now_ms();
        return SO_OK;
    }
    if ((double)S.n_in < r->keyframe_ratio * (double)r->kf_inliers || r->n_tracked % r->keyframe_every == 0) {
        std::vector<uint8_t> fresh((size_t)n, 0);
        int n_fresh = 0;
        for (int k = 0; k < n; k++)
            if (F.kp_mp[(size_t)k] < 0) {
                fresh[(size_t)k] = 1;
                n_fresh++;
            }
        if (n_fresh > 0) {
            const int f0 = add_points(r, S.T, F, fresh, n_fresh);
            if (f0 < 0) return fail(r, "so_map_write");
            int j = 0;
            for (int k = 0; k < n; k++)
                if (fresh[(size_t)k]) F.kp_mp[(size_t)k] = f0 + j++;
        }
        r->kf_inliers = S.n_in > 1 ? S.n_in : 1;
        S.keyframe = 1;
    }
    r->velocity = mul(S.T, rigid_inverse_general(r->T_last));
    S.tmap = now_ms();
    return SO_OK;
}

// The tracked frame as the local-mapping thread's new keyframe (KfSnap).
std::shared_ptr<KfSnap> snapshot_keyframe(so_replay* r, int t) {
    const so_replay::Step& S = r->step;
    const so_replay::FrameHost& F = r->fh[r->cur];
    const int n = F.n;
    if (r->cl && r->pre_kf) {  // keypoints, descriptors (and, by now, feature vector + upload) are there: bindings and pose
        std::shared_ptr<KfSnap> k = r->pre_kf;
        {   // the local-mapping thread may still be at it: wait for its queue to drain (microseconds, if at all)
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv.wait(lk, [r] { return r->queue.empty() && !r->running; });
        }
        for (int i = 0; i < n; i++) {
            const int slot = F.kp_mp[(size_t)i];
            k->mp[(size_t)i] = (slot >= 0 && !F.outlier[(size_t)i]) ? slot : -1;
        }
        to_f12(S.T, k->T);
        return k;
    }
    auto k = std::make_shared<KfSnap>();
    k->n = n;
    k->t = t;
    k->x.resize((size_t)n); k->y.resize((size_t)n); k->angle.resize((size_t)n); k->octave.resize((size_t)n); k->mp.resize((size_t)n);
    k->desc.assign(F.desc.begin(), F.desc.begin() + 32 * (size_t)n);
    if (r->cl) {  // the closed loop: the bindings Tracking leaves in the frame (inliers only); local mapping has the points
        const bool init = r->cl->n_kf == 0;
        for (int i = 0; i < n; i++) {
            k->x[(size_t)i] = F.xy_un[2 * (size_t)i];
            k->y[(size_t)i] = F.xy_un[2 * (size_t)i + 1];
            k->angle[(size_t)i] = F.kps[(size_t)i].angle;
            k->octave[(size_t)i] = F.kps[(size_t)i].octave;
            const int slot = F.kp_mp[(size_t)i];
            k->mp[(size_t)i] = (slot >= 0 && !F.outlier[(size_t)i]) ? slot : -1;
        }
        if (init) {  // keyframe 0 carries the initial map (slot i = keypoint i)
            k->mpX.assign(r->mp_X.begin(), r->mp_X.begin() + 3 * (size_t)n);
            k->mpN.assign(r->new_N.begin(), r->new_N.begin() + 3 * (size_t)n);
            k->mpMax.assign(r->new_max.begin(), r->new_max.begin() + (size_t)n);
            k->mpMin.assign(r->new_min.begin(), r->new_min.begin() + (size_t)n);
            k->mpDesc.assign(F.desc.begin(), F.desc.begin() + 32 * (size_t)n);
        }
        to_f12(S.T, k->T);
        memcpy(k->bounds, r->bounds, sizeof(k->bounds));
        return k;
    }
    k->mpX.assign(3 * (size_t)n, 0.f); k->mpN.assign(3 * (size_t)n, 0.f);
    k->mpMax.assign((size_t)n, 0.f); k->mpMin.assign((size_t)n, 0.f); k->mpDesc.assign(32 * (size_t)n, 0);
    for (int i = 0; i < n; i++) {
        k->x[(size_t)i] = F.xy_un[2 * (size_t)i];
        k->y[(size_t)i] = F.xy_un[2 * (size_t)i + 1];
        k->angle[(size_t)i] = F.kps[(size_t)i].angle;
        k->octave[(size_t)i] = F.kps[(size_t)i].octave;
        const int slot = F.kp_mp[(size_t)i];
        const bool bound = slot >= 0 && slot < S.map_size_at_begin && !F.outlier[(size_t)i];
        k->mp[(size_t)i] = bound ? slot : -1;
        if (bound) {
            memcpy(&k->mpX[3 * (size_t)i], &r->mp_X[3 * (size_t)slot], 12);
            memcpy(&k->mpN[3 * (size_t)i], &r->mp_N[3 * (size_t)slot], 12);
            k->mpMax[(size_t)i] = r->mp_max[(size_t)slot];
            k->mpMin[(size_t)i] = r->mp_min[(size_t)slot];
            memcpy(&k->mpDesc[32 * (size_t)i], &r->mp_desc[32 * (size_t)slot], 32);
        }
    }
    to_f12(S.T, k->T);
    memcpy(k->bounds, r->bounds, sizeof(k->bounds));
    return k;
}

// log, hand a window to the local-mapping thread every lba_every frames, statistics
void step_end(so_replay* r, int t, int timed) {
    so_replay::Step& S = r->step;
    so_replay::FrameHost& F = r->fh[r->cur];
    r->T_last = S.T;
    r->last_tracked = S.hcur;
    {
        float p12[12];
        to_f12(S.T, p12);
        r->poses.insert(r->poses.end(), p12, p12 + 12);
        r->n_m2.push_back(S.nm2);
        r->n_m1.push_back(S.nm1);
        r->n_inl.push_back(S.n_in);
        r->n_map.push_back((int32_t)(r->mp_X.size() / 3));
    }
    r->n_tracked++;
    const double t3 = now_ms();
    r->frame_ms.push_back((float)(t3 - S.t0));
    if (r->cl) {
        if (S.keyframe && !S.kf_queued) queue_keyframe(r, t);  // (the run's first frame: keyframe 0)
        cl_frame_end(r, t);
    } else if (t % r->lba_every == 0 && !r->window.epose.empty()) {
        LmJob job;
        job.timed = timed ? 1 : 0;
        if (!r->vocab.empty()) job.kf = snapshot_keyframe(r, t);
        std::unique_lock<std::mutex> lk(r->mu);
        r->cv.wait(lk, [r] { return r->queue.size() < 3; });  // the running window + two waiting
        r->queue.push_back(job);
        lk.unlock();
        r->cv.notify_all();
    }
    const double t4 = now_ms();
    if (timed) {
        double* st = r->stat;
        st[kSteps] += 1; st[kExtractMs] += S.t1 - S.t0; st[kM2Ms] += S.tm2 - S.t1; st[kPose1Ms] += S.tp1 - S.tm2;
        st[kM1Ms] += S.tm1 - S.tp1; st[kPose2Ms] += S.tp2 - S.tm1; st[kPose3Ms] += S.tp3 - S.tp2; st[kMapMs] += S.tmap - S.tp3;
        st[kSubmitWaitMs] += t4 - t3; st[kKp] += F.n; st[kM2] += S.nm2; st[kM1] += S.nm1; st[kInliers] += S.n_in;
        st[kMatchKernelMs] += S.match_kernel; st[kPoseKernelMs] += S.pose_kernel; st[kPoseTrials] += S.pose_trials;
        st[kPoseCalls] += S.pose_calls; st[kPosePoints] += S.pose_points; st[kPoseTimedCalls] += S.pose_timed_calls;
        st[kTimedFrames] += S.timed_kernels ? 1 : 0; st[kLocalPoints] += S.n_local; st[kInView] += S.n_view;
        st[kKeyframes] += S.keyframe; st[kMapPoints] = (double)(r->mp_X.size() / 3);
        st[kM2EnqMs] += S.mstat[0]; st[kM2WaitMs] += S.mstat[1]; st[kM1EnqMs] += S.mstat[2]; st[kM1WaitMs] += S.mstat[3];
        st[kReruns] += S.reruns; st[kWideM2] += S.wide_m2;
        float prof[SO_EXTRACTOR_N_STAGES];
        if (so_extractor_get_profile(r->ex, prof) == SO_OK)
            for (int i = 0; i < SO_EXTRACTOR_N_STAGES; i++) st[kStage0 + i] += prof[i];
    }
}

}  // namespace

extern "C" {

static int run_one_step(so_replay* r, int t, int timed) {
    r->step_timed = timed ? 1 : 0;
    r->lockstep = false;
    {
        so_replay::Step& S = r->step;
        int rc;
        static const bool submit_early = getenv("SWARMORB_REPLAY_SUBMIT_EARLY") != nullptr;  // A/B: next frame at step begin
        if ((rc = step_begin(r, t, submit_early))) return rc;
        if (!S.first) {
            int32_t inl = 0;
            if (S.stage1_dev) {  // search -> resolve -> PoseOptimization are in flight as one chain (submitted inside step_begin)
                // frame t+1 goes to the extractor while the GPU runs them
                if (!S.next_submitted && (rc = submit_frame(r, t + 1))) return rc;
                if ((rc = step_stage1_wait(r, &inl))) return rc;
            } else {
                if ((rc = step_m2_wait(r))) return rc;  // submitted inside step_begin
                // frame t+1 goes to the extractor while the GPU runs this frame's first PoseOptimization
                if ((rc = pose_single(r, S.Tp, S.Ta, &inl, [r, t, &S] { return S.next_submitted ? 0 : submit_frame(r, t + 1); }))) return rc;
            }
            pose1_apply(r);
            if ((rc = step_m1_submit(r))) return rc;
            if ((rc = step_m1_wait(r))) return rc;
            if (!S.stage2_dev && (rc = pose_single(r, S.Ta, S.Tb, &S.n_in))) return rc;
            pose2_apply(r);
            if (r->third_pose) {  // TrackReferenceKeyFrame's fallback: from the last frame's pose, result unused
                float Tl[12], Tc[12];
                to_f12(r->T_last, Tl);
                int32_t inl3 = 0;
                // the keyframe decision and the new map points only need the second result: they run under this kernel
                bool done3 = false;
                so_matcher* sm = S.stage2_linked ? r->matcher2 : r->matcher;
                if (S.stage2_dev && so_track_stage_pose_again_submit(sm, Tl) == SO_OK) {  // the same edges, still on the device
                    if ((rc = step_keyframe(r))) return rc;
                    std::vector<int32_t>& ek = r->k2l;  // scratch: the edge list comes back unchanged
                    std::vector<uint8_t>& eo = r->excluded;
                    ek.resize((size_t)r->fh[r->cur].n);
                    eo.resize((size_t)r->fh[r->cur].n);
                    int32_t ne = 0, info2[2] = {0, 0};
                    const int rc3 = so_track_stage_wait(sm, nullptr, nullptr, nullptr, &ne, ek.data(), eo.data(), Tc, &inl3, info2);
                    if (rc3 != SO_OK && rc3 != SO_RETRY_ON_HOST) return fail(r, "so_track_stage_wait");
                    done3 = true;
                    if (rc3 == SO_OK) {
                        S.pose_calls++;
                        if (S.timed_kernels) {
                            float kms = 0.f;
                            so_track_stage_last_pose_kernel_ms(sm, &kms);
                            S.pose_kernel += kms;
                            S.pose_trials += info2[1];
                            S.pose_points += ne;
                            S.pose_timed_calls++;
                        }
                    }
                }
                if (!done3 && (rc = pose_single(r, Tl, Tc, &inl3, [r] { return step_keyframe(r); }, !S.stage2_dev))) return rc;
                S.tp3 = S.tmap = now_ms();
            } else {
                S.tp3 = now_ms();
                if ((rc = step_keyframe(r))) return rc;
            }
        }
        step_end(r, t, timed);
        if (!r->error.empty()) return SO_ERR_HIP;
    }
    return SO_OK;
}

int so_replay_run(so_replay* r, int first_t, int n_steps, int timed) {
    if (!r || !r->in_flight || r->frames.empty()) return SO_ERR_INVALID_ARG;
    CallerPin pin(r);  // (the tracking thread is whoever calls; its affinity is restored on return)
    for (int t = first_t; t < first_t + n_steps; t++) {
        const int rc = run_one_step(r, t, timed);
        if (rc) return rc;
    }
    return SO_OK;
}

// The same frames the way a LIVE camera delivers them (System::TrackMonocular is synchronous,
// code/src/System.cc:128-165): frame t is handed over when its step begins - nothing is extracted ahead - and the step
// runs image upload -> extraction -> frame post-processing -> both searches -> PoseOptimization calls back to back.
// pose_ms[i]: image in -> the frame's pose out (the second PoseOptimization has returned); step_ms[i]: through the third
// PoseOptimization and the keyframe bookkeeping.  Leaves no frame in flight (so_replay_prime before the next
// so_replay_run).  Not counted in the statistics.
int so_replay_run_live(so_replay* r, int first_t, int n_steps, float* pose_ms, float* step_ms) {
    if (!r || r->frames.empty() || first_t < 0 || first_t + n_steps > (int)r->frames.size()) return SO_ERR_INVALID_ARG;
    r->live = true;
    CallerPin pin(r);
    int rc = SO_OK;
    for (int t = first_t; t < first_t + n_steps && rc == SO_OK; t++) {
        const double T0 = now_ms();
        if (!r->in_flight) rc = submit_frame(r, t);  // the image arrives now
        if (rc == SO_OK) rc = run_one_step(r, t, 0);
        const so_replay::Step& S = r->step;
        if (pose_ms) pose_ms[t - first_t] = (float)((S.first ? now_ms() : S.tp2) - T0);
        if (step_ms) step_ms[t - first_t] = (float)(now_ms() - T0);
    }
    r->live = false;
    return rc;
}

// Several agents on one GPU, driven in lockstep by the calling thread: per stage the searches of all agents are
// submitted before any is waited for (their kernels overlap on the agents' own streams: create the handles after
// so_runtime_private_streams(1)), and the PoseOptimization problems of all agents go out as ONE launch (a workgroup per
// agent).  Every agent ends up with exactly the results a solo so_replay_run gives it.
int so_replay_drain(so_replay* r);

int so_fleet_run(so_replay** agents, int n_agents, int first_t, int n_steps, int timed) {
    if (!agents || n_agents < 1) return SO_ERR_INVALID_ARG;
    for (int a = 0; a < n_agents; a++)
        if (!agents[a] || !agents[a]->in_flight || agents[a]->frames.empty()) return SO_ERR_INVALID_ARG;
    const size_t A = (size_t)n_agents;
    CallerPin pin(agents[0]);
    std::vector<so_pose_problem> probs(A);
    std::vector<int32_t> inl(A), info(2 * A);
    std::vector<int> live;
    so_ba* batch_opt = agents[0]->tracker_opt;
    auto pose_batch = [&](int which) -> int {  // which: 0 after the motion-model search, 1 after the local map, 2 the third call
        live.clear();
        for (int a = 0; a < n_agents; a++)
            if (!agents[a]->step.first) live.push_back(a);
        if (live.empty()) return SO_OK;
        for (size_t k = 0; k < live.size(); k++) {
            so_replay* r = agents[live[k]];
            so_replay::Step& S = r->step;
            if (which == 2) to_f12(r->T_last, S.Tl);
            const float* Tin = which == 0 ? S.Tp : (which == 1 ? S.Ta : S.Tl);
            float* Tout = which == 0 ? S.Ta : (which == 1 ? S.Tb : S.Tc);
            pose_gather(r, Tin, Tout, which == 1 ? &S.n_in : &inl[(size_t)live[k]], &info[2 * (size_t)live[k]], &probs[k]);
        }
        if (so_pose_optimization_batch(batch_opt, (int32_t)live.size(), probs.data()) != SO_OK)
            return fail(agents[0], "so_pose_optimization_batch");
        float ms = 0.f;
        so_pose_optimization_last_kernel_ms(batch_opt, &ms);
        for (size_t k = 0; k < live.size(); k++) pose_account(agents[live[k]], probs[k], ms / (float)live.size());
        return SO_OK;
    };
    // One extraction chain for the whole fleet (so_extractor_group: every kernel once, the agent as a grid dimension)
    // when the agents' frames are device-visible and of one size; SWARMORB_FLEET_NO_GROUP=1 keeps a chain per agent.
    static const bool no_group = getenv("SWARMORB_FLEET_NO_GROUP") != nullptr;
    so_replay* lead = agents[0];
    bool grouped = !no_group && n_agents > 1 && n_agents <= SO_EXTRACTOR_GROUP_MAX;
    for (int a = 0; a < n_agents && grouped; a++)
        grouped = agents[a]->width == lead->width && agents[a]->height == lead->height && !agents[a]->live;  // (every agent's next frame
                                                                                                          //  goes into ITS next handle: the rotation need not agree)
    if (grouped) {
        std::vector<so_extractor*> members(A);
        for (size_t a = 0; a < A; a++) members[a] = agents[a]->ex;
        if (members != lead->fleet_members) {
            so_extractor_group_destroy(lead->fleet_group);
            lead->fleet_group = nullptr;
            lead->fleet_members.clear();
            if (so_extractor_group_create(members.data(), n_agents, &lead->fleet_group) == SO_OK) lead->fleet_members = members;
        }
        grouped = lead->fleet_group != nullptr;
    }
    std::vector<so_dframe*> gframes(A);
    std::vector<const uint8_t*> gimages(A);
    // The tracking stages of all agents as ONE chain of launches per stage (so_track_group: search with the agent as
    // blockIdx.y, one resolve and one PoseOptimization workgroup per agent) when the stages are chained on the device and the
    // agents' matchers share a stream (handles created on one thread without so_runtime_private_streams); otherwise the
    // separate calls with the PoseOptimization calls batched (round 3's lockstep).
    static const bool no_track_group = getenv("SWARMORB_FLEET_NO_TRACK_GROUP") != nullptr;
    bool chained = !no_track_group && track_chain_on(lead);
    for (int a = 0; a < n_agents && chained; a++)
        chained = track_chain_on(agents[a]) && so_matcher_stream_id(agents[a]->matcher) == so_matcher_stream_id(lead->matcher);
    if (chained && !lead->fleet_track_group && so_track_group_create(lead->device, &lead->fleet_track_group) != SO_OK) chained = false;
    struct LeaveGroup {  // the agents are solo again when the call returns (whatever way)
        so_replay** ag; int n; bool on;
        ~LeaveGroup() {
            for (int a = 0; a < n; a++) {
                ag[a]->fleet_chain = false;
                if (on) so_matcher_set_track_group(ag[a]->matcher, nullptr);
            }
        }
    } leave{agents, n_agents, chained};
    for (int a = 0; a < n_agents && chained; a++) {
        if (so_matcher_set_track_group(agents[a]->matcher, lead->fleet_track_group) != SO_OK) return fail(agents[a], "so_matcher_set_track_group");
        agents[a]->fleet_chain = true;
    }
    // ... and their local bundle adjustments: the agents' local-mapping threads reach so_bundle_adjust at about the same time
    // (keyframes fall on the same frames), their LM chains go out merged (so_ba_group).  Joined once, on the fleet's first run.
    // (from five agents on: with four the merged rounds - a window waits for the others, the round is as slow as its slowest member -
    //  cost more than four separate chains do, 4.4 against 4.9 k frames/s; at eight it is a tie, beyond that the group wins: NOTES G.8.
    //  SWARMORB_FLEET_BA_GROUP_MIN=n moves the threshold.)
    static const bool no_ba_group = getenv("SWARMORB_FLEET_NO_BA_GROUP") != nullptr;
    const int ba_group_min = getenv("SWARMORB_FLEET_BA_GROUP_MIN") ? atoi(getenv("SWARMORB_FLEET_BA_GROUP_MIN")) : 5;  // (per call: tests switch it)
    if (chained && !no_ba_group && n_agents >= std::max(2, ba_group_min) && !lead->fleet_ba_group) {
        const double window_us = getenv("SWARMORB_FLEET_BA_WINDOW_US") ? atof(getenv("SWARMORB_FLEET_BA_WINDOW_US")) : 600.0;
        if (so_ba_group_create(lead->device, window_us, &lead->fleet_ba_group) == SO_OK)
            for (int a = 0; a < n_agents; a++) so_ba_set_group(agents[a]->mapper_opt, lead->fleet_ba_group);
    }
    // The agents' local-mapping matchers (searches, triangulation, UpdateNormalAndDepth of all jobs) on ONE stream of their own,
    // beside the tracking stream: created on the caller's thread they would otherwise all share the tracking stream itself.
    static const bool no_lm_stream = getenv("SWARMORB_FLEET_NO_LM_STREAM") != nullptr;
    if (chained && !no_lm_stream && n_agents > 1 && !lead->fleet_lm_stream_set) {
        lead->fleet_lm_stream_set = true;
        for (int a = 0; a < n_agents; a++) so_replay_drain(agents[a]);  // (the local-mapping threads are idle: their matchers may move)
        if (so_matcher_private_stream(lead->mapper_matcher) == SO_OK) {
            for (int a = 1; a < n_agents; a++) so_matcher_share_stream(agents[a]->mapper_matcher, lead->mapper_matcher);
            // ... and the map tables' (synchronous, rare) writes with them: a stream per table is eight more busy streams
            static const bool own_map_streams = getenv("SWARMORB_FLEET_MAP_STREAMS") != nullptr;
            for (int a = 0; a < n_agents && !own_map_streams; a++) so_map_share_stream(agents[a]->map, lead->mapper_matcher);
        }
    }
    so_track_group* tg = chained ? lead->fleet_track_group : nullptr;
    auto group_launch = [&]() -> int {
        if (so_track_group_pending(tg) > 0 && so_track_group_launch(tg) != SO_OK) return fail(lead, "so_track_group_launch");
        return SO_OK;
    };
    auto group_pose_ms = [&](int n_members) -> float {  // the group's PoseOptimization kernel, dealt to its members
        float sm = 0.f, pm = 0.f;
        if (n_members > 0) so_track_group_last_kernel_ms(tg, &sm, &pm);  // (0 unless a member of the launch asked for events: the asking agent did)
        return n_members > 0 ? pm / (float)n_members : 0.f;
    };
    // Elastic ticks: the agents share the launches, not a clock.  A tick takes every agent whose next frame can start now; an
    // agent whose frame has to wait for its local-mapping job (deterministic schedule: the packet of keyframe k is applied at
    // frame k + delay) sits the tick out - its neighbours do not wait with it, it rejoins when the packet is there, and every
    // agent still tracks exactly its own sequence of frames (bit-equal to its solo run, tests/test_closedloop_gpu.py).  Each
    // agent tracks n_steps frames per call; SWARMORB_FLEET_RIGID=1: every tick takes all agents (waits inside step_begin).
    const bool rigid_env = getenv("SWARMORB_FLEET_RIGID") != nullptr;  // (read per call: tests switch it)
    bool any_cl = false;
    for (int a = 0; a < n_agents; a++) any_cl = any_cl || agents[a]->cl != nullptr;
    const bool elastic = chained && !rigid_env && any_cl && n_agents > 1;
    std::vector<int> done(A, 0);
    std::vector<char> act(A, 1);
    int n_ticks = 0, n_slots = 0;
    while (chained) {
        int rc, n_act = 0, n_left = 0;
        for (int a = 0; a < n_agents; a++) n_left += done[(size_t)a] < n_steps ? 1 : 0;
        if (n_left == 0) break;
        for (int spin = 0; n_act == 0; spin++) {
            for (int a = 0; a < n_agents; a++) {
                act[(size_t)a] = done[(size_t)a] < n_steps &&
                                 (!elastic || cl_frame_ready(agents[a], first_t + agents[a]->fleet_offset + done[(size_t)a]));
                n_act += act[(size_t)a] ? 1 : 0;
            }
            if (n_act == 0)  // every agent left is waiting for its job: the first packet to arrive starts the next tick
                for (int i = 0; i < 32; i++) __builtin_ia32_pause();
        }
        n_ticks++;
        n_slots += n_act;
        auto tf = [&](int a) { return first_t + agents[a]->fleet_offset + done[(size_t)a]; };  // the frame agent a tracks in this tick
        for (int a = 0; a < n_agents; a++) {
            if (!act[(size_t)a]) continue;
            agents[a]->lockstep = true;
            agents[a]->step_timed = timed ? 1 : 0;
            if ((rc = step_begin(agents[a], tf(a), !grouped))) return rc;  // (records the agent's last-frame stage)
        }
        if ((rc = group_launch())) return rc;
        if (grouped) {  // the next frames of the tick's agents as one extraction chain, under the stage that was just launched
            bool all = true;
            for (int a = 0; a < n_agents; a++) all = all && (!act[(size_t)a] || !agents[a]->step.next_submitted);
            if (all) {
                for (size_t a = 0; a < A; a++) {
                    so_replay* r = agents[a];
                    gframes[a] = r->fr[(r->submitted + 1) % 3];
                    gimages[a] = act[a] ? r->frames[(size_t)(tf((int)a) + 1) % r->frames.size()] : nullptr;  // (null: sits out)
                }
                if (so_dframe_group_submit(lead->fleet_group, gframes.data(), gimages.data(), lead->width, lead->height, lead->width) == SO_OK) {
                    for (size_t a = 0; a < A; a++) {
                        if (!act[a]) continue;
                        agents[a]->submitted = (agents[a]->submitted + 1) % 3;
                        agents[a]->in_flight = true;
                        agents[a]->step.next_submitted = true;
                    }
                } else {
                    grouped = false;
                }
            }
        }
        for (int a = 0; a < n_agents; a++)
            if (act[(size_t)a] && !agents[a]->step.next_submitted) {
                if ((rc = submit_frame(agents[a], tf(a) + 1))) return rc;
                agents[a]->step.next_submitted = true;
            }
        // stage 1 back, stage 2 recorded - agent by agent: the host part of one agent runs under the kernels of the others
        int n_dev = 0;
        for (int a = 0; a < n_agents; a++) n_dev += (act[(size_t)a] && !agents[a]->step.first && agents[a]->step.stage1_dev) ? 1 : 0;
        for (int a = 0; a < n_agents; a++) {
            so_replay* r = agents[a];
            so_replay::Step& S = r->step;
            if (!act[(size_t)a] || S.first) continue;
            int32_t inl1 = 0;
            const bool dev = S.stage1_dev;
            if (dev) {
                if ((rc = step_stage1_wait(r, &inl1))) return rc;  // (falls back onto the separate calls by itself)
                if (S.timed_kernels && S.stage1_dev) S.pose_kernel += group_pose_ms(n_dev);
            } else {
                if ((rc = step_m2_wait(r))) return rc;
                if ((rc = pose_single(r, S.Tp, S.Ta, &inl1))) return rc;
            }
            pose1_apply(r);
            if ((rc = step_m1_submit(r))) return rc;
        }
        if ((rc = group_launch())) return rc;
        n_dev = 0;
        for (int a = 0; a < n_agents; a++) n_dev += (act[(size_t)a] && !agents[a]->step.first && agents[a]->step.stage2_dev) ? 1 : 0;
        int n_again = 0;
        for (int a = 0; a < n_agents; a++) {
            so_replay* r = agents[a];
            so_replay::Step& S = r->step;
            if (!act[(size_t)a] || S.first) continue;
            const bool dev = S.stage2_dev;
            if ((rc = step_m1_wait(r))) return rc;
            if (dev && S.stage2_dev && S.timed_kernels) S.pose_kernel += group_pose_ms(n_dev);
            if (!S.stage2_dev && (rc = pose_single(r, S.Ta, S.Tb, &S.n_in))) return rc;
            pose2_apply(r);
            S.third_dev = false;
            if (r->third_pose) {
                to_f12(r->T_last, S.Tl);
                if (S.stage2_dev && so_track_stage_pose_again_submit(r->matcher, S.Tl) == SO_OK) {
                    S.third_dev = true;
                    n_again++;
                }
            }
        }
        if ((rc = group_launch())) return rc;
        for (int a = 0; a < n_agents; a++) {
            so_replay* r = agents[a];
            so_replay::Step& S = r->step;
            if (!act[(size_t)a]) continue;
            if (!S.first) {
                if (r->third_pose && S.third_dev) {
                    if ((rc = step_keyframe(r))) return rc;  // (under the third PoseOptimization, whose result nothing uses)
                    std::vector<int32_t>& ek = r->k2l;
                    std::vector<uint8_t>& eo = r->excluded;
                    ek.resize((size_t)r->fh[r->cur].n);
                    eo.resize((size_t)r->fh[r->cur].n);
                    int32_t ne = 0, inl3 = 0, info2[2] = {0, 0};
                    const int rc3 = so_track_stage_wait(r->matcher, nullptr, nullptr, nullptr, &ne, ek.data(), eo.data(), S.Tc, &inl3, info2);
                    if (rc3 != SO_OK && rc3 != SO_RETRY_ON_HOST) return fail(r, "so_track_stage_wait");
                    if (rc3 == SO_OK) {
                        S.pose_calls++;
                        if (S.timed_kernels) {
                            S.pose_kernel += group_pose_ms(n_again);
                            S.pose_trials += info2[1];
                            S.pose_points += ne;
                            S.pose_timed_calls++;
                        }
                    }
                    S.tp3 = S.tmap = now_ms();
                } else if (r->third_pose) {
                    int32_t inl3 = 0;
                    if ((rc = pose_single(r, S.Tl, S.Tc, &inl3, [r] { return step_keyframe(r); }, false))) return rc;
                    S.tp3 = S.tmap = now_ms();
                } else {
                    S.tp3 = now_ms();
                    if ((rc = step_keyframe(r))) return rc;
                }
            }
            step_end(r, tf(a), timed);
            if (!r->error.empty()) return SO_ERR_HIP;
        }
        static const bool tick_trace = getenv("SWARMORB_CL_TRACE") != nullptr;
        if (tick_trace && timed) {
            int first_act = 0;
            while (!act[(size_t)first_act]) first_act++;
            fprintf(stderr, "[tick] %d agents %d begin %.3f end %.3f\n", n_ticks, n_act, agents[first_act]->step.t0, now_ms());
        }
        for (int a = 0; a < n_agents; a++) done[(size_t)a] += act[(size_t)a] ? 1 : 0;
    }
    if (chained) {
        lead->fleet_ticks += n_ticks;
        lead->fleet_tick_slots += n_slots;
    }
    if (chained) return SO_OK;
    for (int t = first_t; t < first_t + n_steps; t++) {
        int rc;
        for (int a = 0; a < n_agents; a++) {
            agents[a]->lockstep = true;
            agents[a]->step_timed = timed ? 1 : 0;
            if ((rc = step_begin(agents[a], t + agents[a]->fleet_offset, !grouped))) return rc;
        }
        if (grouped) {  // the agents whose next frame is not out yet (all of them, except on a run's first frame)
            bool all = true;
            for (int a = 0; a < n_agents; a++) all = all && !agents[a]->step.next_submitted;
            if (all) {
                for (size_t a = 0; a < A; a++) {
                    so_replay* r = agents[a];
                    gframes[a] = r->fr[(r->submitted + 1) % 3];
                    gimages[a] = r->frames[(size_t)(t + r->fleet_offset + 1) % r->frames.size()];
                }
                if (so_dframe_group_submit(lead->fleet_group, gframes.data(), gimages.data(), lead->width, lead->height, lead->width) == SO_OK) {
                    for (size_t a = 0; a < A; a++) {
                        agents[a]->submitted = (agents[a]->submitted + 1) % 3;
                        agents[a]->in_flight = true;
                        agents[a]->step.next_submitted = true;
                    }
                } else {
                    grouped = false;  // (pageable frames: the group wants device-visible images) - a chain per agent from here on
                }
            }
            for (int a = 0; a < n_agents; a++)
                if (!agents[a]->step.next_submitted) {
                    if ((rc = submit_frame(agents[a], t + agents[a]->fleet_offset + 1))) return rc;
                    agents[a]->step.next_submitted = true;
                }
        }
        for (int a = 0; a < n_agents; a++)  // (each agent's motion-model search went out inside its step_begin)
            if (!agents[a]->step.first && (rc = step_m2_wait(agents[a]))) return rc;
        if ((rc = pose_batch(0))) return rc;
        for (int a = 0; a < n_agents; a++)
            if (!agents[a]->step.first) pose1_apply(agents[a]);
        for (int a = 0; a < n_agents; a++)
            if (!agents[a]->step.first && (rc = step_m1_submit(agents[a]))) return rc;
        for (int a = 0; a < n_agents; a++)
            if (!agents[a]->step.first && (rc = step_m1_wait(agents[a]))) return rc;
        if ((rc = pose_batch(1))) return rc;
        for (int a = 0; a < n_agents; a++)
            if (!agents[a]->step.first) pose2_apply(agents[a]);
        if (agents[0]->third_pose && (rc = pose_batch(2))) return rc;
        for (int a = 0; a < n_agents; a++) {
            so_replay* r = agents[a];
            if (!r->step.first) {
                r->step.tp3 = now_ms();
                if ((rc = step_keyframe(r))) return rc;
            }
            step_end(r, t + r->fleet_offset, timed);
            if (!r->error.empty()) return SO_ERR_HIP;
        }
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
    if (r->cl && r->cl->n_handover > 0 && getenv("SWARMORB_CL_TRACE"))
        fprintf(stderr, "[cycle] packet arrival -> next keyframe handed over: %.3f ms mean over %d keyframes; tracking thread waited %.3f ms per keyframe\n",
                r->cl->handover_ms / r->cl->n_handover, r->cl->n_handover, r->cl->wait_ms / r->cl->n_handover);
    if (r->in_flight) {
        int n = 0;
        so_replay::FrameHost& F = r->fh[r->cur ^ 1];  // scratch: the last frame's host arrays are not needed any more
        if (so_dframe_collect(r->fr[r->submitted], F.kps.data(), nullptr, F.desc.data(), r->cap, &n, nullptr) != SO_OK)
            return fail(r, "so_dframe_collect");
        r->in_flight = false;
    }
    return SO_OK;
}

int so_replay_stats(so_replay* r, double* out48) {
    if (!r || !out48) return SO_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(r->mu);
    memcpy(out48, r->stat, sizeof(r->stat));
    return SO_OK;
}

// Per-frame log of everything tracked so far: poses (12 floats per frame), M2 / M1 match counts, inliers, map size.
// wall time of every tracked frame of the log (ms, step_begin to step_end), for looking at the spread
int so_replay_frame_ms(so_replay* r, float* out, int cap) {
    if (!r || !out) return SO_ERR_INVALID_ARG;
    const int n = std::min(cap, (int)r->frame_ms.size());
    memcpy(out, r->frame_ms.data(), sizeof(float) * (size_t)n);
    return n;
}
int so_replay_log_size(so_replay* r) { return r ? (int)r->n_m2.size() : 0; }
int so_replay_log(so_replay* r, float* poses12, int32_t* n_m2, int32_t* n_m1, int32_t* n_inliers, int32_t* n_map) {
    if (!r) return SO_ERR_INVALID_ARG;
    const size_t n = r->n_m2.size();
    if (poses12) memcpy(poses12, r->poses.data(), sizeof(float) * 12 * n);
    if (n_m2) memcpy(n_m2, r->n_m2.data(), 4 * n);
    if (n_m1) memcpy(n_m1, r->n_m1.data(), 4 * n);
    if (n_inliers) memcpy(n_inliers, r->n_inl.data(), 4 * n);
    if (n_map) memcpy(n_map, r->n_map.data(), 4 * n);
    return SO_OK;
}

// descriptors of the last tracked frame (for the cross-agent exchange tick) and the handles (candidates, exchange)
int so_replay_last_frame(so_replay* r, const uint8_t** desc, int* n) {
    if (!r || !desc || !n) return SO_ERR_INVALID_ARG;
    *desc = r->fh[r->cur].desc.data();
    *n = r->fh[r->cur].n;
    return SO_OK;
}
// keypoint -> map slot bindings of the frame tracked last (-1: none; what mvpMapPoints holds when the frame becomes a
// keyframe, code/src/KeyFrame.cc:47) and its pose: the per-keypoint block of the keyframe record the exchange sends
int so_replay_last_bindings(so_replay* r, const int32_t** kp_mp, int* n, float* Tcw12) {
    if (!r || !kp_mp || !n) return SO_ERR_INVALID_ARG;
    const so_replay::FrameHost& F = r->fh[r->cur];
    *kp_mp = F.kp_mp.data();
    *n = std::min(F.n, (int)F.kp_mp.size());
    if (Tcw12 && !r->poses.empty()) memcpy(Tcw12, r->poses.data() + r->poses.size() - 12, sizeof(float) * 12);
    return SO_OK;
}
// the device-resident frame tracked last (the exchange fills its slot from it without a host hop)
so_dframe* so_replay_last_dframe(so_replay* r) {
    if (!r || r->n_tracked == 0 || r->last_tracked < 0) return nullptr;
    return r->fr[r->last_tracked];
}
so_extractor* so_replay_extractor(so_replay* r) { return r ? r->ex : nullptr; }
so_matcher* so_replay_matcher(so_replay* r) { return r ? r->matcher : nullptr; }

}  // extern "C"
