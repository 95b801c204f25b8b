"""A tracking stage as ONE chain of launches (so_track_stage_*: search -> the order-dependent resolve on the device ->
PoseOptimization over edges read in place) against the separate calls it replaces and against the CPU oracle: the matches
are the oracle's sequential result (bit-exact: the parallel rounds reproduce ORBmatcher.cc:83-85 / 1294-1296 exactly, also
where many map points compete for the same keypoints), the pose is so_pose_optimization's on the same bindings to the bit."""
import numpy as np
import pytest

from swarmmap_amd import dframe as dfm
from swarmmap_amd import synth
from swarmmap_amd.matcher import FrameView
from test_dframe_gpu import LOG_SF, S, _frame_and_view, _make_map, _pose  # noqa: F401 (S is a fixture)

pytestmark = pytest.mark.gpu

K4 = np.asarray(synth.EUROC_K, np.float32)
INV_SIGMA2 = (1.0 / (synth.SCALE_FACTORS.astype(np.float32) ** 2)).astype(np.float32)


def _host_pose(S, Tcw, kp_slot, xy_un, octave, Xw):
    """Optimizer::PoseOptimization the way the host chain gathers it: the keypoints with a map point, ascending."""
    idx = np.nonzero(kp_slot >= 0)[0]
    opt = S.Optimizer()
    n_in, T, outl, info = opt.PoseOptimization(Tcw, K4, Xw[kp_slot[idx]], xy_un[idx], INV_SIGMA2[octave[idx]])
    opt.close()
    return idx, n_in, T.reshape(3, 4), outl, info


@pytest.mark.parametrize("seed,dist", [(41, synth.EUROC_DIST), (43, (0, 0, 0, 0))])
def test_last_frame_stage_equals_search_plus_pose(S, oracle, seed, dist):
    rng = np.random.default_rng(seed)
    ex, last, lk, lxy, ld, _ = _frame_and_view(S, oracle, seed, dist=dist)
    Tl = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, lxy, lk, ld, synth.EUROC_K, Tl)
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    cur = S.DeviceFrame(ex, synth.EUROC_K, dist)
    ck, cxy, cd = [a.copy() for a in cur(synth.make_canvas(seed, 752, 480))]
    F = FrameView(cxy[:, 0], cxy[:, 1], ck["octave"], ck["angle"], cd, cur.bounds, ex.GetScaleFactors())
    n_last = len(lk)
    slot = np.where(rng.random(n_last) < 0.75, np.arange(n_last), -1).astype(np.int32)
    Tc = Tl.copy()
    Tc[3] += 0.004; Tc[7] -= 0.003
    cam = oracle.camera(synth.EUROC_K, dist)
    valid, u, v = oracle.project_last_frame(cam, cur.bounds, Tc, Xw[np.maximum(slot, 0)], slot >= 0)
    lastd = dict(valid=valid, u=u, v=v, octave=lk["octave"], angle=lk["angle"], desc=md[np.maximum(slot, 0)],
                 has_obs=np.ones(n_last, np.uint8))
    for th, ori in ((15.0, True), (30.0, False)):
        m = S.ORBmatcher(0.9, ori)
        r = dfm.track_stage_last_frame(m, cur, last, dmap, Tc, slot, th, K4, INV_SIGMA2)
        assert r is not None
        onm, ok2l = oracle.search_by_projection_lastframe(F, lastd, th, ori)
        assert r["nmatches"] == onm > 200 and np.array_equal(r["kp_to_q"], ok2l)
        # the plain call on the same handle afterwards: same matches, and the pose over them
        nm, k2l = dfm.search_last_frame(m, cur, last, dmap, Tc, slot, th)
        assert nm == onm and np.array_equal(k2l, ok2l)
        kp_slot = np.where(k2l >= 0, slot[np.maximum(k2l, 0)], -1)
        idx, n_in, T, outl, info = _host_pose(S, Tc, kp_slot, cxy, ck["octave"], Xw)
        assert r["n_edges"] == len(idx) and np.array_equal(r["edge_kp"], idx)
        assert np.array_equal(r["Tcw"], T) and np.array_equal(r["edge_outlier"], outl) and r["n_inliers"] == n_in
        assert r["iterations"] == info["iterations"] and r["trials"] == info["lm_trials"]
        # the same edges from another start pose
        T2 = Tc.copy(); T2[3] -= 0.01
        again = dfm.track_stage_pose_again(m, cur, T2)
        _, n_in2, Tb, outl2, _ = _host_pose(S, T2, kp_slot, cxy, ck["octave"], Xw)
        assert again is not None and np.array_equal(again["Tcw"], Tb) and np.array_equal(again["edge_outlier"], outl2)
        assert again["n_inliers"] == n_in2 and np.array_equal(again["edge_kp"], idx)
        # the local-map stage behind it, with the bindings the last-frame stage left ON THE DEVICE (its matches minus its pose's
        # outliers: Tracking.cc:745-760) against the same stage fed the host copy of those bindings
        r = dfm.track_stage_last_frame(m, cur, last, dmap, Tc, slot, th, K4, INV_SIGMA2)
        bound = kp_slot.copy()
        bound[r["edge_kp"][r["edge_outlier"] != 0]] = -1
        skip = np.zeros(len(Xw), np.uint8); skip[bound[bound >= 0]] = 1
        a_dev = dfm.track_stage_local_map(m, cur, bound, dmap, r["Tcw"], len(Xw), 1.0, 0.5, LOG_SF, K4, INV_SIGMA2, skip=skip,
                                          kp_slot_is_last_stage=True)
        a_host = dfm.track_stage_local_map(m, cur, bound, dmap, r["Tcw"], len(Xw), 1.0, 0.5, LOG_SF, K4, INV_SIGMA2, skip=skip)
        assert a_dev is not None and a_host is not None and a_dev["n_edges"] > (bound >= 0).sum() > 100
        for k in ("kp_to_q", "edge_kp", "edge_outlier", "Tcw", "in_view"):
            assert np.array_equal(a_dev[k], a_host[k]), k
        assert a_dev["nmatches"] == a_host["nmatches"] and a_dev["n_inliers"] == a_host["n_inliers"]
        # The host changed the bindings after the last-frame stage (a bad point dropped by SearchLocalPoints' isBad test, a
        # fall-back onto TrackReferenceKeyFrame, a wider re-search: host/glue/Tracking_glue.cc): the device copy is stale.  With
        # so_track_stage_invalidate the flag no longer matters - the stage uploads the host bindings - and the result is the one
        # of an honest flag 0; WITHOUT it a flag 1 would silently optimise over the stale edges (the advisor's round-5 finding).
        r = dfm.track_stage_last_frame(m, cur, last, dmap, Tc, slot, th, K4, INV_SIGMA2)
        changed = bound.copy()
        changed[np.nonzero(changed >= 0)[0][::3]] = -1
        skip2 = np.zeros(len(Xw), np.uint8); skip2[changed[changed >= 0]] = 1
        m._lib.so_track_stage_invalidate.argtypes = [__import__("ctypes").c_void_p]
        assert m._lib.so_track_stage_invalidate(m._h) == 0
        a_inv = dfm.track_stage_local_map(m, cur, changed, dmap, r["Tcw"], len(Xw), 1.0, 0.5, LOG_SF, K4, INV_SIGMA2, skip=skip2,
                                          kp_slot_is_last_stage=True)
        r = dfm.track_stage_last_frame(m, cur, last, dmap, Tc, slot, th, K4, INV_SIGMA2)
        a_ref = dfm.track_stage_local_map(m, cur, changed, dmap, r["Tcw"], len(Xw), 1.0, 0.5, LOG_SF, K4, INV_SIGMA2, skip=skip2)
        for k in ("kp_to_q", "edge_kp", "edge_outlier", "Tcw", "in_view"):
            assert np.array_equal(a_inv[k], a_ref[k]), k
        assert not np.array_equal(a_ref["edge_kp"], a_host["edge_kp"])  # (the changed bindings are a different pose problem)
        m.close()
    dmap.close(); cur.close(); last.close(); ex.close()


@pytest.mark.parametrize("seed,dist,th", [(51, synth.EUROC_DIST, 1.0), (52, (0, 0, 0, 0), 3.0)])
def test_local_map_stage_with_competing_points_equals_search_plus_pose(S, oracle, seed, dist, th):
    """800 duplicated map points compete for the keypoints of their originals, a third of the keypoints is bound on entry."""
    rng = np.random.default_rng(seed)
    ex, cur, ck, cxy, cd, F = _frame_and_view(S, oracle, seed, dist=dist)
    Tc = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, cxy, ck, cd, synth.EUROC_K, Tc)
    dup = rng.integers(0, len(ck), 800)
    Xw = np.concatenate([Xw, Xw[dup] + rng.normal(0, 0.004, (800, 3)).astype(np.float32)])
    normal = np.concatenate([normal, normal[dup]]); mx = np.concatenate([mx, mx[dup]]); mn = np.concatenate([mn, mn[dup]])
    md = np.concatenate([md, synth.flip_bits(rng, md[dup], 0.05)])
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    n_map = len(Xw)
    # bindings on entry: a third of the keypoints hold "their" map point (slot = keypoint index in _make_map)
    kp_slot = np.where(rng.random(len(ck)) < 0.35, np.arange(len(ck)), -1).astype(np.int32)
    excluded = (kp_slot >= 0).astype(np.uint8)
    F.excluded = excluded
    cam = oracle.camera(synth.EUROC_K, dist)
    for local in (None, rng.permutation(n_map)[: n_map * 2 // 3].astype(np.int32)):
        idx = np.arange(n_map) if local is None else local
        bound = np.zeros(n_map, np.uint8); bound[kp_slot[kp_slot >= 0]] = 1
        skip = ((rng.random(len(idx)) < 0.1) | (bound[idx] == 1)).astype(np.uint8)
        fr = oracle.is_in_frustum(cam, cur.bounds, Tc, Xw[idx], normal[idx], mx[idx], mn[idx], 0.5, LOG_SF, 8)
        in_view = fr["in_view"] & (1 - skip)
        mps = dict(in_view=in_view, proj_x=fr["proj_x"], proj_y=fr["proj_y"], view_cos=fr["view_cos"],
                   pred_level=fr["pred_level"], desc=md[idx], has_obs=np.ones(len(idx), np.uint8))
        m = S.ORBmatcher(0.8, True)
        r = dfm.track_stage_local_map(m, cur, kp_slot, dmap, Tc, len(idx), th, 0.5, LOG_SF, K4, INV_SIGMA2, local_slot=local, skip=skip)
        assert r is not None
        onm, ok2m = oracle.search_by_projection_mappoints(F, mps, th, 0.8)
        assert np.array_equal(r["in_view"], in_view)
        assert r["nmatches"] == onm > 100 and np.array_equal(r["kp_to_q"], ok2m)
        after = np.where(kp_slot >= 0, kp_slot, np.where(ok2m >= 0, idx[np.maximum(ok2m, 0)], -1)).astype(np.int32)
        eidx, n_in, T, outl, info = _host_pose(S, Tc, after, cxy, ck["octave"], Xw)
        assert np.array_equal(r["edge_kp"], eidx) and r["n_edges"] > 400
        assert np.array_equal(r["Tcw"], T) and np.array_equal(r["edge_outlier"], outl) and r["n_inliers"] == n_in
        m.close()
    dmap.close(); cur.close(); ex.close()


def test_stage_that_runs_out_of_list_entries_hands_the_call_back(S, oracle):
    """Forty copies of each of fifty map points land in the same windows: the later copies find all eight entries of their
    K-list taken while the window holds more candidates - the stage answers SO_RETRY_ON_HOST, the separate calls (whose host
    resolve re-runs such a query exactly) still give the oracle's matches, and the handle is usable afterwards."""
    rng = np.random.default_rng(77)
    ex, cur, ck, cxy, cd, F = _frame_and_view(S, oracle, 77)
    Tc = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, cxy, ck, cd, synth.EUROC_K, Tc, n_extra=0)
    base = rng.permutation(len(ck))[:50]
    rep = np.repeat(base, 40)
    Xw = np.concatenate([Xw[rep] + rng.normal(0, 0.002, (len(rep), 3)).astype(np.float32)])
    normal, mx, mn = normal[rep], mx[rep], mn[rep]
    md = synth.flip_bits(rng, md[rep], 0.02)
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    kp_slot = np.full(len(ck), -1, np.int32)
    m = S.ORBmatcher(0.8, True)
    r = dfm.track_stage_local_map(m, cur, kp_slot, dmap, Tc, len(Xw), 6.0, 0.5, LOG_SF, K4, INV_SIGMA2)
    cam = oracle.camera(synth.EUROC_K, synth.EUROC_DIST)
    fr = oracle.is_in_frustum(cam, cur.bounds, Tc, Xw, normal, mx, mn, 0.5, LOG_SF, 8)
    mps = dict(in_view=fr["in_view"], proj_x=fr["proj_x"], proj_y=fr["proj_y"], view_cos=fr["view_cos"], pred_level=fr["pred_level"],
               desc=md, has_obs=np.ones(len(Xw), np.uint8))
    onm, ok2m = oracle.search_by_projection_mappoints(F, mps, 6.0, 0.8)
    nm, k2m, _ = dfm.search_local_map(m, cur, dmap, Tc, len(Xw), 6.0, 0.5, LOG_SF)
    assert nm == onm and np.array_equal(k2m, ok2m)
    if r is not None:  # (the lists happened to suffice: then the stage must agree as well)
        assert np.array_equal(r["kp_to_q"], ok2m)
    else:
        assert m.last_stats()["launches"] > 1  # the host resolve re-ran at least one exhausted query
    r2 = dfm.track_stage_local_map(m, cur, kp_slot, dmap, Tc, 60, 1.0, 0.5, LOG_SF, K4, INV_SIGMA2)
    assert r2 is not None and r2["nmatches"] <= 60
    m.close(); dmap.close(); cur.close(); ex.close()


def test_stages_beyond_the_kernels_sizes_hand_the_call_back(S, oracle):
    """More keypoints (> 4096) or more queries (> 4096) than track_resolve_kernel holds, more edges than the launched
    PoseOptimization variant, and a KITTI-sized stage (2000 features, 2500 local points: the 512-thread resolve instance, the
    second PoseOptimization range): the first three answer SO_RETRY_ON_HOST and leave the handle usable, the last runs on the
    device and equals the separate calls."""
    rng = np.random.default_rng(91)
    img = synth.make_canvas(8, 1241, 376)
    K4k = np.asarray(synth.KITTI_K, np.float32)
    # --- 5000 features: more keypoints than the resolve holds
    ex = S.ORBextractor(5000, 1.2, 8, 20, 7)
    cur = S.DeviceFrame(ex, synth.KITTI_K)
    ck, cxy, cd = [a.copy() for a in cur(img)]
    assert len(ck) > 4096
    Tc = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, cxy, ck, cd, synth.KITTI_K, Tc, n_extra=0)
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    m = S.ORBmatcher(0.8, True)
    none = np.full(len(ck), -1, np.int32)
    assert dfm.track_stage_local_map(m, cur, none, dmap, Tc, 2000, 1.0, 0.5, LOG_SF, K4k, INV_SIGMA2) is None
    nm, k2m, _ = dfm.search_local_map(m, cur, dmap, Tc, 2000, 1.0, 0.5, LOG_SF)  # the separate call on the same handle
    assert nm > 500
    m.close(); dmap.close(); cur.close(); ex.close()
    # --- 2000 features (KITTI's setting)
    ex = S.ORBextractor(2000, 1.2, 8, 20, 7)
    cur = S.DeviceFrame(ex, synth.KITTI_K)
    ck, cxy, cd = [a.copy() for a in cur(img)]
    F = FrameView(cxy[:, 0], cxy[:, 1], ck["octave"], ck["angle"], cd, cur.bounds, ex.GetScaleFactors())
    Xw, normal, mx, mn, md = _make_map(rng, cxy, ck, cd, synth.KITTI_K, Tc, n_extra=500)
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    rep = np.concatenate([np.arange(len(Xw))] * 3)  # 3 x ~2500 = more than 4096 queries through a slot list
    m = S.ORBmatcher(0.8, True)
    none = np.full(len(ck), -1, np.int32)
    assert len(rep) > 4096
    assert dfm.track_stage_local_map(m, cur, none, dmap, Tc, len(rep), 1.0, 0.5, LOG_SF, K4k, INV_SIGMA2, local_slot=rep.astype(np.int32)) is None
    # the whole map once (~2500 queries: the 512-thread resolve): ~1700 matches = more edges than the first launch's range
    # (the hint starts at "up to 1024") -> handed back once, then the handle launches the larger range
    cam = oracle.camera(synth.KITTI_K)
    fr = oracle.is_in_frustum(cam, cur.bounds, Tc, Xw, normal, mx, mn, 0.5, LOG_SF, 8)
    mps = dict(in_view=fr["in_view"], proj_x=fr["proj_x"], proj_y=fr["proj_y"], view_cos=fr["view_cos"], pred_level=fr["pred_level"], desc=md,
               has_obs=np.ones(len(Xw), np.uint8))
    onm, ok2m = oracle.search_by_projection_mappoints(F, mps, 1.0, 0.8)
    assert 1024 < onm <= 1792
    first = dfm.track_stage_local_map(m, cur, none, dmap, Tc, len(Xw), 1.0, 0.5, LOG_SF, K4k, INV_SIGMA2)
    assert first is None
    r = dfm.track_stage_local_map(m, cur, none, dmap, Tc, len(Xw), 1.0, 0.5, LOG_SF, K4k, INV_SIGMA2)
    assert r is not None and r["nmatches"] == onm and np.array_equal(r["kp_to_q"], ok2m)
    after = np.where(ok2m >= 0, ok2m, -1).astype(np.int32)
    idx = np.nonzero(after >= 0)[0]
    opt = S.Optimizer()
    n_in, T, outl, _ = opt.PoseOptimization(Tc, K4k, Xw[after[idx]], cxy[idx], INV_SIGMA2[ck["octave"][idx]])
    opt.close()
    assert np.array_equal(r["edge_kp"], idx) and np.array_equal(r["Tcw"], T.reshape(3, 4)) and np.array_equal(r["edge_outlier"], outl)
    assert r["n_inliers"] == n_in
    m.close(); dmap.close(); cur.close(); ex.close()


def test_stages_with_nothing_to_match(S, oracle):
    """A last frame without map points, a local list that is skipped entirely, fewer than three edges: the chains run, report
    no matches, and the pose comes back as it went in (Optimizer.cc:358-359: PoseOptimization returns 0 and touches nothing);
    empty lists are handed back at the submit."""
    rng = np.random.default_rng(5)
    ex, last, lk, lxy, ld, _ = _frame_and_view(S, oracle, 5)
    Tc = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, lxy, lk, ld, synth.EUROC_K, Tc, n_extra=0)
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    cur = S.DeviceFrame(ex, synth.EUROC_K, synth.EUROC_DIST)
    cur(synth.make_canvas(5, 752, 480))
    m = S.ORBmatcher(0.9, True)
    none_last = np.full(len(lk), -1, np.int32)
    r = dfm.track_stage_last_frame(m, cur, last, dmap, Tc, none_last, 15.0, K4, INV_SIGMA2)
    assert r is not None and r["nmatches"] == 0 and r["n_edges"] == 0 and r["n_inliers"] == 0 and (r["kp_to_q"] == -1).all()
    assert np.array_equal(r["Tcw"].reshape(12), Tc)
    two = none_last.copy(); two[:2] = [0, 1]  # at most two matches: still no optimisation
    r = dfm.track_stage_last_frame(m, cur, last, dmap, Tc, two, 15.0, K4, INV_SIGMA2)
    assert r is not None and r["n_edges"] == r["nmatches"] <= 2 and r["n_inliers"] == 0 and np.array_equal(r["Tcw"].reshape(12), Tc)
    kp_slot = np.full(cur.n, -1, np.int32)
    r = dfm.track_stage_local_map(m, cur, kp_slot, dmap, Tc, len(Xw), 1.0, 0.5, LOG_SF, K4, INV_SIGMA2, skip=np.ones(len(Xw), np.uint8))
    assert r is not None and r["nmatches"] == 0 and r["n_edges"] == 0 and not r["in_view"].any()
    assert dfm.track_stage_local_map(m, cur, kp_slot, dmap, Tc, 0, 1.0, 0.5, LOG_SF, K4, INV_SIGMA2) is None  # nothing to search: the plain calls
    nm, k2m, _ = dfm.search_local_map(m, cur, dmap, Tc, len(Xw), 1.0, 0.5, LOG_SF)
    assert nm > 200  # the handle is fine
    m.close(); dmap.close(); cur.close(); last.close(); ex.close()


@pytest.mark.parametrize("seed,dist", [(41, synth.EUROC_DIST), (45, (0, 0, 0, 0)), (46, synth.EUROC_DIST)])
def test_linked_stages_equal_the_stages_with_the_host_in_between(S, oracle, seed, dist):
    """so_track_stage_local_map_submit_after: both stages of a frame enqueued at once - stage 1's pose, its outliers' bindings,
    the excluded keypoints and the already-matched local points travel on the device (track_link_kernel) - against the same two
    stages with the host in between (which test_last_frame_stage_equals_search_plus_pose pins to the oracle): matches, bindings,
    edges, poses, outlier flags, iteration and trial counts of BOTH stages, and the pose repeated over stage 2's edges, to the bit.
    Reference: Tracking::TrackWithMotionModel -> TrackLocalMap, code/src/Tracking.cc:714-807."""
    rng = np.random.default_rng(seed)
    ex, last, lk, lxy, ld, _ = _frame_and_view(S, oracle, seed, dist=dist)
    Tl = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, lxy, lk, ld, synth.EUROC_K, Tl)
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    cur = S.DeviceFrame(ex, synth.EUROC_K, dist)
    cur(synth.make_canvas(seed, 752, 480))
    n_last = len(lk)
    slot = np.where(rng.random(n_last) < 0.6, np.arange(n_last), -1).astype(np.int32)  # 40 % of the last frame's keypoints unbound:
    Tc = Tl.copy()                                                                        # the local-map stage has points left to find
    Tc[3] += 0.004; Tc[7] -= 0.003
    local = rng.permutation(len(Xw))[: len(Xw) * 3 // 4].astype(np.int32)
    bad = (rng.random(len(local)) < 0.05).astype(np.uint8)  # isBad(): known to the host beforehand
    m1, m2, ms = S.ORBmatcher(0.9, True), S.ORBmatcher(0.8, True), S.ORBmatcher(0.8, True)
    # the two stages with the host in between
    r1 = dfm.track_stage_last_frame(m1, cur, last, dmap, Tc, slot, 15.0, K4, INV_SIGMA2)
    assert r1 is not None and r1["nmatches"] > 150
    k2l = r1["kp_to_q"]
    bound = np.where(k2l >= 0, slot[np.maximum(k2l, 0)], -1).astype(np.int32)
    bound[r1["edge_kp"][r1["edge_outlier"] != 0]] = -1
    seen = np.zeros(len(Xw), bool); seen[bound[bound >= 0]] = True
    skip = (bad.astype(bool) | seen[local]).astype(np.uint8)
    r2 = dfm.track_stage_local_map(ms, cur, bound, dmap, r1["Tcw"], len(local), 1.0, 0.5, LOG_SF, K4, INV_SIGMA2, local_slot=local, skip=skip)
    assert r2 is not None and r2["nmatches"] > 50 and r2["n_edges"] > r1["n_edges"] - int((r1["edge_outlier"] != 0).sum())
    T3 = Tc.copy(); T3[3] -= 0.01
    r3 = dfm.track_stage_pose_again(ms, cur, T3)
    # both stages enqueued at once, twice (the link's tables are reused)
    for rep in range(2):
        w1 = dfm.track_stage_last_frame(m1, cur, last, dmap, Tc, slot, 15.0, K4, INV_SIGMA2, wait=False)
        w2 = dfm.track_stage_local_map_after(m2, m1, cur, dmap, len(local), 1.0, 0.5, LOG_SF, K4, INV_SIGMA2, local_slot=local, skip_static=bad)
        assert w1 is not None and w2 is not None
        a1 = w1()
        a2 = w2(a1["Tcw"])
        for k in ("kp_to_q", "edge_kp", "edge_outlier", "Tcw", "nmatches", "n_edges", "n_inliers", "iterations", "trials"):
            assert np.array_equal(a1[k], r1[k]), ("stage 1", k)
            assert np.array_equal(a2[k], r2[k]), ("stage 2", k)
        assert np.array_equal(a2["in_view"], r2["in_view"])
        a3 = dfm.track_stage_pose_again(m2, cur, T3)
        for k in ("edge_kp", "edge_outlier", "Tcw", "n_inliers", "iterations", "trials"):
            assert np.array_equal(a3[k], r3[k]), ("pose again", k)
    for h in (m1, m2, ms, dmap, cur, last, ex):
        h.close()
