"""GPU parity of the device-resident frame path (so_dframe_* / so_map_* / so_track_search_*) against the CPU
oracle, through the C ABI.  Everything here is integer / index work or float work performed in the oracle's
operation order: bit-exact."""
import numpy as np
import pytest

from swarmmap_amd import dframe as dfm
from swarmmap_amd import synth
from swarmmap_amd.matcher import FrameView

pytestmark = pytest.mark.gpu

LOG_SF = float(np.log(np.float32(1.2)))


@pytest.fixture(scope="module")
def S():
    import swarmmap_amd
    assert swarmmap_amd.device_count() > 0, "these tests need a GPU"
    return swarmmap_amd


def _oracle_frame(oracle, img, nfeat, K, dist):
    cfg = oracle.config(nfeat)
    kps, desc = oracle.extract(cfg, img)
    cam = oracle.camera(K, dist)
    xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    un = oracle.undistort_keypoints(cam, xy)
    b = oracle.image_bounds(cam, img.shape[1], img.shape[0])
    g = oracle.assign_features_to_grid(un, b)
    return kps, desc, un, b, g, cam


@pytest.mark.parametrize("size,K,dist,nfeat", [(synth.EUROC, synth.EUROC_K, synth.EUROC_DIST, 1000),
                                               (synth.EUROC, synth.EUROC_K, (0, 0, 0, 0), 1000),
                                               (synth.KITTI, synth.KITTI_K, (0, 0, 0, 0), 2000),
                                               ((376, 240), synth.EUROC_K, synth.EUROC_DIST, 500)])
def test_device_frame_equals_extract_undistort_grid(S, oracle, size, K, dist, nfeat):
    img = synth.make_canvas(5, size[0], size[1])
    ex = S.ORBextractor(nfeat, 1.2, 8, 20, 7)
    f = S.DeviceFrame(ex, K, dist)
    kps, xy_un, desc = f(img)
    okps, odesc, oun, ob, og, _ = _oracle_frame(oracle, img, nfeat, K, dist)
    assert len(kps) == len(okps) > 100
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    assert xy_un.tobytes() == oun.tobytes() and f.bounds.tobytes() == ob.tobytes()
    cs, items = f.grid()
    assert np.array_equal(cs, og["cell_start"]) and np.array_equal(items, og["cell_items"])
    # a second frame through the same handle (buffers are reused), and the pipelined two-handle pattern
    img2 = synth.make_canvas(6, size[0], size[1])
    g = S.DeviceFrame(ex, K, dist)
    g.submit(img2)
    k2, u2, d2 = g.collect()
    o2 = _oracle_frame(oracle, img2, nfeat, K, dist)
    assert k2.tobytes() == o2[0].tobytes() and np.array_equal(d2, o2[1]) and u2.tobytes() == o2[2].tobytes()
    # the first handle still holds frame 1 on the device
    cs1, items1 = f.grid()
    assert np.array_equal(items1, og["cell_items"])
    g.close(); f.close(); ex.close()


@pytest.mark.parametrize("size,nfeat", [(synth.EUROC, 1000), (synth.KITTI, 2000), ((751, 333), 700)])
def test_pinned_host_image_is_ingested_in_place(S, oracle, size, nfeat):
    """A host image in pinned memory is read by the ingest kernel straight over PCIe (no DMA, no landing buffer), for
    row lengths of every alignment; a pageable image takes the copy path.  Both give the oracle's frame."""
    import torch
    img = synth.make_canvas(9, size[0], size[1])
    pinned = torch.from_numpy(img.copy()).pin_memory().numpy()
    ex = S.ORBextractor(nfeat, 1.2, 8, 20, 7)
    f = S.DeviceFrame(ex, synth.EUROC_K)
    k1, u1, d1 = [a.copy() for a in f(pinned)]
    k2, u2, d2 = [a.copy() for a in f(img)]
    okps, odesc = oracle.extract(oracle.config(nfeat), img)
    assert k1.tobytes() == okps.tobytes() and np.array_equal(d1, odesc)
    assert k2.tobytes() == okps.tobytes() and np.array_equal(d2, odesc) and u1.tobytes() == u2.tobytes()
    f.close(); ex.close()


def test_device_frame_flat_image_and_errors(S):
    ex = S.ORBextractor(500, 1.2, 8, 20, 7)
    f = S.DeviceFrame(ex, synth.EUROC_K)
    kps, xy_un, desc = f(np.full((240, 376), 128, np.uint8))
    assert len(kps) == 0 and len(desc) == 0
    with pytest.raises(S.SwarmOrbError):
        f.collect()  # nothing submitted
    kps, _, _ = f(synth.make_canvas(1, 376, 240))
    assert len(kps) > 100
    f.close(); ex.close()


def _frame_and_view(S, oracle, seed, nfeat=1000, size=synth.EUROC, K=synth.EUROC_K, dist=synth.EUROC_DIST):
    img = synth.make_canvas(seed, size[0], size[1])
    ex = S.ORBextractor(nfeat, 1.2, 8, 20, 7)
    f = S.DeviceFrame(ex, K, dist)
    kps, xy_un, desc = [a.copy() for a in f(img)]
    F = FrameView(xy_un[:, 0], xy_un[:, 1], kps["octave"], kps["angle"], desc, f.bounds, ex.GetScaleFactors())
    return ex, f, kps, xy_un, desc, F


def _queries_near(rng, xy_un, kps, desc, nq, jitter, p_flip):
    n = len(kps)
    k = rng.integers(0, n, nq)
    return k, (xy_un[k, 0] + rng.normal(0, jitter, nq)).astype(np.float32), \
        (xy_un[k, 1] + rng.normal(0, jitter, nq)).astype(np.float32), synth.flip_bits(rng, desc[k], p_flip)


@pytest.mark.parametrize("seed,excl", [(31, False), (32, True)])
def test_host_query_searches_on_a_device_frame(S, oracle, seed, excl):
    """so_search_by_projection_*_dframe == the oracle's searches over the same frame given as host arrays."""
    ex, f, kps, xy_un, desc, F = _frame_and_view(S, oracle, seed)
    rng = np.random.default_rng(seed)
    n = len(kps)
    excluded = (rng.random(n) < 0.3).astype(np.uint8) if excl else None
    F.excluded = excluded
    # M2
    k, u, v, qd = _queries_near(rng, xy_un, kps, desc, 900, 3.0, 0.1)
    last = dict(valid=(rng.random(900) < 0.7).astype(np.uint8), u=u, v=v,
                octave=np.clip(kps["octave"][k] + rng.integers(-1, 2, 900), 0, 7).astype(np.int32),
                angle=((kps["angle"][k] + 4) % 360).astype(np.float32), desc=qd, has_obs=np.ones(900, np.uint8))
    m = S.ORBmatcher(0.9, True)
    nm, k2l = dfm.search_lastframe_dframe(m, f, last, 15.0, excluded)
    onm, ok2l = oracle.search_by_projection_lastframe(F, last, 15.0, True)
    assert nm == onm > 100 and np.array_equal(k2l, ok2l)
    # M1
    k, u, v, qd = _queries_near(rng, xy_un, kps, desc, 2500, 2.0, 0.15)
    mps = dict(in_view=(rng.random(2500) < 0.9).astype(np.uint8), proj_x=u, proj_y=v,
               view_cos=np.where(rng.random(2500) < 0.5, 0.9995, 0.9).astype(np.float32),
               pred_level=np.clip(kps["octave"][k] + rng.integers(0, 2, 2500), 0, 7).astype(np.int32), desc=qd,
               has_obs=(rng.random(2500) < 0.97).astype(np.uint8))
    m1 = S.ORBmatcher(0.8, True)
    nm, k2m = dfm.search_mappoints_dframe(m1, f, mps, 1.0, excluded)
    onm, ok2m = oracle.search_by_projection_mappoints(F, mps, 1.0, 0.8)
    assert nm == onm > 100 and np.array_equal(k2m, ok2m)
    # the host-array entry point on the same handle afterwards (switches back to the staged upload)
    nm2, k2m2 = m1.SearchByProjectionMapPoints(F, mps, 1.0)
    assert nm2 == onm and np.array_equal(k2m2, ok2m)
    m.close(); m1.close(); f.close(); ex.close()


def _make_map(rng, xy_un, kps, desc, K, Tcw, n_extra=600):
    """Map points = the frame's keypoints back-projected to random depths through Tcw (so they re-project onto the
    keypoints), plus points that fail each isInFrustum test."""
    fx, fy, cx, cy = K
    n = len(kps)
    z = rng.uniform(1.5, 9.0, n)
    pc = np.stack([(xy_un[:, 0] - cx) / fx * z, (xy_un[:, 1] - cy) / fy * z, z], 1)
    T = np.asarray(Tcw, np.float64).reshape(3, 4)
    R, t = T[:, :3], T[:, 3]
    Xw = (pc - t) @ R  # R^T (pc - t)
    Ow = -R.T @ t
    view = Xw - Ow
    dist = np.linalg.norm(view, axis=1)
    normal = view / dist[:, None]
    tilt = rng.random(n) < 0.15  # seen at a grazing angle
    normal[tilt] = np.roll(normal[tilt], 1, axis=1)
    sf = synth.SCALE_FACTORS
    max_d = dist * sf[kps["octave"]] * rng.uniform(0.95, 1.3, n)
    min_d = max_d / sf[7] * rng.uniform(0.7, 1.0, n)
    far = rng.random(n) < 0.05
    max_d[far] *= 0.3  # outside the scale-invariance range
    extra = rng.normal(0, 6.0, (n_extra, 3))  # behind the camera / outside the image
    Xw = np.concatenate([Xw, extra]); normal = np.concatenate([normal, np.tile([0, 0, 1.0], (n_extra, 1))])
    max_d = np.concatenate([max_d, np.full(n_extra, 20.0)]); min_d = np.concatenate([min_d, np.full(n_extra, 0.1)])
    d = np.concatenate([synth.flip_bits(rng, desc, 0.1), rng.integers(0, 256, (n_extra, 32)).astype(np.uint8)])
    return (Xw.astype(np.float32), normal.astype(np.float32), max_d.astype(np.float32), min_d.astype(np.float32), d)


def _pose(rng):
    R = synth._rodrigues(rng.normal(0, 0.05, 3))
    t = rng.normal(0, 0.2, 3)
    return np.hstack([R, t[:, None]]).astype(np.float32).reshape(12)


@pytest.mark.parametrize("seed,dist", [(41, synth.EUROC_DIST), (42, (0, 0, 0, 0))])
def test_track_search_last_frame_matches_oracle(S, oracle, seed, dist):
    """Fused projection + search == oracle projection (ORBmatcher.cc:1251-1270) + oracle search."""
    rng = np.random.default_rng(seed)
    ex, last, lk, lxy, ld, _ = _frame_and_view(S, oracle, seed, dist=dist)
    Tl = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, lxy, lk, ld, synth.EUROC_K, Tl)
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    # the current frame: another image region, its own device frame (two handles on one extractor)
    cur = S.DeviceFrame(ex, synth.EUROC_K, dist)
    ck, cxy, cd = [a.copy() for a in cur(synth.make_canvas(seed, 752, 480))]  # same scene -> real matches
    F = FrameView(cxy[:, 0], cxy[:, 1], ck["octave"], ck["angle"], cd, cur.bounds, ex.GetScaleFactors())
    n_last = len(lk)
    slot = np.where(rng.random(n_last) < 0.75, np.arange(n_last), -1).astype(np.int32)
    slot[:5] = len(Xw) - 1 - np.arange(5)  # a few far-away points
    Tc = Tl.copy()
    Tc[3] += 0.004; Tc[7] -= 0.003  # small motion
    cam = oracle.camera(synth.EUROC_K, dist)
    valid, u, v = oracle.project_last_frame(cam, cur.bounds, Tc, Xw[np.maximum(slot, 0)], slot >= 0)
    lastd = dict(valid=valid, u=u, v=v, octave=lk["octave"], angle=lk["angle"], desc=md[np.maximum(slot, 0)],
                 has_obs=np.ones(n_last, np.uint8))
    for th, ori in ((15.0, True), (30.0, False)):
        m = S.ORBmatcher(0.9, ori)
        nm, k2l = dfm.search_last_frame(m, cur, last, dmap, Tc, slot, th)
        onm, ok2l = oracle.search_by_projection_lastframe(F, lastd, th, ori)
        assert nm == onm and np.array_equal(k2l, ok2l)
        assert nm > 200
        m.close()
    dmap.close(); cur.close(); last.close(); ex.close()


@pytest.mark.parametrize("seed,dist,th", [(51, synth.EUROC_DIST, 1.0), (52, (0, 0, 0, 0), 3.0)])
def test_track_search_local_map_matches_oracle(S, oracle, seed, dist, th):
    """Fused isInFrustum + search == oracle isInFrustum (Frame.cc:316-375) + oracle search."""
    rng = np.random.default_rng(seed)
    ex, cur, ck, cxy, cd, F = _frame_and_view(S, oracle, seed, dist=dist)
    Tc = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, cxy, ck, cd, synth.EUROC_K, Tc)
    # duplicate part of the map so that several points compete for the same keypoints
    dup = rng.integers(0, len(ck), 800)
    Xw = np.concatenate([Xw, Xw[dup] + rng.normal(0, 0.004, (800, 3)).astype(np.float32)])
    normal = np.concatenate([normal, normal[dup]]); mx = np.concatenate([mx, mx[dup]]); mn = np.concatenate([mn, mn[dup]])
    md = np.concatenate([md, synth.flip_bits(rng, md[dup], 0.05)])
    dmap = S.DeviceMap()
    dmap.append(Xw[:1000], normal[:1000], mx[:1000], mn[:1000], md[:1000])
    dmap.append(Xw[1000:], normal[1000:], mx[1000:], mn[1000:], md[1000:])  # grows the table
    assert len(dmap) == len(Xw)
    n_map = len(Xw)
    excluded = (rng.random(len(ck)) < 0.35).astype(np.uint8)
    F.excluded = excluded
    cam = oracle.camera(synth.EUROC_K, dist)
    for local in (None, rng.permutation(n_map)[: n_map * 2 // 3].astype(np.int32)):
        idx = np.arange(n_map) if local is None else local
        skip = (rng.random(len(idx)) < 0.2).astype(np.uint8)
        fr = oracle.is_in_frustum(cam, cur.bounds, Tc, Xw[idx], normal[idx], mx[idx], mn[idx], 0.5, LOG_SF, 8)
        in_view = fr["in_view"] & (1 - skip)
        mps = dict(in_view=in_view, proj_x=fr["proj_x"], proj_y=fr["proj_y"], view_cos=fr["view_cos"],
                   pred_level=fr["pred_level"], desc=md[idx], has_obs=np.ones(len(idx), np.uint8))
        m = S.ORBmatcher(0.8, True)
        if local is not None:
            m.reserve(4 * n_map)  # so_matcher_reserve: the staging sized up front (one pass with, one without)
        nm, k2m, view = dfm.search_local_map(m, cur, dmap, Tc, len(idx), th, 0.5, LOG_SF, local_slot=local, skip=skip,
                                             excluded=excluded)
        onm, ok2m = oracle.search_by_projection_mappoints(F, mps, th, 0.8)
        assert np.array_equal(view, in_view)
        assert nm == onm and np.array_equal(k2m, ok2m)
        assert nm > 150 and 0 < in_view.sum() < len(idx)
        m.close()
    # SetWorldPos after bundle adjustment: scattered position updates reach the table
    slots = rng.permutation(n_map)[:300].astype(np.int32)
    newX = (Xw[slots] + 0.5).astype(np.float32)
    dmap.write_positions(slots, newX)
    Xr, dr = dmap.read(0, n_map)
    Xe = Xw.copy(); Xe[slots] = newX
    assert np.array_equal(Xr, Xe) and np.array_equal(dr, md)
    dmap.close(); cur.close(); ex.close()


def test_reuse_flag_does_not_leak_to_another_frame(S, oracle):
    """so_matcher_reuse_frame is one-shot even when the next call returns early, and only applies to the
    very frame the handle holds."""
    fr1, mps1 = synth.make_m1_case(61, 800, 1500)
    fr2, mps2 = synth.make_m1_case(62, 800, 1500)  # a different frame with the same keypoint count
    F1 = FrameView(fr1["x"], fr1["y"], fr1["octave"], fr1["angle"], fr1["desc"], fr1["bounds"], fr1["scale_factors"])
    F2 = FrameView(fr2["x"], fr2["y"], fr2["octave"], fr2["angle"], fr2["desc"], fr2["bounds"], fr2["scale_factors"])
    m = S.ORBmatcher(0.8)
    m.SearchByProjectionMapPoints(F1, mps1, 1.0)
    m.reuse_frame()
    empty = {k: v[:0] for k, v in mps1.items()}
    m.SearchByProjectionMapPoints(F1, empty, 1.0)  # returns before any upload: the flag must be consumed here
    nm, k2m = m.SearchByProjectionMapPoints(F2, mps2, 1.0)
    onm, ok2m = oracle.search_by_projection_mappoints(F2, mps2, 1.0, 0.8)
    assert nm == onm and np.array_equal(k2m, ok2m)
    m.reuse_frame()  # armed, but the next view is another frame of equal size: must not be honoured
    nm, k2m = m.SearchByProjectionMapPoints(F1, mps1, 1.0)
    onm, ok2m = oracle.search_by_projection_mappoints(F1, mps1, 1.0, 0.8)
    assert nm == onm and np.array_equal(k2m, ok2m)
    m.reuse_frame()  # the legitimate use: same frame again, different gate
    F1.excluded = (np.arange(800) % 3 == 0).astype(np.uint8)
    nm, k2m = m.SearchByProjectionMapPoints(F1, mps1, 1.0)
    onm, ok2m = oracle.search_by_projection_mappoints(F1, mps1, 1.0, 0.8)
    assert nm == onm and np.array_equal(k2m, ok2m)
    m.close()


def test_large_frame_and_large_local_map_take_the_staged_paths(S, oracle):
    """5000 features per frame: the extractor places DistributeOctTree on the host (level quota > 1020), the
    device-resident frame then reads the host-mapped results, and more than 4096 candidates / 16384 queries make the
    tracking searches stage their gates with a copy kernel instead of carrying them in the kernel arguments."""
    rng = np.random.default_rng(71)
    img = synth.make_canvas(8, 1241, 376)
    ex = S.ORBextractor(5000, 1.2, 8, 20, 7)
    cur = S.DeviceFrame(ex, synth.KITTI_K)
    ck, cxy, cd = [a.copy() for a in cur(img)]
    assert not ex.quadtree_on_device and len(ck) > 4200
    okps, odesc = oracle.extract(oracle.config(5000), img)
    assert ck.tobytes() == okps.tobytes() and np.array_equal(cd, odesc)
    sf = ex.GetScaleFactors()
    F = FrameView(cxy[:, 0], cxy[:, 1], ck["octave"], ck["angle"], cd, cur.bounds, sf)
    Tc = _pose(rng)
    Xw, normal, mx, mn, md = _make_map(rng, cxy, ck, cd, synth.KITTI_K, Tc, n_extra=500)
    rep = 4  # 4 x ~5000 map points: more than 16384 queries
    Xw = np.concatenate([Xw + rng.normal(0, 0.003, Xw.shape).astype(np.float32) for _ in range(rep)])
    normal = np.tile(normal, (rep, 1)); mx = np.tile(mx, rep); mn = np.tile(mn, rep)
    md = np.concatenate([synth.flip_bits(rng, md, 0.03) for _ in range(rep)])
    assert len(Xw) > 16384
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, md)
    excluded = (rng.random(len(ck)) < 0.3).astype(np.uint8)
    F.excluded = excluded
    skip = (rng.random(len(Xw)) < 0.1).astype(np.uint8)
    cam = oracle.camera(synth.KITTI_K)
    fr = oracle.is_in_frustum(cam, cur.bounds, Tc, Xw, normal, mx, mn, 0.5, LOG_SF, 8)
    in_view = fr["in_view"] & (1 - skip)
    mps = dict(in_view=in_view, proj_x=fr["proj_x"], proj_y=fr["proj_y"], view_cos=fr["view_cos"], pred_level=fr["pred_level"],
               desc=md, has_obs=np.ones(len(Xw), np.uint8))
    m = S.ORBmatcher(0.8, True)
    nm, k2m, view = dfm.search_local_map(m, cur, dmap, Tc, len(Xw), 1.0, 0.5, LOG_SF, skip=skip, excluded=excluded)
    onm, ok2m = oracle.search_by_projection_mappoints(F, mps, 1.0, 0.8)
    assert np.array_equal(view, in_view) and nm == onm > 500 and np.array_equal(k2m, ok2m)
    # motion-model search against itself as "last frame" (more than 4096 candidates: staged gate)
    last = S.DeviceFrame(ex, synth.KITTI_K)
    lk, lxy, ld = [a.copy() for a in last(img)]
    slot = np.where(rng.random(len(lk)) < 0.8, np.arange(len(lk)), -1).astype(np.int32)
    valid, u, v = oracle.project_last_frame(cam, cur.bounds, Tc, Xw[np.maximum(slot, 0)], slot >= 0)
    lastd = dict(valid=valid, u=u, v=v, octave=lk["octave"], angle=lk["angle"], desc=md[np.maximum(slot, 0)],
                 has_obs=np.ones(len(lk), np.uint8))
    m2 = S.ORBmatcher(0.9, True)
    nm, k2l = dfm.search_last_frame(m2, cur, last, dmap, Tc, slot, 15.0, excluded=excluded)
    onm, ok2l = oracle.search_by_projection_lastframe(F, lastd, 15.0, True)
    assert nm == onm > 500 and np.array_equal(k2l, ok2l)
    m.close(); m2.close(); dmap.close(); last.close(); cur.close(); ex.close()


def test_tracking_searches_on_empty_inputs(S):
    """No map, no local points, a frame without keypoints, a last frame without map points: every combination
    returns zero matches and touches nothing it should not."""
    ex = S.ORBextractor(500, 1.2, 8, 20, 7)
    full, flat = S.DeviceFrame(ex, synth.EUROC_K), S.DeviceFrame(ex, synth.EUROC_K)
    kps, _, _ = full(synth.make_canvas(2, 376, 240))
    k0, _, _ = flat(np.full((240, 376), 90, np.uint8))
    assert len(k0) == 0 and len(kps) > 100
    dmap = S.DeviceMap()
    T = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)
    m = S.ORBmatcher(0.8, True)
    nm, k2m, view = dfm.search_local_map(m, full, dmap, T, 0, 1.0, 0.5, LOG_SF)           # empty map
    assert nm == 0 and np.all(k2m == -1) and len(view) == 0
    nm, k2l = dfm.search_last_frame(m, full, full, dmap, T, np.full(len(kps), -1, np.int32), 15.0)  # nothing to project
    assert nm == 0 and np.all(k2l == -1)
    X = np.array([[-0.5, -0.3, 3.0]], np.float32)  # projects to (290, 202): inside the 376 x 240 image
    dmap.append(X, np.array([[0, 0, 1.0]], np.float32), np.array([10.0], np.float32), np.array([0.1], np.float32),
                np.zeros((1, 32), np.uint8))
    nm, k2m, view = dfm.search_local_map(m, flat, dmap, T, 1, 1.0, 0.5, LOG_SF)            # a frame without keypoints
    assert nm == 0 and len(k2m) == 0 and view[0] == 1                                       # ... still sees the point
    nm, k2l = dfm.search_last_frame(m, flat, full, dmap, T, np.zeros(len(kps), np.int32), 15.0)
    assert nm == 0 and len(k2l) == 0
    nm, k2l = dfm.search_last_frame(m, full, flat, dmap, T, np.zeros(0, np.int32), 15.0)    # an empty last frame
    assert nm == 0 and np.all(k2l == -1)
    slots = np.full(len(kps), 7, np.int32)  # slots beyond the table are "no map point", not a fault
    nm, k2l = dfm.search_last_frame(m, full, full, dmap, T, slots, 15.0)
    assert nm == 0
    m.close(); dmap.close(); full.close(); flat.close(); ex.close()
