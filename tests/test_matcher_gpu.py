"""GPU parity of the HIP matcher against the CPU oracle, through the C ABI.  Integer work: bit-exact."""
import numpy as np
import pytest

from swarmmap_amd import synth
from swarmmap_amd.matcher import FrameView

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import swarmmap_amd
    assert swarmmap_amd.device_count() > 0, "these tests need a GPU"
    return swarmmap_amd


def _frame(fr, excluded=True):
    return FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"],
                     fr.get("excluded") if excluded else None)


@pytest.mark.parametrize("seed,n_kp,n_mp,th,ratio", [(1, 1000, 2000, 1.0, 0.8), (2, 1000, 3000, 3.0, 0.8),
                                                     (3, 2000, 3000, 5.0, 0.8), (4, 500, 300, 1.0, 0.6)])
def test_m1_search_by_projection_mappoints(S, oracle, seed, n_kp, n_mp, th, ratio):
    fr, mps = synth.make_m1_case(seed, n_kp, n_mp)
    F = _frame(fr)
    m = S.ORBmatcher(ratio)
    nm, kp_to_mp = m.SearchByProjectionMapPoints(F, mps, th)
    onm, okp = oracle.search_by_projection_mappoints(F, mps, th, ratio)
    assert nm == onm and np.array_equal(kp_to_mp, okp)
    assert nm > 0
    m.close()


@pytest.mark.parametrize("seed,th,ori", [(11, 15.0, True), (12, 30.0, True), (13, 7.0, False)])
def test_m2_search_by_projection_lastframe(S, oracle, seed, th, ori):
    fr, last = synth.make_m2_case(seed)
    F = _frame(fr)
    m = S.ORBmatcher(0.9, ori)
    nm, kp_to_last = m.SearchByProjectionLastFrame(F, last, th)
    onm, okp = oracle.search_by_projection_lastframe(F, last, th, ori)
    assert nm == onm and np.array_equal(kp_to_last, okp)
    assert nm > 50
    m.close()


@pytest.mark.parametrize("seed,n_kp,window", [(21, 2000, 100), (22, 1000, 50), (23, 4000, 100)])
def test_m4_search_for_initialization(S, oracle, seed, n_kp, window):
    f1, f2, prev = synth.make_m4_case(seed, n_kp)
    F1, F2 = _frame(f1), _frame(f2)
    m = S.ORBmatcher(0.9, True)
    nm, m12, pm = m.SearchForInitialization(F1, F2, prev, window)
    onm, om12, opm = oracle.search_for_initialization(F1, F2, prev, window, 0.9, True)
    assert nm == onm and np.array_equal(m12, om12) and np.array_equal(pm, opm)
    assert nm > 100
    m.close()


def test_exhausted_topk_list_is_rerun_on_gpu(S, oracle):
    """A tight cluster: 40 near-identical keypoints, 40 map points all projecting onto it with a wide window, so
    every later map point finds its whole K-list taken by earlier ones and needs the exact single-query re-run."""
    rng = np.random.default_rng(5)
    fr = synth.make_frame_arrays(rng, 600, dup_frac=0.0)
    base = rng.integers(0, 256, 32).astype(np.uint8)
    for k in range(40):
        fr["x"][k] = 300 + rng.uniform(-6, 6)
        fr["y"][k] = 200 + rng.uniform(-6, 6)
        fr["octave"][k] = 1
        fr["desc"][k] = synth.flip_bits(rng, base[None, :], 0.02)[0]
    n_mp = 40
    mps = dict(in_view=np.ones(n_mp, np.uint8), proj_x=np.full(n_mp, 300, np.float32),
               proj_y=np.full(n_mp, 200, np.float32), view_cos=np.full(n_mp, 0.9, np.float32),
               pred_level=np.full(n_mp, 1, np.int32), desc=synth.flip_bits(rng, np.tile(base, (n_mp, 1)), 0.02),
               has_obs=np.ones(n_mp, np.uint8))
    F = _frame(fr, excluded=False)
    m = S.ORBmatcher(1.0)  # ratio 1.0: the ratio test never rejects, so all 40 get bound one after another
    nm, kp_to_mp = m.SearchByProjectionMapPoints(F, mps, 5.0)
    onm, okp = oracle.search_by_projection_mappoints(F, mps, 5.0, 1.0)
    assert nm == onm and np.array_equal(kp_to_mp, okp)
    assert nm >= 30
    # same for the last-frame variant
    last = dict(valid=np.ones(n_mp, np.uint8), u=mps["proj_x"], v=mps["proj_y"], octave=np.full(n_mp, 1, np.int32),
                angle=np.zeros(n_mp, np.float32), desc=mps["desc"], has_obs=np.ones(n_mp, np.uint8))
    nm2, k2 = m.SearchByProjectionLastFrame(F, last, 20.0)
    onm2, ok2 = oracle.search_by_projection_lastframe(F, last, 20.0, True)
    assert nm2 == onm2 and np.array_equal(k2, ok2)
    m.close()


def test_topk_building_block_matches_grid_scan(S, oracle):
    rng = np.random.default_rng(9)
    fr = synth.make_frame_arrays(rng, 1200)
    F = _frame(fr, excluded=False)
    nq = 200
    u = rng.uniform(0, 752, nq).astype(np.float32)
    v = rng.uniform(0, 480, nq).astype(np.float32)
    r = rng.uniform(5, 150, nq).astype(np.float32)
    mn = rng.integers(-1, 3, nq).astype(np.int32)
    mx = rng.integers(-1, 8, nq).astype(np.int32)
    qd = rng.integers(0, 256, (nq, 32)).astype(np.uint8)
    m = S.ORBmatcher()
    K = 6
    idx, dist, cnt = m.topk(F, u, v, r, mn, mx, qd, K)
    for i in range(nq):
        cand = oracle.features_in_area(F, u[i], v[i], r[i], mn[i], mx[i])
        assert cnt[i] == len(cand)
        d = np.array([oracle.descriptor_distance(qd[i], fr["desc"][c]) for c in cand], np.int64)
        order = np.argsort(d, kind="stable")[:K]  # stable: ties keep grid-traversal order
        want_idx = cand[order].tolist() + [-1] * (K - len(order))
        want_d = d[order].tolist() + [256] * (K - len(order))
        assert idx[i].tolist() == want_idx and dist[i].tolist() == want_d
    # a window bigger than the LDS list (brute force over everything) takes the exact re-scan path
    big = synth.make_frame_arrays(rng, 3000, dup_frac=0.0)
    FB = _frame(big, excluded=False)
    idx, dist, cnt = m.topk(FB, [376.0], [240.0], [5000.0], [-1], [-1], qd[:1], 5)
    in_grid = len(oracle.features_in_area(FB, 376.0, 240.0, 5000.0, -1, -1))
    assert cnt[0] == in_grid and in_grid > 1024
    cand = oracle.features_in_area(FB, 376.0, 240.0, 5000.0, -1, -1)
    d = np.array([oracle.descriptor_distance(qd[0], big["desc"][c]) for c in cand])
    order = np.argsort(d, kind="stable")[:5]
    assert idx[0].tolist() == cand[order].tolist() and dist[0].tolist() == d[order].tolist()
    m.close()


@pytest.mark.parametrize("na,nb", [(1000, 2000), (37, 5), (3, 0), (2000, 16000)])
def test_bruteforce_top2(S, oracle, na, nb):
    rng = np.random.default_rng(na + nb)
    B = rng.integers(0, 256, (nb, 32)).astype(np.uint8)
    if nb:
        A = synth.flip_bits(rng, B[rng.integers(0, nb, na)], 0.15)
        A[::7] = B[rng.integers(0, nb, len(A[::7]))]  # exact duplicates -> distance 0 and ties
    else:
        A = rng.integers(0, 256, (na, 32)).astype(np.uint8)
    m = S.ORBmatcher()
    bi, bd, sd = m.hamming_top2(A, B)
    obi, obd, osd = oracle.hamming_top2(A, B)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd)
    if nb:
        import torch
        dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        torch.cuda.synchronize()
        bi2, bd2, sd2 = m.hamming_top2_device(dA.data_ptr(), na, dB.data_ptr(), nb)
        assert np.array_equal(bi2, obi) and np.array_equal(bd2, obd) and np.array_equal(sd2, osd)
    m.close()


def test_matcher_edge_cases(S, oracle):
    rng = np.random.default_rng(2)
    fr = synth.make_frame_arrays(rng, 50)
    F = _frame(fr, excluded=False)
    m = S.ORBmatcher(0.8)
    empty = dict(in_view=np.zeros(0, np.uint8), proj_x=np.zeros(0, np.float32), proj_y=np.zeros(0, np.float32),
                 view_cos=np.zeros(0, np.float32), pred_level=np.zeros(0, np.int32),
                 desc=np.zeros((0, 32), np.uint8), has_obs=np.zeros(0, np.uint8))
    nm, k = m.SearchByProjectionMapPoints(F, empty, 1.0)
    assert nm == 0 and np.all(k == -1)
    # every query out of view / projected far outside the image
    fr2, mps = synth.make_m1_case(8, 300, 200)
    mps["proj_x"][:] = 5000.0
    F2 = _frame(fr2)
    nm, k = m.SearchByProjectionMapPoints(F2, mps, 1.0)
    onm, ok = oracle.search_by_projection_mappoints(F2, mps, 1.0, 0.8)
    assert nm == onm == 0 and np.array_equal(k, ok)
    m.close()


# ---- M3 / M5 / M6 / M7 -------------------------------------------------------------------------
@pytest.mark.parametrize("variant,seed,ratio", [(0, 31, 0.7), (1, 32, 0.75), (0, 33, 0.9), (1, 34, 0.6)])
def test_m3_search_by_bow(S, oracle, variant, seed, ratio):
    from swarmmap_amd.matcher import FeatureVector
    kf1, node1, kf2, node2, _ = synth.make_bow_case(seed, 1200, 1000)
    fv1, fv2 = FeatureVector(node1), FeatureVector(node2)
    m = S.ORBmatcher(ratio, True)
    nm, m2, m1 = m.SearchByBoW(variant, kf1, fv1, kf2, fv2)
    onm, om2, om1 = oracle.search_by_bow(variant, kf1, fv1, kf2, fv2, ratio, True)
    assert nm == onm and np.array_equal(m2, om2) and np.array_equal(m1, om1)
    assert nm > 50
    m.close()


def test_m3_dense_node_exhausts_topk(S, oracle):
    """One vocabulary node holding 60 near-identical features on both sides: later queries find their K-list taken."""
    from swarmmap_amd.matcher import FeatureVector
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, 32).astype(np.uint8)
    n = 60
    d2 = synth.flip_bits(rng, np.tile(base, (n, 1)), 0.01)
    d1 = synth.flip_bits(rng, d2, 0.01)
    kf1 = dict(desc=d1, angle=np.zeros(n, np.float32), valid=np.ones(n, np.uint8))
    kf2 = dict(desc=d2, angle=np.zeros(n, np.float32), valid=np.ones(n, np.uint8))
    fv = FeatureVector(np.zeros(n, np.int32))
    m = S.ORBmatcher(1.01, False)  # ratio > 1: the ratio test passes even for equal distances
    for variant in (0, 1):
        nm, m2, m1 = m.SearchByBoW(variant, kf1, fv, kf2, fv)
        onm, om2, om1 = oracle.search_by_bow(variant, kf1, fv, kf2, fv, 1.01, False)
        assert nm == onm and np.array_equal(m2, om2) and np.array_equal(m1, om1)
        assert nm > 30
    m.close()


@pytest.mark.parametrize("seed", [41, 42])
def test_m5_search_for_triangulation(S, oracle, seed):
    from swarmmap_amd.matcher import FeatureVector
    rng = np.random.default_rng(seed)
    kf1, node1, kf2, node2, src = synth.make_bow_case(seed, 1000, 1000, p_flip=0.06)
    # a fundamental matrix for a sideways translation: epipolar lines are (almost) horizontal, y2 ~ y1
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-6, (3, 3)).astype(np.float32)
    kf1["y"] = (kf2["y"][src] + rng.normal(0, 0.8, len(src))).astype(np.float32)
    kf1["x"] = (kf2["x"][src] + rng.uniform(-30, 30, len(src))).astype(np.float32)
    sf = synth.SCALE_FACTORS
    m = S.ORBmatcher(0.6, True)
    fv1, fv2 = FeatureVector(node1), FeatureVector(node2)
    nm, m12 = m.SearchForTriangulation(kf1, fv1, kf2, fv2, F12, (900.0, 240.0), sf, sf * sf)
    onm, om12 = oracle.search_for_triangulation(kf1, fv1, kf2, fv2, F12, (900.0, 240.0), sf, sf * sf, True)
    assert nm == onm and np.array_equal(m12, om12)
    assert nm > 100
    m.close()


@pytest.mark.parametrize("seed,gate", [(51, False), (52, True), (53, True)])
def test_m6_m7_window_best(S, oracle, seed, gate):
    rng = np.random.default_rng(seed)
    fr = synth.make_frame_arrays(rng, 1500)
    KF = _frame(fr, excluded=False)
    q = synth.make_window_queries(seed, fr, 2500, jitter=1.5, th=3.0)
    inv = (1.0 / (synth.SCALE_FACTORS ** 2)).astype(np.float32)
    m = S.ORBmatcher()
    bi, bd = m.SearchWindowBest(KF, q, gate, inv)
    obi, obd = oracle.search_window_best(KF, q, gate, inv)
    assert np.array_equal(bi, obi) and np.array_equal(bd, obd)
    assert (bi >= 0).sum() > 500
    if gate:  # the chi2 gate removes candidates
        bi2, _ = m.SearchWindowBest(KF, q, False, inv)
        assert (bi2 >= 0).sum() >= (bi >= 0).sum()
    m.close()


@pytest.mark.parametrize("seed,max_dist,ori", [(61, 50, False), (62, 100, True), (63, 64, True)])
def test_m7_window_greedy(S, oracle, seed, max_dist, ori):
    rng = np.random.default_rng(seed)
    fr = synth.make_frame_arrays(rng, 1200)
    fr["excluded"] = (rng.random(1200) < 0.2).astype(np.uint8)
    F = _frame(fr)
    q = synth.make_window_queries(seed, fr, 1500, jitter=2.0, th=6.0)
    m = S.ORBmatcher(0.75, ori)
    nm, k2q = m.SearchWindowGreedy(F, q, max_dist)
    onm, ok2q = oracle.search_window_greedy(F, q, max_dist, ori)
    assert nm == onm and np.array_equal(k2q, ok2q)
    assert nm > 200
    m.close()


def test_distinctive_descriptors_batch_matches_oracle(S, oracle):
    """MapPoint::ComputeDistinctiveDescriptors over a batch of map points, bit-exact (indices,
    medians), including empty points, single observations, ties and a point with 512 observations."""
    rng = np.random.default_rng(21)
    counts = np.concatenate([[0, 1, 2, 2, 3, 64, 65, 130, 512], rng.integers(1, 40, 3000)])
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    descs = np.zeros((off[-1], 32), np.uint8)
    for p, n in enumerate(counts):
        if n == 0:
            continue
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        flips = rng.uniform(0.0, 0.25, n)
        descs[off[p]:off[p + 1]] = [np.bitwise_xor(base, np.packbits(rng.random(256) < f)) for f in flips]
    descs[off[3]:off[4]] = descs[off[3]]  # two identical observations: a tie, the first wins
    m = S.ORBmatcher()
    idx, med = m.ComputeDistinctiveDescriptors(off, descs)
    oidx, omed = oracle.distinctive_descriptors(off, descs)
    assert np.array_equal(idx, oidx) and np.array_equal(med, omed)
    assert idx[0] == -1 and idx[1] == 0 and idx[3] == 0
    with pytest.raises(S.SwarmOrbError):
        m.ComputeDistinctiveDescriptors([0, 513], np.zeros((513, 32), np.uint8))
    m.close()


@pytest.mark.parametrize("seed", [41, 42, 43])
def test_window_queries_with_distorted_camera_bounds(S, oracle, seed):
    """A distorted camera: mnMinX / mnMinY are negative, the grid cells are not 11.75 px wide and keypoints lie
    outside [0, w) x [0, h).  The kernel scans only the grid columns GetFeaturesInArea visits (and tests the cell
    row): the cell arithmetic has to agree with the reference's for every offset, including windows that leave the
    grid on either side."""
    rng = np.random.default_rng(seed)
    w, h = synth.EUROC
    bounds = (-41.37, w + 36.81, -27.55, h + 31.02)
    fr, mps = synth.make_m1_case(seed, 1500, 3000)
    fr["x"] = rng.uniform(bounds[0] - 1.0, bounds[1] + 1.0, len(fr["x"])).astype(np.float32)  # a few outside the grid
    fr["y"] = rng.uniform(bounds[2] - 1.0, bounds[3] + 1.0, len(fr["y"])).astype(np.float32)
    fr["bounds"] = bounds
    k = rng.integers(0, len(fr["x"]), len(mps["proj_x"]))
    mps["proj_x"] = (fr["x"][k] + rng.normal(0, 2.0, len(k))).astype(np.float32)
    mps["proj_y"] = (fr["y"][k] + rng.normal(0, 2.0, len(k))).astype(np.float32)
    far = rng.random(len(k)) < 0.1                      # windows hanging over / beyond the borders
    mps["proj_x"][far] = rng.choice([bounds[0] - 30, bounds[0] + 1, bounds[1] - 1, bounds[1] + 30], far.sum()).astype(np.float32)
    mps["desc"] = synth.flip_bits(rng, fr["desc"][k], 0.12)
    F = _frame(fr)
    m = S.ORBmatcher(0.8)
    for th in (1.0, 4.0):
        nm, kp_to_mp = m.SearchByProjectionMapPoints(F, mps, th)
        onm, okp = oracle.search_by_projection_mappoints(F, mps, th, 0.8)
        assert nm == onm and np.array_equal(kp_to_mp, okp)
        assert nm > 100
    fr2, last = synth.make_m2_case(seed + 100, 1500, 1500)
    fr2["x"], fr2["y"], fr2["bounds"] = fr["x"], fr["y"], bounds
    k2 = rng.integers(0, len(fr["x"]), len(last["u"]))
    last["u"] = (fr["x"][k2] + rng.normal(0, 3.0, len(k2))).astype(np.float32)
    last["v"] = (fr["y"][k2] + rng.normal(0, 3.0, len(k2))).astype(np.float32)
    last["desc"] = synth.flip_bits(rng, fr2["desc"][k2], 0.1)
    last["octave"] = np.clip(fr2["octave"][k2] + rng.integers(-1, 2, len(k2)), 0, 7).astype(np.int32)
    last["angle"] = ((fr2["angle"][k2] + rng.normal(0, 2, len(k2))) % 360).astype(np.float32)
    F2 = _frame(fr2)
    nm, kp_to_last = m.SearchByProjectionLastFrame(F2, last, 15.0)
    onm, okp = oracle.search_by_projection_lastframe(F2, last, 15.0, True)
    assert nm == onm and np.array_equal(kp_to_last, okp)
    assert nm > 100
    m.close()


def test_reuse_frame_between_the_two_tracking_searches(S, oracle):
    """so_matcher_reuse_frame: the second search of a frame skips the candidate upload; `excluded` is re-read."""
    fr, last = synth.make_m2_case(51, 1200, 1200)
    _, mps = synth.make_m1_case(52, 1200, 2500)
    F = _frame(fr, excluded=False)
    m = S.ORBmatcher(0.9, True)
    nm2, k2l = m.SearchByProjectionLastFrame(F, last, 15.0)
    onm2, ok2l = oracle.search_by_projection_lastframe(F, last, 15.0, True)
    assert nm2 == onm2 and np.array_equal(k2l, ok2l)
    # keypoints bound by the first search are excluded from the second (Tracking: mvpMapPoints[i] with observations)
    k = np.random.default_rng(5).integers(0, F.n, len(mps["proj_x"]))
    mps["proj_x"] = (fr["x"][k] + 1.5).astype(np.float32); mps["proj_y"] = (fr["y"][k] - 1.0).astype(np.float32)
    mps["desc"] = synth.flip_bits(np.random.default_rng(6), fr["desc"][k], 0.1)
    F2 = FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"],
                   (k2l >= 0).astype(np.uint8))
    staged_full = None
    for reuse in (False, True):
        if reuse:
            m.SearchByProjectionLastFrame(F, last, 15.0)   # the frame is resident again
            m.reuse_frame()
        m.mfNNratio = 0.8
        nm1, k2m = m.SearchByProjectionMapPoints(F2, mps, 1.0)
        onm1, ok2m = oracle.search_by_projection_mappoints(F2, mps, 1.0, 0.8)
        assert nm1 == onm1 and np.array_equal(k2m, ok2m) and nm1 > 100
        staged = m.last_stats()["staged_bytes"]
        if not reuse:
            staged_full = staged
        else:
            assert staged < staged_full - 32 * F.n          # descriptors and positions were not sent again
    # a stale request (different frame size) is ignored, not trusted
    fr3, mps3 = synth.make_m1_case(53, 900, 1500)
    F3 = _frame(fr3)
    m.reuse_frame()
    nm, out = m.SearchByProjectionMapPoints(F3, mps3, 1.0)
    onm, oout = oracle.search_by_projection_mappoints(F3, mps3, 1.0, 0.8)
    assert nm == onm and np.array_equal(out, oout)
    m.close()
