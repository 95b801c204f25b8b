"""GPU parity of Fuse x2, SearchBySim3 and the keyframe-side SearchByProjection overloads
against oracle/project_oracle.c, through the C ABI: the projection half bit-exact on (active, u, v, radius, level) -
float work evaluated in the reference's expression order on both sides, tolerance 0 - and the routines' end results
(best keypoint / distance per map point, bindings per keypoint, agreeing pairs) identical."""
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


def _view(fr, excluded=True):
    return FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"],
                     fr.get("excluded") if excluded else None, grid_bounds=fr.get("grid_bounds"))


def _same_queries(q, oq):
    for k in ("active", "u", "v", "radius", "level"):
        assert q[k].tobytes() == oq[k].tobytes(), k


@pytest.mark.parametrize("seed,n_kp,n_mp,th", [(101, 1000, 1200, 3.0), (102, 2000, 4000, 3.0), (103, 300, 5000, 5.0),
                                               (104, 1000, 1, 3.0)])
def test_m6_fuse(S, oracle, seed, n_kp, n_mp, th):
    c = synth.make_projection_case(seed, n_kp, n_mp)
    KF = _view(c["frame"], False)
    cam = oracle.camera(c["cam"])
    m = S.ORBmatcher()
    n, bi, bd, q = m.Fuse(KF, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], th)
    _same_queries(q, oracle.fuse_queries(KF, cam, c["Tcw"], c["log_scale_factor"], c["mp"], th))
    on, obi, obd = oracle.fuse(KF, cam, c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], th)
    assert n == on and np.array_equal(bi, obi) and np.array_equal(bd, obd)
    if n_mp > 100:
        assert n > 0.3 * n_mp * min(1.0, n_kp / 1000) * 0.5
        a = q["active"].astype(bool)
        assert a.sum() > 0.4 * n_mp and (~a).sum() > 0.1 * n_mp
    m.close()


@pytest.mark.parametrize("seed,scale,th", [(111, 1.0, 4.0), (112, 1.37, 4.0), (113, 0.61, 3.0)])
def test_m6_fuse_with_sim3(S, oracle, seed, scale, th):
    c = synth.make_projection_case(seed, 1200, 2500, sim3_scale=scale)
    KF = _view(c["frame"], False)
    cam = oracle.camera(c["cam"])
    m = S.ORBmatcher()
    n, bi, bd, q = m.FuseSim3(KF, c["cam"], c["Scw"], c["log_scale_factor"], c["mp"], th)
    _same_queries(q, oracle.sim3_world_queries(KF, cam, c["Scw"], c["log_scale_factor"], c["mp"], th))
    on, obi, obd = oracle.fuse_sim3(KF, cam, c["Scw"], c["log_scale_factor"], c["mp"], th)
    assert n == on and np.array_equal(bi, obi) and np.array_equal(bd, obd)
    assert n > 500
    m.close()


@pytest.mark.parametrize("seed,n,s12,th", [(121, 900, 1.3, 7.5), (122, 1500, 0.8, 7.5), (123, 400, 1.0, 5.0)])
def test_m7_search_by_sim3(S, oracle, seed, n, s12, th):
    c = synth.make_sim3_pair_case(seed, n, s12=s12)
    K1, K2 = _view(c["frame1"], False), _view(c["frame2"], False)
    cam = oracle.camera(c["cam"])
    lsf = c["log_scale_factor"]
    m = S.ORBmatcher()
    nf, m12, q1, q2 = m.SearchBySim3(K1, K2, c["cam"], c["T1w"], c["T2w"], c["s12"], c["R12"], c["t12"], lsf, lsf,
                                     c["mp1"], c["mp2"], th)
    _same_queries(q1, oracle.sim3_pair_queries(K2, cam, c["T1w"], c["s12"], c["R12"], c["t12"], True, lsf, c["mp1"], th))
    _same_queries(q2, oracle.sim3_pair_queries(K1, cam, c["T2w"], c["s12"], c["R12"], c["t12"], False, lsf, c["mp2"], th))
    onf, om12 = oracle.search_by_sim3(K1, K2, cam, c["T1w"], c["T2w"], c["s12"], c["R12"], c["t12"], lsf, lsf, c["mp1"],
                                      c["mp2"], th)
    assert nf == onf and np.array_equal(m12, om12)
    assert nf > 0.25 * n
    m.close()


@pytest.mark.parametrize("seed,scale,th", [(131, 1.0, 10), (132, 1.9, 10), (133, 0.7, 4)])
def test_m7_search_by_projection_keyframe_scw(S, oracle, seed, scale, th):
    c = synth.make_projection_case(seed, 1000, 2500, sim3_scale=scale, prebound_frac=0.15)
    KF = _view(c["frame"])
    cam = oracle.camera(c["cam"])
    m = S.ORBmatcher()
    nm, k2p, q = m.SearchByProjectionSim3(KF, c["cam"], c["Scw"], c["log_scale_factor"], c["mp"], th)
    _same_queries(q, oracle.sim3_world_queries(KF, cam, c["Scw"], c["log_scale_factor"], c["mp"], float(th)))
    onm, ok2p = oracle.search_by_projection_sim3(KF, cam, c["Scw"], c["log_scale_factor"], c["mp"], th)
    assert nm == onm and np.array_equal(k2p, ok2p)
    assert nm > 300
    m.close()


@pytest.mark.parametrize("seed,th,orb_dist,ori", [(141, 10.0, 100, True), (142, 3.0, 64, True), (143, 10.0, 100, False)])
def test_m7_search_by_projection_frame_keyframe(S, oracle, seed, th, orb_dist, ori):
    c = synth.make_projection_case(seed, 1000, 1000, prebound_frac=0.2)
    F = _view(c["frame"])
    cam = oracle.camera(c["cam"])
    m = S.ORBmatcher(0.9, ori)
    nm, k2p, q = m.SearchByProjectionKeyFrame(F, c["cam"], c["Tcw"], c["log_scale_factor"], c["mp"], c["mp"]["angle"], th,
                                              orb_dist)
    _same_queries(q, oracle.frame_kf_queries(F, cam, c["Tcw"], c["log_scale_factor"], c["mp"], th))
    onm, ok2p = oracle.search_by_projection_frame_kf(F, cam, c["Tcw"], c["log_scale_factor"], c["mp"], c["mp"]["angle"],
                                                     th, orb_dist, ori)
    assert nm == onm and np.array_equal(k2p, ok2p)
    assert nm > 200
    m.close()


def test_projected_searches_edge_cases(S, oracle):
    """No map points, no keypoints, nothing valid, a crowded window that exhausts the K-lists of the greedy search."""
    c = synth.make_projection_case(151, 800, 600)
    KF = _view(c["frame"], False)
    cam = oracle.camera(c["cam"])
    lsf, inv = c["log_scale_factor"], c["inv_level_sigma2"]
    m = S.ORBmatcher()
    empty = {k: v[:0] for k, v in c["mp"].items()}
    n, bi, bd, _ = m.Fuse(KF, c["cam"], c["Tcw"], lsf, inv, empty, 3.0)
    assert n == 0 and len(bi) == 0
    none = dict(c["mp"], valid=np.zeros(600, np.uint8))
    n, bi, bd, q = m.Fuse(KF, c["cam"], c["Tcw"], lsf, inv, none, 3.0)
    assert n == 0 and (bi == -1).all() and (bd == 256).all() and not q["active"].any()
    fr0 = {k: (v[:0] if isinstance(v, np.ndarray) and k != "scale_factors" else v) for k, v in c["frame"].items()}
    n, bi, bd, q = m.Fuse(_view(fr0, False), c["cam"], c["Tcw"], lsf, inv, c["mp"], 3.0)
    on, obi, obd = oracle.fuse(_view(fr0, False), cam, c["Tcw"], lsf, inv, c["mp"], 3.0)
    assert n == on == 0 and np.array_equal(bi, obi) and np.array_equal(bd, obd)
    # crowded: every map point projects onto the same few keypoints, th large -> later points find their K best taken
    c2 = synth.make_projection_case(152, 60, 900, jitter=0.5)
    F2 = _view(c2["frame"])
    nm, k2p, _ = m.SearchByProjectionKeyFrame(F2, c2["cam"], c2["Tcw"], lsf, c2["mp"], c2["mp"]["angle"], 25.0, 100)
    onm, ok2p = oracle.search_by_projection_frame_kf(F2, oracle.camera(c2["cam"]), c2["Tcw"], lsf, c2["mp"],
                                                     c2["mp"]["angle"], 25.0, 100, True)
    assert nm == onm and np.array_equal(k2p, ok2p)
    m.close()


def test_keyframe_int_bounds_quirk(S, oracle):
    """A KeyFrame queries the grid it copied from its Frame (cells assigned with the Frame's float origin) with its own
    int-truncated bounds (code/include/KeyFrame.h:220, code/src/KeyFrame.cc:58-72, 779-818): all five routines with such
    a target, and the quirk must be visible - the same data with one origin for both gives a different answer."""
    m = S.ORBmatcher()
    c = synth.make_projection_case(161, 3000, 6000, keyframe_bounds=True, jitter=3.0, prebound_frac=0.1)
    cam = oracle.camera(c["cam"])
    lsf, inv = c["log_scale_factor"], c["inv_level_sigma2"]
    KF = _view(c["frame"], False)
    assert KF.has_grid_origin == 1 and KF.grid_min_x != KF.min_x
    n, bi, bd, q = m.Fuse(KF, c["cam"], c["Tcw"], lsf, inv, c["mp"], 3.0)
    _same_queries(q, oracle.fuse_queries(KF, cam, c["Tcw"], lsf, c["mp"], 3.0))
    on, obi, obd = oracle.fuse(KF, cam, c["Tcw"], lsf, inv, c["mp"], 3.0)
    assert n == on and np.array_equal(bi, obi) and np.array_equal(bd, obd) and n > 1000
    n, bi, bd, q = m.FuseSim3(KF, c["cam"], c["Scw"], lsf, c["mp"], 4.0)
    on, obi, obd = oracle.fuse_sim3(KF, cam, c["Scw"], lsf, c["mp"], 4.0)
    assert n == on and np.array_equal(bi, obi) and np.array_equal(bd, obd)
    KFx = _view(c["frame"])
    nm, k2p, _ = m.SearchByProjectionSim3(KFx, c["cam"], c["Scw"], lsf, c["mp"], 10)
    onm, ok2p = oracle.search_by_projection_sim3(KFx, cam, c["Scw"], lsf, c["mp"], 10)
    assert nm == onm and np.array_equal(k2p, ok2p)
    # the quirk is observable: with the float origin on both sides some best keypoints differ
    plain = FrameView(c["frame"]["x"], c["frame"]["y"], c["frame"]["octave"], c["frame"]["angle"], c["frame"]["desc"],
                      c["frame"]["grid_bounds"], c["frame"]["scale_factors"])
    _, bi_plain, _ = oracle.fuse_sim3(plain, cam, c["Scw"], lsf, c["mp"], 4.0)
    assert not np.array_equal(bi_plain, obi)
    p = synth.make_sim3_pair_case(162, 1500, keyframe_bounds=True)
    K1, K2 = _view(p["frame1"], False), _view(p["frame2"], False)
    nf, m12, _, _ = m.SearchBySim3(K1, K2, p["cam"], p["T1w"], p["T2w"], p["s12"], p["R12"], p["t12"], lsf, lsf, p["mp1"],
                                   p["mp2"], 7.5)
    onf, om12 = oracle.search_by_sim3(K1, K2, cam, p["T1w"], p["T2w"], p["s12"], p["R12"], p["t12"], lsf, lsf, p["mp1"],
                                      p["mp2"], 7.5)
    assert nf == onf and np.array_equal(m12, om12) and nf > 300
    m.close()


def test_batched_calls_equal_the_same_calls_one_by_one(S, oracle):
    """so_matcher_batch_begin / _end: a new keyframe's SearchForTriangulation + Fuse calls against its neighbours as one
    staging copy, one projection launch and one search launch - every output equal to the call made alone (and so to
    the oracle); any other matcher call inside a batch is refused; more than 64 calls flush and carry on."""
    from swarmmap_amd.matcher import FeatureVector
    m = S.ORBmatcher(0.6, True)
    sf = synth.SCALE_FACTORS
    fuse_cases = [synth.make_projection_case(200 + i, 600 + 150 * i, 500 + 300 * i, keyframe_bounds=(i % 2 == 0)) for i in range(5)]
    sim_cases = [synth.make_projection_case(220 + i, 900, 1100, sim3_scale=1.2 + 0.1 * i) for i in range(2)]
    tri_cases = []
    for i in range(4):
        kf1, node1, kf2, node2, src = synth.make_bow_case(240 + i, 800 + 100 * i, 900, p_flip=0.06)
        rng = np.random.default_rng(i)
        kf1["y"] = (kf2["y"][src] + rng.normal(0, 0.8, len(src))).astype(np.float32)
        kf1["x"] = (kf2["x"][src] + rng.uniform(-30, 30, len(src))).astype(np.float32)
        F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-6, (3, 3)).astype(np.float32)
        tri_cases.append((kf1, FeatureVector(node1), kf2, FeatureVector(node2), F12))
    alone = []
    for c in fuse_cases:
        alone.append(m.Fuse(_view(c["frame"], False), c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0))
    for c in sim_cases:
        alone.append(m.FuseSim3(_view(c["frame"], False), c["cam"], c["Scw"], c["log_scale_factor"], c["mp"], 4.0))
    for t in tri_cases:
        alone.append(m.SearchForTriangulation(t[0], t[1], t[2], t[3], t[4], (900.0, 240.0), sf, sf * sf))
    for rounds in (1, 6):  # 11 calls, then 66: the 65th flushes the first 64
        m.batch_begin()
        held = []
        for _ in range(rounds):
            for c in fuse_cases:
                held.append(m.Fuse(_view(c["frame"], False), c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0))
            for c in sim_cases:
                held.append(m.FuseSim3(_view(c["frame"], False), c["cam"], c["Scw"], c["log_scale_factor"], c["mp"], 4.0))
            for t in tri_cases:
                held.append(m.SearchForTriangulation(t[0], t[1], t[2], t[3], t[4], (900.0, 240.0), sf, sf * sf))
        if rounds == 1:
            with pytest.raises(Exception):  # not batchable
                c = fuse_cases[0]
                m.SearchByProjectionSim3(_view(c["frame"]), c["cam"], c["Scw"], c["log_scale_factor"], c["mp"], 10)
        m.batch_end()
        for k, h in enumerate(held):
            a = alone[k % len(alone)]
            assert h[0].value == a[0] > 0, k
            for x, y in zip(h[1:], a[1:]):
                if isinstance(x, dict):
                    for key in x:
                        assert x[key].tobytes() == y[key].tobytes(), (k, key)
                else:
                    assert np.array_equal(x, y), k
    # the handle works as before afterwards
    c = fuse_cases[1]
    again = m.Fuse(_view(c["frame"], False), c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0)
    assert again[0] == alone[1][0] and np.array_equal(again[1], alone[1][1])
    on, obi, obd = oracle.fuse(_view(c["frame"], False), oracle.camera(c["cam"]), c["Tcw"], c["log_scale_factor"],
                               c["inv_level_sigma2"], c["mp"], 3.0)
    assert again[0] == on and np.array_equal(again[1], obi)
    m.close()


def test_resident_keyframes_give_the_same_answers(S, oracle):
    """so_kframe_create uploads a keyframe once (grid order + vocabulary-node order); so_fuse_kframe /
    so_search_for_triangulation_kframe against it equal so_fuse / so_search_for_triangulation with the same keyframe as a
    host view - alone and inside a batch, with a KeyFrame's int bounds, and with the free mask changing between calls."""
    from swarmmap_amd.matcher import FeatureVector, KFrame
    m = S.ORBmatcher(0.6, True)
    sf = synth.SCALE_FACTORS
    cases = [synth.make_projection_case(300 + i, 700 + 200 * i, 900 + 250 * i, keyframe_bounds=(i != 1)) for i in range(3)]
    kframes, want = [], []
    for c in cases:
        KF = _view(c["frame"], False)
        kframes.append(KFrame(m, KF))
        want.append(m.Fuse(KF, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0))
    for c, k, w in zip(cases, kframes, want):
        got = m.FuseKFrame(k, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0)
        assert got[0] == w[0] > 100 and np.array_equal(got[1], w[1]) and np.array_equal(got[2], w[2])
        _same_queries(got[3], w[3])
    tri = []
    for i in range(3):
        kf1, node1, kf2, node2, src = synth.make_bow_case(340 + i, 900, 800 + 100 * i, p_flip=0.06)
        rng = np.random.default_rng(i)
        kf1["y"] = (kf2["y"][src] + rng.normal(0, 0.8, len(src))).astype(np.float32)
        kf1["x"] = (kf2["x"][src] + rng.uniform(-30, 30, len(src))).astype(np.float32)
        F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-6, (3, 3)).astype(np.float32)
        fv1, fv2 = FeatureVector(node1), FeatureVector(node2)
        bounds = (0.0, float(synth.EUROC[0]), 0.0, float(synth.EUROC[1]))
        V2 = FrameView(kf2["x"], kf2["y"], kf2["octave"], kf2["angle"], kf2["desc"], bounds, sf)
        k2 = KFrame(m, V2, fv2, sf * sf)
        for trial in range(2):  # the free mask is per call
            if trial == 1:
                kf2 = dict(kf2, free=(rng.random(len(kf2["free"])) < 0.6).astype(np.uint8))
            w = m.SearchForTriangulation(kf1, fv1, kf2, fv2, F12, (900.0, 240.0), sf, sf * sf)
            g = m.SearchForTriangulationKFrame(kf1, fv1, k2, kf2["free"], F12, (900.0, 240.0))
            assert g[0] == w[0] > 50 and np.array_equal(g[1], w[1])
            ow = oracle.search_for_triangulation(kf1, fv1, kf2, fv2, F12, (900.0, 240.0), sf, sf * sf, True)
            assert g[0] == ow[0] and np.array_equal(g[1], ow[1])
        tri.append((kf1, fv1, kf2, k2, F12, w))
    m.batch_begin()
    held = [m.FuseKFrame(k, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0) for c, k in zip(cases, kframes)]
    held_t = [m.SearchForTriangulationKFrame(t[0], t[1], t[3], t[2]["free"], t[4], (900.0, 240.0)) for t in tri]
    mixed = m.Fuse(_view(cases[0]["frame"], False), cases[0]["cam"], cases[0]["Tcw"], cases[0]["log_scale_factor"],
                   cases[0]["inv_level_sigma2"], cases[0]["mp"], 3.0)   # staged and resident jobs side by side
    m.batch_end()
    for h, w in zip(held + [mixed], want + [want[0]]):
        assert h[0].value == w[0] and np.array_equal(h[1], w[1]) and np.array_equal(h[2], w[2])
    for h, t in zip(held_t, tri):
        assert h[0].value == t[5][0] and np.array_equal(h[1], t[5][1])
    for k in kframes + [t[3] for t in tri]:
        k.close()
    m.close()


def test_batch_shares_the_map_points_of_consecutive_fuse_calls(S, oracle):
    """A keyframe's map points go into every neighbour's Fuse with the same arrays and a different `valid` mask
    (LocalMapping.cc:451-457): inside a batch such calls refer to ONE staged copy of the arrays.  The answers must be those
    of the calls made alone - with resident keyframes and host views, with a mask per call, and when the caller changes the
    arrays in place between two calls of the batch (same pointers, other bytes: no sharing then)."""
    from swarmmap_amd.matcher import KFrame
    m = S.ORBmatcher(0.6, True)
    base = synth.make_projection_case(400, 900, 1000, keyframe_bounds=True)
    cases = []
    for i in range(4):  # the same keyframe seen from four slightly different poses: every call has matches to lose
        c = dict(base)
        T = np.array(base["Tcw"], np.float32).copy()
        T.reshape(3, 4)[:, 3] += np.float32(0.004 * i)
        c["Tcw"] = T
        cases.append(c)
    mp = {k: np.ascontiguousarray(v).copy() for k, v in cases[0]["mp"].items()}
    rng = np.random.default_rng(7)
    masks = [(rng.random(len(mp["valid"])) < 0.8).astype(np.uint8) for _ in cases]
    views = [_view(c["frame"], False) for c in cases]
    kframes = [KFrame(m, v) for v in views]

    def args(i, pts):
        c = cases[i]
        return (c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], pts, 3.0)

    def with_mask(i):
        mp["valid"][:] = masks[i]  # in place: the pointer stays, and `valid` is per call by contract
        return mp

    alone = [m.Fuse(views[i], *args(i, with_mask(i))) for i in range(4)]
    for i in range(4):
        on, obi, obd = oracle.fuse(views[i], oracle.camera(cases[i]["cam"]), cases[i]["Tcw"], cases[i]["log_scale_factor"],
                                   cases[i]["inv_level_sigma2"], with_mask(i), 3.0)
        assert alone[i][0] == on and np.array_equal(alone[i][1], obi) and np.array_equal(alone[i][2], obd)
    assert min(a[0] for a in alone) > 100 and len({a[1].tobytes() for a in alone}) == 4
    for resident in (True, False):
        m.batch_begin()
        held = []
        for i in range(4):
            pts = with_mask(i)
            held.append(m.FuseKFrame(kframes[i], *args(i, pts)) if resident else m.Fuse(views[i], *args(i, pts)))
        m.batch_end()
        for h, a in zip(held, alone):
            assert h[0].value == a[0] and np.array_equal(h[1], a[1]) and np.array_equal(h[2], a[2])
    # more calls than a batch holds (64): the flush in the middle takes the shared block with it, the 65th call stages anew
    m.batch_begin()
    held = [m.FuseKFrame(kframes[i % 4], *args(i % 4, with_mask(i % 4))) for i in range(70)]
    m.batch_end()
    for i, h in enumerate(held):
        a = alone[i % 4]
        assert h[0].value == a[0] and np.array_equal(h[1], a[1]) and np.array_equal(h[2], a[2]), i
    # the arrays change in place between two calls of one batch
    moved = mp["Xw"].copy()
    moved[::3] += 0.02
    want_moved = None
    keep = mp["Xw"].copy()
    mp["Xw"][:] = moved
    want_moved = m.Fuse(views[1], *args(1, with_mask(1)))
    mp["Xw"][:] = keep
    m.batch_begin()
    h0 = m.FuseKFrame(kframes[0], *args(0, with_mask(0)))
    mp["Xw"][:] = moved
    h1 = m.FuseKFrame(kframes[1], *args(1, with_mask(1)))
    mp["Xw"][:] = keep
    h2 = m.FuseKFrame(kframes[2], *args(2, with_mask(2)))
    m.batch_end()
    assert h0[0].value == alone[0][0] and np.array_equal(h0[1], alone[0][1])
    assert h1[0].value == want_moved[0] and np.array_equal(h1[1], want_moved[1]) and not np.array_equal(h1[1], alone[1][1])
    assert h2[0].value == alone[2][0] and np.array_equal(h2[1], alone[2][1])
    for k in kframes:
        k.close()
    m.close()


def test_fuse_from_the_resident_map_equals_fuse_from_arrays(S, oracle):
    """so_fuse_kframe_map reads the map points' fields and descriptors from a DeviceMap by slot: same answers as
    so_fuse_kframe with the rows' values as arrays (and so as the oracle) - alone, in a batch next to array jobs, with slots
    in any order, with slots that do not exist (inactive points), with the map appended to between calls."""
    from swarmmap_amd.dframe import DeviceMap
    from swarmmap_amd.matcher import KFrame
    m = S.ORBmatcher(0.6, True)
    cases = [synth.make_projection_case(500 + i, 900, 1100, keyframe_bounds=True) for i in range(3)]
    dmap = DeviceMap(0)
    rng = np.random.default_rng(11)
    # the map: junk rows, then every case's points in shuffled order
    junk = 300
    dmap.append(rng.normal(0, 5, (junk, 3)).astype(np.float32), rng.normal(0, 1, (junk, 3)).astype(np.float32),
                np.full(junk, 9.0, np.float32), np.full(junk, 1.0, np.float32), rng.integers(0, 256, (junk, 32)).astype(np.uint8))
    slots, kframes, want = [], [], []
    for c in cases:
        mp = c["mp"]
        n = len(mp["max_dist"])
        order = rng.permutation(n)
        first = dmap.append(np.asarray(mp["Xw"], np.float32).reshape(n, 3)[order], np.asarray(mp["normal"], np.float32).reshape(n, 3)[order],
                            np.asarray(mp["max_dist"], np.float32)[order], np.asarray(mp["min_dist"], np.float32)[order],
                            np.asarray(mp["desc"], np.uint8).reshape(n, 32)[order])
        sl = np.empty(n, np.int32)
        sl[order] = first + np.arange(n, dtype=np.int32)  # point i lives in row sl[i]
        slots.append(sl)
        KF = _view(c["frame"], False)
        kframes.append(KFrame(m, KF))
        want.append(m.FuseKFrame(kframes[-1], c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], mp, 3.0))
        on, obi, obd = oracle.fuse(KF, oracle.camera(c["cam"]), c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], mp, 3.0)
        assert want[-1][0] == on > 100 and np.array_equal(want[-1][1], obi) and np.array_equal(want[-1][2], obd)
    for c, k, sl, w in zip(cases, kframes, slots, want):
        got = m.FuseKFrameMap(k, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], dmap, sl, c["mp"].get("valid"), 3.0)
        assert got[0] == w[0] and np.array_equal(got[1], w[1]) and np.array_equal(got[2], w[2])
        _same_queries(got[3], w[3])
    # slots that are no rows: those points are inactive, the others unchanged
    c, k, sl, w = cases[0], kframes[0], slots[0].copy(), want[0]
    bad = rng.random(len(sl)) < 0.2
    sl[bad] = np.where(rng.random(bad.sum()) < 0.5, -1, len(dmap) + 7)
    v = np.asarray(c["mp"]["valid"], np.uint8).copy() if c["mp"].get("valid") is not None else np.ones(len(sl), np.uint8)
    mp_masked = dict(c["mp"], valid=(v * ~bad).astype(np.uint8))
    ref = m.FuseKFrame(k, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], mp_masked, 3.0)
    got = m.FuseKFrameMap(k, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], dmap, sl, v, 3.0)
    assert got[0] == ref[0] < w[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    assert not got[3]["active"][bad].any()
    # batched, array jobs and map jobs side by side; the map grows in between (appends do not disturb the rows in use)
    m.batch_begin()
    held = []
    for i, (c, k, sl) in enumerate(zip(cases, kframes, slots)):
        held.append(m.FuseKFrameMap(k, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], dmap, sl, c["mp"].get("valid"), 3.0))
        if i == 1:
            held.append(m.FuseKFrame(kframes[0], cases[0]["cam"], cases[0]["Tcw"], cases[0]["log_scale_factor"],
                                     cases[0]["inv_level_sigma2"], cases[0]["mp"], 3.0))
            dmap.append(rng.normal(0, 5, (70000, 3)).astype(np.float32), rng.normal(0, 1, (70000, 3)).astype(np.float32),
                        np.full(70000, 9.0, np.float32), np.full(70000, 1.0, np.float32),
                        rng.integers(0, 256, (70000, 32)).astype(np.uint8))  # past the first 65536 rows: the tables move
    m.batch_end()
    for h, w in zip(held, [want[0], want[1], want[0], want[2]]):
        assert h[0].value == w[0] and np.array_equal(h[1], w[1]) and np.array_equal(h[2], w[2])
    empty = DeviceMap(0)  # a table without rows: nothing to read, every point inactive
    got = m.FuseKFrameMap(kframes[0], cases[0]["cam"], cases[0]["Tcw"], cases[0]["log_scale_factor"], cases[0]["inv_level_sigma2"],
                          empty, slots[0], None, 3.0)
    assert got[0] == 0 and (got[1] == -1).all() and not got[3]["active"].any()
    empty.close()
    for k in kframes:
        k.close()
    dmap.close()
    m.close()


def test_fuse_from_the_map_while_another_thread_appends_to_it(S, oracle):
    """The local-mapping thread reads the map table (so_fuse_kframe_map) while the tracking thread appends to it
    (so_map_write) - from time to time past the table's capacity, which moves the tables.  The rows a search names are not
    rewritten, so every search must return what it returns on a quiet map."""
    import threading
    from swarmmap_amd.dframe import DeviceMap
    from swarmmap_amd.matcher import KFrame
    m = S.ORBmatcher(0.6, True)
    c = synth.make_projection_case(900, 1000, 1500, keyframe_bounds=True)
    mp = c["mp"]
    n = len(mp["max_dist"])
    dmap = DeviceMap(0)
    first = dmap.append(np.asarray(mp["Xw"], np.float32).reshape(n, 3), np.asarray(mp["normal"], np.float32).reshape(n, 3),
                        mp["max_dist"], mp["min_dist"], np.asarray(mp["desc"], np.uint8).reshape(n, 32))
    slots = (first + np.arange(n)).astype(np.int32)
    kf = KFrame(m, _view(c["frame"], False))
    args = (kf, c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], dmap, slots, mp.get("valid"), 3.0)
    want = m.FuseKFrameMap(*args)
    assert want[0] > 100
    stop, errors, appended = threading.Event(), [], [0]

    def writer():
        rng = np.random.default_rng(3)
        try:
            while not stop.is_set() and len(dmap) < 600000:  # 65536 -> 131072 -> ... -> 1048576 rows: four moves
                k = 9000
                dmap.append(rng.normal(0, 5, (k, 3)).astype(np.float32), rng.normal(0, 1, (k, 3)).astype(np.float32),
                            np.full(k, 9.0, np.float32), np.full(k, 1.0, np.float32), rng.integers(0, 256, (k, 32)).astype(np.uint8))
                appended[0] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=writer)
    t.start()
    rounds = 0
    try:
        while t.is_alive() and rounds < 2000:
            m.batch_begin()
            held = [m.FuseKFrameMap(*args) for _ in range(3)]
            m.batch_end()
            for h in held:
                assert h[0].value == want[0] and np.array_equal(h[1], want[1]) and np.array_equal(h[2], want[2])
            rounds += 1
    finally:
        stop.set()
        t.join()
    assert not errors and appended[0] > 20 and rounds > 20
    got = m.FuseKFrameMap(*args)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    kf.close()
    dmap.close()
    m.close()



@pytest.mark.parametrize("check_ori", [True, False])
def test_triangulation_searches_with_device_built_queries(S, oracle, check_ori):
    """so_search_for_triangulation_kframes: CreateNewMapPoints' searches of one new keyframe against its neighbours with both
    sides resident and the queries (free features, vocabulary node, epipolar line) built by the batch's first launch.
    Five neighbours with different sizes, free masks and fundamental matrices, one of them sharing no vocabulary node with
    the keyframe: the same matches as so_search_for_triangulation_kframe neighbour by neighbour, as the host-view routine
    and as the oracle - alone and inside a batch next to a Fuse call."""
    from swarmmap_amd.matcher import FeatureVector, KFrame
    m = S.ORBmatcher(0.6, check_ori)
    sf = synth.SCALE_FACTORS
    bounds = (0.0, float(synth.EUROC[0]), 0.0, float(synth.EUROC[1]))
    kf1, node1, kf2_0, node2_0, src = synth.make_bow_case(510, 950, 900, p_flip=0.06)
    rng = np.random.default_rng(5)
    kf1["y"] = (kf2_0["y"][src] + rng.normal(0, 0.8, len(src))).astype(np.float32)
    kf1["x"] = (kf2_0["x"][src] + rng.uniform(-30, 30, len(src))).astype(np.float32)
    fv1 = FeatureVector(node1)
    k1 = KFrame(m, FrameView(kf1["x"], kf1["y"], np.zeros(len(kf1["x"]), np.int32), kf1["angle"], kf1["desc"], bounds, sf), fv1, sf * sf)
    nbs, want = [], []
    for j in range(5):
        if j == 0:
            kf2, node2 = kf2_0, node2_0
        else:  # the same scene seen again: permuted keypoints, other descriptors' noise, fewer / more features
            keep = rng.permutation(len(kf2_0["x"]))[:700 + 40 * j]
            kf2 = {k: (v[keep] if isinstance(v, np.ndarray) and len(v) == len(kf2_0["x"]) else v) for k, v in kf2_0.items()}
            kf2["desc"] = synth.flip_bits(rng, kf2["desc"], 0.03)
            node2 = node2_0[keep] if j != 3 else node2_0[keep] + 100000  # j = 3: no vocabulary node in common
        kf2 = dict(kf2, free=(rng.random(len(kf2["x"])) < 0.7).astype(np.uint8))
        F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-6, (3, 3)).astype(np.float32)
        fv2 = FeatureVector(node2)
        k2 = KFrame(m, FrameView(kf2["x"], kf2["y"], kf2["octave"], kf2["angle"], kf2["desc"], bounds, sf), fv2, sf * sf)
        epi = (900.0 - 10 * j, 240.0 + j)
        nbs.append((k2, kf2["free"], F12, epi))
        w = m.SearchForTriangulationKFrame(kf1, fv1, k2, kf2["free"], F12, epi)
        o = oracle.search_for_triangulation(kf1, fv1, kf2, fv2, F12, epi, sf, sf * sf, check_ori)
        assert w[0] == o[0] and np.array_equal(w[1], o[1])
        want.append(w)
    assert want[0][0] > 50 and want[3][0] == 0 and sum(w[0] for w in want[1:]) > 100
    got = m.SearchForTriangulationKFrames(k1, kf1["free"], nbs)
    for g, w in zip(got, want):
        assert g[0] == w[0] and np.array_equal(g[1], w[1])
    c = synth.make_projection_case(77, 600, 800, keyframe_bounds=True)
    with m.batch():
        held = m.SearchForTriangulationKFrames(k1, kf1["free"], nbs[:3])
        fz = m.Fuse(_view(c["frame"], False), c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0)
        held2 = m.SearchForTriangulationKFrames(k1, kf1["free"], nbs[3:])
    for g, w in zip(held + held2, want):
        assert g[0].value == w[0] and np.array_equal(g[1], w[1])
    alone = m.Fuse(_view(c["frame"], False), c["cam"], c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0)
    assert fz[0].value == alone[0] and np.array_equal(fz[1], alone[1])
    m.close()
