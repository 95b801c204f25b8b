"""CPU checks of oracle/project_oracle.c (rows M6 / M7): the restated projection + gating statements against an
independent float64 numpy evaluation of the same geometry, and the five routines end to end on synthetic keyframes
(the searches must actually find the planted correspondences).  The oracle is test infrastructure."""
import numpy as np

from oracle import oracle_py
from swarmmap_amd import synth
from swarmmap_amd.matcher import FrameView


def _view(fr, excluded=None):
    return FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"],
                     fr.get("excluded") if excluded is None else excluded)


def _gates64(c, T, frame_form=False, angle_gate=True, pair=None):
    """float64 evaluation of the gates; returns (active, u, v, level) with a margin mask for borderline points."""
    mp, fr = c["mp"], c["frame"]
    fx, fy, cx, cy = c["cam"]
    T = np.asarray(T, np.float64).reshape(3, 4)
    R, t = T[:, :3], T[:, 3]
    X = mp["Xw"].astype(np.float64)
    Pc = X @ R.T + t
    Ow = -R.T @ t
    z = Pc[:, 2]
    with np.errstate(all="ignore"):
        u = fx * Pc[:, 0] / z + cx
        v = fy * Pc[:, 1] / z + cy
        PO = X - Ow
        d = np.linalg.norm(PO, axis=1)
        mx, mn = mp["max_dist"].astype(np.float64), mp["min_dist"].astype(np.float64)
        ok = mp["valid"].astype(bool)
        margin = np.zeros(len(X), bool)
        b = fr["bounds"]
        if not frame_form:
            ok &= z >= 0
        ok &= (u >= b[0]) & (u <= b[1]) & (v >= b[2]) & (v <= b[3])
        margin |= (np.abs(u - b[0]) < 1e-2) | (np.abs(u - b[1]) < 1e-2) | (np.abs(v - b[2]) < 1e-2) | (np.abs(v - b[3]) < 1e-2)
        ok &= (d >= 0.8 * mn) & (d <= 1.2 * mx)
        margin |= (np.abs(d - 0.8 * mn) < 1e-4 * d) | (np.abs(d - 1.2 * mx) < 1e-4 * d)
        if angle_gate:
            dot = np.einsum("ij,ij->i", PO, mp["normal"].astype(np.float64))
            ok &= dot >= 0.5 * d
            margin |= np.abs(dot - 0.5 * d) < 1e-4 * d
        lv = np.log(mx / d) / np.log(1.2)
        level = np.clip(np.ceil(lv), 0, 7)
        margin |= np.abs(lv - np.round(lv)) < 1e-3
        margin |= np.abs(z) < 1e-3
    return ok, u, v, level, margin


def test_fuse_queries_follow_the_geometry():
    c = synth.make_projection_case(11, 900, 1500)
    KF = _view(c["frame"])
    cam = oracle_py.camera(c["cam"])
    q = oracle_py.fuse_queries(KF, cam, c["Tcw"], c["log_scale_factor"], c["mp"], 3.0)
    ok, u, v, level, margin = _gates64(c, c["Tcw"])
    sel = ~margin
    assert np.array_equal(q["active"][sel].astype(bool), ok[sel])
    a = q["active"].astype(bool) & sel
    assert a.sum() > 600 and (~q["active"].astype(bool)).sum() > 150  # both outcomes well represented
    assert np.abs(q["u"][a] - u[a]).max() < 2e-2 and np.abs(q["v"][a] - v[a]).max() < 2e-2
    assert np.array_equal(q["level"][a], level[a].astype(np.int32))
    assert np.array_equal(q["radius"][a], (np.float32(3.0) * synth.SCALE_FACTORS[q["level"][a]]).astype(np.float32))


def test_sim3_world_queries_do_not_depend_on_the_scale_of_scw():
    """Scw = [s R | s t] decomposes into the same pose for every s: the queries of a scaled Scw equal those of the
    plain pose up to float rounding of the decomposition (levels and gates identical away from the borders)."""
    c1 = synth.make_projection_case(12, 900, 1200, sim3_scale=None)
    c2 = synth.make_projection_case(12, 900, 1200, sim3_scale=2.0)  # power of two: the division is exact
    KF = _view(c1["frame"])
    cam = oracle_py.camera(c1["cam"])
    q1 = oracle_py.sim3_world_queries(KF, cam, c1["Scw"], c1["log_scale_factor"], c1["mp"], 4.0)
    q2 = oracle_py.sim3_world_queries(KF, cam, c2["Scw"], c2["log_scale_factor"], c2["mp"], 4.0)
    for k in q1:
        assert np.array_equal(q1[k], q2[k]), k
    # ... and equal Fuse's projection half for s = 1 (the statements are the same, ORBmatcher.cc:776-815 vs :923-964)
    q3 = oracle_py.fuse_queries(KF, cam, c1["Tcw"], c1["log_scale_factor"], c1["mp"], 4.0)
    R, t, Ow = oracle_py.sim3_decompose(c1["Scw"])
    if np.array_equal(np.hstack([R, t[:, None]]).reshape(12), c1["Tcw"]):
        for k in q1:
            assert np.array_equal(q1[k], q3[k]), k


def test_fuse_finds_the_planted_keypoints():
    c = synth.make_projection_case(13, 1000, 1200, jitter=1.0)
    KF = _view(c["frame"])
    cam = oracle_py.camera(c["cam"])
    n, bi, bd = oracle_py.fuse(KF, cam, c["Tcw"], c["log_scale_factor"], c["inv_level_sigma2"], c["mp"], 3.0)
    assert n == (bi >= 0).sum() and n > 500
    assert (bd[bi >= 0] <= 50).all()
    n2, bi2, bd2 = oracle_py.fuse_sim3(KF, cam, c["Scw"], c["log_scale_factor"], c["mp"], 4.0)
    assert n2 >= n  # wider window, no chi2 gate


def test_search_by_sim3_recovers_the_shuffle():
    c = synth.make_sim3_pair_case(14, 800)
    K1, K2 = _view(c["frame1"]), _view(c["frame2"])
    cam = oracle_py.camera(c["cam"])
    n, m12 = oracle_py.search_by_sim3(K1, K2, cam, c["T1w"], c["T2w"], c["s12"], c["R12"], c["t12"],
                                      c["log_scale_factor"], c["log_scale_factor"], c["mp1"], c["mp2"], 7.5)
    inv = np.empty(800, np.int64)
    inv[c["perm"]] = np.arange(800)  # point i of keyframe 1 is keypoint inv[i] of keyframe 2
    found = m12 >= 0
    assert n == found.sum() and n > 250
    assert (m12[found] == inv[found]).mean() > 0.97


def test_greedy_overloads_bind_each_keypoint_once():
    c = synth.make_projection_case(15, 1000, 1500, sim3_scale=1.7, prebound_frac=0.1)
    KF = _view(c["frame"])
    cam = oracle_py.camera(c["cam"])
    n, k2p = oracle_py.search_by_projection_sim3(KF, cam, c["Scw"], c["log_scale_factor"], c["mp"], 10)
    b = k2p[k2p >= 0]
    assert n == len(b) > 300 and len(np.unique(b)) == len(b)
    assert not (c["frame"]["excluded"].astype(bool) & (k2p >= 0)).any()
    n2, k2p2 = oracle_py.search_by_projection_frame_kf(KF, cam, c["Tcw"], c["log_scale_factor"], c["mp"],
                                                       c["mp"]["angle"], 10.0, 100, True)
    b2 = k2p2[k2p2 >= 0]
    assert n2 == len(b2) > 300 and len(np.unique(b2)) == len(b2)
