"""The closed tracking + local-mapping loop (swarmmap_amd/closedloop.py) over the CPU oracle: the host logic - the map
model, the order of LocalMapping::Run, the window gather and the write-back - without a GPU."""
import numpy as np

from swarmmap_amd import closedloop, minitrack, synth
from swarmmap_amd.replay import make_vocabulary
from trajectory_common import OracleBackend

PLANE_Z = 2.0


def _run(n, **kw):
    K = synth.EUROC_K
    st = synth.FrameStream(seed=20221001, size=synth.EUROC, K=K, dist=synth.EUROC_DIST)
    out = closedloop.track(OracleBackend(K, 1000, synth.EUROC_DIST), st, n, K, make_vocabulary(), plane_z=PLANE_Z, **kw)
    return st, out


def test_closed_loop_over_the_oracle_tracks_and_keeps_its_map_consistent():
    n = 42
    st, o = _run(n)
    K = synth.EUROC_K
    gt = minitrack.ground_truth(st, n, K, PLANE_Z)
    px = PLANE_Z / float(K[0])
    # trajectory: tracked online, and re-expressed through the final keyframe poses (System::SaveTrajectoryTUM)
    assert minitrack.ate_rmse(o["centres"], gt, align=False) < px
    assert minitrack.ate_rmse(o["final_centres"], gt, with_scale=True) < px
    assert o["inliers"][1:].min() > 300
    lm = {k: o["lm_log"][:, i] for i, k in enumerate(closedloop.LM_LOG_COLUMNS)}
    assert list(lm["t"]) == list(range(0, n, 5)) and list(lm["neighbours"]) == list(range(len(lm["t"])))
    assert lm["new_points"][2:].min() > 20 and lm["fused"][2:].min() > 0 and lm["lba_edges"][2:].min() > 1000
    assert list(lm["lba_free"][2:]) == list(range(2, len(lm["t"])))  # every keyframe but the first is free in its own window
    # the map's two-way bookkeeping: every binding of a keyframe is an observation of a live point and vice versa
    M = o["map"]
    for kf in M.kfs:
        for i in np.nonzero(kf["mp"] >= 0)[0]:
            s = int(kf["mp"][i])
            assert not M.bad[s] and (kf["id"], int(i)) in M.obs[s]
    n_obs = n_stray = 0
    for s in range(len(M)):
        if M.bad[s]:
            assert M.obs[s] == []
        for kf, i in M.obs[s]:
            # Two features of a new keyframe matched to ONE keypoint of a neighbour both create their point (LocalMapping.cc:403-416
            # has no test for it): the keypoint's binding is the later point's, both keep the observation, and when one of the two
            # goes bad the binding goes with it (MapPoint::SetBadFlag -> EraseMapPointMatch(idx)).  Rare, and the reference's own.
            n_obs += 1
            n_stray += int(M.kfs[kf]["mp"][i]) != s
        assert len({kf for kf, _ in M.obs[s]}) == len(M.obs[s])  # one observation per keyframe
    assert n_stray <= 0.01 * n_obs, (n_stray, n_obs)
    # the points tracking sees arrive `delay` frames after their keyframe
    assert o["n_map_points"][14] == o["n_map_points"][10] and o["n_map_points"][15] > o["n_map_points"][14]


def test_local_window_follows_the_reference_rules():
    _, o = _run(27)
    M = o["map"]
    c = M.kfs[-1]
    prob, win, pts, (e_kf, e_idx, e_pt) = closedloop.local_window(M, c, n_free=3, n_fixed=2)
    free = [kf for kf, f in zip(win, prob["fixed"]) if not f]
    assert c["id"] in free and len(free) <= 3 and 0 not in free and len(win) <= 5 and win == sorted(win)
    # edges: every observation of a window point by a window keyframe, points with fewer than two edges left out
    assert np.bincount(e_pt).min() >= 2
    for e in range(0, len(e_kf), 97):
        s = int(pts[e_pt[e]])
        assert (int(e_kf[e]), int(e_idx[e])) in M.obs[s]
        assert prob["obs"][e, 0] == M.kfs[e_kf[e]]["x"][e_idx[e]]


def test_batched_searches_equal_the_reference_interleaving():
    """LocalMapping::CreateNewMapPoints searches, triangulates and binds neighbour by neighbour (code/src/LocalMapping.cc:219-416),
    SearchInNeighbors fuses target by target with Replace / AddObservation visible to the next target (:451-481).  The product's
    loop issues a keyframe's searches as batches against ONE snapshot and walks the results in the reference's order with the
    gates on the live state (swarmmap_amd/closedloop.py: why that is the same computation).  Here both forms run over the CPU
    oracle - `interleaved` is the literal one: every search sees the map as the applies before it left it - and must agree in
    everything: poses, every row of the local-mapping log, every keyframe's final bindings, every point's flags and position."""
    n = 62
    _, a = _run(n, third_pose=True)
    _, b = _run(n, third_pose=True, interleaved=True)
    for k in ("poses", "kf_poses", "Tcr", "lm_log", "matches_last", "matches_map", "inliers", "n_map_points"):
        assert np.array_equal(a[k], b[k]), k
    Ma, Mb = a["map"], b["map"]
    assert len(Ma) == len(Mb) and np.array_equal(Ma.bad, Mb.bad) and np.array_equal(Ma.repl, Mb.repl) and np.array_equal(Ma.X, Mb.X)
    assert all(np.array_equal(x["mp"], y["mp"]) for x, y in zip(Ma.kfs, Mb.kfs))
    assert all(x == y for x, y in zip(Ma.obs, Mb.obs))
    lm = {k: a["lm_log"][:, i] for i, k in enumerate(closedloop.LM_LOG_COLUMNS)}
    assert lm["tri_matches"][2:].min() > 100 and lm["fused"][3:].min() > 50 and lm["fused_back"][3:].max() > 0  # (the steps did something)
