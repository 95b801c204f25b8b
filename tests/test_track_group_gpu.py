"""The tracking stages of SEVERAL agents as one chain of launches (so_track_group: one search launch with the agent as
blockIdx.y, one resolve launch and one PoseOptimization launch with a workgroup per agent) against the same stages run
agent by agent (so_track_stage_*, themselves pinned to the oracle by tests/test_track_chain_gpu.py): every member gets the
bits of its solo stage - matches, bindings, edge list, pose, outlier flags, LM iteration and trial counts.
Concurrency model: one process per agent (code/Examples/Monocular/swarm_map.cc:329-337) folded into the launches."""
import numpy as np
import pytest

from swarmmap_amd import dframe as dfm
from swarmmap_amd import synth
from test_dframe_gpu import LOG_SF, S, _frame_and_view, _make_map, _pose  # noqa: F401 (S is a fixture)

pytestmark = pytest.mark.gpu

K4 = np.asarray(synth.EUROC_K, np.float32)
INV_SIGMA2 = (1.0 / (synth.SCALE_FACTORS.astype(np.float32) ** 2)).astype(np.float32)
KEYS = ("kp_to_q", "edge_kp", "edge_outlier", "Tcw", "in_view", "nmatches", "n_edges", "n_inliers", "iterations", "trials", "rounds")


class _Agent:
    """One agent's tracked frame: last frame + map + current frame + matcher, all device-resident."""

    def __init__(self, S, oracle, seed, nfeat, dist):
        rng = np.random.default_rng(seed)
        self.ex, self.last, lk, lxy, ld, _ = _frame_and_view(S, oracle, seed, nfeat=nfeat, dist=dist)
        Tl = _pose(rng)
        self.Xw, normal, mx, mn, md = _make_map(rng, lxy, lk, ld, synth.EUROC_K, Tl)
        self.dmap = S.DeviceMap()
        self.dmap.append(self.Xw, normal, mx, mn, md)
        self.cur = S.DeviceFrame(self.ex, synth.EUROC_K, dist)
        self.cur(synth.make_canvas(seed, 752, 480))
        self.slot = np.where(rng.random(len(lk)) < 0.8, np.arange(len(lk)), -1).astype(np.int32)
        self.Tc = Tl.copy()
        self.Tc[3] += 0.004
        self.Tc[7] -= 0.003
        self.m = S.ORBmatcher(0.9, True)

    def stage1(self, wait=True):
        return dfm.track_stage_last_frame(self.m, self.cur, self.last, self.dmap, self.Tc, self.slot, 15.0, K4, INV_SIGMA2, wait=wait)

    def stage2_inputs(self, r1):
        k2l = r1["kp_to_q"]
        bound = np.where(k2l >= 0, self.slot[np.maximum(k2l, 0)], -1).astype(np.int32)
        bound[r1["edge_kp"][r1["edge_outlier"] != 0]] = -1
        skip = np.zeros(len(self.Xw), np.uint8)
        skip[bound[bound >= 0]] = 1
        return bound, skip

    def stage2(self, r1, on_device, wait=True):
        bound, skip = self.stage2_inputs(r1)
        return dfm.track_stage_local_map(self.m, self.cur, bound, self.dmap, r1["Tcw"], len(self.Xw), 1.0, 0.5, LOG_SF, K4, INV_SIGMA2,
                                         skip=skip, kp_slot_is_last_stage=on_device, wait=wait)

    def again(self, wait=True):
        T2 = self.Tc.copy()
        T2[3] -= 0.01
        return dfm.track_stage_pose_again(self.m, self.cur, T2, wait=wait)

    def close(self):
        for h in (self.m, self.dmap, self.cur, self.last, self.ex):
            h.close()


def _same(a, b, what):
    assert a is not None and b is not None, what
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), "%s: %s differs" % (what, k)


def test_grouped_stages_equal_the_members_solo_stages(S, oracle):
    """Four agents with different frames, keypoint counts (1000 / 700 / 1000 / 500 features) and lens models: last-frame stage,
    the local-map stage behind it on the bindings it left on the device, and the repeated pose - grouped == solo, bit by bit."""
    agents = [_Agent(S, oracle, seed, nfeat, dist) for seed, nfeat, dist in
              ((61, 1000, synth.EUROC_DIST), (62, 700, (0, 0, 0, 0)), (63, 1000, synth.EUROC_DIST), (64, 500, synth.EUROC_DIST))]
    solo1 = [a.stage1() for a in agents]
    solo2 = [a.stage2(r, True) for a, r in zip(agents, solo1)]
    solo3 = [a.again() for a in agents]
    assert all(r is not None and r["nmatches"] > 150 for r in solo1) and all(r["n_edges"] > 200 for r in solo2)
    g = dfm.TrackGroup([a.m for a in agents])
    for rep in range(2):  # (twice: the tables alternate between two blocks)
        waits = [a.stage1(wait=False) for a in agents]
        assert g.pending() == len(agents)
        g.launch()
        assert g.pending() == 0
        grp1 = [w() for w in waits]
        search_ms, pose_ms = g.last_kernel_ms()
        assert 0 < search_ms < 5 and 0 < pose_ms < 5
        for i, (a, b) in enumerate(zip(grp1, solo1)):
            _same(a, b, "agent %d, last-frame stage" % i)
        waits = [a.stage2(r, True, wait=False) for a, r in zip(agents, grp1)]
        g.launch()
        grp2 = [w() for w in waits]
        for i, (a, b) in enumerate(zip(grp2, solo2)):
            _same(a, b, "agent %d, local-map stage" % i)
        waits = [a.again(wait=False) for a in agents]
        assert g.pending() == len(agents)
        g.launch()
        grp3 = [w() for w in waits]
        for i, (a, b) in enumerate(zip(grp3, solo3)):
            for k in ("edge_kp", "edge_outlier", "Tcw", "n_inliers", "iterations", "trials"):
                assert np.array_equal(a[k], b[k]), "agent %d, pose again: %s" % (i, k)
    # a member that leaves the group launches its own stages again
    g.close()
    _same(agents[1].stage1(), solo1[1], "after leaving the group")
    for a in agents:
        a.close()


def test_a_group_of_one_and_a_launch_of_mixed_kinds(S, oracle):
    """One member: the grouped kernels with a grid of one; a launch whose rows are stages of different kinds is refused and
    leaves the members usable."""
    a, b = _Agent(S, oracle, 71, 1000, synth.EUROC_DIST), _Agent(S, oracle, 72, 800, synth.EUROC_DIST)
    s1a, s1b = a.stage1(), b.stage1()
    g = dfm.TrackGroup([a.m])
    w = a.stage1(wait=False)
    g.launch()
    _same(w(), s1a, "group of one")
    g.close()
    g = dfm.TrackGroup([a.m, b.m])
    a.stage1(wait=False)   # a last-frame stage ...
    b.again(wait=False)    # ... and a repeated pose in one launch
    with pytest.raises(S.SwarmOrbError):
        g.launch()
    g.close()
    _same(a.stage1(), s1a, "after a refused launch")
    _same(b.stage1(), s1b, "after a refused launch")
    a.close()
    b.close()


def test_a_member_that_is_handed_back_does_not_disturb_the_others(S, oracle):
    """Three members run the local-map stage on an empty set of bindings; the middle one's map is forty copies of fifty points
    (its K-lists run out: the stage is handed back, SO_RETRY_ON_HOST at its wait) - the other two get their solo results, and the
    handed-back member's plain search afterwards is the oracle's."""
    from swarmmap_amd.matcher import FrameView
    from test_dframe_gpu import _frame_and_view as fv
    agents = []
    for seed, crowded in ((81, False), (77, True), (83, False)):
        rng = np.random.default_rng(seed)
        ex, cur, ck, cxy, cd, F = fv(S, oracle, seed)
        Tc = _pose(rng)
        Xw, normal, mx, mn, md = _make_map(rng, cxy, ck, cd, synth.EUROC_K, Tc, n_extra=0)
        if crowded:
            rep = np.repeat(rng.permutation(len(ck))[:50], 40)
            Xw = (Xw[rep] + rng.normal(0, 0.002, (len(rep), 3))).astype(np.float32)
            normal, mx, mn, md = normal[rep], mx[rep], mn[rep], synth.flip_bits(rng, md[rep], 0.02)
        dmap = S.DeviceMap()
        dmap.append(Xw, normal, mx, mn, md)
        agents.append(dict(ex=ex, cur=cur, F=F, Tc=Tc, Xw=Xw, normal=normal, mx=mx, mn=mn, md=md, dmap=dmap, m=S.ORBmatcher(0.8, True),
                           kp_slot=np.full(len(ck), -1, np.int32), th=6.0 if crowded else 1.0))

    def stage(a, wait=True):
        return dfm.track_stage_local_map(a["m"], a["cur"], a["kp_slot"], a["dmap"], a["Tc"], len(a["Xw"]), a["th"], 0.5, LOG_SF, K4, INV_SIGMA2, wait=wait)

    solo = [stage(a) for a in agents]
    assert solo[0] is not None and solo[2] is not None
    g = dfm.TrackGroup([a["m"] for a in agents])
    waits = [stage(a, wait=False) for a in agents]
    g.launch()
    grp = [w() for w in waits]
    g.close()
    for i in (0, 2):
        _same(grp[i], solo[i], "member %d beside a handed-back one" % i)
    assert (grp[1] is None) == (solo[1] is None)
    if grp[1] is not None:
        _same(grp[1], solo[1], "the crowded member")
    a = agents[1]
    cam = oracle.camera(synth.EUROC_K, synth.EUROC_DIST)
    fr = oracle.is_in_frustum(cam, a["cur"].bounds, a["Tc"], a["Xw"], a["normal"], a["mx"], a["mn"], 0.5, LOG_SF, 8)
    mps = dict(in_view=fr["in_view"], proj_x=fr["proj_x"], proj_y=fr["proj_y"], view_cos=fr["view_cos"], pred_level=fr["pred_level"],
               desc=a["md"], has_obs=np.ones(len(a["Xw"]), np.uint8))
    onm, ok2m = oracle.search_by_projection_mappoints(a["F"], mps, 6.0, 0.8)
    nm, k2m, _ = dfm.search_local_map(a["m"], a["cur"], a["dmap"], a["Tc"], len(a["Xw"]), 6.0, 0.5, LOG_SF)
    assert nm == onm and np.array_equal(k2m, ok2m)
    for a in agents:
        for h in (a["m"], a["dmap"], a["cur"], a["ex"]):
            h.close()
