"""The closed tracking + local-mapping loop: the HIP operators and the CPU oracle chained by the same host logic must
build the same map and the same trajectory from the same images, keyframe by keyframe (BASELINE.json's metric:
tracking + local BA, ATE)."""
import numpy as np
import pytest

from swarmmap_amd import closedloop, minitrack, synth
from swarmmap_amd.replay import make_vocabulary
from trajectory_common import OracleBackend

pytestmark = pytest.mark.gpu

PLANE_Z = 2.0
COUNT_COLUMNS = [closedloop.LM_LOG_COLUMNS.index(k) for k in ("tri_matches", "new_points", "fused", "fused_back", "lba_edges",
                                                               "lba_outliers", "lba_points", "bad_points")]


def _same_jobs(a, b):
    """Local-mapping logs of two chains: same keyframes, same neighbours, same windows; counts may differ by the rare
    decision that flips on the 2e-5 pose / point difference between the two operator sets."""
    assert a.shape == b.shape and np.array_equal(a[:, :2], b[:, :2]), (a, b)
    for col in ("lba_free", "lba_fixed"):
        i = closedloop.LM_LOG_COLUMNS.index(col)
        assert np.array_equal(a[:, i], b[:, i]), (col, a[:, i], b[:, i])
    d = np.abs(a[:, COUNT_COLUMNS] - b[:, COUNT_COLUMNS])
    assert np.all(d <= 4 + b[:, COUNT_COLUMNS] // 50), (a, b)


def test_python_chain_hip_and_oracle_agree_keyframe_by_keyframe():
    n, K = 47, synth.EUROC_K
    st = synth.FrameStream(seed=20221001, size=synth.EUROC, K=K, dist=synth.EUROC_DIST)
    frames = [st.frame(t) for t in range(n)]
    vocab = make_vocabulary()
    hip = minitrack.HipBackend(K, 1000, dist=synth.EUROC_DIST)
    a = closedloop.track(hip, None, n, K, vocab, plane_z=PLANE_Z, third_pose=True, frames=frames)
    hip.close()
    b = closedloop.track(OracleBackend(K, 1000, synth.EUROC_DIST), None, n, K, vocab, plane_z=PLANE_Z, third_pose=True, frames=frames)
    _same_jobs(a["lm_log"], b["lm_log"])
    assert a["lm_log"][2:, closedloop.LM_LOG_COLUMNS.index("lba_edges")].min() > 1000
    assert np.abs(a["poses"] - b["poses"]).max() < 5e-5
    assert np.abs(a["kf_poses"] - b["kf_poses"]).max() < 5e-5
    assert minitrack.ate_rmse(a["centres"], b["centres"], align=False) < 1e-4
    for k in ("matches_last", "matches_map", "inliers"):
        assert np.abs(a[k].astype(int) - b[k].astype(int)).max() <= 5, (k, a[k], b[k])
    assert np.abs(a["n_map_points"].astype(int) - b["n_map_points"].astype(int)).max() <= 8
    gt = minitrack.ground_truth(st, n, K, PLANE_Z)
    px = PLANE_Z / float(K[0])
    assert minitrack.ate_rmse(a["centres"], gt, align=False) < px
    assert abs(minitrack.ate_rmse(a["final_centres"], gt, with_scale=True) - minitrack.ate_rmse(b["final_centres"], gt, with_scale=True)) < 0.02 * px


def _cpp_chain(n, frames_ptrs, st, K, dist, nfeat, vocab, policy=0, track_chain=None, **kw):
    from swarmmap_amd.replay import Replay
    rp = Replay(0, st.w, st.h, nfeat, 5, K, dist, plane_z=PLANE_Z, local_keyframes=12, third_pose=True)
    rp.set_frames(frames_ptrs, on_device=False)
    if track_chain is not None:
        rp.set_track_chain(track_chain)
    rp.set_vocabulary(vocab)
    rp.set_closed_loop(policy=policy, **kw)
    rp.prime(0)
    rp.run(0, n, True)
    rp.drain()
    rp.finish()
    a, cl, stats, lm_stats = rp.log(), rp.closed_loop_log(), rp.stats(), rp.lm_stats()
    rp.close()
    a.update(cl)
    return a, stats, lm_stats


@pytest.mark.parametrize("name", ["euroc", "kitti"])
def test_cpp_closed_loop_in_the_bench_configuration_matches_the_oracle_chain(name):
    """What bench.py times (swarmmap_amd/host/replay.cc + closedloop.cc: tracking thread and local-mapping thread on one
    GPU, frames in pinned host memory, the local map = the last 12 keyframes' points, three PoseOptimization calls per
    frame; per keyframe: SearchForTriangulation batch -> triangulation -> new points; Fuse batch over resident keyframes
    and the resident map -> AddObservation / Replace; local BA over the keyframe's own window -> SetPose / SetWorldPos /
    EraseObservation / UpdateNormalAndDepth; results in the tracked map five frames later) against the same loop over
    the CPU oracle: frame by frame (poses, counts), keyframe by keyframe (window sizes, new / fused / bad point counts,
    final keyframe poses), and the trajectory both ways (online and through the final keyframe poses)."""
    import torch
    euroc = name == "euroc"
    size = synth.EUROC if euroc else synth.KITTI
    K = synth.EUROC_K if euroc else synth.KITTI_K
    dist = synth.EUROC_DIST if euroc else None
    nfeat = 1000 if euroc else 2000
    n = 62 if euroc else 37
    st = synth.FrameStream(seed=20221001, size=size, K=K, dist=dist)
    block = torch.empty((n + 2, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(n + 2):
        view[t] = st.frame(t)
    frames = [view[t] for t in range(n + 2)]
    vocab = make_vocabulary()
    a, stats, lm_stats = _cpp_chain(n, [block.data_ptr() + i * st.w * st.h for i in range(n + 2)], st, K, dist, nfeat, vocab)
    b = closedloop.track(OracleBackend(K, nfeat, dist if dist is not None else (0, 0, 0, 0, 0)), None, n, K, vocab, plane_z=PLANE_Z,
                         third_pose=True, frames=frames)
    assert len(a["poses"]) == n and a["counts"]["jobs"] == len(b["lm_log"]) == (n + 4) // 5
    _same_jobs(a["lm_log"], b["lm_log"])
    assert a["counts"]["windows"] == len(b["lm_log"]) - 2 and a["counts"]["windows_aborted"] == 0
    assert np.array_equal(a["kf_t"], b["kf_t"]) and np.array_equal(a["ref_kf"], b["ref_kf"])
    assert np.abs(a["poses"] - b["poses"]).max() < 5e-5
    assert np.abs(a["kf_poses"] - b["kf_poses"]).max() < 5e-5
    assert np.abs(a["Tcr"] - b["Tcr"]).max() < 5e-5
    assert minitrack.ate_rmse(a["centres"], b["centres"], align=False) < 1e-4
    assert minitrack.ate_rmse(a["final_centres"], b["final_centres"], align=False) < 1e-4
    for k in ("matches_last", "matches_map", "inliers"):
        assert np.abs(a[k].astype(int) - b[k].astype(int)).max() <= 5, (k, a[k], b[k])
    assert np.abs(a["n_map_points"].astype(int) - b["n_map_points"].astype(int)).max() <= 8
    assert a["inliers"][1:].min() > 300
    # the maps themselves: every keyframe's keypoint -> map point bindings as local mapping left them (tracked bindings,
    # triangulated points, Fuse's AddObservation / Replace, local BA's erased observations) and the points' bad / replaced
    # state - slot by slot (a decision that flips on the last bits of a pose would show up here first; none does on these
    # streams, and the bound leaves room for one)
    M = b["map"]
    assert len(a["kf_bindings"]) == len(M.kfs)
    n_bind = n_diff = 0
    for ka, kb in zip(a["kf_bindings"], M.kfs):
        assert len(ka) == len(kb["mp"])
        n_bind += int((kb["mp"] >= 0).sum())
        n_diff += int((ka != kb["mp"]).sum())
    assert n_bind > 5000 and n_diff <= n_bind // 500, (n_diff, n_bind)
    m = min(len(a["point_bad"]), len(M.bad))
    assert abs(len(a["point_bad"]) - len(M.bad)) <= 8 and int((a["point_bad"][:m] != M.bad[:m]).sum()) <= 8
    gt = minitrack.ground_truth(st, n, K, PLANE_Z)
    px = PLANE_Z / float(K[0])
    # online (poses as they were tracked; the first windows have only keyframe 0 fixed - KeyFrame::isFirst() - and shift the
    # young map along its scale gauge before later windows pull it back: a transient of a few pixels) and through the final
    # keyframe poses (what System::SaveTrajectoryTUM writes at shutdown)
    assert minitrack.ate_rmse(a["centres"], gt, with_scale=True) < 2 * px
    assert minitrack.ate_rmse(a["final_centres"], gt, with_scale=True) < px
    assert lm_stats["jobs"] == a["counts"]["jobs"] and lm_stats["cl_solver_ms"] > 0 and lm_stats["batch_kernel_ms"] > 0


def test_cpp_closed_loop_under_the_reference_policy_tracks_and_reports_its_interrupts():
    """policy 1 (Tracking.cc:810-892, LocalMapping.cc:581-583): results arrive when they are ready, a keyframe is made only
    while local mapping is idle, a busy local mapper gets InterruptBA.  Timing-dependent by construction: the checks are
    the trajectory against ground truth and the consistency of the counters."""
    import torch
    K, dist, nfeat, n = synth.EUROC_K, synth.EUROC_DIST, 1000, 80
    st = synth.FrameStream(seed=20221001, size=synth.EUROC, K=K, dist=dist)
    block = torch.empty((n + 2, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(n + 2):
        view[t] = st.frame(t)
    a, stats, _ = _cpp_chain(n, [block.data_ptr() + i * st.w * st.h for i in range(n + 2)], st, K, dist, nfeat, make_vocabulary(), policy=1,
                             kf_every=2)
    c = a["counts"]
    assert c["keyframes"] == c["jobs"] >= 5 and c["windows_aborted"] <= c["windows"] <= c["jobs"] and c["windows_aborted"] <= c["interrupt_ba"]
    # a keyframe every kf_every frames - or earlier when the frame's inliers fall below keyframe_ratio x the last keyframe's
    # (NeedNewKeyFrame's c2), which a map thinned by an interrupted local mapper can trigger on the very next frame
    assert np.all(np.diff(a["kf_t"]) >= 1) and np.median(np.diff(a["kf_t"])) >= 2
    gt = minitrack.ground_truth(st, n, K, PLANE_Z)
    px = PLANE_Z / float(K[0])
    assert a["inliers"][1:].min() > 250
    assert minitrack.ate_rmse(a["centres"], gt, align=False) < 1.5 * px


def test_three_closed_loop_agents_in_threads_equal_their_solo_runs():
    """Agents sharing a GPU (bench.py --agents-per-gpu: a tracking thread + a local-mapping thread each, all in one process):
    every agent's trajectory, local-mapping log and final bindings are those of the same agent run alone - to the bit, the
    C++ loop being deterministic under the deterministic schedule."""
    import threading

    import torch
    K, dist, nfeat, n = synth.EUROC_K, synth.EUROC_DIST, 1000, 37
    vocab = make_vocabulary()
    streams, blocks = [], []
    for a in range(3):
        st = synth.FrameStream(seed=20221001 + 97 * a, size=synth.EUROC, K=K, dist=dist)
        block = torch.empty((n + 2, st.h, st.w), dtype=torch.uint8).pin_memory()
        view = block.numpy()
        for t in range(n + 2):
            view[t] = st.frame(t)
        streams.append(st); blocks.append(block)

    def run(a, out):
        st, block = streams[a], blocks[a]
        out[a] = _cpp_chain(n, [block.data_ptr() + i * st.w * st.h for i in range(n + 2)], st, K, dist, nfeat, vocab)[0]

    solo, par, errs = {}, {}, []
    for a in range(3):
        run(a, solo)

    def guarded(a):
        try:
            run(a, par)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=guarded, args=(a,)) for a in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    for a in range(3):
        s, p = solo[a], par[a]
        assert np.array_equal(s["poses"], p["poses"]) and np.array_equal(s["kf_poses"], p["kf_poses"]), a
        assert np.array_equal(s["lm_log"], p["lm_log"]) and np.array_equal(s["inliers"], p["inliers"]), a
        assert all(np.array_equal(x, y) for x, y in zip(s["kf_bindings"], p["kf_bindings"])), a
        assert s["inliers"][1:].min() > 300
    assert not np.array_equal(solo[0]["poses"], solo[1]["poses"])  # (different streams)


@pytest.mark.parametrize("mode", ["elastic", "elastic-staggered", "rigid"])
def test_closed_loop_agents_in_lockstep_with_grouped_stages_equal_their_solo_runs(mode, monkeypatch):
    """bench.py --agents-per-gpu A --lockstep: ONE thread drives the agents' tracking frame by frame, the tracking stages of all
    agents go out as one chain of launches per stage (so_track_group: the agent is a grid dimension of the search, the resolve
    and the PoseOptimization kernel), every agent keeps its own local-mapping thread.  Every agent's poses, counts,
    local-mapping log and keyframe bindings are those of its solo run, to the bit.  Reference concurrency: one process per agent,
    code/Examples/Monocular/swarm_map.cc:329-337.
    elastic (the default): a tick takes the agents whose next frame does not have to wait for its local-mapping packet - the others
    sit it out (so_dframe_group_submit with a null image, no row in the track group) and catch up; staggered: agent a has been
    run alone for a frames first, so the agents' keyframes fall on different ticks; rigid (SWARMORB_FLEET_RIGID=1): every tick
    takes every agent."""
    import torch
    from swarmmap_amd.replay import Replay
    if mode == "rigid":
        monkeypatch.setenv("SWARMORB_FLEET_RIGID", "1")
        monkeypatch.delenv("SWARMORB_FLEET_BA_GROUP_MIN", raising=False)  # four agents: a chain per window (the default below five)
    else:
        monkeypatch.delenv("SWARMORB_FLEET_RIGID", raising=False)
        monkeypatch.setenv("SWARMORB_FLEET_BA_GROUP_MIN", "2")            # the agents' windows as merged rounds (so_ba_group)
    K, dist, nfeat, n, A = synth.EUROC_K, synth.EUROC_DIST, 1000, 42, 4
    offs = [a if mode == "elastic-staggered" else 0 for a in range(A)]
    vocab = make_vocabulary()
    streams, blocks, ptrs = [], [], []
    for a in range(A):
        st = synth.FrameStream(seed=20221001 + 97 * a, size=synth.EUROC, K=K, dist=dist)
        block = torch.empty((n + 2, st.h, st.w), dtype=torch.uint8).pin_memory()
        view = block.numpy()
        for t in range(n + 2):
            view[t] = st.frame(t)
        streams.append(st); blocks.append(block)
        ptrs.append([block.data_ptr() + i * st.w * st.h for i in range(n + 2)])
    solo = [_cpp_chain(n, ptrs[a], streams[a], K, dist, nfeat, vocab)[0] for a in range(A)]
    fleet = []
    for a in range(A):
        rp = Replay(0, streams[a].w, streams[a].h, nfeat, 5, K, dist, plane_z=PLANE_Z, local_keyframes=12, third_pose=True)
        rp.set_frames(ptrs[a], on_device=False)
        rp.set_vocabulary(vocab)
        rp.set_closed_loop(policy=0)
        rp.prime(0)
        fleet.append(rp)
    for rp, off in zip(fleet, offs):
        if off:
            rp.run(0, off, True)
            rp.set_fleet_offset(off)
    m = n - max(offs)                      # frames per agent inside the fleet
    Replay.fleet_run(fleet, 0, 17, True)   # (two calls: the agents join and leave the group per call)
    Replay.fleet_run(fleet, 17, m - 17, True)
    for rp, off in zip(fleet, offs):       # (the staggered agents' last frames, alone again: n frames each in all)
        if off + m < n:
            rp.run(off + m, n - off - m, True)
    ticks, places = fleet[0].fleet_ticks()
    assert ticks >= m and places == A * m
    if mode == "rigid":
        assert ticks == m
    for rp in fleet:
        rp.drain()
        rp.finish()
    results = []
    for rp in fleet:
        p = rp.log()
        p.update(rp.closed_loop_log())
        results.append((p, rp.stats()))
    for rp in reversed(fleet):  # (the first agent owns what the fleet shares)
        rp.close()
    for a, (p, st) in enumerate(results):
        s = solo[a]
        for k in ("poses", "kf_poses", "Tcr", "lm_log", "matches_last", "matches_map", "inliers", "n_map_points", "point_bad"):
            assert np.array_equal(s[k], p[k]), (a, k)
        assert all(np.array_equal(x, y) for x, y in zip(s["kf_bindings"], p["kf_bindings"])), a
        assert s["inliers"][1:].min() > 300
        assert st["pose_calls"] == 3 * (n - 1), (a, st["pose_calls"])
        assert st["pose_kernel_ms"] > 0, (a, st["pose_kernel_ms"])
    assert not np.array_equal(solo[0]["poses"], solo[1]["poses"])


@pytest.mark.parametrize("name,n", [("euroc", 122), ("kitti", 42)])
def test_stages_chained_on_the_device_equal_the_separate_calls_over_a_whole_run(name, n):
    """so_track_stage_* (search -> resolve on the device -> PoseOptimization, one wait) against so_track_search_* + host resolve
    + host gather + so_pose_optimization inside the same closed loop: every pose, every count, the local-mapping log and the
    final bindings to the bit - the device-side resolve is exact and the pose kernel adds its sums in the same order."""
    import torch
    euroc = name == "euroc"
    size = synth.EUROC if euroc else synth.KITTI
    K = synth.EUROC_K if euroc else synth.KITTI_K
    dist = synth.EUROC_DIST if euroc else None
    nfeat = 1000 if euroc else 2000
    st = synth.FrameStream(seed=20221001, size=size, K=K, dist=dist)
    block = torch.empty((n + 2, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(n + 2):
        view[t] = st.frame(t)
    ptrs = [block.data_ptr() + i * st.w * st.h for i in range(n + 2)]
    vocab = make_vocabulary()
    a = _cpp_chain(n, ptrs, st, K, dist, nfeat, vocab, track_chain=True)[0]
    b = _cpp_chain(n, ptrs, st, K, dist, nfeat, vocab, track_chain=False)[0]
    for k in ("poses", "kf_poses", "Tcr", "lm_log", "matches_last", "matches_map", "inliers", "n_map_points", "point_bad"):
        assert np.array_equal(a[k], b[k]), k
    assert all(np.array_equal(x, y) for x, y in zip(a["kf_bindings"], b["kf_bindings"]))
    assert a["inliers"][1:].min() > 300
