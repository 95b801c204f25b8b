"""Trajectory parity: the HIP operators and the CPU oracle, chained by the same host loop (swarmmap_amd.minitrack),
must produce the same trajectory from the same images (ATE between the HIP path and the CPU path,
and against the renderer's ground truth)."""
import numpy as np
import pytest

from swarmmap_amd import minitrack, synth
from trajectory_common import OracleBackend

pytestmark = pytest.mark.gpu

PLANE_Z = 2.0
# One pixel is plane_z / fx = 4.4 mm (EuRoC) / 2.8 mm (KITTI).  The two paths may differ by PoseOptimization's
# floating-point tolerance (2e-5 per call, tests/test_pose_gpu.py) which a later match decision can amplify;
# measured 1e-8 m over 300 frames (profiles/r1t_ate.jsonl); the bound below is 1/4000 of a pixel.
ATE_HIP_VS_ORACLE = 1e-6


@pytest.mark.parametrize("size,K,nfeat,n", [(synth.EUROC, synth.EUROC_K, 1000, 60), (synth.KITTI, synth.KITTI_K, 2000, 30)])
def test_hip_and_oracle_trajectories_agree(size, K, nfeat, n):
    st = synth.FrameStream(size=size)
    hip = minitrack.HipBackend(K, nfeat)
    a = minitrack.track(hip, st, n, K, plane_z=PLANE_Z)
    hip.close()
    b = minitrack.track(OracleBackend(K, nfeat), st, n, K, plane_z=PLANE_Z)
    gt = minitrack.ground_truth(st, n, K, PLANE_Z)
    ate = minitrack.ate_rmse(a["centres"], b["centres"], align=False)
    assert ate < ATE_HIP_VS_ORACLE, ate
    assert np.abs(a["poses"] - b["poses"]).max() < 2e-5
    # same integer decisions along the way (extractor and matchers are bit-exact; allow the rare match that flips
    # because the two pose estimates differ in the last bits)
    for k in ("matches_last", "matches_map", "inliers"):
        assert np.abs(a[k].astype(int) - b[k].astype(int)).max() <= 3, (k, a[k], b[k])
    assert np.array_equal(a["n_map_points"], b["n_map_points"]) or \
        np.abs(a["n_map_points"] - b["n_map_points"]).max() <= 3
    px = PLANE_Z / float(K[0])
    assert minitrack.ate_rmse(a["centres"], gt, align=False) < 0.75 * px
    assert abs(minitrack.ate_rmse(a["centres"], gt) - minitrack.ate_rmse(b["centres"], gt)) < 0.01 * px


def test_tracking_plus_local_ba_trajectories_agree():
    """The metric's second half: tracking + LocalBundleAdjustment.  LBA agrees with the oracle to 2e-5 per window
    (tests/test_ba_gpu.py), a difference later match decisions can amplify: the bound is 1/40 of a pixel."""
    n, K = 60, synth.EUROC_K
    st = synth.FrameStream()
    hip = minitrack.HipBackend(K, 1000)
    a = minitrack.track(hip, st, n, K, plane_z=PLANE_Z, local_ba=True)
    hip.close()
    b = minitrack.track(OracleBackend(K, 1000), st, n, K, plane_z=PLANE_Z, local_ba=True)
    gt = minitrack.ground_truth(st, n, K, PLANE_Z)
    assert len(a["lba_edges"]) >= 5 and np.array_equal(a["lba_edges"], b["lba_edges"])
    assert minitrack.ate_rmse(a["centres"], b["centres"], align=False) < 1e-4
    assert np.abs(a["inliers"].astype(int) - b["inliers"].astype(int)).max() <= 5
    px = PLANE_Z / float(K[0])
    assert minitrack.ate_rmse(a["centres"], gt, align=False) < px


@pytest.mark.parametrize("local_kfs", [0, 4])
def test_cpp_chain_matches_the_oracle_chain(local_kfs):
    """The C++ host loop of bench.py (swarmmap_amd/host/replay.cc: device-resident frames, map table on the GPU, fused
    tracking searches, pipelined extraction) against the same chain run over the CPU oracle, frame by frame."""
    from swarmmap_amd.replay import Replay
    n, K, nfeat = 48, synth.EUROC_K, 1000
    st = synth.FrameStream()
    frames = [st.frame(t) for t in range(n + 1)]
    rp = Replay(0, st.w, st.h, nfeat, 5, K, dist=None, plane_z=PLANE_Z, local_keyframes=local_kfs, third_pose=True)
    rp.set_host_frames(frames)
    rp.prime(0)
    rp.run(0, n, False)
    rp.finish()
    a = rp.log()
    rp.close()
    b = minitrack.track(OracleBackend(K, nfeat), st, n, K, plane_z=PLANE_Z, local_keyframes=local_kfs, third_pose=True)
    assert len(a["poses"]) == n
    assert minitrack.ate_rmse(a["centres"], b["centres"], align=False) < ATE_HIP_VS_ORACLE
    assert np.abs(a["poses"] - b["poses"]).max() < 2e-5
    for k in ("matches_last", "matches_map", "inliers"):
        assert np.abs(a[k].astype(int) - b[k].astype(int)).max() <= 3, (k, a[k], b[k])
    assert np.abs(a["n_map_points"].astype(int) - b["n_map_points"].astype(int)).max() <= 3
    assert a["matches_last"][1:].min() > 300 and a["inliers"][1:].min() > 400


@pytest.mark.parametrize("name", ["euroc", "kitti"])
def test_cpp_chain_in_the_bench_configuration_matches_the_oracle_chain(name):
    """Exactly what bench.py times (bench.py run_stream): the EuRoC-sized stream rendered through the EuRoC lens model
    (UndistortKeyPoints does real work) / the KITTI-sized stream with 2000 features, frames in pinned host memory read
    by the ingest kernel, the local map limited to the last 12 keyframes, the third PoseOptimization, and the
    local-mapping thread on the same GPU while the tracking thread runs: every 5th frame becomes its new keyframe -
    SearchForTriangulation against each of the last <= 20 keyframes, Fuse into each of them and back
    (code/src/LocalMapping.cc:197-246, 451-481), then an LBA-M window - frame by frame, and matcher job by matcher job,
    against the same chain over the CPU oracle."""
    import torch
    from swarmmap_amd.replay import Replay
    euroc = name == "euroc"
    size = synth.EUROC if euroc else synth.KITTI
    K = synth.EUROC_K if euroc else synth.KITTI_K
    dist = synth.EUROC_DIST if euroc else None
    nfeat = 1000 if euroc else 2000
    n = 40 if euroc else 30
    st = synth.FrameStream(seed=20221001, size=size, K=K, dist=dist)
    block = torch.empty((n + 2, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(n + 2):
        view[t] = st.frame(t)
    frames = [view[t] for t in range(n + 2)]
    rp = Replay(0, st.w, st.h, nfeat, 5, K, dist, plane_z=PLANE_Z, local_keyframes=12, third_pose=True)
    rp.set_frames([block.data_ptr() + i * st.w * st.h for i in range(n + 2)], on_device=False)
    from swarmmap_amd.replay import make_vocabulary
    vocab = make_vocabulary()
    rp.set_window(synth.make_ba_case("LBA-M", seed=100))
    rp.set_vocabulary(vocab)
    rp.preallocate()
    rp.prime(0)
    rp.run(0, n, True)
    rp.drain()
    rp.finish()
    a, stats, a_lm, lm_stats = rp.log(), rp.stats(), rp.lm_log(), rp.lm_stats()
    rp.close()
    assert stats["n_lba"] >= n // 5 - 1, "the local-mapping thread did not run its windows"
    b = minitrack.track(OracleBackend(K, nfeat, dist if dist is not None else (0, 0, 0, 0, 0)), None, n, K, plane_z=PLANE_Z,
                        local_keyframes=12, third_pose=True, frames=frames, lm_every=5, vocab=vocab)
    # the local-mapping thread's matcher jobs: same keyframes, same neighbours; match / fuse counts may differ by the
    # rare decision that flips on the 2e-5 pose difference between the two chains
    b_lm = b["lm_log"]
    assert len(a_lm) == len(b_lm) == (n + 4) // 5 and np.array_equal(a_lm[:, :2], b_lm[:, :2])
    assert np.all(np.abs(a_lm[:, 2:].astype(int) - b_lm[:, 2:].astype(int)) <= 3 + b_lm[:, 2:] // 100), (a_lm, b_lm)
    assert a_lm[-1, 1] == min(20, len(a_lm) - 1) and a_lm[2:, 2].min() > 20 and a_lm[1:, 3].min() > 100
    assert a_lm.shape[1] == 6 and a_lm[2:, 5].max() > 0  # CreateNewMapPoints' triangulation produces new map points
    assert lm_stats["jobs"] == len(a_lm) and lm_stats["tri_calls"] == a_lm[:, 1].sum()
    # all searches of a keyframe travel as one so_matcher batch (SWARMORB_LM_BATCH=0: one by one, per-call kernel times)
    assert lm_stats["batch_kernel_ms"] > 0 or (lm_stats["tri_kernel_ms"] > 0 and lm_stats["fuse_kernel_ms"] > 0)
    assert len(a["poses"]) == n
    assert minitrack.ate_rmse(a["centres"], b["centres"], align=False) < ATE_HIP_VS_ORACLE
    assert np.abs(a["poses"] - b["poses"]).max() < 2e-5
    for k in ("matches_last", "matches_map", "inliers"):
        assert np.abs(a[k].astype(int) - b[k].astype(int)).max() <= 3, (k, a[k], b[k])
    assert np.abs(a["n_map_points"].astype(int) - b["n_map_points"].astype(int)).max() <= 3
    assert a["matches_last"][1:].min() > 300 and a["inliers"][1:].min() > 400


def test_lockstep_fleet_gives_every_agent_its_solo_result():
    """so_fleet_run: three agents of one GPU driven in lockstep by one thread (searches submitted for all agents before
    any is waited for, PoseOptimization of all agents in one launch) - every agent's trajectory is the one a solo run
    of the same stream produces."""
    from swarmmap_amd.replay import Replay, private_streams
    n, K, nfeat = 40, synth.EUROC_K, 1000
    streams = [synth.FrameStream(seed=20221001 + 7 * a) for a in range(3)]
    frames = [[st.frame(t) for t in range(n + 1)] for st in streams]
    solo = []
    for a in range(3):
        rp = Replay(0, streams[a].w, streams[a].h, nfeat, 5, K, plane_z=PLANE_Z, local_keyframes=6, third_pose=True)
        rp.set_host_frames(frames[a])
        rp.prime(0)
        rp.run(0, n, False)
        rp.finish()
        solo.append(rp.log())
        rp.close()
    private_streams(True)
    try:
        fleet = []
        for a in range(3):
            rp = Replay(0, streams[a].w, streams[a].h, nfeat, 5, K, plane_z=PLANE_Z, local_keyframes=6, third_pose=True)
            rp.set_host_frames(frames[a])
            rp.prime(0)
            fleet.append(rp)
        Replay.fleet_run(fleet, 0, n, False)
        for a, rp in enumerate(fleet):
            rp.finish()
            g = rp.log()
            rp.close()
            assert len(g["poses"]) == n
            assert np.abs(g["poses"] - solo[a]["poses"]).max() < 2e-5
            for k in ("matches_last", "matches_map", "inliers", "n_map_points"):
                assert np.abs(g[k].astype(int) - solo[a][k].astype(int)).max() <= 3, (a, k)
    finally:
        private_streams(False)
