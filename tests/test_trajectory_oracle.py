"""Trajectory-level check of the oracle operators chained by swarmmap_amd.minitrack (CPU only, SURVEY.md 8d ATE)."""
import numpy as np

from swarmmap_amd import minitrack, synth
from trajectory_common import OracleBackend

PLANE_Z = 2.0


def test_umeyama_recovers_a_known_similarity():
    rng = np.random.default_rng(3)
    src = rng.normal(size=(50, 3))
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    R = A * np.sign(np.linalg.det(A))
    dst = 1.7 * src @ R.T + np.array([0.3, -2.0, 5.0])
    s, R2, t2 = minitrack.umeyama(src, dst, with_scale=True)
    assert abs(s - 1.7) < 1e-12 and np.allclose(R2, R, atol=1e-12) and np.allclose(t2, [0.3, -2.0, 5.0], atol=1e-12)
    assert minitrack.ate_rmse(src, dst, with_scale=True) < 1e-12
    assert minitrack.ate_rmse(src, dst, with_scale=False) > 0.1


def test_oracle_trajectory_follows_ground_truth():
    n = 24
    K = synth.EUROC_K
    st = synth.FrameStream()
    r = minitrack.track(OracleBackend(K), st, n, K, plane_z=PLANE_Z)
    gt = minitrack.ground_truth(st, n, K, PLANE_Z)
    # one pixel is plane_z / fx = 4.4 mm: the replay must stay sub-pixel over the whole path, without alignment
    assert minitrack.ate_rmse(r["centres"], gt, align=False) < 3e-3
    assert np.abs(r["centres"] - gt).max() < 6e-3
    assert r["inliers"][1:].min() > 400 and r["matches_last"][1:].min() > 300
    assert r["n_map_points"][-1] > r["n_map_points"][0]      # the map grew at "keyframes"


def test_oracle_trajectory_with_local_bundle_adjustment():
    n = 40
    K = synth.EUROC_K
    st = synth.FrameStream()
    r = minitrack.track(OracleBackend(K), st, n, K, plane_z=PLANE_Z, local_ba=True)
    gt = minitrack.ground_truth(st, n, K, PLANE_Z)
    assert len(r["lba_edges"]) >= 3 and r["lba_edges"].max() > 1000      # windows were optimised
    assert r["lba_outliers"].sum() < 0.05 * r["lba_edges"].sum()
    assert minitrack.ate_rmse(r["centres"], gt, align=False) < 4e-3       # still sub-pixel (4.4 mm)
    assert r["inliers"][1:].min() > 400
