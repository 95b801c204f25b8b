"""GPU parity of the Frame post-processing (UndistortKeyPoints, ComputeImageBounds, AssignFeaturesToGrid,
isInFrustum) against the CPU oracle, through the C ABI.  Bar: bit-exact (the kernels perform the oracle's
operations in the oracle's order, contraction-free)."""
import math

import numpy as np
import pytest

from swarmmap_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import swarmmap_amd
    assert swarmmap_amd.device_count() > 0, "these tests need a GPU"
    return swarmmap_amd


def _keypoints(seed, n, size):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.uniform(16, size[0] - 16, n), rng.uniform(16, size[1] - 16, n)], 1)
    return (np.floor(xy) * rng.choice([1.0, 1.2, 1.44, 1.728], (n, 1))).astype(np.float32)  # level-0 units, like the extractor's


@pytest.mark.parametrize("n", [0, 1, 7, 1000, 2047, 4000, 16384])
def test_prepare_matches_oracle_with_distortion(S, oracle, n):
    fp = S.FramePostProcessor(synth.EUROC_K, synth.EUROC_DIST)
    cam = oracle.camera(synth.EUROC_K, synth.EUROC_DIST)
    xy = _keypoints(n, n, synth.EUROC)
    r = fp.prepare(xy, 752, 480)
    b = oracle.image_bounds(cam, 752, 480)
    un = oracle.undistort_keypoints(cam, xy)
    g = oracle.assign_features_to_grid(un, b)
    assert r["bounds"].tobytes() == b.tobytes()
    assert r["xy_un"].tobytes() == un.tobytes()
    assert np.array_equal(r["cell_of"], g["cell_of"]) and np.array_equal(r["cell_start"], g["cell_start"])
    assert np.array_equal(r["cell_items"], g["cell_items"])
    r2 = fp.prepare(xy, 752, 480)  # later frames reuse the bounds (static members, Frame.cc:247)
    assert r2["xy_un"].tobytes() == un.tobytes() and np.array_equal(r2["cell_items"], g["cell_items"])
    fp.close()


def test_prepare_without_distortion_kitti(S, oracle):
    fp = S.FramePostProcessor(synth.KITTI_K)
    xy = _keypoints(5, 2000, synth.KITTI)
    xy[:10] = [[-5, 3], [1241, 376], [1240.9, 375.9], [0, 0], [620.5, 188], [1300, 10], [5, 400], [0.49, 0.49], [9.69, 3.9], [19.4, 7.8]]
    r = fp.prepare(xy, 1241, 376)
    g = oracle.assign_features_to_grid(xy, r["bounds"])
    assert r["bounds"].tolist() == [0.0, 1241.0, 0.0, 376.0] and np.array_equal(r["xy_un"], xy)
    assert np.array_equal(r["cell_of"], g["cell_of"]) and np.array_equal(r["cell_items"], g["cell_items"])
    assert (r["cell_of"] < 0).sum() >= 2  # points outside the image fall off the grid
    with pytest.raises(S.SwarmOrbError):
        fp.prepare(np.zeros((16385, 2), np.float32), 1241, 376)
    fp.close()


@pytest.mark.parametrize("seed,n", [(1, 4000), (2, 1), (3, 257), (4, 20000)])
def test_is_in_frustum_matches_oracle(S, oracle, seed, n):
    c = synth.make_frustum_case(seed, n)
    fp = S.FramePostProcessor(synth.EUROC_K, synth.EUROC_DIST)
    cam = oracle.camera(synth.EUROC_K, synth.EUROC_DIST)
    fp.prepare(np.zeros((0, 2), np.float32), 752, 480, grid=False)  # computes the bounds
    rng = np.random.default_rng(seed)
    init = (rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32),
            rng.normal(size=n).astype(np.float32), rng.integers(0, 99, n).astype(np.int32))  # stale track fields
    lsf = np.float32(math.log(1.2))
    r = fp.is_in_frustum(c["Tcw"], c["Xw"], c["normal"], c["max_dist"], c["min_dist"], 0.5, lsf, 8, init=init)
    o = oracle.is_in_frustum(cam, fp.bounds, c["Tcw"], c["Xw"], c["normal"], c["max_dist"], c["min_dist"], 0.5, lsf, 8,
                             init=init)
    for k in ("in_view", "proj_x", "proj_y", "view_cos", "pred_level"):
        assert r[k].tobytes() == o[k].tobytes(), k
    if n >= 1000:
        assert 0.1 < r["in_view"].mean() < 0.9
    fp.close()
