"""GPU parity of the local-mapping thread's two per-point loops (so_triangulate_matches, so_update_normal_and_depth)
against oracle/mapping_oracle.c through the C ABI: float work in the reference's expression order on both sides, every
operation an IEEE +, -, *, / or sqrt - bit-exact, tolerance 0."""
import numpy as np
import pytest

from swarmmap_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import swarmmap_amd
    assert swarmmap_amd.device_count() > 0, "these tests need a GPU"
    return swarmmap_amd


@pytest.mark.parametrize("seed,n", [(1, 1500), (2, 6000), (3, 1), (4, 129)])
def test_triangulate_matches(S, oracle, seed, n):
    c = synth.make_triangulation_case(seed, n)
    m = S.ORBmatcher()
    ok, X = m.TriangulateMatches(c["kf1"], [c["kf2"]], c["ratio_factor"], np.zeros(n, np.int32), c["xy1"], c["octave1"], c["xy2"],
                                 c["octave2"])
    ook, oX = oracle.triangulate_matches(c["kf1"], c["kf2"], c["ratio_factor"], c["xy1"], c["octave1"], c["xy2"], c["octave2"])
    assert np.array_equal(ok, ook)
    assert X[ok.astype(bool)].tobytes() == oX[ook.astype(bool)].tobytes()
    if n > 1000:
        assert 0.5 * n < ok.sum() < 0.95 * n
    m.close()


def test_triangulate_matches_of_twenty_neighbours_in_one_launch(S, oracle):
    """CreateNewMapPoints walks <= 20 neighbours; their matches go out together, each match naming its neighbour."""
    cases = [synth.make_triangulation_case(20 + j, 150 + 17 * j) for j in range(20)]
    kf1 = cases[0]["kf1"]
    for c in cases:  # the same current keyframe against twenty different neighbours
        c["kf1"] = kf1
    of = np.concatenate([np.full(len(c["octave1"]), j, np.int32) for j, c in enumerate(cases)])
    cat = lambda k: np.concatenate([c[k] for c in cases])  # noqa: E731
    perm = np.random.default_rng(0).permutation(len(of))  # any order of the matches
    m = S.ORBmatcher()
    ok, X = m.TriangulateMatches(kf1, [c["kf2"] for c in cases], cases[0]["ratio_factor"], of[perm], cat("xy1")[perm],
                                 cat("octave1")[perm], cat("xy2")[perm], cat("octave2")[perm])
    want_ok, want_X = [], []
    for c in cases:
        a, b = oracle.triangulate_matches(kf1, c["kf2"], c["ratio_factor"], c["xy1"], c["octave1"], c["xy2"], c["octave2"])
        want_ok.append(a); want_X.append(b)
    want_ok, want_X = np.concatenate(want_ok)[perm], np.concatenate(want_X)[perm]
    assert np.array_equal(ok, want_ok) and X[ok.astype(bool)].tobytes() == want_X[want_ok.astype(bool)].tobytes()
    assert ok.sum() > 100
    with pytest.raises(Exception):  # a match that names a neighbour that was not passed
        m.TriangulateMatches(kf1, [cases[0]["kf2"]], 1.8, np.array([1], np.int32), cases[0]["xy1"][:1], cases[0]["octave1"][:1],
                             cases[0]["xy2"][:1], cases[0]["octave2"][:1])
    m.close()


@pytest.mark.parametrize("seed,n,max_obs", [(1, 3000, 12), (2, 20000, 40), (3, 1, 3)])
def test_update_normal_and_depth(S, oracle, seed, n, max_obs):
    c = synth.make_normal_depth_case(seed, n, max_obs)
    args = (c["offsets"], c["obs_Ow"], c["Xw"], c["ref_Ow"], c["ref_level_scale"], c["ref_last_scale"], c["normal"], c["max_dist"],
            c["min_dist"])
    m = S.ORBmatcher()
    got = m.UpdateNormalAndDepth(*args)
    want = oracle.update_normal_and_depth(*args)
    for g, w in zip(got, want):
        assert g.tobytes() == w.tobytes()
    m.close()
