"""CPU tests of the matcher oracle: KATs that pin it to the reference semantics."""
import numpy as np

from swarmmap_amd import synth
from swarmmap_amd.matcher import FrameView


def _frame(fr):
    return FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"],
                     fr.get("excluded"))


def test_descriptor_distance_is_popcount(oracle):
    rng = np.random.default_rng(0)
    for _ in range(300):
        a = rng.integers(0, 256, 32).astype(np.uint8)
        b = rng.integers(0, 256, 32).astype(np.uint8)
        assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    z = np.zeros(32, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0
    assert oracle.descriptor_distance(z, np.full(32, 255, np.uint8)) == 256


def test_grid_traversal_equals_global_rank_order(oracle):
    """GetFeaturesInArea's result == {in-grid kp : |dx|<r, |dy|<r, level ok} visited in (cell x, cell y, index)
    order — the property the GPU kernel's brute-force mask + rank keys rely on."""
    rng = np.random.default_rng(1)
    fr = synth.make_frame_arrays(rng, 1500)
    F = _frame(fr)
    px = np.round((F.x - F.min_x) * F.grid_inv_w).astype(int)  # PosInGrid (round half away == np.round for .5? no:
    py = np.round((F.y - F.min_y) * F.grid_inv_h).astype(int)  # ties are measure-zero for random floats)
    in_grid = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    for _ in range(200):
        x, y = rng.uniform(-20, 770), rng.uniform(-20, 500)
        r = np.float32(rng.uniform(1, 120))
        mn, mx = int(rng.integers(-1, 4)), int(rng.integers(-1, 8))
        got = oracle.features_in_area(F, x, y, r, mn, mx)
        ok = in_grid & (np.abs(F.x - np.float32(x)) < r) & (np.abs(F.y - np.float32(y)) < r)
        if (mn > 0) or (mx >= 0):
            ok &= F.octave >= mn
            if mx >= 0:
                ok &= F.octave <= mx
        idx = np.nonzero(ok)[0]
        order = np.lexsort((idx, py[idx], px[idx]))
        assert got.tolist() == idx[order].tolist()


def test_three_maxima_and_histogram_quirk(oracle):
    import ctypes as C
    lib = oracle.lib()
    sizes = np.zeros(30, np.int32)
    sizes[[2, 5, 9]] = [100, 50, 5]
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib.orc_three_maxima(sizes.ctypes.data_as(C.c_void_p), 30, C.byref(a), C.byref(b), C.byref(c))
    assert (a.value, b.value, c.value) == (2, 5, -1)  # third < 0.1 * max dropped
    sizes[5] = 9
    lib.orc_three_maxima(sizes.ctypes.data_as(C.c_void_p), 30, C.byref(a), C.byref(b), C.byref(c))
    assert (a.value, b.value, c.value) == (2, -1, -1)


def test_m1_sanity(oracle):
    fr, mps = synth.make_m1_case(3)
    F = _frame(fr)
    nm, kp_to_mp = oracle.search_by_projection_mappoints(F, mps, 1.0, 0.8)
    assert nm > 200
    m = kp_to_mp >= 0
    assert not np.any(fr["excluded"][m])  # pre-bound keypoints are never re-bound
    assert np.all(mps["in_view"][kp_to_mp[m]] == 1)
    # accepted matches obey TH_HIGH
    for k in np.nonzero(m)[0][:200]:
        assert oracle.descriptor_distance(fr["desc"][k], mps["desc"][kp_to_mp[k]]) <= 100
    nm3, _ = oracle.search_by_projection_mappoints(F, mps, 3.0, 0.8)
    assert nm3 != nm  # th changes the window


def test_m2_and_m4_sanity(oracle):
    fr, last = synth.make_m2_case(4)
    nm, kp_to_last = oracle.search_by_projection_lastframe(_frame(fr), last, 15.0, True)
    # nmatches++ also fires when a keypoint bound to an observation-less point is re-bound (reference quirk)
    assert nm >= int((kp_to_last >= 0).sum())
    assert nm > 100
    nm_no, _ = oracle.search_by_projection_lastframe(_frame(fr), last, 15.0, False)
    assert nm_no >= nm  # the rotation histogram only removes matches
    f1, f2, prev = synth.make_m4_case(5)
    nm4, m12, pm = oracle.search_for_initialization(_frame(f1), _frame(f2), prev, 100, 0.9, True)
    assert nm4 == int((m12 >= 0).sum()) and nm4 > 300
    sel = np.nonzero(m12 >= 0)[0]
    assert np.all(f1["octave"][sel] == 0)  # only level-0 keypoints are matched (:393)
    assert np.allclose(pm[sel, 0], f2["x"][m12[sel]])
    assert len(set(m12[sel].tolist())) == len(sel)  # one-to-one after the vnMatches21 back-check


def test_top2(oracle):
    rng = np.random.default_rng(7)
    B = rng.integers(0, 256, (300, 32)).astype(np.uint8)
    A = synth.flip_bits(rng, B[rng.integers(0, 300, 100)], 0.1)
    bi, bd, sd = oracle.hamming_top2(A, B)
    D = np.unpackbits(A[:, None, :] ^ B[None, :, :], axis=2).sum(2)
    assert np.array_equal(bi, D.argmin(1))
    assert np.array_equal(bd, D.min(1))
    assert np.array_equal(sd, np.sort(D, 1)[:, 1])
