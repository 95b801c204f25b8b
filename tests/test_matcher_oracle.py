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


def test_bow_triangulation_window_oracles(oracle):
    from swarmmap_amd.matcher import FeatureVector
    kf1, node1, kf2, node2, src = synth.make_bow_case(9, 800, 700)
    fv1, fv2 = FeatureVector(node1), FeatureVector(node2)
    # FeatureVector flattening: ascending node ids, ascending feature index inside a node (DBoW2 push order)
    assert np.all(np.diff(fv1.node_id) > 0)
    for k in range(len(fv1.node_id)):
        seg = fv1.idx[fv1.off[k]:fv1.off[k + 1]]
        assert np.all(np.diff(seg) > 0) and np.all(node1[seg] == fv1.node_id[k])
    for variant in (0, 1):
        nm, m2, m1 = oracle.search_by_bow(variant, kf1, fv1, kf2, fv2, 0.75, False)
        sel = np.nonzero(m1 >= 0)[0]
        assert nm == len(sel) > 50
        assert len(set(m1[sel].tolist())) == len(sel)          # a target is bound at most once
        assert np.all(kf1["valid"][sel] == 1)
        assert np.all(node1[sel] == node2[m1[sel]])            # only features of the same vocabulary node match
        if variant == 1:
            assert np.all(kf2["valid"][m1[sel]] == 1)
        for i in sel[:100]:
            d = oracle.descriptor_distance(kf1["desc"][i], kf2["desc"][m1[i]])
            assert d <= 50 if variant == 0 else d < 50
    # SearchForTriangulation: ties go to the LAST candidate of the node (":688 dist > bestDist -> continue")
    d = np.zeros((1, 32), np.uint8)
    kfa = dict(x=[100.0], y=[50.0], angle=[0.0], desc=d, free=[1])
    kfb = dict(x=[10.0, 20.0, 30.0], y=[50.0, 50.0, 50.0], octave=[0, 0, 0], angle=[0.0, 0.0, 0.0],
               desc=np.zeros((3, 32), np.uint8), free=[1, 1, 1])
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    sf = synth.SCALE_FACTORS
    nm, m12 = oracle.search_for_triangulation(kfa, FeatureVector([5]), kfb, FeatureVector([5, 5, 5]), F12,
                                              (900.0, 900.0), sf, sf * sf, False)
    assert nm == 1 and m12[0] == 2
    # window_best == brute force over the literal grid scan
    rng = np.random.default_rng(4)
    fr = synth.make_frame_arrays(rng, 600)
    KF = _frame(fr)
    q = synth.make_window_queries(4, fr, 300)
    inv = (1.0 / synth.SCALE_FACTORS ** 2).astype(np.float32)
    bi, bd = oracle.search_window_best(KF, q, True, inv)
    for i in range(300):
        if not q["valid"][i]:
            assert bi[i] == -1
            continue
        cand = oracle.features_in_area(KF, q["u"][i], q["v"][i], q["radius"][i], -1, -1)
        best, bdist = -1, 256
        for c in cand:
            if not (q["pred_level"][i] - 1 <= fr["octave"][c] <= q["pred_level"][i]):
                continue
            e2 = np.float32(np.float32(q["u"][i] - fr["x"][c]) ** 2 + np.float32(q["v"][i] - fr["y"][c]) ** 2)
            if float(np.float32(e2 * inv[fr["octave"][c]])) > 5.99:
                continue
            dd = oracle.descriptor_distance(q["desc"][i], fr["desc"][c])
            if dd < bdist:
                best, bdist = int(c), dd
        assert (bi[i], bd[i]) == (best, bdist)


def test_distinctive_descriptor_against_numpy(oracle):
    """MapPoint::ComputeDistinctiveDescriptors: least median of the Hamming rows, first index on ties."""
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 4, 9, 30):
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        d = np.stack([np.bitwise_xor(base, np.packbits(rng.random(256) < p)) for p in rng.uniform(0.0, 0.3, n)])
        bits = np.unpackbits(d, axis=1)
        D = (bits[:, None, :] != bits[None, :, :]).sum(2)
        med = np.sort(D, axis=1)[:, int(0.5 * (n - 1))]
        idx, m = oracle.distinctive_descriptors([0, n], d)
        assert m[0] == med.min() and idx[0] == int(np.argmax(med == med.min()))
    idx, m = oracle.distinctive_descriptors([0, 0, 2], np.zeros((2, 32), np.uint8))
    assert idx.tolist() == [-1, 0] and m[1] == 0
