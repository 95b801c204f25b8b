"""Known-answer tests that pin the CPU oracle to the reference's own constants.

The reference ships no tests or golden vectors; these KATs are derived from constants in its source:
  FAST lookup table   code/src/cuda/Fast_gpu.cu:58        (tests/golden/fast_table.bin)
  rBRIEF pattern      code/src/ORBextractor.cc:80-338     (tests/golden/brief_pattern_i32.bin)
  umax / level sizes / features-per-level                 code/src/ORBextractor.cc:340-405,824-825
"""
import hashlib
import math
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_fast_table_hash_and_predicate(oracle):
    table = open(os.path.join(GOLD, "fast_table.bin"), "rb").read()
    assert len(table) == 8129
    assert hashlib.sha256(table).hexdigest() == \
        "f4ca464f18605aaaea668d8759ca9e8aeac6adad900895b5eb7077d84d4f8a9d"
    tb = np.frombuffer(table, np.uint8)
    lib = oracle.lib()
    import ctypes as C
    tp = tb.ctypes.data_as(C.c_void_p)
    mism = 0
    checked = 0
    for m in range(1 << 16):
        run9 = lib.orc_fast_is_corner_masks(m, 0)
        if bin(m).count("1") > 8:  # the only masks the reference ever looks up (Fast_gpu.cu:191)
            checked += 1
            mism += int(bool(run9) != bool(lib.orc_fast_table_lookup(tp, m)))
        else:
            assert not run9  # a 9-run needs at least 9 set bits
    assert checked == 26333
    assert mism == 0


def test_brief_pattern_hash():
    blob = open(os.path.join(GOLD, "brief_pattern_i32.bin"), "rb").read()
    assert hashlib.sha256(blob).hexdigest() == \
        "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
    pat = np.frombuffer(blob, "<i4")
    assert np.abs(pat).max() == 13
    r = np.sqrt((pat.reshape(-1, 2).astype(np.float64) ** 2).sum(1)).max()
    assert abs(r - 18.385) < 1e-3
    # both embedded copies (oracle + product) carry exactly these numbers
    root = os.path.dirname(os.path.dirname(__file__))
    for rel in ("oracle/brief_pattern.inc", "swarmmap_amd/csrc/brief_pattern.inc"):
        txt = open(os.path.join(root, rel)).read()
        txt = txt[txt.index("*/") + 2:]
        vals = np.array([int(v) for v in txt.replace("\n", " ").split(",") if v.strip()], np.int32)
        assert np.array_equal(vals, pat), rel


def test_tables(oracle):
    t = oracle.make_tables(oracle.config(1000))
    assert list(t.umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # circular patch = 749 px
    assert sum(2 * u + 1 for u in list(t.umax)[1:]) * 2 + 31 == 749
    assert list(t.features_per_level)[:8] == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(oracle.make_tables(oracle.config(2000)).features_per_level)[:8] == \
        [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(oracle.make_tables(oracle.config(4000)).features_per_level)[:8] == \
        [869, 724, 603, 503, 419, 349, 291, 242]
    assert oracle.level_sizes(oracle.config(), 752, 480) == \
        [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]
    assert oracle.level_sizes(oracle.config(), 1241, 376) == \
        [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]
    s = np.array(list(t.scale)[:8], np.float32)
    assert s[0] == 1.0 and abs(s[7] - 1.2 ** 7) < 1e-5
    assert np.allclose(np.array(list(t.sigma2)[:8]), s * s)


def test_fast_score_closed_form(oracle):
    """cornerScore's binary search == max over 9-runs of min |diff| - 1 (what the HIP kernel computes)."""
    rng = np.random.default_rng(5)
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    n_corner = 0
    for it in range(4000):
        img = rng.integers(0, 256, (7, 7)).astype(np.uint8)
        if it % 2:  # make corners likely
            img[:] = rng.integers(60, 200)
            k0 = rng.integers(0, 16)
            ln = rng.integers(7, 14)
            delta = int(rng.integers(8, 56)) * (1 if it % 4 == 1 else -1)
            for k in range(ln):
                kk = (k0 + k) % 16
                img[3 + dy[kk], 3 + dx[kk]] = np.clip(int(img[3, 3]) + delta + rng.integers(-3, 4), 0, 255)
        v = int(img[3, 3])
        d = np.array([int(img[3 + dy[k], 3 + dx[k]]) - v for k in range(16)])
        best = -10 ** 9
        for s in range(16):
            w = [d[(s + k) % 16] for k in range(9)]
            best = max(best, min(w), min(-x for x in w))
        closed = best - 1
        for th in (7, 20):
            got = oracle.fast_score(img, 3, 3, th)
            want = closed if closed >= th else 0
            assert got == want, (it, th, got, want)
            n_corner += got > 0
    assert n_corner > 500


def test_fast_detect_rules(oracle):
    """Deterministic tile rule: a tile with no high-threshold survivor falls back to the low threshold."""
    rng = np.random.default_rng(11)
    img = np.full((160, 200), 100, np.uint8)
    # strong isolated corner (bright 4x4 block corner) far from anything else -> high-threshold tile
    img[40:60, 40:60] = 180
    # weak structure (contrast 12: between low=7 and high=20) in a different tile
    img[100:120, 130:150] = 112
    img = (img.astype(np.int16) + rng.integers(-1, 2, img.shape)).astype(np.uint8)  # break NMS ties
    xs, ys, sc = oracle.fast_detect(img, 20, 7)
    pts = set(zip((xs + 16).tolist(), (ys + 16).tolist()))
    assert len(pts) > 0
    strong = [(x, y) for (x, y) in pts if 35 <= x <= 65 and 35 <= y <= 65]
    weak = [(x, y) for (x, y) in pts if 95 <= y <= 125 and 125 <= x <= 155]
    assert strong and weak
    # with both thresholds high, the weak block disappears
    xs2, ys2, _ = oracle.fast_detect(img, 20, 20)
    pts2 = set(zip((xs2 + 16).tolist(), (ys2 + 16).tolist()))
    assert not [(x, y) for (x, y) in pts2 if 95 <= y <= 125 and 125 <= x <= 155]
    # raster order
    key = ys.astype(np.int64) * 100000 + xs
    assert np.all(np.diff(key) > 0)
    # coordinates stay inside the tested ROI range [3, dim-3)
    assert xs.min() >= 3 and ys.min() >= 3 and xs.max() < 200 - 32 - 3 and ys.max() < 160 - 32 - 3


def test_det_math_accuracy(oracle):
    lib = oracle.lib()
    import ctypes as C
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(20000):
        y, x = rng.integers(-3000000, 3000000, 2)
        got = lib.orc_atan2f(float(y), float(x))
        worst = max(worst, abs(got - math.atan2(float(y), float(x))))
    assert worst < 4e-7, worst
    assert lib.orc_atan2f(0.0, 0.0) == 0.0
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for deg in np.linspace(0, 360, 7201):
        lib.orc_sincosf_deg(float(deg), C.byref(s), C.byref(c))
        r = float(np.float32(np.float32(deg) * np.float32(math.pi / 180.0)))
        worst = max(worst, abs(s.value - math.sin(r)), abs(c.value - math.cos(r)))
    assert worst < 3e-7, worst


def test_resize_and_blur_conventions(oracle):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (48, 60)).astype(np.uint8)
    # identity-size resize reproduces the source
    assert np.array_equal(oracle.resize_linear(src, 60, 48), src)
    # constant image stays constant through resize and blur
    c = np.full((40, 50), 77, np.uint8)
    assert np.all(oracle.resize_linear(c, 42, 33) == 77)
    assert np.all(oracle.gaussian7(c) == 77)
    # blur == numpy float64 separable blur within 1 grey level, reflect-101 borders
    k = np.array([math.exp(-(i - 3) ** 2 / 8.0) for i in range(7)])
    k /= k.sum()
    pad = np.pad(src.astype(np.float64), 3, mode="reflect")
    tmp = sum(k[i] * pad[:, i:i + 60] for i in range(7))
    ref = sum(k[i] * tmp[i:i + 48, :] for i in range(7))
    assert np.abs(oracle.gaussian7(src).astype(np.float64) - ref).max() <= 0.5 + 1e-3
    # border: reflect-101
    b = oracle.border_reflect101(src, 19)
    assert np.array_equal(b, np.pad(src, 19, mode="reflect"))


def test_octree_basic(oracle):
    rng = np.random.default_rng(3)
    W, H = 720, 448
    pts = set()
    while len(pts) < 3000:
        pts.add((int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))))
    pts = sorted(pts, key=lambda p: (p[1], p[0]))
    xs = np.array([p[0] for p in pts], np.int16)
    ys = np.array([p[1] for p in pts], np.int16)
    sc = rng.integers(7, 200, len(pts)).astype(np.uint8)
    for N in (60, 217, 500):
        idx = oracle.distribute_octree(xs, ys, sc, W, H, N)
        assert N <= len(idx) <= N + 3  # a split adds <= 3 nodes before the >=N check
        assert len(set(idx.tolist())) == len(idx)
    # fewer candidates than requested: every candidate in its own node
    idx = oracle.distribute_octree(xs[:50], ys[:50], sc[:50], W, H, 217)
    assert sorted(idx.tolist()) == list(range(50))
    # best response per node: the global maximum response always survives
    idx = oracle.distribute_octree(xs, ys, sc, W, H, 217)
    assert sc[idx].max() == sc.max()


def test_extract_end_to_end_shape(oracle):
    from swarmmap_amd import synth
    img = synth.make_image(7)
    cfg = oracle.config(1000)
    kps, desc = oracle.extract(cfg, img)
    assert 900 <= len(kps) <= 1000 + 16
    assert desc.shape == (len(kps), 32)
    assert np.all(np.diff(kps["octave"]) >= 0)  # levels concatenated 0..7
    assert np.all((kps["angle"] >= 0) & (kps["angle"] <= 360))
    t = oracle.make_tables(cfg)
    for l in range(8):
        m = kps["octave"] == l
        assert np.all(kps["size"][m] == float(int(31 * t.scale[l])))
    # level-0 points are integer pixel coordinates inside [19, dim-19)
    m0 = kps["octave"] == 0
    assert np.all(kps["x"][m0] == np.rint(kps["x"][m0]))
    assert kps["x"][m0].min() >= 19 and kps["x"][m0].max() < 752 - 19
    # deterministic
    kps2, desc2 = oracle.extract(cfg, img)
    assert kps.tobytes() == kps2.tobytes() and np.array_equal(desc, desc2)


def test_oracle_regression_pins():
    """tests/golden/oracle_pins.json (tools/make_oracle_pins.py): the oracle's outputs on seeded inputs have not moved."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_oracle_pins", os.path.join(root, "tools", "make_oracle_pins.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(root, "tests", "golden", "oracle_pins.json")))
    got = mod.compute()
    def same(a, b):
        if isinstance(a, dict):
            return a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        if isinstance(a, list):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, float):
            return abs(a - b) <= 1e-6 * max(1.0, abs(a))  # libm may differ in the last bits between hosts
        return a == b

    for k in want:
        assert same(got[k], want[k]), (k, got[k], want[k])
