"""GPU parity of the HIP ORB extractor against the CPU oracle, through the C ABI.

Bar: bit-exact (integer / byte / index work; the float stages are defined contraction-free so they are
bit-exact too).  Stage-wise: pyramid levels, FAST candidates (x, y, score, raster order), final keypoints
(28-byte records) and 32-byte descriptors.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import swarmmap_amd
    assert swarmmap_amd.device_count() > 0, "these tests need a GPU"
    return swarmmap_amd


def _compare(S, oracle, img, nfeatures, ini=20, mn=7, nlevels=8, device_qt=True):
    ex = S.ORBextractor(nfeatures, 1.2, nlevels, ini, mn)
    kps, desc = ex(img)
    assert ex.quadtree_on_device == device_qt
    cfg = oracle.config(nfeatures, 1.2, nlevels, ini, mn)
    okps, odesc, ocands, olevels = oracle.extract(cfg, img, debug=True)
    for l in range(nlevels):
        assert np.array_equal(ex.level(l), olevels[l]), "pyramid level %d differs" % l
    for l in range(nlevels):
        xs, ys, sc = ex.candidates(l)
        oxs, oys, osc = ocands[l]
        assert len(xs) == len(oxs), "level %d: %d vs %d candidates" % (l, len(xs), len(oxs))
        assert np.array_equal(xs, oxs) and np.array_equal(ys, oys), "level %d candidate coords" % l
        assert np.array_equal(sc, osc), "level %d candidate scores" % l
    assert len(kps) == len(okps)
    for f in ("octave", "x", "y", "response", "size", "class_id"):
        assert np.array_equal(kps[f], okps[f]), f
    assert np.array_equal(kps["angle"], okps["angle"]), np.abs(kps["angle"] - okps["angle"]).max()
    assert kps.tobytes() == okps.tobytes()
    assert np.array_equal(desc, odesc)
    ex.close()
    return kps, desc


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_euroc_size_bit_exact(S, oracle, seed):
    from swarmmap_amd import synth
    kps, _ = _compare(S, oracle, synth.make_image(seed, synth.EUROC), 1000)
    assert len(kps) > 900


@pytest.mark.parametrize("nf", [2000, 4000])
def test_kitti_size_bit_exact(S, oracle, nf):
    from swarmmap_amd import synth
    kps, _ = _compare(S, oracle, synth.make_image(10 + nf, synth.KITTI), nf)
    assert len(kps) > 0.8 * nf


def test_init_extractor_2000_on_euroc(S, oracle):
    from swarmmap_amd import synth
    _compare(S, oracle, synth.make_image(5, synth.EUROC), 2000)


def test_noise_image_dense_corners(S, oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (480, 752)).astype(np.uint8)
    _compare(S, oracle, img, 1000)


def test_low_contrast_uses_low_threshold(S, oracle):
    from swarmmap_amd import synth
    img = synth.make_image(9, synth.EUROC).astype(np.float32)
    img = np.clip(110 + (img - 110) * 0.18, 0, 255).astype(np.uint8)  # contrast between minTh and iniTh
    kps, _ = _compare(S, oracle, img, 1000)
    assert len(kps) > 0


def test_half_textured_image_mixes_tile_thresholds(S, oracle):
    from swarmmap_amd import synth
    img = synth.make_image(4, synth.EUROC)
    soft = np.clip(110 + (img.astype(np.float32) - 110) * 0.15, 0, 255).astype(np.uint8)
    img[:, 376:] = soft[:, 376:]
    _compare(S, oracle, img, 1000)


def test_host_quadtree_paths_agree(S, oracle, monkeypatch):
    """The host tree serves quotas / aspect ratios the LDS tree does not hold; same output either way."""
    from swarmmap_amd import synth
    _compare(S, oracle, synth.make_image(31, synth.KITTI), 6000, device_qt=False)   # level-0 quota 1302 > 1020
    _compare(S, oracle, synth.make_canvas(32, 1000, 200), 800, device_qt=False)     # 6 root cells
    monkeypatch.setenv("SWARMORB_HOST_QUADTREE", "1")
    _compare(S, oracle, synth.make_image(2, synth.EUROC), 1000, device_qt=False)


def test_flat_image_no_keypoints(S, oracle):
    img = np.full((480, 752), 128, np.uint8)
    kps, desc = _compare(S, oracle, img, 1000)
    assert len(kps) == 0 and desc.shape == (0, 32)


def test_small_and_odd_sizes(S, oracle):
    from swarmmap_amd import synth
    for (w, h) in [(160, 120), (333, 217), (641, 479)]:
        _compare(S, oracle, synth.make_canvas(3, w, h), 300)


def test_strided_input_and_repeat_frames(S, oracle):
    from swarmmap_amd import synth
    big = synth.make_canvas(21, 900, 600)
    view = big[50:530, 100:852]  # non-contiguous rows (stride 900)
    ex = S.ORBextractor(1000, 1.2, 8, 20, 7)
    cfg = oracle.config(1000)
    for _ in range(3):  # the context is reused frame after frame
        kps, desc = ex(view)
        okps, odesc = oracle.extract(cfg, np.ascontiguousarray(view))
        assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    stream = synth.FrameStream(seed=3)
    for t in (0, 7, 19):
        f = stream.frame(t)
        kps, desc = ex(f)
        okps, odesc = oracle.extract(cfg, f)
        assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    ex.close()


def test_device_resident_input(S, oracle):
    import torch
    from swarmmap_amd import synth
    img = synth.make_image(8, synth.EUROC)
    d = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    ex = S.ORBextractor(1000, 1.2, 8, 20, 7)
    kps, desc = ex.run_device(d.data_ptr(), 752, 480, 752)
    okps, odesc = oracle.extract(oracle.config(1000), img)
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    ex.close()


def test_submit_collect_pipeline(S, oracle):
    """Asynchronous form: frame t+1 is submitted before frame t's results are used; identical output, one frame in
    flight, both quadtree placements."""
    from swarmmap_amd import synth
    stream = synth.FrameStream(seed=5)
    frames = [stream.frame(t) for t in range(4)]
    cfg = oracle.config(1000)
    want = [oracle.extract(cfg, f) for f in frames]
    for nf, frames_, want_ in ((1000, frames, want), (6000, [synth.make_image(31, synth.KITTI)], None)):
        ex = S.ORBextractor(nf, 1.2, 8, 20, 7)
        if want_ is None:
            want_ = [oracle.extract(oracle.config(nf), f) for f in frames_]
        ex.submit(frames_[0])
        for t in range(len(frames_)):
            kps, desc = ex.collect()
            kps, desc = kps.copy(), desc.copy()
            if t + 1 < len(frames_):
                ex.submit(frames_[t + 1])  # in flight while frame t is "tracked"
            assert kps.tobytes() == want_[t][0].tobytes() and np.array_equal(desc, want_[t][1])
        with pytest.raises(S.SwarmOrbError):
            ex.collect()  # nothing submitted
        ex.submit(frames_[0])
        with pytest.raises(S.SwarmOrbError):
            ex.submit(frames_[0])  # one frame in flight per extractor
        with pytest.raises(S.SwarmOrbError):
            ex(frames_[0])
        ex.collect()
        ex.close()


def test_errors(S):
    from swarmmap_amd import synth
    ex = S.ORBextractor(500, 1.2, 8, 20, 7)
    kps, desc = ex(np.zeros((0, 0), np.uint8))  # empty image: nothing happens (ORBextractor.cc:750)
    assert len(kps) == 0
    ex(synth.make_canvas(1, 320, 240))
    with pytest.raises(S.SwarmOrbError):
        ex(synth.make_canvas(1, 400, 240))  # size change is an error, not silent corruption
    ex.close()
    with pytest.raises(S.SwarmOrbError):
        S.ORBextractor(500, 1.2, 9, 20, 7)  # > SO_MAX_LEVELS
    t = S.ORBextractor(1000, 1.2, 8, 20, 7)
    assert t.GetLevels() == 8 and abs(t.GetScaleFactors()[7] - 1.2 ** 7) < 1e-5
    assert t.GetFeaturesPerLevel().tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    t.close()


@pytest.mark.parametrize("fuse_from", [0, 2])
def test_fused_pyramid_option_is_bit_identical(fuse_from):
    """pyramid_fused_kernel (SWARMORB_PYRAMID_FUSE_FROM, read once per process: hence the child process) produces the
    same level bytes as the chained resize launches, i.e. as the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, swarmmap_amd\n"
            "from swarmmap_amd import synth\n"
            "from oracle import oracle_py as orc\n"
            "for size, nf in ((synth.EUROC, 1000), (synth.KITTI, 2000), ((377, 241), 500)):\n"
            "    img = synth.make_canvas(4, size[0], size[1])\n"
            "    ex = swarmmap_amd.ORBextractor(nf, 1.2, 8, 20, 7)\n"
            "    k, d = ex(img)\n"
            "    ok, od, _, lv = orc.extract(orc.config(nf), img, debug=True)\n"
            "    assert all(np.array_equal(ex.level(l), lv[l]) for l in range(8))\n"
            "    assert k.tobytes() == ok.tobytes() and np.array_equal(d, od)\n"
            "    ex.close()\n"
            "print('ok')\n" % root)
    env = dict(os.environ, SWARMORB_PYRAMID_FUSE_FROM=str(fuse_from))
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True)
    assert out.strip().endswith("ok")
