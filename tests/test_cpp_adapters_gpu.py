"""The C++ adapter classes (reference signatures: ORBextractor / ORBmatcher / Optimizer) compile against the
C ABI and give the same results as the Python binding on the same inputs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "swarmmap_amd", "host")


def _fnv(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def _build(tmp_path):
    exe = str(tmp_path / "adapter_smoke")
    srcs = [os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp")] + \
           [os.path.join(HOST, f) for f in ("ORBextractor.cc", "ORBmatcher.cc", "Optimizer.cc", "Frame.cc")]
    lib = os.path.join(ROOT, "swarmmap_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-o", exe] + srcs +
                          ["-L" + lib, "-lswarmorb", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_adapters_compile_with_plain_gcc(tmp_path):
    """Host adapters are plain C++14 over the C ABI: g++ (no hipcc, no OpenCV, no Eigen) must build them."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_adapters_match_python_binding(tmp_path):
    import swarmmap_amd
    from swarmmap_amd import synth
    exe = _build(tmp_path)
    img = synth.make_image(12, synth.EUROC)
    raw = tmp_path / "img.raw"
    raw.write_bytes(img.tobytes())
    out = subprocess.check_output([exe, str(raw), "752", "480"], text=True)
    lines = dict(l.split(" ", 1) for l in out.strip().splitlines())
    ex = swarmmap_amd.ORBextractor(1000, 1.2, 8, 20, 7)
    kps, desc = ex(img)
    n, h = lines["keypoints"].split()
    assert int(n) == len(kps) and h == _fnv(kps.tobytes())
    n, h = lines["descriptors"].split()
    assert int(n) == len(desc) and h == _fnv(desc.tobytes())
    assert int(lines["distance"]) == int(np.unpackbits(desc[0] ^ desc[1]).sum())
    nm, self_m, lvl0 = [int(v) for v in lines["init_matches"].split()[::2]]
    assert self_m == nm and nm > 0.8 * lvl0  # level-0 keypoints re-find themselves
    assert "its" in lines["ba"] and float(lines["ba"].split()[3]) < 1e-3 * float(lines["ba"].split()[1])
    # Frame post-processing, PoseOptimization, distinctive descriptors through the C++ adapters
    fp = swarmmap_amd.FramePostProcessor(synth.EUROC_K, synth.EUROC_DIST)
    pr = fp.prepare(np.stack([kps["x"], kps["y"]], 1), 752, 480)
    f = lines["undistorted"].split()
    assert int(f[0]) == len(kps) and f[1] == _fnv(pr["xy_un"].tobytes())
    assert [np.float32(v) for v in f[3:7]] == pr["bounds"].tolist()
    assert int(f[8]) == len(pr["cell_items"]) and f[9] == _fnv(pr["cell_items"].tobytes())
    p = lines["pose"].split()
    assert int(p[1]) == 40 and max(abs(float(v)) for v in p[3:6]) < 1e-3  # converges back to the identity pose
    m = swarmmap_amd.ORBmatcher()
    idx, _ = m.ComputeDistinctiveDescriptors([0, 3, 3, 8], desc[:8])
    assert [int(v) for v in lines["distinctive"].split()] == idx.tolist()
