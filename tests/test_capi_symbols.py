"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/swarmorb.h declares, and
fails loudly (no CPU fallback) when asked to compute without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "swarmorb.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(so_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_entry_points():
    syms = _declared_symbols()
    for must in ("so_extractor_create", "so_extractor_run", "so_extractor_run_device", "so_extractor_destroy",
                 "so_extractor_tables", "so_status_string", "so_device_count"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import swarmmap_amd
    lib = ctypes.CDLL(swarmmap_amd.library_path())
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_gpu():
    import swarmmap_amd
    if swarmmap_amd.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(swarmmap_amd.SwarmOrbError):
        swarmmap_amd.ORBextractor(1000, 1.2, 8, 20, 7)


def test_product_does_not_import_oracle():
    """The product package must never reach into oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "swarmmap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".cc")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "orb_oracle" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_keyframe_record_round_trip_and_corruption():
    """so_keyframe_record_{size,pack,unpack}: host-side format functions, exercised without a GPU."""
    import numpy as np
    import pytest
    import swarmmap_amd
    from swarmmap_amd.parallel import pack_keyframe_record, unpack_keyframe_record
    rng = np.random.default_rng(5)
    for n in (0, 1, 1000):
        xy = rng.uniform(0, 752, (n, 2)).astype(np.float32)
        ang = rng.uniform(0, 360, n).astype(np.float32)
        octv = rng.integers(0, 8, n).astype(np.int32)
        desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
        T = rng.normal(size=12).astype(np.float32)
        rec = pack_keyframe_record(3, 2 ** 40 + 7, 12.5, T, (1, 2, 3, 4), xy, ang, octv, desc)
        assert rec.nbytes == 128 + 48 * n and rec[:4].tobytes() == b"SOKF"
        assert np.array_equal(rec[128:128 + 32 * n].reshape(-1, 32), desc)  # descriptors in place, on a 32-byte row
        u = unpack_keyframe_record(rec)
        assert u["agent_id"] == 3 and u["keyframe_id"] == 2 ** 40 + 7 and u["timestamp"] == 12.5
        assert np.array_equal(u["xy"], xy) and np.array_equal(u["angle"], ang) and np.array_equal(u["octave"], octv)
        assert np.array_equal(u["desc"], desc) and np.array_equal(u["Tcw"], T) and u["K"].tolist() == [1, 2, 3, 4]
        if n:
            bad = rec.copy()
            bad[130] ^= 1  # a flipped payload bit is caught by the checksum
            with pytest.raises(swarmmap_amd.SwarmOrbError):
                unpack_keyframe_record(bad)
    with pytest.raises(swarmmap_amd.SwarmOrbError):
        unpack_keyframe_record(np.zeros(128, np.uint8))


def test_keyframe_record_v2_round_trip_and_corruption():
    """Version 2 = version 1 + the map-point id of every keypoint: same descriptor / geometry blocks at the same
    offsets, ids behind them, length a multiple of 32, header counts the bound keypoints; version-1 records stay
    readable by the version-2 reader (every keypoint counts as bound)."""
    import numpy as np
    import pytest
    import swarmmap_amd
    from swarmmap_amd.kfstore import pack_keyframe_record2, record_size2, unpack_keyframe_record2
    from swarmmap_amd.parallel import pack_keyframe_record
    rng = np.random.default_rng(6)
    for n in (0, 1, 7, 1000):
        xy = rng.uniform(0, 752, (n, 2)).astype(np.float32)
        ang = rng.uniform(0, 360, n).astype(np.float32)
        octv = rng.integers(0, 8, n).astype(np.int32)
        desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
        mp = np.where(rng.random(n) < 0.5, rng.integers(0, 1 << 30, n), -1).astype(np.int32)
        T = rng.normal(size=12).astype(np.float32)
        rec = pack_keyframe_record2(3, 2 ** 40 + 7, 12.5, T, (1, 2, 3, 4), xy, ang, octv, desc, mp)
        assert rec.nbytes == record_size2(n) == (128 + 52 * n + 31) // 32 * 32
        v1 = pack_keyframe_record(3, 2 ** 40 + 7, 12.5, T, (1, 2, 3, 4), xy, ang, octv, desc)
        assert np.array_equal(rec[128:128 + 48 * n], v1[128:])  # descriptor and geometry blocks are version 1's
        assert np.array_equal(rec[128 + 48 * n:128 + 52 * n].view(np.int32), mp) and not rec[128 + 52 * n:].any()
        u = unpack_keyframe_record2(rec)
        assert u["version"] == 2 and u["n_map_points"] == int((mp >= 0).sum()) and u["agent_id"] == 3
        for k, want in (("xy", xy), ("angle", ang), ("octave", octv), ("desc", desc), ("map_point_id", mp), ("Tcw", T)):
            assert np.array_equal(u[k], want), k
        u1 = unpack_keyframe_record2(v1)
        assert u1["version"] == 1 and (u1["map_point_id"] == 0).all() and np.array_equal(u1["desc"], desc)
        if n:
            bad = rec.copy()
            bad[128 + 48 * n] ^= 1  # a flipped binding is caught by the checksum
            with pytest.raises(swarmmap_amd.SwarmOrbError):
                unpack_keyframe_record2(bad)


def test_replay_library_loads_and_exports():
    """The C++ host loop of bench.py (swarmmap_amd/host/replay.cc) is built next to the C-ABI library."""
    import ctypes
    import os
    import swarmmap_amd
    swarmmap_amd.load_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = ctypes.CDLL(os.path.join(root, "swarmmap_amd", "libswarmorb_replay.so"))
    for name in ("so_replay_create", "so_replay_destroy", "so_replay_set_frames", "so_replay_set_window",
                 "so_replay_preallocate", "so_replay_prime", "so_replay_run", "so_replay_drain", "so_replay_finish",
                 "so_replay_stats", "so_replay_log", "so_replay_log_size", "so_replay_frame_ms", "so_replay_last_frame", "so_replay_last_dframe",
                 "so_replay_last_bindings"):
        assert hasattr(lib, name), name
