#!/usr/bin/env python3
"""A local-BA window as the CLOSED loop builds it (swarmmap_amd/closedloop.py local_window: the keyframe's own covisible
keyframes, their points, the other observers fixed), taken from the loop run over the CPU oracle on the bench's stream:
the window of the keyframe at frame 300.  tools/lba_bench.py and tests time / check so_bundle_adjust on it
(tests/golden/closed_loop_window.npz)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from swarmmap_amd import closedloop, synth  # noqa: E402
from swarmmap_amd.replay import make_vocabulary  # noqa: E402
from trajectory_common import OracleBackend  # noqa: E402

K = synth.EUROC_K
st = synth.FrameStream(seed=20221001, size=synth.EUROC, K=K, dist=synth.EUROC_DIST)
grabbed = {}
orig = closedloop.local_window


def spy(M, c, n_free, n_fixed):
    w = orig(M, c, n_free, n_fixed)
    if w is not None and c["t"] == 300:
        prob = dict(w[0])
        prob["intr"] = np.tile(np.asarray(K, np.float32), (len(w[1]), 1))
        grabbed.update(prob)
    return w


closedloop.local_window = spy
closedloop.track(OracleBackend(K, 1000, synth.EUROC_DIST), st, 302, K, make_vocabulary(), plane_z=2.0, third_pose=False)
assert grabbed, "no window at frame 300"
out = os.path.join(ROOT, "tests", "golden", "closed_loop_window.npz")
np.savez_compressed(out, **{k: np.asarray(v) for k, v in grabbed.items()})
print(out, {k: np.asarray(v).shape for k, v in grabbed.items()}, os.path.getsize(out))
