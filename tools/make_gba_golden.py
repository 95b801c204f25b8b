"""Golden fixtures for whole-map bundle adjustment at BASELINE configs[4] size (code/src/Optimizer.cc:42-237): the CPU
oracle (oracle/ba_oracle.c: the line-cited restatement of g2o's Levenberg / Schur / Cholesky path) run ONCE in the
build container on the full-size maps - GBA-2 (1499 keyframes looking at one cloud, 780 k observations) and GBA-2r (the
8-agent street-grid map, 1503 keyframes, 710 k observations) - because it takes minutes per map, not seconds.
    python tools/make_gba_golden.py GBA-2        # -> tests/golden/gba2.npz       (~15 min on one core)
    python tools/make_gba_golden.py GBA-2r       # -> tests/golden/gba2r.npz      (~10 min)
    python tools/make_gba_golden.py --no-robust GBA-2 GBA-2r   # -> gba2_norobust.npz, gba2r_norobust.npz
--no-robust is the reference's own server-side call: GlobalBundleAdjustemnt(map, 10, &stop, kf, false)
(code/src/MediatorScheduler.cc:122, code/src/LoopClosing.cc:606) - no Huber kernel; the default (Huber on) is the
function's default argument (code/include/Optimizer.h).
The fixture holds what tests/test_ba_gpu.py compares so_bundle_adjust with: every optimised pose, every 8th point,
the outlier flags (packed), chi2 before / after, iteration counts - plus a digest of the generated problem, so a
drifting generator is noticed instead of compared.  Inputs are regenerated in the test from the same seed
(swarmmap_amd.synth.make_ba_case(name, 1)); nothing of /root/reference is read."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py  # noqa: E402
from swarmmap_amd import synth  # noqa: E402

POINT_STRIDE = 8
FILES = {"GBA-2": "gba2.npz", "GBA-2r": "gba2r.npz", "GBA-1": "gba1.npz"}


def problem_digest(p):
    h = hashlib.sha256()
    for k in ("Tcw", "Xw", "obs", "edge_pose", "edge_point", "inv_sigma2", "fixed"):
        h.update(np.ascontiguousarray(p[k]).tobytes())
    return h.hexdigest()


def main():
    args = sys.argv[1:]
    robust = "--no-robust" not in args
    for name in [a for a in args if not a.startswith("--")]:
        p = synth.make_ba_case(name, 1)
        t0 = time.perf_counter()
        # Optimizer::BundleAdjustment(..., nIterations = 10, bRobust): one optimize(10), thHuber2D = sqrt(5.99) when robust
        o = oracle_py.bundle_adjust(p, its1=10, its2=0, robust=robust, huber_delta=np.float32(np.sqrt(np.float32(5.99))))
        dt = time.perf_counter() - t0
        out = os.path.join(ROOT, "tests", "golden", FILES[name] if robust else FILES[name].replace(".npz", "_norobust.npz"))
        inf = o["info"]
        np.savez_compressed(out, name=name, seed=1, robust=int(robust), digest=problem_digest(p), Tcw=o["Tcw"].astype(np.float32),
                            Xw_every8=o["Xw"][::POINT_STRIDE].astype(np.float32), outlier_bits=np.packbits(o["outlier"].astype(np.uint8)),
                            n_edges=len(p["edge_pose"]), chi2=np.asarray(o["chi2"], np.float64)[::64],
                            info_keys=np.array(sorted(inf.keys())), info_vals=np.array([float(inf[k]) for k in sorted(inf.keys())]),
                            oracle_seconds=dt)
        print(name, "oracle %.0f s" % dt, {k: inf[k] for k in sorted(inf.keys())}, "->", out, os.path.getsize(out), "bytes", flush=True)


if __name__ == "__main__":
    main()
