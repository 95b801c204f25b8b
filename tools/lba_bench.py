"""Wall time of Optimizer::LocalBundleAdjustment on the synthetic windows of SURVEY.md 8d (developer tool, GPU box):
python tools/lba_bench.py            (SWARMORB_BA_NO_MFMA_SOLVER=1 for the register solvers)"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from swarmmap_amd import synth
from swarmmap_amd.optimizer import Optimizer
o = Optimizer()
for name in ("LBA-S", "LBA-M", "LBA-L"):
    w = synth.make_ba_case(name)
    for _ in range(3):
        r = o.LocalBundleAdjustment(w)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); r = o.LocalBundleAdjustment(w); ts.append(time.perf_counter() - t0)
    print(name, "free", int((w["fixed"] == 0).sum()), "edges", len(w["edge_pose"]), "ms %.3f" % (np.median(ts) * 1e3),
          "trials", r["info"]["lm_trials"], "chi2 %.6e" % r["info"]["chi2_final"])
