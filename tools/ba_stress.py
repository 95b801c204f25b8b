"""Stress of bundle adjustment's determinism: the same windows solved many times must give the same bytes every time
(the whole call is one enqueue of gated launches; the dataflow solve's workgroups meet through flags).  GPU box.
    python tools/ba_stress.py [repetitions, default 200]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swarmmap_amd import synth  # noqa: E402
from swarmmap_amd.optimizer import Optimizer  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
o = Optimizer()
cases = [("LBA-M", synth.make_ba_case("LBA-M")), ("64", synth.make_ba_problem(0, 64, 96, 9600, max_obs="auto", max_yaw=0.6)),
         ("rejecting", synth.make_ba_problem(4, 8, 10, 800, pose_noise=(0.6, 15.0), point_noise=0.4, max_obs="auto"))]
bad = 0
for name, w in cases:
    r0 = o.LocalBundleAdjustment(w)
    ref = (r0["Tcw"].tobytes(), r0["Xw"].tobytes(), r0["outlier"].tobytes(), r0["info"]["lm_trials"], r0["info"]["chi2_final"])
    for k in range(reps):
        r = o.LocalBundleAdjustment(w)
        got = (r["Tcw"].tobytes(), r["Xw"].tobytes(), r["outlier"].tobytes(), r["info"]["lm_trials"], r["info"]["chi2_final"])
        if got != ref:
            bad += 1
            print("MISMATCH", name, k, r["info"]["lm_trials"], r["info"]["chi2_final"], flush=True)
    print(name, "trials", r0["info"]["lm_trials"], "chi2 %.6e" % r0["info"]["chi2_final"], "repetitions", reps, flush=True)
print("mismatches", bad)
sys.exit(1 if bad else 0)
