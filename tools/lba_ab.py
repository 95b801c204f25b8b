"""Wall and GPU time (so_ba_info.gpu_ms: HIP events around the whole solve) of LocalBundleAdjustment, 40 calls: an A/B
tool - run it with and without a switch (SWARMORB_BA_NO_CHAIN=1: the host sees every stage end).  GPU box."""
import sys, time
sys.path.insert(0,'.')
import numpy as np
from swarmmap_amd import synth
from swarmmap_amd.optimizer import Optimizer
o = Optimizer()
for name, w in (("64", synth.make_ba_problem(0, 64, 96, 9600, max_obs="auto", max_yaw=0.6)), ("LBA-M", synth.make_ba_case("LBA-M"))):
    for _ in range(5): o.LocalBundleAdjustment(w)
    ts, gs = [], []
    for _ in range(40):
        t0 = time.perf_counter(); r = o.LocalBundleAdjustment(w); ts.append(time.perf_counter() - t0); gs.append(r["info"]["gpu_ms"])
    print(name, "wall median %.3f min %.3f | gpu_ms median %.3f min %.3f" % (np.median(ts)*1e3, np.min(ts)*1e3, np.median(gs), np.min(gs)))
