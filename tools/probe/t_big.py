import sys, time
sys.path.insert(0, '.')
import numpy as np
import swarmmap_amd
from swarmmap_amd import synth
o = swarmmap_amd.Optimizer()
for nf, npts in ((120, 6000), (299, 30000)):
    p = synth.make_ba_problem(200 + nf, nf, 1, npts, max_obs="auto")
    t0 = time.perf_counter()
    r = o.BundleAdjustment(p, nIterations=3, bRobust=True)
    print(nf, "path", r["info"]["solver_path"], "tiles", r["info"]["nnz_tiles"], "ms %.2f" % ((time.perf_counter() - t0) * 1e3), "chi2", r["info"]["chi2_final"], "solve_ms", r["info"]["solve_ms"] / max(r["info"]["n_solves"], 1), flush=True)
