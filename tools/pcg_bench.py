#!/usr/bin/env python3
"""Direct block-skyline Cholesky vs block-Jacobi PCG (so_ba_set_linear_solver) on the whole-map bundle adjustments of
BASELINE configs[4]: wall / GPU time of BundleAdjustment(10 iterations, bRobust = false), time per solve (HIP events),
PCG iterations, and how far the two results are apart."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swarmmap_amd  # noqa: E402
from swarmmap_amd import synth  # noqa: E402


def run(o, p):
    o.BundleAdjustment(p, nIterations=2, bRobust=False)
    ts, best = [], None
    for _ in range(3):
        t0 = time.perf_counter()
        r = o.BundleAdjustment(p, nIterations=10, bRobust=False)
        ts.append(time.perf_counter() - t0)
        best = r
    o.set_solve_timing(True)
    it = o.BundleAdjustment(p, nIterations=10, bRobust=False)["info"]
    o.set_solve_timing(False)
    inf = best["info"]
    return best, {"wall_ms": float(np.median(ts)) * 1e3, "gpu_ms": inf["gpu_ms"], "lm_trials": inf["lm_trials"], "chi2_final": inf["chi2_final"],
                  "solver_path": inf["solver_path"], "ms_per_solve": it["solve_ms"] / max(it["n_solves"], 1),
                  "pcg_iterations": inf["pcg_iterations"], "pcg_iterations_per_solve": inf["pcg_iterations"] / max(inf["lm_trials"], 1),
                  "nnz": inf["nnz_tiles"]}


def main():
    cases = sys.argv[1].split(",") if len(sys.argv) > 1 else ["GBA-1", "GBA-1r", "GBA-2r", "GBA-2"]
    tols = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1e-7]
    d = swarmmap_amd.Optimizer()
    for name in cases:
        p = synth.make_ba_case(name, 1)
        rd, di = run(d, p)
        rec = {"case": name, "edges": int(len(p["edge_pose"])), "direct": di, "pcg": {}}
        for tol in tols:
            q = swarmmap_amd.Optimizer()
            q.set_linear_solver("pcg", tol, 2000)
            rp, pi = run(q, p)
            pi["max_pose_entry_difference_to_direct"] = float(np.abs(rp["Tcw"] - rd["Tcw"]).max())
            pi["max_point_difference_to_direct"] = float(np.abs(rp["Xw"] - rd["Xw"]).max())
            rec["pcg"]["%g" % tol] = pi
            q.close()
        print(json.dumps(rec), flush=True)
    d.close()


if __name__ == "__main__":
    main()
