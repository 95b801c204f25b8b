"""Global bundle adjustment (BASELINE configs[4]: the whole-map optimisation after a merge / loop closure) on one GPU:
wall time per BundleAdjustment(10 iterations) and the FP64 rate of the blocked reduced-camera-system solve.
    python tools/gba_bench.py            # GBA-1 and GBA-2 (SURVEY 8d sizes), JSON lines
    python tools/gba_bench.py max        # + the largest map the blocked solver accepts (2047 free keyframes)
Not part of bench.py's per-frame metric; the numbers are quoted in DESIGN.md 5."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swarmmap_amd  # noqa: E402
from swarmmap_amd import synth  # noqa: E402

FP64_PEAK_TF = 78.6

def main():
    o = swarmmap_amd.Optimizer()
    o.set_solve_timing(True)
    cases = ["GBA-1", "GBA-2", "GBA-1r", "GBA-2r"] + (["GBA-max", "GBA-4k"] if "max" in sys.argv[1:] else [])
    named = [a for a in sys.argv[1:] if a.startswith("GBA-")]  # python tools/gba_bench.py GBA-1 GBA-1r: those only
    if named:
        cases = named
    for name in cases:
        if name == "GBA-max":  # the old dense limit: 12288 reduced-system rows = 2048 keyframes, one of them fixed
            p = synth.make_ba_problem(1, n_free=2047, n_fixed=1, n_points=160000, max_obs="auto")
        else:
            p = synth.make_ba_case(name, 1)
        # the reference's server-side call has bRobust = false (code/src/MediatorScheduler.cc:122, LoopClosing.cc:606);
        # "huber" on the command line times the function's default (Huber on) instead
        robust = "huber" in sys.argv[1:]
        o.BundleAdjustment(p, nIterations=2, bRobust=robust)  # warm-up: buffers
        t0 = time.perf_counter()
        r = o.BundleAdjustment(p, nIterations=10, bRobust=robust)
        wall = time.perf_counter() - t0
        inf = r["info"]
        n = 6 * int(inf["n_free_keyframes"])  # keyframes the solver gave a hessian index (not fixed AND observed)
        flop = n ** 3 / 3.0 + 2.0 * n ** 2
        sflop = inf["solve_gflop_structural"] * 1e9
        ms = inf["solve_ms"] / max(inf["n_solves"], 1)
        T = (n + 95) // 96
        print(json.dumps({"case": name, "robust": robust, "free_keyframes": n // 6, "keyframes_not_fixed": int((p["fixed"] == 0).sum()), "points": int(len(p["Xw"])), "edges": int(len(p["edge_pose"])),
                          "wall_ms": wall * 1e3, "gpu_ms": inf["gpu_ms"], "lm_trials": inf["lm_trials"],
                          "chi2_initial": inf["chi2_initial"], "chi2_final": inf["chi2_final"],
                          "solve": {"n": n, "ms_per_solve": ms, "tiles_in_skyline": inf["nnz_tiles"], "tiles_dense": T * (T + 1) // 2,
                                    "structural_flop": sflop, "dense_flop": flop,
                                    "achieved_tflops": min(sflop, flop) / (ms * 1e-3) / 1e12,
                                    "skyline_tiles_tflops": sflop / (ms * 1e-3) / 1e12,
                                    "dense_equivalent_tflops": flop / (ms * 1e-3) / 1e12,
                                    "peak_tflops": FP64_PEAK_TF, "frac": min(sflop, flop) / (ms * 1e-3) / 1e12 / FP64_PEAK_TF, "bound": "mfma"}}), flush=True)
    o.close()

if __name__ == "__main__":
    main()
