"""Wall time of Optimizer::LocalBundleAdjustment on the synthetic windows of SURVEY.md 8d (developer tool, GPU box):
python tools/lba_bench.py            (SWARMORB_BA_NO_MFMA_SOLVER=1 for the register solvers)
python tools/lba_bench.py --sweep    windows of 30..128 free keyframes (150 points and 1.5 fixed keyframes per free one,
                                     the proportions of LBA-L): the solver changes at 29|30, 43|44
python tools/lba_bench.py --only 64  one window size (for rocprofv3 --kernel-trace --stats)"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from swarmmap_amd import synth
from swarmmap_amd.optimizer import Optimizer
o = Optimizer()


def run(name, w, reps=10):
    for _ in range(3):
        r = o.LocalBundleAdjustment(w)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = o.LocalBundleAdjustment(w); ts.append(time.perf_counter() - t0)
    o.set_solve_timing(True)  # one extra call with HIP events around the solves (not part of the wall time above)
    i = o.LocalBundleAdjustment(w)["info"]
    o.set_solve_timing(False)
    print(name, "free", int((w["fixed"] == 0).sum()), "with a Hessian index", i["n_free_keyframes"], "tiles", int(i["nnz_tiles"]),
          "edges", len(w["edge_pose"]), "ms %.3f" % (np.median(ts) * 1e3), "solve us %.1f" % (1e3 * i["solve_ms"] / max(i["n_solves"], 1)),
          "trials", i["lm_trials"], "chi2 %.6e" % i["chi2_final"], flush=True)


def window(nf):
    # max_yaw: every free keyframe observes points (without it the keyframes at the ends of a long arc look away from the
    # cloud and drop out of the reduced system: the "128-keyframe" window of round 2's sweep had 78 keyframes in it and
    # the "96-keyframe" one 86 - five panels against six, the whole of the inversion profiles/r2_lba_sweep.txt shows)
    return synth.make_ba_problem(0, nf, (3 * nf) // 2, 150 * nf, max_obs="auto", max_yaw=0.6)


if "--only" in sys.argv:
    nf = int(sys.argv[sys.argv.index("--only") + 1])
    run("window", window(nf), reps=5)
elif "--sweep" in sys.argv:
    for nf in (29, 30, 40, 43, 44, 48, 64, 80, 96, 128):
        run("window", window(nf))
else:
    for name in ("LBA-S", "LBA-M", "LBA-L"):
        run(name, synth.make_ba_case(name))
