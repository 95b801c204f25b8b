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
    print(name, "free", int((w["fixed"] == 0).sum()), "edges", len(w["edge_pose"]), "ms %.3f" % (np.median(ts) * 1e3),
          "trials", r["info"]["lm_trials"], "chi2 %.6e" % r["info"]["chi2_final"], flush=True)


def window(nf):
    return synth.make_ba_problem(0, nf, (3 * nf) // 2, 150 * nf, max_obs="auto")


if "--only" in sys.argv:
    nf = int(sys.argv[sys.argv.index("--only") + 1])
    run("window", window(nf), reps=5)
elif "--sweep" in sys.argv:
    for nf in (29, 30, 40, 43, 44, 48, 64, 80, 96, 128):
        run("window", window(nf))
else:
    for name in ("LBA-S", "LBA-M", "LBA-L"):
        run(name, synth.make_ba_case(name))
