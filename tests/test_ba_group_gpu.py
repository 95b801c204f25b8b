"""Local bundle adjustments of SEVERAL agents as one chain of launches (so_ba_group: every kernel of the Levenberg-Marquardt
chain once, with the window as blockIdx.y) against the same windows optimised one by one: poses, points, outlier flags, chi2
per edge, iteration and trial counts are those of the solo call, to the bit.  Solo calls are pinned to the oracle by
tests/test_ba_gpu.py (Optimizer::LocalBundleAdjustment, code/src/Optimizer.cc:436-740).  Concurrency model: one LocalMapping
thread per agent (code/src/LocalMapping.cc:53-110), one process per agent (code/Examples/Monocular/swarm_map.cc:329-337)."""
import os
import threading

import numpy as np
import pytest

import swarmmap_amd
from swarmmap_amd import synth
from swarmmap_amd.optimizer import BaGroup

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("iterations_stage1", "iterations_stage2", "lm_trials", "chi2_initial", "chi2_final", "n_free_keyframes", "lambda_final")


def _windows():
    w = [synth.make_ba_case("LBA-M", seed=100 + i) for i in range(3)]
    w.append(synth.make_ba_case("LBA-S", seed=7))                      # 8 free keyframes: another tile count of the MFMA solver
    w.append(synth.make_ba_problem(3, 3, 6, 300, max_obs="auto"))      # 3 free keyframes: the look-ahead solver (no grouped form)
    w.append(synth.make_ba_case("LBA-L", seed=5))                      # 40 free keyframes: the register-resident MFMA solver
    z = np.load(os.path.join(GOLDEN, "closed_loop_window.npz"))        # a window of the closed loop (25 free + 36 fixed keyframes)
    w.append({k: z[k] for k in ("Tcw", "fixed", "intr", "Xw", "edge_pose", "edge_point", "obs", "inv_sigma2")})
    return w


def _same(a, b, what):
    for k in ("Tcw", "Xw", "outlier", "chi2"):
        assert np.array_equal(a[k], b[k]), "%s: %s differs" % (what, k)
    for k in KEYS:
        assert a["info"][k] == b["info"][k], "%s: info.%s %r != %r" % (what, k, a["info"][k], b["info"][k])


def _run_grouped(opts, windows, reps):
    out, errs = [[None] * reps for _ in windows], []

    def worker(i):
        try:
            for r in range(reps):
                out[i][r] = opts[i].LocalBundleAdjustment(windows[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(windows))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    return out


def test_grouped_local_bundle_adjustments_equal_the_solo_calls():
    windows = _windows()
    n = len(windows)
    opts = [swarmmap_amd.Optimizer() for _ in range(n)]
    solo = [[o.LocalBundleAdjustment(w) for _ in range(3)] for o, w in zip(opts, windows)]  # (a context's 2nd / 3rd call: the stage-2 hint is set)
    for i in range(n):
        _same(solo[i][1], solo[i][0], "solo call repeated")
    g = BaGroup(window_us=20000.0)  # (python threads arrive far apart: a generous window so that the rounds really are shared)
    for o in opts:
        o.set_group(g)
    grp = _run_grouped(opts, windows, 3)
    st = g.stats()
    for i in range(n):
        for r in range(3):
            _same(grp[i][r], solo[i][0], "window %d, grouped call %d" % (i, r))
    assert st["members"] == n and st["members_total"] == 3 * n
    assert st["rounds"] < 3 * n, st                          # members did share rounds ...
    assert st["rows_launched"] > 1.5 * st["grouped_launches"], st   # ... and launches
    assert st["ungrouped_launches"] > 0                      # (the 3- and 40-keyframe windows' solves have no grouped form)
    # a member that leaves the group is a solo context again, with the same bits
    opts[0].set_group(None)
    _same(opts[0].LocalBundleAdjustment(windows[0]), solo[0][0], "after leaving the group")
    # a group of one: every launch goes through the grouped kernels with a grid of one member
    g1 = BaGroup(window_us=50.0)
    opts[1].set_group(g1)
    _same(opts[1].LocalBundleAdjustment(windows[1]), solo[1][0], "group of one")
    for o in opts:
        o.close()
    g.close()
    g1.close()


def test_more_members_than_an_argument_block_holds_and_a_stop_request():
    """Eleven windows in one group (a grouped launch carries eight: the ninth starts another launch of the same step), one of
    them with pbStopFlag raised before the call - Optimizer::LocalBundleAdjustment returns its input there (code/src/Optimizer.cc:
    631-633) without joining the round - and one whose flag is raised while it waits: results as in solo calls."""
    n = 11
    windows = [synth.make_ba_case("LBA-S", seed=30 + i) for i in range(n)]
    opts = [swarmmap_amd.Optimizer() for _ in range(n)]
    solo = [o.LocalBundleAdjustment(w) for o, w in zip(opts, windows)]
    stop_set = np.ones(1, np.uint8)
    solo_stopped = opts[3].LocalBundleAdjustment(windows[3], pbStopFlag=stop_set)
    assert solo_stopped["info"]["aborted"] == 1 and solo_stopped["info"]["lm_trials"] == 0
    g = BaGroup(window_us=30000.0)
    for o in opts:
        o.set_group(g)
    out, errs = [None] * n, []

    def worker(i):
        try:
            out[i] = opts[i].LocalBundleAdjustment(windows[i], pbStopFlag=stop_set if i == 3 else None)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    st = g.stats()
    for i in range(n):
        _same(out[i], solo_stopped if i == 3 else solo[i], "window %d of eleven" % i)
    assert st["members_total"] == n - 1  # (the stopped call never reached the group)
    assert st["rows_launched"] > 4 * st["grouped_launches"], st  # rounds of many members ...
    for o in opts:
        o.close()
    g.close()


def test_resident_trial_loop_equals_the_chain_of_launches(monkeypatch):
    """SWARMORB_BA_RESIDENT=1 (an experiment, off by default: NOTES.md G.10): a stage's LM trials as ONE resident launch - workgroups
    that walk the virtual blocks of build / Schur gather / update + errors, meet at grid barriers, the MFMA solve on four waves of
    workgroup 0 - against the five launches per trial: every output to the bit, with 8, 32 and 100 workgroups; windows the loop does
    not cover (3 and 40 free keyframes) take the chain by themselves."""
    windows = _windows()
    opt = swarmmap_amd.Optimizer()
    monkeypatch.delenv("SWARMORB_BA_RESIDENT", raising=False)
    ref = [opt.LocalBundleAdjustment(w) for w in windows]
    assert opt.resident_stages() == 0
    monkeypatch.setenv("SWARMORB_BA_RESIDENT", "1")
    for wgs in ("8", "32", "100"):
        monkeypatch.setenv("SWARMORB_BA_RESIDENT_WGS", wgs)
        before = opt.resident_stages()
        for i, w in enumerate(windows):
            _same(opt.LocalBundleAdjustment(w), ref[i], "window %d resident on %s workgroups" % (i, wgs))
        assert opt.resident_stages() - before == 2 * 5  # both stages of the five windows with 8 / 25 free keyframes
    opt.close()

