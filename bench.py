#!/usr/bin/env python3
"""bench.py — frames/s per agent of the per-frame hot path (tracking + local BA) on synthetic EuRoC-sized streams,
one agent per GPU.

Contract (task prompt): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU, RCCL).
A step = ONE TRACKED FRAME of one agent, chained and device-resident (swarmmap_amd/host/replay.cc over the C ABI):
    host image -> HBM upload -> HIP ORB extract (E0-E9) -> UndistortKeyPoints + AssignFeaturesToGrid (device)
    -> SearchByProjection(cur, last) incl. the projection (M2) -> PoseOptimization over ITS matches
    -> isInFrustum over the local map + SearchByProjection(F, local map) (M1) -> PoseOptimization over all matches
    -> a third PoseOptimization (TrackReferenceKeyFrame's fallback, SURVEY 8d counts three) -> keyframe decision /
    new map points, and, every 5th frame, one local bundle adjustment of an LBA-M window (B1) on a local-mapping
    thread of the same GPU.  The image starts in pinned host memory: the upload is inside the step.
Agents are independent: weak scaling, no per-frame collective.  Every `--exchange-every` frames
the ranks all-gather their newest keyframe's record over RCCL, append what they receive to a keyframe store in HBM and
look their own keyframe up in the whole store (the cross-agent loop/merge candidate search of
code/src/AgentMediator.cc:177-262); that exchange is inside the timed region.
Rank 0 prints ONE JSON line; with one GPU it also carries `configs`: the KITTI-sized stream, LBA-S/M/L windows and
global BA (BASELINE.json configs[3], configs[4]) measured once each after the headline region.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (pinned / device memory, barrier, RCCL plumbing only)

import swarmmap_amd  # noqa: E402
from swarmmap_amd import minitrack, synth  # noqa: E402
from swarmmap_amd.replay import Replay, make_vocabulary  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_PEAK_TF = 78.6    # MI355X FP64 vector/matrix peak (AMD datasheet; not tabulated in the guide)
INT_PEAK_TOPS = 78.6   # 32-bit integer VALU: 256 CUs x 4 SIMD-32 x 2.4 GHz lane-ops/s (the guide's execution model)
LIVE_STEPS = 41        # frames tracked the live way after the timed region (the first one is dropped: it was in flight)
STORE_KEYFRAMES = 4096  # keyframe store per rank: 8 agents x 512 keyframes (218 MB of records + 134 MB of search rows)
# one LBA-M window per ~5 frames (SURVEY.md 8d end-to-end replay); the override is a diagnostic (no local mapping)
LBA_EVERY = int(os.environ.get("SWARMORB_BENCH_LBA_EVERY", "5"))
# the local-mapping thread's matcher job per new keyframe (SearchForTriangulation against the last <= 20 keyframes, Fuse
# into them and back: code/src/LocalMapping.cc:197-246, 451-481) before its window; 0 switches it off
LM_MATCHER = int(os.environ.get("SWARMORB_BENCH_LM_MATCHER", "1"))
LM_NEIGHBOURS = 20  # nn = 20, LocalMapping.cc:207,455 (monocular)
# Untimed frames tracked before the driver's own warm-up so that the timed region - however short (the driver runs
# --steps 20 --warmup 5) - sees a local-mapping thread whose keyframe ring is full: 20 neighbours per new keyframe, not the
# two or three a cold start has.  Makes the timed steps heavier, never lighter.
LM_PREFILL_FRAMES = LM_NEIGHBOURS * LBA_EVERY if LM_MATCHER else 0
LOCAL_KEYFRAMES = 12   # local map = points created at (open loop) / seen by (closed loop) the last 12 keyframes (~3-5 k map points)
PLANE_Z = 2.0
# The measured chain is CLOSED (swarmmap_amd/host/closedloop.cc; swarmmap_amd/closedloop.py is the same loop in Python): what
# a keyframe's local-mapping job computes - triangulated points, fused duplicates, the poses and points local BA moved over
# the keyframe's OWN window - is in the tracked map five frames later.  "open": round 4's workload (a fixed synthetic LBA-M
# window per keyframe, nothing fed back), kept as configs.open_loop_synthetic_window.
LOOP = os.environ.get("SWARMORB_BENCH_LOOP", "closed")
SEED_BASE = int(os.environ.get("SWARMORB_BENCH_SEED", "20221001"))  # the synthetic stream of rank r is FrameStream(seed = SEED_BASE + r)
CL_N_FREE, CL_N_FIXED = 25, 40  # caps of a window's free / fixed keyframes (LBA-M's proportions)


def cgroup_cpu():
    """(usage_usec, nr_throttled, throttled_usec) of this container's CPU controller, or None: the GPU boxes cap a container at
    a CPU quota (cpu.max), and threads that spin while they wait count against it."""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d["usage_usec"]), int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except (OSError, ValueError, KeyError):
        return None


class GpuBusy:
    """Samples the driver's busy figure of the GPU (sysfs gpu_busy_percent: share of the last sampling period with anything running
    on the device) every few milliseconds over a timed region: mean / max / samples, or None where the file is not readable.  It is
    coarse (the SMU's own period) and says "something was running", not how many CUs - the per-kernel times are in `roofline`."""

    def __init__(self, device_index=0, period_s=0.004):
        self.path, self.period, self.vals, self.th, self.stop_flag = None, period_s, [], None, False
        self.device_index = device_index

    def locate(self):
        """The sysfs node of the HIP device: the card whose PCI address is the device's (a box shows every GPU of the node in sysfs,
        the container sees one of them)."""
        if self.path is not None:
            return
        self.path = ""
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(self.device_index)) != 0:
                return
            bus = buf.value.decode().lower()
            for c in os.listdir("/sys/class/drm"):
                if not (c.startswith("card") and c[4:].isdigit()):
                    continue
                dev = os.path.join("/sys/class/drm", c, "device")
                if os.path.basename(os.path.realpath(dev)).lower() == bus and os.path.exists(os.path.join(dev, "gpu_busy_percent")):
                    self.path = os.path.join(dev, "gpu_busy_percent")
                    return
        except (OSError, AttributeError):
            pass

    def _read(self):
        try:
            with open(self.path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def start(self):
        self.locate()
        if not self.path or self._read() is None:
            return
        self.vals, self.stop_flag = [], False

        def loop():
            while not self.stop_flag:
                v = self._read()
                if v is not None:
                    self.vals.append(v)
                time.sleep(self.period)
        self.th = threading.Thread(target=loop, daemon=True)
        self.th.start()

    def stop(self):
        if not self.th:
            return None
        self.stop_flag = True
        self.th.join()
        self.th = None
        if not self.vals:
            return None
        return {"mean_percent": round(float(np.mean(self.vals)), 1), "max_percent": float(np.max(self.vals)), "samples": len(self.vals),
                "source": "sysfs gpu_busy_percent, sampled every %d ms over the timed region" % int(self.period * 1e3)}


GPU_BUSY = GpuBusy()


def cgroup_delta(a, b, dt):
    if a is None or b is None:
        return None
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    return {"cpu_cores_used": (b[0] - a[0]) * 1e-6 / dt, "cpu_quota_cores": quota, "throttled_periods": b[1] - a[1],
            "throttled_thread_ms": (b[2] - a[2]) * 1e-3}


def level_pixels(inv_scale, w, h):
    """Sum of pyramid level pixels = bytes the FAST kernel must read at least once."""
    return int(sum(int(np.rint(np.float32(w) * s)) * int(np.rint(np.float32(h) * s)) for s in inv_scale))


def pinned_frames(stream, n):
    """n frames of the stream in ONE pinned host block (asynchronous H2D from the tracking thread)."""
    block = torch.empty((n, stream.h, stream.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(n):
        view[t] = stream.frame(t)
    return block, [view[t] for t in range(n)]


class LocalMapper:
    """CPU-baseline side of the reference's local-mapping thread (LocalMapping::Run, code/src/LocalMapping.cc:53-110):
    windows are queued by the tracking loop and optimised in order, two at most waiting."""

    def __init__(self, fn):
        import queue
        self.fn, self.q = fn, queue.Queue(maxsize=2)
        self.n = 0
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                self.q.task_done()
                return
            self.fn()
            self.n += 1
            self.q.task_done()

    def submit(self):
        self.q.put(True)

    def drain(self):
        self.q.join()

    def close(self):
        self.q.put(None)
        self.th.join()


def cpu_baseline(frames, K, dist, nfeatures, lba_window, size, budget_s=20.0, closed=None):
    """The same chained per-frame workload through the CPU oracle ("port": the reference has no CPU extractor and its
    g2o needs Eigen): tracking on one thread, local mapping on a second, like the reference; bounded sample.
    Closed loop: returns the chain's trajectory too (the ATE of the HIP chain against it is computed by the caller)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle_py
    from trajectory_common import OracleBackend
    closed = (LOOP == "closed") if closed is None else closed
    be = OracleBackend(K, nfeatures, dist if dist is not None else (0, 0, 0, 0, 0))
    state = {"n": 0, "t0": time.perf_counter()}
    if closed:
        from concurrent.futures import ThreadPoolExecutor
        from swarmmap_amd import closedloop
        pool = ThreadPoolExecutor(1)  # the local-mapping thread (the oracle's C operators release the interpreter lock)

        def on_frame(t):
            state["n"] = t + 1
            return time.perf_counter() - state["t0"] < budget_s

        traj = closedloop.track(be, None, len(frames), K, make_vocabulary(), plane_z=PLANE_Z, kf_every=LBA_EVERY, delay=LBA_EVERY,
                                local_keyframes=LOCAL_KEYFRAMES, neighbours=LM_NEIGHBOURS, n_free=CL_N_FREE, n_fixed=CL_N_FIXED,
                                third_pose=True, frames=frames, on_frame=on_frame, run_job=pool.submit)
        pool.shutdown(wait=True)
        dt = time.perf_counter() - state["t0"]
        n = state["n"]
        return {"value": n / dt, "unit": "frames/s", "cores": 2, "kind": "port",
                "sample_short": "%d frames %dx%d of the same stream in %.1f s: the same closed loop (tracking thread + local-mapping thread, "
                                "%d keyframes) through oracle/*.c (gcc -O3), python-driven; host has %d cores"
                                % (n, size[0], size[1], dt, len(traj["kf_t"]), os.cpu_count()),
                "sample": "%d frames %dx%d of the same stream in %.1f s: the CLOSED loop through the CPU oracle (extract nFeatures %d + "
                          "undistort + grid + M2 + isInFrustum + M1 + 3 PoseOptimization per frame on the tracking thread; %d "
                          "keyframes (1 per %d frames) on a local-mapping thread, each: SearchForTriangulation against the last <= "
                          "20 keyframes + triangulation + Fuse into them and back + LocalBundleAdjustment over its own window + "
                          "write-back, results in the tracked map %d frames later); operators in C (gcc -O3), the loop around them "
                          "python-driven (swarmmap_amd/closedloop.py) while the HIP arm runs the C++ loop; host has %d cores"
                          % (n, size[0], size[1], dt, nfeatures, len(traj["kf_t"]), LBA_EVERY, LBA_EVERY, os.cpu_count()),
                "_trajectory": traj}
    jobs = []  # matcher jobs handed over by the tracking loop, run by the local-mapping thread in front of its window

    def window():
        if jobs:
            jobs.pop(0)()
        oracle_py.bundle_adjust(lba_window)

    lm = LocalMapper(window)

    def on_frame(t):
        state["n"] = t + 1
        if t % LBA_EVERY == 0:
            lm.submit()
        return time.perf_counter() - state["t0"] < budget_s

    minitrack.track(be, None, len(frames), K, plane_z=PLANE_Z,
                    local_keyframes=LOCAL_KEYFRAMES, third_pose=True, frames=frames, on_frame=on_frame,
                    lm_every=LBA_EVERY if LM_MATCHER else 0, vocab=make_vocabulary() if LM_MATCHER else None,
                    lm_neighbours=LM_NEIGHBOURS, on_keyframe=jobs.append)
    lm.drain()
    dt = time.perf_counter() - state["t0"]
    lm.close()
    n = state["n"]
    return {"value": n / dt, "unit": "frames/s", "cores": 2, "kind": "port",
            "sample_short": "%d frames %dx%d of the same stream in %.1f s: the same open-loop chain (tracking thread + local-mapping "
                            "thread, %d windows) through oracle/*.c (gcc -O3), python-driven; host has %d cores"
                            % (n, size[0], size[1], dt, lm.n, os.cpu_count()),
            "sample": "%d frames %dx%d of the same stream in %.1f s: CPU oracle chain (extract nFeatures %d + undistort + grid + "
                      "M2 + isInFrustum + M1 + 3 PoseOptimization per frame on the tracking thread; %d keyframes (1 per "
                      "%d frames) on a local-mapping thread, each: SearchForTriangulation + Fuse against the last <= 20 "
                      "keyframes, then an LBA-M window); operators in C (gcc -O3), the loop around them python-driven "
                      "(swarmmap_amd/minitrack.py: ~1-3 %% of a ~30 ms frame) while the HIP arm runs the C++ loop; host has "
                      "%d cores" % (n, size[0], size[1], dt, nfeatures, lm.n, LBA_EVERY, os.cpu_count())}


def ate_records(log, cl, stream, K, oracle=None):
    """BASELINE.json's second half: ATE RMSE of the tracked stream against the renderer's ground truth and against the same
    loop run over the CPU oracle.  `online`: the poses as they were tracked; `final`: every frame's pose relative to its
    reference keyframe composed with that keyframe's FINAL pose - what System::SaveTrajectoryTUM writes at shutdown
    (code/src/System.cc:225-252), the trajectory `evo` is run on; `keyframes`: SaveKeyFrameTrajectoryTUM (:259-296).
    Monocular: Sim3-aligned (Umeyama with scale), the unaligned figure beside it (the synthetic map has metric scale)."""
    n = len(log["centres"])
    gt = minitrack.ground_truth(stream, n, K, PLANE_Z)

    def both(est, ref):
        return {"sim3_aligned_m": minitrack.ate_rmse(est, ref, with_scale=True), "unaligned_m": minitrack.ate_rmse(est, ref, align=False)}
    out = {"frames": n, "pixel_m": PLANE_Z / float(K[0]),
           "vs_ground_truth": {"online": both(log["centres"], gt), "final": both(cl["final_centres"][:n], gt),
                               "keyframes": both(cl["kf_centres"], gt[np.minimum(cl["kf_t"], n - 1)])}}
    if oracle is not None:
        m = min(n, len(oracle["centres"]))
        nk = int(min((cl["kf_t"] < m).sum(), (oracle["kf_t"] < m).sum()))
        differ = [t for t in range(m) if any(log[k][t] != oracle[k][t] for k in ("matches_last", "matches_map", "inliers"))]
        first = differ[0] if differ else m
        out["vs_oracle_chain"] = {
            "frames": m,
            "online_unaligned_m": minitrack.ate_rmse(log["centres"][:m], oracle["centres"][:m], align=False),
            "max_pose_entry_difference": float(np.abs(log["poses"][:m] - oracle["poses"][:m]).max()),
            "frames_with_different_match_or_inlier_counts": len(differ),
            "first_frame_with_different_counts": differ[0] if differ else None,
            "max_pose_entry_difference_before_it": float(np.abs(log["poses"][:first] - oracle["poses"][:first]).max()) if first else 0.0,
            "why_counts_can_differ": "the two chains' optimisers add their f64 sums in a different order, so float32 poses / points "
                                     "differ in the last bit from the first frames on; a Fuse / culling decision that sits on a "
                                     "threshold flips on that bit, one map point more or less shifts the slot numbering and the "
                                     "counts after it by a handful (tools/loop_diff.py finds the decision; tools/loop_sensitivity.py: "
                                     "the oracle chain against itself with one-ulp nudges parts at the same keyframe)",
            "oracle_vs_ground_truth_online": both(oracle["centres"][:m], gt[:m]),
            "note": "the oracle chain's final keyframe poses are final for ITS (shorter) run: only the online trajectories are compared"}
        del nk
    return out


def run_fleet(dev, size, K, dist, nfeatures, steps, warmup, seed, lba_window, barrier, agents, closed=None, fleet_threads=1):
    """A agents of one GPU in lockstep: ONE thread drives their tracking frame by frame (so_fleet_run) - the tracking stages of
    all agents are one chain of launches per stage (so_track_group), their frames one extraction chain (so_extractor_group) -
    and every agent keeps its own local-mapping thread.  Same return values as run_stream."""
    w, h = size
    closed = (LOOP == "closed") if closed is None else closed
    warmup = warmup + LM_PREFILL_FRAMES
    n_frames = warmup + steps + 2 + LBA_EVERY
    fleet, keep, streams = [], [], []
    for a in range(agents):
        stream = synth.FrameStream(seed=seed + 97 * a, size=size, K=K, dist=dist)
        block, frames = pinned_frames(stream, n_frames)
        keep.append((block, frames))
        streams.append(stream)
        rp = Replay(dev, w, h, nfeatures, LBA_EVERY, K, dist, plane_z=PLANE_Z, local_keyframes=LOCAL_KEYFRAMES, third_pose=True)
        rp.set_frames([block.data_ptr() + i * w * h for i in range(n_frames)], on_device=False)
        if closed:
            rp.set_vocabulary(make_vocabulary(), LM_NEIGHBOURS)
            rp.set_closed_loop(kf_every=LBA_EVERY, delay=LBA_EVERY, n_free=CL_N_FREE, n_fixed=CL_N_FIXED, policy=0)
        else:
            rp.set_window(lba_window)
            if LM_MATCHER:
                rp.set_vocabulary(make_vocabulary(), LM_NEIGHBOURS)
            rp.preallocate()
        rp.prime(0)
        fleet.append(rp)
    # Staggered keyframes: agent a is run alone for (a mod kf_every) frames first, so that inside the fleet it is that many frames
    # ahead of the clock and its keyframes - every kf_every-th frame of ITS stream, as in its solo run - fall on other ticks than
    # its neighbours': the local-mapping jobs of the agents do not all start (and queue their kernels) in the same instant, and no
    # tick waits for the slowest of eight jobs.  Real agents are not synchronised either.  Opt-in (SWARMORB_FLEET_STAGGER=1): measured
    # at 8 agents it removes the trackers' waits (141 -> 5-20 ms per agent) but a lockstep tick then stalls on whichever agent's job is
    # late, 5.6-5.9 k frames/s against 6.3 k in step (NOTES.md G.8).
    stagger = closed and bool(os.environ.get("SWARMORB_FLEET_STAGGER"))
    if stagger:
        for a, rp in enumerate(fleet):
            off = a % LBA_EVERY
            if off:
                rp.run(0, off, False)
                rp.set_fleet_offset(off)
    # `fleet_threads` driving threads, each with an equal share of the agents in lockstep (its own so_track_group and stream:
    # the handles a thread uses were created by it).  One thread: everything above ran on this one.
    T = max(1, min(fleet_threads, agents))
    if T == 1:
        Replay.fleet_run(fleet, 0, warmup, False)
        for rp in fleet:
            rp.drain()
            rp.set_profiling(False)
        barrier()
        cg0 = cgroup_cpu()
        GPU_BUSY.start()
        t0 = time.perf_counter()
        Replay.fleet_run(fleet, warmup, steps, True)
        for rp in fleet:
            rp.drain()  # every queued window is optimised inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        busy = GPU_BUSY.stop()
        cg = cgroup_delta(cg0, cgroup_cpu(), dt)
    else:
        raise RuntimeError("internal: fleets of several driving threads are built by run_fleet_threads")
    results = []
    fleet_lm = fleet[0].lm_stats()
    cl0 = fleet[0].closed_loop_log() if closed else None
    ticks = fleet[0].fleet_ticks()  # (elastic ticks: how many there were and how many agent places they filled - warm-up included)
    for rp in fleet:
        rp.finish()
        results.append((rp.stats(), rp.candidates_total(), rp.log()))
    if closed:
        cl0 = fleet[0].closed_loop_log()
    for rp in reversed(fleet):  # (the first agent owns what the fleet shares: the groups, the local-mapping stream)
        rp.close()
    stats = {k: sum(r[0][k] for r in results) / agents for k in results[0][0] if k not in ("stages", "frame_ms")}
    stats["frame_ms"] = np.concatenate([np.asarray(r[0]["frame_ms"])[-steps:] for r in results])
    stats.update({"n_xchg": 0, "xchg_ms": 0.0, "lm": fleet_lm, "closed": closed, "cgroup": cg, "gpu_busy": busy,
                  "fleet_ticks": {"ticks": ticks[0], "agents_per_tick": round(ticks[1] / max(ticks[0], 1), 3), "staggered": bool(stagger)}})
    if closed:
        stats.update({"cl": cl0, "stream": streams[0], "timed_from": warmup})
    return dt, stats, results[0][1], keep[0][1], results[0][2]


def run_fleet_threads(dev, size, K, dist, nfeatures, steps, warmup, seed, lba_window, barrier, agents, fleet_threads, closed=None):
    """`fleet_threads` lockstep fleets side by side (agents split evenly), one driving thread each; the clock runs from the
    moment all fleets are warm until the last one has drained.  Returns what run_fleet returns (statistics averaged over fleets)."""
    T = max(1, min(fleet_threads, agents))
    share = [agents // T + (1 if i < agents % T else 0) for i in range(T)]
    first = [sum(share[:i]) for i in range(T)]
    gate = threading.Barrier(T + 1)
    out, errors = [None] * T, []

    def fleet_barrier():
        gate.wait()   # all fleets warm / all drained
        gate.wait()   # the main thread has read the clock

    def run(i):
        try:
            out[i] = run_fleet(dev, size, K, dist, nfeatures, steps, warmup, seed + 97 * first[i], lba_window, fleet_barrier, share[i], closed)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            gate.abort()

    ths = [threading.Thread(target=run, args=(i,), daemon=True) for i in range(T)]
    for th in ths:
        th.start()
    try:
        gate.wait(); barrier(); t0 = time.perf_counter(); gate.wait()
        gate.wait(); barrier(); dt = time.perf_counter() - t0; gate.wait()
    except threading.BrokenBarrierError:
        raise errors[0] if errors else RuntimeError("a fleet thread failed")
    for th in ths:
        th.join()
    if errors:
        raise errors[0]
    st = dict(out[0][1])
    for k, v in st.items():
        if isinstance(v, float):
            st[k] = sum(o[1][k] * share[i] for i, o in enumerate(out)) / agents
    st["frame_ms"] = np.concatenate([o[1]["frame_ms"] for o in out])
    return dt, st, out[0][2], out[0][3], out[0][4]


def run_stream(dev, size, K, dist, nfeatures, steps, warmup, seed, lba_window, barrier, agents=1, xchg=None,
               exchange_every=20, m1=None, live_steps=0, closed=None, policy=0):
    """Timed region of the per-frame path on one GPU.  Returns (dt seconds, per-agent stats, candidate count, frames).
    live_steps > 0: after the timed region that many more frames are tracked the way a live camera delivers them
    (so_replay_run_live: nothing extracted ahead) and their image-in -> pose-out latencies land in stats["live_*"]."""
    w, h = size
    closed = (LOOP == "closed") if closed is None else closed
    warmup = warmup + LM_PREFILL_FRAMES
    n_frames = warmup + steps + 2 + live_steps
    A = max(1, agents)
    gate = threading.Barrier(A + 1)
    results, errors, frame_sets = [None] * A, [], [None] * A
    clock = {}
    acc_x = {"n_xchg": 0, "xchg_ms": 0.0}

    def sync(tag):
        """Agent side of the rendezvous points.  With one agent the loop runs on this (the main) thread, so that no
        stream beyond the agent's own exists, and the clock is handled right here."""
        if A > 1:
            gate.wait()
        elif tag == "warm":
            barrier()
            clock["cg0"] = cgroup_cpu()
            GPU_BUSY.start()
            clock["t0"] = time.perf_counter()
        elif tag == "done":
            barrier()  # (the frame submitted ahead by the last step finishes inside the timed region too)
            clock["dt"] = time.perf_counter() - clock["t0"]
            clock["busy"] = GPU_BUSY.stop()
            clock["cg"] = cgroup_delta(clock["cg0"], cgroup_cpu(), clock["dt"])

    def agent(a):
        # created in the thread that runs it: the library gives every thread its own tracking streams
        try:
            stream = synth.FrameStream(seed=seed + 97 * a, size=size, K=K, dist=dist)
            block, frames = pinned_frames(stream, n_frames)
            frame_sets[a] = (block, frames)
            rp = Replay(dev, w, h, nfeatures, LBA_EVERY, K, dist, plane_z=PLANE_Z, local_keyframes=LOCAL_KEYFRAMES,
                        third_pose=True)
            rp.set_frames([block.data_ptr() + i * w * h for i in range(n_frames)], on_device=False)
            if closed:  # the closed loop: every keyframe's local-mapping job over its own neighbours and its own window
                rp.set_vocabulary(make_vocabulary(), LM_NEIGHBOURS)
                rp.set_closed_loop(kf_every=LBA_EVERY, delay=LBA_EVERY, n_free=CL_N_FREE, n_fixed=CL_N_FIXED, policy=policy)
            else:
                rp.set_window(lba_window)
                if LM_MATCHER:  # the local-mapping thread's matcher job: SearchForTriangulation + Fuse per new keyframe
                    rp.set_vocabulary(make_vocabulary(), LM_NEIGHBOURS)
                rp.preallocate()  # device buffers of the local-mapping solver sized once, before any step is counted

            def run_span(first, n, timed):
                t_ = first
                while t_ < first + n:
                    if xchg is None or a > 0:
                        m_ = first + n - t_
                    else:  # agent 0 of the rank stops after the next exchange tick
                        nxt = (t_ // exchange_every + 1) * exchange_every
                        m_ = min(first + n, nxt + 1) - t_
                    rp.run(t_, m_, timed)
                    t_ += m_
                    if xchg is not None and a == 0 and (t_ - 1) % exchange_every == 0:
                        tx = time.perf_counter()
                        # the frame tracked last goes out as a keyframe record assembled on the device (descriptors and
                        # undistorted keypoints from HBM, map-point bindings from the tracker), every rank appends what
                        # it receives to its keyframe store and looks its own keyframe up in the WHOLE store
                        mp, Tcw = rp.last_bindings()
                        cands = xchg.tick_keyframe(rp.last_dframe(), len(mp), agent_id=xchg.rank, keyframe_id=t_,
                                                   map_point_id=mp, timestamp=t_ / 20.0, Tcw=Tcw, K=K)
                        if timed:
                            acc_x["n_xchg"] += 1
                            acc_x["xchg_ms"] += (time.perf_counter() - tx) * 1e3
                            acc_x["xchg_candidates"] = acc_x.get("xchg_candidates", 0) + len(cands)
                            st_x = xchg.store.last_stats()
                            acc_x["xchg_scan_ms"] = acc_x.get("xchg_scan_ms", 0.0) + st_x["scan_ms"]
                            acc_x["xchg_pairs"] = acc_x.get("xchg_pairs", 0.0) + st_x["pairs"]
                            acc_x["xchg_store_keyframes"] = xchg.store.size()[0]

            if xchg is not None and a == 0:
                # ranks finish rendering their streams at different times: meet before the first (collective) exchange tick,
                # and give the ticks a budget that start-up skew cannot exhaust (a dead peer is still noticed: SO_ERR_TIMEOUT)
                if A == 1:  # (with several agents per GPU this is not the main thread: no process-group call from here)
                    barrier()
                xchg.set_timeout(int(os.environ.get("SWARMORB_COLLECTIVE_TIMEOUT_MS", "30000")))
            rp.prime(0)
            run_span(0, warmup, False)
            rp.drain()
            rp.set_profiling(False)  # stage events off: the frame goes out as one hipGraph launch
            sync("warm")  # all agents warmed up
            sync("go")    # the main thread has passed the barrier and started the clock
            run_span(warmup, steps, True)
            rp.drain()    # every queued window is optimised inside the timed region
            sync("done")
            live = None
            if live_steps > 0 and a == 0:  # (outside the timed region; the first frame of the span is already in flight)
                live = rp.run_live(warmup + steps, live_steps)
            rp.finish()
            results[a] = (rp.stats(), rp.candidates_total(), rp.log())
            if a == 0:
                acc_x["lm"] = rp.lm_stats()
                if closed:
                    acc_x["cl"] = rp.closed_loop_log()
                    acc_x["stream"] = stream
                    acc_x["timed_from"] = warmup
            if live is not None:
                acc_x["live_pose_ms"], acc_x["live_step_ms"] = live[0][1:], live[1][1:]
            rp.close()
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(e)
            if A > 1:
                gate.abort()

    if A == 1:
        agent(0)
        if errors:
            raise errors[0]
        dt = clock["dt"]
    else:
        threads = [threading.Thread(target=agent, args=(a,), daemon=True) for a in range(A)]
        for th in threads:
            th.start()
        try:
            gate.wait()
            barrier()
            clock["cg0"] = cgroup_cpu()
            GPU_BUSY.start()
            t0 = time.perf_counter()
            gate.wait()
            gate.wait()
            barrier()  # (the frames submitted ahead by the last steps finish inside the timed region too)
            dt = time.perf_counter() - t0
            clock["busy"] = GPU_BUSY.stop()
            clock["cg"] = cgroup_delta(clock["cg0"], cgroup_cpu(), dt)
        except threading.BrokenBarrierError:
            raise errors[0] if errors else RuntimeError("an agent thread failed")
        for th in threads:
            th.join()
        if errors:
            raise errors[0]
    stats = {k: sum(r[0][k] for r in results) / A for k in results[0][0] if k not in ("stages", "frame_ms")}  # per-agent averages
    stats["frame_ms"] = np.concatenate([np.asarray(r[0]["frame_ms"])[-steps:] for r in results])  # timed frames of every agent
    stats.update(acc_x)
    stats["closed"] = closed
    stats["cgroup"] = clock.get("cg")
    stats["gpu_busy"] = clock.get("busy")
    return dt, stats, results[0][1], frame_sets[0][1], results[0][2]


def closed_loop_record(st, steps):
    """config.closed_loop: what the local-mapping thread did with the keyframes of the TIMED region (rows of its log whose
    keyframe lies in it) and how its time splits; counters of the whole run."""
    from swarmmap_amd.closedloop import LM_LOG_COLUMNS
    cl, L = st["cl"], st.get("lm", {})
    lm = cl["lm_log"]
    rows = lm[lm[:, 0] >= st["timed_from"]]
    col = {k: rows[:, i].astype(np.float64) for i, k in enumerate(LM_LOG_COLUMNS)}
    mean = lambda k: float(col[k].mean()) if len(rows) else 0.0  # noqa: E731
    jobs = max(L.get("jobs", 0.0), 1.0)
    return {
        "schedule": "deterministic: a keyframe every %d frames, its results in the tracked map %d frames later (the tracking "
                    "thread waits if the job is not done)" % (LBA_EVERY, LBA_EVERY),
        "keyframes_in_timed_region": int(len(rows)), "neighbours": mean("neighbours"),
        "per_keyframe": {"triangulation_matches": mean("tri_matches"), "new_map_points": mean("new_points"),
                         "fused_into_neighbours": mean("fused"), "fused_back": mean("fused_back"), "points_gone_bad": mean("bad_points"),
                         "lba_edges": mean("lba_edges"), "lba_outlier_edges": mean("lba_outliers"), "lba_free_keyframes": mean("lba_free"),
                         "lba_fixed_keyframes": mean("lba_fixed"), "lba_points": mean("lba_points")},
        "whole_run": dict(cl["counts"], tracking_thread_waited_ms=cl["wait_ms"]),
        "local_mapping_ms_per_keyframe": {
            "whole_job": L.get("cl_job_ms", 0.0) / jobs, "process_new_keyframe_and_culling": L.get("cl_process_ms", 0.0) / jobs,
            "feature_vector_and_upload": L.get("node_ms", 0.0) / jobs,
            "search_for_triangulation_batch": L.get("stage_tri_ms", 0.0) / jobs,
            "search_for_triangulation_batch_launch_wait_resolve": L.get("cl_tri_batch_end_ms", 0.0) / jobs, "triangulation_and_new_points": L.get("triangulate_ms", 0.0) / jobs,
            "fuse_batch_stage": L.get("stage_fuse_ms", 0.0) / jobs, "fuse_batch_launch_wait_resolve": L.get("batch_end_ms", 0.0) / jobs,
            "fuse_apply": L.get("cl_apply_ms", 0.0) / jobs, "window_gather": L.get("cl_gather_ms", 0.0) / jobs,
            "so_bundle_adjust": L.get("cl_solver_ms", 0.0) / jobs, "write_back_and_update_normal_depth": L.get("cl_writeback_ms", 0.0) / jobs,
            "write_back_split": {"set_pose_set_world_pos_erase": L.get("cl_wb_apply_ms", 0.0) / jobs, "observers_per_point": L.get("cl_und_build_ms", 0.0) / jobs,
                                 "so_update_normal_and_depth": L.get("cl_und_call_ms", 0.0) / jobs},
            "packet": L.get("cl_packet_ms", 0.0) / jobs,
            "kernels": {"search_for_triangulation_batch": L.get("batch_kernel_ms", 0.0) / jobs, "triangulation": L.get("triangulate_kernel_ms", 0.0) / jobs,
                        "fuse_batch": L.get("cl_fuse_kernel_ms", 0.0) / jobs},
            "batches_enqueue": L.get("batch_enqueue_ms", 0.0) / jobs, "batches_wait": L.get("batch_wait_ms", 0.0) / jobs},
    }


def extractor_stage_profile(dev, frames, size, nfeatures, n_prof=64):
    """Per-stage HIP-event times of the extractor: an untimed profiled pass (stage events split the frame's graph back
    into single launches, so they stay out of the timed loop).  Returns (ms per frame per stage, inverse scale factors)."""
    w, h = size
    ex = swarmmap_amd.ORBextractor(nfeatures, 1.2, 8, 20, 7, device=dev)
    dev_frames = [torch.from_numpy(f).cuda(dev) for f in frames[:n_prof]]
    ex.run_device(dev_frames[0].data_ptr(), w, h, w)
    ex.set_profiling(True)
    stage = {}
    for f in dev_frames:
        ex.run_device(f.data_ptr(), w, h, w)
        for k, v in ex.profile().items():
            stage[k] = stage.get(k, 0.0) + v / len(dev_frames)
    inv = ex.GetInverseScaleFactors()
    ex.close()
    return stage, inv


def stream_record(size, nfeatures, steps, dt, st, n_cand, stage, inv_scale, pmc, agents=1):
    """Common per-stream fields: throughput, per-frame host times, roofline objects of the three candidate kernels."""
    w, h = size
    fast_ms = stage["fast_score"]
    alg_bytes = level_pixels(inv_scale, w, h) + 8 * n_cand  # every level pixel read once + 8 B per candidate written
    achieved = alg_bytes / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else 0.0
    roof_fast = {"bound": "hbm", "kernel": "fast_score_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                 "traffic": pmc.get("fast_score_kernel", {}).get("hbm_bytes_per_launch"),
                 "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes of this command)",
                 "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": fast_ms,
                 "measured": "untimed profiled pass of 64 frames right after the timed region (the timed loop launches "
                             "the frame as one hipGraph, without stage events)",
                 "total_ms_in_timed_region": fast_ms * steps}
    # PoseOptimization kernel: per LM trial every matched point costs ~250 flop (projection, 2x6 Jacobian, the 27
    # accumulations of J^T w J | J^T w e, chi2 and Huber weight), all FP64, plus a 6x6 solve
    # HIP events bracket the kernel on every 4th frame of the timed region (swarmmap_amd/host/replay.cc kEventEvery): the
    # averages below are over those calls, the total is scaled to all calls of the region
    calls = max(st["pose_timed_calls"], 1)
    n_pose = st["pose_points"] / calls
    pose_flop = 250.0 * n_pose * (st["pose_trials"] / calls + 4)  # +4: one pass per round
    # (lockstep with grouped stages: one launch optimises the poses of all agents - every agent is dealt 1/A of the launch's
    #  time per call, so flop / ms below is the launch's flop over all its workgroups divided by the launch's duration)
    pose_ms = st["pose_kernel_ms"] / calls
    pose_total_ms = pose_ms * st["pose_calls"]
    pose_tf = pose_flop / (pose_ms * 1e-3) / 1e12 if pose_ms > 0 else 0.0
    chain = os.environ.get("SWARMORB_TRACK_CHAIN", "1") != "0"
    grouped = chain and agents > 1 and "--lockstep" in sys.argv and not os.environ.get("SWARMORB_FLEET_NO_TRACK_GROUP")
    pose_kernel = "pose_opt_chain_group_kernel" if grouped else ("pose_opt_chain_kernel" if chain else "pose_opt_reg_kernel")
    roof_pose = {"bound": "fp64-valu (latency: one workgroup, no MFMA)", "kernel": pose_kernel, "achieved": pose_tf,
                 "peak": FP64_PEAK_TF, "unit": "TFLOP/s", "frac": pose_tf / FP64_PEAK_TF,
                 "traffic": pmc.get(pose_kernel, pmc.get("pose_opt_reg_kernel", pmc.get("pose_opt_lds_kernel", {}))).get("hbm_bytes_per_launch"),
                 "launched_as": ("the last launch of a tracking stage's chain (search -> resolve on the device -> this kernel, edges read in "
                                 "place from the device-resident frame and map table; so_track_stage_*)" if chain else
                                 "so_pose_optimization after a host-side resolve and gather"),
                 "algorithmic_flop_per_launch": pose_flop, "avg_launch_ms": pose_ms, "avg_points": n_pose,
                 "avg_lm_trials": st["pose_trials"] / calls, "event_timed_launches": st["pose_timed_calls"],
                 "launches_in_timed_region": st["pose_calls"], "total_ms_in_timed_region": pose_total_ms,
                 "note": "one workgroup runs g2o's 4 x optimize(10) on one 6-dof vertex: serial LM trials, latency-bound "
                         "by construction (DESIGN.md 5b); peak is the FP64 vector rate"}
    rec = {
        "frames_per_s": steps * agents / dt, "ms_per_frame": dt / steps * 1e3, "frame": [w, h], "nfeatures": nfeatures,
        "keypoints_per_frame": st["n_kp"] / steps, "m2_matches_per_frame": st["n_m2"] / steps,
        "m1_matches_per_frame": st["n_m1"] / steps, "inliers_per_frame": st["n_inliers"] / steps,
        "local_map_points_per_frame": st["n_local"] / steps, "in_view_per_frame": st["n_in_view"] / steps,
        "frame_ms_percentiles": (lambda f: {k: float(np.percentile(f, q)) for k, q in (("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))}
                                 if len(f) else {})(np.asarray(st.get("frame_ms", []), np.float64)),
        # a LIVE frame (System::TrackMonocular is synchronous, code/src/System.cc:128-165): the image is handed over when
        # its step begins, nothing is extracted ahead: upload + extraction + frame post-processing + both searches + the
        # two PoseOptimization calls that produce the frame's pose, back to back (untimed frames right after the region)
        "latency_ms_image_to_pose": (lambda f: {k: float(np.percentile(f, q)) for k, q in (("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))}
                                     if len(f) else None)(np.asarray(st.get("live_pose_ms", []), np.float64)),
        "latency_ms_live_step": (lambda f: {k: float(np.percentile(f, q)) for k, q in (("p50", 50), ("p99", 99))}
                                 if len(f) else None)(np.asarray(st.get("live_step_ms", []), np.float64)),
        "search_launches_per_frame": st.get("n_reruns", 0.0) / steps, "wide_window_m2_per_frame": st.get("n_wide_m2", 0.0) / steps,
        "keyframes": st["n_keyframes"], "map_points_at_end": st["n_map_points"],
        "lba_windows": st["n_lba"],
        "host_ms_per_frame": {"collect_and_submit": st["extract_ms"] / steps, "match": (st["m2_ms"] + st["m1_ms"]) / steps,
                              "match_m2": st["m2_ms"] / steps, "match_m1": st["m1_ms"] / steps,
                              "match_m2_enqueue": st["m2_enqueue_ms"] / steps, "match_m2_sync_wait": st["m2_wait_ms"] / steps,
                              "match_m1_enqueue": st["m1_enqueue_ms"] / steps, "match_m1_sync_wait": st["m1_wait_ms"] / steps,
                              "pose_optimization_x3": (st["pose1_ms"] + st["pose2_ms"] + st["pose3_ms"]) / steps,
                              "keyframe_map_insert": st["map_ms"] / steps, "lba_submit_wait": st["lba_ms"] / steps,
                              "lba_thread_busy": st["lba_busy_ms"] / steps,
                              "exchange_amortised": st.get("xchg_ms", 0.0) / steps},
        "lba_ms_per_window": ({"wall": st.get("lm", {}).get("cl_solver_ms", 0.0) / max(st["n_lba"], 1),
                               "gpu": st["lba_gpu_ms"] / max(st["n_lba"], 1)} if st.get("closed") else
                              {"wall": (st["lba_busy_ms"] - st.get("lm", {}).get("wall_ms", 0.0)) / max(st["n_lba"], 1),
                               "gpu": st["lba_gpu_ms"] / max(st["n_lba"], 1)}),
        # the local-mapping thread's matcher job in front of every window (inside lba_thread_busy): per new keyframe
        # SearchForTriangulation against each of the last <= 20 keyframes (M5), Fuse into each and back (M6)
        "local_mapping_matcher": (lambda L: None if not L or not L.get("jobs") or st.get("closed") else {
            "keyframes": L["jobs"], "neighbours": LM_NEIGHBOURS, "wall_ms_per_keyframe": L["wall_ms"] / L["jobs"],
            "untimed_prefill_frames": LM_PREFILL_FRAMES,  # tracked before the warm-up: the keyframe ring is full when the clock starts
            # all searches of a keyframe go out as one so_matcher batch (one staging copy, one projection launch, one
            # search launch, one wait); SWARMORB_LM_BATCH=0 issues them one by one (then the per-call kernel times below)
            "batched": bool(L.get("batch_ms")),
            "batch": None if not L.get("batch_ms") else {"stage_and_launch_and_resolve_ms_per_keyframe": L["batch_ms"] / L["jobs"],
                                                          "launch_wait_resolve_ms_per_keyframe": L["batch_end_ms"] / L["jobs"],
                                                          "kernels_ms_per_keyframe": L["batch_kernel_ms"] / L["jobs"],
                                                          # where the host time of the batch goes: staging the searches'
                                                          # inputs (per group of calls), then enqueue / wait / resolve
                                                          "stage_triangulation_searches_ms": L.get("stage_tri_ms", 0.0) / L["jobs"],
                                                          "stage_fuse_into_neighbours_ms": L.get("stage_fuse_ms", 0.0) / L["jobs"],
                                                          "stage_fuse_back_ms": L.get("stage_back_ms", 0.0) / L["jobs"],
                                                          "enqueue_ms": L.get("batch_enqueue_ms", 0.0) / L["jobs"],
                                                          "wait_ms": L.get("batch_wait_ms", 0.0) / L["jobs"],
                                                          "resolve_ms": (L["batch_end_ms"] - L.get("batch_enqueue_ms", 0.0)
                                                                         - L.get("batch_wait_ms", 0.0)) / L["jobs"]},
            "feature_vector_ms_per_keyframe": L["node_ms"] / L["jobs"],
            "search_for_triangulation": {"calls_per_keyframe": L["tri_calls"] / L["jobs"], "host_ms_per_call": L["tri_ms"] / max(L["tri_calls"], 1),
                                         "match_kernel_ms_per_call": L["tri_kernel_ms"] / max(L["tri_calls"], 1),
                                         "matches_per_keyframe": L["tri_matches"] / L["jobs"]},
            # CreateNewMapPoints' per-match body for the matches of all neighbours (one launch) + UpdateNormalAndDepth of the
            # new points
            "triangulation": {"host_ms_per_keyframe": L.get("triangulate_ms", 0.0) / L["jobs"],
                              "kernel_ms_per_keyframe": L.get("triangulate_kernel_ms", 0.0) / L["jobs"],
                              "new_map_points_per_keyframe": L.get("new_points", 0.0) / L["jobs"]},
            "fuse": {"calls_per_keyframe": L["fuse_calls"] / L["jobs"], "host_ms_per_call": L["fuse_ms"] / max(L["fuse_calls"], 1),
                     "match_kernel_ms_per_call": L["fuse_kernel_ms"] / max(L["fuse_calls"], 1),
                     "map_points_per_call": L["fuse_points"] / max(L["fuse_calls"], 1), "fused_per_keyframe": L["fused"] / L["jobs"]}})(st.get("lm")),
        "match_kernel_ms_per_frame": st["match_kernel_ms"] / max(st["timed_frames"], 1),
        "pose_kernel_ms_per_call": pose_ms,
        "extract_stage_ms_per_frame": stage,
    }
    return rec, roof_fast, roof_pose


def random_keyframe_records(rng, agents, per_agent, n_kp, bound_frac, first_agent=1):
    """Version-2 keyframe records of random descriptors (no two keyframes look alike: the search finds no candidate and
    the scan does all of its work), bound_frac of the keypoints carrying a map point."""
    from swarmmap_amd.kfstore import pack_keyframe_record2
    recs = []
    for a in range(agents):
        for k in range(per_agent):
            mp = np.where(rng.random(n_kp) < bound_frac, rng.integers(0, 1 << 30, n_kp), -1).astype(np.int32)
            recs.append(pack_keyframe_record2(first_agent + a, k, 0.0, np.zeros(12, np.float32), synth.EUROC_K,
                                              rng.uniform(0, 752, (n_kp, 2)).astype(np.float32),
                                              rng.uniform(0, 360, n_kp).astype(np.float32), rng.integers(0, 8, n_kp).astype(np.int32),
                                              rng.integers(0, 256, (n_kp, 32), dtype=np.uint8), mp))
    return recs


def prefill_store(store, agents, per_agent, n_kp, bound_frac=0.4, seed=5):
    rng = np.random.default_rng(seed)
    for a in range(agents):
        store.append(random_keyframe_records(rng, 1, per_agent, n_kp, bound_frac, first_agent=1 + a))


def candidate_search_records(dev):
    """SURVEY 8e / BASELINE configs[2]-[4]: the cross-agent candidate search on one GPU - a keyframe store as eight agents
    fill it (512 keyframes each, 1000 keypoints, 40 % of them bound to map points; `dense`: every keypoint bound), one new
    keyframe looked up in all of it.  The detection scan is the one throughput kernel of the path: its roofline is the
    integer VALU rate (16 instructions per descriptor pair at least: 8 v_xor + 8 accumulating v_bcnt)."""
    from swarmmap_amd.kfstore import KeyframeStore, search_params
    out = {}
    for name, per_agent, frac in (("store_8x512_kf_40pct_bound", 512, 0.4), ("store_8x128_kf_all_bound", 128, 1.0)):
        rng = np.random.default_rng(11)
        store = KeyframeStore(8 * per_agent, 1024, device=dev)
        t0 = time.perf_counter()
        for a in range(8):
            store.append(random_keyframe_records(rng, 1, per_agent, 1000, frac, first_agent=1 + a))
        fill_s = time.perf_counter() - t0
        q = random_keyframe_records(rng, 1, 1, 1000, frac, first_agent=0)[0]
        p = search_params()
        for _ in range(3):
            store.search(q, p, want_pairs=False)
        scan, wall = [], []
        for _ in range(20):
            t0 = time.perf_counter()
            store.search(q, p, want_pairs=False)
            wall.append(time.perf_counter() - t0)
            scan.append(store.last_stats()["scan_ms"])
        st = store.last_stats()
        ms = float(np.median(scan))
        ops = 16.0 * st["pairs"]
        n_kf, n_desc = store.size()
        out[name] = {"keyframes": n_kf, "store_descriptors": int(n_desc), "query_descriptors": int(st["pairs"] / max(n_desc, 1)),
                     "descriptor_pairs": st["pairs"], "scan_kernel_ms": ms, "search_wall_ms": float(np.median(wall)) * 1e3,
                     "keyframes_per_s": n_kf / (ms * 1e-3) if ms > 0 else 0.0, "store_fill_s": fill_s,
                     "roofline": {"bound": "int-valu", "kernel": "kf_scan_kernel", "achieved": ops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                                  "peak": INT_PEAK_TOPS, "unit": "Tlane-op/s", "frac": ops / (ms * 1e-3) / 1e12 / INT_PEAK_TOPS if ms > 0 else 0.0,
                                  "algorithmic_ops_per_pair": 16, "issued_ops_per_pair": 18,
                                  "hbm_bytes_algorithmic": 32.0 * n_desc, "hbm_gbs": 32.0 * n_desc / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                                  "note": "256-bit Hamming distance = 8 v_xor + 8 v_bcnt (the popcount accumulates); best / "
                                          "second add v_med3 + v_min; peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz"}}
        store.close()
    return out


def batched_front_end_records(dev, members=(1, 8, 32), reps=40):
    """Verdict item 9 (round 2): A agents' frames through ONE extraction chain (so_extractor_group + so_dframe_group_submit,
    every kernel once with the agent as a grid dimension) - wall time from the group submit to the last frame complete on
    the device, images in pinned host memory (361 KB per frame cross PCIe inside the timed region).  Not the headline's
    deployment (one agent per GPU); what the front end does when agents share a GPU."""
    import ctypes as C
    st = synth.FrameStream(seed=20221001, size=synth.EUROC, K=synth.EUROC_K, dist=synth.EUROC_DIST)
    nimg = 8
    block = torch.empty((nimg, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(nimg):
        view[t] = st.frame(t)
    out = {}
    for A in members:
        exs = [swarmmap_amd.ORBextractor(1000, 1.2, 8, 20, 7, device=dev) for _ in range(A)]
        frs = [swarmmap_amd.DeviceFrame(ex, synth.EUROC_K, synth.EUROC_DIST) for ex in exs]
        grp = swarmmap_amd.ExtractorGroup(exs)
        lib = frs[0]._lib
        lib.so_dframe_wait.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        ts = []
        for r in range(reps + 8):
            imgs = [view[(r + a) % nimg] for a in range(A)]
            t0 = time.perf_counter()
            grp.submit(imgs, frames=frs)
            nk = C.c_int(0)
            for f in frs:
                lib.so_dframe_wait(f._h, C.byref(nk), None)
            t1 = time.perf_counter()
            for f in frs:
                f.collect()
            if r >= 8:
                ts.append(t1 - t0)
        grp.close()
        for f in frs:
            f.close()
        for ex in exs:
            ex.close()
        ms = float(np.median(ts)) * 1e3
        out["agents_%d" % A] = {"ms_per_chain": ms, "frames_per_s": A / (ms * 1e-3),
                                "algorithmic_gbs": A * 7.96e6 / (ms * 1e-3) / 1e9, "hbm_frac": A * 7.96e6 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    out["note"] = ("752x480, nFeatures 1000, EuRoC lens model; 7.96 MB algorithmic bytes per frame (SURVEY 8d); a lone frame is "
                   "latency-bound, 32 per chain are bound by PCIe ingest + the FAST and descriptor kernels' ALU work")
    return out


def lba_records(dev):
    """BASELINE configs[2]/[3] local-BA leg: LBA-S / LBA-M / LBA-L windows, wall time of one
    Optimizer::LocalBundleAdjustment through the C ABI (median of 5 after 2 warm-up calls)."""
    o = swarmmap_amd.Optimizer(device=dev)
    out = {}
    for name in ("LBA-S", "LBA-M", "LBA-L", "LBA-64"):
        # LBA-64: a 64-keyframe window with LBA-L's proportions (150 points and 1.5 fixed keyframes per free one): beyond the
        # single-workgroup solvers, the size the single-launch tile-dataflow solve was built for
        wnd = synth.make_ba_problem(0, 64, 96, 9600, max_obs="auto") if name == "LBA-64" else synth.make_ba_case(name)
        for _ in range(2):
            o.LocalBundleAdjustment(wnd)
        ts, infos = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            r = o.LocalBundleAdjustment(wnd)
            ts.append(time.perf_counter() - t0)
            infos.append(r["info"])
        inf = dict(infos[int(np.argsort(ts)[len(ts) // 2])])
        o.set_solve_timing(True)  # one more call with HIP events around the solves (they idle the stream: not in wall_ms)
        it = o.LocalBundleAdjustment(wnd)["info"]
        o.set_solve_timing(False)
        inf["solve_ms"], inf["n_solves"] = it["solve_ms"], it["n_solves"]
        n = 6 * int(inf["n_free_keyframes"])
        flop = n ** 3 / 3.0 + 2.0 * n ** 2
        ms = inf["solve_ms"] / max(inf["n_solves"], 1)
        tf = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        out[name] = {"free_keyframes": n // 6, "solver_path": int(inf["solver_path"]), "fixed_keyframes": int((wnd["fixed"] != 0).sum()), "points": int(len(wnd["Xw"])),
                     "edges": int(len(wnd["edge_pose"])), "wall_ms": float(np.median(ts)) * 1e3, "gpu_ms": inf["gpu_ms"],
                     "lm_trials": inf["lm_trials"], "chi2_final": inf["chi2_final"],
                     "solve": {"n": n, "ms_per_solve": ms, "algorithmic_flop": flop, "achieved_tflops": tf,
                               "peak_tflops": FP64_PEAK_TF, "frac": tf / FP64_PEAK_TF,
                               "bound": "mfma (latency-bound: one workgroup)" if inf["solver_path"] == 0 else "mfma (latency-bound: a chain of 96-column panels)"}}
    o.close()
    return out


def gba_records(dev, cases):
    """BASELINE configs[4]: whole-map BundleAdjustment(10 iterations) on one GPU; FP64 rate of the reduced-camera solve."""
    o = swarmmap_amd.Optimizer(device=dev)
    o.set_solve_timing(True)  # (two event records per solve: nothing next to a multi-millisecond solve)
    out = {}
    for name in cases:
        t0 = time.perf_counter()
        p = synth.make_ba_case(name, 1)
        gen_s = time.perf_counter() - t0
        o.BundleAdjustment(p, nIterations=2, bRobust=False)  # warm-up: buffers
        # the function's default (Huber on) as a second key; the headline is the reference's own server-side call,
        # GlobalBundleAdjustemnt(map, 10, &stop, kf, false): code/src/MediatorScheduler.cc:122, LoopClosing.cc:606
        t0 = time.perf_counter()
        rh = o.BundleAdjustment(p, nIterations=10, bRobust=True)
        wall_huber = time.perf_counter() - t0
        t0 = time.perf_counter()
        r = o.BundleAdjustment(p, nIterations=10, bRobust=False)
        wall = time.perf_counter() - t0
        inf = r["info"]
        n = 6 * int(inf["n_free_keyframes"])  # keyframes the solver gave a hessian index (not fixed AND observed)
        flop = n ** 3 / 3.0 + 2.0 * n ** 2
        ms = inf["solve_ms"] / max(inf["n_solves"], 1)
        tf = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        sflop = inf["solve_gflop_structural"] * 1e9  # FP64 work over the nonzero tiles of the block skyline only
        stf = sflop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        T = (n + 95) // 96
        # the headline is the SMALLER of the two counts: the flop over the skyline's 96x96 tiles include the zero padding
        # of the last tile row (GBA-1: 1308 unknowns in 14 x 96 = 1344 rows), the dense n^3/3 + 2 n^2 of the un-padded
        # system exceed what a sparse map needs (GBA-2r) - neither may flatter the kernel
        atf = min(stf, tf)
        out[name] = {"free_keyframes": n // 6, "points": int(len(p["Xw"])), "edges": int(len(p["edge_pose"])),
                     "robust": False,
                     "huber_on": {"wall_ms": wall_huber * 1e3, "gpu_ms": rh["info"]["gpu_ms"], "lm_trials": rh["info"]["lm_trials"],
                                  "chi2_final": rh["info"]["chi2_final"],
                                  "ms_per_solve": rh["info"]["solve_ms"] / max(rh["info"]["n_solves"], 1)},
                     "wall_ms": wall * 1e3, "gpu_ms": inf["gpu_ms"], "lm_trials": inf["lm_trials"],
                     "chi2_initial": inf["chi2_initial"], "chi2_final": inf["chi2_final"], "generator_s": gen_s,
                     "solver_path": int(inf["solver_path"]),
                     "solve": {"n": n, "ms_per_solve": ms, "tiles_in_skyline": inf["nnz_tiles"], "tiles_dense": T * (T + 1) // 2,
                               "structural_flop": sflop, "dense_flop": flop, "achieved_tflops": atf,
                               "skyline_tiles_tflops": stf, "dense_equivalent_tflops": tf, "peak_tflops": FP64_PEAK_TF,
                               "frac": atf / FP64_PEAK_TF, "bound": "mfma",
                               "evidence": "profiles/r5_gba_kernel_stats.csv, profiles/r5_gba_pmc_mfma.json (rocprofv3 "
                                           "--kernel-trace --stats and separate --pmc passes of tools/gba_bench.py)",
                               "note": "achieved = min(flop over the nonzero 96x96 tiles of the block skyline, n^3/3 + 2 n^2 "
                                       "of the un-padded system) / solve time (HIP events around the solve kernel)"}}
        # north_star's "PCG solve" of the reduced camera system, built and measured (so_ba_set_linear_solver, ba_pcg.hip): block-
        # Jacobi PCG over the nonzero 6 x 6 blocks of S, |r| / |b| <= 1e-7.  The direct solve stays the default: it is faster
        # on all four maps (the damped systems of the late LM iterations need hundreds of CG iterations)
        q = swarmmap_amd.Optimizer(device=dev)
        q.set_linear_solver("pcg", 1e-7, 4000)
        q.set_solve_timing(True)
        q.BundleAdjustment(p, nIterations=2, bRobust=False)
        t0 = time.perf_counter()
        rp = q.BundleAdjustment(p, nIterations=10, bRobust=False)
        wall_pcg = time.perf_counter() - t0
        pi = rp["info"]
        out[name]["pcg"] = {"wall_ms": wall_pcg * 1e3, "gpu_ms": pi["gpu_ms"], "lm_trials": pi["lm_trials"], "chi2_final": pi["chi2_final"],
                            "ms_per_solve": pi["solve_ms"] / max(pi["n_solves"], 1), "cg_iterations_per_solve": pi["pcg_iterations"] / max(pi["lm_trials"], 1),
                            "nonzero_6x6_blocks": pi["nnz_tiles"], "rel_tolerance": 1e-7,
                            "max_pose_entry_difference_to_direct": float(np.abs(rp["Tcw"] - r["Tcw"]).max()),
                            "evidence": "profiles/r5_pcg_bench.jsonl, r5_pcg_iterations.json (CPU study of the iteration counts), "
                                        "r5_pcg_spmv_probe.txt (time of one iteration on the block structure)"}
        q.close()
        del p
    o.close()
    return out


def _cpu_ranges(cpus):
    """{0,1,2,5} -> ["0-2", "5"]"""
    out, run = [], []
    for c in sorted(cpus):
        if run and c == run[-1] + 1:
            run.append(c)
        else:
            if run:
                out.append("%d-%d" % (run[0], run[-1]) if len(run) > 1 else str(run[0]))
            run = [c]
    if run:
        out.append("%d-%d" % (run[0], run[-1]) if len(run) > 1 else str(run[0]))
    return out


HEADLINE_MAX_BYTES = 6000  # the driver's parser lost round 5's 27.9 KB line (BENCH_r05.json: parsed null); r04's 19.6 KB parsed


def _strict(o):
    """NaN / inf -> null, numpy scalars -> python: the line must be strict JSON."""
    if isinstance(o, dict):
        return {str(k): _strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_strict(v) for v in o]
    if isinstance(o, (np.floating, float)):
        f = float(o)
        return f if np.isfinite(f) else None
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, np.ndarray):
        return _strict(o.tolist())
    return o


def _sig(x, n=5):
    """Scalars of the headline carry n significant digits (the full record keeps every bit)."""
    if isinstance(x, bool) or x is None or isinstance(x, (str, int)):
        return x
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    f = float(x)
    return float("%.*g" % (n, f)) if np.isfinite(f) else None


def headline(full, full_path):
    """The ONE line the driver parses: contract keys, config.workload + scalars, the dominant kernel's roofline,
    cpu_baseline, the two ATE figures and compact scalars of the other configs.  Everything else is in `full_path`."""
    g = lambda d, *ks: (lambda v: v)(__import__("functools").reduce(lambda a, k: a.get(k) if isinstance(a, dict) else None, ks, d))  # noqa: E731
    cfg = full["config"]
    cl = cfg.get("closed_loop") or {}
    lmms = cl.get("local_mapping_ms_per_keyframe") or {}
    roof = full["roofline"]
    hconfig = {
        "workload": cfg["workload_short"], "agents": cfg["agents"], "frame": cfg["frame"], "nfeatures": cfg["nfeatures"],
        "loop": "closed" if cl else "open", "keyframe_every": LBA_EVERY, "lba_edges": cfg["lba_edges"],
        "lba_free_keyframes": g(cl, "per_keyframe", "lba_free_keyframes"), "lba_fixed_keyframes": g(cl, "per_keyframe", "lba_fixed_keyframes"),
        "keypoints_per_frame": cfg["keypoints_per_frame"], "m2_matches_per_frame": cfg["m2_matches_per_frame"],
        "m1_matches_per_frame": cfg["m1_matches_per_frame"], "inliers_per_frame": cfg["inliers_per_frame"],
        "local_map_points_per_frame": cfg["local_map_points_per_frame"],
        "frame_ms_percentiles": cfg["frame_ms_percentiles"], "latency_ms_image_to_pose": cfg["latency_ms_image_to_pose"],
        "local_mapping_ms_per_keyframe": lmms.get("whole_job"), "so_bundle_adjust_ms": lmms.get("so_bundle_adjust"),
        "lba_ms_per_window": cfg["lba_ms_per_window"], "tracking_thread_waited_ms": g(cl, "whole_run", "tracking_thread_waited_ms"),
        "descriptor_exchanges": cfg["descriptor_exchanges"], "exchange": cfg["exchange"],
        "pose_kernel_ms_per_call": cfg["pose_kernel_ms_per_call"], "match_kernel_ms_per_frame": cfg["match_kernel_ms_per_frame"],
    }
    hroof = {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                      "algorithmic_flop_per_launch", "algorithmic_bytes_per_launch", "launches_in_timed_region",
                                      "event_timed_launches", "total_ms_in_timed_region") if k in roof}
    hroof["evidence"] = "profiles/r6_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this command), profiles/pmc_traffic.json"
    hroof["others"] = {r["kernel"]: {"bound": r["bound"].split(" ")[0], "frac": r["frac"], "avg_launch_ms": r["avg_launch_ms"], "traffic": r.get("traffic")}
                       for r in (full["roofline_secondary"], full["roofline_tertiary"])}
    head = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    head.update({"fps_per_agent": full["fps_per_agent"], "agents_per_gpu": full["agents_per_gpu"], "host_loop": full["host_loop"], "host_cpu": full.get("host_cpu"), "fleet_ticks": full.get("fleet_ticks"), "gpu_busy": full.get("gpu_busy"),
                 "launch": full["launch"][:120], "config": hconfig, "roofline": hroof})
    if "cpu_baseline" in full:
        c = full["cpu_baseline"]
        head["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": c["sample_short"]}
    for k in ("ate_rmse_vs_ground_truth", "ate_rmse_vs_oracle_chain"):
        if k in full:
            head[k] = full[k]
    if "ate_rmse" in full:
        voc = full["ate_rmse"].get("vs_oracle_chain") or {}
        head["ate"] = {"frames": full["ate_rmse"]["frames"], "unit": "m", "pixel_m": full["ate_rmse"]["pixel_m"],
                       "online_vs_ground_truth": g(full["ate_rmse"], "vs_ground_truth", "online", "sim3_aligned_m"),
                       "keyframes_vs_ground_truth": g(full["ate_rmse"], "vs_ground_truth", "keyframes", "sim3_aligned_m"),
                       "frames_with_different_counts_vs_oracle": voc.get("frames_with_different_match_or_inlier_counts"),
                       "max_pose_entry_difference_vs_oracle": voc.get("max_pose_entry_difference")}
    cf = full.get("configs")
    if cf:
        k3 = cf.get("kitti_stream_1241x376") or {}
        gb = cf.get("global_ba") or {}
        lb = cf.get("local_ba_windows") or {}
        head["other_configs"] = {
            "steady_state_frames_per_s": g(cf, "steady_state", "frames_per_s"), "steady_state_steps": g(cf, "steady_state", "steps"),
            "steady_state_lba_edges": g(cf, "steady_state", "lba_edges"),
            "kitti_frames_per_s": k3.get("frames_per_s"), "kitti_frame_ms_p90": g(k3, "frame_ms_percentiles", "p90"),
            "kitti_cpu_baseline_frames_per_s": g(k3, "cpu_baseline", "value"),
            "kitti_ate_rmse_vs_ground_truth": g(k3, "ate_rmse", "vs_ground_truth", "final", "sim3_aligned_m"),
            "reference_policy_frames_per_s": g(cf, "reference_policy", "frames_per_s"),
            "open_loop_frames_per_s": g(cf, "open_loop_synthetic_window", "frames_per_s"),
            "agents_per_gpu_frames_per_s": {k: v for k, v in (cf.get("agents_per_gpu") or {}).items() if k != "note"} or None,
            "agents_per_gpu_busy_percent": cf.get("agents_per_gpu_busy_percent"),
            "front_end_batched_frames_per_s": {k[7:]: v["frames_per_s"] for k, v in (cf.get("front_end_batched") or {}).items() if k.startswith("agents_")},
            "kf_scan_frac_int_valu": g(cf, "candidate_search", "store_8x512_kf_40pct_bound", "roofline", "frac"),
            "kf_scan_ms_4096_keyframes": g(cf, "candidate_search", "store_8x512_kf_40pct_bound", "scan_kernel_ms"),
            "lba_wall_ms": {k: v["wall_ms"] for k, v in lb.items()},
            "gba_wall_ms": {k: v["wall_ms"] for k, v in gb.items()},
            "gba_solve_ms": {k: v["solve"]["ms_per_solve"] for k, v in gb.items()},
            "gba_solve_frac_fp64_mfma": {k: v["solve"]["frac"] for k, v in gb.items()},
            "error": cf.get("error"), "seconds": cf.get("seconds")}
    head["full_record"] = full_path
    head = _sig(_strict(head))
    line = json.dumps(head, allow_nan=False, separators=(",", ":"))
    if len(line) > HEADLINE_MAX_BYTES:  # never again: drop the optional blocks before the contract keys go unparsed
        for k in ("other_configs", "ate", "launch", "host_loop"):
            head.pop(k, None)
            line = json.dumps(head, allow_nan=False, separators=(",", ":"))
            if len(line) <= HEADLINE_MAX_BYTES:
                break
    return line


def write_full_record(full):
    """The whole record (configs, every roofline object, the ATE block) beside the profiles; a second copy under
    gpurun_out/ so that a gpurun call brings it back."""
    txt = json.dumps(_strict(full), allow_nan=False, indent=1)
    rel = os.path.join("profiles", "last_bench_full.json")
    for path in (os.path.join(ROOT, rel), os.path.join(ROOT, "gpurun_out", "last_bench_full.json")):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(txt + "\n")
        except OSError:
            pass
    return rel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--size", default="euroc", choices=["euroc", "kitti"])
    ap.add_argument("--exchange-every", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the configs[3] / configs[4] sub-records (KITTI-sized stream, LBA-S/M/L, global BA)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the full record as a stdout line BEFORE the headline (it is always written to profiles/last_bench_full.json)")
    ap.add_argument("--lockstep", action="store_true",
                    help="with --agents-per-gpu A > 1: ONE thread drives the A agents frame by frame (so_fleet_run: searches "
                         "of all agents in flight together, PoseOptimization of all agents in one launch) instead of A "
                         "independent tracking threads")
    ap.add_argument("--fleet-threads", type=int, default=1,
                    help="with --lockstep: split the agents of a GPU over this many driving threads (a lockstep fleet each)")
    ap.add_argument("--agents-per-gpu", type=int, default=1,
                    help="run this many independent agents (tracking + local-mapping thread pairs, own contexts and "
                         "streams) on each GPU; value stays the aggregate frames/s over all agents")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or bool(os.environ.get("SWARMORB_BENCH_FORCE_DIST"))  # (the latter: 1-rank RCCL self-test)
    if distributed:
        for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(key, val)  # torchrun sets all four; the 1-rank self-test runs without it
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = local_rank if distributed else 0
    torch.cuda.set_device(dev)
    GPU_BUSY.device_index = dev
    torch.cuda.synchronize()  # the runtime's threads exist from here on
    from swarmmap_amd import _lib as _so_lib
    # thread placement: this rank's threads behind one L3 (one per agent of the rank) next to its GPU; "pinned_cpus" in
    # the JSON line says which (null: left to the OS - SWARMORB_NO_PIN=1 or a single-node host)
    pinned_cpus = _so_lib.pin_process_near_device(dev, max(1, args.agents_per_gpu))

    import ctypes
    libc = ctypes.CDLL(None)

    # Independent bench PROCESSES sharing one GPU (tools/agents_processes.py) meet at the same points through a directory:
    # BENCH_FILE_BARRIER="dir:index:count" - every process drops a file per barrier and waits until all of them are there, so
    # that the timed regions of all processes cover the same interval and their rates may be added
    fb = os.environ.get("BENCH_FILE_BARRIER")
    fb_state = {"n": 0}

    def barrier():
        if distributed:
            torch.distributed.barrier()
            libc.fflush(None)  # RCCL prints a version banner through C stdio when a communicator comes up: out it goes
                               # now, on every rank, so that rank 0's JSON line stays the last line of the job's stdout
        torch.cuda.synchronize()
        if fb:
            d, idx, cnt = fb.rsplit(":", 2)
            fb_state["n"] += 1
            open(os.path.join(d, "b%d.%s" % (fb_state["n"], idx)), "w").close()
            t_end = time.time() + 600.0
            while sum(os.path.exists(os.path.join(d, "b%d.%d" % (fb_state["n"], i))) for i in range(int(cnt))) < int(cnt):
                if time.time() > t_end:
                    raise RuntimeError("file barrier: a peer process did not arrive")
                time.sleep(0.0005)

    euroc = args.size == "euroc"
    size = synth.EUROC if euroc else synth.KITTI
    K = synth.EUROC_K if euroc else synth.KITTI_K
    dist = synth.EUROC_DIST if euroc else None  # code/Examples/Monocular/EuRoC.yaml has a lens model, KITTI00-02.yaml none
    nfeatures = 1000 if euroc else 2000
    A = max(1, args.agents_per_gpu)
    lba_window = synth.make_ba_case("LBA-M", seed=100 + rank)
    m1 = None
    xchg = None
    if distributed:  # RCCL all-gather + keyframe store + candidate search behind the C ABI (so_exchange_*);
        from swarmmap_amd.exchange import StoreExchange  # torch.distributed only carries the communicator's id
        xchg = StoreExchange.from_process_group(dev, nfeatures + 24, records_per_tick=1, store_keyframes=STORE_KEYFRAMES)
        if world == 1:
            # 1-rank self-test: nobody sends anything, so the store gets what seven peers would have sent by now
            # (SWARMORB_BENCH_PREFILL keyframes each) - the scan inside the tick then has something to read
            prefill_store(xchg.store, 7, int(os.environ.get("SWARMORB_BENCH_PREFILL", "64")), nfeatures)

    try:
        if args.lockstep and A > 1 and args.fleet_threads > 1:
            dt, st, n_cand, frames, log0 = run_fleet_threads(dev, size, K, dist, nfeatures, args.steps, args.warmup, SEED_BASE + rank,
                                                             lba_window, barrier, A, args.fleet_threads)
        elif args.lockstep and A > 1:
            dt, st, n_cand, frames, log0 = run_fleet(dev, size, K, dist, nfeatures, args.steps, args.warmup, SEED_BASE + rank,
                                                     lba_window, barrier, A)
        else:
            dt, st, n_cand, frames, log0 = run_stream(dev, size, K, dist, nfeatures, args.steps, args.warmup, SEED_BASE + rank,
                                                      lba_window, barrier, A, xchg, args.exchange_every, m1,
                                                      live_steps=LIVE_STEPS if rank == 0 else 0)
    except swarmmap_amd.SwarmOrbError as e:
        if "timed out" not in str(e):
            raise
        # a peer did not enter an exchange tick within the budget (SWARMORB_COLLECTIVE_TIMEOUT_MS): the exchange handle
        # is dead.  No result, no hang, no re-exec of a process that holds the GPU: one JSON line with the error, exit != 0
        # (os._exit: the other ranks may be stuck in torch.distributed's own barrier, which a normal exit would join)
        print(json.dumps({"metric": "frames/sec/agent (tracking+localBA)", "value": None, "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "error": "rank %d: %s" % (rank, e)}), flush=True)
        os._exit(3)
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        steps = args.steps
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pmc = json.load(open(tpath)) if os.path.exists(tpath) else {}
        stage, inv_scale = extractor_stage_profile(dev, frames, size, nfeatures)
        rec, roof_fast, roof_pose = stream_record(size, nfeatures, steps, dt, st, n_cand, stage, inv_scale, pmc, A)
        # reduced-camera-system solve of local BA: dense Cholesky of a (6 n_free)^2 FP64 system, n^3/3 + 2 n^2 flop
        cl_rec = closed_loop_record(st, steps) if st.get("closed") else None
        n_red = 6 * (int(round(cl_rec["per_keyframe"]["lba_free_keyframes"])) if cl_rec else int((lba_window["fixed"] == 0).sum()))
        solve_flop = n_red ** 3 / 3.0 + 2.0 * n_red ** 2
        solve_ms = st["solve_ms"] / max(st["n_solves"], 1)
        solve_tf = solve_flop / (solve_ms * 1e-3) / 1e12 if solve_ms > 0 else 0.0
        # ba_kernels.hip launch_ba_solve: 4..29 free keyframes -> single-workgroup MFMA solver, 30..43 -> its
        # register-resident sibling (ba_dense.hip)
        solve_kernel = ("ba_solve_mfma_kernel" if 24 <= n_red <= 174 else
                        "ba_solve_mfma_reg_kernel" if 174 < n_red <= 258 else "ba_solve_la_kernel")
        roof_solve = {"bound": "mfma", "kernel": solve_kernel, "achieved": solve_tf, "peak": FP64_PEAK_TF,
                      "unit": "TFLOP/s", "frac": solve_tf / FP64_PEAK_TF,
                      "traffic": pmc.get(solve_kernel, {}).get("hbm_bytes_per_launch"),
                      "algorithmic_flop_per_launch": solve_flop, "avg_launch_ms": solve_ms,
                      "total_ms_in_timed_region": solve_ms * st.get("lba_trials", st["n_solves"]),
                      "event_timed_launches": st["n_solves"],
                      "note": "150x150 FP64 system per launch: latency-bound by construction (DESIGN.md 5); peak is "
                              "AMD's FP64 datasheet figure (the guide lists no FP64 MFMA peak)"}
        # the dominant kernel = the one with the largest accumulated HIP-event time inside the timed region
        ranked = sorted([(roof_pose["total_ms_in_timed_region"], roof_pose), (roof_solve["total_ms_in_timed_region"], roof_solve),
                         (roof_fast["total_ms_in_timed_region"], roof_fast)], key=lambda kv: -kv[0])
        out = {
            "metric": "frames/sec/agent (tracking+localBA), aggregate over agents (per agent: fps_per_agent); ATE RMSE in ate_rmse_*",
            "value": steps * world * A / dt, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (extract, match) + f64 (PoseOptimization, local BA)", "data": "synthetic",
            "host_loop": "c++ (swarmmap_amd/host/replay.cc)" + (", %d agents in lockstep on one thread (so_fleet_run)" % A
                                                                  if args.lockstep and A > 1 else ""),
            "fps_per_agent": steps / dt, "agents_per_gpu": A,
            "fleet_ticks": st.get("fleet_ticks"),
            "gpu_busy": st.get("gpu_busy"),  # the driver's busy figure over the timed region (coarse: "something was running")
            "host_cpu": st.get("cgroup"),  # cores used / the container's quota / CFS throttling inside the timed region
            "pinned_cpus": None if not pinned_cpus else ",".join(_cpu_ranges(pinned_cpus)),
            "config": dict({
                "workload": ("BASELINE.json configs[1]+[2] on one GPU per agent: 752x480 EuRoC-sized stream seen through "
                             "the EuRoC lens model, each step = host->HBM image upload + HIP ORB extract (nFeatures %d) + "
                             "UndistortKeyPoints/AssignFeaturesToGrid on the device + SearchByProjection(last frame) -> "
                             "PoseOptimization over its matches -> isInFrustum + SearchByProjection(local map) -> "
                             "PoseOptimization -> third PoseOptimization (TrackReferenceKeyFrame fallback) -> keyframe decision, "
                             "chained device-resident; on a local-mapping thread, as in the reference, every %d-th frame becomes a "
                             "keyframe: " % (nfeatures, LBA_EVERY) +
                             ("SearchForTriangulation against the last <= 20 keyframes -> triangulation -> new map points; Fuse into "
                              "them and back -> AddObservation / Replace; HIP LocalBA over the keyframe's OWN window -> SetPose / "
                              "SetWorldPos / EraseObservation / UpdateNormalAndDepth; the results are in the tracked map %d frames "
                              "later (CLOSED loop)" % LBA_EVERY if st.get("closed") else
                              "SearchForTriangulation against the last <= 20 keyframes, Fuse into them and back, then HIP LocalBA "
                              "(a fixed synthetic LBA-M window; open loop)")) if euroc else
                            "KITTI-sized 1241x376 stream, nFeatures %d, same chained per-frame path" % nfeatures,
                "workload_short": ("BASELINE.json configs[1]+[2], one agent per GPU: synthetic %dx%d stream (%s), step = one tracked frame: "
                                   "image upload + HIP ORB extract (nFeatures %d) + undistort/grid + SearchByProjection(last frame) + "
                                   "PoseOptimization + SearchByProjection(local map) + 2 PoseOptimization on the tracking thread; every "
                                   "%d-th frame a keyframe on a local-mapping thread: %s"
                                   % (size[0], size[1], "EuRoC lens model" if euroc else "KITTI-sized", nfeatures, LBA_EVERY,
                                      "SearchForTriangulation x <=20 + triangulation + Fuse x <=40 + HIP LocalBundleAdjustment over its own "
                                      "window, results in the tracked map %d frames later (closed loop)" % LBA_EVERY if st.get("closed") else
                                      "SearchForTriangulation + Fuse + HIP LocalBundleAdjustment of a fixed LBA-M window (open loop)")),
                "agents": world * A,
                "lba_edges": int(round(cl_rec["per_keyframe"]["lba_edges"])) if cl_rec else int(len(lba_window["edge_pose"])),
                "lba_window": "the chain's own keyframes (config.closed_loop)" if cl_rec else "synthetic LBA-M (SURVEY 8d)",
                "closed_loop": cl_rec,
                "descriptor_exchanges": st.get("n_xchg", 0),
                "exchange": {"ticks": st.get("n_xchg", 0), "candidates": st.get("xchg_candidates", 0),
                             "store_keyframes_at_end": st.get("xchg_store_keyframes", 0),
                             "scan_kernel_ms_per_tick": st.get("xchg_scan_ms", 0.0) / max(st.get("n_xchg", 0), 1),
                             "descriptor_pairs_per_tick": st.get("xchg_pairs", 0.0) / max(st.get("n_xchg", 0), 1),
                             "wall_ms_per_tick": st.get("xchg_ms", 0.0) / max(st.get("n_xchg", 0), 1)}},
                **{k: v for k, v in rec.items() if k != "frames_per_s"}),
            "launch": ("torch.distributed, %d rank(s): every %d-th frame is followed by an exchange tick (RCCL all-gather of the "
                       "newest keyframe record + store append + candidate search) INSIDE the timed region - a 1-rank line "
                       "launched this way is therefore not the plain `python bench.py` line" % (world, args.exchange_every)
                       if distributed else "single process, no exchange ticks"),
            "roofline": ranked[0][1],
            "roofline_secondary": ranked[1][1],
            "roofline_tertiary": ranked[2][1],
        }
        if world == 1 and A == 1 and not args.no_configs:
            cfgs = {}
            t0 = time.perf_counter()
            try:
                # configs[3]: KITTI-sized stream (1241x376, nFeatures 2000), same chained path, one agent
                k_window = synth.make_ba_case("LBA-M", seed=101)
                ksteps = 150
                kdt, kst, kcand, kframes, klog = run_stream(dev, synth.KITTI, synth.KITTI_K, None, 2000, ksteps, 20, 20221001,
                                                            k_window, barrier, live_steps=LIVE_STEPS)
                kstage, kinv = extractor_stage_profile(dev, kframes, synth.KITTI, 2000, 32)
                krec, kfast, kpose = stream_record(synth.KITTI, 2000, ksteps, kdt, kst, kcand, kstage, kinv, pmc)
                krec["roofline_fast_score"] = {k: kfast[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "avg_launch_ms")}
                krec["algorithmic_front_end_bytes_per_frame"] = 11.83e6
                kor = None
                if not args.no_cpu_baseline:  # the same KITTI-sized chain through the CPU oracle, a shorter sample
                    krec["cpu_baseline"] = cpu_baseline(kframes, synth.KITTI_K, None, 2000, k_window, synth.KITTI, budget_s=8.0)
                    kor = krec["cpu_baseline"].pop("_trajectory", None)
                if kst.get("closed"):
                    krec["closed_loop"] = closed_loop_record(kst, ksteps)
                    krec["ate_rmse"] = ate_records(klog, kst["cl"], kst["stream"], synth.KITTI_K, kor)
                cfgs["kitti_stream_1241x376"] = krec
                del kframes
                if st.get("closed"):
                    # the same closed loop over 400 timed frames (the driver's 20 cover 4 keyframes of a young map: ~13 k LBA
                    # edges against 22-26 k in steady state)
                    ssteps = 400
                    sdt, sst, _, _, _ = run_stream(dev, size, K, dist, nfeatures, ssteps, 40, SEED_BASE, lba_window, barrier)
                    srec = closed_loop_record(sst, ssteps)
                    cfgs["steady_state"] = {"frames_per_s": ssteps / sdt, "ms_per_frame": sdt / ssteps * 1e3, "steps": ssteps, "warmup": 40,
                                            "lba_edges": srec["per_keyframe"]["lba_edges"], "inliers_per_frame": sst["n_inliers"] / ssteps,
                                            "local_mapping_ms_per_keyframe": srec["local_mapping_ms_per_keyframe"]["whole_job"],
                                            "so_bundle_adjust_ms": srec["local_mapping_ms_per_keyframe"]["so_bundle_adjust"]}
                if st.get("closed"):
                    # the reference's own policy when tracking outpaces mapping (Tracking.cc:810-892, LocalMapping.cc:581-583):
                    # results arrive when ready, a keyframe only while local mapping is idle, InterruptBA otherwise -
                    # timing-dependent by construction, so it is a second key, not the headline
                    psteps = 200
                    pdt, pst, _, _, plog = run_stream(dev, size, K, dist, nfeatures, psteps, 20, 20221001, lba_window, barrier, policy=1)
                    cfgs["reference_policy"] = {
                        "frames_per_s": psteps / pdt, "ms_per_frame": pdt / psteps * 1e3, "inliers_per_frame": pst["n_inliers"] / psteps,
                        "policy": "NeedNewKeyFrame: a keyframe after >= %d frames or when inliers fall below 0.7 of the last "
                                  "keyframe's - only while local mapping is idle; a busy local mapper gets InterruptBA (the stop "
                                  "flag so_bundle_adjust polls between LM trials) and the keyframe waits" % LBA_EVERY,
                        "closed_loop": closed_loop_record(pst, psteps)["whole_run"],
                        "ate_rmse": ate_records(plog, pst["cl"], pst["stream"], K)}
                    # round 4's workload: a fixed synthetic LBA-M window per keyframe, nothing fed back
                    odt, ost, _, _, _ = run_stream(dev, size, K, dist, nfeatures, psteps, 20, 20221001, lba_window, barrier, closed=False)
                    cfgs["open_loop_synthetic_window"] = {
                        "frames_per_s": psteps / odt, "ms_per_frame": odt / psteps * 1e3, "lba_edges": int(len(lba_window["edge_pose"])),
                        "lba_thread_busy_ms_per_frame": ost["lba_busy_ms"] / psteps, "inliers_per_frame": ost["n_inliers"] / psteps}
                if st.get("closed"):
                    # several agents on THIS GPU in lockstep (the metric's "1/2/4/8 agents" on the hardware at hand): one thread
                    # drives the agents' tracking with the stages of all agents as one chain of launches per stage
                    # (so_track_group), their local bundle adjustments merged per round (so_ba_group), a local-mapping thread each
                    apg = {"1": cfgs["steady_state"]["frames_per_s"]}
                    fill, busy_apg = {}, {}
                    for A_ in (2, 4, 8, 16):
                        fdt, fst, _, _, _ = run_fleet(dev, size, K, dist, nfeatures, 200, 20, SEED_BASE, lba_window, barrier, A_)
                        apg[str(A_)] = 200 * A_ / fdt
                        fill[str(A_)] = fst["fleet_ticks"]["agents_per_tick"]
                        busy_apg[str(A_)] = (fst.get("gpu_busy") or {}).get("mean_percent")
                    cfgs["agents_per_gpu"] = dict(apg, note="aggregate frames/s, closed loop, --lockstep (elastic ticks: an agent whose frame "
                                                            "waits for its local-mapping packet sits the tick out); 1 = the headline's steady-state figure")
                    cfgs["agents_per_gpu_tick_fill"] = fill  # agents a tick took on average
                    cfgs["agents_per_gpu_busy_percent"] = busy_apg  # sysfs gpu_busy_percent, mean over the fleet's timed region
                cfgs["front_end_batched"] = batched_front_end_records(dev)
                cfgs["candidate_search"] = candidate_search_records(dev)
                cfgs["local_ba_windows"] = lba_records(dev)
                # GBA-1 / GBA-2: SURVEY 8d's sizes with every camera looking at one cloud (reduced system nearly dense);
                # GBA-1r / GBA-2r: the same sizes as merged street-grid maps of 4 / 8 agents (banded + inter-agent links)
                cfgs["global_ba"] = gba_records(dev, ["GBA-1", "GBA-2", "GBA-1r", "GBA-2r"])
            except Exception as e:  # noqa: BLE001 - the headline stays valid; the failure is reported in the line
                cfgs["error"] = repr(e)
            cfgs["seconds"] = time.perf_counter() - t0
            out["configs"] = cfgs
        oracle_traj = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, K, dist, nfeatures, lba_window, size)
            oracle_traj = out["cpu_baseline"].pop("_trajectory", None)
        if st.get("closed"):
            # ATE RMSE of the whole tracked stream (prefill + warm-up + timed frames: one trajectory), BASELINE.json's second half
            ate = ate_records(log0, st["cl"], st["stream"], K, oracle_traj)
            out["ate_rmse"] = ate
            out["ate_rmse_vs_ground_truth"] = ate["vs_ground_truth"]["final"]["sim3_aligned_m"]
            out["ate_rmse_vs_oracle_chain"] = ate.get("vs_oracle_chain", {}).get("online_unaligned_m")
            if oracle_traj is not None:
                out["cpu_baseline"]["ate_rmse_vs_ground_truth_online"] = ate["vs_oracle_chain"]["oracle_vs_ground_truth_online"]
        full_path = write_full_record(out)
        if args.full_line:
            print(json.dumps(_strict(out), allow_nan=False), flush=True)
        libc.fflush(None)  # C-side stdout (RCCL's version banner) goes out first: the JSON line is the last line
        print(headline(out, full_path), flush=True)
    if xchg is not None:
        xchg.close()
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
