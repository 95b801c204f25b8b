#!/usr/bin/env python3
"""A agents on ONE GPU as A PROCESSES (the deployment: one process per agent; `bench.py --agents-per-gpu A` runs them as
thread pairs of one process, where the HIP runtime's locks make them contend).  Each process runs `bench.py --steps S` with
its own L3 group (SWARMORB_PIN_SLOT_BASE); the processes enter and leave their timed regions together (a file barrier,
BENCH_FILE_BARRIER in bench.py: without it the regions of processes that start a second apart overlap only partly and the
sum of their rates overstates the aggregate), so the per-process rates may be added.
Usage: python tools/agents_processes.py [A ...]        (default 1 2 4 8)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(A, steps=300):
    import tempfile
    procs = []
    bdir = tempfile.mkdtemp(prefix="bench_barrier_")  # the processes' timed regions start and end together (bench.py: BENCH_FILE_BARRIER)
    for a in range(A):
        env = dict(os.environ, SWARMORB_PIN_SLOT_BASE=str(a), HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_FILE_BARRIER="%s:%d:%d" % (bdir, a, A))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "20", "--no-configs",
                                       "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    rates, jobs = [], []
    for p in procs:
        out = p.communicate(timeout=900)[0].strip().splitlines()
        d = json.loads(out[-1])
        rates.append(d["value"])
        jobs.append(d["config"]["closed_loop"]["local_mapping_ms_per_keyframe"]["whole_job"])
    return {"processes": A, "frames_per_s_aggregate": sum(rates), "frames_per_s_per_agent": rates, "local_mapping_ms_per_keyframe": jobs}


if __name__ == "__main__":
    for A in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]:
        print(json.dumps(run(A)), flush=True)
