#!/usr/bin/env python3
"""bench.py — frames/s of the per-frame hot path (one agent per GPU) on synthetic EuRoC-sized streams.

Contract (see the task prompt): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU, RCCL).
A step = one frame of one agent through the hot path, inputs already resident in HBM.  Rank 0 prints ONE
JSON line.  Agents are independent: weak scaling, no data-path collective per frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (device memory, barrier, RCCL plumbing only)

import swarmmap_amd  # noqa: E402
from swarmmap_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def level_pixels(ex, w, h):
    """Sum of pyramid level pixels = algorithmic bytes the FAST kernel must read once."""
    inv = ex.GetInverseScaleFactors()
    return int(sum(int(np.rint(np.float32(w) * s)) * int(np.rint(np.float32(h) * s)) for s in inv))


def cpu_baseline(frames, nfeatures, budget_s=12.0):
    """The CPU oracle ("port": the reference has no CPU extractor) timed on the host, 1 thread."""
    from oracle import oracle_py
    cfg = oracle_py.config(nfeatures)
    oracle_py.extract(cfg, frames[0])  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        oracle_py.extract(cfg, frames[n % len(frames)])
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 400:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames 752x480, ORB extract (nFeatures %d, 8 levels, FAST 20/7), CPU oracle -O2, "
                      "1 thread, host has %d cores" % (n, nfeatures, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--size", default="euroc", choices=["euroc", "kitti"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = local_rank if distributed else 0
    torch.cuda.set_device(dev)

    size = synth.EUROC if args.size == "euroc" else synth.KITTI
    nfeatures = 1000 if args.size == "euroc" else 2000
    w, h = size
    # each agent sees its own seeded stream; 32 distinct frames cycle, resident in HBM before timing
    stream = synth.FrameStream(seed=20221001 + rank, size=size)
    host_frames = [stream.frame(t) for t in range(32)]
    dev_frames = [torch.from_numpy(f).cuda(dev) for f in host_frames]
    torch.cuda.synchronize()

    ex = swarmmap_amd.ORBextractor(nfeatures, 1.2, 8, 20, 7, device=dev)

    def step(i):
        d = dev_frames[i % len(dev_frames)]
        return ex.run_device(d.data_ptr(), w, h, w)

    for i in range(args.warmup):
        step(i)
    ex.set_profiling(True)
    stage_ms = {}
    n_kp = 0
    n_cand = 0

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        kps, _ = step(i)
        n_kp += len(kps)
        for k, v in ex.profile().items():
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    for l in range(8):
        n_cand += len(ex.candidates(l)[0])

    if rank == 0:
        steps = args.steps
        total_frames = steps * world
        fast_ms = stage_ms["fast_score"] / steps
        alg_bytes = level_pixels(ex, w, h) + 8 * n_cand  # read every level pixel once + 8 B per candidate
        achieved = alg_bytes / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else 0.0
        out = {
            "metric": "frames/sec (ORB front-end per frame; aggregate over agents, per-agent = value/n_gpus)",
            "value": total_frames / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "fps_per_agent": steps / dt,
            "config": {"workload": "EuRoC-sized 752x480 single-agent-per-GPU stream, HIP ORB extract "
                                   "(nFeatures %d, 8 levels, 1.2, FAST 20/7); BASELINE.json configs[1]" % nfeatures
                       if args.size == "euroc" else
                       "KITTI-sized 1241x376 stream, HIP ORB extract (nFeatures %d)" % nfeatures,
                       "agents": world, "frame": [w, h], "keypoints_per_frame": n_kp / steps,
                       "stage_ms_per_frame": {k: v / steps for k, v in stage_ms.items()}},
            "roofline": {"bound": "hbm", "kernel": "fast_score_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": fast_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host_frames, nfeatures)
        print(json.dumps(out), flush=True)
    ex.close()
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
