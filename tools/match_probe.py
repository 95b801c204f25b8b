"""Per-call time of the two tracking matchers (M1 / M2) through the ctypes mirror: wall, enqueue, wait, kernel.
Developer tool:  python tools/match_probe.py  (on the GPU box)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from swarmmap_amd import synth
from swarmmap_amd.matcher import ORBmatcher, FrameView
m = ORBmatcher(0.8, True)
fr, mps = synth.make_m1_case(1, 1000, 2000)
F = FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"], fr["excluded"])
for _ in range(20): m.SearchByProjectionMapPoints(F, mps, 1.0)
s0 = m.last_stats(); N = 300
t0 = time.perf_counter()
for _ in range(N): m.SearchByProjectionMapPoints(F, mps, 1.0)
t1 = time.perf_counter()
s1 = m.last_stats()
print(s1); print("M1 wall/call us", (t1 - t0) / N * 1e6, "enqueue us", (s1["enqueue_ms"] - s0["enqueue_ms"]) / N * 1e3, "wait us", (s1["wait_ms"] - s0["wait_ms"]) / N * 1e3,
      "launches/call", (s1["launches"] - s0["launches"]) / N, "bytes/call", (s1["staged_bytes"] - s0["staged_bytes"]) / N, "kernel ms", m.last_kernel_ms())
fr, last = synth.make_m2_case(2, 1000, 1000)
F = FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"], fr["excluded"])
m2 = ORBmatcher(0.9, True)
for _ in range(20): m2.SearchByProjectionLastFrame(F, last, 15.0)
s0 = m2.last_stats()
t0 = time.perf_counter()
for _ in range(N): m2.SearchByProjectionLastFrame(F, last, 15.0)
t1 = time.perf_counter()
s1 = m2.last_stats()
print(s1); print("M2 wall/call us", (t1 - t0) / N * 1e6, "enqueue us", (s1["enqueue_ms"] - s0["enqueue_ms"]) / N * 1e3, "wait us", (s1["wait_ms"] - s0["wait_ms"]) / N * 1e3,
      "launches/call", (s1["launches"] - s0["launches"]) / N, "bytes/call", (s1["staged_bytes"] - s0["staged_bytes"]) / N, "kernel ms", m2.last_kernel_ms())
