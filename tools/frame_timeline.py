#!/usr/bin/env python3
"""Per-frame timeline of the extraction chain from a rocprofv3 --kernel-trace run of tools/extract_latency.py:
for every kernel of the chain its mean start offset from the frame's first kernel, its duration and the idle gap in
front of it.  Usage: frame_timeline.py <dir with *kernel_trace.csv> [first kernel name, default ingest]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "ingest"
p = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
for r in csv.DictReader(open(p)):
    m = re.search(r"so::(?:\(anonymous namespace\)::)?(\w+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
rows.sort()
frames, cur = [], None
for s, e, n in rows:
    if n.startswith(first):
        cur = []
        frames.append(cur)
    if cur is not None:
        cur.append((s, e, n))
frames = frames[len(frames) // 4:]  # skip warm-up
shape = defaultdict(int)
for f in frames:
    shape[tuple(n for _, _, n in f)] += 1
names = max(shape, key=shape.get)
sel = [f for f in frames if tuple(n for _, _, n in f) == names]
print("%d frames of %d with the common chain of %d kernels" % (len(sel), len(frames), len(names)))
print("%-34s %9s %9s %9s" % ("kernel", "start_us", "dur_us", "gap_us"))
tot_gap = 0.0
for i, n in enumerate(names):
    st = sum(f[i][0] - f[0][0] for f in sel) / len(sel) / 1e3
    du = sum(f[i][1] - f[i][0] for f in sel) / len(sel) / 1e3
    gp = sum(f[i][0] - f[i - 1][1] for f in sel) / len(sel) / 1e3 if i else 0.0
    tot_gap += gp
    print("%-34s %9.2f %9.2f %9.2f" % (n, st, du, gp))
end = sum(f[-1][1] - f[0][0] for f in sel) / len(sel) / 1e3
print("chain %.2f us, of which idle between kernels %.2f us" % (end, tot_gap))
