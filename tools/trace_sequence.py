#!/usr/bin/env python3
"""Kernel sequence of a rocprofv3 --kernel-trace run: start offset, duration and the idle gap in front of every kernel
between two occurrences of a marker kernel.  Usage: trace_sequence.py <dir> <marker kernel> [occurrence, default 3]"""
import csv
import glob
import os
import re
import sys

d, marker = sys.argv[1], sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
for r in csv.DictReader(open(p)):
    m = re.search(r"so::(?:\(anonymous namespace\)::)?(\w+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
a, b = idx[occ], idx[occ + 1]
t0 = rows[a][0]
prev_end = None
busy = 0
for s, e, n in rows[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-34s %9.2f %9.2f %9.2f" % (n, (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev_end = e
    busy += e - s
print("span %.2f us, kernels %.2f us" % ((rows[b][0] - t0) / 1e3, busy / 1e3))
