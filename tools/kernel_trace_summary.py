#!/usr/bin/env python3
"""Print a rocprofv3 *kernel_stats.csv as `kernel calls avg_us min_us`.  Usage: kernel_trace_summary.py <dir or csv>"""
import csv
import glob
import os
import re
import sys

p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
for r in csv.DictReader(open(p)):
    m = re.search(r"so::(?:\(anonymous namespace\)::)?(\w+)", r["Name"])
    print("%-34s %6s %9.2f %9.2f" % (m.group(1) if m else r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
