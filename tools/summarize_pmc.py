#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel name, number of dispatches and the mean
counter value per dispatch.  Usage: summarize_pmc.py <counter_collection.csv> [out.csv]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in rows:
    name = r.get("Kernel_Name") or r.get("Kernel Name") or r.get("kernel_name")
    cname = r.get("Counter_Name") or r.get("Counter Name")
    val = float(r.get("Counter_Value") or r.get("Counter Value") or 0)
    a = acc[name][cname]
    a[0] += 1
    a[1] += val
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "total"])
for name in sorted(acc):
    for cname, (n, tot) in sorted(acc[name].items()):
        w.writerow([name[:100], cname, n, tot / n, tot])
