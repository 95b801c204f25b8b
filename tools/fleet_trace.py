#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of a multi-agent run: kernels per second, per-queue busy fractions, how many
kernels run at once, the largest kernel families.  Usage: fleet_trace.py <dir with *kernel_trace.csv> [last_fraction=0.4]
(only the last fraction of the trace is looked at: the timed region comes last in bench.py --no-configs --no-cpu-baseline)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
p = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
for r in csv.DictReader(open(p)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
rows.sort()
t_end = max(r[1] for r in rows)
t_beg = min(r[0] for r in rows)
cut = t_end - int((t_end - t_beg) * frac)
rows = [r for r in rows if r[0] >= cut]
span = (t_end - cut) * 1e-9
print("window %.3f s, %d kernels = %.0f kernels/s" % (span, len(rows), len(rows) / span))
busy = defaultdict(int)
fam = defaultdict(lambda: [0, 0])
for s, e, q, name in rows:
    busy[q] += e - s
    m = re.search(r"so::(?:\(anonymous namespace\)::)?(\w+)", name)
    k = m.group(1) if m else name[:40]
    fam[k][0] += 1
    fam[k][1] += e - s
print("queues:", {q: round(b * 1e-9 / span, 3) for q, b in sorted(busy.items())})
# concurrency: sweep
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, hist = 0, ev[0][0], defaultdict(int)
for t, dlt in ev:
    hist[cur] += t - last
    last = t
    cur += dlt
tot = sum(hist.values())
print("kernels in flight (fraction of time):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print("sum of kernel durations / wall = %.2f" % (sum(e - s for s, e, _, _ in rows) * 1e-9 / span))
for k, (n, ns) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:22]:
    print("%-36s %7d calls %8.2f us avg  %6.1f ms total" % (k, n, ns / n / 1e3, ns / 1e6))
