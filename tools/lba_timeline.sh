#!/bin/bash
# Where a local-BA call's GPU time goes: kernel-trace of the closed-loop bench, then per so_bundle_adjust call (a run of ba_*
# kernels on the local-mapping stream) the span, the summed kernel time and the idle gaps between dependent launches.
#   bash tools/lba_timeline.sh
set -u
export TMPDIR=/tmp
out=gpurun_out/tl_$$
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-configs > $out/log.txt 2>&1
f=$(find $out -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ba = [r for r in rows if "ba_" in r["Kernel_Name"] and "pose_opt" not in r["Kernel_Name"]]
# split into calls: a gap of more than 300 us between consecutive ba_ kernels starts a new call
calls, cur = [], []
for r in ba:
    if cur and int(r["Start_Timestamp"]) - int(cur[-1]["End_Timestamp"]) > 300000:
        calls.append(cur); cur = []
    cur.append(r)
if cur: calls.append(cur)
calls = [c for c in calls if len(c) > 20]
print("so_bundle_adjust calls seen:", len(calls))
tot = collections.Counter(); cnt = collections.Counter(); gaps = []; spans = []; busy = []; nk = []
for c in calls[2:]:
    spans.append((int(c[-1]["End_Timestamp"]) - int(c[0]["Start_Timestamp"])) / 1e3)
    b = 0
    for i, r in enumerate(c):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = r["Kernel_Name"].split("(")[0].split("::")[-1]
        tot[name] += d; cnt[name] += 1; b += d
        if i: gaps.append((int(r["Start_Timestamp"]) - int(c[i - 1]["End_Timestamp"])) / 1e3)
    busy.append(b); nk.append(len(c))
n = max(len(spans), 1)
print("per call: span %.1f us, kernels %.1f us in %.1f launches, idle between launches %.1f us (mean gap %.2f us, median %.2f)" % (
    sum(spans) / n, sum(busy) / n, sum(nk) / n, (sum(spans) - sum(busy)) / n, sum(gaps) / max(len(gaps), 1), sorted(gaps)[len(gaps) // 2]))
for k, v in tot.most_common():
    print("  %-34s %6.1f launches/call  %6.2f us each  %7.1f us/call" % (k, cnt[k] / n, v / cnt[k], v / n))
P
rm -rf $out
