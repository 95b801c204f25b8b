#!/bin/bash
# Per-kernel time of the closed-loop bench:  bash tools/kernel_stats.sh [n=14]   (rocprofv3 --kernel-trace --stats, top n rows)
set -u
export TMPDIR=/tmp
out=gpurun_out/ks_$$
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 200 --no-cpu-baseline --no-configs > $out/log.txt 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" "${1:-14}" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2])]:
    print("%-64s %6s calls  %8.2f us  %5s %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
rm -rf $out
