#!/bin/bash
# Profiler evidence for the whole-map solver (BASELINE configs[4]):  bash tools/profile_gba.sh <tag>
# kernel trace + stats in one pass, the MFMA counters in their own passes (never combined with another trace domain).
set -u
tag=${1:-rX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 tools/gba_bench.py > $out/${tag}_gba.json 2> $out/${tag}_gba.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_gba -- python3 tools/gba_bench.py > $out/${tag}_gba_prof.log 2>&1
f=$(find $out/prof_${tag}_gba -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $out/${tag}_gba_kernel_stats.csv
rm -rf $out/prof_${tag}_gba
for c in SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_gba_$c -- python3 tools/gba_bench.py GBA-2 GBA-2r > $out/${tag}_gba_pmc_$c.log 2>&1
  f=$(find $out/pmc_${tag}_gba_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" $out/${tag}_gba_pmc_${c}_summary.csv
  rm -rf $out/pmc_${tag}_gba_$c
done
ls -la $out | tail -20
