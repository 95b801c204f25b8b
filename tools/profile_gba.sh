#!/bin/bash
# Profiler evidence for the whole-map solver (BASELINE configs[4]):  bash tools/profile_gba.sh <tag>
# One case per profiler pass (the dataflow kernels have the same name for every map): kernel trace + stats in one pass,
# the MFMA / busy counters in their own passes (never combined with another trace domain).  Summaries land in
# gpurun_out/<tag>_gba_<case>_*; tools/make_gba_pmc.py turns them into profiles/<tag>_gba_pmc_mfma.json.
set -u
tag=${1:-rX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 tools/gba_bench.py > $out/${tag}_gba.json 2> $out/${tag}_gba.err
for case in GBA-1 GBA-2 GBA-1r GBA-2r; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_$case -- python3 tools/gba_bench.py $case > $out/${tag}_gba_${case}_prof.log 2>&1
  f=$(find $out/prof_${tag}_$case -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $out/${tag}_gba_${case}_kernel_stats.csv
  rm -rf $out/prof_${tag}_$case
  for c in SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_${case}_$c -- python3 tools/gba_bench.py $case > $out/${tag}_gba_${case}_pmc_$c.log 2>&1
    f=$(find $out/pmc_${tag}_${case}_$c -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" $out/${tag}_gba_${case}_pmc_${c}.csv
    rm -rf $out/pmc_${tag}_${case}_$c
  done
done
ls $out | grep ${tag}_gba | head -60
