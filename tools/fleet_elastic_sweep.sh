#!/bin/bash
# on the GPU box: so_fleet_run's elastic ticks against rigid ones, staggered keyframes, the BA group, 4..32 agents per GPU, several
# driving threads -> gpurun_out/fleet_elastic.jsonl (profiles/r6_fleet_elastic.jsonl is one such run, NOTES.md G.8 reads it)
cd /root/repo
mkdir -p gpurun_out
out=gpurun_out/fleet_elastic.jsonl
: > $out
for A in 8 4 12 16; do
  for envs in "SWARMORB_FLEET_RIGID=1" "X=1" "SWARMORB_FLEET_STAGGER=1" "SWARMORB_FLEET_STAGGER=1 SWARMORB_FLEET_NO_BA_GROUP=1" "SWARMORB_FLEET_NO_BA_GROUP=1"; do
    bash tools/fleet_ab.sh "$envs" --agents-per-gpu $A --lockstep >> $out 2>&1
  done
done
for cfg in "8 2" "16 2" "16 4" "24 3" "32 4"; do
  set -- $cfg
  bash tools/fleet_ab.sh "X=1" --agents-per-gpu $1 --lockstep --fleet-threads $2 >> $out 2>&1
done
cat $out
