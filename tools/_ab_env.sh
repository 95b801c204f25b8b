#!/bin/bash
# A/B an environment switch on the closed-loop bench:  bash tools/_ab_env.sh VAR v1 v2 ...   (prints value, frames/s, LM job ms, BA ms, gather ms | tracking thread: match ms, pose x3 ms, live latency p50, pose kernel ms)
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open('profiles/last_bench_full.json')); L=f['config']['closed_loop']['local_mapping_ms_per_keyframe']
H=f['config']['host_ms_per_frame']
print('$var=$v', round(d['value']), round(L['whole_job'],3), round(L['so_bundle_adjust'],3), round(L['window_gather'],3), '| track: match', round(H['match'],4), 'pose x3', round(H['pose_optimization_x3'],4), 'live p50', round(d['config']['latency_ms_image_to_pose']['p50'],4), 'pose kernel', round(d['roofline']['avg_launch_ms'],4))"
done
