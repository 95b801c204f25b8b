#!/bin/bash
# A/B an environment switch on the closed-loop bench:  bash tools/_ab_env.sh VAR v1 v2 ...   (prints value, frames/s, LM job ms, BA ms)
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); L=d['config']['closed_loop']['local_mapping_ms_per_keyframe']
print('$var=$v', round(d['value']), round(L['whole_job'],3), round(L['so_bundle_adjust'],3), round(L['window_gather'],3))"
done
