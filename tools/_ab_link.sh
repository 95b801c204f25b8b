cd /root/repo
for v in 0 1 0 1; do
SWARMORB_TRACK_LINK=$v python bench.py --steps 300 --warmup 20 --no-configs --no-cpu-baseline 2>/dev/null | tail -1 > /dev/null
python - <<P
import json
f=json.load(open('profiles/last_bench_full.json'))
c=f['config']; L=c['closed_loop']['local_mapping_ms_per_keyframe']; H=c['host_ms_per_frame']
print('LINK=$v', round(f['value']), 'frame', {k:round(v,3) for k,v in c['frame_ms_percentiles'].items()}, 'job', round(L['whole_job'],3), 'waited', round(c['closed_loop']['whole_run']['tracking_thread_waited_ms'],1), 'collect', round(H['collect_and_submit'],4), 'm2', round(H['match_m2'],4), 'm1', round(H['match_m1'],4), 'pose3', round(H['pose_optimization_x3'],4), 'lat', c['latency_ms_image_to_pose'])
P
done
