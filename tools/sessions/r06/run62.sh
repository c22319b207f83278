# GPU session r06/62: the bench step on 2048 DISTINCT frames (the batch was always 64 distinct frames repeated 32 times) beside the 64-frame workload, same box
export TMPDIR=/tmp
O=gpurun_out/r06distinct; mkdir -p $O
for d in 64 2048 64 2048; do timeout 300 python bench.py --distinct $d --steps 20 --warmup 4 --no-cpu-baseline --no-extras --verify 64 2>/dev/null | tail -1 > $O/d$d.json; python -c "
import json;j=json.load(open('$O/d$d.json'));c=j['config'];s=j['roofline']['stage_ms_per_batch']
print('distinct',c['distinct_frames'],j['value'],j['ms_per_step'],'verified',j['verified_frames'],j['verified_halo_rows'],'kp',c['keypoints_mean'],'lines',c['lines_mean'],'matches',c['matches_mean'],'grow',s['lsd_grow'],'sort',s['lsd_order'],'stable',j['other_seed_order']['value'])"; done
