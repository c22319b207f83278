# GPU session 32: single-frame matcher latency against the queries per workgroup of k_match_topk_cells (the step numbers of these runs are not the point)
export TMPDIR=/tmp
O=gpurun_out/r03x12; mkdir -p $O
P() { timeout 200 env $1 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); l=j['latency_ms_median_mean']; print('$1 |', j['value'], '| last', l['match_current_and_last_frames'], 'lm', l['match_frame_and_landmarks'], 'orb', l['orb_extract'], 'line', l['line_extract'])"; }
{
P PLP_NONE=1
P PLP_MATCH_QPB=32
P PLP_MATCH_QPB=64
P PLP_MATCH_QPB=128
P PLP_MATCH_QPB=256
} > $O/lat.log 2>&1
cat $O/lat.log
(timeout 120 python -m pytest tests/test_gpu_match.py -q -p no:cacheprovider -x 2>&1 | tail -1) > $O/pytest.log; cat $O/pytest.log
