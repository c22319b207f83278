# GPU session r04/11: two configurations of the seed sort in one library (4 waves / 4096 entries for batches, 16 waves / 24576 for <= 256 frames)
export TMPDIR=/tmp
O=gpurun_out/r04k; mkdir -p $O
(timeout 60 python tools/experiments/dbg_seed3.py 2>&1 | tail -1) > $O/dbg.log; cat $O/dbg.log
grep -q "20000 1 failures of 100: 0" $O/dbg.log || { echo "debug cases failed or hung: stopping"; exit 1; }
(timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python tools/seed_order_cost.py --batch 2048 2>&1 | tail -1) > $O/cost.json
python -c "import json; j=json.load(open('$O/cost.json')); print('lsd_order ms', j['libstdcxx']['stage_ms']['lsd_order'], 'batch', j['libstdcxx']['batch_ms_unprofiled'], 'single', j['libstdcxx']['single_frame_ms_median'], 'stable single', j['stable']['single_frame_ms_median'])"
(timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> $O/bench.err | tail -1) > $O/bench.json
python -c "import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['other_seed_order'], j['verified_frames'], j.get('pcie_inclusive_value'), j.get('latency_ms_median_mean'))"
