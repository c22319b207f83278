# GPU session r04/7: seed sort with a small LDS window (8192 entries, 56 KB) against 16384 / 24576: parity, cost alone, the step
export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
(timeout 90 python tools/experiments/dbg_seed3.py 2>&1 | tail -6) > $O/dbg.log; cat $O/dbg.log
grep -q "20000 1 failures of 100: 0" $O/dbg.log || { echo "debug cases failed or hung: stopping"; exit 1; }
(timeout 300 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
for v in main t16k t24k; do
  if [ $v = main ]; then unset PLP_FRONT_LIB; else export PLP_FRONT_LIB=build_exp/$v.so; fi
  (timeout 200 python tools/seed_order_cost.py --batch 2048 2>&1 | tail -1) > $O/cost_$v.json
  python -c "import json; j=json.load(open('$O/cost_$v.json')); print('$v', 'lsd_order ms', j['libstdcxx']['stage_ms']['lsd_order'], 'batch', j['libstdcxx']['batch_ms_unprofiled'], 'single', j['libstdcxx']['single_frame_ms_median'])"
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$v.err | tail -1) > $O/bench_$v.json
  python -c "import json; j=json.load(open('$O/bench_$v.json')); print('$v', j['value'], j['ms_per_step'], j['other_seed_order'], j['verified_frames'])"
done
