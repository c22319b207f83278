# GPU session r04/9: seed sort with 8 waves per workgroup (two workgroups per CU) against 16 waves
export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
for v in w8t4k w8t6k w4t4k w4t8k; do
  if [ $v = main ]; then unset PLP_FRONT_LIB; else export PLP_FRONT_LIB=build_exp/$v.so; fi
  (timeout 60 python tools/experiments/dbg_seed3.py 2>&1 | tail -1) > $O/dbg_$v.log; cat $O/dbg_$v.log
  grep -q "20000 1 failures of 100: 0" $O/dbg_$v.log || { echo "$v: debug cases failed or hung: skipping"; continue; }
  (timeout 200 python tools/seed_order_cost.py --batch 2048 2>&1 | tail -1) > $O/cost_$v.json
  python -c "import json; j=json.load(open('$O/cost_$v.json')); print('$v', 'lsd_order ms', j['libstdcxx']['stage_ms']['lsd_order'], 'batch', j['libstdcxx']['batch_ms_unprofiled'], 'single', j['libstdcxx']['single_frame_ms_median'])"
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$v.err | tail -1) > $O/bench_$v.json
  python -c "import json; j=json.load(open('$O/bench_$v.json')); print('$v', j['value'], j['ms_per_step'], j['other_seed_order'], j['verified_frames'])"
done
