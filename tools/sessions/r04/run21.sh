# GPU session r04/21: k_lsd_grow with the record gather issued before the USED test (experiment)
export TMPDIR=/tmp
O=gpurun_out/r04v; mkdir -p $O
for v in main gfirst; do
  if [ $v = main ]; then unset PLP_FRONT_LIB; else export PLP_FRONT_LIB=build_exp/$v.so; fi
  (timeout 200 python -m pytest tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -1) > $O/pytest_$v.log; echo "$v: $(cat $O/pytest_$v.log)"
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$v.err | tail -1) > $O/bench_$v.json
  python -c "import json; j=json.load(open('$O/bench_$v.json')); s=j['roofline']['stage_ms_per_batch']; print('$v', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'grow alone', s['lsd_grow'])" || tail -2 $O/bench_$v.err
done
