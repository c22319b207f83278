# GPU session r04/24: seed sort -- segments of 17..128 entries by rows of 16 lanes (four per wave) instead of one lane / one wave each
export TMPDIR=/tmp
O=gpurun_out/r04y; mkdir -p $O
(timeout 240 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -6) > $O/pytest_sort.log; cat $O/pytest_sort.log
grep -q " passed" $O/pytest_sort.log && ! grep -q "failed\|error" $O/pytest_sort.log || { echo "seed sort tests not green: stop"; exit 1; }
(timeout 120 python tools/experiments/seed_sort_prof.py 2>&1 | tail -4) > $O/prof.log; cat $O/prof.log
for v in r04z main gu3; do
  if [ $v = main ]; then unset PLP_FRONT_LIB; else export PLP_FRONT_LIB=build_exp/$v.so; fi
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$v.err | tail -1) > $O/bench_$v.json
  python -c "import json; j=json.load(open('$O/bench_$v.json')); s=j['roofline']['stage_ms_per_batch']; print('$v', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'seed order alone', s.get('lsd_order'), 'line_extract latency', j.get('latency_ms_median_mean'))" || tail -2 $O/bench_$v.err
done
