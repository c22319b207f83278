# GPU session r04/12: the steps of configs[2], [3], [4] pinned to the oracle (tests + bench tool with --verify); launch-bound variant of the seed sort
export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_config_steps.py -q -x -p no:cacheprovider 2>&1 | tail -12) > $O/pytest.log; cat $O/pytest.log
(timeout 600 python tools/bench_configs.py --batch 1024 --steps 3 --verify 8 2> $O/configs.err) > $O/configs.jsonl; cat $O/configs.jsonl | cut -c1-400; tail -3 $O/configs.err
for v in minw4 main; do
  if [ $v = main ]; then unset PLP_FRONT_LIB; else export PLP_FRONT_LIB=build_exp/$v.so; fi
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$v.err | tail -1) > $O/bench_$v.json
  python -c "import json; j=json.load(open('$O/bench_$v.json')); print('$v', j['value'], j['ms_per_step'], j['other_seed_order'], j['verified_frames'], j['roofline']['stage_ms_per_batch']['lsd_order'])"
done
