# GPU session r05/10: k_lsd_grow's lone-seed path (a seed none of whose neighbours can pass the first round is a region of one: no round is run for it) -- parity, bench line
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_line.py tests/test_gpu_golden_ref.py tests/test_gpu_bench_step.py tests/test_gpu_config_steps.py tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 $2 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'order', round(s['lsd_order'],2), 'grow', round(s['lsd_grow'],2))" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
B lone1 ""
B lone2 ""
(timeout 90 python tools/fuzz_gpu.py --only lines --seconds 60 --seed 84 2>&1 | tail -3) > $O/fuzz.log; cat $O/fuzz.log
