# GPU session r05/17: FAST staging with (row, dword) by additions instead of a division by the runtime row length per trip -- parity, bench line
export TMPDIR=/tmp
O=gpurun_out/r05q; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_orb.py tests/test_gpu_line.py tests/test_gpu_golden_ref.py tests/test_gpu_bench_step.py tests/test_gpu_config_steps.py tests/test_gpu_rectify.py tests/test_gpu_facade.py -q -x -p no:cacheprovider 2>&1 | tail -1) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 $2 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'fast', round(s['fast_cells'],2), 'blur7', round(s['blur7'],2))" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
B new1 ""
PLP_FRONT_LIB=build_exp/prev.so B prev1 ""
B new2 ""
(timeout 100 python tools/fuzz_gpu.py --seconds 70 --seed 87 2>&1 | tail -4) > $O/fuzz.log; cat $O/fuzz.log
