# GPU session r05/14: the acceptance block two scalar instructions shorter (s_bitset1 for the accepted set, the count from the set afterwards) -- parity, bench line
export TMPDIR=/tmp
O=gpurun_out/r05n; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_line.py tests/test_gpu_golden_ref.py tests/test_gpu_bench_step.py tests/test_gpu_facade.py -q -x -p no:cacheprovider 2>&1 | tail -1) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 $2 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'order', round(s['lsd_order'],2), 'grow', round(s['lsd_grow'],2))" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
B new1 ""
PLP_FRONT_LIB=build_exp/nolone.so B prev1 ""
B new2 ""
PLP_FRONT_LIB=build_exp/nolone.so B prev2 ""
(timeout 70 python tools/fuzz_gpu.py --only lines --seconds 45 --seed 85 2>&1 | tail -3) > $O/fuzz.log; cat $O/fuzz.log
