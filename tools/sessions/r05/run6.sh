# GPU session r05/6: (1) region_grow's settle block (the whole round accepted together, verified lane by lane) -- parity and the bench line;
# (2) the 2-wave experiment build of the seed sort in three variants that separate hypotheses about its fault:
#     x1 SGPR spills to memory instead of VGPR lanes | x2 the prefix arrays of a global partition in LDS of their own (no overlay) | x3 the scan pass staged through LDS (few spills)
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_line.py tests/test_gpu_golden_ref.py tests/test_gpu_bench_step.py tests/test_gpu_config_steps.py tests/test_gpu_facade.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'order', round(s['lsd_order'],2), 'grow', round(s['lsd_grow'],2))" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
B settle1
PLP_FRONT_LIB=build_exp/head.so B head1
B settle2
(timeout 70 python tools/fuzz_gpu.py --only lines --seconds 45 --seed 83 2>&1 | tail -3) > $O/fuzz.log; cat $O/fuzz.log
for v in x1 x2 x3; do for r in 1 2; do PLP_FRONT_LIB=build_exp/$v.so B ${v}_$r; done; done
