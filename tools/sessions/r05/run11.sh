# GPU session r05/11: the lone-seed path with the alignment decided in f32 degrees (f64 only inside a 0.01-degree band), with and without the check of the USED bits at the seed's turn, against the kernel without it
export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_line.py tests/test_gpu_bench_step.py -q -x -p no:cacheprovider 2>&1 | tail -1) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 $2 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'order', round(s['lsd_order'],2), 'grow', round(s['lsd_grow'],2))" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
B lone_full ""
PLP_FRONT_LIB=build_exp/lone_static.so B lone_static ""
PLP_FRONT_LIB=build_exp/nolone.so B nolone ""
B lone_full2 ""
