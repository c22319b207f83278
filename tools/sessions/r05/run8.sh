# GPU session r05/8: waves per workgroup x LDS window of the seed sort again, now that the kernel holds 85 registers (round 4's sweep was taken at 128 and more)
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
B() {
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'order', round(s['lsd_order'],2), 'grow', round(s['lsd_grow'],2))" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
B base1
for v in s8x4096 s8x8192 s4x8192 s4x2048 s2x4096 s2x2048; do PLP_FRONT_LIB=build_exp/$v.so B $v; done
B base2
