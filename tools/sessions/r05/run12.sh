# GPU session r05/12: frames per step x line sub-blocks with the 113-register grower (does a third or fourth grower per SIMD pay when the sub-blocks stay at 1024 frames?)
export TMPDIR=/tmp
O=gpurun_out/r05l; mkdir -p $O
B() {
  (timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --verify 8 $2 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'])" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
PLP_BENCH_LINE_SPLIT=3 B b3072s3 "--batch 3072"
PLP_BENCH_LINE_SPLIT=4 B b4096s4 "--batch 4096"
PLP_BENCH_LINE_SPLIT=2 B b4096s2 "--batch 4096"
PLP_BENCH_LINE_SPLIT=2 PLP_BENCH_NBUF=3 B b2048n3 "--batch 2048"
PLP_BENCH_LINE_SPLIT=2 B b1024s2 "--batch 1024"
