# GPU session r05/3: wave-uniform indices declared scalar (k_lsd_grow 155 -> 113 VGPRs, seed sort 128 -> 110, matchers, LBD, ...), rectangle refits from the list,
# 32-bit gather offsets, lazy f64 conversion, unconditional USED read; the seed sort's partner check compiled in.  Full GPU suite + same-box A/B against HEAD.
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], {k: round(v, 2) for k, v in s.items()})" || tail -2 $O/bench_$1.err
}
for r in 1 2; do
  B new$r
  PLP_FRONT_LIB=build_exp/head.so B head$r
done
(timeout 90 python tools/fuzz_gpu.py --seconds 60 --seed 82 2>&1 | tail -4) > $O/fuzz.log; cat $O/fuzz.log
