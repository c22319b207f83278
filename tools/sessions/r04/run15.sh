# GPU session r04/15: k_blur7 with its level table in the kernel arguments (no dependent scalar loads before the first pixel load)
export TMPDIR=/tmp
O=gpurun_out/r04o; mkdir -p $O
for n in 0 1 0 1; do
  export PLP_BLUR7_TAB=$n
  (timeout 120 python -m pytest tests/test_gpu_orb.py -q -x -p no:cacheprovider 2>&1 | tail -1) > $O/pytest_$n.log; echo "tab $n: $(cat $O/pytest_$n.log)"
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$n.err | tail -1) > $O/bench_$n.json
  python -c "import json; j=json.load(open('$O/bench_$n.json')); s=j['roofline']['stage_ms_per_batch']; print('tab $n:', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'blur7 alone', s['blur7'])" || tail -2 $O/bench_$n.err
done
