# GPU session r04/13: region growing confined to a subset of the CUs (hipExtStreamCreateWithCUMask), VERDICT r03 item 3b
export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
for n in 0 128 192 224 160; do
  export PLP_GROW_CUS=$n
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$n.err | tail -1) > $O/bench_$n.json
  python -c "import json; j=json.load(open('$O/bench_$n.json')); print('grow CUs $n:', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'grow alone', j['roofline']['stage_ms_per_batch']['lsd_grow'])" || tail -2 $O/bench_$n.err
done
