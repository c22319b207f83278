# GPU session r04/20: line sub-blocks per step (2 shipped) with the exact seed sort in the chain
export TMPDIR=/tmp
O=gpurun_out/r04u; mkdir -p $O
for n in 2 3 4 1; do
  export PLP_BENCH_LINE_SPLIT=$n
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$n.err | tail -1) > $O/bench_$n.json
  python -c "import json; j=json.load(open('$O/bench_$n.json')); print('line split $n:', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'])" || tail -2 $O/bench_$n.err
done
