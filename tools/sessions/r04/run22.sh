# GPU session r04/22: the two line streams out of phase (one sub-block's front beside the other's region growing)
export TMPDIR=/tmp
O=gpurun_out/r04w; mkdir -p $O
for ms in 0 4 8 12 16; do
  export PLP_BENCH_LINE_PHASE_MS=$ms
  (timeout 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$ms.err | tail -1) > $O/bench_$ms.json
  python -c "import json; j=json.load(open('$O/bench_$ms.json')); print('phase $ms ms:', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'])" || tail -2 $O/bench_$ms.err
done
