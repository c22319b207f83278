# GPU session r04/26: the line chains of two consecutive steps in flight together (two sets of line contexts and streams, three feature sets)
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
run() { # name, env...
  name=$1; shift
  (env "$@" timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$name.err | tail -1) > $O/bench_$name.json
  python -c "import json; j=json.load(open('$O/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'])" || tail -3 $O/bench_$name.err
}
run base X=1
run depth2 PLP_BENCH_LINE_DEPTH=2 PLP_BENCH_NBUF=3
run depth2_q8 PLP_BENCH_LINE_DEPTH=2 PLP_BENCH_NBUF=3 GPU_MAX_HW_QUEUES=8
run q8 GPU_MAX_HW_QUEUES=8
run depth2_split1 PLP_BENCH_LINE_DEPTH=2 PLP_BENCH_NBUF=3 PLP_BENCH_LINE_SPLIT=1
