# GPU session r04/27: the step by parts (which streams cost what beside which)
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
run() { # name, env...
  name=$1; shift
  (env "$@" timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --verify 0 2> $O/parts_$name.err | tail -1) > $O/parts_$name.json
  python -c "import json; j=json.load(open('$O/parts_$name.json')); print('$name', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'])" || tail -3 $O/parts_$name.err
}
run all X=1
run lines PLP_BENCH_PARTS=lines
run orb PLP_BENCH_PARTS=orb
run orb_lines PLP_BENCH_PARTS=orb,lines
run lines_match PLP_BENCH_PARTS=lines,match
run orb_match PLP_BENCH_PARTS=orb,match
run lines_split1 PLP_BENCH_PARTS=lines PLP_BENCH_LINE_SPLIT=1
run lines_split4 PLP_BENCH_PARTS=lines PLP_BENCH_LINE_SPLIT=4
