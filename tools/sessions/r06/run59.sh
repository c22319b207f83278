# GPU session r06/59: sub-blocks of the line path (PLP_BENCH_LINE_SPLIT) x line chains of consecutive steps in flight (PLP_BENCH_LINE_DEPTH), re-measured on the final tree:
# round 4 found no effect at register packing 0.75; the kernels hold a quarter less now
export TMPDIR=/tmp
B() { timeout 150 env "$@" python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], j['ms_per_step'], 'verified', j.get('verified_frames'))"; }
for pass in 1 2; do
B PLP_BENCH_LINE_SPLIT=2 PLP_BENCH_LINE_DEPTH=1
B PLP_BENCH_LINE_SPLIT=1 PLP_BENCH_LINE_DEPTH=1
B PLP_BENCH_LINE_SPLIT=4 PLP_BENCH_LINE_DEPTH=1
B PLP_BENCH_LINE_SPLIT=8 PLP_BENCH_LINE_DEPTH=1
B PLP_BENCH_LINE_SPLIT=1 PLP_BENCH_LINE_DEPTH=2
B PLP_BENCH_LINE_SPLIT=2 PLP_BENCH_LINE_DEPTH=2
B PLP_BENCH_LINE_SPLIT=4 PLP_BENCH_LINE_DEPTH=2
B PLP_BENCH_LINE_SPLIT=2 PLP_BENCH_LINE_DEPTH=3
done
