# GPU session r06/33: the step's arrangement again with this round's sort -- line sub-blocks 1 .. 4 x frames per step 2048 / 3072 / 4096, line stream priority
export TMPDIR=/tmp
O=gpurun_out/r06arr; mkdir -p $O
R() { env "$@" timeout 200 python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 12 --warmup 3 --batch $BATCH 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', 'batch $BATCH', j['value'], j['ms_per_step'], 'verified', j['verified_frames'])"; }
for BATCH in 2048; do for s in 2 1 3 4; do R PLP_BENCH_LINE_SPLIT=$s; done; R PLP_BENCH_LINE_SPLIT=2 PLP_BENCH_LINE_PRIO=-1; R PLP_BENCH_LINE_SPLIT=2 PLP_BENCH_NBUF=3; done 2>&1 | tee $O/arr.log
for BATCH in 3072 4096; do for s in 2 3 4; do R PLP_BENCH_LINE_SPLIT=$s; done; done 2>&1 | tee -a $O/arr.log
