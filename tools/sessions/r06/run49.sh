# GPU session r06/49: the two line streams out of phase (PLP_BENCH_LINE_PHASE_MS: stream 2 starts that much later, once) -- one sub-block sorts while the other grows
export TMPDIR=/tmp
O=gpurun_out/r06arr; mkdir -p $O
R() { env "$@" timeout 200 python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', j['value'], j['ms_per_step'], 'verified', j['verified_frames'])"; }
for p in 0 4 8 11 0 6 10; do R PLP_BENCH_LINE_PHASE_MS=$p; done 2>&1 | tee $O/phase.log
R PLP_BENCH_LINE_SPLIT=3 PLP_BENCH_LINE_PHASE_MS=6 | tee -a $O/phase.log
R PLP_BENCH_LINE_SPLIT=4 PLP_BENCH_LINE_PHASE_MS=5 | tee -a $O/phase.log
