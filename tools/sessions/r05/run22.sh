# GPU session r05/22: the driver's sequence on a fresh box: smoke(), then `python bench.py` with no flags (wall time of each)
export TMPDIR=/tmp
O=gpurun_out/r05v; mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -5 $O/smoke.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.log; tail -3 $O/bench_time.log; cut -c1-300 $O/bench_default.json
