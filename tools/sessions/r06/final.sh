# GPU session r06/final: the round's closing checks and measurements on the final tree -- whole GPU suite, smoke + default bench as the driver runs them, round profile (tag r06z), fuzz sweep
export TMPDIR=/tmp
O=gpurun_out/r06z; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default_time.log; tail -3 $O/bench_default_time.log; cut -c1-200 $O/bench_default.json
bash tools/run_prof.sh r06z > $O/run_prof.log 2>&1; tail -2 $O/run_prof.log | cut -c1-300
(timeout 200 python tools/fuzz_gpu.py --seconds 120 --seed 101 2>&1 | tail -6) > $O/fuzz.log; cat $O/fuzz.log
