# GPU session r04/18: current tree: full GPU suite, seed-sort cost (two-wide lane scans), bench
export TMPDIR=/tmp
O=gpurun_out/r04s; mkdir -p $O
(timeout 60 python tools/experiments/dbg_seed3.py 2>&1 | tail -1) > $O/dbg.log; cat $O/dbg.log
grep -q "20000 1 failures of 100: 0" $O/dbg.log || { echo "debug cases failed or hung: stopping"; exit 1; }
(timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench.err | tail -1) > $O/bench.json
python -c "import json; j=json.load(open('$O/bench.json')); s=j['roofline']['stage_ms_per_batch']; print(j['value'], j['ms_per_step'], 'stable:', j['other_seed_order'], 'verified', j['verified_frames']); print(s)" || tail -2 $O/bench.err
