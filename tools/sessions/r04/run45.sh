# GPU session r04/45: full GPU suite, a lines-only fuzz sweep and the bench line on the shipped seed sort with the store drain before its global partitions' barriers
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04v2; mkdir -p $O
(timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python tools/fuzz_gpu.py --only lines --seconds 60 --seed 81 2>&1 | grep "lines:" ) > $O/fuzz_lines.log; cat $O/fuzz_lines.log
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $O/bench.err | tail -1) > $O/bench.json; python -c "import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['other_seed_order'], j['verified_frames'], j['pcie_inclusive_value'], j['latency_ms_median_mean']['line_extract'], j['roofline']['stage_ms_per_batch']['lsd_order'])" || tail -3 $O/bench.err
