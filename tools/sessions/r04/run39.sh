# GPU session r04/39: the full GPU suite, smoke() and the default bench line on the tree after the late experiments (kernels unchanged; knobs added)
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04w; mkdir -p $O
(timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.log; cat $O/smoke.log
(timeout 400 python bench.py 2> $O/bench.err | tail -1) > $O/bench.json; python -c "import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['other_seed_order'], j['verified_frames'], j['pcie_inclusive_value'], j['latency_ms_median_mean']['line_extract'], j['cpu_baseline']['value'])" || tail -3 $O/bench.err
