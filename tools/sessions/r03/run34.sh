# GPU session 34: the tree as committed -- smoke(), the matcher / step tests, the default bench line
export TMPDIR=/tmp
O=gpurun_out/r03x14; mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
(timeout 200 python -m pytest tests/test_gpu_match.py tests/test_gpu_bench_step.py tests/test_gpu_replay_sharded.py -q -p no:cacheprovider -x 2>&1 | tail -1) > $O/pytest.log; cat $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['verified_frames'], j['pcie_inclusive_value'], j['latency_ms_median_mean'])"
