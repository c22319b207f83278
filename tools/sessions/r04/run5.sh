# GPU session r04/5: seed sort after the equal-keys shortcut and the unrolled workgroup passes: parity, phase cycles, cost
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
(timeout 90 python tools/experiments/dbg_seed3.py 2>&1 | tail -6) > $O/dbg.log; cat $O/dbg.log
grep -q "20000 1 failures of 100: 0" $O/dbg.log || { echo "debug cases failed or hung: stopping"; exit 1; }
(timeout 120 python tools/experiments/seed_sort_prof.py 2>&1 | tail -3) > $O/prof.log; cat $O/prof.log
(timeout 300 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python tools/seed_order_cost.py --batch 2048 2>&1 | tail -1) > $O/cost.json; cat $O/cost.json
