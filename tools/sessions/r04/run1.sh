# GPU session r04/1: the exact seed order (D1): kernel vs libstdc++, the line front-end in both orders, its cost
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
(timeout 90 python tools/experiments/dbg_seed3.py 2>&1 | tail -6) > $O/dbg4.log; cat $O/dbg4.log
grep -q "20000 1 failures of 100: 0" $O/dbg4.log || { echo "debug cases failed or hung: stopping"; exit 1; }
(timeout 500 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_line.py -q -p no:cacheprovider 2>&1 | tail -15) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python tools/seed_order_cost.py --batch 2048 2>&1 | tail -3) > $O/cost.json; cat $O/cost.json
