# GPU session r06/1: the new concurrent single-frame test, the line / seed-sort tests on the tree with ADVICE r05's changes, the bench line of this box as the round's baseline
export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_concurrent_single_frame.py -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/concurrent.log; cat $O/concurrent.log
(timeout 900 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_line.py tests/test_gpu_seed_sort_soak.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/line.log; cat $O/line.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06a/bench.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["stage_ms_per_batch"], j.get("latency_ms_median_mean"))
PY
