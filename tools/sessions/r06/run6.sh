# GPU session r06/6: seed sort -- a wave with no task to take finishes a small subtree instead of sleeping (ticket of the list); LDS capacities follow the window (35 -> 24.5 KB); tests, phase clocks, bench
export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/ss.log; cat $O/ss.log
timeout 300 python tools/experiments/seed_sort_prof.py > $O/prof_cand.log 2>&1; tail -2 $O/prof_cand.log
(timeout 900 python -m pytest tests/test_gpu_line.py tests/test_gpu_seed_sort_soak.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/line.log; cat $O/line.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06f/bench.json"))
print(j["value"], j["ms_per_step"], j["verified_frames"], j["roofline"]["stage_ms_per_batch"])
PY
