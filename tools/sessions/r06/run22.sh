# GPU session r06/22: no FLAT memory instruction left in the library (k_quadtree, k_lsd_grow_mw, k_match_topk_cells specialised per address space) + the seed sort with its masks in LDS through DS instructions: the whole GPU suite, bench
export TMPDIR=/tmp
O=gpurun_out/r06v; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06v/bench.json"))
print(j["value"], j["ms_per_step"], j["verified_frames"], j["roofline"]["stage_ms_per_batch"], j.get("latency_ms_median_mean"))
PY
