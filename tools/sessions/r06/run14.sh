# GPU session r06/14: seed sort -- chunk masks of a global partition in LDS when they fit (all but the first of a frame), popcounts left by the classifying pass; tests, clocks alone / chip full, bench
export TMPDIR=/tmp
O=gpurun_out/r06n; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/ss.log; cat $O/ss.log
for c in 1 2048; do PLP_SEED_SORT_DBG_COPIES=$c timeout 300 python tools/experiments/seed_sort_prof.py > $O/prof_$c.log 2>&1; echo "copies $c"; tail -1 $O/prof_$c.log; done
(timeout 900 python -m pytest tests/test_gpu_line.py tests/test_gpu_seed_sort_soak.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/line.log; cat $O/line.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06n/bench.json"))
print(j["value"], j["ms_per_step"], j["verified_frames"], j["roofline"]["stage_ms_per_batch"])
PY
