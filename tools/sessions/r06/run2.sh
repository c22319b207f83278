# GPU session r06/2: seed sort -- subtrees of at most 256 entries finished by ONE wave with scalar chunk masks (wave_sub_sort) instead of wave tasks through the queue (65..256) and one lane per segment (17..64)
export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/ss.log; cat $O/ss.log
(timeout 900 python -m pytest tests/test_gpu_line.py tests/test_gpu_seed_sort_soak.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/line.log; cat $O/line.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06b/bench.json"))
print(j["value"], j["ms_per_step"], j["verified_frames"], j["roofline"]["stage_ms_per_batch"])
PY
