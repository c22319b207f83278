# GPU session r06/23: one packed halo exchange per step (replay.halo_exchanger, ring / allgather), counted matcher candidates, bench line: step tests, two-rank bench in both modes, bench
export TMPDIR=/tmp
O=gpurun_out/r06w; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_bench_step.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_replay_sharded.py tests/test_gpu_seed_sort_soak.py tests/test_gpu_config_steps.py -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r06w/bench.json"))
print(j["value"], j["ms_per_step"], j["verified_frames"], j["config"]["sharding"], j["roofline"].get("match_candidates_per_query_counted"), j["roofline"]["stage_GBps"].get("match_4x"))
PY
