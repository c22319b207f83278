# GPU session r04/6: exact seed order as the default: full GPU suite, bench (headline in the reference's order, the stable order beside it)
export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6) > $O/pytest.log; cat $O/pytest.log
(timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> $O/bench.err | tail -1) > $O/bench.json; python - <<'PY'
import json
j = json.load(open("gpurun_out/r04f/bench.json"))
print({k: j[k] for k in ("value", "ms_per_step", "seed_order", "other_seed_order", "verified_frames")})
print(j["roofline"]["stage_ms_per_batch"])
print(j.get("latency_ms_median_mean"), j.get("pcie_inclusive_value"))
PY
tail -3 $O/bench.err
