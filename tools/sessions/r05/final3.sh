# GPU session r05/final3: the tree after the latency-path series (only k_lsd_grow_mw differs from final2's binary: ISA of every other kernel compared) -- full GPU suite,
# the bench line with its latency pass, a line sweep
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest3.log; cat $O/pytest3.log
(timeout 300 python bench.py --verify 64 > $O/bench.json 2> $O/bench.err); cut -c1-200 $O/bench.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05y/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "verified", d.get("verified_frames"), d.get("verified_halo_rows"), "latency", d.get("latency_ms_median_mean"))
PY
(timeout 200 python tools/fuzz_gpu.py --only lines --seconds 60 --seed 116 2>&1 | tail -3) > $O/fuzz3.log; cat $O/fuzz3.log
