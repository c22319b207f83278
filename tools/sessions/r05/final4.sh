# GPU session r05/final4: what is kept of the latency series -- the hand-over at workgroup scope, the main wave's clocks only when profiling, the watchdog flag read where
# a wave waits (only k_lsd_grow_mw differs from final2's binary: ISA of every other kernel compared) -- against the committed tree before it (build_exp/r05z_base.so) on
# one box; full GPU suite; the bench line with its latency pass; a line sweep
export TMPDIR=/tmp
O=gpurun_out/r05x; mkdir -p $O
for LIB in build_exp/r05z_base.so "" build_exp/r05z_base.so ""; do
  echo "== lib=${LIB:-shipped}" >> $O/calls.log
  (PLP_FRONT_LIB=$LIB timeout 60 python tools/experiments/latency_calls.py 96 2>&1 | grep -v amdgpu.ids | tail -2) >> $O/calls.log
done
cat $O/calls.log
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python bench.py --verify 64 > $O/bench.json 2> $O/bench.err); python - <<'PY'
import json
d = json.load(open("gpurun_out/r05x/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "verified", d.get("verified_frames"), d.get("verified_halo_rows"), "latency", d.get("latency_ms_median_mean"))
PY
(timeout 100 python tools/fuzz_gpu.py --only lines --seconds 35 --seed 117 2>&1 | tail -3) > $O/fuzz.log; cat $O/fuzz.log
