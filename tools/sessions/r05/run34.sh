# GPU session r05/34: the latency path -- main takes a small result with one LDS round trip less (entry read travels with the seed's committed bit), refreshes its seed
# set only after regions of 8+ pixels; wall time against build_exp/agentscope.so (the state before this series: 4.712 / 4.240 ms on run30's box where the series so far gave
# 4.539 / 4.100); parity incl. the policy / park switches
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for LIB in build_exp/agentscope.so ""; do
  echo "== lib=${LIB:-shipped}" >> $O/calls3.log
  (PLP_FRONT_LIB=$LIB timeout 120 python tools/experiments/latency_calls.py 128 2>&1 | grep -v amdgpu.ids | tail -2) >> $O/calls3.log
done
cat $O/calls3.log
(timeout 400 python -m pytest tests/test_gpu_line.py tests/test_gpu_line_mw_options.py -x -q 2>&1 | tail -3) > $O/pytest_main.log; cat $O/pytest_main.log
(timeout 200 python tools/fuzz_gpu.py --only lines --seconds 40 --seed 115 2>&1 | tail -3) > $O/fuzz_main.log; cat $O/fuzz_main.log
