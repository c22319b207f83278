# GPU session r05/30: the latency path -- the wave clocks only when profiling is on (they were always taken); wall time of the synchronous call, profiling off:
# build_exp/agentscope.so (clocks always on, agent scope) against the shipped build, claim policies 0 / 3; parity
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for LIB in build_exp/agentscope.so ""; do for POL in 0 3; do
  echo "== lib=${LIB:-shipped} PLP_LSD_MW_POLICY=$POL" >> $O/calls.log
  (PLP_FRONT_LIB=$LIB PLP_LSD_MW_POLICY=$POL timeout 120 python tools/experiments/latency_calls.py 128 2>&1 | grep -v amdgpu.ids | tail -2) >> $O/calls.log
done; done
cat $O/calls.log
(timeout 300 python -m pytest tests/test_gpu_line.py -x -q 2>&1 | tail -3) > $O/pytest_clk.log; cat $O/pytest_clk.log
(PLP_LSD_MW_POLICY=3 timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -4) > $O/latency7.log; cat $O/latency7.log
