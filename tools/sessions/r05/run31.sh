# GPU session r05/31: the latency path -- a wave reads its own short list from the LDS ring in the hand-over (was: three HBM round trips per attempt); wall time of the
# synchronous call (profiling off) against build_exp/agentscope.so (the state before this series), the helpers' clocks, parity
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for LIB in build_exp/agentscope.so ""; do
  echo "== lib=${LIB:-shipped}" >> $O/calls2.log
  (PLP_FRONT_LIB=$LIB timeout 120 python tools/experiments/latency_calls.py 128 2>&1 | grep -v amdgpu.ids | tail -2) >> $O/calls2.log
done
echo "== lib=shipped PLP_LSD_MW_POLICY=3" >> $O/calls2.log
(PLP_LSD_MW_POLICY=3 timeout 120 python tools/experiments/latency_calls.py 128 2>&1 | grep -v amdgpu.ids | tail -2) >> $O/calls2.log
cat $O/calls2.log
(timeout 300 python -m pytest tests/test_gpu_line.py -x -q 2>&1 | tail -3) > $O/pytest_ring.log; cat $O/pytest_ring.log
(timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -4) > $O/latency8.log; cat $O/latency8.log
(timeout 200 python tools/fuzz_gpu.py --only lines --seconds 40 --seed 114 2>&1 | tail -3) > $O/fuzz_ring.log; cat $O/fuzz_ring.log
