# GPU session r05/28: the latency path -- the hand-over of a helper's lists at WORKGROUP scope (was agent: sc1 loads + buffer_wbl2 per finished region); A/B against a
# build with the old scope (build_exp/agentscope.so), claim policies 0 / 3; parity
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for LIB in build_exp/agentscope.so ""; do for POL in 0 3; do
  echo "== lib=${LIB:-shipped} PLP_LSD_MW_POLICY=$POL" >> $O/latency6.log
  (PLP_FRONT_LIB=$LIB PLP_LSD_MW_POLICY=$POL timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -6) >> $O/latency6.log
done; done
cat $O/latency6.log
(timeout 300 python -m pytest tests/test_gpu_line.py -x -q 2>&1 | tail -3) > $O/pytest_wg.log; cat $O/pytest_wg.log
(timeout 200 python tools/fuzz_gpu.py --only lines --seconds 50 --seed 112 2>&1 | tail -3) > $O/fuzz_wg.log; cat $O/fuzz_wg.log
(PLP_LSD_MW_POLICY=3 timeout 200 python tools/fuzz_gpu.py --only lines --seconds 30 --seed 113 2>&1 | tail -3) > $O/fuzz_wg3.log; cat $O/fuzz_wg3.log
