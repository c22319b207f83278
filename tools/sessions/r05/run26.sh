# GPU session r05/26: the latency path -- main's seed order arrives eight loads at a time, a chunk ahead (was: one load ahead); claim policies 0 / 3; parity
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for POL in 0 3; do
  echo "== PLP_LSD_MW_POLICY=$POL" >> $O/latency4.log
  (PLP_LSD_MW_POLICY=$POL timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -6) >> $O/latency4.log
done
cat $O/latency4.log
(timeout 300 python -m pytest tests/test_gpu_line.py -x -q 2>&1 | tail -3) > $O/pytest_pre.log; cat $O/pytest_pre.log
(PLP_LSD_MW_POLICY=3 timeout 200 python tools/fuzz_gpu.py --only lines --seconds 40 --seed 111 2>&1 | tail -3) > $O/fuzz_pre.log; cat $O/fuzz_pre.log
