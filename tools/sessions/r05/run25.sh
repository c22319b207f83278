# GPU session r05/25: the latency path -- a FINISHED region's claim: bet that it is an earlier seed's and will be committed (assume used, checked by main)
# instead of giving up: policy 2 (that helper has a finished region from an earlier seed), policy 3 (that helper is at an earlier seed now); park 0 / 4
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for POL in 2 3; do for P in 0 4; do
  echo "== PLP_LSD_MW_POLICY=$POL PLP_LSD_MW_PARK=$P" >> $O/latency3.log
  (PLP_LSD_MW_POLICY=$POL PLP_LSD_MW_PARK=$P timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -6) >> $O/latency3.log
done; done
cat $O/latency3.log
(PLP_LSD_MW_POLICY=2 PLP_LSD_MW_PARK=4 timeout 300 python -m pytest tests/test_gpu_line.py -x -q 2>&1 | tail -3) > $O/pytest_pol2.log; cat $O/pytest_pol2.log
(PLP_LSD_MW_POLICY=3 PLP_LSD_MW_PARK=4 timeout 300 python -m pytest tests/test_gpu_line.py -x -q 2>&1 | tail -3) > $O/pytest_pol3.log; cat $O/pytest_pol3.log
