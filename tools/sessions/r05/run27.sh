# GPU session r05/27: the latency path -- what the HELPERS spend their time on (claiming groups / growing / waiting for room), claim policies 0 / 3
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for POL in 0 3; do
  echo "== PLP_LSD_MW_POLICY=$POL" >> $O/latency5.log
  (PLP_LSD_MW_POLICY=$POL timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -6) >> $O/latency5.log
done
cat $O/latency5.log
