# GPU session r05/29: the latency path -- main's clocks ALONG the seed list (diagnostic build with device printf), claim policy 0 / 3
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for POL in 0 3; do
  echo "== PLP_LSD_MW_POLICY=$POL" >> $O/timeline.log
  (PLP_FRONT_LIB=build_exp/timeline.so PLP_LSD_MW_POLICY=$POL timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | grep "timeline\|order" | tail -40) >> $O/timeline.log
done
tail -42 $O/timeline.log
