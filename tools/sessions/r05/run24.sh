# GPU session r05/24: why do the helpers of the latency path give up?  Counters: give-ups over a FINISHED region's claim (no park for those), parks
# that ran out, main's self-grown regions in groups nobody claimed; the two claim policies x park 0 / 4
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for POL in 0 1; do for P in 0 4; do
  echo "== PLP_LSD_MW_POLICY=$POL PLP_LSD_MW_PARK=$P" >> $O/latency2.log
  (PLP_LSD_MW_POLICY=$POL PLP_LSD_MW_PARK=$P timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -6) >> $O/latency2.log
done; done
cat $O/latency2.log
