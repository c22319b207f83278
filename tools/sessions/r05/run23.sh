# GPU session r05/23: the latency path's park / resume (region_grow<MW>: wait for an earlier seed's growing region instead of giving up;
# PLP_LSD_MW_PARK = waits per attempt, 0 = the old give-up): single-frame stage times + the main wave's own counters, then parity
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
for P in 0 1 2 4 8; do
  echo "== PLP_LSD_MW_PARK=$P" >> $O/latency.log
  (PLP_LSD_MW_PARK=$P timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -4) >> $O/latency.log
done
cat $O/latency.log
for P in 2 4; do
  (PLP_LSD_MW_PARK=$P timeout 300 python -m pytest tests/test_gpu_line.py -x -q 2>&1 | tail -3) > $O/pytest_park$P.log; cat $O/pytest_park$P.log
  (PLP_LSD_MW_PARK=$P timeout 200 python tools/fuzz_gpu.py --only lines --seconds 45 --seed $((100+P)) 2>&1 | tail -3) > $O/fuzz_park$P.log; cat $O/fuzz_park$P.log
done
