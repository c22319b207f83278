# GPU session r05/36: the latency series against the TRUE baseline (the committed tree before it, build_exp/r05z_base.so), same box: 128 synchronous calls (profiling off)
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
for LIB in build_exp/r05z_base.so "" build_exp/r05z_base.so ""; do
  echo "== lib=${LIB:-shipped}" >> $O/calls_base.log
  (PLP_FRONT_LIB=$LIB timeout 120 python tools/experiments/latency_calls.py 128 2>&1 | grep -v amdgpu.ids | tail -2) >> $O/calls_base.log
done
cat $O/calls_base.log
