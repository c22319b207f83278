# GPU session r04/40: where the single-frame line call spends its time (both seed orders)
export TMPDIR=/tmp
(timeout 120 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -6) > gpurun_out/latency_profile.log; cat gpurun_out/latency_profile.log
