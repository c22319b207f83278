# GPU session r04/51: k_lsd_grow's own cycle counters (growth / rectangle fits / refinement), one wave per frame
export TMPDIR=/tmp
(timeout 120 python tools/experiments/grow_profile.py 2>&1 | grep -v amdgpu.ids | tail -7) > gpurun_out/grow_profile.log; cat gpurun_out/grow_profile.log
