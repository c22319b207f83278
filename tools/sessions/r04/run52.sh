# GPU session r04/52: what the refinement of k_lsd_grow is made of (diagnostic build -DPLP_GROW_PROF_REFINE)
export TMPDIR=/tmp
export PLP_FRONT_LIB=build_exp/profref.so
(timeout 120 python tools/experiments/grow_profile.py 2>&1 | grep -v amdgpu.ids | tail -7) > gpurun_out/grow_profile2.log; cat gpurun_out/grow_profile2.log
