# GPU session r05/2: cycles of a region-growing round by phase (diagnostic build -DPLP_GROW_PROF_ROUND, frame 0 of 2048), with and without the record cache
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
export PLP_FRONT_LIB=build_exp/rprof.so
(timeout 120 python tools/grow_profile.py 2048 2>&1 | grep -v amdgpu.ids | tail -3) > $O/rprof_cache.log; cat $O/rprof_cache.log
(PLP_LSD_CACHE=0 timeout 120 python tools/grow_profile.py 2048 2>&1 | grep -v amdgpu.ids | tail -3) > $O/rprof_nocache.log; cat $O/rprof_nocache.log
