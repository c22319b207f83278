# GPU session r05/33: the latency path -- a round of region growing by phase, helper 1 of the several-waves kernel against the one-wave kernel (diagnostic build)
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
(PLP_FRONT_LIB=build_exp/rprof.so timeout 120 python tools/experiments/mw_round_profile.py 2>&1 | grep -v amdgpu.ids | tail -6) > $O/mw_round.log; cat $O/mw_round.log
