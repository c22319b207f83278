# GPU session r06/3: phase clocks of the seed sort (debug entry, one workgroup alone): the round's base library against the tree with wave_sub_sort
export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.cand.so
cp build_exp/base.so $L; timeout 300 python tools/experiments/seed_sort_prof.py > $O/prof_base.log 2>&1; tail -4 $O/prof_base.log
cp build_exp/.cand.so $L; timeout 300 python tools/experiments/seed_sort_prof.py > $O/prof_cand.log 2>&1; tail -4 $O/prof_cand.log
