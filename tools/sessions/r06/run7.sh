# GPU session r06/7: seed sort -- clocks inside the small-subtree path (register sort of <= 64 entries, scalar-mask sort of 65..256)
export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
timeout 300 python tools/experiments/seed_sort_prof.py > $O/prof.log 2>&1; tail -2 $O/prof.log
