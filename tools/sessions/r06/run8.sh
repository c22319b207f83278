# GPU session r06/8: seed sort phase clocks of workgroup 0 alone and with the chip full (2048 workgroups sorting 2048 copies: PLP_SEED_SORT_DBG_COPIES)
export TMPDIR=/tmp
O=gpurun_out/r06h; mkdir -p $O
for c in 1 2048; do PLP_SEED_SORT_DBG_COPIES=$c timeout 300 python tools/experiments/seed_sort_prof.py > $O/prof_$c.log 2>&1; echo "copies $c"; tail -1 $O/prof_$c.log; done
