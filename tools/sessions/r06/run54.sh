# GPU session r06/54: the cause of the second-dispatch failure confirmed.  The failing builds lack `s_waitcnt lgkmcnt(0)` before the barrier that heads the loop over a frame's
# global partitions (tools/isa_barrier_check.py); with the hard wait of csrc/plp_barrier.hpp the same three builds must pass, and with -DPLP_SOFT_BARRIERS fail as before.
# Then the shipped library (hard waits): sort tests, soaks, line + concurrent tests, bench.
export TMPDIR=/tmp
O=gpurun_out/r06nb5; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.cand.so
for v in soft_v7 fix_v7 fix_asm0 fix_flat; do
  cp build_exp/$v.so $L
  for i in 1 2 3; do FLN_CASES="lines:2" timeout 300 python tools/experiments/flat_neighbours.py > $O/${v}_$i.log 2>&1; echo "$v run $i: $(grep '^parts' $O/${v}_$i.log || echo 'process died (memory fault)')"; done
done
cp build_exp/.cand.so $L
timeout 1500 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_seed_sort_soak.py tests/test_gpu_line.py tests/test_gpu_concurrent_single_frame.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_$i.json; python -c "import json;d=json.load(open('$O/bench_$i.json'));print('bench',d['value'],d['ms_per_step'],d['roofline']['stage_ms_per_batch']['lsd_order'])"; done
