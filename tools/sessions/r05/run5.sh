# GPU session r05/5: the soak test of the shipped seed sort, bench.py at N = 2 on one GPU (verified on every rank), and the 2-wave experiment build of the seed sort
# (the one that faulted in round 4) rebuilt from this round's sources: does it still fault beside the other kernels?
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_seed_sort_soak.py tests/test_gpu_bench_two_ranks.py -q -x -p no:cacheprovider 2>&1 | tail -15) > $O/pytest.log; cat $O/pytest.log
for r in 1 2 3; do
  (PLP_FRONT_LIB=build_exp/ss2w.so timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_ss2w_$r.err | tail -1) > $O/bench_ss2w_$r.json
  echo "ss2w run $r: rc=$? $(cut -c1-120 $O/bench_ss2w_$r.json) | $(grep -i -m2 'fault\|error\|abort' $O/bench_ss2w_$r.err | cut -c1-200)"
done
