# GPU session r06/11: the two-kernel sort -- why is it slower?  Kernel trace of the isolated stage passes (split and single kernel), the failing tests in full
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_seed_sort_soak.py -q -x -p no:cacheprovider 2>&1 | tail -30) > $O/ss.log; tail -12 $O/ss.log
cd /tmp
for sp in 1 0; do
PLP_SS_SPLIT=$sp PLP_BENCH_LINE_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt$sp -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --verify 0 > $O/kt$sp.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_isolated.py $O/kt$sp/kt_results.db "split $sp" > $O/iso$sp.md; grep -E "seed|order" $O/iso$sp.md
rm -rf $O/kt$sp
done
