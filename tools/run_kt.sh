export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kt1
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
cd $R
python tools/rocpd_summary.py $O/kt/kt_results.db "kt1" > $O/kt1_kernel_stats.md
python tools/rocpd_isolated.py $O/kt/kt_results.db "kt1" > $O/kt1_isolated.md 2>&1
rm -rf $O/kt
cat $O/kt1_isolated.md | head -60
