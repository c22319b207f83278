# Short round-end check: bench line without the CPU leg, then a kernel trace.  Usage (GPU box): bash tools/run_kt.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-kt1}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R && timeout 40 python bench.py --no-cpu-baseline > $O/${T}_bench.json 2> $O/bench.err
cat $O/${T}_bench.json
cd /tmp
timeout 50 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
cd $R
python tools/rocpd_summary.py $O/kt/kt_results.db "$T (bench.py --steps 3 --warmup 1)" > $O/${T}_full_kernel_stats.md
rm -rf $O/kt
head -12 $O/${T}_full_kernel_stats.md
