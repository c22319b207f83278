export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/kt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/kt/kt.log 2>&1
cd $R && python tools/rocpd_summary.py gpurun_out/kt/kt_results.db | head -28
