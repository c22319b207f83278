# GPU session r06/63: the timeline of an overlapped step on the final tree (kernel trace of bench.py --steps 8): which queue is busy for how much of the step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06tl; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --verify 0 > $O/kt.log 2>&1
tail -1 $O/kt.log | cut -c1-200
cd $R
python tools/rocpd_timeline.py $O/kt/kt_results.db 6 > $O/r06_step_timeline.md 2> $O/timeline.err; head -12 $O/r06_step_timeline.md
python tools/rocpd_timeline.py $O/kt/kt_results.db 5 | head -10
rm -rf $O/kt
