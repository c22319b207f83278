# GPU session r04/54: the round's profile again after the late change to k_lsd_grow (bench line with the CPU leg, rocprofv3 kernel trace, SQ counters; the PMC
# traffic passes are not repeated: the change does not touch what the kernels read and write)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=r04f
O=$R/gpurun_out/$T
mkdir -p $O
cd $R && timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --verify 0 > $O/kt.log 2>&1
export PLP_BENCH_LINE_SPLIT=1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq -o sq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --verify 0 > $O/sq.log 2>&1
cd $R
python tools/rocpd_summary.py $O/kt/kt_results.db "$T (bench.py --steps 3 --warmup 1)" > $O/${T}_full_kernel_stats.md
python tools/sq_table.py $O/sq/sq_results.db > $O/${T}_sq_counters.md
cp $O/bench.json $O/${T}_bench.json
rm -rf $O/kt $O/sq
tail -1 $O/bench.json | cut -c1-400
