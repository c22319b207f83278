# Round profile: default bench line, kernel trace, PMC traffic passes, SQ counters.  Usage (GPU box): bash tools/run_prof.sh <tag>
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-prof}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R && timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --verify 0 > $O/kt.log 2>&1
# counter passes: one launch = the whole 2048-frame batch for every kernel (no sub-block split of the line path)
export PLP_BENCH_LINE_SPLIT=1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --verify 0 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --verify 0 > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq -o sq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --verify 0 > $O/sq.log 2>&1
cat $O/bench.json
# post-process on the box and drop the databases (hundreds of MB): what travels back is what gets committed under profiles/
cd $R
python tools/rocpd_summary.py $O/kt/kt_results.db "$T (bench.py --steps 3 --warmup 1)" > $O/${T}_full_kernel_stats.md
python tools/pmc_traffic.py $O/pmc_fetch/f_results.db $O/pmc_write/w_results.db --batch 2048 --md $O/${T}_pmc_traffic.md --json $O/pmc_traffic.json --source profiles/${T}_pmc_traffic.md
python tools/sq_table.py $O/sq/sq_results.db > $O/${T}_sq_counters.md
python tools/step_profile.py $O/kt/kt_results.db $O/sq/sq_results.db --bench $O/bench.json --out $O/step_profile.json --source profiles/${T}_full_kernel_stats.md,profiles/${T}_sq_counters.md > $O/step_profile.log 2>&1; tail -30 $O/step_profile.log
cp $O/bench.json $O/${T}_bench.json
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/sq
ls -la $O
