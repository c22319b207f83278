# GPU session 20: which of session 19's ORB changes pay in the STEP (A/B switches), isolated per-kernel times of the new state (SQ pass), PCIe arrangements
export TMPDIR=/tmp
O=gpurun_out/r03x2; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 |', j['value'], j['ms_per_step'], '| grow', s['lsd_grow'], 'fast', s['fast_cells'], 'quadtree', s['quadtree'], 'rbrief', s['orient_rbrief'], 'match_4x', s['match_4x'])"; }
{
for pass in 1 2; do
B base
B new
B fs0
B fp0
B f00
B rb0
B f00rb0
done
} > $O/ab.log 2>&1
cat $O/ab.log
cp build_exp/new.so $L
P() { timeout 200 env $1 python bench.py --no-cpu-baseline --no-latency --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 |', j['value'], j['ms_per_step'], '| pcie', j.get('pcie_inclusive_value'), j.get('pcie_inclusive_ms_per_step'))"; }
{
P PLP_BENCH_PREFETCH=0
P PLP_BENCH_PREFETCH=1
P "PLP_BENCH_PREFETCH=1 GPU_MAX_HW_QUEUES=8"
P "PLP_BENCH_PREFETCH=0 GPU_MAX_HW_QUEUES=8"
} > $O/pcie.log 2>&1
cat $O/pcie.log
cd /tmp
export PLP_BENCH_LINE_SPLIT=1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/$O/sq -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --verify 0 > $GRAFT_REPO_ROOT/$O/sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/sq_table.py $O/sq/sq_results.db > $O/sq_counters_new.md; rm -rf $O/sq
cat $O/sq_counters_new.md
cp build_exp/.orig.so $L
