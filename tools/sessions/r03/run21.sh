# GPU session 21: old FAST / rBRIEF back (session 20), lane candidates of k_match_topk_cells (2: 63 VGPRs, 1: 54), grower wave priority; PCIe: staging buffers
export TMPDIR=/tmp
O=gpurun_out/r03x3; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 |', j['value'], j['ms_per_step'], '| grow', s['lsd_grow'], 'match_4x', s['match_4x'])"; }
{
for pass in 1 2 3; do
B base
B cur
B lc2
B lc1
B p1
B p2
B p3
B lc2p1
done
} > $O/ab.log 2>&1
cat $O/ab.log
cp build_exp/cur.so $L
P() { timeout 200 env $1 python bench.py --no-cpu-baseline --no-latency --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 |', j['value'], j['ms_per_step'], '| pcie', j.get('pcie_inclusive_value'), j.get('pcie_inclusive_ms_per_step'))"; }
{
P PLP_BENCH_STAGES=2
P PLP_BENCH_STAGES=3
P PLP_BENCH_STAGES=2
P PLP_BENCH_STAGES=3
P "PLP_BENCH_STAGES=2 GPU_MAX_HW_QUEUES=16"
} > $O/pcie.log 2>&1
cat $O/pcie.log
(timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
cp build_exp/.orig.so $L
