# GPU session 37: single-frame line latency with 16 LBD workgroups per frame for small batches (2 for batches); line tests on the committed tree
export TMPDIR=/tmp
O=gpurun_out/r03x17; mkdir -p $O
P() { timeout 200 env $1 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); l=j['latency_ms_median_mean']; print('$1 |', j['value'], '| line', l['line_extract'], 'pair', l['orb_par_line_extract'])"; }
{
P PLP_NONE=1
P PLP_LBD_BLOCKS=4
P PLP_LBD_BLOCKS=2
} > $O/lat.log 2>&1
cat $O/lat.log
(timeout 120 python -m pytest tests/test_gpu_line.py tests/test_gpu_bench_step.py -q -p no:cacheprovider -x 2>&1 | tail -1) > $O/pytest.log; cat $O/pytest.log
