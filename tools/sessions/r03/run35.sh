# GPU session 35: workgroups per frame of k_lbd (PLP_LBD_BLOCKS; 4 was tuned when the kernel held 76 VGPRs, it holds 59 now)
export TMPDIR=/tmp
O=gpurun_out/r03x15; mkdir -p $O
B() { timeout 120 env $1 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 |', j['value'], j['ms_per_step'], '| lbd', s['lbd'], 'grow', s['lsd_grow'])"; }
{
for pass in 1 2 3; do
B PLP_LBD_BLOCKS=4
B PLP_LBD_BLOCKS=2
B PLP_LBD_BLOCKS=8
B PLP_LBD_BLOCKS=16
done
} > $O/ab.log 2>&1
cat $O/ab.log
