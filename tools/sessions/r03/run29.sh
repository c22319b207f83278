# GPU session 29: queries per block of k_match_topk_cells, 256 against 512, six passes
export TMPDIR=/tmp
O=gpurun_out/r03x10; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 env $2 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 $2 |', j['value'], j['ms_per_step'], '| grow', s['lsd_grow'], 'match_4x', s['match_4x'])"; }
{
for pass in 1 2 3 4 5 6; do
B lanes2 PLP_MATCH_QPB=256
B lanes2 PLP_MATCH_QPB=512
B cur7 PLP_MATCH_QPB=256
done
} > $O/ab.log 2>&1
cat $O/ab.log
cp build_exp/.orig.so $L
