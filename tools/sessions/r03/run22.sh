# GPU session 22: k_match_topk_cells with one candidate per lane + LDS sized by the caller's count hint; orientation + rBRIEF variants (serialized disc rows,
# with / without the 40-byte patch rows); grower priority 2 once more
export TMPDIR=/tmp
O=gpurun_out/r03x4; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
(timeout 300 python -m pytest tests/test_gpu_match.py tests/test_gpu_bench_step.py tests/test_gpu_replay_sharded.py tests/test_gpu_golden_ref.py -q -p no:cacheprovider -x 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
for v in rbA rbC; do cp build_exp/$v.so $L; (timeout 200 python -m pytest tests/test_gpu_orb.py tests/test_gpu_bench_step.py -q -p no:cacheprovider -x 2>&1 | tail -1) >> $O/pytest.log; done; cat $O/pytest.log
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 |', j['value'], j['ms_per_step'], '| grow', s['lsd_grow'], 'rbrief', s['orient_rbrief'], 'match_4x', s['match_4x'])"; }
{
for pass in 1 2 3; do
B cur
B cur2
B cur2p2
B rbA
B rbC
done
} > $O/ab.log 2>&1
cat $O/ab.log
cp build_exp/.orig.so $L
