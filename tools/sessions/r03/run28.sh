# GPU session 28: key-line search with one or two waves per frame walking the query blocks (was a workgroup per block of the capacity); queries per block of k_match_topk_cells
export TMPDIR=/tmp
O=gpurun_out/r03x9; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
(timeout 300 python -m pytest tests/test_gpu_match.py tests/test_gpu_bench_step.py tests/test_gpu_replay_sharded.py tests/test_gpu_golden_ref.py tests/test_gpu_facade.py -q -p no:cacheprovider -x 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
(timeout 80 python tools/fuzz_gpu.py --only match --seconds 40 --seed 93 2>&1 | grep -i "match" | tail -1) > $O/fuzz.log; cat $O/fuzz.log
B() { cp build_exp/$1.so $L; timeout 120 env $2 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 $2 |', j['value'], j['ms_per_step'], '| grow', s['lsd_grow'], 'match_4x', s['match_4x'])"; }
{
for pass in 1 2 3; do
B cur7
B lanes2
B lanes2 PLP_MATCH_QPB=128
B lanes2 PLP_MATCH_QPB=512
done
} > $O/ab.log 2>&1
cat $O/ab.log
cp build_exp/.orig.so $L
