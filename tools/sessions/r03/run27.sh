# GPU session 27: pyramid with a 64 x 32 strip per WAVE (was a 256 x 32 tile per workgroup, one 256-pixel row of lanes per wave)
export TMPDIR=/tmp
O=gpurun_out/r03x8; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
(timeout 300 python -m pytest tests/test_gpu_orb.py tests/test_gpu_bench_step.py tests/test_gpu_golden_ref.py tests/test_gpu_facade.py tests/test_gpu_stereo_lbdmatch.py tests/test_gpu_replay_driver.py -q -p no:cacheprovider -x 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
(timeout 100 python tools/fuzz_gpu.py --only orb --seconds 60 --seed 73 2>&1 | grep "orb:" | tail -1) > $O/fuzz.log; cat $O/fuzz.log
B() { cp build_exp/$1.so $L; timeout 120 env $2 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 $2 |', j['value'], j['ms_per_step'], '| pyramid', s['pyramid'], 'grow', s['lsd_grow'], 'match_4x', s['match_4x'])"; }
{
for pass in 1 2 3; do
B cur6
B pyr
done
} > $O/ab.log 2>&1
cat $O/ab.log
cp build_exp/.orig.so $L
