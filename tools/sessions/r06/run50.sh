# GPU session r06/50: k_fast_cells -- the barrier behind the "found anything?" read only where a second attempt follows; ORB tests, same-box A/B
export TMPDIR=/tmp
O=gpurun_out/r06fast; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_golden_ref.py -q -x -p no:cacheprovider 2>&1 | tail -3) | tee $O/pytest.log
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'fast', s['fast_cells'], 'verified', j['verified_frames'])"; }
for pass in 1 2 3; do for v in fast1 pre_fast; do B $v; done; done 2>&1 | tee $O/ab.log
cp build_exp/.orig.so $L
