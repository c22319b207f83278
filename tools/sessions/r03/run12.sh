# GPU session 12: batched LDS loads in the rectangle fit (chain0 = one load per add, chain4, chain8), same box
O=gpurun_out/r03m; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 $2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 $2 |', j['value'], j['ms_per_step'], s['lsd_grow'], s['match_4x'])"; }
{ for pass in 1 2; do for v in chain8 chain16; do B $v; done; done; } > $O/ab.log 2>&1
cp build_exp/chain16.so $L
(timeout 200 python -m pytest tests/test_gpu_line.py tests/test_gpu_golden_ref.py tests/test_gpu_bench_step.py -m gpu -x -q 2>&1 | tail -2) >> $O/ab.log
(timeout 60 python tools/fuzz_gpu.py --only lines --seconds 30 --seed 57 2>&1 | grep lines) >> $O/ab.log
cp build_exp/.orig.so $L
cat $O/ab.log
