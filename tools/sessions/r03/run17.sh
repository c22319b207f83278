# GPU session 17: k_match_topk_cells with fewer descriptor fetches in flight per lane but two waves per SIMD beside the growers
O=gpurun_out/r03w; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 |', j['value'], j['ms_per_step'], s['lsd_grow'], s['match_4x'])"; }
{ for pass in 1 2; do for v in rect170 rect147; do B $v; done; done; } > $O/ab.log 2>&1
cp build_exp/rect147.so $L
(timeout 200 python -m pytest tests/test_gpu_line.py tests/test_gpu_bench_step.py tests/test_gpu_golden_ref.py -m gpu -x -q 2>&1 | tail -2) >> $O/ab.log
cp build_exp/.orig.so $L
cat $O/ab.log
(timeout 60 python tools/fuzz_gpu.py --only lines --seconds 30 --seed 60 2>&1 | grep lines:)
