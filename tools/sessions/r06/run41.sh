# GPU session r06/41: seed sort -- bins from a threshold table (f32 estimate corrected against T[b]) instead of the f64 square root per pixel; with the fused first partition (l0_tab) and without (tab_nol0) against the tree before (pre_l0); tests, same-box A/B
export TMPDIR=/tmp
O=gpurun_out/r06l0b; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_line.py tests/test_gpu_seed_sort_soak.py tests/test_gpu_concurrent_single_frame.py tests/test_gpu_bench_step.py -q -x -p no:cacheprovider 2>&1 | tail -4) | tee $O/pytest.log
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 64 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'order', s['lsd_order'], 'verified', j['verified_frames'])"; }
for pass in 1 2 3; do for v in l0_tab tab_nol0 pre_l0; do B $v; done; done 2>&1 | tee $O/ab.log
cp build_exp/.orig.so $L
timeout 120 python tools/experiments/latency_stages.py 2>&1 | grep order | tee $O/lat.log
