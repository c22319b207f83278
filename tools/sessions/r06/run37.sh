# GPU session r06/37: seed sort -- segments above 512 / 1024 / 2048 entries inside the LDS window partitioned by the whole workgroup (wg_partition on LDS) instead of one wave (4096: shipped); same box, sort tests of the last variant
export TMPDIR=/tmp
O=gpurun_out/r06task; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 16 --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'order', s['lsd_order'], 'verified', j['verified_frames'])"; }
for pass in 1 2; do for v in task4096 task2048 task1024 task512; do B $v; done; done 2>&1 | tee $O/ab.log
cp build_exp/task1024.so $L; (timeout 600 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -3) | tee $O/ss.log
cp build_exp/.orig.so $L
