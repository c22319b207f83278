# GPU session r06/13: the single-kernel sort again after the rewrite of its global phase -- 2 / 4 / 8 waves per frame x occupancy, same box
export TMPDIR=/tmp
O=gpurun_out/r06m; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; PLP_SS_SPLIT=0 timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 16 --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'order', s['lsd_order'], 'verified', j['verified_frames'])"; }
for pass in 1 2; do for v in f4_w4 f4_w8 f8_w8 f8_w6 f2_w8; do B $v; done; done 2>&1 | tee $O/ab.log
cp build_exp/.orig.so $L
