# GPU session r06/5: seed sort occupancy -- LDS window 4096 / 2048 entries x at least 4 / 6 / 8 waves per SIMD (launch bounds; 80 / 64 VGPRs with 12 / 48 bytes of scratch), with the capacities that follow the window; same box, round robin twice
export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 16 --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'order', s['lsd_order'], 'grow', s['lsd_grow'], 'verified', j['verified_frames'], 'stable', j['other_seed_order']['value'] if j.get('other_seed_order') else None)"; }
for pass in 1 2; do for v in base t4096_w4 t4096_w6 t4096_w8 t2048_w6 t2048_w8; do B $v; done; done 2>&1 | tee $O/ab.log
cp build_exp/.orig.so $L
