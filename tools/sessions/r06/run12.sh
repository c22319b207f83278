# GPU session r06/12: the two-kernel sort -- waves per frame of the global-phase kernel x occupancy (16 waves at 7 / 8 waves per SIMD: one / two workgroups per CU; 8 waves; 4 waves), against the single kernel, same box
export TMPDIR=/tmp
O=gpurun_out/r06l; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; PLP_SS_SPLIT=$2 timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 16 --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 split=$2', j['value'], j['ms_per_step'], 'order', s['lsd_order'], 'verified', j['verified_frames'])"; }
for pass in 1 2; do B top16_w2 0; for v in top16_w2 top16_w8 top8_w8 top8_w4 top4_w8; do B $v 1; done; done 2>&1 | tee $O/ab.log
cp build_exp/.orig.so $L
