# GPU session r06/65: eight frames per workgroup of k_lsd_grow (launch bounds 512, 85 KB of LDS per workgroup: one workgroup per CU at 2048 frames) against the shipped four
export TMPDIR=/tmp
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 150 env PLP_LSD_WPB=$2 python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 wpb $2', j['value'], j['ms_per_step'], 'grow alone', round(s['lsd_grow'],3), 'verified', j.get('verified_frames'))"; }
for pass in 1 2; do B base 4; B wpb8 4; B wpb8 8; done
cp build_exp/.orig.so $L
