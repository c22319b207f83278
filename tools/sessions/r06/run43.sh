# GPU session r06/43: bins from the threshold table without the fused first partition (tab_nol0) against the tree before (pre_l0) and with the fusion (l0_tab); same box, three passes
export TMPDIR=/tmp
O=gpurun_out/r06l0b; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 200 python bench.py --no-cpu-baseline --no-extras --verify 64 --steps 16 --warmup 4 2>$O/err_$1.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'order', s['lsd_order'], 'verified', j['verified_frames'])"; }
for pass in 1 2 3; do for v in tab_nol0 pre_l0 l0_tab; do B $v; done; done 2>&1 | tee $O/ab3.log
cp build_exp/.orig.so $L
