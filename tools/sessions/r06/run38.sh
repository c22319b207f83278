# GPU session r06/38: what the per-address-space specialisation of k_quadtree costs inside the step (two inlined copies, 83 spilled SGPRs) against round 5's run-time pointers (qt_flat); same box, three passes
export TMPDIR=/tmp
O=gpurun_out/r06qt; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'quadtree', s['quadtree'], 'verified', j['verified_frames'])"; }
for pass in 1 2 3; do for v in qt_spec qt_flat; do B $v; done; done 2>&1 | tee $O/ab.log
cp build_exp/.orig.so $L
