# GPU session r06/60: k_lsd_grow's waves at raised issue priority (s_setprio 1 / 3) in the step -- round 4 measured this inside the spread at register packing 0.75; + the GPU test of the barrier reproducer
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_barrier_reproducer.py -q -m gpu -s 2>&1 | tail -5
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'grow alone', round(s['lsd_grow'],3), 'verified', j.get('verified_frames'))"; }
for pass in 1 2 3; do for v in base prio1 prio3; do B $v; done; done
cp build_exp/.orig.so $L
