# GPU session r06/57: the sort with 2 / 3 / 8 waves per frame in the step, now that the loop-header barrier is sound (rounds 4 - 6 could not time 2 waves in the overlapped step: it faulted)
export TMPDIR=/tmp
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'sort alone', round(s['lsd_order'],3), 'verified', j.get('verified_frames'))"; }
for pass in 1 2; do for v in base w2 w2m w3 w8; do B $v; done; done
cp build_exp/.orig.so $L
