# GPU session r06/61: the classifying pass of a global partition software-pipelined (the next trip's loads in flight while this trip is classified): 16 + 16 and 8 + 8 chunks per wave
export TMPDIR=/tmp
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
cp build_exp/pipeA8.so $L; timeout 600 python -m pytest tests/test_gpu_seed_sort.py -x -q -m gpu 2>&1 | tail -2
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 8 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'sort alone', round(s['lsd_order'],3), 'verified', j.get('verified_frames'))"; }
for pass in 1 2 3; do for v in base pipeA pipeA8; do B $v; done; done
cp build_exp/.orig.so $L
