# GPU session r06/64: frames per workgroup of k_lsd_grow (PLP_LSD_WPB = 1 / 2 / 4; 4 shipped since round 3) in the step, on the 64-frame workload and on 2048 distinct frames
export TMPDIR=/tmp
B() { timeout 150 env PLP_LSD_WPB=$1 python bench.py --distinct $2 --no-cpu-baseline --no-extras --verify 8 --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('wpb $1 distinct $2', j['value'], j['ms_per_step'], 'grow alone', round(s['lsd_grow'],3), 'verified', j.get('verified_frames'))"; }
for pass in 1 2; do for d in 64 2048; do for w in 4 2 1; do B $w $d; done; done; done
