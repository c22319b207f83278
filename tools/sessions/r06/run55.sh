# GPU session r06/55: wg_barrier() (hard LDS wait) in EVERY kernel of the library: the whole GPU suite, then a same-box A/B against the build that has it in the sort only
export TMPDIR=/tmp
O=gpurun_out/r06hard; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], {k: round(v,3) for k,v in s.items()})"; }
for pass in 1 2 3; do for v in sort_only all_hard; do B $v; done; done
cp build_exp/.orig.so $L
timeout 200 python tools/experiments/latency_stages.py 2>&1 | tail -8
