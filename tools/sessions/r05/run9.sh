# GPU session r05/9: the full GPU suite on the current tree (per-level quotas up to 1960 among it), line sub-blocks per step 1 / 2 / 3 / 4 again with this round's kernels,
# the other BASELINE configs (every step verified) and the TUM RGB-D full step at K = 2000
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 $2 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'order', round(s['lsd_order'],2), 'grow', round(s['lsd_grow'],2))" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
for n in 2 1 3 4; do PLP_BENCH_LINE_SPLIT=$n B split$n ""; done
B k2000 "--keypoints 2000"
(timeout 600 python tools/bench_configs.py --batch 1024 --steps 3 --verify 8 2>&1 | grep -v amdgpu.ids | tail -12) > $O/other_configs.log; cat $O/other_configs.log
