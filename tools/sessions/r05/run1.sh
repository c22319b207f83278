# GPU session r05/1: k_lsd_grow with the per-wave record cache + ring-2 prefetch (default) against the kernel without it (PLP_LSD_CACHE=0),
# and the two small variants: angle test without v_rsq (norsq), rectangle refits from the list (refit)
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_line.py tests/test_gpu_golden_ref.py tests/test_gpu_bench_step.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/pytest_cache.log; echo "cache: $(tail -1 $O/pytest_cache.log)"
(PLP_LSD_CACHE=0 timeout 300 python -m pytest tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -1) > $O/pytest_nocache.log; echo "nocache: $(cat $O/pytest_nocache.log)"
(timeout 120 python tools/grow_stats.py 2>&1 | grep -v amdgpu.ids | tail -6) > $O/grow_stats_cache.log; cat $O/grow_stats_cache.log
(PLP_LSD_CACHE=0 timeout 120 python tools/grow_stats.py 2>&1 | grep -v amdgpu.ids | tail -6) > $O/grow_stats_nocache.log; cat $O/grow_stats_nocache.log
B() {
  (timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'grow alone', s['lsd_grow'])" || tail -2 $O/bench_$1.err
}
B cache
PLP_LSD_CACHE=0 B nocache
for v in norsq refit norsq_refit; do
  export PLP_FRONT_LIB=build_exp/$v.so
  (timeout 200 python -m pytest tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -1) > $O/pytest_$v.log; echo "$v: $(cat $O/pytest_$v.log)"
  B $v
  unset PLP_FRONT_LIB
done
B cache2
(timeout 60 python tools/fuzz_gpu.py --only lines --seconds 40 --seed 81 2>&1 | tail -4) > $O/fuzz.log; cat $O/fuzz.log
