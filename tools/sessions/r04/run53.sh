# GPU session r04/53: reduce_region_radius as a rank pairing (k_lsd_grow): parity (line tests in both seed orders, also with a 64-word ring = the sequential fallback), fuzz, bench
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04u; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_line.py tests/test_gpu_seed_sort.py tests/test_gpu_bench_step.py tests/test_gpu_config_steps.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
(PLP_LSD_RING=64 timeout 300 python -m pytest tests/test_gpu_line.py tests/test_gpu_bench_step.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/pytest_ring64.log; cat $O/pytest_ring64.log
(timeout 200 python tools/fuzz_gpu.py --only lines --seconds 45 --seed 91 2>&1 | grep "lines:") > $O/fuzz.log; cat $O/fuzz.log
for k in 1 2; do
(timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --verify 64 2> $O/bench$k.err | tail -1) > $O/bench$k.json; python -c "import json; j=json.load(open('$O/bench$k.json')); print(j['value'], j['ms_per_step'], j['other_seed_order']['value'], j['verified_frames'], j['roofline']['stage_ms_per_batch']['lsd_grow'])" || tail -3 $O/bench$k.err
done
