# GPU session r05/7: seed sort with the scan pass staged through LDS (85 VGPRs, 8 spilled SGPRs) -- parity, the bench line with its extras (single-frame latencies, PCIe pass)
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
(timeout 500 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_line.py tests/test_gpu_bench_step.py tests/test_gpu_seed_sort_soak.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 8 $2 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'order', round(s['lsd_order'],2), 'grow', round(s['lsd_grow'],2), 'lat', j.get('latency_ms_median_mean'), 'pcie', j.get('pcie_inclusive_value'))" || (grep -i -m2 'fault\|PlpError\|error' $O/bench_$1.err | cut -c1-220)
}
B a1 ""
B a2 "--no-extras"
