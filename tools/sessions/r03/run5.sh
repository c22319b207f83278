# GPU session 5 of round 3: several waves per frame with ordered yielding -- parity, latency, where a frame's time goes.
O=gpurun_out/r03h; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_line.py tests/test_gpu_bench_step.py tests/test_gpu_golden_ref.py tests/test_gpu_pack_rows.py -m gpu -x -q 2>&1 | tail -8) > $O/line_tests.log
cat $O/line_tests.log
python - > $O/latency.log 2>&1 <<'PY'
import importlib, time, numpy as np
plp = importlib.import_module("structure-plp-slam_amd"); synth = importlib.import_module("structure-plp-slam_amd.synth")
frames = synth.replay(1234, 16, 480, 640)
for w in (1, 0, 4, 6):
    lt = plp.LineFeatureTracker(); lt.set_grow_waves(w)
    lt.extract_LSD_LBD(frames[0])
    ts = []
    for i in range(48):
        t = time.perf_counter(); kl = lt.extract_LSD_LBD(frames[i % 16])[0]; ts.append(time.perf_counter() - t)
    lt.set_profiling(True)
    for i in range(8): lt.extract_LSD_LBD(frames[i % 16])
    ms, _ = lt.stage_times_ms(); lt.set_profiling(False)
    p = lt.grow_profile()
    extra = "" if w == 1 else f" | main: total {p['cycles_total']} wait {p['cycles_grow']} self {p['cycles_rect']} cycles; helper attempts {p['cycles_refine'] & 0xffffffff} give-ups {p['cycles_refine'] >> 32}; main grew {p['regions']}, took {p['pixels'] & 0xffffffff}, rejected {p['pixels'] >> 32}"
    print(f"grow_waves {w}: plp_line_extract median {1e3 * np.median(ts):.3f} ms; lsd_grow stage {ms['lsd_grow']:.3f} ms per frame; {len(kl)} key lines" + (f" | {p}" if w == 1 else extra), flush=True)
PY
cat $O/latency.log
(timeout 150 python tools/fuzz_gpu.py --only lines --seconds 60 --seed 54 2>&1 | tail -3) > $O/fuzz_lines.log; cat $O/fuzz_lines.log
P() { timeout 150 python bench.py --no-cpu-baseline --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 |', j['value'], j['ms_per_step'], 'pcie-inclusive', j.get('pcie_inclusive_value'), j.get('pcie_inclusive_ms_per_step'), j.get('pcie_bytes_per_step'), 'line latency', j['latency_ms_median_mean']['line_extract'])"; }
{ P "default"; GPU_MAX_HW_QUEUES=8 P "8 hw queues"; } > $O/pcie.log 2>&1; cat $O/pcie.log
