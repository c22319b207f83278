# GPU session 5 of round 3: several waves per frame with ordered yielding -- parity, latency, where a frame's time goes.
O=gpurun_out/r03f; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_line.py tests/test_gpu_bench_step.py tests/test_gpu_golden_ref.py -m gpu -x -q 2>&1 | tail -8) > $O/line_tests.log
cat $O/line_tests.log
python - > $O/latency.log 2>&1 <<'PY'
import importlib, time, numpy as np
plp = importlib.import_module("structure-plp-slam_amd"); synth = importlib.import_module("structure-plp-slam_amd.synth")
frames = synth.replay(1234, 16, 480, 640)
for w in (1, 0, 2, 4, 6):
    lt = plp.LineFeatureTracker(); lt.set_grow_waves(w)
    lt.extract_LSD_LBD(frames[0])
    ts = []
    for i in range(48):
        t = time.perf_counter(); kl = lt.extract_LSD_LBD(frames[i % 16])[0]; ts.append(time.perf_counter() - t)
    lt.set_profiling(True)
    for i in range(8): lt.extract_LSD_LBD(frames[i % 16])
    ms, _ = lt.stage_times_ms(); lt.set_profiling(False)
    p = lt.grow_profile()
    extra = "" if w == 1 else f" | main: total {p['cycles_total']} wait {p['cycles_grow']} self {p['cycles_rect']} cycles; helper attempts {p['cycles_refine'] & 0xffffffff} give-ups {p['cycles_refine'] >> 32}; main grew {p['regions']}, took {p['pixels']}"
    print(f"grow_waves {w}: plp_line_extract median {1e3 * np.median(ts):.3f} ms; lsd_grow stage {ms['lsd_grow']:.3f} ms per frame; {len(kl)} key lines" + (f" | {p}" if w == 1 else extra), flush=True)
PY
cat $O/latency.log
(timeout 150 python tools/fuzz_gpu.py --only lines --seconds 90 --seed 53 2>&1 | tail -3) > $O/fuzz_lines.log; cat $O/fuzz_lines.log
