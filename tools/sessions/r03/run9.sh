# GPU session 9 of round 3: helper policy A/B of the several-waves-per-frame region growing
O=gpurun_out/r03q; mkdir -p $O
for pol in 0; do
unset PLP_LSD_MW_POLICY
timeout 200 python -m pytest tests/test_gpu_line.py tests/test_gpu_golden_ref.py -m gpu -x -q 2>&1 | tail -2
PLP_LSD_MW_POLICY=0 python - <<'PY'
import importlib, time, os, numpy as np
plp = importlib.import_module("structure-plp-slam_amd"); synth = importlib.import_module("structure-plp-slam_amd.synth")
frames = synth.replay(1234, 16, 480, 640)
for w in (0,):
    lt = plp.LineFeatureTracker(); lt.set_grow_waves(w)
    lt.extract_LSD_LBD(frames[0])
    ts = []
    for i in range(48):
        t = time.perf_counter(); kl = lt.extract_LSD_LBD(frames[i % 16])[0]; ts.append(time.perf_counter() - t)
    acc = np.zeros(11)
    for i in range(16):
        lt.extract_LSD_LBD(frames[i]); p = lt.grow_profile()
        acc += [p['cycles_total'], p['cycles_grow'], p['cycles_rect'], p['cycles_refine'] & 0xffffffff, p['cycles_refine'] >> 32, p['regions'], p['pixels'] & 0xffffffff, p['pixels'] >> 32] + p['more'][:3]
    acc /= 16
    print(f"policy {os.environ['PLP_LSD_MW_POLICY']} grow_waves {w}: plp_line_extract median {1e3 * np.median(ts):.3f} ms | main (mean of 16 frames): total {acc[0]:.0f} wait {acc[1]:.0f} self {acc[2]:.0f} cycles; helper attempts {acc[3]:.0f} give-ups {acc[4]:.0f}; main grew {acc[5]:.0f}, took {acc[6]:.0f}, rejected {acc[7]:.0f}; cycles taking results {acc[8]:.0f}, publishing own {acc[9]:.0f}, group set-up {acc[10]:.0f}", flush=True)
PY
done > $O/policy.log 2>&1
cat $O/policy.log

(timeout 100 python tools/fuzz_gpu.py --only lines --seconds 50 --seed 56 2>&1 | tail -3) > $O/fuzz_lines.log; cat $O/fuzz_lines.log
