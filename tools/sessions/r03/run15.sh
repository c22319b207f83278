# GPU session 15: group size / buffers / entries of the several-waves path, same box
O=gpurun_out/r03r; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
for v in g16b8e4 g16b4e8 g32b4e8 g32b2e16 g64b2e16 g64b4e8; do
cp build_exp/$v.so $L
V=$v python - <<'PY'
import importlib, time, os, numpy as np
plp = importlib.import_module("structure-plp-slam_amd"); synth = importlib.import_module("structure-plp-slam_amd.synth")
frames = synth.replay(1234, 16, 480, 640)
lt = plp.LineFeatureTracker()
lt.extract_LSD_LBD(frames[0])
ts = []
for i in range(64):
    t = time.perf_counter(); kl = lt.extract_LSD_LBD(frames[i % 16])[0]; ts.append(time.perf_counter() - t)
acc = np.zeros(11)
for i in range(16):
    lt.extract_LSD_LBD(frames[i]); p = lt.grow_profile()
    acc += [p['cycles_total'], p['cycles_grow'], p['cycles_rect'], p['cycles_refine'] & 0xffffffff, p['cycles_refine'] >> 32, p['regions'], p['pixels'] & 0xffffffff, p['pixels'] >> 32] + p['more'][:3]
acc /= 16
print(f"{os.environ['V']}: plp_line_extract median {1e3 * np.median(ts):.3f} ms | main: total {acc[0]/1e6:.2f} M wait {acc[1]/1e6:.2f} self {acc[2]/1e6:.2f} commit {acc[8]/1e6:.2f} publish {acc[9]/1e6:.2f} groups {acc[10]/1e6:.2f}; attempts {acc[3]:.0f} give-ups {acc[4]:.0f}; main grew {acc[5]:.0f}, took {acc[6]:.0f}, rejected {acc[7]:.0f}", flush=True)
PY
done > $O/sweep.log 2>&1
cp build_exp/.orig.so $L
cat $O/sweep.log
