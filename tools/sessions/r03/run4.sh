# GPU session 4 of round 3: where the several-waves-per-frame region growing spends a frame; scheduling matrix of the overlapped step.
O=gpurun_out/r03d; mkdir -p $O
python - > $O/mw_diag.log 2>&1 <<'PY'
import importlib, time, numpy as np
plp = importlib.import_module("structure-plp-slam_amd"); synth = importlib.import_module("structure-plp-slam_amd.synth")
frames = synth.replay(1234, 4, 480, 640)
for w in (1, 2, 4, 8):
    lt = plp.LineFeatureTracker(); lt.set_grow_waves(w)
    for f in frames[:2]:
        lt.extract_LSD_LBD(f)
        p = lt.grow_profile(); gs = lt.debug_read(7) if hasattr(lt, "DBG_GROW_STATS") else None
        print("grow_waves", w, {k: v for k, v in p.items()}, flush=True)
PY
cat $O/mw_diag.log
L=structure-plp-slam_amd/libplp_front.so
B() { timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 $2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 |', j['value'], j['ms_per_step'], s['lsd_grow'], s['match_4x'])"; }
{
B "default (sobel first, 2 line ctx)"
PLP_LINE_SIDE_STREAM=1 B "side stream (round 2 arrangement)"
PLP_LINE_SIDE_STREAM=1 GPU_MAX_HW_QUEUES=8 B "side stream + 8 hw queues"
GPU_MAX_HW_QUEUES=8 B "8 hw queues"
PLP_BENCH_LINE_PRIO=-1 B "line streams high priority"
PLP_BENCH_LINE_SPLIT=1 B "1 line ctx"
PLP_BENCH_LINE_SPLIT=4 GPU_MAX_HW_QUEUES=8 B "4 line ctx, 8 hw queues"
PLP_BENCH_LINE_SPLIT=4 B "4 line ctx"
PLP_BENCH_NBUF=3 B "3 feature sets"
PLP_BENCH_NBUF=3 PLP_BENCH_LINE_PRIO=-1 B "3 feature sets + line prio"
B "default again"
cp $L build_exp/.orig.so
} > $O/sched.log 2>&1
cat $O/sched.log
