# GPU session 3 of round 3: parity of the several-waves-per-frame region growing, its latency, and the timeline of the overlapped step.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
(timeout 300 python -m pytest tests/test_gpu_line.py tests/test_gpu_bench_step.py tests/test_gpu_golden_ref.py -m gpu -x -q 2>&1 | tail -15) > $O/line_tests.log
cat $O/line_tests.log
python - > $O/latency.log 2>&1 <<'PY'
import importlib, time, numpy as np
plp = importlib.import_module("structure-plp-slam_amd"); synth = importlib.import_module("structure-plp-slam_amd.synth")
frames = synth.replay(1234, 16, 480, 640)
for w in (1, 0, 2, 3, 4, 6, 8):
    lt = plp.LineFeatureTracker(); lt.set_grow_waves(w)
    lt.extract_LSD_LBD(frames[0])
    ts = []
    for i in range(48):
        t = time.perf_counter(); kl = lt.extract_LSD_LBD(frames[i % 16])[0]; ts.append(time.perf_counter() - t)
    lt.set_profiling(True)
    for i in range(8): lt.extract_LSD_LBD(frames[i % 16])
    ms, _ = lt.stage_times_ms(); lt.set_profiling(False)
    print(f"grow_waves {w}: plp_line_extract median {1e3 * np.median(ts):.3f} ms, mean {1e3 * np.mean(ts):.3f}; lsd_grow stage {ms['lsd_grow']:.3f} ms per frame; {len(kl)} key lines")
PY
cat $O/latency.log
(timeout 120 python tools/fuzz_gpu.py --only lines --seconds 60 --seed 51 2>&1 | tail -4) > $O/fuzz_lines.log; cat $O/fuzz_lines.log
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --verify 0 > $O/kt.log 2>&1
cd $R
python tools/rocpd_timeline.py $O/kt/kt_results.db 3 > $O/r03c_step_timeline.md 2> $O/timeline.err
python tools/rocpd_summary.py $O/kt/kt_results.db "r03c (bench.py --steps 4 --warmup 2)" > $O/r03c_full_kernel_stats.md
rm -rf $O/kt
head -60 $O/r03c_step_timeline.md; cat $O/timeline.err | tail -5
