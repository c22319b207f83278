"""single-frame line call, profiling OFF: median / mean wall time of N synchronous plp_line_extract calls (the way bench.py's latency pass times them)"""
import sys, time, numpy as np
sys.path[:0] = ["tests", "."]
from plp import plp, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
frames = synth.replay(1234, 16, 480, 640)
for order, name in ((plp.SEED_ORDER_LIBSTDCXX, "reference order"), (plp.SEED_ORDER_STABLE, "stable order")):
    lt = plp.LineFeatureTracker()
    lt.set_seed_order(order)
    for f in frames[:4]:
        lt.extract_LSD_LBD(f)
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); lt.extract_LSD_LBD(frames[i % len(frames)]); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print(f"{name}: {n} calls, median {np.median(ts):.3f} ms, mean {ts.mean():.3f} ms, min {ts.min():.3f} ms")
