"""single-frame line path: stage times and the several-waves grower's own counters (frame 0 of the replay)"""
import sys, numpy as np
sys.path[:0] = ["tests", "."]
from plp import plp, synth
frames = synth.replay(1234, 8, 480, 640)
for order in (plp.SEED_ORDER_LIBSTDCXX, plp.SEED_ORDER_STABLE):
    lt = plp.LineFeatureTracker()
    lt.set_seed_order(order)
    for f in frames[:3]:
        lt.extract_LSD_LBD(f)
    lt.set_profiling(True)
    for f in frames:
        lt.extract_LSD_LBD(f)
    ms, n = lt.stage_times_ms()
    print("order", order, {k: round(v, 3) for k, v in ms.items()}, n)
    p = lt.grow_profile()
    m = p["more"]
    print("  mw: cycles total/wait/self", p["cycles_total"], p["cycles_grow"], p["cycles_rect"], "helper attempts|giveups", p["cycles_refine"] & 0xffffffff, p["cycles_refine"] >> 32,
          "self-grown", p["regions"], "spec ok|bad", p["pixels"] & 0xffffffff, p["pixels"] >> 32, "commit/publish/group cycles", m[:3])
