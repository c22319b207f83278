"""one wave per frame (k_lsd_grow): cycles of growth / rectangle fits / refinement for a few replay frames"""
import sys, numpy as np
sys.path[:0] = ["tests", "."]
from plp import plp, synth
frames = synth.replay(1234, 6, 480, 640)
lt = plp.LineFeatureTracker()
lt.set_grow_waves(1)
lt.set_profiling(True)
for f in frames:
    lt.extract_LSD_LBD(f)
    p = lt.grow_profile()
    print({k: p[k] for k in ("cycles_total", "cycles_grow", "cycles_rect", "cycles_refine", "regions", "pixels")}, dict(zip(("regrow_cycles", "reduce_lane_cycles", "refined", "reduce_iterations", "reduce_points", "fitted"), p["more"])))
