"""Stage times of the single-frame line call (plp_line_extract with profiling on: HIP events around every stage), median over the replay frames, both seed orders."""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")
frames = synth.replay(1234, 64, 480, 640)
for order, name in ((plp.SEED_ORDER_LIBSTDCXX, "std::sort order"), (plp.SEED_ORDER_STABLE, "stable order")):
    lt = plp.LineFeatureTracker(); lt.set_seed_order(order)
    for f in frames[:4]:
        lt.extract_LSD_LBD(f)
    rows = []
    for f in frames:
        lt.set_profiling(True); lt.extract_LSD_LBD(f); ms, n = lt.stage_times_ms(); lt.set_profiling(False)
        rows.append([ms[k] for k in lt.STAGES])
    med = np.median(np.array(rows), 0)
    print(name, {k: round(float(v), 3) for k, v in zip(lt.STAGES, med)})
