"""Diagnostic: per-frame statistics and phase cycles of k_lsd_grow (frame 0 of a batch).  Usage (GPU box): python tools/grow_stats.py"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")


def main():
    dev = torch.device("cuda", 0)
    base = synth.replay(7, 8, 480, 640)
    lt = plp.LineFeatureTracker()
    B = len(base)
    d = torch.from_numpy(np.ascontiguousarray(base)).to(dev)
    kl = torch.empty((B, 256, 68), dtype=torch.uint8, device=dev); lbd = torch.empty((B, 256, 32), dtype=torch.uint8, device=dev)
    fn = torch.empty((B, 256, 3), dtype=torch.float64, device=dev); cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    lt.set_profiling(True)
    for _ in range(2):
        lt.extract_batch(d, kl, lbd, fn, cnt)
    torch.cuda.synchronize()
    print("profile frame 0:", lt.grow_profile())
    for f in range(4):
        print("frame", f, "regions, pixels, exact tests, speculative commits:", lt.debug_read(lt.DBG_GROW_STATS, f).tolist(), "seeds", len(lt.debug_read(lt.DBG_ORDER, f)))


if __name__ == "__main__":
    main()
