"""Diagnostic: k_lsd_grow's stage time against the number of frames in flight (one wave per frame).  Separates what one wave
costs alone (L2-resident planes) from what the batch costs (2048 planes = 2.5 GB behind 2048 dependent gather chains).
Usage (GPU box): python tools/grow_scaling.py [--frames 1 64 256 ...]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, nargs="*", default=[1, 64, 256, 512, 1024, 2048, 3072])
    ap.add_argument("--same-frame", action="store_true", help="every slot of the batch holds the same image (same trip counts everywhere)")
    ap.add_argument("--transpose", action="store_true", help="transposed frames (480 wide, 640 high): horizontal structures become vertical")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    base = synth.replay(7, 64, 480, 640)
    if a.transpose:
        base = np.ascontiguousarray(base.transpose(0, 2, 1))
    lt = plp.LineFeatureTracker()
    for B in a.frames:
        idx = np.zeros(B, np.int64) if a.same_frame else np.arange(B) % len(base)
        d = torch.from_numpy(np.ascontiguousarray(base[idx])).to(dev)
        kl = torch.empty((B, 256, 68), dtype=torch.uint8, device=dev); lbd = torch.empty((B, 256, 32), dtype=torch.uint8, device=dev)
        fn = torch.empty((B, 256, 3), dtype=torch.float64, device=dev); cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        lt.extract_batch(d, kl, lbd, fn, cnt); torch.cuda.synchronize()
        lt.set_profiling(True)
        for _ in range(3):
            lt.extract_batch(d, kl, lbd, fn, cnt)
        ms, _ = lt.stage_times_ms(); lt.set_profiling(False)
        print(f"B={B:5d}  lsd_grow {ms['lsd_grow']:8.3f} ms  order {ms['lsd_order']:.3f}  gradient_bins {ms['lsd_gradient_bins']:.3f}  lbd {ms['lbd']:.3f}", flush=True)


if __name__ == "__main__":
    main()
