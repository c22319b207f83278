#!/usr/bin/env python3
"""What the exact seed order (PLP_SEED_ORDER_LIBSTDCXX) costs beside the stable one: per-stage HIP-event times of the line front-end for
a batch of replay frames in both modes, and the single-frame latency of plp_line_extract.  python tools/seed_order_cost.py [--batch 2048]"""
import argparse, importlib, json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    plp = importlib.import_module("structure-plp-slam_amd")
    synth = importlib.import_module("structure-plp-slam_amd.synth")
    dev = torch.device("cuda:0")
    uniq = min(a.batch, 64)
    frames = synth.replay(1234, uniq, a.rows, a.cols)
    d = torch.from_numpy(frames).to(dev)
    B = a.batch
    if uniq < B:
        d = d.repeat((B + uniq - 1) // uniq, 1, 1)[:B].contiguous()
    cap = 512
    d_kl = torch.zeros((B, cap, 68), dtype=torch.uint8, device=dev); d_lbd = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_fn = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev); d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    out = {"batch": B, "shape": [a.rows, a.cols]}
    for name, order in (("stable", plp.SEED_ORDER_STABLE), ("libstdcxx", plp.SEED_ORDER_LIBSTDCXX)):
        lt = plp.LineFeatureTracker()
        lt.set_seed_order(order)
        lt.extract_batch(d, d_kl, d_lbd, d_fn, d_cnt); torch.cuda.synchronize()
        lt.last_batch_status()
        lt.set_profiling(True)
        for _ in range(a.reps):
            lt.extract_batch(d, d_kl, d_lbd, d_fn, d_cnt)
        ms, n = lt.stage_times_ms()
        lt.set_profiling(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            lt.extract_batch(d, d_kl, d_lbd, d_fn, d_cnt)
        torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) / a.reps * 1e3
        lat = []
        l1 = plp.LineFeatureTracker(); l1.set_seed_order(order)
        for i in range(40):
            t0 = time.perf_counter(); l1.extract_LSD_LBD(frames[i % uniq]); lat.append((time.perf_counter() - t0) * 1e3)
        out[name] = {"stage_ms": {k: round(v, 4) for k, v in ms.items()}, "batch_ms_unprofiled": round(whole, 3),
                     "single_frame_ms_median": round(float(np.median(lat[8:])), 3), "mean_lines": float(d_cnt.float().mean().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
