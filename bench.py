#!/usr/bin/env python3
"""bench.py — frames/s of the front-end hot path on synthetic 640x480 replay (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of `--batch` frames that are already resident
in HBM.  Frames shard across ranks with no data-path collective (weak scaling: every rank owns its
own batch); the timed region is bracketed by a barrier + synchronize and the max over ranks is used.
Rank 0 prints ONE JSON line.  The `roofline` object is the dominant kernel's algorithmic bytes per
launch divided by its HIP-event duration (measured live on the launch stream), `cpu_baseline` is the
oracle restatement timed on this box's host cores on a bounded sample of the same frames.
"""
import argparse
import importlib
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def level_pixels(rows, cols, sf, n_levels):
    s, out = np.float32(1.0), []
    for l in range(n_levels):
        if l:
            s = np.float32(np.float32(sf) * s)
        out.append((int(round(cols * 1.0 / float(s))) if l else cols) * (int(round(rows * 1.0 / float(s))) if l else rows))
    return out


def algorithmic_bytes(rows, cols, n_levels, mean_kp, mean_cand):
    """SURVEY.md §8(d) per-frame figures, split per kernel (bytes the algorithm inherently moves)."""
    px = level_pixels(rows, cols, 1.2, n_levels)
    P = sum(px)
    return {
        "pyramid": (P - px[-1]) + (P - px[0]),          # read every source level once, write every resized level
        "fast_cells": P + 4 * mean_cand,                 # read each level once, write packed candidates
        "blur7": 2 * P,                                  # read + write every level
        "quadtree": 3 * 4 * mean_cand + 4 * mean_kp,     # candidates in, keys/indices once, selection out
        "orient_rbrief": mean_kp * (749 + 512 + 28 + 32),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per rank per step")
    ap.add_argument("--keypoints", type=int, default=1000, help="Feature.max_num_keypoints (TUM RGB-D YAML: 1000)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    plp = importlib.import_module("structure-plp-slam_amd")
    synth = importlib.import_module("structure-plp-slam_amd.synth")

    B, K = args.batch, args.keypoints
    # every rank replays its own contiguous block of the sequence (frame f -> rank f // B)
    uniq = min(B, 64)
    frames_np = synth.replay(1234 + rank, uniq, args.rows, args.cols)
    d_frames = torch.from_numpy(frames_np).to(dev)
    if uniq < B:
        d_frames = d_frames.repeat((B + uniq - 1) // uniq, 1, 1)[:B].contiguous()
    cap = 2 * K + 64
    d_kps = torch.empty((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    ex = plp.orb_extractor(K, device=local_rank)
    stream = torch.cuda.current_stream(dev)

    def step():
        ex.extract_batch(d_frames, d_kps, d_desc, d_cnt, stream=stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ex.last_batch_status()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fps = world * B * args.steps / elapsed

    # ---- per-kernel HIP-event timing (separate, synchronous pass) -> roofline of the dominant kernel
    cnt = d_cnt.cpu().numpy()
    mean_kp = float(cnt.mean())
    n_prof = 5
    ex.set_profiling(True)
    for _ in range(n_prof):
        step()
    stage_ms, nb = ex.stage_times_ms()
    ex.set_profiling(False)
    mean_cand = float(np.mean([len(ex.debug_read(ex.DBG_CANDIDATES, l, 0)) for l in range(ex.get_num_scale_levels())]) * ex.get_num_scale_levels())
    per_frame = algorithmic_bytes(args.rows, args.cols, ex.get_num_scale_levels(), mean_kp, mean_cand)
    kern = {k: v for k, v in stage_ms.items() if k in per_frame}
    dominant = max(kern, key=kern.get)
    launches = 7 if dominant == "pyramid" else 1
    dom_bytes = per_frame[dominant] * B
    achieved = dom_bytes / (kern[dominant] * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "launch_ms": round(kern[dominant] / launches, 4), "bytes_per_launch": int(dom_bytes / launches),
                "stage_ms_per_batch": {k: round(v, 4) for k, v in stage_ms.items()},
                "stage_GBps": {k: round(per_frame[k] * B / (kern[k] * 1e-3) / 1e9, 1) for k in kern if kern[k] > 0}}

    out = {
        "metric": "frames/sec ORB+LSD extract+match, 640x480 TUM-RGBD, 1/2/4/8 GPU",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"TUM-RGBD-shaped replay {args.cols}x{args.rows}, ORB extract only (K={K}, 8 levels, 1.2); "
                               "LSD/LBD + matchers not in the timed region yet",
                   "frames_per_rank_per_step": B, "keypoints_mean": round(mean_kp, 1), "sharding": "frame blocks per rank, no collective"},
        "roofline": roofline,
    }
    if rank == 0 and not args.no_cpu_baseline:
        import ctypes as C
        import oracle_lib as O
        cores = os.cpu_count() or 1
        n_cpu = min(uniq, 64)
        sample = np.ascontiguousarray(frames_np[:n_cpu])
        reps = max(1, int(8 * cores / n_cpu))     # ~10-20 s of CPU work at ~25 ms/frame/core
        tot = C.c_long()
        tiled = np.ascontiguousarray(np.tile(sample, (reps, 1, 1)))
        sec = O.lib().oracle_orb_time_frames(tiled.ctypes.data_as(C.c_void_p), len(tiled), args.rows, args.cols, K, cores, C.byref(tot))
        sec1 = O.lib().oracle_orb_time_frames(sample.ctypes.data_as(C.c_void_p), min(n_cpu, 16), args.rows, args.cols, K, 1, C.byref(tot))
        out["cpu_baseline"] = {"value": round(len(tiled) / sec, 1), "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": f"{len(tiled)} frames of the same replay, frame-parallel on {cores} threads, ORB extract only (oracle restatement, -O2 strict FP)",
                               "single_thread_fps": round(min(n_cpu, 16) / sec1, 2)}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
