#!/usr/bin/env python3
"""What do the kernels of a step lose to 2048 co-resident region-growing waves, resource by resource?  The ORB chain (and the line front
without region growing, and the matcher stage) is timed alone and beside a kernel that only HOLDS what the growers hold -- wave slots,
~10.6 KB of LDS per wave, ~160 VGPRs per wave -- in the four combinations.  GPU box:  python tools/experiments/corun_probe.py"""
import ctypes as C, importlib, os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "build_exp", "liblds_hog.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "experiments", "lds_hog.hip"), "-o", so], check=True)
hog = C.CDLL(so)
hog.hog_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_void_p]
plp = importlib.import_module("structure-plp-slam_amd"); synth = importlib.import_module("structure-plp-slam_amd.synth")
dev = torch.device("cuda", 0)
B, K = 2048, 1000
fr = torch.from_numpy(synth.replay(1234, 64, 480, 640)).to(dev).repeat(32, 1, 1).contiguous()
cap = 2 * K + 64
kps = torch.empty((B, cap, 28), dtype=torch.uint8, device=dev); desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev); cnt = torch.zeros(B, dtype=torch.int32, device=dev)
ex = plp.orb_extractor(K)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
sA, sH = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ex.extract_batch(fr, kps, desc, cnt, stream=sA); torch.cuda.synchronize()


def orb_ms(hog_cfg):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if hog_cfg:
        blocks, lds, regs = hog_cfg
        assert hog.hog_launch(sH.cuda_stream, blocks, lds, 60_000_000, regs, sink.data_ptr()) == 0      # ~25 ms at 2.4 GHz: longer than the chain
        time.sleep(0.002)
    e0.record(sA)
    ex.extract_batch(fr, kps, desc, cnt, stream=sA)
    e1.record(sA)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for name, cfg in (("alone", None), ("2048 sleeping waves, no LDS, few registers", (2048, 64, 0)), ("2048 sleeping waves, 10.6 KB LDS each", (2048, 10624, 0)),
                  ("2048 waves, ~160 VGPRs each, no LDS", (2048, 64, 1)), ("2048 waves, 10.6 KB LDS and ~160 VGPRs each", (2048, 10624, 1)),
                  ("1024 sleeping waves, 10.6 KB LDS each", (1024, 10624, 0)), ("2048 sleeping waves, 5.3 KB LDS each", (2048, 5312, 0))):
    ts = [orb_ms(cfg) for _ in range(4)]
    print(f"ORB chain of 2048 frames, {name}: {np.median(ts):.2f} ms (runs: {', '.join(f'{t:.2f}' for t in ts)})", flush=True)
