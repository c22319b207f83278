#!/usr/bin/env python3
"""Supplementary measurements of the BASELINE.json configs that are NOT the bench.py line (those are parity-test cases
for the judge; the numbers here are extra evidence, recorded in profiles/):
   configs[0]  TUM mono, ORB only, K = 2000                      -> extract
   configs[2]  EuRoC stereo 752x480 x 2, K = 1000 per image       -> 2 x ORB, 2 x LSD+LBD, stereo::compute, LBD 1-NN L<->R
   configs[3]  KITTI mono 1241x376, K = 4000 (and 2000)           -> ORB, LSD+LBD, match_current_and_last_frames
Inputs resident in HBM, synthetic replay, one GPU.   python tools/bench_configs.py [--batch 1024] [--steps 4]"""
import argparse, ctypes as C, importlib, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")
replay = importlib.import_module("structure-plp-slam_amd.replay")


def frames(seed, B, rows, cols, dev):
    uniq = min(B, 32)
    f = torch.from_numpy(synth.replay(seed, uniq, rows, cols)).to(dev)
    return f.repeat((B + uniq - 1) // uniq, 1, 1)[:B].contiguous()


def timeit(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = a.batch
    out = []
    sA, sB, sC, sD = (torch.cuda.Stream(dev) for _ in range(4))
    cur = torch.cuda.current_stream(dev)

    def orb_buffers(K):
        cap = 2 * K + 64
        return cap, torch.empty((B, cap, 28), dtype=torch.uint8, device=dev), torch.empty((B, cap, 32), dtype=torch.uint8, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)

    def line_buffers():
        return (torch.empty((B, 512, 68), dtype=torch.uint8, device=dev), torch.empty((B, 512, 32), dtype=torch.uint8, device=dev),
                torch.empty((B, 512, 3), dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))

    # ---- configs[0]: ORB only, 640x480, K = 2000
    fr = frames(1, B, 480, 640, dev)
    ex = plp.orb_extractor(2000)
    cap, k, d, c = orb_buffers(2000)
    sec = timeit(lambda: ex.extract_batch(fr, k, d, c, stream=cur), a.steps)
    out.append({"config": "configs[0] TUM mono ORB-only 640x480 K=2000", "frames_per_s": round(B / sec, 1), "ms_per_batch": round(sec * 1e3, 3),
                "keypoints_mean": round(float(c.float().mean()), 1)})
    del ex, fr

    # ---- configs[2]: EuRoC stereo 752x480 x 2, K = 1000
    left = frames(2, B, 480, 752 + 16, dev)
    # right image = left shifted by an integer disparity field d(y) = 8 + round(4 sin(y / 60))
    disp = (8 + np.rint(4 * np.sin(np.arange(480) / 60.0))).astype(int)
    right = torch.empty((B, 480, 752), dtype=torch.uint8, device=dev)
    for y in range(480):
        right[:, y, :] = left[:, y, disp[y]:disp[y] + 752]
    left = left[:, :, :752].contiguous()
    exl, exr = plp.orb_extractor(1000), plp.orb_extractor(1000)
    ltl, ltr = plp.LineFeatureTracker(), plp.LineFeatureTracker()
    mt = plp.matcher()
    cap, kl_, dl, cl = orb_buffers(1000); _, kr_, dr, cr = orb_buffers(1000)
    LL, LR = line_buffers(), line_buffers()
    xr = torch.empty((B, cap), dtype=torch.float32, device=dev); dep = torch.empty((B, cap), dtype=torch.float32, device=dev)
    tidx = torch.empty((B, 512), dtype=torch.int32, device=dev); tdist = torch.empty((B, 512), dtype=torch.int32, device=dev)
    L = plp.lib()

    def stereo_step():
        for s in (sA, sB, sC, sD):
            s.wait_stream(cur)
        exl.extract_batch(left, kl_, dl, cl, stream=sA)
        exr.extract_batch(right, kr_, dr, cr, stream=sB)
        ltl.extract_batch(left, *LL, stream=sC)
        ltr.extract_batch(right, *LR, stream=sD)
        sA.wait_stream(sB); sC.wait_stream(sD)
        plp._check(L.plp_stereo_compute_batch_device(exl._h, exr._h, kl_.data_ptr(), cl.data_ptr(), kr_.data_ptr(), cr.data_ptr(), dl.data_ptr(), dr.data_ptr(),
                                                    cap, B, C.c_float(435.2 * 0.11), C.c_float(0.11), xr.data_ptr(), dep.data_ptr(), C.c_void_p(sA.cuda_stream)))
        plp._check(L.plp_lbd_match_1nn_device(mt._h, LL[1].data_ptr(), LL[3].data_ptr(), 512, LR[1].data_ptr(), LR[3].data_ptr(), 512, B,
                                              tidx.data_ptr(), tdist.data_ptr(), C.c_void_p(sC.cuda_stream)))
        cur.wait_stream(sA); cur.wait_stream(sC)
    sec = timeit(stereo_step, a.steps)
    torch.cuda.synchronize()
    out.append({"config": "configs[2] EuRoC stereo 752x480 x2 K=1000: 2x ORB, 2x LSD+LBD, stereo::compute, LBD 1-NN L<->R", "stereo_frames_per_s": round(B / sec, 1),
                "ms_per_batch": round(sec * 1e3, 3), "keypoints_mean": round(float(cl.float().mean()), 1), "lines_mean": round(float(LL[3].float().mean()), 1),
                "stereo_matches_mean": round(float((xr >= 0).float().sum(1).mean()), 1),
                "line_matches_mean": round(float(((tidx >= 0) & (torch.arange(tidx.shape[1], device=tidx.device)[None, :] < LL[3][:, None])).float().sum(1).mean()), 1)})
    del exl, exr, ltl, ltr, left, right

    # ---- configs[3]: KITTI mono 1241x376
    fr = frames(3, B, 376, 1241, dev)
    for K in (4000, 2000):
        ex = plp.orb_extractor(K); lt = plp.LineFeatureTracker(); mtk = plp.matcher(0.9, True)
        cap, k, d, c = orb_buffers(K)
        LB = line_buffers()
        m1 = torch.empty((B, cap), dtype=torch.int32, device=dev); n1 = torch.zeros(B, dtype=torch.int32, device=dev)
        grid = plp.make_grid(1241, 376)
        sf = ex.get_scale_factors()
        shift = torch.tensor([3.0, 0.0], device=dev)

        def kitti_step():
            sA.wait_stream(cur); sB.wait_stream(cur)
            ex.extract_batch(fr, k, d, c, stream=sA)
            lt.extract_batch(fr, *LB, stream=sB)
            with torch.cuda.stream(sA):
                kf = k.view(torch.float32).view(B, cap, 7)
                prev = torch.roll(kf, 1, 0); prevd = torch.roll(d, 1, 0); prevc = torch.roll(c, 1, 0).contiguous()
                q = dict(q_reproj=(prev[:, :, 0:2] + shift).contiguous(), q_level=prev.view(torch.int32)[:, :, 5].contiguous(), q_angle=prev[:, :, 3].contiguous(),
                         q_desc=prevd.contiguous(), q_counts=prevc)
                mtk.match_device(plp.MODE_LAST_FRAME, cap, cap, {**dict(t_kps=k, t_desc=d, t_counts=c), **q}, m1, n1, margin=20.0, direction=0,
                                 scale_factors=sf, grid=grid, B=B, stream=sA)
            cur.wait_stream(sA); cur.wait_stream(sB)
        sec = timeit(kitti_step, a.steps)
        out.append({"config": f"configs[3] KITTI mono 1241x376 K={K}: ORB || LSD+LBD, match_current_and_last_frames", "frames_per_s": round(B / sec, 1),
                    "ms_per_batch": round(sec * 1e3, 3), "keypoints_mean": round(float(c.float().mean()), 1), "lines_mean": round(float(LB[3].float().mean()), 1),
                    "matches_mean": round(float(n1.float().mean()), 1)})
        del ex, lt, mtk
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
