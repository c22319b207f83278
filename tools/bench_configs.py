#!/usr/bin/env python3
"""Supplementary measurements of the BASELINE.json configs that are NOT the bench.py line (those are parity-test cases
for the judge; the numbers here are extra evidence, recorded in profiles/):
   configs[0]  TUM mono, ORB only, K = 2000                      -> extract
   configs[2]  EuRoC stereo 752x480 x 2, K = 1000 per image       -> 2 x ORB, 2 x LSD+LBD, stereo::compute, LBD 1-NN L<->R
   configs[3]  KITTI mono 1241x376, K = 4000 (and 2000)           -> ORB, LSD+LBD, match_current_and_last_frames
   configs[4]  ICL-NUIM RGB-D 640x480 + plane instance masks      -> ORB, LSD+LBD, undistort/bearings/stereo-from-depth (plp_post_extract_device),
                                                                     match_current_and_last_frames[_line], plane colour vote (plp_color_vote_device)
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
    # ---- configs[4]: ICL-NUIM living_room RGB-D with plane segmentation masks (example/run_slam_planeSeg.cc; planar_mapping_module.cc:185-345)
    fr = frames(4, B, 480, 640, dev)
    K = 1000
    ex = plp.orb_extractor(K); lt = plp.LineFeatureTracker(); mtk = plp.matcher(0.9, True); mtl = plp.matcher(0.9, True)
    cap, k, d, c = orb_buffers(K)
    LB = line_buffers()
    lcap = 512
    cam = plp.camera_c()
    for name, v in (("fx", 481.2), ("fy", -480.0), ("cx", 319.5), ("cy", 239.5), ("focal_x_baseline", 40.0)):   # ICL-NUIM living room intrinsics
        setattr(cam, name, v)
    rng = np.random.default_rng(4)
    depth = torch.from_numpy(rng.uniform(0.5, 4.0, (32, 480, 640)).astype(np.float32)).to(dev).repeat((B + 31) // 32, 1, 1)[:B].contiguous()
    seg = np.zeros((32, 480, 640, 3), np.uint8)                      # six planar regions per frame + unlabelled background
    yy, xx = np.ogrid[:480, :640]
    for f in range(32):
        for _ in range(6):
            cy, cx, ry, rx = rng.integers(0, 480), rng.integers(0, 640), rng.integers(40, 200), rng.integers(40, 250)
            seg[f][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = rng.integers(1, 256, 3)
    d_seg = torch.from_numpy(seg).to(dev).repeat((B + 31) // 32, 1, 1, 1)[:B].contiguous()
    und = torch.empty_like(k); bear = torch.empty((B, cap, 3), dtype=torch.float64, device=dev)
    xr = torch.empty((B, cap), dtype=torch.float32, device=dev); dp = torch.empty((B, cap), dtype=torch.float32, device=dev)
    kld = torch.empty((B, lcap, 2), dtype=torch.float32, device=dev); klx = torch.empty((B, lcap, 2), dtype=torch.float32, device=dev)
    lab = torch.empty((B, cap), dtype=torch.int32, device=dev)
    m1 = torch.empty((B, cap), dtype=torch.int32, device=dev); n1 = torch.zeros(B, dtype=torch.int32, device=dev)
    m3 = torch.empty((B, lcap), dtype=torch.int32, device=dev); n3 = torch.zeros(B, dtype=torch.int32, device=dev)
    grid = plp.make_grid(640, 480); sf = ex.get_scale_factors(); sf_lsd = np.ones(1, np.float32)
    shift = torch.tensor([-3.0, 0.0], device=dev)
    L = plp.lib()

    def icl_step():
        sA.wait_stream(cur); sB.wait_stream(cur)
        ex.extract_batch(fr, k, d, c, stream=sA)
        lt.extract_batch(fr, *LB, stream=sB)
        sA.wait_stream(sB)
        with torch.cuda.stream(sA):
            st = sA.cuda_stream
            plp._check(L.plp_post_extract_device(mtk._h, C.byref(cam), k.data_ptr(), c.data_ptr(), cap, B, depth.data_ptr(), 480, 640, 640 * 4, 480 * 640 * 4, und.data_ptr(),
                                                 bear.data_ptr(), xr.data_ptr(), dp.data_ptr(), LB[0].data_ptr(), LB[3].data_ptr(), lcap, kld.data_ptr(), klx.data_ptr(), st))
            plp._check(L.plp_color_vote_device(mtk._h, d_seg.data_ptr(), 480, 640, 640 * 3, 480 * 640 * 3, und.data_ptr(), None, c.data_ptr(), cap, B, 1, lab.data_ptr(), st))
            uf = und.view(torch.float32).view(B, cap, 7)
            prev = torch.roll(uf, 1, 0); prevd = torch.roll(d, 1, 0); prevc = torch.roll(c, 1, 0).contiguous()
            q = dict(q_reproj=(prev[:, :, 0:2] + shift).contiguous(), q_level=prev.view(torch.int32)[:, :, 5].contiguous(), q_angle=prev[:, :, 3].contiguous(),
                     q_desc=prevd.contiguous(), q_counts=prevc)
            mtk.match_device(plp.MODE_LAST_FRAME, cap, cap, {**dict(t_kps=und, t_desc=d, t_counts=c), **q}, m1, n1, margin=20.0, direction=0, scale_factors=sf, grid=grid, B=B, stream=sA)
            klf = LB[0].view(torch.float32).view(B, lcap, 17)
            pk = torch.roll(klf, 1, 0); pl = torch.roll(LB[1], 1, 0); pc = torch.roll(LB[3], 1, 0).contiguous()
            ql = dict(q_reproj=(pk[:, :, 7:9] + shift).contiguous(), q_reproj2=(pk[:, :, 9:11] + shift).contiguous(), q_level=pk.view(torch.int32)[:, :, 2].contiguous(),
                      q_desc=pl.contiguous(), q_counts=pc, is_rgbd=0, num_levels_lsd=1)
            mtl.match_device(plp.MODE_LAST_FRAME_LINE, lcap, lcap, {**dict(t_kl=LB[0], t_desc=LB[1], t_counts=LB[3]), **ql}, m3, n3, margin=20.0, direction=0,
                             scale_factors=sf_lsd, B=B, stream=sA)
        cur.wait_stream(sA)
    sec = timeit(icl_step, a.steps)
    slot = torch.arange(cap, device=dev)[None, :]
    out.append({"config": "configs[4] ICL-NUIM RGB-D 640x480 + plane masks K=1000: ORB || LSD+LBD, post-extract (undistort, bearings, depth), plane colour vote, "
                          "match_current_and_last_frames + _line", "frames_per_s": round(B / sec, 1), "ms_per_batch": round(sec * 1e3, 3),
                "keypoints_mean": round(float(c.float().mean()), 1), "lines_mean": round(float(LB[3].float().mean()), 1),
                "keypoints_with_depth_mean": round(float(((dp > 0) & (slot < c[:, None])).float().sum(1).mean()), 1),
                "keypoints_on_a_plane_mean": round(float(((lab != 0) & (slot < c[:, None])).float().sum(1).mean()), 1),
                "matches_mean": [round(float(n1.float().mean()), 1), round(float(n3.float().mean()), 1)]})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
