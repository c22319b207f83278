"""Replay of an RGB-D sequence in the TUM layout through the device path, batch by batch ("a
real-data replay driver once datasets are available").  Mirrors the front half of example/run_tum_rgbd_slam.cc +
tracking_module::track_RGBD_image + the data::frame constructor, for B frames at a time:

    rgb.txt / depth.txt association (io_formats.tum_rgbd_sequence)   host
    PNG decode (io_formats.read_image)                               host
    convert_to_grayscale, convert_to_true_depth                      device (plp_convert_*_device)
    ORB extract, LSD + LBD extract                                   device (plp_orb/line_extract_batch_device)
    undistort_keypoints, bearings, compute_stereo_from_depth         device (plp_post_extract_device)
    [compute_bow]                                                    device (plp_bow_transform_device, when a vocabulary is given)
    match_current_and_last_frames of frame b against frame b - 1     device (PLP_MATCH_MODE_LAST_FRAME)

The matcher step stands in for the tracker: without a pose estimate the "reprojection" of a last-frame feature is its own
undistorted position (a static-camera motion model, no stereo gate), with a correspondingly larger margin.  Everything between the upload
of the decoded frames and the per-frame counts stays in HBM.

    python -m structure-plp-slam_amd.replay_driver <sequence dir> [--batch 64] [--fx .. --fy .. --cx .. --cy ..]
"""
import argparse
import ctypes as C
import importlib

import numpy as np

plp = importlib.import_module(__package__ or "structure-plp-slam_amd")
io = importlib.import_module((__package__ or "structure-plp-slam_amd") + ".io_formats")

# TUM RGB-D freiburg3 defaults of the reference's example/tum_rgbd/TUM_RGBD_rgbd_3.yaml
FR3 = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0, focal_x_baseline=40.0, depthmap_factor=5000.0,
           color_order="RGB", max_num_keypts=1000)


class rgbd_replay:
    def __init__(self, camera=None, vocabulary=None, device=0, match_margin=30.0):
        import torch
        self.cfg = dict(FR3, **(camera or {}))
        self.dev = torch.device("cuda", device)
        self.ex = plp.orb_extractor(self.cfg["max_num_keypts"], device=device)
        self.lt = plp.LineFeatureTracker(device=device)
        self.mt = plp.matcher(0.9, True, device=device)
        self.vocab = vocabulary
        self.margin = match_margin
        self.cam = plp.camera_c()
        for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "focal_x_baseline"):
            setattr(self.cam, k, float(self.cfg[k]))
        self.prev = None          # undistorted key points / descriptors / count of the last frame of the previous batch

    def process(self, color_frames, depth_frames):
        """color_frames: uint8 [B, rows, cols, 3] as cv::imread returns them (BGR) or [B, rows, cols] gray; depth_frames: uint16
        [B, rows, cols].  Returns per-frame numpy arrays: n_keypts, n_keylines, n_depth (key points with a valid depth),
        n_matches (against the previous frame; -1 for the very first frame), and the device tensors of the batch."""
        import torch
        L = plp.lib()
        h = self.mt._h
        color = torch.from_numpy(np.ascontiguousarray(color_frames)).to(self.dev)
        depth_raw = torch.from_numpy(np.ascontiguousarray(depth_frames).view(np.int16)).to(self.dev)
        B, rows, cols = depth_raw.shape
        st = torch.cuda.current_stream(self.dev).cuda_stream
        if color.dim() == 4:
            gray = torch.empty((B, rows, cols), dtype=torch.uint8, device=self.dev)
            # cv::imread hands out BGR; the yaml's Camera.color_order says what the sensor delivered (image_converter.cc:33-75)
            plp._check(L.plp_convert_to_grayscale_device(h, color.data_ptr(), rows, cols, cols * 3, rows * cols * 3, 3,
                                                         0 if self.cfg["color_order"] == "RGB" else 1, B, gray.data_ptr(), cols, rows * cols, st))
        else:
            gray = color
        depth = torch.empty((B, rows, cols), dtype=torch.float32, device=self.dev)
        plp._check(L.plp_convert_to_true_depth_device(h, depth_raw.data_ptr(), 1, rows, cols, cols * 2, rows * cols * 2,
                                                      C.c_double(self.cfg["depthmap_factor"]), B, depth.data_ptr(), cols * 4, rows * cols * 4, st))
        cap, lcap = 2 * self.cfg["max_num_keypts"] + 64, 512
        kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=self.dev); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=self.dev)
        cnt = torch.zeros(B, dtype=torch.int32, device=self.dev)
        kl = torch.zeros((B, lcap, 68), dtype=torch.uint8, device=self.dev); lbd = torch.zeros((B, lcap, 32), dtype=torch.uint8, device=self.dev)
        fn = torch.zeros((B, lcap, 3), dtype=torch.float64, device=self.dev); lcnt = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.ex.extract_batch(gray, kps, desc, cnt)
        self.lt.extract_batch(gray, kl, lbd, fn, lcnt)
        und = torch.zeros_like(kps); bear = torch.zeros((B, cap, 3), dtype=torch.float64, device=self.dev)
        xr = torch.full((B, cap), -1.0, dtype=torch.float32, device=self.dev); dp = torch.full((B, cap), -1.0, dtype=torch.float32, device=self.dev)
        kld = torch.full((B, lcap, 2), -1.0, dtype=torch.float32, device=self.dev); klx = torch.full((B, lcap, 2), -1.0, dtype=torch.float32, device=self.dev)
        plp._check(L.plp_post_extract_device(h, C.byref(self.cam), kps.data_ptr(), cnt.data_ptr(), cap, B, depth.data_ptr(), rows, cols, cols * 4,
                                             rows * cols * 4, und.data_ptr(), bear.data_ptr(), xr.data_ptr(), dp.data_ptr(), kl.data_ptr(),
                                             lcnt.data_ptr(), lcap, kld.data_ptr(), klx.data_ptr(), st))
        out = dict(gray=gray, depth=depth, keypts=kps, undist_keypts=und, descriptors=desc, counts=cnt, bearings=bear, stereo_x_right=xr, depths=dp,
                   keylines=kl, lbd=lbd, line_counts=lcnt)
        if self.vocab is not None:
            out["bow"] = self.vocab.transform_device(desc, cnt, 4)
        # frame b against frame b - 1 (the first frame of the batch against the last frame of the previous batch)
        undf = und.view(torch.float32).view(B, cap, 7)
        first = self.prev is None
        pk, pd, pc = (undf[:1], desc[:1], cnt[:1]) if first else self.prev
        qk = torch.cat([pk, undf[:-1]], 0); qd = torch.cat([pd, desc[:-1]], 0); qc = torch.cat([pc, cnt[:-1]], 0).contiguous()
        m = torch.empty((B, cap), dtype=torch.int32, device=self.dev); nm = torch.zeros(B, dtype=torch.int32, device=self.dev)
        q = dict(q_reproj=qk[:, :, 0:2].contiguous(), q_level=qk.view(torch.int32)[:, :, 5].contiguous(), q_angle=qk[:, :, 3].contiguous(),
                 q_desc=qd.contiguous(), q_counts=qc)
        self.mt.match_device(plp.MODE_LAST_FRAME, cap, cap, {**dict(t_kps=und, t_desc=desc, t_counts=cnt), **q}, m, nm, margin=self.margin,
                             direction=0, scale_factors=self.ex.get_scale_factors(), grid=plp.make_grid(cols, rows), B=B,
                             stream=torch.cuda.current_stream(self.dev))
        self.prev = (undf[-1:].clone(), desc[-1:].clone(), cnt[-1:].clone())
        torch.cuda.synchronize(self.dev)
        n_matches = nm.cpu().numpy().copy()
        if first:
            n_matches[0] = -1
        slot = torch.arange(cap, device=self.dev)[None, :]
        n_depth = ((dp > 0) & (slot < cnt[:, None])).sum(1).cpu().numpy()
        out["matches"] = m
        return dict(n_keypts=cnt.cpu().numpy(), n_keylines=lcnt.cpu().numpy(), n_depth=n_depth, n_matches=n_matches), out


def replay_sequence(seq_dir, batch=64, camera=None, vocabulary=None, device=0, max_frames=None):
    frames = io.tum_rgbd_sequence(seq_dir).get_frames()
    if max_frames:
        frames = frames[:max_frames]
    rp = rgbd_replay(camera, vocabulary, device)
    stats = []
    for i in range(0, len(frames), batch):
        chunk = frames[i:i + batch]
        color = np.stack([io.read_image(f.rgb_img_path) for f in chunk])
        depth = np.stack([io.read_image(f.depth_img_path) for f in chunk])
        s, _ = rp.process(color, depth)
        stats.append(s)
    return {k: np.concatenate([s[k] for s in stats]) for k in stats[0]} if stats else {}, [f.timestamp for f in frames]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sequence")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--max-frames", type=int, default=None)
    for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3"):
        ap.add_argument("--" + k, type=float, default=None)
    a = ap.parse_args()
    cam = {k: getattr(a, k) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3") if getattr(a, k) is not None}
    stats, ts = replay_sequence(a.sequence, a.batch, cam, max_frames=a.max_frames)
    for i, t in enumerate(ts):
        print(f"{t:.6f} keypts {stats['n_keypts'][i]} keylines {stats['n_keylines'][i]} with_depth {stats['n_depth'][i]} matches {stats['n_matches'][i]}")


if __name__ == "__main__":
    main()
