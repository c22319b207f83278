"""The per-frame steps of BASELINE.json's other configurations, batched over B frames that are resident in HBM -- the code
`tools/bench_configs.py` times and `tests/test_gpu_config_steps.py` checks frame by frame against the oracle (the step of configs[1],
the bench line, is replay_step.tracker_step).

  stereo_step      configs[2]  EuRoC stereo (run_euroc_slam_with_line): the stereo frame constructor, data/frame.cc:267-360:
                               ORB left || ORB right || LSD+LBD left || LSD+LBD right (frame.cc:277-281, 351-358),
                               match::stereo::compute (match/stereo.cc:45-150) -> stereo_x_right / depths,
                               BinaryDescriptorMatcher::match left -> right on the LBD rows (frame.cc:505)
  mono_step        configs[3]  KITTI mono (run_kitti_slam_with_line): ORB || LSD+LBD, match_current_and_last_frames against frame b-1
                               (match/projection.cc:214-358, margin 20, orientation check, as frame_tracker.cc:66-87)
  rgbd_plane_step  configs[4]  ICL-NUIM RGB-D + plane masks (run_slam_planeSeg): ORB || LSD+LBD, undistort / bearings / stereo from depth
                               (frame.cc:110-128, 1169-1219), plane colour vote (planar_mapping_module.cc:185-345),
                               match_current_and_last_frames and _line against frame b-1
Frame b-1 of frame 0 is frame B-1 (the batch is treated as a ring, as the bench tool always did).  Host code: Python over the C ABI."""
import ctypes as C
import importlib

import numpy as np


def _torch():
    import torch
    return torch


def euroc_disparity(rows=480):
    """integer disparity per row of the synthetic stereo pair: d(y) = 8 + round(4 sin(y / 60))"""
    return (8 + np.rint(4 * np.sin(np.arange(rows) / 60.0))).astype(int)


def stereo_pair_from_wide(wide, cols=752):
    """wide: [B, rows, cols + 16] u8 on the device -> (left, right) [B, rows, cols]: the right image is the left one shifted by euroc_disparity"""
    torch = _torch()
    B, rows, _ = wide.shape
    disp = euroc_disparity(rows)
    right = torch.empty((B, rows, cols), dtype=torch.uint8, device=wide.device)
    for y in range(rows):
        right[:, y, :] = wide[:, y, disp[y]:disp[y] + cols]
    return wide[:, :, :cols].contiguous(), right


def icl_inputs(seed, n, rows=480, cols=640):
    """n synthetic depth maps (uniform 0.5 .. 4 m) and plane-instance colour masks (six elliptic regions over an unlabelled background)"""
    rng = np.random.default_rng(seed)
    depth = rng.uniform(0.5, 4.0, (n, rows, cols)).astype(np.float32)
    seg = np.zeros((n, rows, cols, 3), np.uint8)
    yy, xx = np.ogrid[:rows, :cols]
    for f in range(n):
        for _ in range(6):
            cy, cx, ry, rx = rng.integers(0, rows), rng.integers(0, cols), rng.integers(40, 200), rng.integers(40, 250)
            seg[f][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = rng.integers(1, 256, 3)
    return depth, seg


ICL_CAMERA = dict(fx=481.2, fy=-480.0, cx=319.5, cy=239.5, focal_x_baseline=40.0)     # ICL-NUIM living room (example/icl_nuim yaml)
EUROC_FXB, EUROC_TB = 435.2 * 0.11, 0.11                                               # focal_x_baseline, true_baseline


class _base:
    LCAP = 512

    def __init__(self, plp, B, K, device_index=0):
        torch = _torch()
        self.plp, self.B, self.K = plp, B, K
        self.torch = torch
        self.dev = dev = torch.device("cuda", device_index)
        self.cap = 2 * K + 64
        self.cur = torch.cuda.current_stream(dev)
        self.streams = [torch.cuda.Stream(dev) for _ in range(4)]
        self.L = plp.lib()

    def orb_buffers(self):
        t, B, cap = self.torch, self.B, self.cap
        return (t.empty((B, cap, 28), dtype=t.uint8, device=self.dev), t.empty((B, cap, 32), dtype=t.uint8, device=self.dev), t.zeros(B, dtype=t.int32, device=self.dev))

    def line_buffers(self):
        t, B, lc = self.torch, self.B, self.LCAP
        return (t.zeros((B, lc, 68), dtype=t.uint8, device=self.dev), t.zeros((B, lc, 32), dtype=t.uint8, device=self.dev),
                t.empty((B, lc, 3), dtype=t.float64, device=self.dev), t.zeros(B, dtype=t.int32, device=self.dev))


class stereo_step(_base):
    def __init__(self, plp, B, K=1000, device_index=0, fxb=EUROC_FXB, tb=EUROC_TB):
        super().__init__(plp, B, K, device_index)
        t = self.torch
        self.exl, self.exr = plp.orb_extractor(K, device=device_index), plp.orb_extractor(K, device=device_index)
        self.ltl, self.ltr = plp.LineFeatureTracker(device=device_index), plp.LineFeatureTracker(device=device_index)
        self.mt = plp.matcher(device=device_index)
        self.kl, self.dl, self.cl = self.orb_buffers(); self.kr, self.dr, self.cr = self.orb_buffers()
        self.LL, self.LR = self.line_buffers(), self.line_buffers()
        self.xr = t.empty((B, self.cap), dtype=t.float32, device=self.dev); self.dep = t.empty((B, self.cap), dtype=t.float32, device=self.dev)
        self.tidx = t.empty((B, self.LCAP), dtype=t.int32, device=self.dev); self.tdist = t.empty((B, self.LCAP), dtype=t.int32, device=self.dev)
        self.fxb, self.tb = fxb, tb

    def run(self, left, right):
        sA, sB, sC, sD = self.streams
        cur, plp, L = self.cur, self.plp, self.L
        for s in self.streams:
            s.wait_stream(cur)
        self.exl.extract_batch(left, self.kl, self.dl, self.cl, stream=sA)
        self.exr.extract_batch(right, self.kr, self.dr, self.cr, stream=sB)
        self.ltl.extract_batch(left, *self.LL, stream=sC)
        self.ltr.extract_batch(right, *self.LR, stream=sD)
        sA.wait_stream(sB); sC.wait_stream(sD)
        plp._check(L.plp_stereo_compute_batch_device(self.exl._h, self.exr._h, self.kl.data_ptr(), self.cl.data_ptr(), self.kr.data_ptr(), self.cr.data_ptr(),
                                                    self.dl.data_ptr(), self.dr.data_ptr(), self.cap, self.B, C.c_float(self.fxb), C.c_float(self.tb),
                                                    self.xr.data_ptr(), self.dep.data_ptr(), C.c_void_p(sA.cuda_stream)))
        plp._check(L.plp_lbd_match_1nn_device(self.mt._h, self.LL[1].data_ptr(), self.LL[3].data_ptr(), self.LCAP, self.LR[1].data_ptr(), self.LR[3].data_ptr(), self.LCAP,
                                              self.B, self.tidx.data_ptr(), self.tdist.data_ptr(), C.c_void_p(sC.cuda_stream)))
        cur.wait_stream(sA); cur.wait_stream(sC)

    def status(self):
        self.ltl.last_batch_status(); self.ltr.last_batch_status()


class mono_step(_base):
    def __init__(self, plp, B, K, rows, cols, device_index=0, shift=(3.0, 0.0)):
        super().__init__(plp, B, K, device_index)
        t = self.torch
        self.ex = plp.orb_extractor(K, device=device_index); self.lt = plp.LineFeatureTracker(device=device_index); self.mt = plp.matcher(0.9, True, device=device_index)
        self.k, self.d, self.c = self.orb_buffers()
        self.LB = self.line_buffers()
        self.m1 = t.empty((B, self.cap), dtype=t.int32, device=self.dev); self.n1 = t.zeros(B, dtype=t.int32, device=self.dev)
        self.grid = plp.make_grid(cols, rows)
        self.sf = self.ex.get_scale_factors()
        self.shift = shift
        self._shift = t.tensor(list(shift), dtype=t.float32, device=self.dev)

    def run(self, frames):
        t, plp = self.torch, self.plp
        sA, sB = self.streams[:2]
        cur, B, cap = self.cur, self.B, self.cap
        sA.wait_stream(cur); sB.wait_stream(cur)
        self.ex.extract_batch(frames, self.k, self.d, self.c, stream=sA)
        self.lt.extract_batch(frames, *self.LB, stream=sB)
        with t.cuda.stream(sA):
            kf = self.k.view(t.float32).view(B, cap, 7)
            prev = t.roll(kf, 1, 0); prevd = t.roll(self.d, 1, 0); prevc = t.roll(self.c, 1, 0).contiguous()
            q = dict(q_reproj=(prev[:, :, 0:2] + self._shift).contiguous(), q_level=prev.view(t.int32)[:, :, 5].contiguous(), q_angle=prev[:, :, 3].contiguous(),
                     q_desc=prevd.contiguous(), q_counts=prevc)
            self.mt.match_device(plp.MODE_LAST_FRAME, cap, cap, {**dict(t_kps=self.k, t_desc=self.d, t_counts=self.c), **q}, self.m1, self.n1, margin=20.0, direction=0,
                                 scale_factors=self.sf, grid=self.grid, B=B, stream=sA)
        cur.wait_stream(sA); cur.wait_stream(sB)

    def status(self):
        self.lt.last_batch_status()


class rgbd_plane_step(_base):
    def __init__(self, plp, B, K=1000, rows=480, cols=640, device_index=0, shift=(-3.0, 0.0), camera=None):
        super().__init__(plp, B, K, device_index)
        t = self.torch
        self.rows, self.cols = rows, cols
        self.ex = plp.orb_extractor(K, device=device_index); self.lt = plp.LineFeatureTracker(device=device_index)
        self.mtk = plp.matcher(0.9, True, device=device_index); self.mtl = plp.matcher(0.9, True, device=device_index)
        self.k, self.d, self.c = self.orb_buffers()
        self.LB = self.line_buffers()
        self.cam = plp.camera_c()
        self.cam_values = dict(camera or ICL_CAMERA)
        for name, v in self.cam_values.items():
            setattr(self.cam, name, v)
        cap, lcap, dev = self.cap, self.LCAP, self.dev
        self.und = t.empty_like(self.k); self.bear = t.empty((B, cap, 3), dtype=t.float64, device=dev)
        self.xr = t.empty((B, cap), dtype=t.float32, device=dev); self.dp = t.empty((B, cap), dtype=t.float32, device=dev)
        self.kld = t.empty((B, lcap, 2), dtype=t.float32, device=dev); self.klx = t.empty((B, lcap, 2), dtype=t.float32, device=dev)
        self.lab = t.empty((B, cap), dtype=t.int32, device=dev)
        self.m1 = t.empty((B, cap), dtype=t.int32, device=dev); self.n1 = t.zeros(B, dtype=t.int32, device=dev)
        self.m3 = t.empty((B, lcap), dtype=t.int32, device=dev); self.n3 = t.zeros(B, dtype=t.int32, device=dev)
        self.grid = plp.make_grid(cols, rows); self.sf = self.ex.get_scale_factors(); self.sf_lsd = np.ones(1, np.float32)
        self.shift = shift
        self._shift = t.tensor(list(shift), dtype=t.float32, device=dev)

    def run(self, frames, depth, seg):
        t, plp, L = self.torch, self.plp, self.L
        sA, sB = self.streams[:2]
        cur, B, cap, lcap, rows, cols = self.cur, self.B, self.cap, self.LCAP, self.rows, self.cols
        LB = self.LB
        sA.wait_stream(cur); sB.wait_stream(cur)
        self.ex.extract_batch(frames, self.k, self.d, self.c, stream=sA)
        self.lt.extract_batch(frames, *LB, stream=sB)
        sA.wait_stream(sB)
        with t.cuda.stream(sA):
            st = sA.cuda_stream
            self.kld.fill_(-1.0); self.klx.fill_(-1.0)       # the frame constructor's initial values; compute_stereo_from_depth overwrites the end points that have depth
            plp._check(L.plp_post_extract_device(self.mtk._h, C.byref(self.cam), self.k.data_ptr(), self.c.data_ptr(), cap, B, depth.data_ptr(), rows, cols, cols * 4,
                                                 rows * cols * 4, self.und.data_ptr(), self.bear.data_ptr(), self.xr.data_ptr(), self.dp.data_ptr(), LB[0].data_ptr(),
                                                 LB[3].data_ptr(), lcap, self.kld.data_ptr(), self.klx.data_ptr(), st))
            plp._check(L.plp_color_vote_device(self.mtk._h, seg.data_ptr(), rows, cols, cols * 3, rows * cols * 3, self.und.data_ptr(), None, self.c.data_ptr(), cap, B, 1,
                                               self.lab.data_ptr(), st))
            uf = self.und.view(t.float32).view(B, cap, 7)
            prev = t.roll(uf, 1, 0); prevd = t.roll(self.d, 1, 0); prevc = t.roll(self.c, 1, 0).contiguous()
            q = dict(q_reproj=(prev[:, :, 0:2] + self._shift).contiguous(), q_level=prev.view(t.int32)[:, :, 5].contiguous(), q_angle=prev[:, :, 3].contiguous(),
                     q_desc=prevd.contiguous(), q_counts=prevc)
            self.mtk.match_device(plp.MODE_LAST_FRAME, cap, cap, {**dict(t_kps=self.und, t_desc=self.d, t_counts=self.c), **q}, self.m1, self.n1, margin=20.0, direction=0,
                                  scale_factors=self.sf, grid=self.grid, B=B, stream=sA)
            klf = LB[0].view(t.float32).view(B, lcap, 17)
            pk = t.roll(klf, 1, 0); pl = t.roll(LB[1], 1, 0); pc = t.roll(LB[3], 1, 0).contiguous()
            ql = dict(q_reproj=(pk[:, :, 7:9] + self._shift).contiguous(), q_reproj2=(pk[:, :, 9:11] + self._shift).contiguous(), q_level=pk.view(t.int32)[:, :, 2].contiguous(),
                      q_desc=pl.contiguous(), q_counts=pc, is_rgbd=0, num_levels_lsd=1)
            self.mtl.match_device(plp.MODE_LAST_FRAME_LINE, lcap, lcap, {**dict(t_kl=LB[0], t_desc=LB[1], t_counts=LB[3]), **ql}, self.m3, self.n3, margin=20.0, direction=0,
                                  scale_factors=self.sf_lsd, B=B, stream=sA)
        cur.wait_stream(sA)

    def status(self):
        self.lt.last_batch_status()
