"""Post-extract per-key-point step (undistort_keypoints, convert_keypoints_to_bearings, compute_stereo_from_depth):
HIP path vs oracle, bit-exact (f64 arithmetic with +, -, *, /, sqrt only)."""
import numpy as np
import pytest

import oracle_lib as O
from plp import plp, synth

pytestmark = pytest.mark.gpu

# TUM RGB-D freiburg1 / freiburg2 (distorted) and freiburg3 (rectified) intrinsics as in the reference's example configs
CAMS = {
    "fr1": (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314, 40.0),
    "fr2": (520.908620, 521.007327, 325.141442, 249.701764, 0.231222, -0.784899, -0.003257, -0.000105, 0.917205, 40.0),
    "fr3": (535.4, 539.2, 320.1, 247.6, 0.0, 0.0, 0.0, 0.0, 0.0, 40.0),
    "kitti": (718.856, 718.856, 607.1928, 185.2157, 0.0, 0.0, 0.0, 0.0, 0.0, 386.1448),
    "wide": (300.0, 305.0, 322.0, 241.0, -0.35, 0.12, 0.001, -0.0007, -0.02, 30.0),
}


def cam_struct(v):
    c = plp.camera_c()
    for name, val in zip(("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "focal_x_baseline"), v):
        setattr(c, name, float(val))
    return c


@pytest.mark.parametrize("name", list(CAMS))
def test_post_extract_matches_oracle(name):
    cam = CAMS[name]
    rng = np.random.default_rng(5)
    img = synth.replay(3, 1, 480, 640)[0]
    kps, _ = O.OrbOracle(1500).extract(img)
    # a few exact corners / borders in addition to the extractor's key points
    extra = np.zeros(6, O.KP_DTYPE)
    extra["x"] = [0, 639, 0, 639, 320.25, 19.5]; extra["y"] = [0, 0, 479, 479, 240.75, 460.0]; extra["octave"] = [0, 1, 2, 3, 4, 5]
    kps = np.concatenate([kps, extra])
    depth = rng.uniform(0.3, 8.0, (480, 640)).astype(np.float32)
    depth[rng.uniform(size=depth.shape) < 0.2] = 0.0            # missing depth
    depth[rng.uniform(size=depth.shape) < 0.02] = -1.0
    kl = np.zeros(40, O.KL_DTYPE)
    kl["startPointX"] = rng.uniform(0, 639, 40); kl["startPointY"] = rng.uniform(0, 479, 40)
    kl["endPointX"] = rng.uniform(0, 639, 40); kl["endPointY"] = rng.uniform(0, 479, 40)
    pre_d = np.full((40, 2), -1, np.float32); pre_x = np.full((40, 2), -1, np.float32)
    want = O.post_extract(cam, kps, depth, kl, pre_d, pre_x)
    got = plp.matcher().post_extract(cam_struct(cam), kps, depth, kl, pre_d, pre_x)
    for key in want:
        assert np.array_equal(got[key], want[key]), key
    assert (want["stereo_x_right"] >= 0).sum() > 500 and (want["depths"] < 0).sum() > 100
    if name in ("fr3", "kitti"):     # no distortion: the fixed-point iteration returns the input up to the final float rounding
        assert np.abs(want["undist_keypts"]["x"] - kps["x"]).max() < 1e-3
    else:
        assert np.abs(want["undist_keypts"]["x"] - kps["x"]).max() > 0.5
    nb = np.linalg.norm(want["bearings"], axis=1)
    assert np.abs(nb - 1).max() < 1e-12


def test_post_extract_without_depth_and_empty():
    cam = CAMS["fr1"]
    kps = np.zeros(3, O.KP_DTYPE); kps["x"] = [10, 300, 630]; kps["y"] = [20, 240, 470]
    want = O.post_extract(cam, kps)
    got = plp.matcher().post_extract(cam_struct(cam), kps)
    assert set(got) == {"undist_keypts", "bearings"}
    assert np.array_equal(got["undist_keypts"], want["undist_keypts"]) and np.array_equal(got["bearings"], want["bearings"])
    assert len(plp.matcher().post_extract(cam_struct(cam), np.zeros(0, O.KP_DTYPE))["undist_keypts"]) == 0


def test_post_extract_batched_device():
    import torch
    cam = CAMS["fr2"]
    frames = synth.replay(9, 3, 480, 640)
    ex = plp.orb_extractor(1000)
    B, cap = len(frames), 2064
    dev = torch.device("cuda", 0)
    d_fr = torch.from_numpy(frames).to(dev)
    d_k = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev); d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_c = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch(d_fr, d_k, d_d, d_c)
    rng = np.random.default_rng(2)
    depth = rng.uniform(0.5, 6.0, (B, 480, 640)).astype(np.float32)
    d_depth = torch.from_numpy(depth).to(dev)
    d_u = torch.zeros_like(d_k); d_b = torch.zeros((B, cap, 3), dtype=torch.float64, device=dev)
    d_x = torch.zeros((B, cap), dtype=torch.float32, device=dev); d_z = torch.zeros((B, cap), dtype=torch.float32, device=dev)
    mt = plp.matcher()
    c = cam_struct(cam)
    import ctypes as C
    torch.cuda.synchronize()   # the tensor fills above ran on torch's stream, the call below uses the context's
    plp._check(plp.lib().plp_post_extract_device(mt._h, C.byref(c), d_k.data_ptr(), d_c.data_ptr(), cap, B, d_depth.data_ptr(), 480, 640, 640 * 4, 480 * 640 * 4,
                                                d_u.data_ptr(), d_b.data_ptr(), d_x.data_ptr(), d_z.data_ptr(), None, None, 0, None, None, None))
    torch.cuda.synchronize()
    cnt = d_c.cpu().numpy()
    for b in range(B):
        n = int(cnt[b])
        kps = d_k[b].cpu().numpy().view(plp.KP_DTYPE).reshape(cap)[:n]
        want = O.post_extract(cam, kps, depth[b])
        assert np.array_equal(d_u[b].cpu().numpy().view(plp.KP_DTYPE).reshape(cap)[:n], want["undist_keypts"])
        assert np.array_equal(d_b[b].cpu().numpy()[:n], want["bearings"])
        assert np.array_equal(d_x[b].cpu().numpy()[:n], want["stereo_x_right"]) and np.array_equal(d_z[b].cpu().numpy()[:n], want["depths"])


def test_landmark_descriptor_selection():
    rng = np.random.default_rng(21)
    sizes = [0, 1, 2, 3, 4, 5, 7, 8, 15, 33, 64, 65, 100, 200] + rng.integers(1, 40, 300).tolist()
    descs, offsets = [], [0]
    for n in sizes:
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        d = np.tile(base, (n, 1))
        for i in range(n):                      # observations of one landmark: noisy copies (ties in the medians are common)
            for b in rng.choice(256, int(rng.integers(0, 30)), replace=False):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        if n >= 4 and rng.uniform() < 0.3:
            d[1] = d[0]                          # exact duplicates
        descs.append(d); offsets.append(offsets[-1] + n)
    alld = np.concatenate(descs) if descs else np.zeros((0, 32), np.uint8)
    got = plp.matcher().landmark_descriptors(alld, offsets)
    want = [O.landmark_descriptor(d) if len(d) else -1 for d in descs]
    assert got.tolist() == want


def test_input_side_grayscale_and_true_depth():
    import ctypes as C
    import torch
    rng = np.random.default_rng(33)
    dev = torch.device("cuda", 0)
    mt = plp.matcher()
    L = plp.lib()
    cur = None   # the context's own stream: synchronise torch's fills first (torch's default stream is the NULL stream = 'use the context stream')
    for (rows, cols, ch, bgr) in [(480, 640, 3, 0), (480, 640, 3, 1), (376, 1241, 4, 1), (61, 77, 4, 0), (5, 3, 3, 1)]:
        B = 2
        src = rng.integers(0, 256, (B, rows, cols, ch), dtype=np.uint8)
        src[0, 0, 0] = 255; src[0, 0, min(1, cols - 1)] = 0
        d_src = torch.from_numpy(src).to(dev); d_g = torch.zeros((B, rows, cols), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        plp._check(L.plp_convert_to_grayscale_device(mt._h, d_src.data_ptr(), rows, cols, cols * ch, rows * cols * ch, ch, bgr, B, d_g.data_ptr(), cols, rows * cols, cur))
        torch.cuda.synchronize()
        want = np.zeros((B, rows, cols), np.uint8)
        for b in range(B):
            O._call("oracle_convert_to_grayscale", [np.ascontiguousarray(src[b]), rows, cols, ch, bgr, want[b]])
        assert np.array_equal(d_g.cpu().numpy(), want)
    for factor in (5000.0, 5208.0, 1000.0, 1.0):
        raw = rng.integers(0, 65536, (2, 120, 161), dtype=np.uint16); raw[0, 0, :3] = [0, 65535, 5000]
        d_raw = torch.from_numpy(raw.view(np.int16)).to(dev); d_f = torch.zeros((2, 120, 161), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        plp._check(L.plp_convert_to_true_depth_device(mt._h, d_raw.data_ptr(), 1, 120, 161, 161 * 2, 120 * 161 * 2, C.c_double(factor), 2, d_f.data_ptr(), 161 * 4, 120 * 161 * 4, cur))
        torch.cuda.synchronize()
        want = np.zeros(raw.size, np.float32)
        fn = O.lib().oracle_convert_to_true_depth_u16; fn.restype = None
        fn(C.c_void_p(raw.ctypes.data), C.c_size_t(raw.size), C.c_double(factor), C.c_void_p(want.ctypes.data))
        got = d_f.cpu().numpy().ravel()
        assert np.array_equal(got, want), (factor, raw.ravel()[got != want][:4], got[got != want][:4], want[got != want][:4])
        f32 = rng.uniform(0, 40000, (2, 120, 161)).astype(np.float32)
        d_in = torch.from_numpy(f32).to(dev)
        torch.cuda.synchronize()
        plp._check(L.plp_convert_to_true_depth_device(mt._h, d_in.data_ptr(), 0, 120, 161, 161 * 4, 120 * 161 * 4, C.c_double(factor), 2, d_f.data_ptr(), 161 * 4, 120 * 161 * 4, cur))
        torch.cuda.synchronize()
        fn2 = O.lib().oracle_convert_to_true_depth_f32; fn2.restype = None
        fn2(C.c_void_p(f32.ctypes.data), C.c_size_t(f32.size), C.c_double(factor), C.c_void_p(want.ctypes.data))
        assert np.array_equal(d_f.cpu().numpy().ravel(), want)


def test_plane_colour_vote():
    import ctypes as C
    import torch
    rng = np.random.default_rng(44)
    dev = torch.device("cuda", 0)
    rows, cols, B, cap = 480, 640, 2, 1500
    mask = np.zeros((B, rows, cols, 3), np.uint8)
    for b in range(B):                                   # six random convex colour regions + unlabelled background
        for k in range(6):
            cy, cx, ry, rx = rng.integers(0, rows), rng.integers(0, cols), rng.integers(30, 200), rng.integers(30, 250)
            yy, xx = np.ogrid[:rows, :cols]
            mask[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = rng.integers(1, 256, 3)
    und = np.zeros((B, cap), O.KP_DTYPE)
    und["x"] = rng.uniform(-5, cols + 5, (B, cap)).astype(np.float32); und["y"] = rng.uniform(-5, rows + 5, (B, cap)).astype(np.float32)
    und["x"][:, :8] = [0, 0.5, 1.2, cols - 1, cols - 0.5, cols, 320, 1]; und["y"][:, :8] = [0, 1, 0.7, rows - 1, rows - 0.2, 100, rows, 1]
    valid = (rng.uniform(size=(B, cap)) > 0.1).astype(np.uint8)
    counts = np.array([cap, cap - 77], np.int32)
    d_mask = torch.from_numpy(mask).to(dev); d_und = torch.from_numpy(und.view(np.uint8).reshape(B, cap, 28)).to(dev)
    d_valid = torch.from_numpy(valid).to(dev); d_cnt = torch.from_numpy(counts).to(dev)
    mt = plp.matcher()
    for check in (1, 0):
        d_lab = torch.full((B, cap), -5, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        plp._check(plp.lib().plp_color_vote_device(mt._h, d_mask.data_ptr(), rows, cols, cols * 3, rows * cols * 3, d_und.data_ptr(), d_valid.data_ptr(),
                                                  d_cnt.data_ptr(), cap, B, check, d_lab.data_ptr(), None))
        torch.cuda.synchronize()
        got = d_lab.cpu().numpy()
        for b in range(B):
            want = np.zeros(cap, np.int32)
            n = int(counts[b])
            O._call("oracle_color_vote", [np.ascontiguousarray(mask[b]), rows, cols, ("z", cols * 3), np.ascontiguousarray(und[b]), np.ascontiguousarray(valid[b]),
                                          n, check, want])
            assert np.array_equal(got[b], want), (check, b)
            assert (want != 0).sum() > 100
