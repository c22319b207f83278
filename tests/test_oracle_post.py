"""Oracle of the post-extract step (oracle/post_oracle.cpp): properties that hold for any correct cv::undistortPoints /
bearing / depth-lookup restatement.  (The reference has no test for this step: parity unpinned, see the file header.)"""
import numpy as np
import pytest

import oracle_lib as O

FR1 = (517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314, 40.0)
FR3 = (535.4, 539.2, 320.1, 247.6, 0.0, 0.0, 0.0, 0.0, 0.0, 40.0)


def distort(cam, xu, yu):
    """forward Brown-Conrady model with the FLOAT-rounded parameters the reference hands to OpenCV"""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = [float(np.float32(v)) for v in cam[:9]]
    x = (xu - cx) / fx; y = (yu - cy) / fy
    r2 = x * x + y * y
    cd = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return xd * fx + cx, yd * fy + cy


def test_undistort_inverts_the_distortion_model():
    rng = np.random.default_rng(0)
    xu = rng.uniform(40, 600, 2000); yu = rng.uniform(40, 440, 2000)
    xd, yd = distort(FR1, xu, yu)
    kps = np.zeros(2000, O.KP_DTYPE); kps["x"] = xd; kps["y"] = yd; kps["angle"] = 12.5; kps["size"] = 31; kps["octave"] = 3; kps["response"] = 77
    out = O.post_extract(FR1, kps)
    u = out["undist_keypts"]
    assert np.abs(u["x"] - xu).max() < 2e-3 and np.abs(u["y"] - yu).max() < 2e-3      # EPS 1e-6 px + float storage
    assert (u["angle"] == 12.5).all() and (u["size"] == 31).all() and (u["octave"] == 3).all()
    assert (u["response"] == 0).all() and (u["class_id"] == -1).all()                   # resize() defaults, not copied (:153-160)


def test_zero_distortion_is_the_identity_up_to_float_rounding_and_bearings_are_unit():
    rng = np.random.default_rng(1)
    kps = np.zeros(500, O.KP_DTYPE); kps["x"] = rng.uniform(0, 639, 500).astype(np.float32); kps["y"] = rng.uniform(0, 479, 500).astype(np.float32)
    out = O.post_extract(FR3, kps)
    assert np.abs(out["undist_keypts"]["x"] - kps["x"]).max() < 1e-4 and np.abs(out["undist_keypts"]["y"] - kps["y"]).max() < 1e-4
    b = out["bearings"]
    assert np.abs(np.linalg.norm(b, axis=1) - 1).max() < 1e-15 * 10
    # bearing of the principal point is the optical axis
    pp = np.zeros(1, O.KP_DTYPE); pp["x"] = 320.1; pp["y"] = 247.6
    bp = O.post_extract(FR3, pp)["bearings"][0]
    assert abs(bp[2] - 1) < 1e-9 and abs(bp[0]) < 1e-6 and abs(bp[1]) < 1e-6


def test_stereo_from_depth_lookup_and_invalid_values():
    kps = np.zeros(4, O.KP_DTYPE); kps["x"] = [10.9, 20.2, 30.5, 5.0]; kps["y"] = [7.9, 8.1, 9.5, 3.0]
    depth = np.zeros((16, 40), np.float32)
    depth[7, 10] = 2.0      # (int)7.9 = 7, (int)10.9 = 10: truncation, not rounding
    depth[8, 20] = -1.0     # invalid
    depth[9, 30] = 0.0      # missing
    depth[3, 5] = 4.0
    out = O.post_extract(FR3, kps, depth)
    und = out["undist_keypts"]
    assert out["depths"].tolist() == [2.0, -1.0, -1.0, 4.0]
    assert out["stereo_x_right"][1] == -1 and out["stereo_x_right"][2] == -1
    assert out["stereo_x_right"][0] == np.float32(float(und["x"][0]) - 40.0 / 2.0) and out["stereo_x_right"][3] == np.float32(float(und["x"][3]) - 40.0 / 4.0)
    kl = np.zeros(2, O.KL_DTYPE); kl["startPointX"] = [10.9, 20.2]; kl["startPointY"] = [7.9, 8.1]; kl["endPointX"] = [5.0, 5.0]; kl["endPointY"] = [3.0, 3.0]
    o2 = O.post_extract(FR3, kps, depth, kl, np.full((2, 2), -7, np.float32), np.full((2, 2), -7, np.float32))
    assert o2["kl_depths"].tolist() == [[2.0, 4.0], [-7.0, -7.0]]                       # a negative end-point depth skips the line
    assert o2["kl_x_right"][0].tolist() == [np.float32(10.9) - np.float32(20.0), np.float32(5.0 - 10.0)]


def test_grayscale_known_values_and_depth_scale():
    import ctypes as C
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 77]]], np.uint8)
    for bgr in (0, 1):
        out = np.zeros((1, 6), np.uint8)
        O._call("oracle_convert_to_grayscale", [np.ascontiguousarray(px), 1, 6, 3, bgr, out])
        r, g, b = (px[0, :, 2], px[0, :, 1], px[0, :, 0]) if bgr else (px[0, :, 0], px[0, :, 1], px[0, :, 2])
        want = (r.astype(np.int64) * 4899 + g.astype(np.int64) * 9617 + b.astype(np.int64) * 1868 + 8192) >> 14
        assert out[0].tolist() == want.tolist()
        assert out[0, 0] == 255 and out[0, 1] == 0                    # the three weights sum to 1 << 14
        assert abs(int(out[0, 5]) - round(0.299 * int(r[5]) + 0.587 * int(g[5]) + 0.114 * int(b[5]))) <= 1
    raw = np.array([0, 1, 5000, 65535], np.uint16); dst = np.zeros(4, np.float32)
    fn = O.lib().oracle_convert_to_true_depth_u16; fn.restype = None
    fn(C.c_void_p(raw.ctypes.data), C.c_size_t(4), C.c_double(5000.0), C.c_void_p(dst.ctypes.data))
    assert dst.tolist() == [0.0, float(np.float32(1) * np.float32(1.0 / 5000.0)), float(np.float32(5000) * np.float32(1.0 / 5000.0)),
                            float(np.float32(65535) * np.float32(1.0 / 5000.0))]


def test_landmark_descriptor_and_colour_vote_small_cases():
    d = np.zeros((5, 32), np.uint8)
    d[1, 0] = 0x01; d[2, 0] = 0x03; d[3, 0] = 0x07; d[4, 0] = 0xff      # distances to row 0: 0 1 2 3 8
    # medians (rank 2 of 5): row0 {0,1,2,3,8}->2, row1 {1,0,1,2,7}->1, row2 {2,1,0,1,6}->1, row3 {3,2,1,0,5}->2, row4 {8,7,6,5,0}->6: first minimum = row 1
    assert O.landmark_descriptor(d) == 1
    assert O.landmark_descriptor(d[:1]) == 0 and O.landmark_descriptor(d[:2]) == 0      # rank (unsigned)(0.5 * 1) = 0: every median is 0
    mask = np.zeros((6, 8, 3), np.uint8); mask[1:5, 1:6] = (10, 20, 30); mask[2, 3] = (10, 20, 31)
    kps = np.zeros(5, O.KP_DTYPE)
    kps["x"] = [1.9, 4.2, 2.5, 7.0, 1.0]; kps["y"] = [1.2, 3.9, 2.0, 5.0, 4.9]      # (1,1) (3,4) (2,2) background (4,1)
    lab = np.zeros(5, np.int32)
    O._call("oracle_color_vote", [mask, 6, 8, ("z", 24), kps, np.ones(5, np.uint8), 5, 0, lab])
    h = 10 + (20 << 8) + (30 << 16)
    assert lab.tolist() == [h, h, h, 0, h]
    O._call("oracle_color_vote", [mask, 6, 8, ("z", 24), kps, np.ones(5, np.uint8), 5, 1, lab])
    # (1,1): neighbours in row 0 / column 0 are never looked at, the others are its own colour; (3,4) and (2,2) touch the odd pixel (2,3);
    # (4,1): its lower neighbours (5, .) are background
    assert lab.tolist() == [h, 0, 0, 0, 0]


def test_rectify_map_and_remap_known_cases():
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    yy, xx = np.mgrid[0:48, 0:64].astype(np.float32)
    # identity and whole-pixel shifts copy pixels; what falls outside is the constant border 0
    assert np.array_equal(O.remap_linear(img, xx, yy), img)
    sh = O.remap_linear(img, xx + 3, yy - 2)
    assert np.array_equal(sh[2:, :61], img[:46, 3:]) and not sh[:2].any() and not sh[:, 61:].any()
    # half-pixel: exact bilinear mean, rounded half up by (sum + 2^14) >> 15
    hp = O.remap_linear(img, xx + 0.5, yy)
    a, b = img[:, :-1].astype(np.int32), img[:, 1:].astype(np.int32)
    assert np.array_equal(hp[:, :-1], ((a + b + 1) >> 1).astype(np.uint8))
    assert np.array_equal(hp[:, -1], ((img[:, -1].astype(np.int32) + 1) >> 1).astype(np.uint8))     # right neighbour is the border
    # the 1/32 grid: 0.49 px rounds to 16/32, closed-form weights 32 (32 - ax)(32 - ay) reproduce the table path
    mx = (xx + rng.uniform(-1.5, 1.5, xx.shape)).astype(np.float32); my = (yy + rng.uniform(-1.5, 1.5, yy.shape)).astype(np.float32)
    got = O.remap_linear(img, mx, my)
    fx = np.rint(mx.astype(np.float64) * 32).astype(np.int64); fy = np.rint(my.astype(np.float64) * 32).astype(np.int64)
    sx, sy, ax, ay = fx >> 5, fy >> 5, fx & 31, fy & 31
    pad = np.zeros((48 + 8, 64 + 8), np.int64); pad[4:52, 4:68] = img
    g = lambda dy, dx: pad[np.clip(sy + dy + 4, 0, 55), np.clip(sx + dx + 4, 0, 71)]
    want = (g(0, 0) * 32 * (32 - ax) * (32 - ay) + g(0, 1) * 32 * ax * (32 - ay) + g(1, 0) * 32 * (32 - ax) * ay + g(1, 1) * 32 * ax * ay + (1 << 14)) >> 15
    assert np.array_equal(got, want.astype(np.uint8))
    # no distortion, R = I, K = float(K_rect): the map is the pixel grid
    cam = dict(fx=435.25, fy=435.25, cx=367.5, cy=252.25)
    K = [435.25, 0, 367.5, 0, 435.25, 252.25, 0, 0, 1]
    mx, my = O.rectify_map(K, [], np.eye(3).ravel(), cam, 40, 60)
    gy, gx = np.mgrid[0:40, 0:60]
    assert np.abs(mx - gx).max() < 1e-3 and np.abs(my - gy).max() < 1e-3
    # EuRoC: the rectified principal point looks along R^T e_z, and the distortion model is applied (forward check of one pixel)
    E = O.EUROC
    mx, my = O.rectify_map(E["StereoRectifier.K_left"], E["StereoRectifier.D_left"], E["StereoRectifier.R_left"], E["camera"], 480, 752)
    Kl = np.array(E["StereoRectifier.K_left"]).reshape(3, 3); R = np.array(E["StereoRectifier.R_left"]).reshape(3, 3)
    k1, k2, p1, p2, k3 = E["StereoRectifier.D_left"]
    c = E["camera"]
    for (i, j) in [(0, 0), (479, 751), (252, 367), (100, 600)]:
        ray = R.T @ np.array([(j - np.float32(c["cx"])) / np.float32(c["fx"]), (i - np.float32(c["cy"])) / np.float32(c["fy"]), 1.0])
        x, y = ray[0] / ray[2], ray[1] / ray[2]
        r2 = x * x + y * y
        kr = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
        xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        assert abs(mx[i, j] - (Kl[0, 0] * xd + Kl[0, 2])) < 2e-3 and abs(my[i, j] - (Kl[1, 1] * yd + Kl[1, 2])) < 2e-3
    with pytest.raises(ValueError):
        O.rectify_map(K, [], np.zeros(9), cam, 4, 4)


def test_fisheye_rectify_map_follows_the_equidistant_model():
    E = O.TUM_VI
    c = E["camera"]
    mx, my = O.fisheye_rectify_map(E["StereoRectifier.K_left"], E["StereoRectifier.D_left"], E["StereoRectifier.R_left"], c, 512, 512)
    Kl = np.array(E["StereoRectifier.K_left"]).reshape(3, 3); R = np.array(E["StereoRectifier.R_left"]).reshape(3, 3)
    k = E["StereoRectifier.D_left"]
    for (i, j) in [(0, 0), (511, 511), (256, 240), (100, 400), (500, 20)]:
        ray = R.T @ np.array([(j - np.float32(c["cx"])) / np.float32(c["fx"]), (i - np.float32(c["cy"])) / np.float32(c["fy"]), 1.0])
        x, y = ray[0] / ray[2], ray[1] / ray[2]
        r = np.hypot(x, y); th = np.arctan(r)
        thd = th * (1 + k[0] * th ** 2 + k[1] * th ** 4 + k[2] * th ** 6 + k[3] * th ** 8)
        s = thd / r if r > 0 else 1.0
        assert abs(mx[i, j] - (Kl[0, 0] * x * s + Kl[0, 2])) < 2e-3 and abs(my[i, j] - (Kl[1, 1] * y * s + Kl[1, 2])) < 2e-3
    # the wide rectified view (fx = 61.8 on a 512 px image, +-76 degrees) still lands on (or a few pixels off) the fisheye sensor
    assert np.isfinite(mx).all() and np.isfinite(my).all() and -8 < min(mx.min(), my.min()) and max(mx.max(), my.max()) < 520
    # rays behind the camera (a rectifying rotation by more than 90 degrees) map to +-inf, which remap treats as outside
    Rb = np.diag([1.0, -1.0, -1.0])
    bx, by = O.fisheye_rectify_map(E["StereoRectifier.K_left"], E["StereoRectifier.D_left"], Rb.ravel(), c, 8, 8)
    assert np.isinf(bx).all() and np.isinf(by).all()
    assert not O.remap_linear(np.full((16, 16), 200, np.uint8), bx, by).any()
