"""Oracle of the post-extract step (oracle/post_oracle.cpp): properties that hold for any correct cv::undistortPoints /
bearing / depth-lookup restatement.  (The reference has no test for this step: parity unpinned, see the file header.)"""
import numpy as np

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
