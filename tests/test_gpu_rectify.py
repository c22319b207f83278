"""util::stereo_rectifier (cv::initUndistortRectifyMap + cv::remap INTER_LINEAR): HIP path vs oracle.
Maps: f64 arithmetic with +, *, / only -> bit-exact floats.  Remap: integer arithmetic -> bit-exact bytes."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from plp import plp, synth

pytestmark = pytest.mark.gpu


def test_euroc_rectifier_maps_and_rectified_frames_match_oracle():
    import torch
    E = O.EUROC
    cam = E["camera"]
    rect = plp.stereo_rectifier(cam, E)
    frames = synth.replay(21, 3, 480, 752)
    dev = torch.device("cuda", 0)
    d_l = torch.from_numpy(frames).to(dev); d_r = torch.from_numpy(np.ascontiguousarray(frames[:, :, ::-1])).to(dev)
    out_l, out_r = rect.rectify(d_l, d_r)
    torch.cuda.synchronize()
    for eye, src, out in (("left", frames, out_l), ("right", frames[:, :, ::-1], out_r)):
        mx, my = O.rectify_map(E[f"StereoRectifier.K_{eye}"], E[f"StereoRectifier.D_{eye}"], E[f"StereoRectifier.R_{eye}"], cam, 480, 752)
        gx, gy = rect.maps[eye][0].cpu().numpy(), rect.maps[eye][1].cpu().numpy()
        assert np.array_equal(gx, mx) and np.array_equal(gy, my), (np.abs(gx - mx).max(), np.abs(gy - my).max())
        for b in range(3):
            assert np.array_equal(out[b].cpu().numpy(), O.remap_linear(src[b], mx, my)), (eye, b)
    # a single 2-D frame goes through the same path
    one_l, one_r = rect.rectify(d_l[1], d_r[1])
    assert torch.equal(one_l, out_l[1]) and torch.equal(one_r, out_r[1])


@pytest.mark.parametrize("n_dist", [0, 4, 8, 12])
def test_rectify_map_distortion_vector_lengths(n_dist):
    import torch
    rng = np.random.default_rng(n_dist)
    full = np.array([-0.3, 0.1, 1e-3, -7e-4, -0.02, 0.01, -0.004, 0.002, 1e-3, -2e-4, 5e-4, 1e-4])
    D = full[:n_dist]
    K = np.array([300.0, 0, 322.0, 0, 305.0, 241.0, 0, 0, 1])
    ax = rng.normal(size=3) * 0.02
    th = np.linalg.norm(ax); k = ax / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).ravel()
    cam = dict(fx=290.123456789, fy=291.987654321, cx=318.2, cy=239.9)
    rows, cols = 97, 131
    mt = plp.matcher()
    dev = torch.device("cuda", 0)
    step = (cols + 5) * 4                                                 # padded map rows
    d_mx = torch.zeros((rows, cols + 5), dtype=torch.float32, device=dev); d_my = torch.zeros_like(d_mx)
    c = plp.camera_c()
    for kk, v in cam.items():
        setattr(c, kk, v)
    torch.cuda.synchronize()
    plp._check(plp.lib().plp_rectify_map_device(mt._h, plp._p(K), plp._p(D) if n_dist else None, n_dist, plp._p(R), C.byref(c), rows, cols, d_mx.data_ptr(),
                                                d_my.data_ptr(), step, None))
    torch.cuda.synchronize()
    mx, my = O.rectify_map(K, D, R, cam, rows, cols)
    assert np.array_equal(d_mx.cpu().numpy()[:, :cols], mx) and np.array_equal(d_my.cpu().numpy()[:, :cols], my)
    assert not d_mx.cpu().numpy()[:, cols:].any()


def test_remap_borders_fractions_strides_and_errors():
    import torch
    rng = np.random.default_rng(4)
    dev = torch.device("cuda", 0)
    mt = plp.matcher()
    L = plp.lib()
    for (rows, cols, drows, dcols) in [(48, 64, 48, 64), (33, 47, 29, 53), (7, 5, 9, 11), (1, 1, 3, 3), (480, 752, 480, 752)]:
        B = 2
        src = rng.integers(0, 256, (B, rows, cols + 3), dtype=np.uint8)      # source rows padded by 3 bytes
        yy, xx = np.mgrid[0:drows, 0:dcols].astype(np.float32)
        mx = (xx * (cols / dcols) + rng.uniform(-3, 3, xx.shape)).astype(np.float32)
        my = (yy * (rows / drows) + rng.uniform(-3, 3, yy.shape)).astype(np.float32)
        mx[0, 0], my[0, 0] = -1.0, -1.0                                     # just outside: only the (1,1) tap is inside
        mx[-1, -1], my[-1, -1] = cols - 1, rows - 1                         # last pixel exactly
        mx[0, -1], my[0, -1] = 1e6, -1e6                                    # far outside
        mx[-1, 0] = cols - 0.5                                              # half-way into the right border
        d_src = torch.from_numpy(src).to(dev); d_mx = torch.from_numpy(mx).to(dev); d_my = torch.from_numpy(my).to(dev)
        d_out = torch.full((B, drows, dcols + 1), 77, dtype=torch.uint8, device=dev)     # odd destination step: byte-store path
        torch.cuda.synchronize()
        plp._check(L.plp_remap_linear_device(mt._h, d_src.data_ptr(), rows, cols, cols + 3, rows * (cols + 3), d_mx.data_ptr(), d_my.data_ptr(), dcols * 4,
                                             drows, dcols, B, d_out.data_ptr(), dcols + 1, drows * (dcols + 1), None))
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert (got[:, :, dcols] == 77).all()
        for b in range(B):
            want = O.remap_linear(np.ascontiguousarray(src[b][:, :cols]), mx, my)
            assert np.array_equal(got[b][:, :dcols], want), (rows, cols, drows, dcols, b)
    with pytest.raises(plp.PlpError):
        plp._check(L.plp_remap_linear_device(mt._h, d_src.data_ptr(), 0, cols, cols, 0, d_mx.data_ptr(), d_my.data_ptr(), dcols * 4, drows, dcols, 1,
                                             d_out.data_ptr(), dcols, 0, None))
    with pytest.raises(plp.PlpError):
        plp.stereo_rectifier(O.EUROC["camera"], {**O.EUROC, "StereoRectifier.model": "equirectangular"})
    with pytest.raises(plp.PlpError):                                        # the fisheye model has exactly 4 coefficients
        plp.stereo_rectifier(O.EUROC["camera"], {**O.EUROC, "StereoRectifier.model": "fisheye"})
    c = plp.camera_c(); c.fx = c.fy = 1.0
    with pytest.raises(plp.PlpError):                                        # singular K_rect * R
        plp._check(L.plp_rectify_map_device(mt._h, plp._p(np.eye(3).ravel()), None, 0, plp._p(np.zeros(9)), C.byref(c), 4, 4, d_mx.data_ptr(), d_my.data_ptr(), 16, None))
    with pytest.raises(plp.PlpError):                                        # 3 distortion coefficients
        plp._check(L.plp_rectify_map_device(mt._h, plp._p(np.eye(3).ravel()), plp._p(np.zeros(3)), 3, plp._p(np.eye(3).ravel()), C.byref(c), 4, 4,
                                            d_mx.data_ptr(), d_my.data_ptr(), 16, None))


def test_tum_vi_fisheye_rectifier():
    """cv::fisheye::initUndistortRectifyMap on the TUM-VI calibration: the map goes through atan() and is held to 1e-4 px
    against the oracle (in practice all but a handful of the 2 x 512 x 512 floats are identical); the remap on the device's
    own maps is exact."""
    import torch
    E = O.TUM_VI
    cam = E["camera"]
    rect = plp.stereo_rectifier(cam, E)
    frames = synth.replay(8, 2, 512, 512)
    dev = torch.device("cuda", 0)
    d_l = torch.from_numpy(frames).to(dev); d_r = torch.from_numpy(np.ascontiguousarray(frames[:, ::-1])).to(dev)
    out_l, out_r = rect.rectify(d_l, d_r)
    torch.cuda.synchronize()
    for eye, src, out in (("left", frames, out_l), ("right", frames[:, ::-1], out_r)):
        mx, my = O.fisheye_rectify_map(E[f"StereoRectifier.K_{eye}"], E[f"StereoRectifier.D_{eye}"], E[f"StereoRectifier.R_{eye}"], cam, 512, 512)
        gx, gy = rect.maps[eye][0].cpu().numpy(), rect.maps[eye][1].cpu().numpy()
        assert np.abs(gx - mx).max() <= 1e-4 and np.abs(gy - my).max() <= 1e-4
        assert (gx == mx).mean() > 0.99 and (gy == my).mean() > 0.99
        for b in range(2):
            assert np.array_equal(out[b].cpu().numpy(), O.remap_linear(src[b], gx, gy)), (eye, b)
        assert out[0].cpu().numpy().any()
