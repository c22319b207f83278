"""The OpenCV restatements (oracle/cv_restated.hpp, lsd_restated.hpp, post_oracle.cpp) against a REAL OpenCV, when its
dump is present: tests/golden/opencv_crosscheck.npz is written by tools/opencv_crosscheck.cpp (run wherever OpenCV 3.4.16
is installed, packed by tools/opencv_crosscheck_pack.py).  The build image has no OpenCV, so the file is not committed yet
and this module is skipped: the primitives below stay "parity unpinned" (README.md) until someone runs the tool.
Each check names the reference call site whose arithmetic it pins."""
import os
import pathlib

import numpy as np
import pytest

import oracle_lib as O

G = pathlib.Path(__file__).resolve().parent / "golden"
NPZ = pathlib.Path(os.environ.get("PLP_OPENCV_CROSSCHECK", G / "opencv_crosscheck.npz"))
pytestmark = pytest.mark.skipif(not NPZ.exists(), reason="tests/golden/opencv_crosscheck.npz absent (needs a machine with OpenCV: tools/opencv_crosscheck.cpp)")
NAMES = ("equirect1_640x480", "equirect1_crop_640x480", "equirect2_640x480", "equirect2_crop_640x480")


def frames():
    from PIL import Image
    return {n: np.asarray(Image.open(G / f"{n}.png").convert("L"), dtype=np.uint8) for n in NAMES}


@pytest.fixture(scope="module")
def z():
    return np.load(NPZ)


def test_pyramid_resize(z):                       # feature/orb_extractor.cc:315-326
    for n, img in frames().items():
        ex = O.OrbOracle(1000)
        ex.extract(img)
        for l in range(1, 8):
            assert np.array_equal(ex.level_image(l), z[f"{n}__pyr{l}"]), (n, l)


def test_fast(z):                                 # feature/orb_extractor.cc:404,410
    for n, img in frames().items():
        ex = O.OrbOracle(1000)
        ex.extract(img)
        for l in (0, 3):
            lvl = img if l == 0 else ex.level_image(l)
            for thr in (20, 7):
                want = z[f"{n}__fast_l{l}_t{thr}"]
                got = O.fast9_16(lvl, thr)
                if len(got) == 0:
                    assert (want == -1).all()
                else:
                    assert np.array_equal(got, want), (n, l, thr)


def test_gaussian_blur(z):                        # orb_extractor.cc:149, binary_descriptor_custom.cpp:355, lsd.cpp
    for n, img in frames().items():
        assert np.array_equal(O.gaussian_blur_u8(img, 7, 2.0), z[f"{n}__blur7"]), n
        assert np.array_equal(O.gaussian_blur_u8(img, 5, 1.0), z[f"{n}__blur5"]), n
        assert np.array_equal(O.gaussian_blur_u8(img, 11, 0.6 / 0.5), z[f"{n}__blur11"]), n


def test_fast_atan2(z):                           # feature/orb_extractor.cc:734
    t = z["fast_atan2"]
    got = np.array([O.lib().oracle_fast_atan2(float(y), float(x)) for y, x, _ in t], np.float32)
    assert np.array_equal(got, t[:, 2])


def test_lsd_and_its_resize(z):                   # LSDDetector_custom.cpp:241-257, lsd.cpp
    for n, img in frames().items():
        lo = O.LineOracle(img, stable_order=False)      # std::sort order, as OpenCV runs it (definition D1 is the stable variant)
        if f"{n}__lsd_scaled" in z.files:
            assert np.array_equal(lo.scaled, z[f"{n}__lsd_scaled"]), n
        if f"{n}__lsd_lines" in z.files:
            want = z[f"{n}__lsd_lines"]
            assert lo.raw.shape == want.shape and np.abs(lo.raw - want).max() <= 1e-4, n


def test_sobel(z):                                # binary_descriptor_custom.cpp:392-393
    for n, img in frames().items():
        lo = O.LineOracle(img)
        assert np.array_equal(lo.dx, z[f"{n}__sobel_dx"]) and np.array_equal(lo.dy, z[f"{n}__sobel_dy"]), n


def test_remap_and_rectification(z):              # feature/line_extractor.cc:65-103, util/stereo_rectifier.cc:61-84
    K = np.array([458.654, 0, 367.215, 0, 457.296, 248.375, 0, 0, 1]); D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0])
    R = np.array([0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127, 0.9999741760894847, 0.007055629199258132,
                  -0.008089410156878961, -0.007044357138835809, 0.9999424675829176])
    cam = dict(fx=np.float32(435.2046959714599), fy=np.float32(435.2046959714599), cx=np.float32(367.4517211914062), cy=np.float32(252.2008514404297))
    for n, img in frames().items():
        assert np.array_equal(img, z[f"{n}__remap_identity"]), n
        mx, my = O.rectify_map(K, D, R, cam, *img.shape)
        assert np.abs(mx - z[f"{n}__rectify_map_x"]).max() <= 1e-4 and np.abs(my - z[f"{n}__rectify_map_y"]).max() <= 1e-4
        assert np.array_equal(O.remap_linear(img, z[f"{n}__rectify_map_x"], z[f"{n}__rectify_map_y"]), z[f"{n}__rectified"]), n


def test_undistort_points(z):                     # camera/perspective.cc:130-162
    pts = z["undistort_in"]
    kps = np.zeros(len(pts), O.KP_DTYPE); kps["x"] = pts[:, 0]; kps["y"] = pts[:, 1]
    cam = np.array([517.306408, 516.469215, 318.643040, 255.313989, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314, 40.0], np.float64)
    und = O.post_extract(cam, kps)["undist_keypts"]
    want = z["undistort_out"]
    assert np.abs(und["x"] - want[:, 0]).max() <= 2e-3 and np.abs(und["y"] - want[:, 1]).max() <= 2e-3


def test_colour_and_depth_conversion(z):          # util/image_converter.cc:33-80
    import ctypes as C
    col = np.ascontiguousarray(z["color_src"])
    for bgr, key in ((0, "gray_rgb"), (1, "gray_bgr")):
        out = np.zeros(col.shape[:2], np.uint8)
        O.lib().oracle_convert_to_grayscale(C.c_void_p(col.ctypes.data), col.shape[0], col.shape[1], 3, bgr, C.c_void_p(out.ctypes.data))
        assert np.array_equal(out, z[key])
    d16 = np.ascontiguousarray(z["depth_u16"]); out = np.zeros(d16.shape, np.float32)
    O.lib().oracle_convert_to_true_depth_u16(C.c_void_p(d16.ctypes.data), C.c_size_t(d16.size), C.c_double(5208.0), C.c_void_p(out.ctypes.data))
    assert np.array_equal(out, z["depth_f32"])
