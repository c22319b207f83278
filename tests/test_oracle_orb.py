"""Pins the ORB oracle against everything the reference's own tests hold for this path
(SURVEY.md §4, §8c): test/PLPSLAM/feature/orb_extractor.cc (toy corners, masks),
test/PLPSLAM/feature/orb_params.cc:159-211 (scale tables), test/PLPSLAM/util/trigonometric.cc,
and the quota comment orb_extractor.cc:255-264.  CPU only."""
import math
import numpy as np
import pytest
from PIL import Image

import oracle_lib as O


def toy(rows, cols, x0, y0, x1, y1):
    """white image + filled black rectangle, corners inclusive and clipped (cv::rectangle(..., -1))"""
    img = np.full((rows, cols), 255, np.uint8)
    img[max(y0, 0):min(y1 + 1, rows), max(x0, 0):min(x1 + 1, cols)] = 0
    return img


def ulp_close(a, b, ulps=4):  # EXPECT_FLOAT_EQ
    a = np.float32(a); b = np.float32(b)
    return abs(int(a.view(np.int32)) - int(b.view(np.int32))) <= ulps


@pytest.mark.parametrize("rows,cols,rect,cx,cy", [
    (600, 600, (300, 300, 600, 600), 300, 300),          # extract_toy_sample_1  :27
    (2000, 2000, (0, 0, 1800, 1800), 1800, 1800),        # extract_toy_sample_2  :56
    (1200, 600, (300, 600, 600, 1200), 300, 600),        # extract_toy_sample_3  :361 (portrait quirk)
])
def test_toy_corner(rows, cols, rect, cx, cy):
    ex = O.OrbOracle()
    kps, desc = ex.extract(toy(rows, cols, *rect))
    assert len(kps) > 0 and len(kps) == len(desc) and desc.dtype == np.uint8
    sf = ex.tables()["scale_factors"]
    for k in kps:
        assert abs(k["x"] - cx) <= 2.0 * sf[k["octave"]]
        assert abs(k["y"] - cy) <= 2.0 * sf[k["octave"]]


@pytest.fixture(scope="module")
def frames(golden_dir):
    return [np.asarray(Image.open(golden_dir / n)) for n in
            ("equirect1_640x480.png", "equirect2_640x480.png", "equirect1_crop_640x480.png")]


def test_real_images_nonempty(frames):
    for img in frames:
        for im in (img, np.ascontiguousarray(np.rot90(img, -1))):  # landscape + portrait (:381-420)
            kps, desc = O.OrbOracle().extract(im)
            assert len(kps) > 0 and len(kps) == len(desc)


def test_image_mask_rows(frames):  # extract_with_image_mask_1 :125
    img = frames[0]
    rows = img.shape[0]
    mask = np.ones_like(img)
    mask[0:rows // 4] = 0
    mask[3 * rows // 4:rows - 1] = 0
    kps, _ = O.OrbOracle().extract(img, mask)
    assert len(kps) > 0
    assert (kps["y"] >= rows // 4).all() and (kps["y"] <= 3 * rows // 4).all()


def test_image_mask_cols(frames):  # extract_with_image_mask_2 :165
    img = frames[1]
    cols = img.shape[1]
    mask = np.ones_like(img)
    mask[:, 0:cols // 4] = 0
    mask[:, 3 * cols // 4:cols - 1] = 0
    kps, _ = O.OrbOracle().extract(img, mask)
    assert len(kps) > 0
    assert (kps["x"] >= cols // 4).all() and (kps["x"] <= 3 * cols // 4).all()


@pytest.mark.parametrize("rects,axis", [
    ([[0.0, 1.0, 0.0, 0.2], [0.0, 1.0, 0.8, 1.0]], "y"),   # extract_with_rectangle_mask_1 :248
    ([[0.0, 0.2, 0.0, 1.0], [0.8, 1.0, 0.0, 1.0]], "x"),   # extract_with_rectangle_mask_2 :285
])
def test_rect_mask(frames, rects, axis):
    img = frames[0]
    ext = img.shape[0] if axis == "y" else img.shape[1]
    kps, _ = O.OrbOracle(mask_rects=rects).extract(img)
    assert len(kps) > 0
    assert (kps[axis] >= ext // 5).all() and (kps[axis] <= 4 * ext // 5).all()


def test_scale_tables():  # orb_params.cc tests :159-211
    import ctypes as C
    n, sf = 10, np.float32(1.26)
    t = [np.zeros(n, np.float32) for _ in range(4)]
    O.lib().oracle_scale_tables(n, float(sf), *[a.ctypes.data_as(C.c_void_p) for a in t])
    s = np.float32(1.0)
    for level in range(n):
        assert ulp_close(t[0][level], np.float32(math.pow(float(sf), level)))
        assert ulp_close(t[1][level], np.float32(math.pow(float(np.float32(1.0) / sf), level)))
        assert ulp_close(t[2][level], s * s)
        assert ulp_close(t[3][level], np.float32(1.0) / (s * s))
        s = np.float32(sf * s)


def test_quota_known_answer():  # comment orb_extractor.cc:255-264
    t = O.OrbOracle(1000).tables()
    assert t["quota"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert t["u_max"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert O.OrbOracle(2000).tables()["quota"].sum() == 2000


def test_trig_tolerance():  # test/PLPSLAM/util/trigonometric.cc: 0..360 deg step 0.1, 1e-3
    L = O.lib()
    for i in range(3601):
        v = i * 0.1 * math.pi / 180.0
        assert abs(L.oracle_trig_cos(v) - math.cos(v)) < 1e-3
        assert abs(L.oracle_trig_sin(v) - math.sin(v)) < 1e-3


def test_fast_atan2_accuracy():
    L = O.lib()
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1000
        ref = math.degrees(math.atan2(y, x)) % 360.0
        got = L.oracle_fast_atan2(y, x)
        assert abs(((got - ref + 180) % 360) - 180) < 0.02


def test_fast_known_corner():
    """a bright square corner on dark ground (with a little noise so that NMS ties break):
    FAST fires at the corner; the score does not depend on the threshold; thr-20 set is a subset of thr-7"""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 12, (40, 40)).astype(np.uint8)
    img[20:, 20:] += 190
    pts = O.fast9_16(img, 20)
    assert len(pts) >= 1
    assert any(abs(p[0] - 20) <= 2 and abs(p[1] - 20) <= 2 for p in pts)
    p7 = O.fast9_16(img, 7)
    s20 = {(x, y): s for x, y, s in pts}
    s7 = {(x, y): s for x, y, s in p7}
    assert set(s20) <= set(s7)
    for k, s in s20.items():
        assert s7[k] == s and s >= 20
    # brute-force definition of the score: largest t for which 9 contiguous circle pixels are all
    # brighter than v+t or all darker than v-t
    off = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
           (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    for (x, y), s in s7.items():
        v = int(img[y, x])
        d = [int(img[y + dy, x + dx]) - v for dx, dy in off]
        best = 0
        for k in range(16):
            arc = [d[(k + i) % 16] for i in range(9)]
            best = max(best, min(arc), min(-a for a in arc))
        assert s == best - 1


def test_blur_constant_and_taps():
    assert O.gaussian_taps_q8(7, 2.0).tolist() == [18, 34, 48, 56, 48, 34, 18]
    assert O.gaussian_taps_q8(5, 1.0).sum() == 256 and O.gaussian_taps_q8(11, 1.2).sum() == 256
    img = np.full((30, 50), 137, np.uint8)
    assert (O.gaussian_blur_u8(img, 7, 2.0) == 137).all()


def test_resize_identity_and_constant():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    assert (O.resize_linear_u8(img, 48, 64) == img).all()
    c = np.full((48, 64), 99, np.uint8)
    assert (O.resize_linear_u8(c, 40, 53) == 99).all()


def test_extract_structure(frames):
    """keypoints concatenated by level, level-0 coordinates, size/octave consistent, count ~ quota"""
    ex = O.OrbOracle(1000)
    kps, desc = ex.extract(frames[2])
    sf = ex.tables()["scale_factors"]
    assert (np.diff(kps["octave"]) >= 0).all()
    assert 900 <= len(kps) <= 1100
    for lvl in range(8):
        sel = kps[kps["octave"] == lvl]
        if len(sel):
            assert (sel["size"] == np.float32(int(31 * sf[lvl]))).all()
            lk = ex.level_keypts(lvl)
            assert len(lk) == len(sel)
            if lvl:
                assert np.array_equal(sel["x"], lk["x"] * sf[lvl])
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360)).all()
    assert (kps["class_id"] == -1).all()
    assert desc.any()


def test_empty_image_is_noop():
    ex = O.OrbOracle()
    kps, desc = ex.extract(np.zeros((0, 0), np.uint8))
    assert len(kps) == 0


def _golden_ref_cases(golden_dir):
    z = np.load(golden_dir / "ref_orb.npz")
    for key in sorted(k for k in z.files if k.endswith("__kps")):
        name, K = key.split("__")[0], int(key.split("__")[1][1:])
        kps = np.ascontiguousarray(z[key]).view(O.KP_DTYPE).reshape(-1)
        yield name, K, kps, z[key.replace("__kps", "__desc")]


def test_oracle_equals_committed_reference_vectors(golden_dir):
    """tests/golden/ref_orb.npz = key points and descriptors of the reference's OWN ORB sources (oracle/_ref, made by
    tools/make_golden_ref.py where /root/reference is mounted) on the fixture frames: the pin travels with the repository."""
    n = 0
    for name, K, kps, desc in _golden_ref_cases(golden_dir):
        img = np.asarray(Image.open(golden_dir / f"{name}.png").convert("L"), dtype=np.uint8)
        ok, od = O.OrbOracle(K).extract(img)
        assert len(ok) == len(kps) and np.array_equal(ok, kps) and np.array_equal(od, desc), (name, K)
        n += 1
    assert n == 6
