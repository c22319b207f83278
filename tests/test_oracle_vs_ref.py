"""The oracle against the REFERENCE'S OWN ORB extractor sources (oracle/_ref/libplpref.so = /root/reference's
orb_extractor.cc, orb_extractor_node.cc, orb_params.cc, trigonometric.h, match/base.h, match/angle_checker.h compiled
against oracle/ref_shim by oracle/ref_build.sh).  Pins every reference-owned line of the ORB path; the OpenCV
primitives underneath are the same restatement on both sides (cv_restated.hpp), so they stay unpinned.
Runs wherever the prebuilt library is present (it is built here, where /root/reference is mounted, and travels)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from plp import synth

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "libplpref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(_REF), reason="oracle/_ref not built (needs /root/reference)")


def ref():
    L = C.CDLL(_REF)
    L.ref_orb_extract.restype = C.c_int
    L.ref_orb_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int]
    L.ref_orb_tables.argtypes = [C.c_int, C.c_float, C.c_int, C.c_void_p]
    L.ref_cos.restype = C.c_float; L.ref_cos.argtypes = [C.c_float]
    L.ref_sin.restype = C.c_float; L.ref_sin.argtypes = [C.c_float]
    L.ref_hamming32.restype = C.c_uint; L.ref_hamming32.argtypes = [C.c_void_p, C.c_void_p]
    L.ref_hamming64.restype = C.c_uint; L.ref_hamming64.argtypes = [C.c_void_p, C.c_void_p]
    L.ref_angle_checker.restype = C.c_int
    L.ref_angle_checker.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return L


def ref_extract(img, K, scale=1.2, levels=8, ini=20, mn=7, rects=(), mask=None):
    img = np.ascontiguousarray(img, np.uint8)
    cap = 2 * K + 64
    kps = np.zeros(cap, O.KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    r = np.ascontiguousarray(np.asarray(rects, np.float32).reshape(-1, 4))
    m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
    n = ref().ref_orb_extract(img.ctypes.data, img.shape[0], img.shape[1], K, scale, levels, ini, mn, r.ctypes.data if len(r) else None, len(r),
                              m.ctypes.data if m is not None else None, kps.ctypes.data, desc.ctypes.data, cap)
    assert n >= 0
    return kps[:n].copy(), desc[:n].copy()


def same(a, b):
    ka, da = a; kb, db = b
    assert len(ka) == len(kb)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(ka[f], kb[f]), f
    assert np.array_equal(da, db)


@pytest.mark.parametrize("K", [500, 1000, 2000])
def test_extract_matches_reference_sources_on_fixture_frames(K):
    frames = synth.replay(77, 3, 480, 640)
    for f in frames:
        same(O.OrbOracle(K).extract(f), ref_extract(f, K))


def test_extract_other_shapes_levels_thresholds():
    rng = np.random.default_rng(5)
    canvas = np.ascontiguousarray(synth.canvas(1, 700, 1500)).astype(np.uint8)
    for (rows, cols, K, scale, levels, ini, mn) in [(376, 1241, 2000, 1.2, 8, 20, 7), (480, 752, 1000, 1.2, 8, 20, 7), (240, 320, 300, 1.5, 4, 30, 10),
                                                   (200, 150, 400, 1.2, 6, 20, 7), (120, 160, 100, 1.1, 3, 12, 5)]:
        y0 = int(rng.integers(0, canvas.shape[0] - rows)); x0 = int(rng.integers(0, canvas.shape[1] - cols))
        img = np.ascontiguousarray(canvas[y0:y0 + rows, x0:x0 + cols])
        same(O.OrbOracle(K, scale, levels, ini, mn).extract(img), ref_extract(img, K, scale, levels, ini, mn))


def test_extract_noise_flat_and_masks():
    rng = np.random.default_rng(6)
    noise = rng.integers(0, 256, (240, 320), dtype=np.uint8)            # thousands of candidates: quadtree pool ties
    same(O.OrbOracle(600).extract(noise), ref_extract(noise, 600))
    flat = np.full((240, 320), 90, np.uint8); flat[100:140, 150:200] = 200   # mostly empty cells: the threshold fallback
    same(O.OrbOracle(500).extract(flat), ref_extract(flat, 500))
    img = synth.replay(9, 1, 480, 640)[0]
    rects = [[0.1, 0.5, 0.2, 0.6], [0.7, 0.95, 0.0, 0.3]]
    same(O.OrbOracle(1000, mask_rects=rects).extract(img), ref_extract(img, 1000, rects=rects))
    mask = np.full(img.shape, 255, np.uint8); mask[:, 200:330] = 0; mask[300:, :] = 0
    same(O.OrbOracle(1000).extract(img, mask), ref_extract(img, 1000, mask=mask))


def test_tables_trig_hamming_angle_checker():
    L = ref()
    for (K, scale, levels) in [(1000, 1.2, 8), (2000, 1.2, 8), (700, 1.5, 5)]:
        out = np.zeros(4 * levels, np.float32)
        L.ref_orb_tables(K, scale, levels, out.ctypes.data)
        t = O.OrbOracle(K, scale, levels).tables()
        assert np.array_equal(out[:levels], t["scale_factors"]) and np.array_equal(out[levels:2 * levels], t["inv_scale_factors"])
        assert np.array_equal(out[2 * levels:3 * levels], t["level_sigma_sq"]) and np.array_equal(out[3 * levels:], t["inv_level_sigma_sq"])
    rng = np.random.default_rng(8)
    for v in np.concatenate([rng.uniform(-20, 20, 2000), [0, np.pi / 2, np.pi, -np.pi, 2 * np.pi, 1e-8]]).astype(np.float32):
        assert O.lib().oracle_trig_cos(float(v)) == L.ref_cos(float(v)) and O.lib().oracle_trig_sin(float(v)) == L.ref_sin(float(v))
    d = rng.integers(0, 256, (400, 32), dtype=np.uint8)
    for i in range(0, 400, 2):
        want = int(np.unpackbits(d[i] ^ d[i + 1]).sum())
        assert L.ref_hamming32(d[i].ctypes.data, d[i + 1].ctypes.data) == want == L.ref_hamming64(d[i].ctypes.data, d[i + 1].ctypes.data)
        assert O.hamming32(d[i], d[i + 1]) == want
    for n in (0, 1, 50, 2000):
        deltas = (rng.uniform(-359.9, 359.9, n)).astype(np.float32)   # differences of two angles in [0, 360)
        for valid in (0, 1):
            out = np.zeros(max(n, 1), np.int32)
            k = L.ref_angle_checker(deltas.ctypes.data, n, 30, 3, valid, out.ctypes.data)
            got = O.angle_checker(deltas, 30, 3, bool(valid))
            # the reference's std::sort of the bins has an unspecified order for equally full bins (D3): compare as sets
            # only when the top-3 is unambiguous
            assert sorted(out[:k].tolist()) == sorted(got.tolist()) or _ambiguous_bins(deltas)


def _ambiguous_bins(deltas):
    d = np.asarray(deltas, np.float32).copy()
    d[d < 0] += np.float32(360.0); d[d >= 360] -= np.float32(360.0)
    b = np.rint(d * np.float32(1.0 / 30)).astype(int)
    h = np.bincount(b[b < 30], minlength=30)
    s = np.sort(h)[::-1]
    return s[2] == s[3]
