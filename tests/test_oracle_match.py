"""Pins the matcher oracle to the reference's own known-answer tests: test/PLPSLAM/match/base.cc (3 Hamming
vectors), test/PLPSLAM/match/angle_checker.cc:13-192 and test/PLPSLAM/data/common_get_cell_indices.cc (the
zero-distortion cases; the distorted-bounds cases need cv::undistortPoints, which is outside the path)."""
import numpy as np
import pytest

import oracle_lib as O
from plp import plp


@pytest.mark.parametrize("a,b,want", [(0b01010101, 0b01010101, 0), (0b01010101, 0b10101010, 256), (0b01100110, 0b00111100, 128)])
def test_hamming_known_answers(a, b, want):   # base.cc:11-66
    da, db = np.full(32, a, np.uint8), np.full(32, b, np.uint8)
    assert O.hamming32(da, db) == want
    assert O.hamming64(da, db) == want


def test_hamming_random_vs_numpy():
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        want = int(np.unpackbits(a ^ b).sum())
        assert O.hamming32(a, b) == want == O.hamming64(a, b)


def _angle_case():
    # angle_checker.cc:17-41: five matches near 35 deg, four near 324, three near 127, four scattered
    deltas = [34.8, 34.9, 35.0, 35.1, 35.2, 323.8, 323.9, 324.1, 324.2, 126.9, 127.0, 127.1, 0.0, 90.0, 180.0, 270.0]
    label = [35] * 5 + [324] * 4 + [127] * 3 + [0, 60, 180, 270]
    return np.array(deltas, np.float32), np.array(label)


@pytest.mark.parametrize("top,valid_labels", [(1, {35}), (2, {35, 324}), (3, {35, 324, 127})])
def test_angle_checker_top_n(top, valid_labels):   # angle_checker.cc:13-151
    d, lab = _angle_case()
    valid = O.angle_checker(d, 30, top, valid=True)
    invalid = O.angle_checker(d, 30, top, valid=False)
    assert len(valid) and set(lab[valid]) == valid_labels
    assert set(lab[invalid]).isdisjoint(valid_labels)
    assert len(valid) + len(invalid) == len(d)


def test_angle_checker_all_bins_valid():   # angle_checker.cc:153-192
    d, lab = _angle_case()
    assert len(O.angle_checker(d, 30, 30, valid=True)) == len(d)
    assert len(O.angle_checker(d, 30, 30, valid=False)) == 0


def test_angle_checker_wraps_negative_and_full_turn():
    assert len(O.angle_checker(np.array([-10.0, 350.0, 370.0], np.float32), 30, 1, valid=True)) == 2   # -10 == 350 (bin 12), 370 -> 10 (bin 0)


def test_get_cell_indices_zero_distortion():   # common_get_cell_indices.cc:66-125 (valid_cases_2)
    import ctypes as C
    cols, rows = 2000, 1000
    g = plp.make_grid(cols, rows)
    eps = np.float32(0.01)
    cw, ch = 1.0 / g.inv_cell_width, 1.0 / g.inv_cell_height
    cx, cy = C.c_int(), C.c_int()
    for ix in range(g.cols):
        for iy in range(g.rows):
            for x, y in [(np.float32(ix * cw + eps), np.float32(iy * ch + eps)), (np.float32((ix + 1) * cw - eps), np.float32(iy * ch + eps)),
                         (np.float32(ix * cw + eps), np.float32((iy + 1) * ch - eps)), (np.float32((ix + 1) * cw - eps), np.float32((iy + 1) * ch - eps))]:
                ok = O.lib().oracle_get_cell_indices(g.min_x, g.min_y, g.inv_cell_width, g.inv_cell_height, g.cols, g.rows, x, y, C.byref(cx), C.byref(cy))
                assert ok and (cx.value, cy.value) == (ix, iy)
    # outside the bounds (invalid_cases with zero distortion)
    for x, y in [(-0.01, -0.01), (cols, -0.01), (-0.01, rows), (cols, rows), (cols / 2.0, -0.01), (cols / 2.0, rows)]:
        assert not O.lib().oracle_get_cell_indices(g.min_x, g.min_y, g.inv_cell_width, g.inv_cell_height, g.cols, g.rows, x, y, C.byref(cx), C.byref(cy))


def test_window_query_equals_bruteforce_definition():
    """get_keypoints_in_cell: result = {in grid, level window, |dx|<m, |dy|<m} ordered by (cell col, cell row, index)"""
    rng = np.random.default_rng(2)
    g = plp.make_grid(640, 480)
    kps = np.zeros(800, O.KP_DTYPE)
    kps["x"] = rng.uniform(-5, 645, 800).astype(np.float32); kps["y"] = rng.uniform(-5, 485, 800).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, 800)
    for _ in range(50):
        rx, ry = np.float32(rng.uniform(0, 640)), np.float32(rng.uniform(0, 480))
        m = np.float32(rng.uniform(5, 60)); lo = int(rng.integers(-1, 7)); hi = lo + int(rng.integers(0, 3))
        got = O.keypoints_in_cell(O.grid6(g), kps, rx, ry, m, lo, hi)
        cx = np.floor((kps["x"] - np.float32(g.min_x)).astype(np.float64) * g.inv_cell_width).astype(int)
        cy = np.floor((kps["y"] - np.float32(g.min_y)).astype(np.float64) * g.inv_cell_height).astype(int)
        ok = (cx >= 0) & (cx < 64) & (cy >= 0) & (cy < 48) & (np.abs(kps["x"] - rx) < m) & (np.abs(kps["y"] - ry) < m)
        if lo > 0 or hi >= 0:
            ok &= kps["octave"] >= lo
            if hi >= 0:
                ok &= kps["octave"] <= hi
        idx = np.nonzero(ok)[0]
        want = idx[np.lexsort((idx, cy[idx], cx[idx]))]
        assert np.array_equal(got, want)
