"""The matchers on EMPTY sides (a dark frame has no key points, a fresh map no landmarks, a wall no key lines): zero targets, zero queries, both -- through the host C ABI, against
the oracle (the reference's loops simply do not run: 0 matches, every slot -1)."""
import importlib

import numpy as np
import pytest

import oracle_lib as O
from match_cases import random_line_problem, random_problem

pytestmark = pytest.mark.gpu
plp = importlib.import_module("structure-plp-slam_amd")
SF = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
SF_LSD = np.array([1.0, 2.0], np.float32)


def cut(d, n):
    return {k: v[:n] for k, v in d.items()}


@pytest.mark.parametrize("n, m", [(0, 40), (40, 0), (0, 0), (1, 0), (0, 1)])
def test_point_matchers_with_an_empty_side(n, m):
    rng = np.random.default_rng(5)
    t_full, q_full = random_problem(rng, 64, 64, n_words=3)
    t, q = cut(t_full, n), cut(q_full, m)
    grid = plp.make_grid(640, 480)
    want, wn = O.match_frame_and_landmarks(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"], q["q_reproj"],
                                           q["q_x_right"], q["q_level"], q["q_desc"], q["q_has_obs"], 12.0, 0.8)
    got, gn = plp.matcher(0.8, True).match_host(plp.MODE_LANDMARKS, n, m, {**t, **q}, margin=12.0, scale_factors=SF, grid=grid)
    assert gn[0] == wn == 0 and np.array_equal(got[0][:n], want)
    for direction in (0, 1, 2):
        want, wn = O.match_current_and_last(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"], q["q_reproj"],
                                            q["q_x_right"], q["q_level"], q["q_angle"], q["q_desc"], q["q_has_obs"], 15.0, direction, True)
        got, gn = plp.matcher(0.9, True).match_host(plp.MODE_LAST_FRAME, n, m, {**t, **q}, margin=15.0, direction=direction, scale_factors=SF, grid=grid)
        assert gn[0] == wn == 0 and np.array_equal(got[0][:n], want)


@pytest.mark.parametrize("n, m", [(0, 30), (30, 0), (0, 0)])
def test_line_matchers_with_an_empty_side(n, m):
    rng = np.random.default_rng(6)
    t_full, q_full = random_line_problem(rng, 48, 48, 2)
    t, q = cut(t_full, n), cut(q_full, m)
    want, wn = O.match_frame_and_landmarks_line(t["t_kl"], t["t_desc"], t["t_kp_octave"], t["t_occupied"], SF_LSD, q["q_valid"], q["q_reproj"], q["q_reproj2"],
                                                q["q_level"], q["q_desc"], q["q_has_obs"], 10.0, 0.8)
    got, gn = plp.matcher(0.8, False).match_host(plp.MODE_LANDMARKS_LINE, n, m, {**t, **q}, margin=10.0, scale_factors=SF_LSD)
    assert gn[0] == wn == 0 and np.array_equal(got[0][:n], want)
    xr_pair = np.stack([t["t_x_right"], t["t_x_right2"]], 1) if n else np.zeros((0, 2), np.float32)
    want, wn = O.match_current_and_last_line(t["t_kl"], t["t_desc"], xr_pair, t["t_occupied"], SF_LSD, 1, q["q_valid"], q["q_reproj"], q["q_reproj2"], q["q_x_right"],
                                             q["q_x_right2"], q["q_level"], q["q_desc"], q["q_has_obs"], 10.0, 0, 1)
    got, gn = plp.matcher(0.9, True).match_host(plp.MODE_LAST_FRAME_LINE, n, m, {**t, **q, "is_rgbd": 1, "num_levels_lsd": 1}, margin=10.0, direction=0, scale_factors=SF_LSD)
    assert gn[0] == wn == 0 and np.array_equal(got[0][:n], want)


def test_the_device_entry_with_an_empty_side_writes_the_defined_result_on_the_callers_stream():
    import torch
    dev = torch.device("cuda", 0)
    mt = plp.matcher(0.8, True)
    B, n = 3, 40
    out_match = torch.zeros((B, n), dtype=torch.int32, device=dev)
    out_num = torch.full((B,), 7, dtype=torch.int32, device=dev)
    # no queries at all: the target-side arrays are there, the query side is NULL
    rng = np.random.default_rng(9)
    t, _ = random_problem(rng, n, 8, n_words=2)
    fields = {"t_kps": torch.from_numpy(np.tile(t["t_kps"].view(np.uint8).reshape(1, n, -1), (B, 1, 1))).to(dev), "t_desc": torch.from_numpy(np.tile(t["t_desc"][None], (B, 1, 1))).to(dev)}
    mt.match_device(plp.MODE_LANDMARKS, n, 0, fields, out_match, out_num, margin=10.0, scale_factors=SF, grid=plp.make_grid(640, 480), B=B)
    torch.cuda.synchronize(dev)
    assert int((out_match != -1).sum()) == 0 and int(out_num.abs().sum()) == 0
    # no targets at all
    out_num.fill_(7)
    mt.match_device(plp.MODE_LAST_FRAME, 0, 16, {}, torch.zeros((B, 0), dtype=torch.int32, device=dev), out_num, margin=10.0, scale_factors=SF, grid=plp.make_grid(640, 480), B=B)
    torch.cuda.synchronize(dev)
    assert int(out_num.abs().sum()) == 0
    # a call with both sides present and nothing to read is still refused
    with pytest.raises(plp.PlpError):
        mt.match_host(plp.MODE_LANDMARKS, 5, 5, {})
