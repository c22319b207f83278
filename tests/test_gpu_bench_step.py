"""The step bench.py times, pinned to the oracle: tracker_step (structure-plp-slam_amd/replay_step.py) is run exactly as bench.py runs it --
HALO = 2 feature buffers, NBUF = 2 sets, two line contexts on two streams, replay.exchange_halo_into, the device-side query builders
(plp_replay_point_queries_device / plp_replay_line_queries_device), descriptors read in place through q_desc_stride, the four matcher
calls on their own stream, steps software-pipelined -- at B = 12 distinct frames, and EVERY frame of the last step (frames 0 and 1 read the
circular halo) is compared with the CPU oracle's rendering of the tracker's per-frame sequence (tests/bench_step_check.py): extracted
features, match_current_and_last_frames, match_frame_and_landmarks with the queries in replay_kernels.hip's [b-2, b-1] order,
match_current_and_last_frames_line and match_frame_and_landmarks_line (module/frame_tracker.cc:66-87, tracking_module.cc:983,1060;
match/projection.cc:37-212, 214-527).  Integer work: every array identical."""
import importlib

import numpy as np
import pytest
import torch

import bench_step_check as BC
from plp import plp, synth

rs = importlib.import_module("structure-plp-slam_amd.replay_step")
pytestmark = pytest.mark.gpu
SHIFT = (-3.0, 0.0)


def run_steps(frames, K, n_steps, **kw):
    B, rows, cols = frames.shape
    ts = rs.tracker_step(plp, B, K, rows, cols, shift=SHIFT, **kw)
    d_frames = torch.from_numpy(frames).to(ts.dev)
    buf = 0
    for _ in range(n_steps):
        buf = ts.step(d_frames)
    torch.cuda.synchronize(ts.dev)
    ts.last_batch_status()
    return ts, buf


@pytest.mark.parametrize("K,n_steps", [(1000, 1), (1000, 4), (2000, 3)])
def test_every_frame_of_the_benchmarked_step_equals_the_oracle_chain(K, n_steps):
    frames = synth.replay(4321 + K, 12, 480, 640)
    ts, buf = run_steps(frames, K, n_steps)
    assert buf == (n_steps - 1) % ts.NBUF and ts.n_line == 2 and ts.NBUF == 2
    h = BC.fetch(ts, buf)
    # single rank: the halo rows are the block's own last two frames (circular replay)
    for name in ("kps", "desc", "cnt", "kl", "lbd", "lcnt"):
        assert np.array_equal(h[name][:rs.HALO].view(np.uint8), h[name][-rs.HALO:].view(np.uint8)), name   # bytes: slots past a frame's count are uninitialised
    n, bad = BC.verify(ts, buf, range(len(frames)), frames)
    assert n == len(frames) and not bad, bad
    # the problems are not vacuous: hundreds of point matches and some line matches per frame
    assert h["n"][0].min() > 200 and h["n"][1].min() > 200, (h["n"][0], h["n"][1])
    assert h["n"][2].sum() > 50 and h["n"][3].sum() > 50, (h["n"][2], h["n"][3])
    assert h["lcnt"][rs.HALO:].min() > 3


def test_other_stream_arrangements_give_the_same_step():
    """one stream for everything (PLP_BENCH_SERIAL), one line context, three feature sets, region growing with one wave per frame (what a
    2048-frame step of bench.py runs; small batches default to several waves per frame) or four: same results as the default arrangement"""
    frames = synth.replay(77, 9, 480, 640)
    ts0, b0 = run_steps(frames, 1000, 2)
    want = BC.fetch(ts0, b0)
    for kw in (dict(serial=True), dict(n_line=1), dict(nbuf=3, n_line=3), dict(line_grow_waves=1), dict(line_grow_waves=4)):
        ts, b = run_steps(frames, 1000, 2, **kw)
        got = BC.fetch(ts, b)
        for f in range(len(frames)):
            c, l = int(want["cnt"][rs.HALO + f]), int(want["lcnt"][rs.HALO + f])
            assert int(got["cnt"][rs.HALO + f]) == c and int(got["lcnt"][rs.HALO + f]) == l
            for i, lim in enumerate((c, c, l, l)):
                assert got["n"][i][f] == want["n"][i][f] and np.array_equal(got["m"][i][f][:lim], want["m"][i][f][:lim]), (kw, f, i)


def test_orb_only_step():
    frames = synth.replay(5, 6, 480, 640)
    ts, buf = run_steps(frames, 2000, 2, orb_only=True)
    n, bad = BC.verify(ts, buf, range(len(frames)), frames)
    assert n == 6 and not bad, bad
