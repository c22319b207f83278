"""Soak of the SHIPPED exact seed sort (4 waves per workgroup, 4096-entry LDS window: the configuration of batches above 256 frames) inside the overlapped step.

Why it exists: an EXPERIMENT build of this kernel with two waves per workgroup page-faulted in its swap pass whenever other
kernels ran beside it, and the cause was never identified.  The shipped kernel is the same source with other constants, so it is held to the result here, under the
conditions in which the experiment failed: 200 steps of the benchmark's tracker_step (ORB on one stream, two line sub-blocks of 512 frames on two more, the matchers of
the previous step on a fourth), steps enqueued in pairs so that consecutive steps overlap as in bench.py.  After every pair the seed order of 32 sampled frames is
compared with what std::sort leaves (the oracle's LSD on the same pixels), and the batch status must be clean: a partner position outside its segment would set
status bit 32 (seed_sort_impl.inc, the check in the swap pass) and make last_batch_status() raise.
"""
import importlib

import numpy as np
import pytest
import torch

import oracle_lib as O
from plp import plp, synth

pytestmark = pytest.mark.gpu
rs = importlib.import_module("structure-plp-slam_amd.replay_step")


def defined_seed(scaled):
    s = scaled.astype(np.int64)
    DA = s[1:, 1:] - s[:-1, :-1]; BC = s[:-1, 1:] - s[1:, :-1]
    gx = DA + BC; gy = DA - BC
    d = np.zeros(scaled.shape, bool)
    d[:-1, :-1] = ~(np.sqrt((gx * gx + gy * gy) / 4.0) <= 2.0 / np.sin(np.pi * 22.5 / 180))
    return d.ravel()


def test_two_hundred_overlapped_steps_leave_the_seed_order_of_std_sort():
    B, K, rows, cols, uniq, n_line = 1024, 1000, 480, 640, 32, 2
    frames = synth.replay(4321, uniq, rows, cols)
    want = []
    for f in frames:
        ora = O.LineOracle(f, stable_order=False)
        want.append(np.asarray(ora.order)[defined_seed(ora.scaled)[ora.order]].astype(np.int32))
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(frames).to(dev).repeat(B // uniq, 1, 1).contiguous()
    ts = rs.tracker_step(plp, B, K, rows, cols, n_line=n_line, seed_order=plp.SEED_ORDER_LIBSTDCXX)
    assert len(ts.lts) == n_line and B // n_line > 256, "sub-blocks above 256 frames take the 4-wave configuration of the sort"
    per = B // n_line
    rng = np.random.default_rng(11)
    n_checked = 0
    for it in range(100):
        ts.step(d_frames); ts.step(d_frames)          # two steps in flight: the second one's extractors run beside the first one's matchers
        torch.cuda.synchronize(dev)
        ts.last_batch_status()                         # raises on any status bit (32 = a partner position outside its segment, or a queue overflow)
        for b in rng.choice(B, 32, replace=False):
            lt, local = ts.lts[int(b) // per], int(b) % per
            got = lt.debug_read(lt.DBG_ORDER, local)
            assert np.array_equal(got, want[int(b) % uniq]), f"step pair {it}, frame {b}: seed order differs from std::sort"
            n_checked += 1
    assert n_checked == 3200
