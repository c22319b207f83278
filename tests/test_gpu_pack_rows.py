"""plp_pack_rows_device: the live rows of padded per-frame arrays packed back to back (the host boundary of the batched replay)."""
import importlib

import numpy as np
import pytest
import torch

from plp import plp

replay = importlib.import_module("structure-plp-slam_amd.replay")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,cap,row", [(1, 7, 4), (5, 300, 28), (2048, 64, 32), (1500, 17, 68), (3, 2064, 24)])
def test_pack_rows_equals_numpy(B, cap, row):
    rng = np.random.default_rng(B * 1000 + cap)
    src = rng.integers(0, 256, (B, cap, row), dtype=np.uint8)
    counts = rng.integers(-2, cap + 5, B).astype(np.int32)       # below 0 and above cap: clamped
    counts[rng.integers(0, B)] = 0
    dev = torch.device("cuda", 0)
    d_src, d_cnt = torch.from_numpy(src).to(dev), torch.from_numpy(counts).to(dev)
    d_dst = torch.full((B * cap * row,), 0xAB, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)
    assert replay.pack_rows(plp, d_src, d_cnt, d_dst, d_off, True, st) == row
    d_dst2 = torch.zeros_like(d_dst)
    replay.pack_rows(plp, d_src, d_cnt, d_dst2, d_off, False, st)          # offsets reused
    torch.cuda.synchronize()
    c = np.clip(counts, 0, cap)
    want_off = np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
    assert np.array_equal(d_off.cpu().numpy(), want_off)
    want = np.concatenate([src[b, :c[b]] for b in range(B)]).reshape(-1)
    got = d_dst.cpu().numpy()
    assert np.array_equal(got[:len(want)], want) and np.array_equal(d_dst2.cpu().numpy()[:len(want)], want)
    assert (got[len(want):] == 0xAB).all()                                 # nothing is written behind the packed rows
