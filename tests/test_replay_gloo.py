"""Multi-rank path on CPU: world_size 2, gloo.  Checks the frame sharding and the halo exchange that the
batched-replay matcher needs at block boundaries (bench.py uses the same code with backend nccl = RCCL)."""
import os
import socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plp import plp  # noqa: F401  (import path set-up)
import importlib
replay = importlib.import_module("structure-plp-slam_amd.replay")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = replay.frame_block(rank, world, n_frames)
    # per-frame "features": frame index in every element, so provenance is checkable
    kps = torch.arange(a, b, dtype=torch.float32)[:, None, None].repeat(1, 5, 7)
    desc = torch.arange(a, b, dtype=torch.int64)[:, None, None].repeat(1, 5, 32).to(torch.uint8)
    cnt = torch.arange(a, b, dtype=torch.int32)
    hk, hd, hc = replay.exchange_halo([kps, desc, cnt], halo=2)
    ak, ad, ac = replay.exchange_halo([kps, desc, cnt], halo=2, mode="allgather")      # the two transports agree
    assert torch.equal(hk, ak) and torch.equal(hd, ad) and torch.equal(hc, ac)
    assert hk.dtype == kps.dtype and hd.dtype == desc.dtype and hc.dtype == cnt.dtype and hk.shape == (2, 5, 7)
    # the step's form (replay.halo_exchanger: persistent buffers, ONE exchange for all arrays) fills rows 0..1 of [halo + B, ...] arrays with the same tails, both modes
    for mode in ("ring", "allgather"):
        bufs = [torch.cat([torch.zeros_like(t[:2]), t], 0) for t in (kps, desc, cnt)]
        ex = replay.halo_exchanger(bufs, halo=2, mode=mode)
        assert ex.record_bytes == 5 * 7 * 4 + 5 * 32 + 4 and ex.bytes_per_step == 2 * ex.record_bytes
        ex(bufs)
        assert torch.equal(bufs[0][:2], hk) and torch.equal(bufs[1][:2], hd) and torch.equal(bufs[2][:2], hc), mode
        assert torch.equal(bufs[0][2:], kps) and torch.equal(bufs[2][2:], cnt)
    full = replay.with_halo(cnt, hc)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the max-over-ranks timing reduction of bench.py
    q.put((rank, a, b, hk[:, 0, 0].tolist(), hd[:, 0, 0].tolist(), hc.tolist(), full.tolist(), t.item()))
    dist.destroy_process_group()


def test_sharding_and_halo_exchange_world2():
    world, n_frames = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    blocks = [(a, b) for _, a, b, *_ in res]
    assert blocks == [(0, 6), (6, 11)]                                  # contiguous, covering, balanced
    (_, _, _, hk0, hd0, hc0, full0, t0), (_, _, _, hk1, hd1, hc1, full1, t1) = res
    assert hc0 == [9, 10] and hk0 == [9.0, 10.0] and hd0 == [9, 10]      # rank 0 gets the last rank's tail (circular replay)
    assert hc1 == [4, 5] and hk1 == [4.0, 5.0]                            # rank 1 gets rank 0's tail
    assert full1 == [4, 5, 6, 7, 8, 9, 10]
    assert t0 == t1 == 2.0


def test_single_rank_halo_is_circular():
    cnt = torch.arange(5)
    (h,) = replay.exchange_halo([cnt], halo=2)
    assert h.tolist() == [3, 4]
    buf = torch.cat([torch.zeros(2, dtype=cnt.dtype), cnt])
    replay.halo_exchanger([buf], halo=2)([buf])
    assert buf.tolist() == [3, 4, 0, 1, 2, 3, 4]
    assert replay.frame_block(0, 1, 7) == (0, 7)
    assert [replay.frame_block(r, 3, 10) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
