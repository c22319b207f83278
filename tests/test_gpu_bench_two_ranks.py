"""bench.py at N = 2 on ONE GPU (PLP_BENCH_SHARE_GPU=1: both ranks on cuda:0 over gloo), launched the way the driver launches it: the line must carry
`verified_frames` and `verified_halo_rows` -- every rank re-derives frames of its own block (frames 0 and 1, whose predecessors arrived over the halo exchange,
among them) and its halo rows (against the predecessor RANK's last two frames, regenerated from that rank's seed) with the CPU oracle after the timed region.
With backend nccl (one GPU per rank) exactly this code runs on an 8-GPU node (tools/scale8.sh)."""
import json
import os
import pathlib
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("halo", ["ring", "allgather"])     # the step's one exchange as a neighbour shift (default) and as the all-gather north_star names (PLP_BENCH_HALO)
def test_two_rank_bench_line_is_verified_on_every_rank(halo):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, PLP_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", PLP_BENCH_HALO=halo)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "128", "--no-cpu-baseline", "--no-extras", "--verify", "8"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert j["verified_frames"] == 16, j["verified_frames"]          # 8 per rank
    assert j["verified_halo_rows"] == 4, j["verified_halo_rows"]      # 2 per rank
    assert j["value"] > 0
    assert j["config"]["halo_mode"] == halo and j["config"]["halo_bytes_per_rank_per_step"] > 2 * (2064 * 60 + 512 * 100)     # one record per frame: points AND lines
