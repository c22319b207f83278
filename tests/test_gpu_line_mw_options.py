"""The several-waves region growing (the latency path, k_lsd_grow_mw) reads one policy switch from the environment once per process: PLP_LSD_MW_POLICY, what a
helper makes of a FINISHED region's claim (0 default: judged like a growing one's; 1: by the position range of that helper's finished regions).  It only changes
how much speculation is useful -- the main wave validates every result against the committed map -- so either setting must give the oracle's lines, bit for
bit.  One subprocess per setting.  (Round 5 tried two more policies and a park / resume of attempts that run into an earlier seed's growing region: exact as
well, no faster -- profiles/r05_latency_path.md, profiles/r05_latency_path_series.patch; the protocol with parking: tests/test_spec_grow_model.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SNIPPET = """
import sys, numpy as np
sys.path[:0] = [%r, %r]
from PIL import Image
import test_gpu_line as T
from plp import synth
n = len(T.compare(np.asarray(Image.open(%r)), grow_waves=(0, 3, 5)))
n += len(T.compare(synth.canvas(77, 480, 640), grow_waves=(0, 8)))
rng = np.random.default_rng(5)
T.compare(rng.integers(0, 256, (240, 320), dtype=np.uint8), grow_waves=(0,))
assert n > 20
print("lines", n)
"""


@pytest.mark.parametrize("policy", [0, 1])
def test_claim_policies_give_the_same_lines(policy, golden_dir):
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PLP_LSD_MW_POLICY=str(policy))
    code = SNIPPET % (here, os.path.dirname(here), str(golden_dir / "equirect1_640x480.png"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "lines" in r.stdout
