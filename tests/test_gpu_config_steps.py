"""GPU parity of the batched steps that tools/bench_configs.py times for BASELINE.json's configs[2], [3] and [4]
(structure-plp-slam_amd/config_steps.py): every frame of a small batch against the oracle chain -- the features of every extractor, stereo
x_right / depth and the LBD 1-NN association (config 2: data/frame.cc:277-281, match/stereo.cc:45-150), K = 4000 key points at 1241 x 376 and
the last-frame matcher (config 3), post-extract + plane colour vote + both last-frame matchers (config 4: planar_mapping_module.cc:185-345)."""
import importlib

import numpy as np
import pytest

import config_step_check as CC
from plp import plp, synth

pytestmark = pytest.mark.gpu
cs = importlib.import_module("structure-plp-slam_amd.config_steps")


def _dev():
    import torch
    return torch, torch.device("cuda", 0)


@pytest.mark.parametrize("K", [1000, 2000])
def test_config2_euroc_stereo_step(K):
    torch, dev = _dev()
    B = 8
    wide = torch.from_numpy(synth.replay(2, B, 480, 752 + 16)).to(dev)
    left, right = cs.stereo_pair_from_wide(wide, 752)
    st = cs.stereo_step(plp, B, K)
    st.run(left, right); st.run(left, right)              # twice: buffers reused as the bench reuses them
    torch.cuda.synchronize(); st.status()
    bad = CC.check_stereo(st, left.cpu().numpy(), right.cpu().numpy(), range(B))
    assert not bad, bad[:5]
    assert float((st.xr >= 0).float().sum(1).mean()) > 200, "vacuous: no stereo matches"


@pytest.mark.parametrize("K", [4000, 2000])
def test_config3_kitti_mono_step(K):
    torch, dev = _dev()
    B = 8
    fr = synth.replay(3, B, 376, 1241)
    d = torch.from_numpy(fr).to(dev)
    st = cs.mono_step(plp, B, K, 376, 1241)
    st.run(d); st.run(d)
    torch.cuda.synchronize(); st.status()
    bad = CC.check_mono(st, fr, range(B))
    assert not bad, bad[:5]
    assert float(st.n1.float().mean()) > 100, "vacuous: few matches"


def test_config4_icl_rgbd_plane_step():
    torch, dev = _dev()
    B = 8
    fr = synth.replay(4, B, 480, 640)
    depth, seg = cs.icl_inputs(4, B)
    depth[:, ::7, ::5] = 0.0                                 # missing depth somewhere
    st = cs.rgbd_plane_step(plp, B, 1000)
    args = (torch.from_numpy(fr).to(dev), torch.from_numpy(depth).to(dev), torch.from_numpy(seg).to(dev))
    st.run(*args); st.run(*args)
    torch.cuda.synchronize(); st.status()
    bad = CC.check_rgbd_plane(st, fr, depth, seg, range(B))
    assert not bad, bad[:5]
    slot = torch.arange(st.cap, device=dev)[None, :]
    assert float(((st.lab != 0) & (slot < st.c[:, None])).float().sum(1).mean()) > 50, "vacuous: no key point on a plane"
