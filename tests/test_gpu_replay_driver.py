"""End-to-end replay of a (synthetic) TUM-layout RGB-D sequence through the device path: files on disk -> association ->
decode -> grayscale / depth conversion -> ORB + LSD/LBD -> undistortion / stereo from depth -> last-frame matcher.  Every
stage's output is compared with the oracle chained the same way on the host."""
import importlib

import numpy as np
import pytest

import oracle_lib as O
from plp import plp, synth

pytestmark = pytest.mark.gpu
drv = importlib.import_module("structure-plp-slam_amd.replay_driver")


def write_sequence(root, n):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    gray = synth.replay(77, n, 480, 640)
    (root / "rgb").mkdir(); (root / "depth").mkdir()
    rgb_lines, depth_lines = ["# color", "# file", "# timestamp filename"], ["# depth", "# file", "# timestamp filename"]
    colors, depths = [], []
    for i in range(n):
        t = 1305031102.175304 + i / 30.0
        # a colour image whose channels differ, so that the channel order matters to the grayscale conversion
        rgbimg = np.stack([gray[i], np.roll(gray[i], 3, 1), 255 - gray[i] // 2], 2).astype(np.uint8)
        d = (rng.uniform(0.4, 6.0, (480, 640)) * 5000).astype(np.uint16)
        d[rng.uniform(size=d.shape) < 0.2] = 0
        Image.fromarray(rgbimg).save(root / "rgb" / f"{t:.6f}.png")
        Image.fromarray(d).save(root / "depth" / f"{t + 0.004:.6f}.png")
        rgb_lines.append(f"{t:.6f} rgb/{t:.6f}.png"); depth_lines.append(f"{t + 0.004:.6f} depth/{t + 0.004:.6f}.png")
        colors.append(rgbimg); depths.append(d)
    (root / "rgb.txt").write_text("\n".join(rgb_lines) + "\n"); (root / "depth.txt").write_text("\n".join(depth_lines) + "\n")
    return colors, depths


def test_tum_layout_sequence_replays_like_the_oracle_chain(tmp_path):
    n = 5
    colors, depths = write_sequence(tmp_path, n)
    cam = dict(drv.FR3, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, k1=0.262383, k2=-0.953104, p1=-0.005358, p2=0.002628,
               k3=1.163314)                                   # freiburg1: with distortion
    stats, ts = drv.replay_sequence(str(tmp_path), batch=3, camera=cam)          # two batches: 3 + 2 frames, the seam is matched too
    assert len(ts) == n and all(abs(ts[i] - (1305031102.175304 + i / 30.0 + 0.002)) < 1e-6 for i in range(n))
    grid = plp.make_grid(640, 480)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    cam10 = tuple(cam[k] for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "focal_x_baseline"))
    prev = None
    for i in range(n):
        bgr = np.ascontiguousarray(colors[i][:, :, ::-1])                         # what cv::imread returns
        g = np.zeros((480, 640), np.uint8)
        O._call("oracle_convert_to_grayscale", [bgr, 480, 640, 3, 0, g])         # Camera.color_order RGB -> cv::COLOR_RGB2GRAY on it
        kps, desc = O.OrbOracle(1000).extract(g)
        lo = O.LineOracle(g)
        dep = (depths[i].astype(np.float32) * np.float32(1.0 / 5000.0)).astype(np.float32)
        pe = O.post_extract(cam10, kps, dep)
        assert stats["n_keypts"][i] == len(kps) and stats["n_keylines"][i] == len(lo.keylsd)
        assert stats["n_depth"][i] == int((pe["depths"] > 0).sum())
        if prev is None:
            assert stats["n_matches"][i] == -1
        else:
            pu, pdsc = prev
            m = len(pu)
            want, wn = O.match_current_and_last(O.grid6(grid), pe["undist_keypts"], desc, np.full(len(kps), -1, np.float32), np.zeros(len(kps), np.uint8),
                                                sf, np.ones(m, np.uint8), np.stack([pu["x"], pu["y"]], 1), np.full(m, -1, np.float32),
                                                pu["octave"].astype(np.int32), pu["angle"], pdsc, np.ones(m, np.uint8), 30.0, 0, True)
            assert stats["n_matches"][i] == wn, (i, stats["n_matches"][i], wn)
        prev = (pe["undist_keypts"], desc)
    assert (stats["n_matches"][1:] > 100).all()
