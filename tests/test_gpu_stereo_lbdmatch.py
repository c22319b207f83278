"""GPU parity of match::stereo::compute and BinaryDescriptorMatcher::match (exact 1-NN) against the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from plp import plp, synth

pytestmark = pytest.mark.gpu


def stereo_pair(seed, rows=480, cols=752):
    """left = synthetic canvas, right = left seen with disparity d(y) = 8 + round(4 sin(y/60)) px (scene moves left)"""
    wide = synth.canvas(seed, rows, cols + 32)
    left = np.ascontiguousarray(wide[:, 16:16 + cols])
    right = np.empty_like(left)
    for y in range(rows):
        d = 8 + int(round(4 * np.sin(y / 60.0)))
        right[y] = wide[y, 16 + d:16 + d + cols]
    return left, right


@pytest.mark.parametrize("seed,K", [(3, 1000), (4, 2000)])
def test_stereo_compute_matches_oracle(seed, K):
    left, right = stereo_pair(seed)
    ol, orr = O.OrbOracle(K), O.OrbOracle(K)
    kl, dl = ol.extract(left); kr, dr = orr.extract(right)
    fxb, tb = 435.2 * 0.11 * 10, 0.11 * 10      # focal_x_baseline / true_baseline -> max disparity 435 px
    want_x, want_d = O.stereo_compute(ol, orr, kl, kr, dl, dr, fxb, tb)
    el, er = plp.orb_extractor(K), plp.orb_extractor(K)
    gkl, gdl = el.extract(left); gkr, gdr = er.extract(right)
    assert np.array_equal(gkl, kl) and np.array_equal(gkr, kr)
    got_x, got_d = el.stereo_compute(er, gkl, gkr, gdl, gdr, fxb, tb)
    assert (want_x > 0).sum() > 100, "vacuous test"
    assert np.array_equal(got_x > 0, want_x > 0)
    assert np.abs(got_x - want_x).max() <= 1e-4 and np.abs(got_d - want_d).max() <= 1e-4 * np.abs(want_d).max()
    assert np.array_equal(got_x, want_x) and np.array_equal(got_d, want_d)      # identical in practice
    # small baseline: the disparity window [x - fxb/tb, x] excludes most candidates
    want_x, want_d = O.stereo_compute(ol, orr, kl, kr, dl, dr, 9.5, 1.0)
    got_x, got_d = el.stereo_compute(er, gkl, gkr, gdl, gdr, 9.5, 1.0)
    assert np.array_equal(got_x, want_x) and np.array_equal(got_d, want_d)


def test_stereo_no_right_keypoints():
    left, _ = stereo_pair(5, 480, 640)
    flat = np.full_like(left, 100)
    el, er = plp.orb_extractor(500), plp.orb_extractor(500)
    kl, dl = el.extract(left); kr, dr = er.extract(flat)
    assert len(kr) == 0
    x, d = el.stereo_compute(er, kl, kr, dl, dr, 400.0, 1.0)
    assert (x == -1).all() and (d == -1).all()


@pytest.mark.parametrize("seed", range(4))
def test_lbd_match_1nn_with_ties(seed):
    rng = np.random.default_rng(60 + seed)
    mt = plp.matcher()
    for nq, nt, words, flips in [(50, 70, 0, 0), (200, 300, 6, 2), (400, 120, 3, 1), (1, 1, 1, 0), (64, 500, 12, 3)]:
        if words:
            vocab = rng.integers(0, 256, (words, 32), dtype=np.uint8)
            t = vocab[rng.integers(0, words, nt)].copy()
            q = vocab[rng.integers(0, words, nq)].copy()
            for arr in (t, q):     # a few random bit flips: many exact distance ties with different discovery bytes / patterns
                for _ in range(flips):
                    r = np.arange(len(arr)); arr[r, rng.integers(0, 32, len(arr))] ^= (np.uint8(1) << rng.integers(0, 8, len(arr)).astype(np.uint8))
        else:
            t = rng.integers(0, 256, (nt, 32), dtype=np.uint8); q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        wi, wd = O.lbd_match_1nn(q, t)
        gi, gd = mt.lbd_match_1nn(q, t)
        assert np.array_equal(gd, wd)
        assert np.array_equal(gi, wi), (nq, nt, words)
        bf = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
        ok = wd <= 128
        assert np.array_equal(wd[ok], bf.min(1)[ok])


def test_lbd_match_on_real_stereo_lines():
    left, right = stereo_pair(9, 480, 640)
    a, b = O.LineOracle(left), O.LineOracle(right)
    assert len(a.lbd) > 10 and len(b.lbd) > 10
    wi, wd = O.lbd_match_1nn(a.lbd, b.lbd)
    gi, gd = plp.matcher().lbd_match_1nn(a.lbd, b.lbd)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    assert (wd < 30).sum() > 5      # the stereo line association of data/frame.cc:505 would keep these
