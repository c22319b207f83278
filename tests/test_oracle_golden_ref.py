"""The oracle against the committed answers of the REFERENCE BUILD (tests/golden/ref_match.npz, ref_line.npz, ref_stereo.npz,
written by tools/make_golden_ref.py from oracle/_ref/libplpref2.so): this check travels -- it needs neither /root/reference nor
oracle/_ref, so it also runs on the GPU box, next to tests/test_gpu_golden_ref.py which holds the HIP path to the same files."""
import pathlib

import numpy as np

import oracle_lib as O
import match_cases as MC
from plp import synth

G = pathlib.Path(__file__).resolve().parent / "golden"


def test_oracle_matchers_equal_the_committed_reference_answers():
    cases = MC.load_golden_match()
    assert len(cases) >= 60 and len({c[1] for c in cases}) == 15
    for seed, label, fn, args, outs, extras in cases:
        got = getattr(O, fn)(*args)
        got = got if isinstance(got, tuple) else (got,)
        for g, w in zip(got, outs):
            if label == "lbd_1nn":
                ok = extras["defined"]
                assert np.array_equal(np.asarray(g)[ok], np.asarray(w)[ok]), (seed, label)
            elif isinstance(w, np.ndarray):
                assert np.array_equal(g, w), (seed, label)
            else:
                assert int(g) == w, (seed, label)


def golden_frames():
    from PIL import Image
    fr = {n: np.asarray(Image.open(G / f"{n}.png").convert("L"), dtype=np.uint8)
          for n in ("equirect1_640x480", "equirect1_crop_640x480", "equirect2_640x480", "equirect2_crop_640x480")}
    fr["canvas7_480x640"] = synth.canvas(7, 480, 640)
    fr["canvas3_376x1241"] = synth.canvas(3, 376, 1241)
    return fr


def check_lines(name, kl, lbd, fn, z):
    """key lines / LBD / line functions against the reference build's: bit-exact, KeyLine::angle to one ulp (atan2f, D2)"""
    wkl = np.ascontiguousarray(z[name + "__kl"]).view(O.KL_DTYPE).reshape(-1)
    assert len(kl) == len(wkl), name
    for f in O.KL_DTYPE.names:
        if f == "angle":
            assert np.all(np.abs(kl[f] - wkl[f]) <= np.spacing(np.abs(wkl[f]).astype(np.float32))), (name, f)
        else:
            assert np.array_equal(kl[f], wkl[f]), (name, f)
    assert np.array_equal(lbd, z[name + "__lbd"]) and np.array_equal(fn, z[name + "__fn"]), name


def test_oracle_lines_equal_the_committed_reference_answers():
    z = np.load(G / "ref_line.npz")
    for name, img in golden_frames().items():
        lo = O.LineOracle(img)
        check_lines(name, lo.keylsd, lo.lbd, lo.linefn, z)


def stereo_pairs():
    for seed, K in ((3, 1000), (4, 2000)):
        rows, cols = 480, 752
        wide = synth.canvas(seed, rows, cols + 32)
        left = np.ascontiguousarray(wide[:, 16:16 + cols]); right = np.empty_like(left)
        for y in range(rows):
            d = 8 + int(round(4 * np.sin(y / 60.0)))
            right[y] = wide[y, 16 + d:16 + d + cols]
        yield seed, K, left, right


def test_oracle_stereo_equals_the_committed_reference_answers():
    z = np.load(G / "ref_stereo.npz")
    for seed, K, left, right in stereo_pairs():
        ol, orr = O.OrbOracle(K), O.OrbOracle(K)
        kl, dl = ol.extract(left); kr, dr = orr.extract(right)
        for tag in ("wide", "narrow"):
            fxb, tb = (float(v) for v in z[f"seed{seed}_K{K}_{tag}__params"])
            xr, dp = O.stereo_compute(ol, orr, kl, kr, dl, dr, fxb, tb)
            assert np.array_equal(xr, z[f"seed{seed}_K{K}_{tag}__x_right"]) and np.array_equal(dp, z[f"seed{seed}_K{K}_{tag}__depth"])
