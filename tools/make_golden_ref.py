#!/usr/bin/env python3
"""Dev-time (this container only: needs oracle/_ref/libplpref.so, i.e. /root/reference at build time): run the REFERENCE'S
OWN ORB extractor sources (feature/orb_extractor.cc, orb_extractor_node.cc, orb_params.cc, compiled unmodified by
oracle/ref_build.sh) on the committed fixture frames and store their key points and descriptors as
tests/golden/ref_orb.npz.  The reference tree and, after a fresh build elsewhere, oracle/_ref do not exist on the GPU
box: the stored vectors are what the CPU suite checks the oracle against and the GPU suite checks the HIP path against.
Also (oracle/_ref/libplpref2.so = the reference's matcher / line / LBD / MIH / stereo sources, oracle/ref_driver2.cpp):
  tests/golden/ref_match.npz   random problems for every array-form matcher (tests/match_cases.py) with the REFERENCE'S answers
  tests/golden/ref_line.npz    LineFeatureTracker::extract_LSD_LBD of the reference build on the fixture frames and two synthetic ones
  tests/golden/ref_stereo.npz  match::stereo::compute of the reference build on two synthetic stereo pairs
    python tools/make_golden_ref.py
"""
import pathlib
import sys

import numpy as np
from PIL import Image

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_oracle_vs_ref as R   # noqa: E402  (ctypes binding of oracle/_ref/libplpref.so)

CASES = [("equirect1_640x480", 1000), ("equirect1_crop_640x480", 1000), ("equirect2_640x480", 1000), ("equirect2_crop_640x480", 1000),
         ("equirect1_crop_640x480", 2000), ("equirect2_640x480", 500)]


def golden_match(seeds=(1, 2, 3, 4), scale=0.45):
    """arguments are stored once per distinct array (many matchers share the frame); scalars as 0-d arrays"""
    import hashlib
    import oracle_lib as O
    import match_cases as MC
    blobs, index, out = [], {}, {}

    def ref_of(a):
        a = np.ascontiguousarray(a)
        key = (a.dtype.str if a.dtype.names is None else "kp" + str(a.dtype.itemsize), a.shape, hashlib.sha1(a.tobytes()).hexdigest())
        if key not in index:
            index[key] = len(blobs)
            blobs.append(a)
        return index[key]
    n_d3 = 0
    for seed in seeds:
        rng = np.random.default_rng(424_000 + seed)
        for label, fn, args in MC.matcher_cases(rng, scale):
            checked = isinstance(args[-1], (bool, np.bool_)) and bool(args[-1])
            if checked:                       # D3: count the problems whose orientation check is decided by the order std::sort gives equal bins
                getattr(O, fn)(*args)
                n_d3 += int(O.angle_checker_last_tie())
            with O.reference():
                res = getattr(O, fn)(*args)
            res = res if isinstance(res, tuple) else (res,)
            base = f"s{seed}__{label}"
            out[base + "__fn"] = np.array(fn)
            spec = []
            for i, a in enumerate(args):
                if isinstance(a, np.ndarray):
                    spec.append(ref_of(a))
                else:
                    spec.append(-1)
                    out[f"{base}__arg{i}"] = np.array(a)
            out[base + "__args"] = np.array(spec, np.int32)
            for k, r in enumerate(res):
                out[f"{base}__out{k}"] = np.asarray(r)
            if label == "lbd_1nn":            # queries the reference leaves undefined (nothing within the MIH reach)
                out[base + "__defined"] = (getattr(O, fn)(*args)[0] >= 0)
    for k, b in enumerate(blobs):
        if b.dtype.names is not None:
            out[f"blob{k}__rec{b.dtype.itemsize}"] = b.view(np.uint8).reshape(len(b), b.dtype.itemsize)
        else:
            out[f"blob{k}"] = b
    np.savez_compressed(ROOT / "tests" / "golden" / "ref_match.npz", **out)
    print("ref_match.npz:", len(seeds), "seeds,", len(blobs), "arrays,", n_d3, "orientation checks decided by ties (D3)")


def golden_line_and_stereo():
    import importlib
    import oracle_lib as O
    synth = importlib.import_module("structure-plp-slam_amd.synth")
    out = {}
    frames = {n: np.asarray(Image.open(ROOT / "tests" / "golden" / f"{n}.png").convert("L"), dtype=np.uint8)
              for n in ("equirect1_640x480", "equirect1_crop_640x480", "equirect2_640x480", "equirect2_crop_640x480")}
    frames["canvas7_480x640"] = synth.canvas(7, 480, 640)
    frames["canvas3_376x1241"] = synth.canvas(3, 376, 1241)
    for name, img in frames.items():
        kl, lbd, fn = O.ref_line_extract(img)
        out[name + "__kl"] = kl.view(np.uint8).reshape(len(kl), 68); out[name + "__lbd"] = lbd; out[name + "__fn"] = fn
        print(name, len(kl), "key lines")
    np.savez_compressed(ROOT / "tests" / "golden" / "ref_line.npz", **out)
    out = {}
    for seed, K in ((3, 1000), (4, 2000)):
        rows, cols = 480, 752
        wide = synth.canvas(seed, rows, cols + 32)
        left = np.ascontiguousarray(wide[:, 16:16 + cols]); right = np.empty_like(left)
        for y in range(rows):
            d = 8 + int(round(4 * np.sin(y / 60.0)))
            right[y] = wide[y, 16 + d:16 + d + cols]
        ol, orr = O.OrbOracle(K), O.OrbOracle(K)
        kl, dl = ol.extract(left); kr, dr = orr.extract(right)
        tb = ol.tables()
        lv_l = [left] + [ol.level_image(l) for l in range(1, 8)]; lv_r = [right] + [orr.level_image(l) for l in range(1, 8)]
        for tag, (fxb, b) in (("wide", (435.2 * 1.1, 1.1)), ("narrow", (9.5, 1.0))):
            xr, dp = O.ref_stereo_compute(lv_l, lv_r, kl, kr, dl, dr, tb["scale_factors"], tb["inv_scale_factors"], fxb, b)
            out[f"seed{seed}_K{K}_{tag}__x_right"] = xr; out[f"seed{seed}_K{K}_{tag}__depth"] = dp
            out[f"seed{seed}_K{K}_{tag}__params"] = np.array([fxb, b], np.float32)
            print("stereo", seed, K, tag, int((xr > 0).sum()), "matches")
    np.savez_compressed(ROOT / "tests" / "golden" / "ref_stereo.npz", **out)


def main():
    golden_match()
    golden_line_and_stereo()
    out = {}
    for name, K in CASES:
        img = np.asarray(Image.open(ROOT / "tests" / "golden" / f"{name}.png").convert("L"), dtype=np.uint8)
        kps, desc = R.ref_extract(img, K)
        out[f"{name}__K{K}__kps"] = kps.view(np.uint8).reshape(len(kps), 28)
        out[f"{name}__K{K}__desc"] = desc
        print(name, K, len(kps))
    np.savez_compressed(ROOT / "tests" / "golden" / "ref_orb.npz", **out)


if __name__ == "__main__":
    main()
