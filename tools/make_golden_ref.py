#!/usr/bin/env python3
"""Dev-time (this container only: needs oracle/_ref/libplpref.so, i.e. /root/reference at build time): run the REFERENCE'S
OWN ORB extractor sources (feature/orb_extractor.cc, orb_extractor_node.cc, orb_params.cc, compiled unmodified by
oracle/ref_build.sh) on the committed fixture frames and store their key points and descriptors as
tests/golden/ref_orb.npz.  The reference tree and, after a fresh build elsewhere, oracle/_ref do not exist on the GPU
box: the stored vectors are what the CPU suite checks the oracle against and the GPU suite checks the HIP path against.
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


def main():
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
