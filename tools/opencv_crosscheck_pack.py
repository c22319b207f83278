#!/usr/bin/env python3
"""Packs the .npy files written by tools/opencv_crosscheck (a run against a real OpenCV, see that file's header) into
tests/golden/opencv_crosscheck.npz, the file tests/test_oracle_opencv_crosscheck.py looks for.
    python tools/opencv_crosscheck_pack.py <out_dir>"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]


def main():
    d = pathlib.Path(sys.argv[1])
    arrays = {p.stem: np.load(p) for p in sorted(d.glob("*.npy"))}
    assert "opencv_version" in arrays, "not an opencv_crosscheck output directory"
    np.savez_compressed(ROOT / "tests" / "golden" / "opencv_crosscheck.npz", **arrays)
    print(len(arrays), "arrays, OpenCV", bytes(arrays["opencv_version"]).decode())


if __name__ == "__main__":
    main()
