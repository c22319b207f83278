#!/usr/bin/env python3
"""Dev-time: derive small gray test frames from the reference's own test images
(test/data/equirectangular_image_00{1,2}.jpg, used by
test/PLPSLAM/feature/orb_extractor.cc) and store them as PNG under tests/golden/.
The reference tree is absent on the GPU box, so the derived frames are committed.
Recipe: gray = PIL 'L' conversion, 2x box downscale, fixed crops.
"""
from PIL import Image
import numpy as np, pathlib
out = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"
out.mkdir(parents=True, exist_ok=True)
for idx, name in enumerate(["equirectangular_image_001.jpg", "equirectangular_image_002.jpg"], 1):
    im = Image.open(f"/root/reference/test/data/{name}").convert("L")
    a = np.asarray(im, dtype=np.uint16)
    half = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) // 4).astype(np.uint8)  # 480x960
    Image.fromarray(half[:, 160:800]).save(out / f"equirect{idx}_640x480.png", optimize=True)
    # full-res crop keeps fine texture (more corners / lines)
    full = np.asarray(im, dtype=np.uint8)
    Image.fromarray(full[240:720, 600:1240]).save(out / f"equirect{idx}_crop_640x480.png", optimize=True)
print(sorted(p.name for p in out.glob("*.png")))
