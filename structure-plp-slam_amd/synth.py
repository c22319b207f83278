"""Deterministic synthetic gray frames (no dataset is reachable offline).

`canvas(seed, H, W)` = 5 octaves of bilinearly up-sampled uniform noise plus 40 random
dark/bright rectangles (corner- and line-rich, like indoor TUM scenes); `replay(...)`
slides a window over a wider canvas by 3 px/frame, mimicking a camera pan.
"""
import numpy as np


def _upsample_bilinear(g, H, W):
    gh, gw = g.shape
    ys = np.linspace(0, gh - 1, H)
    xs = np.linspace(0, gw - 1, W)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, gh - 1); x1 = np.minimum(x0 + 1, gw - 1)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x1]; c = g[y1][:, x0]; d = g[y1][:, x1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def canvas(seed, H, W, n_rect=40):
    rng = np.random.default_rng(seed)
    img = np.zeros((H, W), np.float64)
    amp = 1.0
    for o in range(5):
        gh, gw = max(2, H >> (6 - o)), max(2, W >> (6 - o))
        img += amp * _upsample_bilinear(rng.uniform(0, 1, (gh, gw)), H, W)
        amp *= 0.55
    img = (img - img.min()) / (img.max() - img.min())
    img = 40 + 170 * img
    for _ in range(n_rect):
        w = int(rng.integers(12, max(13, W // 4))); h = int(rng.integers(12, max(13, H // 4)))
        x = int(rng.integers(0, W - w)); y = int(rng.integers(0, H - h))
        delta = float(rng.uniform(35, 90)) * (1 if rng.uniform() < 0.5 else -1)
        img[y:y + h, x:x + w] += delta
    img += rng.normal(0, 1.5, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def replay(seed, n_frames, H=480, W=640, step_px=3):
    """n_frames x H x W uint8: a window sliding `step_px` px per frame over a wide canvas."""
    span = step_px * (n_frames - 1)
    wide = canvas(seed, H, W + min(span, 4 * W))
    period = wide.shape[1] - W + 1
    out = np.empty((n_frames, H, W), np.uint8)
    for f in range(n_frames):
        x = (f * step_px) % period
        out[f] = wide[:, x:x + W]
    return out
