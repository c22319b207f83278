"""Shared builders of matcher test problems (used by the GPU parity tests and the bench)."""
import numpy as np
import oracle_lib as O


def features_from_oracle(img, K=1000):
    kps, desc = O.OrbOracle(K).extract(img)
    return kps, desc


def make_queries_from_prev(prev_kps, prev_desc, rng, shift=(3.0, 0.0), jitter=1.5, drop=0.1):
    """queries = the previous frame's key points 'reprojected' into the current frame"""
    m = len(prev_kps)
    reproj = np.stack([prev_kps["x"] + np.float32(shift[0]), prev_kps["y"] + np.float32(shift[1])], 1).astype(np.float32)
    reproj += rng.normal(0, jitter, reproj.shape).astype(np.float32)
    valid = (rng.uniform(size=m) > drop).astype(np.uint8)
    return dict(q_valid=valid, q_reproj=reproj, q_x_right=np.full(m, -1, np.float32), q_level=prev_kps["octave"].astype(np.int32),
                q_angle=prev_kps["angle"].astype(np.float32), q_desc=prev_desc.copy(), q_has_obs=np.ones(m, np.uint8))


def random_problem(rng, n, m, n_words=0, cols=640, rows=480, stereo=False):
    """fully synthetic problem; n_words>0 draws descriptors from a tiny vocabulary -> many exact distance ties and conflicts"""
    kps = np.zeros(n, O.KP_DTYPE)
    kps["x"] = rng.uniform(0, cols, n).astype(np.float32); kps["y"] = rng.uniform(0, rows, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n); kps["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    if n_words:
        vocab = rng.integers(0, 256, (n_words, 32), dtype=np.uint8)
        desc = vocab[rng.integers(0, n_words, n)].copy()
        flip = rng.integers(0, 32, n)
        desc[np.arange(n), flip] ^= (1 << rng.integers(0, 8, n)).astype(np.uint8) * (rng.uniform(size=n) < 0.5)
    else:
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    src = rng.integers(0, n, m)
    q_desc = desc[src].copy()
    noise_bits = rng.integers(0, 40, m)
    for i in range(m):
        bits = rng.choice(256, noise_bits[i], replace=False)
        for b in bits:
            q_desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    reproj = np.stack([kps["x"][src], kps["y"][src]], 1) + rng.normal(0, 4, (m, 2))
    t = dict(t_kps=kps, t_desc=desc, t_x_right=(rng.uniform(10, 600, n).astype(np.float32) * (rng.uniform(size=n) < 0.5) - (rng.uniform(size=n) < 0.2)).astype(np.float32) if stereo else np.full(n, -1, np.float32),
             t_occupied=(rng.uniform(size=n) < 0.15).astype(np.uint8))
    q = dict(q_valid=(rng.uniform(size=m) > 0.1).astype(np.uint8), q_reproj=reproj.astype(np.float32),
             q_x_right=(reproj[:, 0] - rng.uniform(0, 30, m)).astype(np.float32) if stereo else np.full(m, -1, np.float32),
             q_level=np.clip(kps["octave"][src] + rng.integers(-1, 2, m), 0, 7).astype(np.int32),
             q_angle=(kps["angle"][src] + rng.normal(0, 20, m)).astype(np.float32) % np.float32(360), q_desc=q_desc,
             q_has_obs=(rng.uniform(size=m) > 0.05).astype(np.uint8))
    return t, q
