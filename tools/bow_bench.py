"""Times plp_bow_transform_device on B frames of extractor output against a full k = 10, L = 6 tree (1,111,111 nodes, the
shape of the ORB vocabulary; random node descriptors) and prints ms per batch for the two kernels' sum."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")


def main(B=2048, K=1000):
    rng = np.random.default_rng(0)
    k, L = 10, 6
    n = (k ** (L + 1) - 1) // (k - 1)
    parents = np.concatenate([[-1], (np.arange(1, n) - 1) // k])
    is_leaf = np.arange(n) >= (k ** L - 1) // (k - 1)
    descs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weights = np.where(is_leaf, rng.uniform(0.5, 12.0, n), 0.0)
    v = plp.bow_vocabulary(L, parents, is_leaf, descs, weights)
    dev = torch.device("cuda", 0)
    uniq = 64
    frames = torch.from_numpy(synth.replay(5, uniq, 480, 640)).to(dev).repeat(B // uniq, 1, 1).contiguous()
    ex = plp.orb_extractor(K)
    cap = 2 * K + 64
    d_kps = torch.empty((B, cap, 28), dtype=torch.uint8, device=dev); d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch(frames, d_kps, d_desc, d_cnt)
    torch.cuda.synchronize()
    for it in range(3):
        t0 = time.perf_counter()
        out = v.transform_device(d_desc, d_cnt, 4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
    print(f"bow transform: B={B} mean features {d_cnt.float().mean().item():.0f} nodes {n}: {dt:.2f} ms per batch, "
          f"{B / dt * 1e3:.0f} frames/s, mean distinct words {out['n_bow'].float().mean().item():.0f}")


if __name__ == "__main__":
    main()
