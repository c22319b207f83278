#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: HIP path vs oracle on random frames / parameters / matcher problems.
   python tools/fuzz_gpu.py --seconds 240 [--seed N]      -> prints one summary line per family, exit code 1 on any mismatch"""
import argparse, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O
import match_cases as MC
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")


def rand_image(rng, h, w):
    kind = rng.integers(0, 4)
    if kind == 0:
        return synth.canvas(int(rng.integers(1 << 30)), h, w).astype(np.uint8)
    if kind == 1:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        return img if rng.uniform() < 0.3 else np.repeat(np.repeat(img[::4, ::4], 4, 0), 4, 1)[:h, :w].copy()
    fr = synth.replay(int(rng.integers(1 << 30)), 1, h, w)[0]
    if kind == 3:   # low contrast: threshold fallback cells
        fr = (fr.astype(np.float32) * rng.uniform(0.05, 0.4) + rng.uniform(0, 150)).astype(np.uint8)
    return fr


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120); ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    budget = a.seconds / 3
    bad = 0
    # ---- ORB
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        h, w = int(rng.integers(120, 720)), int(rng.integers(160, 1300))
        K = int(rng.choice([100, 500, 1000, 2000, 4000])); sfac = float(rng.choice([1.2, 1.2, 1.1, 1.5])); nl = int(rng.integers(1, 9))
        ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1))
        img = rand_image(rng, h, w)
        mask = None
        if rng.uniform() < 0.25:
            mask = np.full((h, w), 255, np.uint8); x0 = int(rng.integers(0, w - 20)); mask[:, x0:x0 + int(rng.integers(10, w // 2))] = 0
        try:
            ex = plp.orb_extractor(K, sfac, nl, ini, mn)
            got = ex.extract(img, mask)
        except Exception as e:   # documented kernel limits (quota per level <= 1022, ...): refused loudly, never wrong
            if "limits" not in str(e) and "too small" not in str(e) and "overflow" not in str(e):
                raise
            skipped = locals().get("skipped", 0) + 1
            continue
        want = O.OrbOracle(K, sfac, nl, ini, mn).extract(img, mask)
        if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
            bad += 1; print("ORB MISMATCH", h, w, K, sfac, nl, ini, mn, len(got[0]), len(want[0]))
        n += 1
    print(f"orb: {n} random frames/parameter sets compared (refusals by documented limits are skipped), mismatches so far {bad}")
    # ---- lines
    t0, n = time.time(), 0
    lt = plp.LineFeatureTracker()
    while time.time() - t0 < budget:
        h, w = int(rng.integers(200, 600)), int(rng.integers(240, 900))
        img = rand_image(rng, h, w)
        kl, lbd, fn = lt.extract_LSD_LBD(img)
        o = O.LineOracle(img)
        ok = len(kl) == len(o.keylsd) and np.array_equal(lbd, o.lbd) and np.array_equal(kl, o.keylsd) and np.array_equal(fn, o.linefn)
        if not ok:
            bad += 1; print("LINE MISMATCH", h, w, len(kl), len(o.keylsd))
        n += 1
    print(f"lines: {n} random frames, mismatches so far {bad}")
    # ---- matchers
    t0, n = time.time(), 0
    grid = plp.make_grid(640, 480)
    SF = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    while time.time() - t0 < budget:
        nt, m = int(rng.integers(1, 2500)), int(rng.integers(1, 3000))
        t, q = MC.random_problem(rng, nt, m, n_words=int(rng.choice([0, 0, 3, 20])), stereo=bool(rng.integers(0, 2)))
        margin, ratio = float(rng.uniform(2, 40)), float(rng.choice([0.6, 0.75, 0.9]))
        want, wn = O.match_frame_and_landmarks(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"], q["q_reproj"],
                                               q["q_x_right"], q["q_level"], q["q_desc"], q["q_has_obs"], margin, ratio)
        got, gn = plp.matcher(ratio, False).match_host(plp.MODE_LANDMARKS, nt, m, {**t, **q}, margin=margin, scale_factors=SF, grid=grid)
        if gn[0] != wn or not np.array_equal(got[0], want):
            bad += 1; print("LANDMARKS MISMATCH", nt, m, margin, ratio)
        d = int(rng.integers(0, 3)); chk = bool(rng.integers(0, 2))
        want, wn = O.match_current_and_last(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"], q["q_reproj"],
                                            q["q_x_right"], q["q_level"], q["q_angle"], q["q_desc"], q["q_has_obs"], margin, d, chk)
        got, gn = plp.matcher(0.9, chk).match_host(plp.MODE_LAST_FRAME, nt, m, {**t, **q}, margin=margin, direction=d, scale_factors=SF, grid=grid)
        if gn[0] != wn or not np.array_equal(got[0], want):
            bad += 1; print("LAST_FRAME MISMATCH", nt, m, margin, d, chk)
        n += 1
    print(f"matchers: {n} random problems x 2 modes, mismatches so far {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
