#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: HIP path vs oracle on random frames / parameters / matcher problems.
   python tools/fuzz_gpu.py --seconds 240 [--seed N] [--aux-seconds 120]
   -> prints one summary line per family, exit code 1 on any mismatch.  --aux-seconds adds the steps either side of the
   path: bag-of-words transform on random vocabularies, stereo rectification maps + remap on random calibrations."""
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


def strided_view(rng, img):
    """the image inside a larger array of noise, as a view with a row step (cv::Mat ROI)"""
    h, w = img.shape
    top, left, right = int(rng.integers(0, 6)), int(rng.integers(0, 70)), int(rng.integers(1, 300))
    big = rng.integers(0, 256, (h + top + 2, w + left + right), dtype=np.uint8)
    big[top:top + h, left:left + w] = img
    return big[top:top + h, left:left + w]


def aux_families(rng, seconds):
    import ctypes as C
    import torch
    bad = 0
    dev = torch.device("cuda", 0)
    # ---- bag of words
    t0, n = time.time(), 0
    while time.time() - t0 < seconds / 2:
        k, L = int(rng.integers(2, 24)), int(rng.integers(1, 7))
        if k ** L > 200000:
            L = max(1, int(np.log(200000) / np.log(k)))
        weighting, scoring = int(rng.integers(0, 4)), int(rng.integers(0, 6))
        parents, is_leaf, descs, weights = O.random_vocab(rng, k, L, p_leaf=float(rng.uniform(0, 0.4)), p_dup=float(rng.uniform(0, 0.3)),
                                                          p_stop=float(rng.uniform(0, 0.2)), k_jitter=int(rng.integers(0, 3)) if k > 3 else 0)
        if len(parents) < 2:
            continue
        v = plp.bow_vocabulary(L, parents, is_leaf, descs, weights, weighting, scoring)
        B, cap = int(rng.integers(1, 6)), int(rng.integers(1, 2500))
        levelsup = int(rng.integers(0, L + 2))
        desc = rng.integers(0, 256, (B, cap, 32), dtype=np.uint8)
        leaves = np.flatnonzero(is_leaf)
        rep = rng.uniform(size=(B, cap)) < 0.3
        desc[rep] = descs[rng.choice(leaves, int(rep.sum()))]
        counts = rng.integers(0, cap + 1, B).astype(np.int32)
        out = v.transform_device(torch.from_numpy(desc).to(dev), torch.from_numpy(counts).to(dev), levelsup)
        torch.cuda.synchronize()
        got = {kk: vv.cpu().numpy() for kk, vv in out.items()}
        for b in range(B):
            c = int(counts[b])
            wid, nid, bw, bv, fn, ff = O.bow_transform(v.child_offset, v.children, v.node_desc, v.node_weight, v.node_word, v.L, desc[b][:c], levelsup,
                                                       v.accumulate, v.norm)
            ok = (np.array_equal(got["word_id"][b][:c].view(np.uint32), wid) and np.array_equal(got["node_id"][b][:c].view(np.uint32), nid)
                  and got["n_bow"][b] == len(bw) and got["n_fv"][b] == len(fn) and np.array_equal(got["bow_word"][b][:len(bw)].view(np.uint32), bw)
                  and np.array_equal(got["bow_value"][b][:len(bw)], bv) and np.array_equal(got["fv_node"][b][:len(fn)].view(np.uint32), fn)
                  and np.array_equal(got["fv_feat"][b][:len(ff)].view(np.uint32), ff))
            if not ok:
                bad += 1; print("BOW MISMATCH", k, L, weighting, scoring, levelsup, cap, c)
            n += 1
    print(f"bow: {n} random frames on random vocabularies, mismatches so far {bad}")
    # ---- rectification maps and remap
    t0, n = time.time(), 0
    mt = plp.matcher()
    while time.time() - t0 < seconds / 2:
        rows, cols = int(rng.integers(8, 520)), int(rng.integers(8, 800))
        f = float(rng.uniform(150, 900))
        K = np.array([f, 0, cols / 2 + rng.uniform(-20, 20), 0, f * rng.uniform(0.97, 1.03), rows / 2 + rng.uniform(-20, 20), 0, 0, 1])
        nd = int(rng.choice([0, 4, 5, 8, 12]))
        D = (rng.normal(size=12) * np.array([0.2, 0.1, 2e-3, 2e-3, 0.05, 0.02, 0.01, 0.01, 1e-3, 1e-3, 1e-3, 1e-3]))[:nd]
        ax = rng.normal(size=3) * 0.03
        th = np.linalg.norm(ax); kx = ax / th
        Kx = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
        R = (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx).ravel()
        cam = dict(fx=f * rng.uniform(0.9, 1.1), fy=f * rng.uniform(0.9, 1.1), cx=cols / 2 + rng.uniform(-5, 5), cy=rows / 2 + rng.uniform(-5, 5))
        c = plp.camera_c()
        for kk, vv in cam.items():
            setattr(c, kk, float(vv))
        d_mx = torch.zeros((rows, cols), dtype=torch.float32, device=dev); d_my = torch.zeros_like(d_mx)
        plp._check(plp.lib().plp_rectify_map_device(mt._h, plp._p(K), plp._p(D) if nd else None, nd, plp._p(R), C.byref(c), rows, cols, d_mx.data_ptr(),
                                                    d_my.data_ptr(), cols * 4, None))
        src = rand_image(rng, rows, cols) if rows >= 120 and cols >= 160 else rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        d_src = torch.from_numpy(np.ascontiguousarray(src)).to(dev); d_out = torch.zeros((rows, cols), dtype=torch.uint8, device=dev)
        plp._check(plp.lib().plp_remap_linear_device(mt._h, d_src.data_ptr(), rows, cols, cols, rows * cols, d_mx.data_ptr(), d_my.data_ptr(), cols * 4,
                                                     rows, cols, 1, d_out.data_ptr(), cols, rows * cols, None))
        torch.cuda.synchronize()
        mx, my = O.rectify_map(K, D, R, cam, rows, cols)
        ok = np.array_equal(d_mx.cpu().numpy(), mx) and np.array_equal(d_my.cpu().numpy(), my) and np.array_equal(d_out.cpu().numpy(), O.remap_linear(src, mx, my))
        if not ok:
            bad += 1; print("RECTIFY MISMATCH", rows, cols, nd)
        n += 1
    print(f"rectify: {n} random calibrations / frames, mismatches so far {bad}")
    return bad


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120); ap.add_argument("--seed", type=int, default=0); ap.add_argument("--aux-seconds", type=float, default=0.0)
    ap.add_argument("--only", choices=["orb", "lines", "match"], default=None, help="spend all of --seconds on one family")
    ap.add_argument("--soak-calls", type=int, default=0, help="robustness soak instead of a parity sweep: N plp_orb_extract calls, each with a fresh context, a freshly "
                    "allocated random image of a random size and (half of the calls) a fresh mask, no oracle in between -- the pattern that hit the rare GPU "
                    "page fault of round 2 (DESIGN.md section 5); every 16th call is also compared with the oracle")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    if a.soak_calls:
        t0, checked, bad = time.time(), 0, 0
        for n in range(a.soak_calls):
            h, w = int(rng.integers(120, 720)), int(rng.integers(160, 1300))
            K = int(rng.choice([100, 500, 1000, 2000])); sfac = float(rng.choice([1.2, 1.1, 1.1, 1.5])); nl = int(rng.integers(1, 9))
            base = rng.integers(0, 256, (h // 4 + 1, w // 4 + 1), dtype=np.uint8)
            img = np.repeat(np.repeat(base, 4, 0), 4, 1)[:h, :w].copy()          # fresh allocation per call
            mask = None
            if n & 1:
                mask = np.full((h, w), 255, np.uint8); x0 = int(rng.integers(0, w - 20)); mask[:, x0:x0 + int(rng.integers(10, w // 2))] = 0
            try:
                ex = plp.orb_extractor(K, sfac, nl, 20, 7)
                got = ex.extract(img, mask)
            except Exception as e:
                if "limits" not in str(e) and "too small" not in str(e) and "overflow" not in str(e):
                    raise
                continue
            if n % 16 == 0:
                want = O.OrbOracle(K, sfac, nl, 20, 7).extract(img, mask)
                checked += 1
                if not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
                    bad += 1; print("ORB MISMATCH (soak)", h, w, K, sfac, nl)
            del ex, img, mask
            if (n + 1) % 5000 == 0:
                print(f"soak: {n + 1} calls, {time.time() - t0:.0f} s", flush=True)
        print(f"soak: {a.soak_calls} plp_orb_extract calls with fresh contexts / images / masks in {time.time() - t0:.0f} s, no fault; {checked} compared with the oracle, {bad} mismatches")
        sys.exit(1 if bad else 0)
    budget = a.seconds / 3
    share = lambda fam: (a.seconds if a.only == fam else 0.0) if a.only else budget
    bad = 0
    # ---- ORB
    t0, n = time.time(), 0
    while time.time() - t0 < share("orb"):
        h, w = int(rng.integers(120, 720)), int(rng.integers(160, 1300))
        K = int(rng.choice([100, 500, 1000, 2000, 4000])); sfac = float(rng.choice([1.2, 1.2, 1.1, 1.5])); nl = int(rng.integers(1, 9))
        ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1))
        img = rand_image(rng, h, w)
        mask = None
        if os.environ.get("PLP_FUZZ_VERBOSE"): print("orb case", n, h, w, K, sfac, nl, ini, mn, flush=True)
        if rng.uniform() < 0.25:
            mask = np.full((h, w), 255, np.uint8); x0 = int(rng.integers(0, w - 20)); mask[:, x0:x0 + int(rng.integers(10, w // 2))] = 0
            if os.environ.get("PLP_FUZZ_NOMASK"): mask = None   # bisection aid: same random sequence, no image masks
        img_in, mask_in = (strided_view(rng, img), None if mask is None else strided_view(rng, mask)) if n % 5 == 3 else (img, mask)   # one case in five: views with a row step
        try:
            ex = plp.orb_extractor(K, sfac, nl, ini, mn)
            got = ex.extract(img_in, mask_in)
        except Exception as e:   # documented kernel limits (quota per level <= 1960, ...): refused loudly, never wrong
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
    while time.time() - t0 < share("lines"):
        h, w = int(rng.integers(200, 600)), int(rng.integers(240, 900))
        img = rand_image(rng, h, w)
        lt.set_grow_waves((0, 1, 3, 5)[n % 4])          # several waves per frame (automatic / 3 / 5) and one wave per frame: same results
        stable = n % 5 == 4                            # the reference's seed order (std::sort, the default) four times in five, the stable order once
        lt.set_seed_order(plp.SEED_ORDER_STABLE if stable else plp.SEED_ORDER_LIBSTDCXX)
        kl, lbd, fn = lt.extract_LSD_LBD(strided_view(rng, img) if n % 5 == 2 else img)
        o = O.LineOracle(img, stable_order=stable)
        ok = len(kl) == len(o.keylsd) and np.array_equal(lbd, o.lbd) and np.array_equal(kl, o.keylsd) and np.array_equal(fn, o.linefn)
        if not ok:
            bad += 1; print("LINE MISMATCH", h, w, len(kl), len(o.keylsd))
        n += 1
    print(f"lines: {n} random frames, mismatches so far {bad}")
    # ---- matchers
    t0, n = time.time(), 0
    grid = plp.make_grid(640, 480)
    SF = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    while time.time() - t0 < share("match"):
        nt, m = int(rng.integers(1, 2500)), int(rng.integers(1, 3000))
        t, q = MC.random_problem(rng, nt, m, n_words=int(rng.choice([0, 0, 3, 20])), stereo=bool(rng.integers(0, 2)))
        if n % 25 == 24:   # an empty side (no key points / no landmarks): defined result, no kernel
            if rng.integers(0, 2): nt = 0; t = {k: v[:0] for k, v in t.items()}
            else: m = 0; q = {k: v[:0] for k, v in q.items()}
        margin, ratio = float(rng.uniform(2, 40)), float(rng.choice([0.6, 0.75, 0.9]))
        # the LDS-size hint never changes a result: none / far too small / a little too small / generous, at random
        t = {**t, "t_count_hint": int(rng.choice([0, max(1, nt // 4), max(1, nt - 1), nt + 100]))}
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
    if a.aux_seconds > 0:
        bad += aux_families(rng, a.aux_seconds)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
