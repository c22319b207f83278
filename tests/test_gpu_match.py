"""GPU parity of the array-form Hamming matchers (through the C ABI) against the oracle restatement of
match::projection / match::robust.  Integer work: the association arrays and num_matches must be identical."""
import numpy as np
import pytest

import oracle_lib as O
from plp import plp, synth
import match_cases as MC

pytestmark = pytest.mark.gpu
SF = (np.float32(1.2) ** np.arange(8)).astype(np.float32)


def run_landmarks(t, q, margin, ratio, grid):
    want, wn = O.match_frame_and_landmarks(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"],
                                           q["q_reproj"], q["q_x_right"], q["q_level"], q["q_desc"], q["q_has_obs"], margin, ratio)
    mt = plp.matcher(ratio, True)
    got, gn = mt.match_host(plp.MODE_LANDMARKS, len(t["t_kps"]), len(q["q_level"]), {**t, **q}, margin=margin, scale_factors=SF, grid=grid)
    assert gn[0] == wn
    assert np.array_equal(got[0], want)
    same_with_a_count_hint(mt, plp.MODE_LANDMARKS, t, q, want, wn, margin=margin, scale_factors=SF, grid=grid)
    return wn


def same_with_a_count_hint(mt, mode, t, q, want, wn, **kw):
    """plp_match_args.t_count_hint sizes the LDS copy of a frame's targets; a frame with MORE targets than the hint reads the rest from memory:
    the hint (too small by a lot, too small by one, larger than needed) never changes a result"""
    n = len(t["t_kps"])
    for hint in sorted({max(1, n // 3), max(1, n - 1), n + 40}):
        got, gn = mt.match_host(mode, n, len(q["q_level"]), {**t, **q, "t_count_hint": hint}, **kw)
        assert gn[0] == wn and np.array_equal(got[0], want), hint


def run_last(t, q, margin, direction, check, grid):
    want, wn = O.match_current_and_last(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"],
                                        q["q_reproj"], q["q_x_right"], q["q_level"], q["q_angle"], q["q_desc"], q["q_has_obs"], margin,
                                        direction, check)
    mt = plp.matcher(0.9, check)
    got, gn = mt.match_host(plp.MODE_LAST_FRAME, len(t["t_kps"]), len(q["q_level"]), {**t, **q}, margin=margin, direction=direction,
                            scale_factors=SF, grid=grid)
    assert gn[0] == wn
    assert np.array_equal(got[0], want)
    same_with_a_count_hint(mt, plp.MODE_LAST_FRAME, t, q, want, wn, margin=margin, direction=direction, scale_factors=SF, grid=grid)
    if check:
        # PLP_MATCH_FLAG_MARK_INVALIDATED: -2 exactly where the orientation check removed a match (projection.cc:350-354)
        raw, _ = O.match_current_and_last(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"],
                                          q["q_reproj"], q["q_x_right"], q["q_level"], q["q_angle"], q["q_desc"], q["q_has_obs"], margin,
                                          direction, False)
        marked, mn = mt.match_host(plp.MODE_LAST_FRAME, len(t["t_kps"]), len(q["q_level"]), {**t, **q, "flags": plp.FLAG_MARK_INVALIDATED},
                                   margin=margin, direction=direction, scale_factors=SF, grid=grid)
        assert mn[0] == wn and np.array_equal(marked[0], np.where((raw >= 0) & (want < 0), -2, want))
    return wn


def run_brute(desc1, angle1, desc2, angle2, valid2, ratio, check):
    want, wn = O.brute_force_match(desc1, angle1, desc2, angle2, valid2, ratio, check)
    mt = plp.matcher(ratio, check)
    got, gn = mt.match_host(plp.MODE_BRUTE_FORCE, len(desc1), len(desc2),
                            dict(t_desc=desc1, t_angle=angle1, q_desc=desc2, q_angle=angle2, q_valid=valid2))
    assert gn[0] == wn
    assert np.array_equal(got[0], want)
    return wn


def test_hamming_matrix():
    rng = np.random.default_rng(0)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8); t = rng.integers(0, 256, (517, 32), dtype=np.uint8)
    d = plp.matcher().hamming_matrix(q, t)
    want = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    assert np.array_equal(d, want)
    for a, b, w in [(0b01010101, 0b01010101, 0), (0b01010101, 0b10101010, 256), (0b01100110, 0b00111100, 128)]:   # base.cc vectors
        assert plp.matcher().hamming_matrix(np.full((1, 32), a, np.uint8), np.full((1, 32), b, np.uint8))[0, 0] == w


def test_tracking_shaped_problem_from_real_features():
    frames = synth.replay(21, 3)
    rng = np.random.default_rng(1)
    grid = plp.make_grid(640, 480)
    feats = [MC.features_from_oracle(f, 1000) for f in frames]
    total = 0
    for a, b in [(0, 1), (1, 2)]:
        pk, pd = feats[a]; ck, cd = feats[b]
        q = MC.make_queries_from_prev(pk, pd, rng, shift=(-3.0, 0.0))
        t = dict(t_kps=ck, t_desc=cd, t_x_right=np.full(len(ck), -1, np.float32), t_occupied=np.zeros(len(ck), np.uint8))
        total += run_landmarks(t, q, 10.0, 0.8, grid)
        total += run_last(t, q, 20.0, 0, True, grid)
        total += run_last(t, q, 40.0, 1, False, grid)
        total += run_brute(cd, ck["angle"], pd, pk["angle"], q["q_valid"], 0.75, True)
    assert total > 500    # the problems are not vacuous


@pytest.mark.parametrize("seed", range(4))
def test_random_problems_with_ties_conflicts_and_stereo(seed):
    rng = np.random.default_rng(100 + seed)
    grid = plp.make_grid(640, 480)
    for trial in range(6):
        n, m = int(rng.integers(1, 1500)), int(rng.integers(1, 2500))
        t, q = MC.random_problem(rng, n, m, n_words=(0, 3, 12)[trial % 3], stereo=bool(trial % 2))
        run_landmarks(t, q, float(rng.uniform(5, 60)), float(rng.choice([0.6, 0.8, 0.9])), grid)
        run_last(t, q, float(rng.uniform(5, 60)), trial % 3, bool(trial & 1), grid)


@pytest.mark.parametrize("seed", range(3))
def test_brute_force_heavy_conflicts(seed):
    rng = np.random.default_rng(7 + seed)
    for n1, n2, words in [(300, 400, 5), (1200, 900, 40), (2000, 2000, 0), (1, 5, 2), (64, 1, 1)]:
        t, q = MC.random_problem(rng, n1, n2, n_words=words)
        run_brute(t["t_desc"], t["t_kps"]["angle"], q["q_desc"], q["q_angle"], q["q_valid"], 0.75, bool(seed & 1))


def test_batched_device_matches_per_problem_host():
    import torch
    rng = np.random.default_rng(5)
    grid = plp.make_grid(640, 480)
    B, n_cap, m_cap = 5, 700, 900
    ts, qs = zip(*[MC.random_problem(rng, int(rng.integers(200, n_cap)), int(rng.integers(200, m_cap)), n_words=(0, 8)[b % 2]) for b in range(B)])
    dev = torch.device("cuda:0")

    def pad(arrs, cap, dt):
        out = np.zeros((B, cap) + arrs[0].shape[1:], dt)
        for b, a in enumerate(arrs):
            out[b, :len(a)] = a
        return out
    f = {k: pad([t[k] for t in ts], n_cap, ts[0][k].dtype) for k in ts[0]}
    f.update({k: pad([q[k] for q in qs], m_cap, qs[0][k].dtype) for k in qs[0]})
    f["t_counts"] = np.array([len(t["t_kps"]) for t in ts], np.int32)
    f["q_counts"] = np.array([len(q["q_level"]) for q in qs], np.int32)
    d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype == plp.KP_DTYPE else v).to(dev) for k, v in f.items()}
    out_match = torch.full((B, n_cap), -7, dtype=torch.int32, device=dev)
    out_num = torch.zeros(B, dtype=torch.int32, device=dev)
    mt = plp.matcher(0.8, True)
    mt.match_device(plp.MODE_LANDMARKS, n_cap, m_cap, d, out_match, out_num, margin=15.0, scale_factors=SF, grid=grid, B=B)
    torch.cuda.synchronize()
    om, on = out_match.cpu().numpy(), out_num.cpu().numpy()
    for b in range(B):
        t, q = ts[b], qs[b]
        want, wn = O.match_frame_and_landmarks(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"],
                                               q["q_reproj"], q["q_x_right"], q["q_level"], q["q_desc"], q["q_has_obs"], 15.0, 0.8)
        assert on[b] == wn and np.array_equal(om[b, :len(want)], want)


# ---------------------------------------------------------------------------------------- line variants
def random_line_problem(rng, n, m, words=0):
    kl = np.zeros(n, O.KL_DTYPE)
    x1 = rng.uniform(0, 640, n); y1 = rng.uniform(0, 480, n); ang = rng.uniform(0, np.pi, n); ln = rng.uniform(60, 250, n)
    kl["startPointX"], kl["startPointY"] = x1, y1
    kl["endPointX"], kl["endPointY"] = x1 + ln * np.cos(ang), y1 + ln * np.sin(ang)
    kl["octave"] = rng.integers(0, 2, n)
    vocab = rng.integers(0, 256, (max(words, 1), 32), dtype=np.uint8)
    lbd = vocab[rng.integers(0, len(vocab), n)].copy() if words else rng.integers(0, 256, (n, 32), dtype=np.uint8)
    src = rng.integers(0, n, m)
    sp = np.stack([kl["startPointX"][src], kl["startPointY"][src]], 1) + rng.normal(0, 3, (m, 2))
    ep = np.stack([kl["endPointX"][src], kl["endPointY"][src]], 1) + rng.normal(0, 3, (m, 2))
    qd = lbd[src].copy()
    flip = rng.integers(0, 32, m); qd[np.arange(m), flip] ^= np.uint8(1) << rng.integers(0, 8, m).astype(np.uint8)
    t = dict(t_kl=kl, t_desc=lbd, t_kp_octave=rng.integers(0, 8, n).astype(np.int32), t_occupied=(rng.uniform(size=n) < 0.1).astype(np.uint8),
             t_x_right=(rng.uniform(5, 600, n) * (rng.uniform(size=n) < 0.7) - 1).astype(np.float32),
             t_x_right2=(rng.uniform(5, 600, n) * (rng.uniform(size=n) < 0.7) - 1).astype(np.float32))
    q = dict(q_valid=(rng.uniform(size=m) > 0.1).astype(np.uint8), q_reproj=sp.astype(np.float32), q_reproj2=ep.astype(np.float32),
             q_level=rng.integers(0, 2, m).astype(np.int32), q_desc=qd, q_has_obs=(rng.uniform(size=m) > 0.05).astype(np.uint8),
             q_x_right=(sp[:, 0] - rng.uniform(0, 40, m)).astype(np.float32), q_x_right2=(ep[:, 0] - rng.uniform(0, 40, m)).astype(np.float32))
    return t, q


@pytest.mark.parametrize("seed", range(3))
def test_line_projection_matchers(seed):
    rng = np.random.default_rng(40 + seed)
    sf_lsd = np.array([1.0, 2.0], np.float32)
    for n, m, words in [(60, 80, 0), (300, 500, 4), (1, 3, 1), (150, 40, 2)]:
        t, q = random_line_problem(rng, n, m, words)
        margin, ratio = float(rng.uniform(3, 30)), 0.8
        want, wn = O.match_frame_and_landmarks_line(t["t_kl"], t["t_desc"], t["t_kp_octave"], t["t_occupied"], sf_lsd, q["q_valid"], q["q_reproj"],
                                                    q["q_reproj2"], q["q_level"], q["q_desc"], q["q_has_obs"], margin, ratio)
        got, gn = plp.matcher(ratio, False).match_host(plp.MODE_LANDMARKS_LINE, n, m, {**t, **q}, margin=margin, scale_factors=sf_lsd)
        assert gn[0] == wn and np.array_equal(got[0], want)
        for direction in (0, 1, 2):
            for rgbd in (0, 1):
                xr_pair = np.stack([t["t_x_right"], t["t_x_right2"]], 1)
                want, wn = O.match_current_and_last_line(t["t_kl"], t["t_desc"], xr_pair, t["t_occupied"], sf_lsd, 1, q["q_valid"], q["q_reproj"],
                                                         q["q_reproj2"], q["q_x_right"], q["q_x_right2"], q["q_level"], q["q_desc"], q["q_has_obs"],
                                                         margin, direction, rgbd)
                got, gn = plp.matcher(0.9, True).match_host(plp.MODE_LAST_FRAME_LINE, n, m, {**t, **q, "is_rgbd": rgbd, "num_levels_lsd": 1},
                                                            margin=margin, direction=direction, scale_factors=sf_lsd)
                assert gn[0] == wn and np.array_equal(got[0], want), (n, m, direction, rgbd)


def test_batched_line_matchers_small_target_sets_one_lane_per_query():
    """64 frames x (<= 120 key lines against <= 200 projected lines): the batch path with one lane per query (k_match_topk_lanes: target
    capacity <= 512 and >= 64 problems) must equal the per-problem oracle, for both line projection matchers"""
    import torch
    rng = np.random.default_rng(77)
    sf_lsd = np.array([1.0, 2.0], np.float32)
    B, n_cap, m_cap = 64, 128, 256
    ts, qs = zip(*[random_line_problem(rng, int(rng.integers(1, 120)), int(rng.integers(1, 200)), words=(0, 3)[b % 2]) for b in range(B)])
    dev = torch.device("cuda:0")

    def pad(arrs, cap, dt):
        out = np.zeros((B, cap) + arrs[0].shape[1:], dt)
        for b, a in enumerate(arrs):
            out[b, :len(a)] = a
        return out
    f = {k: pad([t[k] for t in ts], n_cap, ts[0][k].dtype) for k in ts[0]}
    f.update({k: pad([q[k] for q in qs], m_cap, qs[0][k].dtype) for k in qs[0]})
    f["t_counts"] = np.array([len(t["t_kl"]) for t in ts], np.int32)
    f["q_counts"] = np.array([len(q["q_level"]) for q in qs], np.int32)
    d = {k: torch.from_numpy(v.view(np.uint8) if v.dtype in (plp.KP_DTYPE, plp.KL_DTYPE) else v).to(dev) for k, v in f.items()}
    out_match = torch.full((B, n_cap), -7, dtype=torch.int32, device=dev)
    out_num = torch.zeros(B, dtype=torch.int32, device=dev)
    plp.matcher(0.8, False).match_device(plp.MODE_LANDMARKS_LINE, n_cap, m_cap, d, out_match, out_num, margin=12.0, scale_factors=sf_lsd, B=B)
    torch.cuda.synchronize()
    om, on = out_match.cpu().numpy(), out_num.cpu().numpy()
    for b in range(B):
        t, q = ts[b], qs[b]
        want, wn = O.match_frame_and_landmarks_line(t["t_kl"], t["t_desc"], t["t_kp_octave"], t["t_occupied"], sf_lsd, q["q_valid"], q["q_reproj"],
                                                    q["q_reproj2"], q["q_level"], q["q_desc"], q["q_has_obs"], 12.0, 0.8)
        assert on[b] == wn and np.array_equal(om[b, :len(want)], want), b
    plp.matcher(0.9, True).match_device(plp.MODE_LAST_FRAME_LINE, n_cap, m_cap, {**d, "is_rgbd": 1, "num_levels_lsd": 1}, out_match, out_num, margin=12.0,
                                        direction=0, scale_factors=sf_lsd, B=B)
    torch.cuda.synchronize()
    om, on = out_match.cpu().numpy(), out_num.cpu().numpy()
    for b in range(B):
        t, q = ts[b], qs[b]
        want, wn = O.match_current_and_last_line(t["t_kl"], t["t_desc"], np.stack([t["t_x_right"], t["t_x_right2"]], 1), t["t_occupied"], sf_lsd, 1,
                                                 q["q_valid"], q["q_reproj"], q["q_reproj2"], q["q_x_right"], q["q_x_right2"], q["q_level"], q["q_desc"],
                                                 q["q_has_obs"], 12.0, 0, 1)
        assert on[b] == wn and np.array_equal(om[b, :len(want)], want), b


# ---------------------------------------------------------------------------------------- BoW-guided, fuse, area
@pytest.mark.parametrize("seed", range(3))
def test_bow_guided_matcher(seed):
    rng = np.random.default_rng(80 + seed)
    for n, m, nodes, words in [(800, 700, 40, 0), (1500, 1500, 12, 30), (50, 60, 3, 4), (1, 2, 1, 1)]:
        t, q = MC.random_problem(rng, n, m, n_words=words)
        # node id = a hash of the descriptor's first byte (as a vocabulary tree would group similar descriptors)
        t_node = (t["t_desc"][:, 0].astype(np.int32) * 7 + 3) % nodes
        q_node = (q["q_desc"][:, 0].astype(np.int32) * 7 + 3) % nodes
        order = np.argsort(q_node, kind="stable")      # the reference walks the key-frame features in node order
        qd, qa, qn, qv = q["q_desc"][order], q["q_angle"][order], q_node[order], q["q_valid"][order]
        for skip in (None, t["t_occupied"]):
            for check in (True, False):
                tskip = np.zeros(n, np.uint8) if skip is None else skip
                want, wn = O.match_bow(qd, qa, qn, qv, t["t_desc"], t["t_kps"]["angle"], t_node, tskip, 0.75, check)
                got, gn = plp.matcher(0.75, check).match_host(plp.MODE_BOW, n, m, dict(t_desc=t["t_desc"], t_angle=t["t_kps"]["angle"], t_group=t_node,
                                                              t_occupied=skip, q_desc=qd, q_angle=qa, q_group=qn, q_valid=qv))
                assert gn[0] == wn and np.array_equal(got[0], want), (n, m, nodes, check)


@pytest.mark.parametrize("seed", range(3))
def test_fuse_search(seed):
    rng = np.random.default_rng(90 + seed)
    grid = plp.make_grid(640, 480)
    inv_sigma = (1.0 / (SF * SF)).astype(np.float32)
    for n, m in [(900, 1200), (1500, 300), (3, 10)]:
        t, q = MC.random_problem(rng, n, m, n_words=(0, 6)[seed % 2], stereo=True)
        reproj_d = q["q_reproj"].astype(np.float64) + rng.normal(0, 0.7, (m, 2))
        pred = rng.integers(0, 8, m).astype(np.uint32)            # includes 0: the unsigned-wrap quirk rejects everything
        want = O.fuse_search(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], SF, inv_sigma, q["q_valid"], reproj_d, q["q_x_right"], pred,
                             q["q_desc"], 3.0)
        got = plp.matcher().match_host(plp.MODE_FUSE, n, m, dict(t_kps=t["t_kps"], t_desc=t["t_desc"], t_x_right=t["t_x_right"], q_valid=q["q_valid"],
                                       q_reproj_d=reproj_d, q_x_right=q["q_x_right"], q_level=pred.astype(np.int32), q_desc=q["q_desc"],
                                       inv_level_sigma_sq=inv_sigma), margin=3.0, scale_factors=SF, grid=grid)
        assert np.array_equal(got[0], want)
        assert (want[pred == 0] == -1).all()
    assert (want >= 0).sum() >= 0


@pytest.mark.parametrize("seed", range(3))
def test_area_matcher(seed):
    rng = np.random.default_rng(70 + seed)
    grid = plp.make_grid(640, 480)
    frames = synth.replay(50 + seed, 2, step_px=5)
    f1, f2 = MC.features_from_oracle(frames[0], 2000), MC.features_from_oracle(frames[1], 2000)
    prev = np.stack([f1[0]["x"], f1[0]["y"]], 1).astype(np.float32)          # initializer: prev_matched_pts = key points of frame 1
    for check in (True, False):
        want, wpp, wn = O.match_area(O.grid6(grid), f1[0], f1[1], f2[0], f2[1], prev, 100, 0.9, check)
        got, gpp, gn = plp.matcher(0.9, check).match_in_consistent_area(f1[0], f1[1], f2[0], f2[1], prev, 100, grid)
        assert gn == wn and np.array_equal(got, want) and np.array_equal(gpp, wpp)
        assert wn > 50
    # synthetic: few distinct descriptors -> many "steal if closer" events
    t1, _ = MC.random_problem(rng, 600, 5, n_words=5); t2, _ = MC.random_problem(rng, 700, 5, n_words=5)
    t1["t_kps"]["octave"] = rng.integers(0, 2, 600); t2["t_kps"]["octave"] = rng.integers(0, 2, 700)
    prev = np.stack([t1["t_kps"]["x"], t1["t_kps"]["y"]], 1).astype(np.float32)
    want, wpp, wn = O.match_area(O.grid6(grid), t1["t_kps"], t1["t_desc"], t2["t_kps"], t2["t_desc"], prev, 150, 0.9, True)
    got, gpp, gn = plp.matcher(0.9, True).match_in_consistent_area(t1["t_kps"], t1["t_desc"], t2["t_kps"], t2["t_desc"], prev, 150, grid)
    assert gn == wn and np.array_equal(got, want) and np.array_equal(gpp, wpp)


# ---------------------------------------------------------------------------------------- relocalisation / loop closing / mapping variants
@pytest.mark.parametrize("seed", range(3))
def test_frame_and_keyframe_and_sim3(seed):
    rng = np.random.default_rng(110 + seed)
    grid = plp.make_grid(640, 480)
    for n, m, words in [(1000, 900, 0), (1500, 1500, 8), (40, 70, 2)]:
        t, q = MC.random_problem(rng, n, m, n_words=words)
        pred = q["q_level"].astype(np.uint32)                                     # contains 0: (int)(pred - 1) = -1 / the unsigned wrap
        for thr, check in ((50, True), (100, False)):
            want, wn = O.match_frame_and_keyframe(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_occupied"], SF, q["q_valid"], q["q_reproj"], pred,
                                                  q["q_angle"], q["q_desc"], 10.0, thr, check)
            got, gn = plp.matcher(0.9, check).match_host(plp.MODE_LAST_FRAME, n, m, dict(t_kps=t["t_kps"], t_desc=t["t_desc"], t_occupied=t["t_occupied"],
                                  q_valid=q["q_valid"], q_reproj=q["q_reproj"], q_level=q["q_level"], q_angle=q["q_angle"], q_desc=q["q_desc"],
                                  hamm_dist_thr=thr), margin=10.0, direction=0, scale_factors=SF, grid=grid)
            assert gn[0] == wn and np.array_equal(got[0], want), (n, m, thr)
        want, wn = O.match_by_sim3(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_occupied"], SF, q["q_valid"], q["q_reproj"], pred, q["q_desc"], 7.5)
        got, gn = plp.matcher(0.9, False).match_host(plp.MODE_LAST_FRAME, n, m, dict(t_kps=t["t_kps"], t_desc=t["t_desc"], t_occupied=t["t_occupied"],
                              q_valid=q["q_valid"], q_reproj=q["q_reproj"], q_level=q["q_level"], q_desc=q["q_desc"], hamm_dist_thr=50, level_window=1,
                              flags=plp.FLAG_UNSIGNED_LEVEL), margin=7.5, scale_factors=SF, grid=grid)
        assert gn[0] == wn and np.array_equal(got[0], want), (n, m)
        assert not np.isin(want, np.nonzero(pred == 0)[0]).any()


@pytest.mark.parametrize("seed", range(2))
def test_mutual_and_detect_duplication(seed):
    rng = np.random.default_rng(120 + seed)
    grid = plp.make_grid(640, 480)
    n1, n2 = 1200, 1100
    a, qa = MC.random_problem(rng, n1, n2, n_words=(0, 10)[seed])     # key frame 1 key points, landmarks of key frame 2 projected into it
    b, qb = MC.random_problem(rng, n2, n1, n_words=(0, 10)[seed])
    best = []
    for t, q in ((b, qb), (a, qa)):          # landmarks of 1 searched in 2, then landmarks of 2 searched in 1
        rd = q["q_reproj"].astype(np.float64) + rng.normal(0, 0.3, q["q_reproj"].shape)
        pred = q["q_level"].astype(np.uint32)
        want = O.project_best(O.grid6(grid), t["t_kps"], t["t_desc"], SF, q["q_valid"], rd, pred, q["q_desc"], 7.5, 100, 0)
        got = plp.matcher().match_host(plp.MODE_FUSE, len(t["t_kps"]), len(pred), dict(t_kps=t["t_kps"], t_desc=t["t_desc"], q_valid=q["q_valid"],
                                       q_reproj_d=rd, q_level=q["q_level"], q_desc=q["q_desc"], inv_level_sigma_sq=np.ones(8, np.float32),
                                       flags=plp.FLAG_NO_CHI2, hamm_dist_thr=100), margin=7.5, scale_factors=SF, grid=grid)
        assert np.array_equal(got[0], want)
        best.append(want)
        # fuse::detect_duplication: signed level window, threshold 50
        want = O.project_best(O.grid6(grid), t["t_kps"], t["t_desc"], SF, q["q_valid"], rd, pred, q["q_desc"], 4.0, 50, 1)
        got = plp.matcher().match_host(plp.MODE_FUSE, len(t["t_kps"]), len(pred), dict(t_kps=t["t_kps"], t_desc=t["t_desc"], q_valid=q["q_valid"],
                                       q_reproj_d=rd, q_level=q["q_level"], q_desc=q["q_desc"], inv_level_sigma_sq=np.ones(8, np.float32),
                                       flags=plp.FLAG_NO_CHI2 | plp.FLAG_SIGNED_LEVEL), margin=4.0, scale_factors=SF, grid=grid)
        assert np.array_equal(got[0], want)
        assert (want[pred == 0] >= 0).any()
    m21, num = O.cross_check(best[0], best[1])
    assert num == (m21 >= 0).sum()


@pytest.mark.parametrize("seed", range(3))
def test_line_keyframe_and_fuse(seed):
    rng = np.random.default_rng(130 + seed)
    sf_lsd = np.array([1.0, 2.0], np.float32)
    inv_sigma = np.array([1.0, 0.25], np.float32)
    for n, m, words in [(80, 120, 0), (300, 400, 4), (2, 5, 1)]:
        t, q = random_line_problem(rng, n, m, words)
        pred = q["q_level"].astype(np.uint32)
        want, wn = O.match_frame_and_keyframe_line(t["t_kl"], t["t_desc"], t["t_occupied"], sf_lsd, q["q_valid"], q["q_reproj"], q["q_reproj2"], pred,
                                                   q["q_desc"], 12.0, 60)
        got, gn = plp.matcher(0.9, False).match_host(plp.MODE_LAST_FRAME_LINE, n, m, dict(t_kl=t["t_kl"], t_desc=t["t_desc"], t_occupied=t["t_occupied"],
                              q_valid=q["q_valid"], q_reproj=q["q_reproj"], q_reproj2=q["q_reproj2"], q_level=q["q_level"], q_desc=q["q_desc"],
                              hamm_dist_thr=60, is_rgbd=0, num_levels_lsd=1), margin=12.0, direction=0, scale_factors=sf_lsd)
        assert gn[0] == wn and np.array_equal(got[0], want)
        sp_d = q["q_reproj"].astype(np.float64) + rng.normal(0, 0.2, (m, 2)); ep_d = q["q_reproj2"].astype(np.float64) + rng.normal(0, 0.2, (m, 2))
        want = O.fuse_search_line(t["t_kl"], t["t_desc"], sf_lsd, inv_sigma, q["q_valid"], sp_d, ep_d, pred, q["q_desc"], 6.0)
        got = plp.matcher().match_host(plp.MODE_FUSE_LINE, n, m, dict(t_kl=t["t_kl"], t_desc=t["t_desc"], q_valid=q["q_valid"], q_reproj_d=sp_d,
                                       q_reproj2_d=ep_d, q_level=q["q_level"], q_desc=q["q_desc"], inv_level_sigma_sq=inv_sigma),
                                       margin=6.0, scale_factors=sf_lsd)
        assert np.array_equal(got[0], want)
    assert (want >= 0).any()


@pytest.mark.parametrize("seed", range(3))
def test_match_for_triangulation(seed):
    rng = np.random.default_rng(140 + seed)
    for n, m, nodes, words in [(900, 800, 30, 0), (1200, 1200, 10, 25), (30, 40, 2, 3)]:
        t, q = MC.random_problem(rng, n, m, n_words=words)
        t_node = (t["t_desc"][:, 0].astype(np.int32) * 7 + 3) % nodes
        q_node = (q["q_desc"][:, 0].astype(np.int32) * 7 + 3) % nodes
        order = np.argsort(q_node, kind="stable")
        qd, qa, qn = q["q_desc"][order], q["q_angle"][order], q_node[order]
        q_has_lm = (rng.uniform(size=m) < 0.3).astype(np.uint8); t_has_lm = (rng.uniform(size=n) < 0.3).astype(np.uint8)
        q_xr = np.where(rng.uniform(size=m) < 0.3, 100.0, -1.0).astype(np.float32); t_xr = np.where(rng.uniform(size=n) < 0.3, 100.0, -1.0).astype(np.float32)
        q_oct = rng.integers(0, 8, m).astype(np.int32)
        # geometry: camera 2 = camera 1 translated by tr (R = I): E_12 = [tr]_x; bearings of common 3-D points + angular noise around the 0.2 deg gate
        tr = np.array([0.3, 0.02, 0.05])
        E = np.array([[0, -tr[2], tr[1]], [tr[2], 0, -tr[0]], [-tr[1], tr[0], 0]], np.float64)
        epipole = -tr / np.linalg.norm(tr)                      # camera centre 1 seen from camera 2 (p2 = p1 - tr)
        def bearings(k):
            p1 = np.stack([rng.uniform(-2, 2, k), rng.uniform(-1.5, 1.5, k), rng.uniform(1, 8, k)], 1)
            return p1
        pts_t = bearings(n)
        src = rng.integers(0, n, m)
        pts_q = pts_t[src] + rng.normal(0, 0.004, (m, 3)) * pts_t[src][:, 2:3]
        b1 = pts_q / np.linalg.norm(pts_q, axis=1, keepdims=True)                     # key frame 1 bearings (queries)
        p2 = pts_t - tr
        p2[rng.uniform(size=n) < 0.05] = epipole * 3.0 + rng.normal(0, 0.01, 3)      # a few targets next to the epipole
        b2 = p2 / np.linalg.norm(p2, axis=1, keepdims=True)
        for check in (True, False):
            want, wn = O.match_for_triangulation(qd, qa, qn, q_has_lm, q_xr, q_oct, b1, t["t_desc"], t["t_kps"]["angle"], t_node, t_has_lm, t_xr, b2,
                                                 SF, E.ravel(), epipole, check)
            got, gn = plp.matcher(0.9, check).match_host(plp.MODE_TRIANGULATION, n, m, dict(t_desc=t["t_desc"], t_angle=t["t_kps"]["angle"],
                                  t_group=t_node, t_occupied=t_has_lm, t_x_right=t_xr, t_bearing=b2, q_desc=qd, q_angle=qa, q_group=qn,
                                  q_valid=(1 - q_has_lm).astype(np.uint8), q_x_right=q_xr, q_level=q_oct, q_bearing=b1,
                                  epipolar=np.concatenate([E.ravel(), epipole])), scale_factors=SF)
            want_t = np.full(n, -1, np.int32); sel = want >= 0; want_t[want[sel]] = np.nonzero(sel)[0]
            assert gn[0] == wn and np.array_equal(got[0], want_t), (n, m, check)
    assert wn > 0


def test_large_frames_take_the_global_memory_path():
    """more targets than the LDS staging holds (KITTI, K = 4000: up to 8064 slots): k_match_topk + unsorted rescans"""
    rng = np.random.default_rng(150)
    grid = plp.make_grid(1241, 376)
    for n, m, words in [(5000, 4000, 0), (8100, 3000, 12)]:
        t, q = MC.random_problem(rng, n, m, n_words=words, cols=1241, rows=376, stereo=True)
        run_landmarks(t, q, 12.0, 0.8, grid)
        for check in (True, False):
            want, wn = O.match_current_and_last(O.grid6(grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF, q["q_valid"], q["q_reproj"],
                                                q["q_x_right"], q["q_level"], q["q_angle"], q["q_desc"], q["q_has_obs"], 15.0, 0, check)
            got, gn = plp.matcher(0.9, check).match_host(plp.MODE_LAST_FRAME, n, m, {**t, **q}, margin=15.0, direction=0, scale_factors=SF, grid=grid)
            assert gn[0] == wn and np.array_equal(got[0], want), (n, m, check)


def test_counts_beyond_the_capacity_are_clamped():
    """t_counts / q_counts come from device memory and cannot be validated on the host: a count larger than the array
    capacity must behave like the capacity (no out-of-bounds reads), a negative one like zero."""
    rng = np.random.default_rng(9)
    grid = plp.make_grid(640, 480)
    n, m = 700, 900
    t, q = MC.random_problem(rng, n, m, n_words=6)
    ref, rn = plp.matcher(0.8, True).match_host(plp.MODE_LANDMARKS, n, m, {**t, **q}, margin=10.0, scale_factors=SF, grid=grid)
    big = dict(t_counts=np.array([n + 1000], np.int32), q_counts=np.array([m + 5000], np.int32))
    got, gn = plp.matcher(0.8, True).match_host(plp.MODE_LANDMARKS, n, m, {**t, **q, **big}, margin=10.0, scale_factors=SF, grid=grid)
    assert gn[0] == rn[0] and np.array_equal(got[0], ref[0])
    got, gn = plp.matcher(0.9, True).match_host(plp.MODE_LAST_FRAME, n, m, {**t, **q, **big}, margin=10.0, direction=0, scale_factors=SF, grid=grid)
    ref2, rn2 = plp.matcher(0.9, True).match_host(plp.MODE_LAST_FRAME, n, m, {**t, **q}, margin=10.0, direction=0, scale_factors=SF, grid=grid)
    assert gn[0] == rn2[0] and np.array_equal(got[0], ref2[0])
    neg = dict(t_counts=np.array([n], np.int32), q_counts=np.array([-3], np.int32))
    got, gn = plp.matcher(0.8, True).match_host(plp.MODE_LANDMARKS, n, m, {**t, **q, **neg}, margin=10.0, scale_factors=SF, grid=grid)
    assert gn[0] == 0 and (got[0] == -1).all()
