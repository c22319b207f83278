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


class _Grid:
    def __init__(self, cols_px=640, rows_px=480, cols=64, rows=48):
        self.min_x, self.min_y = 0.0, 0.0
        self.inv_cell_width, self.inv_cell_height = cols / float(cols_px), rows / float(rows_px)
        self.cols, self.rows = cols, rows


SF8 = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
SF_LSD = np.array([1.0, 2.0], np.float32)


def triangulation_problem(rng, n, m, nodes, words):
    t, q = random_problem(rng, n, m, n_words=words)
    t_node = (t["t_desc"][:, 0].astype(np.int32) * 7 + 3) % nodes
    q_node = (q["q_desc"][:, 0].astype(np.int32) * 7 + 3) % nodes
    order = np.argsort(q_node, kind="stable")
    qd, qa, qn = q["q_desc"][order], q["q_angle"][order], q_node[order]
    q_has_lm = (rng.uniform(size=m) < 0.3).astype(np.uint8); t_has_lm = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    q_xr = np.where(rng.uniform(size=m) < 0.3, 100.0, -1.0).astype(np.float32); t_xr = np.where(rng.uniform(size=n) < 0.3, 100.0, -1.0).astype(np.float32)
    q_oct = rng.integers(0, 8, m).astype(np.int32)
    tr = np.array([0.3, 0.02, 0.05])
    E = np.array([[0, -tr[2], tr[1]], [tr[2], 0, -tr[0]], [-tr[1], tr[0], 0]], np.float64)
    epipole = -tr / np.linalg.norm(tr)
    pts_t = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1, 8, n)], 1)
    src = rng.integers(0, n, m)
    pts_q = pts_t[src] + rng.normal(0, 0.004, (m, 3)) * pts_t[src][:, 2:3]
    b1 = pts_q / np.linalg.norm(pts_q, axis=1, keepdims=True)
    p2 = pts_t - tr
    p2[rng.uniform(size=n) < 0.05] = epipole * 3.0 + rng.normal(0, 0.01, 3)
    b2 = p2 / np.linalg.norm(p2, axis=1, keepdims=True)
    return (qd, qa, qn, q_has_lm, q_xr, q_oct, b1, t["t_desc"], t["t_kps"]["angle"], t_node, t_has_lm, t_xr, b2, SF8, E.ravel(), epipole)


def matcher_cases(rng, scale=1.0):
    """One random problem for every array-form matcher: a list of (label, name of the wrapper in oracle_lib, argument tuple).
    The same list is run through the oracle, through the reference build (oracle_lib.reference()) and -- from the committed golden
    file -- through the HIP path."""
    g = _Grid()
    g6 = O.grid6(g)
    S = lambda v: max(1, int(v * scale))
    out = []
    words = int(rng.choice([0, 3, 12]))
    n, m = int(rng.integers(1, S(1200))), int(rng.integers(1, S(1800)))
    t, q = random_problem(rng, n, m, n_words=words, stereo=bool(rng.integers(0, 2)))
    margin, ratio = float(rng.uniform(5, 60)), float(rng.choice([0.6, 0.8, 0.9]))
    out.append(("landmarks", "match_frame_and_landmarks", (g6, t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF8, q["q_valid"], q["q_reproj"],
                                                           q["q_x_right"], q["q_level"], q["q_desc"], q["q_has_obs"], margin, ratio)))
    direction, check = int(rng.integers(0, 3)), bool(rng.integers(0, 2))
    out.append(("last_frame", "match_current_and_last", (g6, t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], SF8, q["q_valid"], q["q_reproj"],
                                                         q["q_x_right"], q["q_level"], q["q_angle"], q["q_desc"], q["q_has_obs"], margin, direction, check)))
    pred = q["q_level"].astype(np.uint32)
    thr = int(rng.choice([50, 100]))
    out.append(("frame_keyframe", "match_frame_and_keyframe", (g6, t["t_kps"], t["t_desc"], t["t_occupied"], SF8, q["q_valid"], q["q_reproj"], pred, q["q_angle"],
                                                               q["q_desc"], 10.0, thr, check)))
    out.append(("sim3", "match_by_sim3", (g6, t["t_kps"], t["t_desc"], t["t_occupied"], SF8, q["q_valid"], q["q_reproj"], pred, q["q_desc"], 7.5)))
    reproj_d = q["q_reproj"].astype(np.float64) + rng.normal(0, 0.7, (m, 2))
    predr = rng.integers(0, 8, m).astype(np.uint32)
    inv_sigma = (1.0 / (SF8 * SF8)).astype(np.float32)
    out.append(("fuse", "fuse_search", (g6, t["t_kps"], t["t_desc"], t["t_x_right"], SF8, inv_sigma, q["q_valid"], reproj_d, q["q_x_right"], predr, q["q_desc"], 3.0)))
    out.append(("detect_duplication", "project_best", (g6, t["t_kps"], t["t_desc"], SF8, q["q_valid"], reproj_d, pred, q["q_desc"], 4.0, 50, 1)))
    out.append(("brute_force", "brute_force_match", (t["t_desc"][:S(500)], t["t_kps"]["angle"][:S(500)], q["q_desc"][:S(500)], q["q_angle"][:S(500)],
                                                     q["q_valid"][:S(500)], 0.75, check)))
    nodes = int(rng.integers(1, 40))
    t_node = (t["t_desc"][:, 0].astype(np.int32) * 7 + 3) % nodes
    q_node = (q["q_desc"][:, 0].astype(np.int32) * 7 + 3) % nodes
    order = np.argsort(q_node, kind="stable")
    tskip = t["t_occupied"] if rng.integers(0, 2) else np.zeros(n, np.uint8)
    out.append(("bow", "match_bow", (q["q_desc"][order], q["q_angle"][order], q_node[order], q["q_valid"][order], t["t_desc"], t["t_kps"]["angle"], t_node,
                                     tskip, 0.75, check)))
    out.append(("triangulation", "match_for_triangulation", triangulation_problem(rng, int(rng.integers(1, S(900))), int(rng.integers(1, S(900))), nodes, words)
                + (check,)))
    # area: level-0 key points only take part; few distinct descriptors -> "steal if closer"
    t1, _ = random_problem(rng, int(rng.integers(1, S(600))), 3, n_words=5); t2, _ = random_problem(rng, int(rng.integers(1, S(700))), 3, n_words=5)
    t1["t_kps"]["octave"] = rng.integers(0, 2, len(t1["t_kps"])); t2["t_kps"]["octave"] = rng.integers(0, 2, len(t2["t_kps"]))
    prev = np.stack([t1["t_kps"]["x"], t1["t_kps"]["y"]], 1).astype(np.float32)
    out.append(("area", "match_area", (g6, t1["t_kps"], t1["t_desc"], t2["t_kps"], t2["t_desc"], prev, int(rng.integers(20, 200)), 0.9, check)))
    # lines
    nl, ml = int(rng.integers(1, S(300))), int(rng.integers(1, S(400)))
    tl, ql = random_line_problem(rng, nl, ml, int(rng.choice([0, 2, 4])))
    lmargin = float(rng.uniform(3, 30))
    out.append(("landmarks_line", "match_frame_and_landmarks_line", (tl["t_kl"], tl["t_desc"], tl["t_kp_octave"], tl["t_occupied"], SF_LSD, ql["q_valid"], ql["q_reproj"],
                                                                     ql["q_reproj2"], ql["q_level"], ql["q_desc"], ql["q_has_obs"], lmargin, 0.8)))
    xr_pair = np.stack([tl["t_x_right"], tl["t_x_right2"]], 1)
    out.append(("last_frame_line", "match_current_and_last_line", (tl["t_kl"], tl["t_desc"], xr_pair, tl["t_occupied"], SF_LSD, 1, ql["q_valid"], ql["q_reproj"],
                                                                   ql["q_reproj2"], ql["q_x_right"], ql["q_x_right2"], ql["q_level"], ql["q_desc"], ql["q_has_obs"],
                                                                   lmargin, direction, int(rng.integers(0, 2)))))
    lpred = ql["q_level"].astype(np.uint32)
    out.append(("frame_keyframe_line", "match_frame_and_keyframe_line", (tl["t_kl"], tl["t_desc"], tl["t_occupied"], SF_LSD, ql["q_valid"], ql["q_reproj"], ql["q_reproj2"],
                                                                         lpred, ql["q_desc"], 12.0, 60)))
    sp_d = ql["q_reproj"].astype(np.float64) + rng.normal(0, 0.2, (ml, 2)); ep_d = ql["q_reproj2"].astype(np.float64) + rng.normal(0, 0.2, (ml, 2))
    out.append(("fuse_line", "fuse_search_line", (tl["t_kl"], tl["t_desc"], SF_LSD, np.array([1.0, 0.25], np.float32), ql["q_valid"], sp_d, ep_d, lpred, ql["q_desc"], 6.0)))
    # MIH 1-NN over LBD: few words -> distance ties decided by the discovery order
    nq, nt = int(rng.integers(1, S(200))), int(rng.integers(1, S(200)))
    vocab = rng.integers(0, 256, (int(rng.integers(1, 6)), 32), dtype=np.uint8)
    tq = vocab[rng.integers(0, len(vocab), nq)].copy(); tt = vocab[rng.integers(0, len(vocab), nt)].copy()
    for arr in (tq, tt):
        k = len(arr)
        for _ in range(int(rng.integers(0, 4))):
            arr[np.arange(k), rng.integers(0, 32, k)] ^= (np.uint8(1) << rng.integers(0, 8, k).astype(np.uint8)) * (rng.uniform(size=k) < 0.5).astype(np.uint8)
    out.append(("lbd_1nn", "lbd_match_1nn", (tq, tt)))
    return out


def load_golden_match(path=None):
    """tests/golden/ref_match.npz (tools/make_golden_ref.py): [(seed, label, wrapper name, args tuple, reference outputs tuple, extras)]"""
    import pathlib
    path = path or pathlib.Path(__file__).resolve().parent / "golden" / "ref_match.npz"
    z = np.load(path)
    blobs = {}
    for k in z.files:
        if k.startswith("blob"):
            idx = int(k[4:].split("__")[0])
            a = z[k]
            if "__rec" in k:
                a = np.ascontiguousarray(a).view(O.KP_DTYPE if a.shape[1] == 28 else O.KL_DTYPE).reshape(-1)
            blobs[idx] = a
    cases = []
    for k in z.files:
        if not k.endswith("__fn"):
            continue
        base = k[:-4]
        seed, label = base.split("__")
        spec = z[base + "__args"]
        args = []
        for i, b in enumerate(spec):
            if b >= 0:
                args.append(blobs[int(b)])
            else:
                v = z[f"{base}__arg{i}"]
                args.append(bool(v) if v.dtype == np.bool_ else (float(v) if v.dtype.kind == "f" else int(v)))
        outs = []
        while f"{base}__out{len(outs)}" in z.files:
            o = z[f"{base}__out{len(outs)}"]
            outs.append(o if o.ndim else int(o))
        extras = {"defined": z[base + "__defined"]} if base + "__defined" in z.files else {}
        cases.append((int(seed[1:]), label, str(z[k]), tuple(args), tuple(outs), extras))
    return sorted(cases, key=lambda c: (c[0], c[1]))
