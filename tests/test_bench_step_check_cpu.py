"""CPU check of the checker (tests/bench_step_check.py): a step's feature set and match arrays are fabricated from the oracle itself, with the
local-landmark queries given to the oracle in COMPACT form (no padding slots) -- the checker, which feeds the padded 2 x cap layout of the
device path with validity flags, must accept them (padding changes nothing but the query indices) and must notice a corrupted entry."""
import numpy as np

import bench_step_check as BC
import oracle_lib as O
from plp import synth

HALO, K, CAP, LCAP = BC.HALO, 500, 2 * 500 + 64, 64
SHIFT = (np.float32(-3.0), np.float32(0.0))


def fabricate(frames):
    B = len(frames)
    feats = [O.OrbOracle(K).extract(f) for f in frames]
    lines = [O.LineOracle(f) for f in frames]
    kps = np.zeros((HALO + B, CAP), O.KP_DTYPE); desc = np.zeros((HALO + B, CAP, 32), np.uint8); cnt = np.zeros(HALO + B, np.int32)
    kl = np.zeros((HALO + B, LCAP), O.KL_DTYPE); lbd = np.zeros((HALO + B, LCAP, 32), np.uint8); lcnt = np.zeros(HALO + B, np.int32)
    fn = np.zeros((B, LCAP, 3), np.float64)
    rng = np.random.default_rng(0)
    kps["x"] = rng.uniform(0, 640, kps.shape); kps["octave"] = 7      # stale bytes in the padding slots
    for b in range(B):
        k, d = feats[b]
        kps[HALO + b, :len(k)] = k; desc[HALO + b, :len(k)] = d; cnt[HALO + b] = len(k)
        n = min(len(lines[b].keylsd), LCAP)
        kl[HALO + b, :n] = lines[b].keylsd[:n]; lbd[HALO + b, :n] = lines[b].lbd[:n]; lcnt[HALO + b] = n; fn[b, :n] = lines[b].linefn[:n]
    for a in (kps, desc, cnt, kl, lbd, lcnt):
        a[:HALO] = a[-HALO:]
    return dict(kps=kps, desc=desc, cnt=cnt, kl=kl, lbd=lbd, lcnt=lcnt, fn=fn)


def oracle_matches(h, B, g6, sf):
    m = [np.full((B, CAP), -1, np.int32), np.full((B, CAP), -1, np.int32), np.full((B, LCAP), -1, np.int32), np.full((B, LCAP), -1, np.int32)]
    n = [np.zeros(B, np.int32) for _ in range(4)]
    neg = lambda k: np.full(k, -1, np.float32)
    for b in range(B):
        r0, r1, r2 = HALO + b, HALO + b - 1, HALO + b - 2
        c0, c1, c2 = (int(h["cnt"][r]) for r in (r0, r1, r2))
        k0, d0 = h["kps"][r0][:c0], h["desc"][r0][:c0]
        p1, p2 = h["kps"][r1][:c1], h["kps"][r2][:c2]
        re1 = np.stack([p1["x"] + SHIFT[0], p1["y"] + SHIFT[1]], 1).astype(np.float32)
        re2 = np.stack([p2["x"] + np.float32(2) * SHIFT[0], p2["y"] + np.float32(2) * SHIFT[1]], 1).astype(np.float32)
        w, wn = O.match_current_and_last(g6, k0, d0, neg(c0), np.zeros(c0, np.uint8), sf, np.ones(c1, np.uint8), re1, neg(c1), p1["octave"], p1["angle"], h["desc"][r1][:c1],
                                         np.ones(c1, np.uint8), 20.0, 0, True)
        m[0][b, :c0] = w; n[0][b] = wn
        # compact queries [b-2 | b-1]; the device layout puts frame b-1's queries at slots CAP.. of the padded array
        w, wn = O.match_frame_and_landmarks(g6, k0, d0, neg(c0), np.zeros(c0, np.uint8), sf, np.ones(c2 + c1, np.uint8), np.concatenate([re2, re1]), neg(c2 + c1),
                                            np.concatenate([p2["octave"], p1["octave"]]), np.concatenate([h["desc"][r2][:c2], h["desc"][r1][:c1]]), np.ones(c2 + c1, np.uint8),
                                            10.0, 0.8)
        m[1][b, :c0] = np.where(w >= c2, w - c2 + CAP, w); n[1][b] = wn
        l0, l1, l2 = (int(h["lcnt"][r]) for r in (r0, r1, r2))
        kl0, lb0 = h["kl"][r0][:l0], h["lbd"][r0][:l0]
        q1, q2 = h["kl"][r1][:l1], h["kl"][r2][:l2]
        ends = lambda q, mul: (np.stack([q["startPointX"] + np.float32(mul) * SHIFT[0], q["startPointY"] + np.float32(mul) * SHIFT[1]], 1).astype(np.float32),
                               np.stack([q["endPointX"] + np.float32(mul) * SHIFT[0], q["endPointY"] + np.float32(mul) * SHIFT[1]], 1).astype(np.float32))
        sp1, ep1 = ends(q1, 1); sp2, ep2 = ends(q2, 2)
        sf_lsd = np.ones(1, np.float32)
        w, wn = O.match_current_and_last_line(kl0, lb0, np.full((l0, 2), -1, np.float32), np.zeros(l0, np.uint8), sf_lsd, 1, np.ones(l1, np.uint8), sp1, ep1, neg(l1), neg(l1), q1["octave"],
                                              h["lbd"][r1][:l1], np.ones(l1, np.uint8), 20.0, 0, 0)
        m[2][b, :l0] = w; n[2][b] = wn
        kpo = np.zeros(l0, np.int32); kpo[:min(l0, c0)] = k0["octave"][:min(l0, c0)]
        w, wn = O.match_frame_and_landmarks_line(kl0, lb0, kpo, np.zeros(l0, np.uint8), sf_lsd, np.ones(l2 + l1, np.uint8), np.concatenate([sp2, sp1]), np.concatenate([ep2, ep1]),
                                                 np.concatenate([q2["octave"], q1["octave"]]), np.concatenate([h["lbd"][r2][:l2], h["lbd"][r1][:l1]]), np.ones(l2 + l1, np.uint8), 10.0, 0.8)
        m[3][b, :l0] = np.where(w >= l2, w - l2 + LCAP, w); n[3][b] = wn
    return m, n


def test_checker_accepts_the_oracle_chain_and_sees_a_corrupted_match():
    frames = synth.replay(11, 4, 480, 640)
    h = fabricate(frames)
    sf = O.OrbOracle(K).tables()["scale_factors"]

    class G:
        min_x = min_y = 0.0
        inv_cell_width = float(np.float64(64) / np.float64(np.float32(640))); inv_cell_height = float(np.float64(48) / np.float64(np.float32(480)))
        cols, rows = 64, 48
    g6 = O.grid6(G)
    h["m"], h["n"] = oracle_matches(h, len(frames), g6, sf)
    assert h["n"][0].min() > 100 and h["n"][1].min() > 100 and h["n"][2].sum() > 0 and h["n"][3].sum() > 0
    for b in range(len(frames)):
        assert BC.check_frame(h, b, K, g6, SHIFT, sf, frames[b]) == []
    h["m"][1][2, int(np.argmax(h["m"][1][2] >= 0))] += 1
    h["n"][3][1] += 1
    assert len(BC.check_frame(h, 2, K, g6, SHIFT, sf)) == 1 and len(BC.check_frame(h, 1, K, g6, SHIFT, sf)) == 1


def test_halo_checker_accepts_the_predecessors_tail_and_sees_a_wrong_or_stale_row():
    """check_halo (bench.py --verify at N > 1): the halo rows 0..HALO-1 must be the oracle's extraction of the predecessor rank's last two frames.  The fabricated
    step is circular (rows 0..1 = the block's own tail), so its own last two frames are the right answer, any other frames are not, and a damaged row is seen."""
    frames = synth.replay(12, 3, 480, 640)
    h = fabricate(frames)
    assert BC.check_halo(h, K, frames[-HALO:]) == []
    assert len(BC.check_halo(h, K, frames[:HALO])) >= 2                      # the rows of some other frames: both feature kinds of at least one row differ
    h["desc"][1, 3, 0] ^= 1
    bad = BC.check_halo(h, K, frames[-HALO:])
    assert len(bad) == 1 and "halo row 1" in bad[0]
    h["lcnt"][0] -= 1
    assert len(BC.check_halo(h, K, frames[-HALO:])) == 2
