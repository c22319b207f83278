"""One gather round of region_grow: the two renderings the kernel has of "accept the neighbours in scan order" against
the reference's plain statement (structure-plp-slam_amd/csrc/line_kernels.hip region_grow; cv lsd.cpp region_grow).

Reference semantics for the 63 lanes of a round (7 region points x 3x3 neighbours, ascending lane = the reference's scan
order): a lane is accepted iff it is a candidate, its pixel has not been accepted from a lower lane, and
isAligned(pixel, fastAtan2(current sum)) holds; an acceptance adds the pixel's unit vector to the sum (f32).

Rendering 1 (C++ loop): cosine against cos(prec -+ band); the lowest certain-pass lane is accepted unless a lane inside the
band comes before it, in which case the reference's test decides all eligible lanes.
Rendering 2 (hand-scheduled block): the same with fused multiply-adds, taken only while NO eligible lane is inside the band;
otherwise one decision by rendering 1, then back.

Rendering 3 (round 5, settle_block): the whole round settled together from the start sum and verified lane by lane against each lane's own sum; kept only
when every check holds, otherwise the loop (rendering 2 with rendering 1) runs unchanged.

This is a model of the control logic (the device code itself is checked on the GPU by tests/test_gpu_line.py); it shows that
both renderings are the reference's scan for any placement of band pixels and duplicates.
"""
import numpy as np
import pytest

from test_angle_band_model import F, aligned_to, constants_from_source, fast_atan2_deg


def f32(x):
    return F(x)


def exact_test(a, s, prec):
    theta = float(fast_atan2_deg(np.array([s[1]], F), np.array([s[0]], F))[0]) * (np.pi / 180)
    return bool(aligned_to(np.array([a]), np.array([theta]), prec)[0])


def cos_sep(c, s):
    n2 = f32(f32(s[0] * s[0]) + f32(s[1] * s[1]))
    dot = f32(f32(c[0] * s[0]) + f32(c[1] * s[1]))
    return f32(dot * f32(1.0 / np.sqrt(np.float64(n2))))


def cos_fma(c, s):
    d = np.float64
    n2 = f32(d(s[1]) * d(s[1]) + d(f32(s[0] * s[0])))
    dot = f32(d(c[1]) * d(s[1]) + d(f32(c[0] * s[0])))
    return f32(dot * f32(1.0 / np.sqrt(d(n2))))


def reference_round(cand, pix, ang, cs, s, prec):
    taken, acc = set(), []
    for lane in range(63):
        if cand[lane] and pix[lane] not in taken and exact_test(ang[lane], s, prec):
            acc.append(lane); taken.add(pix[lane])
            s = (f32(s[0] + cs[lane][0]), f32(s[1] + cs[lane][1]))
    return acc, s


def careful_step(elig, pix, ang, cs, s, prec, c_pass, c_fail, cosine):
    """one pass of the C++ loop body: -> accepted lane or None"""
    cosang = {l: cosine(cs[l], s) for l in elig}
    P = [l for l in elig if cosang[l] >= c_pass]
    U = [l for l in elig if not cosang[l] >= c_pass and not cosang[l] < c_fail]
    first = P[0] if P else 64
    bal = P
    if any(u < first for u in U):
        bal = [l for l in elig if exact_test(ang[l], s, prec)]
    return bal[0] if bal else None


def run_round(cand, pix, ang, cs, s, prec, band, hand_scheduled):
    c_pass, c_fail = f32(np.cos(prec - band)), f32(np.cos(prec + band))
    live = [l for l in range(63) if cand[l]]
    gt, acc = -1, []

    def accept(k):
        nonlocal s, gt, live
        acc.append(k)
        s = (f32(s[0] + cs[k][0]), f32(s[1] + cs[k][1]))
        gt = k
        live = [l for l in live if pix[l] != pix[k]]

    while True:
        if hand_scheduled:
            while True:   # the block: certain decisions only
                elig = [l for l in live if l > gt]
                if not elig:
                    return acc, s
                cosang = {l: cos_fma(cs[l], s) for l in elig}
                if any(not cosang[l] >= c_pass and not cosang[l] < c_fail for l in elig):
                    break   # an eligible lane inside the band: one careful decision
                P = [l for l in elig if cosang[l] >= c_pass]
                if not P:
                    return acc, s
                accept(P[0])
        elig = [l for l in live if l > gt]
        if not elig:
            return acc, s
        k = careful_step(elig, pix, ang, cs, s, prec, c_pass, c_fail, cos_sep)
        if k is None:
            return acc, s
        accept(k)


def settle_block(cand, pix, cs, s, prec, band):
    """The round settled together (round 5, region_grow's first block): A = the candidates that pass with certainty against the START sum, one per pixel; a chain over A
    in lane order gives every lane the sum after the accepted lanes below it (same f32 additions, same order); every lane whose pixel nobody took is tested against ITS
    sum and must decide as A says, none inside the band.  -> (accepted lanes, sum) or None (nothing kept: the loop runs as if the block did not exist)."""
    c_pass, c_fail = f32(np.cos(prec - band)), f32(np.cos(prec + band))
    live = [l for l in range(63) if cand[l]]
    cos0 = {l: cos_fma(cs[l], s) for l in live}
    if any(not cos0[l] >= c_pass and not cos0[l] < c_fail for l in live):
        return None
    p = [l for l in live if cos0[l] >= c_pass]
    acc, shadow, lane_sum = [], set(), {l: s for l in range(63)}
    while p:
        k = p[0]
        acc.append(k)
        s = (f32(s[0] + cs[k][0]), f32(s[1] + cs[k][1]))
        for l in range(k + 1, 63):
            lane_sum[l] = s
        same = {l for l in range(63) if pix[l] == pix[k]}
        shadow |= same
        p = [l for l in p if l not in same]
    checked = [l for l in live if l not in shadow] + acc
    for l in checked:
        c = cos_fma(cs[l], lane_sum[l])
        if not c >= c_pass and not c < c_fail:
            return None
        if (c >= c_pass) != (l in acc):
            return None
    return acc, s


def random_round(rng, prec, band, p_near=0.35):
    # seven region points close together on a grid: their 3x3 neighbourhoods overlap (the same pixel in several lanes)
    base = np.array([50, 50]) + rng.integers(-1, 2, (7, 2)).cumsum(0)
    pix, cand = [], []
    for p in base:
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = p + (dx, dy)
                pix.append(int(q[1]) * 1000 + int(q[0])); cand.append(bool(rng.uniform() < 0.6))
    nreg = int(rng.choice([1, 2, 5, 20, 300]))
    t0 = rng.uniform(0, 2 * np.pi)
    s = (f32(nreg * np.cos(t0)), f32(nreg * np.sin(t0)))
    # per PIXEL angle (duplicates share it): aligned, clearly off, or within a few bands of the tolerance
    per_pixel = {}
    for q in set(pix):
        kind = rng.uniform()
        off = rng.uniform(-0.8 * prec, 0.8 * prec) if kind < 0.4 else rng.choice([-1, 1]) * (prec + rng.uniform(-3 * band, 3 * band)) if kind < 0.4 + p_near \
            else rng.uniform(prec * 1.2, np.pi)
        deg = F(np.degrees((t0 + off) % (2 * np.pi)))
        per_pixel[q] = float(deg) * (np.pi / 180)
    ang = [per_pixel[q] for q in pix]
    cs = [(f32(np.cos(a)), f32(np.sin(a))) for a in ang]
    return cand, pix, ang, cs, s


@pytest.mark.parametrize("prec", [np.pi / 8, 0.05, 0.3])
def test_both_renderings_of_a_round_are_the_reference_scan(prec):
    band, lo, hi = constants_from_source()
    assert lo <= prec < hi
    rng = np.random.default_rng(int(prec * 1000))
    multi, band_rounds, settled, settled_multi = 0, 0, 0, 0
    for it in range(800):
        # the first 400 rounds are adversarial (a third of the pixels within three bands of the tolerance: the block rarely keeps its result there),
        # the second 400 look like an image (few pixels near the tolerance)
        cand, pix, ang, cs, s = random_round(rng, prec, band, 0.35 if it < 400 else 0.01)
        want = reference_round(cand, pix, ang, cs, s, prec)
        for hand in (False, True):
            got = run_round(cand, pix, ang, cs, s, prec, band, hand)
            assert got[0] == want[0] and got[1] == want[1], (hand, got, want)
        st = settle_block(cand, pix, cs, s, prec, band)      # whenever the block keeps its result, it is the reference's
        if st is not None:
            assert st[0] == want[0] and st[1] == want[1], (st, want)
            settled += 1
            settled_multi += len(want[0]) >= 2
        multi += len(want[0]) >= 3
        c_pass, c_fail = f32(np.cos(prec - band)), f32(np.cos(prec + band))
        band_rounds += any(cand[l] and not cos_sep(cs[l], s) >= c_pass and not cos_sep(cs[l], s) < c_fail for l in range(63))
    assert multi > 50 and band_rounds > 20   # rounds with several acceptances and rounds with band pixels both occur
    assert settled > 150 and settled_multi > 60   # and the block does settle rounds, several acceptances among them
