"""What the choice of LSD's seed order costs the caller (DESIGN.md section 5, D1 -- closed in round 4: the library's default is now the std::sort
order measured here as "unstable", replayed on the device; the stable order is the opt-in PLP_SEED_ORDER_STABLE mode; tests/test_gpu_seed_sort.py
requires 0 differing key lines between the library's default mode and std::sort on exactly these frames).  Original text:
What definition D1 costs the caller.  OpenCV's LSD orders its seeds with std::sort on the gradient bin alone -- an
UNSTABLE sort, so the order inside a bin is whatever the C++ library's algorithm leaves (reached via LSDDetector_custom.cpp:244-257).  The
repository defines the order inside a bin as row-major (oracle and HIP path agree on it); `stable_order=False` makes the oracle call
std::sort as the reference does, i.e. reproduces a reference built with THIS libstdc++.  This test measures, on the fixture frames and the
replay frames of the bench, how far the two are apart in what the caller sees -- key lines after the >= 60 px filter
(feature/line_extractor.cc:134-141), LBD rows, and match_current_and_last_frames_line between consecutive frames -- and pins the bound that
DESIGN.md quotes.  CPU only."""
import numpy as np
from PIL import Image

import oracle_lib as O
from plp import synth

EP = ("startPointX", "startPointY", "endPointX", "endPointY")


def ends(kl):
    return np.stack([kl[k] for k in EP], 1).astype(np.float64) if len(kl) else np.zeros((0, 4))


def test_effect_of_the_seed_order_definition_on_the_callers_output(golden_dir):
    frames = [np.asarray(Image.open(golden_dir / n)) for n in ("equirect1_640x480.png", "equirect2_640x480.png", "equirect1_crop_640x480.png", "equirect2_crop_640x480.png")]
    frames += [synth.canvas(1234, 480, 640)] + list(synth.replay(1234, 64, 480, 640))      # the bench's 64 distinct frames
    n_lines = n_diff = n_far = n_lbd_same = raw_total = raw_diff = 0
    shifts = []
    prev = None
    match_stable = match_unstable = 0
    for i, f in enumerate(frames):
        a, b = O.LineOracle(f, True), O.LineOracle(f, False)
        raw_total += len(a.raw)
        raw_diff += len({tuple(r) for r in a.raw} - {tuple(r) for r in b.raw})
        A, Bm = ends(a.keylsd), ends(b.keylsd)
        n_lines += len(A)
        for k, row in enumerate(A):
            e = np.abs(Bm - row).max(1) if len(Bm) else np.array([1e9])
            j = int(np.argmin(e))
            if e[j] == 0:
                continue
            n_diff += 1
            shifts.append(float(e[j]))
            n_far += e[j] > 3.0
            n_lbd_same += len(Bm) > 0 and np.array_equal(a.lbd[k], b.lbd[j])
        if i >= 5:      # the replay frames: the line matcher of the tracker between consecutive frames, in either order definition
            if prev is not None:
                for cur, old, tag in ((a, prev[0], 0), (b, prev[1], 1)):
                    l0, l1 = len(cur.keylsd), len(old.keylsd)
                    if l0 == 0 or l1 == 0:
                        continue
                    sp = np.stack([old.keylsd["startPointX"] - np.float32(3), old.keylsd["startPointY"]], 1).astype(np.float32)
                    ep = np.stack([old.keylsd["endPointX"] - np.float32(3), old.keylsd["endPointY"]], 1).astype(np.float32)
                    _, wn = O.match_current_and_last_line(cur.keylsd, cur.lbd, np.full((l0, 2), -1, np.float32), np.zeros(l0, np.uint8), np.ones(1, np.float32), 1, np.ones(l1, np.uint8),
                                                          sp, ep, np.full(l1, -1, np.float32), np.full(l1, -1, np.float32), old.keylsd["octave"], old.lbd, np.ones(l1, np.uint8), 20.0, 0, 0)
                    if tag == 0:
                        match_stable += wn
                    else:
                        match_unstable += wn
            prev = (a, b)
    shifts = np.array(shifts)
    report = dict(frames=len(frames), raw_segments=raw_total, raw_differing=raw_diff, key_lines=n_lines, key_lines_differing=n_diff,
                  median_shift_px=float(np.median(shifts)) if len(shifts) else 0.0, p90_shift_px=float(np.percentile(shifts, 90)) if len(shifts) else 0.0,
                  differing_by_more_than_3px=int(n_far), differing_with_identical_lbd=int(n_lbd_same), line_matches_stable=int(match_stable), line_matches_unstable=int(match_unstable))
    print(report)
    # the bound DESIGN.md section 5 states: a few percent of the raw segments and of the key lines differ at all (most of those by a fraction of
    # a pixel: the same edge grown from a neighbouring seed), about 1 % of the key lines are present in one order and absent or elsewhere in the
    # other, and the tracker's line matcher finds the same number of matches to within 2 %
    assert raw_diff <= 0.06 * raw_total
    assert n_diff <= 0.08 * n_lines
    assert n_far <= 0.015 * n_lines
    assert abs(match_stable - match_unstable) <= 0.02 * max(match_stable, 1)
