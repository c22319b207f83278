"""Deliberate definition D2, MEASURED (VERDICT r05 "What's weak" 4).

The reference calls single-precision libm in two places of the line path: `cos(float)` / `sin(float)` in LSD's `region_grow` (lsd.cpp, reached from
`/root/reference/src/PLPSLAM/feature/line_descriptor/LSDDetector_custom.cpp:241-257`) and for the LBD direction (`binary_descriptor_custom.cpp`); the oracle and the
HIP path evaluate them as `(float)cos((double)x)` because glibc picks its float variants at run time.  This machine's glibc: `cosf` differs from the rounded f64
result on a fraction of a percent of arguments.  Does that ever change a line?  `liboracle_d2.so` is the strict oracle build with the literal `cosf / sinf`
(`-DORACLE_D2_LIBM_FLOAT`, oracle/lsd_restated.hpp); both builds run on the bench's replay frames and every LSD segment, KeyLine record and LBD row is compared.
Result on this glibc (2.35): 0 differing segments / records / rows -- D2 costs nothing on this replay; the test holds that and prints the counts."""
import importlib

import numpy as np

import oracle_lib as O

synth = importlib.import_module("structure-plp-slam_amd.synth")


def test_cosf_sinf_differ_from_the_rounded_f64_results_on_some_arguments_of_this_glibc():
    a = np.linspace(2.0 ** -6, 2 * np.pi, 200_001).astype(np.float32)
    c = np.zeros_like(a); s = np.zeros_like(a)
    O.lib().oracle_f_cos_sin(a.ctypes.data, len(a), c.ctypes.data, s.ctypes.data)
    with O.variant("liboracle_d2.so"):
        c2 = np.zeros_like(a); s2 = np.zeros_like(a)
        O.lib().oracle_f_cos_sin(a.ctypes.data, len(a), c2.ctypes.data, s2.ctypes.data)
    fc, fs = float(np.mean(c != c2)), float(np.mean(s != s2))
    print(f"cosf != (float)cos((double)x) on {100 * fc:.2f} %, sinf on {100 * fs:.2f} % of {len(a)} arguments in [2^-6, 2 pi)")
    # the two definitions are not the same function (otherwise D2 would not be a definition at all) and never differ by more than one ulp
    assert np.all(np.abs(c.view(np.int32).astype(np.int64) - c2.view(np.int32)) <= 1) and np.all(np.abs(s.view(np.int32).astype(np.int64) - s2.view(np.int32)) <= 1)


def test_no_line_of_the_replay_depends_on_d2():
    frames = synth.replay(1234, 64, 480, 640)            # the frames bench.py replays (rank 0)
    n_seg = n_kept = n_rows = 0
    diff = []
    for i, f in enumerate(frames):
        a = O.LineOracle(f)
        with O.variant("liboracle_d2.so"):
            b = O.LineOracle(f)
        n_seg += len(a.raw); n_kept += len(a.keylsd); n_rows += len(a.all_lbd)
        same = (a.raw.shape == b.raw.shape and np.array_equal(a.raw, b.raw) and np.array_equal(a.all_kl, b.all_kl) and np.array_equal(a.all_lbd, b.all_lbd)
                and np.array_equal(a.keylsd, b.keylsd) and np.array_equal(a.lbd, b.lbd) and np.array_equal(a.linefn, b.linefn))
        if not same:
            diff.append(i)
    print(f"D2 on {len(frames)} replay frames: {n_seg} LSD segments, {n_kept} kept key lines, {n_rows} LBD rows; frames with any difference: {diff}")
    assert n_seg > 10_000 and n_kept > 2_000
    assert not diff, f"frames whose lines depend on cosf / sinf vs (float)cos((double)x): {diff}"
