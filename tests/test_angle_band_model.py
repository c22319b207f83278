"""The guard band of region_grow's angle test (structure-plp-slam_amd/csrc/line_kernels.hip, region_grow).

The reference decides `|angle(pixel) - theta| <= prec` with theta = cv::fastAtan2(sum) (modules/imgproc/src/lsd.cpp
isAligned / region_grow).  The kernel decides most pixels WITHOUT theta: it compares the cosine between the pixel's unit
vector and the sum vector with cos(prec -+ band) and evaluates the reference's test only for pixels inside the band.  That
is bit-exact with the reference iff no pixel that is "certain" by the cosine is decided differently by the reference.
This test checks exactly that on CPU, densely around the tolerance, for both renderings of the cosine the kernel has
(separate f32 multiplies and adds: the C++ path; fused multiply-adds: the hand-scheduled block) and with the reciprocal
square root off by up to two ulps either way (v_rsq_f32 is good to one).
"""
import pathlib
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = pathlib.Path(__file__).resolve().parents[1]
F = np.float32


def constants_from_source():
    src = (ROOT / "structure-plp-slam_amd" / "csrc" / "line_device.hpp").read_text()
    get = lambda name: float(re.search(name + r"\s*=\s*([0-9.eE+-]+)", src).group(1))
    return get("kLsdAngleBand"), get("kLsdBandMinPrec"), get("kLsdBandMaxPrec")


def fast_atan2_deg(y, x):
    """cv::fastAtan2 in f32, vectorised (oracle/cv_restated.hpp fast_atan2f_deg is the scalar form it is checked against)."""
    y, x = y.astype(F), x.astype(F)
    k = F(180 / np.pi)
    p1, p3, p5, p7 = F(0.9997878412794807) * k, F(-0.3258083974640975) * k, F(0.1555786518463281) * k, F(-0.04432655554792128) * k
    eps = F(2.2204460492503131e-16)
    ax, ay = np.abs(x), np.abs(y)
    big = ax >= ay
    num, den = np.where(big, ay, ax), np.where(big, ax, ay) + eps
    c = (num / den).astype(F)
    c2 = (c * c).astype(F)
    poly = ((((p7 * c2).astype(F) + p5).astype(F) * c2).astype(F) + p3).astype(F)
    poly = (((poly * c2).astype(F) + p1).astype(F) * c).astype(F)
    a = np.where(big, poly, (F(90) - poly).astype(F))
    a = np.where(x < 0, (F(180) - a).astype(F), a)
    a = np.where(y < 0, (F(360) - a).astype(F), a)
    return a.astype(F)


def aligned_to(a, theta, prec):
    n = np.abs(theta - a)
    n = np.where(n > 1.5 * np.pi, np.abs(n - 2 * np.pi), n)
    return n <= prec


def test_vectorised_fast_atan2_equals_the_oracle():
    rng = np.random.default_rng(5)
    y, x = rng.normal(0, 50, 4000).astype(F), rng.normal(0, 50, 4000).astype(F)
    got = fast_atan2_deg(y, x)
    want = np.array([O.fast_atan2(float(a), float(b)) for a, b in zip(y, x)], F)
    assert np.array_equal(got, want)


def nudge(v, ulps):
    return (v.view(np.int32) + np.int32(ulps)).view(F)


def wrong_certain_decisions(prec, band, n, seed):
    """samples around the tolerance -> (certain decisions that contradict the reference, largest share of uncertain samples, both outcomes seen)"""
    rng = np.random.default_rng(seed)
    # pixel angles as the gradient stage stores them: f32 degrees from fastAtan2, used as (double)deg * pi/180
    deg = rng.uniform(0, 360, n).astype(F)
    a = deg.astype(np.float64) * (np.pi / 180)
    # the region sums what the reference sums: (float)cos / (float)sin of the pixel's angle
    cx, cy = np.cos(a).astype(F), np.sin(a).astype(F)
    # sum vectors: direction at the tolerance on either side, densely within a few bands of it (plus a wide sprinkle), any length
    off = np.where(rng.uniform(size=n) < 0.85, rng.uniform(-4 * band, 4 * band, n), rng.uniform(-prec, np.pi - prec, n))
    ang = a + rng.choice([-1.0, 1.0], n) * (prec + off)
    r = np.exp(rng.uniform(np.log(0.9), np.log(6000.0), n))
    sx, sy = (r * np.cos(ang)).astype(F), (r * np.sin(ang)).astype(F)
    theta = fast_atan2_deg(sy, sx).astype(np.float64) * (np.pi / 180)
    ref = aligned_to(a, theta, prec)
    c_pass, c_fail = F(np.cos(prec - band)), F(np.cos(prec + band))

    d = lambda v: v.astype(np.float64)
    # C++ path: separate roundings (-ffp-contract=off)
    n2_sep = ((sx * sx).astype(F) + (sy * sy).astype(F)).astype(F)
    dot_sep = ((cx * sx).astype(F) + (cy * sy).astype(F)).astype(F)
    # hand-scheduled block: v_mul + v_fmac (one rounding of the exact a*b + c)
    n2_fma = (d(sy) * d(sy) + d((sx * sx).astype(F))).astype(F)
    dot_fma = (d(cy) * d(sy) + d((cx * sx).astype(F))).astype(F)
    wrong, uncertain = 0, 0
    for n2, dot in ((n2_sep, dot_sep), (n2_fma, dot_fma)):
        inv0 = (1.0 / np.sqrt(d(n2))).astype(F)
        for ulps in (-2, -1, 0, 1, 2):
            cosang = (dot * nudge(inv0, ulps)).astype(F)
            certain_pass, certain_fail = cosang >= c_pass, cosang < c_fail
            wrong += int(np.count_nonzero(certain_pass & ~ref)) + int(np.count_nonzero(certain_fail & ref))
            uncertain = max(uncertain, int(np.count_nonzero(~certain_pass & ~certain_fail)))
    return wrong, uncertain / n, bool(np.any(ref) and np.any(~ref))


@pytest.mark.parametrize("prec", ["min", 0.0125, 0.02, 0.05, 0.2, np.pi / 8, 0.7, 1.4, "max"])
def test_a_pixel_certain_by_the_cosine_is_decided_the_same_by_the_reference(prec):
    band, lo, hi = constants_from_source()
    prec = lo if prec == "min" else np.nextafter(hi, 0) if prec == "max" else prec
    assert lo <= prec < hi
    wrong, uncertain, both = wrong_certain_decisions(prec, band, 600_000, int(prec * 1e6))
    assert wrong == 0
    assert uncertain < 0.3   # the band is narrow: with 85 % of the samples within four bands of the tolerance most are certain
    assert both              # not vacuous


def test_tolerances_below_the_limit_are_why_there_is_a_limit():
    """A tolerance of 0.0009 rad (above twice the band, the limit this kernel had first) is NOT safe: cos(t -+ band) are a few f32
    steps apart there.  The kernel sends such tolerances down the exact path (kLsdBandMinPrec)."""
    band, lo, _ = constants_from_source()
    assert 0.0009 < lo
    wrong, _, _ = wrong_certain_decisions(0.0009, band, 200_000, 9)
    assert wrong > 0
