"""CPU-only checks of the product side: the C-ABI library loads and exports every symbol that
include/plp_front.h declares, the no-GPU behaviour is loud, and the host model of the quadtree
kernel agrees with the oracle's std::list restatement."""
import ctypes as C
import re
import numpy as np
import pytest

import oracle_lib as O
from plp import plp


def test_library_exports_every_declared_symbol():
    L = plp.lib()
    header = (plp.ROOT / "include" / "plp_front.h").read_text()
    declared = set(re.findall(r"\b(plp_[a-z0-9_]+)\s*\(", header))
    declared -= {"plp_status"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in plp_front.h but not exported"
    for name in plp.api_symbols():
        assert hasattr(L, name)
    assert L.plp_version() >= 1
    assert L.plp_strerror(0) == b"ok"


def test_no_gpu_is_loud():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert plp.lib().plp_device_count() == 0
    with pytest.raises(plp.PlpError) as e:
        plp.orb_extractor()
    assert e.value.status == plp.PLP_ERR_NO_DEVICE
    # every other context of the product path refuses as well: there is no CPU fallback anywhere
    for make in (plp.LineFeatureTracker, plp.matcher,
                 lambda: plp.bow_vocabulary(1, [-1, 0, 0], [False, True, True], np.zeros((3, 32), np.uint8), [0.0, 1.0, 1.0])):
        with pytest.raises(plp.PlpError) as e:
            make()
        assert e.value.status in (plp.PLP_ERR_NO_DEVICE, plp.PLP_ERR_HIP)


def test_param_validation_matches_orb_params():
    """orb_params.cc:40-54: bad mask rectangles throw -> PLP_ERR_INVALID_ARG before any device use."""
    for bad in ([[0.5, 0.2, 0.3, 0.8]], [[0.2, 0.5, 0.8, 0.3]]):
        with pytest.raises(ValueError):
            plp.orb_extractor(mask_rects=bad)


@pytest.mark.parametrize("seed", range(6))
def test_quadtree_model_equals_oracle_list(seed):
    rng = np.random.default_rng(seed)
    ex = O.OrbOracle()
    for trial in range(25):
        w, h = [(640, 480), (1241, 376), (600, 1200), (int(rng.integers(60, 700)), int(rng.integers(60, 700)))][trial % 4]
        bw, bh = w - 38, h - 38
        n = int(rng.integers(0, min(3000, (bw - 6) * (bh - 6) // 4)))
        quota = int(rng.integers(0, 900))
        flat = rng.choice((bw - 6) * (bh - 6), size=n, replace=False)
        pts = np.stack([flat % (bw - 6) + 3, flat // (bw - 6) + 3], 1)
        if trial % 3 == 0 and n:   # clustered
            c = pts[rng.integers(0, n, 4)]
            pts = np.unique(np.clip(c[rng.integers(0, 4, n)] + rng.normal(0, 12, (n, 2)).astype(int), 3, [bw - 4, bh - 4]), axis=0)
            rng.shuffle(pts)
            n = len(pts)
        score = rng.integers(7, 13 if trial % 2 else 255, n)   # narrow range -> many response ties
        cands = np.zeros(n, O.KP_DTYPE)
        cands["x"], cands["y"], cands["response"] = pts[:, 0], pts[:, 1], score
        ref = ex.distribute(cands, 19, w - 19, 19, h - 19, quota)
        xys = np.stack([pts[:, 0], pts[:, 1], score], 1).astype(np.int32) if n else np.zeros((0, 3), np.int32)
        pick = plp.model_quadtree(xys, w, h, quota)
        got = xys[pick]
        want = np.stack([ref["x"], ref["y"], ref["response"]], 1).astype(np.int32)
        assert np.array_equal(got, want), (w, h, n, quota)


def test_sincos_fast_path_equals_libm_on_every_gradient_angle():
    """csrc/sincos_ziv.hpp (the gradient kernel's cos / sin of the level-line angle): wherever its rounding test succeeds the result
    must be (float)cos((double)a) / (float)sin((double)a) exactly (definition D2) -- checked for EVERY angle the kernel can see (all
    (gx, gy) with |g| <= 510, through cv::fastAtan2 and the degree -> radian conversion of lsd.cpp) and for 4 M random floats in [0, 2 pi];
    where the test fails (a few arguments per million) the kernel calls the general f64 routine."""
    g = np.arange(-510, 511, dtype=np.float32)
    gx, gy = np.meshgrid(g, g)
    deg = np.array([O.lib().oracle_fast_atan2(float(y), float(x)) for x, y in zip(gx.ravel()[::97], gy.ravel()[::97])], np.float32)
    # the full grid through a vectorised restatement of cv::fastAtan2 (checked against the oracle's on the sample above)
    def fast_atan2(y, x):
        f = np.float32
        p1, p3, p5, p7 = (f(0.9997878412794807) * f(180 / np.pi), f(-0.3258083974640975) * f(180 / np.pi), f(0.1555786518463281) * f(180 / np.pi),
                          f(-0.04432655554792128) * f(180 / np.pi))
        ax, ay = np.abs(x), np.abs(y)
        eps = f(2.2204460492503131e-16)
        big = ax >= ay
        num, den = np.where(big, ay, ax), np.where(big, ax, ay) + eps
        c = (num / den).astype(f); c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
        a = np.where(big, a, f(90) - a)
        a = np.where(x < 0, f(180) - a, a)
        a = np.where(y < 0, f(360) - a, a)
        return a.astype(f)
    full = fast_atan2(gy.ravel(), gx.ravel())
    assert np.array_equal(full[::97], deg)
    rng = np.random.default_rng(5)
    args = np.concatenate([(full.astype(np.float64) * (np.pi / 180)).astype(np.float32), rng.uniform(0, 2 * np.pi, 4_000_000).astype(np.float32),
                           np.array([0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi], np.float32)])
    c, s, ok = plp.model_sincos(args)
    want_c, want_s = O.f_cos_sin(args)
    assert ok.mean() > 0.9999
    assert np.array_equal(c[ok], want_c[ok]) and np.array_equal(s[ok], want_s[ok])


def test_bin_ranking_model_equals_std_sort():
    """csrc/libstdcxx_sort.hpp (the matchers' orientation check ranks the 30 histogram bins as the reference's std::sort does) against
    the real std::sort of this machine's library: random bin sizes with many ties, sorted / reversed / organ-pipe shapes, other
    lengths up to 64."""
    rng = np.random.default_rng(3)
    for trial in range(20000):
        n = 30 if trial % 4 else int(rng.integers(1, 65))
        sizes = rng.integers(0, (2, 3, 5, 50, 1000)[trial % 5], n).astype(np.int32)
        if trial % 7 == 4: sizes = np.sort(sizes)
        if trial % 7 == 5: sizes = np.sort(sizes)[::-1].copy()
        if trial % 7 == 6: sizes = np.concatenate([np.sort(sizes[:n // 2]), np.sort(sizes[n // 2:])[::-1]]).astype(np.int32)
        assert np.array_equal(plp.model_index_sort(sizes), O.index_sort_by_size(sizes)), (trial, sizes)


def test_facade_check_programs_are_built_and_load():
    """The C++ facade check programs (oracle/facade_*_check, oracle/_ref/facade_orb_check) are compiled by build(); without
    arguments they only print nothing and return 2, which proves that they link against libplp_front.so / liboracle.so."""
    import os
    import subprocess
    root = plp.ROOT
    exes = [root / "oracle" / "facade_match_check", root / "oracle" / "facade_line_check"]
    if (root / "oracle" / "_ref" / "facade_orb_check").exists():
        exes.append(root / "oracle" / "_ref" / "facade_orb_check")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    for exe in exes:
        assert exe.exists(), f"{exe} missing: run python -c 'import __graft_entry__ as g; g.build()'"
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60, env=env)
        assert r.returncode == 2, (exe, r.returncode, r.stderr)
