"""The shipped C++ facade header (structure-plp-slam_amd/facade/PLPSLAM/feature/orb_extractor.h), compiled against the
reference's own orb_params and the OpenCV type shim into oracle/_ref/facade_orb_check (oracle/ref_build.sh), run on the GPU
and compared with the oracle: cv::Mat in, std::vector<cv::KeyPoint> / cv::Mat out, getters, setter, public image_pyramid_."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from plp import synth

pytestmark = pytest.mark.gpu
_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "facade_orb_check")


@pytest.mark.skipif(not os.path.exists(_EXE), reason="oracle/_ref/facade_orb_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("K", [1000, 2000])
def test_cpp_facade_equals_oracle(tmp_path, K):
    img = synth.replay(31, 1, 480, 640)[0]
    raw, out = tmp_path / "img.raw", tmp_path / "out.bin"
    raw.write_bytes(np.ascontiguousarray(img).tobytes())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([_EXE, str(raw), "480", "640", str(K), str(out)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    b = out.read_bytes()
    n, nl, mk = struct.unpack_from("<iii", b, 0)
    off = 12
    sf = np.frombuffer(b, np.float32, nl, off); off += 4 * nl
    isq = np.frombuffer(b, np.float32, nl, off); off += 4 * nl
    kps = np.frombuffer(b, O.KP_DTYPE, n, off); off += 28 * n
    desc = np.frombuffer(b, np.uint8, 32 * n, off).reshape(n, 32); off += 32 * n
    orc = O.OrbOracle(K)
    ok, od = orc.extract(img)
    t = orc.tables()
    assert mk == K and nl == 8
    assert np.array_equal(sf, t["scale_factors"]) and np.array_equal(isq, t["inv_level_sigma_sq"])
    assert n == len(ok) and np.array_equal(kps, ok) and np.array_equal(desc, od)
    for l in range(nl):                       # image_pyramid_ host copies
        rows, cols, s = struct.unpack_from("<iiI", b, off); off += 12
        lvl = orc.level_image(l)
        assert (rows, cols) == lvl.shape and s == int(lvl.astype(np.uint64).sum())


_MATCH_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "facade_match_check")


@pytest.mark.skipif(not os.path.exists(_MATCH_EXE), reason="oracle/facade_match_check not built (make -C oracle after the product library)")
@pytest.mark.parametrize("seed,n,m", [(1, 900, 1200), (2, 2000, 2600), (3, 40, 25), (4, 1500, 300)])
def test_cpp_matcher_facade_equals_oracle(seed, n, m):
    """structure-plp-slam_amd/facade/PLPSLAM/match/projection.h on stand-in frame / landmark objects: landmarks_ after the
    call and the returned match count equal the array-form oracle's (match_frame_and_landmarks; match_current_and_last_frames
    for the monocular, forward and backward cases, including the nullptr left by the orientation check; the two line
    variants on key lines / 3D lines, with partially visible lines and the RGB-D stereo gate; bow_tree::match_frame_and_keyframe
    (facade/PLPSLAM/match/bow_tree.h) on DBoW2-shaped feature vectors and area::match_in_consistent_area (facade/PLPSLAM/match/area.h), each with and without the
    orientation check; fuse::detect_duplication, projection::match_by_Sim3_transform and match_keyframes_mutually through a Sim3)."""
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([_MATCH_EXE, str(seed), str(n), str(m), "sim3+mutual"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 25 and lines[0].startswith("match_frame_and_landmarks") and lines[4].startswith("bow_tree::match_frame_and_keyframe")
    assert lines[6].startswith("bow_tree::match_keyframes") and lines[7].startswith("fuse::replace_duplication:")
    assert lines[8].startswith("fuse::detect_duplication") and lines[9].startswith("projection::match_by_Sim3_transform")
    assert lines[10].startswith("projection::match_keyframes_mutually")
    assert lines[11].startswith("projection::match_frame_and_keyframe[") and lines[13].startswith("projection::match_frame_and_keyframe_line")
    assert lines[14].startswith("robust::match_for_triangulation") and lines[16].startswith("robust::brute_force_match") and lines[17].startswith("robust::brute_force_match")
    assert lines[18].startswith("fuse::replace_duplication_line")
    assert lines[19].startswith("area::match_in_consistent_area") and lines[21].startswith("match_frame_and_landmarks_line")
    if n >= 900:     # the scenes are built so that the matchers have work to do
        import re
        count = lambda ln: int(re.search(r"(\d+) (matches|fused)", ln).group(1))
        point_lines = lines[:13] + lines[14:18] + lines[19:21]
        assert all(count(ln) > 30 for ln in point_lines)
        assert all(count(ln) > 10 for ln in [lines[13], lines[18]] + lines[21:])


@pytest.mark.skipif(not os.path.exists(_EXE), reason="oracle/_ref/facade_orb_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("seed,K", [(0, 1000), (3, 2000)])
def test_cpp_stereo_facade_equals_oracle(tmp_path, seed, K):
    """Two feature::orb_extractor facades and the match::stereo facade wired like the stereo data::frame constructor
    (data/frame.cc:250-281): stereo_x_right / depths equal the oracle's; a pyramid vector that does not belong to an
    extractor is refused."""
    from test_gpu_stereo_lbdmatch import stereo_pair
    left, right = stereo_pair(seed)
    rows, cols = left.shape
    (tmp_path / "l.raw").write_bytes(np.ascontiguousarray(left).tobytes()); (tmp_path / "r.raw").write_bytes(np.ascontiguousarray(right).tobytes())
    fxb, tb = 435.2 * 0.11 * 10, 0.11 * 10
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = tmp_path / "s.bin"
    r = subprocess.run([_EXE, "stereo", str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(rows), str(cols), str(K), repr(fxb), repr(tb), str(out)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    b = out.read_bytes()
    n, nr = struct.unpack_from("<ii", b, 0)
    xr = np.frombuffer(b, np.float32, n, 8); dp = np.frombuffer(b, np.float32, n, 8 + 4 * n)
    ol, orr = O.OrbOracle(K), O.OrbOracle(K)
    kl, dl = ol.extract(left); kr, dr = orr.extract(right)
    want_x, want_d = O.stereo_compute(ol, orr, kl, kr, dl, dr, np.float32(fxb), np.float32(tb))
    assert n == len(kl) and nr == len(kr) and (want_x > 0).sum() > 100
    assert np.array_equal(xr, want_x) and np.array_equal(dp, want_d)


_LBDMATCH_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "facade_lbdmatch_check")


@pytest.mark.skipif(not os.path.exists(_LBDMATCH_EXE), reason="oracle/_ref/facade_lbdmatch_check not built (needs /root/reference at build time)")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_cpp_binary_descriptor_matcher_equals_oracle(seed):
    """cv::line_descriptor::BinaryDescriptorMatcher from the REFERENCE'S OWN descriptor_custom.hpp, linked with the shipped
    replacement translation unit (facade/src/binary_descriptor_matcher_plp.cpp) instead of the reference's
    binary_descriptor_matcher.cpp, driven like data/frame.cc:392-398: DMatch (queryIdx, trainIdx, distance) equal the oracle's."""
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([_LBDMATCH_EXE, str(seed)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout


_LINE_EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "facade_line_check")


@pytest.mark.skipif(not os.path.exists(_LINE_EXE), reason="oracle/facade_line_check not built (make -C oracle after the product library)")
@pytest.mark.parametrize("seed,shape", [(5, (480, 640)), (6, (376, 1241))])
def test_cpp_line_facade_equals_oracle(tmp_path, seed, shape):
    """structure-plp-slam_amd/facade/PLPSLAM/feature/line_extractor.h driven like data/frame.cc:1143-1167: key lines (all 17
    fields), LBD rows and line functions equal the oracle's."""
    img = synth.replay(seed, 1, *shape)[0]
    raw, out = tmp_path / "img.raw", tmp_path / "out.bin"
    raw.write_bytes(np.ascontiguousarray(img).tobytes())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([_LINE_EXE, str(raw), str(shape[0]), str(shape[1]), str(out)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    b = out.read_bytes()
    n, nl = struct.unpack_from("<ii", b, 0)
    sf = struct.unpack_from("<f", b, 8)[0]
    off = 12
    kl = np.frombuffer(b, O.KL_DTYPE, n, off); off += 68 * n
    lbd = np.frombuffer(b, np.uint8, 32 * n, off).reshape(n, 32); off += 32 * n
    fn = np.frombuffer(b, np.float64, 3 * n, off).reshape(n, 3)
    o = O.LineOracle(img)
    assert nl == 1 and sf == 2.0 and n == len(o.keylsd) and n > 20
    assert np.array_equal(kl, o.keylsd) and np.array_equal(lbd, o.lbd) and np.array_equal(fn, o.linefn)
