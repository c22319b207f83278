"""The HIP path (through the C ABI) against the committed answers of the REFERENCE BUILD -- tests/golden/ref_match.npz,
ref_line.npz, ref_stereo.npz, written by tools/make_golden_ref.py from oracle/_ref/libplpref2.so (= the reference's own
match/*.cc, data/common.cc, feature/line_extractor.cc, feature/line_descriptor/*.cpp compiled unmodified).  No oracle in
between: problem arrays in, the product library's answer compared with what the reference's code returned.
Bars: association arrays, counts, key-line fields, LBD bytes, line functions, stereo x_right / depth bit-exact;
KeyLine::angle <= 1 ulp (the reference calls atan2f, definition D2)."""
import numpy as np
import pytest

import oracle_lib as O
import match_cases as MC
from plp import plp
from test_oracle_golden_ref import G, golden_frames, check_lines, stereo_pairs

pytestmark = pytest.mark.gpu


def grid_of(g6):
    return plp.match_grid_c(float(g6[0]), float(g6[1]), float(g6[2]), float(g6[3]), int(g6[4]), int(g6[5]))


def hip(label, a):
    """one matcher problem (arguments in the order of the oracle / reference entry point) through the product library"""
    M = plp.matcher
    if label == "landmarks":
        g6, kps, desc, xr, occ, sf, valid, reproj, qxr, lvl, qd, hobs, margin, ratio = a
        o, n = M(ratio, True).match_host(plp.MODE_LANDMARKS, len(kps), len(lvl), dict(t_kps=kps, t_desc=desc, t_x_right=xr, t_occupied=occ, q_valid=valid, q_reproj=reproj,
                                         q_x_right=qxr, q_level=lvl, q_desc=qd, q_has_obs=hobs), margin=margin, scale_factors=sf, grid=grid_of(g6))
        return o[0], int(n[0])
    if label == "last_frame":
        g6, kps, desc, xr, occ, sf, valid, reproj, qxr, lvl, ang, qd, hobs, margin, direction, check = a
        o, n = M(0.9, check).match_host(plp.MODE_LAST_FRAME, len(kps), len(lvl), dict(t_kps=kps, t_desc=desc, t_x_right=xr, t_occupied=occ, q_valid=valid, q_reproj=reproj,
                                        q_x_right=qxr, q_level=lvl, q_angle=ang, q_desc=qd, q_has_obs=hobs), margin=margin, direction=direction, scale_factors=sf,
                                        grid=grid_of(g6))
        return o[0], int(n[0])
    if label == "frame_keyframe":
        g6, kps, desc, occ, sf, valid, reproj, pred, ang, qd, margin, thr, check = a
        o, n = M(0.9, check).match_host(plp.MODE_LAST_FRAME, len(kps), len(pred), dict(t_kps=kps, t_desc=desc, t_occupied=occ, q_valid=valid, q_reproj=reproj,
                                        q_level=pred.astype(np.int32), q_angle=ang, q_desc=qd, hamm_dist_thr=thr), margin=margin, direction=0, scale_factors=sf,
                                        grid=grid_of(g6))
        return o[0], int(n[0])
    if label == "sim3":
        g6, kps, desc, occ, sf, valid, reproj, pred, qd, margin = a
        o, n = M(0.9, False).match_host(plp.MODE_LAST_FRAME, len(kps), len(pred), dict(t_kps=kps, t_desc=desc, t_occupied=occ, q_valid=valid, q_reproj=reproj,
                                        q_level=pred.astype(np.int32), q_desc=qd, hamm_dist_thr=50, level_window=1, flags=plp.FLAG_UNSIGNED_LEVEL),
                                        margin=margin, scale_factors=sf, grid=grid_of(g6))
        return o[0], int(n[0])
    if label == "fuse":
        g6, kps, desc, xr, sf, inv_sigma, valid, rd, qxr, pred, qd, margin = a
        o = M().match_host(plp.MODE_FUSE, len(kps), len(pred), dict(t_kps=kps, t_desc=desc, t_x_right=xr, q_valid=valid, q_reproj_d=rd, q_x_right=qxr,
                           q_level=pred.astype(np.int32), q_desc=qd, inv_level_sigma_sq=inv_sigma), margin=margin, scale_factors=sf, grid=grid_of(g6))
        return (o[0],)
    if label == "detect_duplication":
        g6, kps, desc, sf, valid, rd, pred, qd, margin, thr, signed = a
        assert signed == 1 and thr == 50
        o = M().match_host(plp.MODE_FUSE, len(kps), len(pred), dict(t_kps=kps, t_desc=desc, q_valid=valid, q_reproj_d=rd, q_level=pred.astype(np.int32), q_desc=qd,
                           inv_level_sigma_sq=np.ones(len(sf), np.float32), flags=plp.FLAG_NO_CHI2 | plp.FLAG_SIGNED_LEVEL), margin=margin, scale_factors=sf,
                           grid=grid_of(g6))
        return (o[0],)
    if label == "brute_force":
        d1, a1, d2, a2, v2, ratio, check = a
        o, n = M(ratio, check).match_host(plp.MODE_BRUTE_FORCE, len(d1), len(d2), dict(t_desc=d1, t_angle=a1, q_desc=d2, q_angle=a2, q_valid=v2))
        return o[0], int(n[0])
    if label == "bow":
        qd, qa, qn, qv, td, ta, tn, tskip, ratio, check = a
        o, n = M(ratio, check).match_host(plp.MODE_BOW, len(td), len(qd), dict(t_desc=td, t_angle=ta, t_group=tn, t_occupied=tskip, q_desc=qd, q_angle=qa, q_group=qn,
                                          q_valid=qv))
        return o[0], int(n[0])
    if label == "triangulation":
        qd, qa, qn, qhas, qxr, qoct, b1, td, ta, tn, thas, txr, b2, sf, E, epi, check = a
        o, n = M(0.9, check).match_host(plp.MODE_TRIANGULATION, len(td), len(qd), dict(t_desc=td, t_angle=ta, t_group=tn, t_occupied=thas, t_x_right=txr, t_bearing=b2,
                                        q_desc=qd, q_angle=qa, q_group=qn, q_valid=(1 - qhas).astype(np.uint8), q_x_right=qxr, q_level=qoct, q_bearing=b1,
                                        epipolar=np.concatenate([E.ravel(), epi])), scale_factors=sf)
        per_q = np.full(len(qd), -1, np.int32)          # the library reports per target, the reference per query
        t_idx = np.nonzero(o[0] >= 0)[0]
        per_q[o[0][t_idx]] = t_idx
        return per_q, int(n[0])
    if label == "area":
        g6, k1, d1, k2, d2, prev, margin, ratio, check = a
        return M(ratio, check).match_in_consistent_area(k1, d1, k2, d2, prev, margin, grid_of(g6))
    if label == "landmarks_line":
        kl, lbd, kpo, occ, sf, valid, sp, ep, lvl, qd, hobs, margin, ratio = a
        o, n = M(ratio, False).match_host(plp.MODE_LANDMARKS_LINE, len(kl), len(lvl), dict(t_kl=kl, t_desc=lbd, t_kp_octave=kpo, t_occupied=occ, q_valid=valid, q_reproj=sp,
                                          q_reproj2=ep, q_level=lvl, q_desc=qd, q_has_obs=hobs), margin=margin, scale_factors=sf)
        return o[0], int(n[0])
    if label == "last_frame_line":
        kl, lbd, xrp, occ, sf, nlv, valid, sp, ep, xsp, xep, lvl, qd, hobs, margin, direction, rgbd = a
        o, n = M(0.9, True).match_host(plp.MODE_LAST_FRAME_LINE, len(kl), len(lvl), dict(t_kl=kl, t_desc=lbd, t_occupied=occ, t_x_right=np.ascontiguousarray(xrp[:, 0]),
                                       t_x_right2=np.ascontiguousarray(xrp[:, 1]), q_valid=valid, q_reproj=sp, q_reproj2=ep, q_x_right=xsp, q_x_right2=xep, q_level=lvl,
                                       q_desc=qd, q_has_obs=hobs, is_rgbd=rgbd, num_levels_lsd=nlv), margin=margin, direction=direction, scale_factors=sf)
        return o[0], int(n[0])
    if label == "frame_keyframe_line":
        kl, lbd, occ, sf, valid, sp, ep, pred, qd, margin, thr = a
        o, n = M(0.9, False).match_host(plp.MODE_LAST_FRAME_LINE, len(kl), len(pred), dict(t_kl=kl, t_desc=lbd, t_occupied=occ, q_valid=valid, q_reproj=sp, q_reproj2=ep,
                                        q_level=pred.astype(np.int32), q_desc=qd, hamm_dist_thr=thr, is_rgbd=0, num_levels_lsd=1), margin=margin, direction=0,
                                        scale_factors=sf)
        return o[0], int(n[0])
    if label == "fuse_line":
        kl, lbd, sf, inv_sigma, valid, spd, epd, pred, qd, margin = a
        o = M().match_host(plp.MODE_FUSE_LINE, len(kl), len(pred), dict(t_kl=kl, t_desc=lbd, q_valid=valid, q_reproj_d=spd, q_reproj2_d=epd, q_level=pred.astype(np.int32),
                           q_desc=qd, inv_level_sigma_sq=inv_sigma), margin=margin, scale_factors=sf)
        return (o[0],)
    if label == "lbd_1nn":
        return M().lbd_match_1nn(a[0], a[1])
    raise KeyError(label)


CASES = MC.load_golden_match()


@pytest.mark.parametrize("case", CASES, ids=[f"s{c[0]}-{c[1]}" for c in CASES])
def test_matcher_equals_the_reference_build(case):
    seed, label, fn, args, outs, extras = case
    got = hip(label, args)
    assert len(got) == len(outs)
    for g, w in zip(got, outs):
        if label == "lbd_1nn":
            ok = extras["defined"]
            assert np.array_equal(np.asarray(g)[ok], np.asarray(w)[ok])
        elif isinstance(w, np.ndarray):
            assert np.array_equal(np.asarray(g)[:len(w)], w)
        else:
            assert int(g) == w


def test_line_extractor_equals_the_reference_build():
    z = np.load(G / "ref_line.npz")
    tracker = plp.LineFeatureTracker()
    for name, img in golden_frames().items():
        kl, lbd, fn = tracker.extract_LSD_LBD(img)
        check_lines(name, kl, lbd, fn, z)


def test_stereo_equals_the_reference_build():
    z = np.load(G / "ref_stereo.npz")
    for seed, K, left, right in stereo_pairs():
        el, er = plp.orb_extractor(K), plp.orb_extractor(K)
        kl, dl = el.extract(left); kr, dr = er.extract(right)
        for tag in ("wide", "narrow"):
            fxb, tb = (float(v) for v in z[f"seed{seed}_K{K}_{tag}__params"])
            xr, dp = el.stereo_compute(er, kl, kr, dl, dr, fxb, tb)
            assert np.array_equal(xr, z[f"seed{seed}_K{K}_{tag}__x_right"]) and np.array_equal(dp, z[f"seed{seed}_K{K}_{tag}__depth"])
