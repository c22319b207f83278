"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os, subprocess, pathlib
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
ORACLE_DIR = ROOT / "oracle"

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True)


def lib():
    global _lib
    if _lib is None:
        so = ORACLE_DIR / "liboracle.so"
        if not so.exists():
            build()
        L = C.CDLL(str(so))
        L.oracle_orb_create.restype = C.c_void_p
        L.oracle_orb_create.argtypes = [C.c_uint, C.c_float, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_int]
        L.oracle_orb_destroy.argtypes = [C.c_void_p]
        L.oracle_orb_extract.restype = C.c_int
        L.oracle_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_long,
                                         C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_orb_tables.argtypes = [C.c_void_p] * 7
        L.oracle_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_orb_level_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_orb_level_blurred.restype = C.c_int
        L.oracle_orb_level_blurred.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_orb_num_candidates.argtypes = [C.c_void_p, C.c_int]
        L.oracle_orb_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_orb_num_level_keypts.argtypes = [C.c_void_p, C.c_int]
        L.oracle_orb_level_keypts.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_orb_distribute.restype = C.c_int
        L.oracle_orb_distribute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_uint, C.c_void_p]
        L.oracle_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.oracle_gaussian_blur_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.oracle_gaussian_taps_q8.argtypes = [C.c_int, C.c_double, C.c_void_p]
        L.oracle_fast9_16.restype = C.c_int
        L.oracle_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oracle_fast_atan2.restype = C.c_float
        L.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.oracle_trig_cos.restype = C.c_float
        L.oracle_trig_cos.argtypes = [C.c_float]
        L.oracle_trig_sin.restype = C.c_float
        L.oracle_trig_sin.argtypes = [C.c_float]
        L.oracle_scale_tables.argtypes = [C.c_uint, C.c_float] + [C.c_void_p] * 4
        L.oracle_orb_time_frames.restype = C.c_double
        L.oracle_orb_time_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbOracle:
    """Python face of oracle::OrbOracle (restates feature::orb_extractor)."""

    def __init__(self, max_num_keypts=2000, scale_factor=1.2, num_levels=8, ini_fast_thr=20, min_fast_thr=7,
                 mask_rects=()):
        r = np.ascontiguousarray(np.asarray(mask_rects, np.float32).reshape(-1, 4))
        self.num_levels = num_levels
        self.max_num_keypts = max_num_keypts
        self.h = lib().oracle_orb_create(max_num_keypts, scale_factor, num_levels, ini_fast_thr, min_fast_thr,
                                         _p(r) if len(r) else None, len(r))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_orb_destroy(self.h)
            self.h = None

    def extract(self, img, mask=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = max(self.max_num_keypts * 2, 16)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        n = lib().oracle_orb_extract(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0],
                                     _p(mask) if mask is not None else None,
                                     mask.strides[0] if mask is not None else 0, _p(kps), _p(desc), cap)
        assert n >= 0
        return kps[:n].copy(), desc[:n].copy()

    def tables(self):
        n = self.num_levels
        f = [np.zeros(n, np.float32) for _ in range(4)]
        q = np.zeros(n, np.uint32)
        u = np.zeros(16, np.int32)
        lib().oracle_orb_tables(self.h, *[_p(a) for a in f], _p(q), _p(u))
        return dict(scale_factors=f[0], inv_scale_factors=f[1], level_sigma_sq=f[2], inv_level_sigma_sq=f[3],
                    quota=q, u_max=u)

    def level_image(self, level):
        r, c = C.c_int(), C.c_int()
        lib().oracle_orb_level_size(self.h, level, C.byref(r), C.byref(c))
        a = np.zeros((r.value, c.value), np.uint8)
        lib().oracle_orb_level_image(self.h, level, _p(a))
        return a

    def level_blurred(self, level):
        a = np.zeros_like(self.level_image(level))
        ok = lib().oracle_orb_level_blurred(self.h, level, _p(a))
        return a if ok else None

    def candidates(self, level):
        n = lib().oracle_orb_num_candidates(self.h, level)
        a = np.zeros(n, KP_DTYPE)
        lib().oracle_orb_candidates(self.h, level, _p(a))
        return a

    def level_keypts(self, level):
        n = lib().oracle_orb_num_level_keypts(self.h, level)
        a = np.zeros(n, KP_DTYPE)
        lib().oracle_orb_level_keypts(self.h, level, _p(a))
        return a

    def distribute(self, cands, min_x, max_x, min_y, max_y, num_keypts):
        cands = np.ascontiguousarray(cands, KP_DTYPE)
        out = np.zeros(max(len(cands), 1), KP_DTYPE)
        n = lib().oracle_orb_distribute(self.h, _p(cands), len(cands), min_x, max_x, min_y, max_y, num_keypts, _p(out))
        return out[:n].copy()


def resize_linear_u8(src, dh, dw):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oracle_resize_linear_u8(_p(src), src.shape[0], src.shape[1], _p(dst), dh, dw)
    return dst


def gaussian_blur_u8(src, ksize, sigma):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().oracle_gaussian_blur_u8(_p(src), src.shape[0], src.shape[1], ksize, float(sigma), _p(dst))
    return dst


def gaussian_taps_q8(n, sigma):
    out = np.zeros(n, np.int32)
    lib().oracle_gaussian_taps_q8(n, float(sigma), _p(out))
    return out


def fast9_16(roi, threshold):
    """cv::FAST(roi, thr, nms=True) restated; roi may be a strided 2-D view."""
    assert roi.dtype == np.uint8 and roi.strides[1] == 1
    cap = roi.shape[0] * roi.shape[1]
    out = np.zeros((cap, 3), np.int32)
    n = lib().oracle_fast9_16(C.c_void_p(roi.ctypes.data), roi.strides[0], roi.shape[1], roi.shape[0], threshold,
                              _p(out), cap)
    return out[:n].copy()
