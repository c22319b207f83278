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


_backend = None          # set by reference(): the oracle_* names resolve to the ref_* entry points of oracle/_ref/libplpref2.so
_ref2 = None


def ref2_path():
    return ORACLE_DIR / "_ref" / "libplpref2.so"


class _RefProxy:
    """Resolves oracle_NAME to ref_NAME in libplpref2.so (the reference's own matcher / line sources compiled unmodified,
    oracle/ref_driver2.cpp) with the argtypes / restype of the oracle's entry point: every wrapper below then runs the
    REFERENCE on the same arrays."""

    def __init__(self, ref, orc):
        self._ref, self._orc = ref, orc

    def __getattr__(self, name):
        if not name.startswith("oracle_"):
            raise AttributeError(name)
        fn = getattr(self._ref, "ref_" + name[len("oracle_"):])
        src = getattr(self._orc, name)
        if src.argtypes is not None:
            fn.argtypes = src.argtypes
        fn.restype = src.restype
        return fn


class reference:
    """context manager: inside it the matcher / line wrappers of this module call the reference build instead of the oracle"""

    def __enter__(self):
        global _backend, _ref2
        if _ref2 is None:
            _ref2 = C.CDLL(str(ref2_path()))
        _backend = _RefProxy(_ref2, _load())
        return _ref2

    def __exit__(self, *exc):
        global _backend
        _backend = None
        return False


class _VariantProxy:
    """another build of the SAME oracle sources (e.g. liboracle_d2.so): same entry points, the argtypes / restype of the default build"""

    def __init__(self, other, orc):
        self._other, self._orc = other, orc

    def __getattr__(self, name):
        if not name.startswith("oracle_"):
            raise AttributeError(name)
        fn, src = getattr(self._other, name), getattr(self._orc, name)
        if src.argtypes is not None:
            fn.argtypes = src.argtypes
        fn.restype = src.restype
        return fn


class variant:
    """context manager: inside it the wrappers of this module call the oracle build `so_name` (a file beside liboracle.so)"""

    def __init__(self, so_name):
        self.path = ORACLE_DIR / so_name

    def __enter__(self):
        global _backend
        orc = _load()
        if not self.path.exists():
            build()
        _backend = _VariantProxy(C.CDLL(str(self.path)), orc)
        return self

    def __exit__(self, *exc):
        global _backend
        _backend = None
        return False


def lib():
    return _backend if _backend is not None else _load()


def _load():
    global _lib
    if _lib is None:
        # PLP_ORACLE_SO: another build of the same sources in the default's place for a whole run (tools/oracle_sanitized_tests.sh: the AddressSanitizer / UBSan build)
        so = ORACLE_DIR / os.environ.get("PLP_ORACLE_SO", "liboracle.so")
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
        L.oracle_f_cos_sin.restype = None
        L.oracle_f_cos_sin.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        L.oracle_trig_cos.restype = C.c_float
        L.oracle_trig_cos.argtypes = [C.c_float]
        L.oracle_trig_sin.restype = C.c_float
        L.oracle_trig_sin.argtypes = [C.c_float]
        L.oracle_scale_tables.argtypes = [C.c_uint, C.c_float] + [C.c_void_p] * 4
        L.oracle_orb_time_frames.restype = C.c_double
        L.oracle_orb_time_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p]
        L.oracle_hamming32.restype = C.c_uint
        L.oracle_hamming32.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_hamming64.restype = C.c_uint
        L.oracle_hamming64.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_angle_checker.restype = C.c_int
        L.oracle_angle_checker.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_get_cell_indices.restype = C.c_int
        L.oracle_get_cell_indices.argtypes = [C.c_float] * 2 + [C.c_double] * 2 + [C.c_int] * 2 + [C.c_float] * 2 + [C.c_void_p] * 2
        L.oracle_keypoints_in_cell.restype = C.c_int
        L.oracle_keypoints_in_cell.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.oracle_match_frame_and_landmarks.restype = C.c_uint
        L.oracle_match_frame_and_landmarks.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.oracle_match_current_and_last.restype = C.c_uint
        L.oracle_match_current_and_last.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.oracle_brute_force_match.restype = C.c_uint
        L.oracle_brute_force_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.oracle_keylines_in_cell.restype = C.c_int
        L.oracle_keylines_in_cell.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 5 + [C.c_int, C.c_int, C.c_void_p]
        L.oracle_match_frame_and_landmarks_line.restype = C.c_uint
        L.oracle_match_frame_and_landmarks_line.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.oracle_match_current_and_last_line.restype = C.c_uint
        L.oracle_match_current_and_last_line.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
        L.oracle_stereo_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_lbd_match_1nn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_match_bow.restype = C.c_uint
        L.oracle_match_bow.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.oracle_fuse_search.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_void_p]
        L.oracle_match_area.restype = C.c_uint
        L.oracle_match_area.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.oracle_front_time_frames.restype = C.c_double
        L.oracle_front_time_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_float, C.c_void_p]
        L.oracle_line_extract.restype = C.c_void_p
        L.oracle_line_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int]
        L.oracle_line_free.argtypes = [C.c_void_p]
        L.oracle_line_count.argtypes = [C.c_void_p, C.c_int]
        L.oracle_line_get.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.oracle_line_raw.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_line_scaled_size.argtypes = [C.c_void_p] * 3
        L.oracle_line_scaled.argtypes = [C.c_void_p] * 2
        L.oracle_line_order.argtypes = [C.c_void_p] * 2
        L.oracle_line_sobel.argtypes = [C.c_void_p] * 3
        L.oracle_resize_linear_exact_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.oracle_lbd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
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
        cap = 2 * self.max_num_keypts + 64      # (a level yields at least the four nodes of its first split, whatever its quota: K = 2 returns 32 key points)
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


# ---- matchers (oracle/match_oracle.cpp)
def hamming32(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().oracle_hamming32(_p(a), _p(b))


def hamming64(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().oracle_hamming64(_p(a), _p(b))


def angle_checker(deltas, hist_len=30, n_bins_thr=3, valid=False):
    d = np.ascontiguousarray(deltas, np.float32)
    out = np.zeros(max(len(d), 1), np.int32)
    n = lib().oracle_angle_checker(_p(d), len(d), hist_len, n_bins_thr, int(valid), _p(out))
    return out[:n].copy()


def grid6(g):
    return np.array([g.min_x, g.min_y, g.inv_cell_width, g.inv_cell_height, g.cols, g.rows], np.float64)


def keypoints_in_cell(g6, kps, ref_x, ref_y, margin, min_level=-1, max_level=-1):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros(max(len(kps), 1), np.uint32)
    n = lib().oracle_keypoints_in_cell(_p(g6), _p(kps), len(kps), ref_x, ref_y, margin, min_level, max_level, _p(out))
    return out[:n].copy()


def _c(a, dt):
    return np.ascontiguousarray(a, dt)


def match_frame_and_landmarks(g6, kps, desc, x_right, occupied, scale_factors, lm_valid, lm_reproj, lm_x_right, lm_level, lm_desc,
                              lm_has_obs, margin, lowe_ratio):
    n, m = len(kps), len(lm_level)
    out = np.zeros(max(n, 1), np.int32)
    a = [_c(g6, np.float64), _c(kps, KP_DTYPE), _c(desc, np.uint8), _c(x_right, np.float32), _c(occupied, np.uint8)]
    b = [_c(scale_factors, np.float32), _c(lm_valid, np.uint8), _c(lm_reproj, np.float32), _c(lm_x_right, np.float32),
         _c(lm_level, np.int32), _c(lm_desc, np.uint8), _c(lm_has_obs, np.uint8)]
    num = lib().oracle_match_frame_and_landmarks(*[_p(v) for v in a], n, *[_p(v) for v in b], m, margin, lowe_ratio, _p(out))
    return out[:n].copy(), num


def match_current_and_last(g6, kps, desc, x_right, occupied, scale_factors, valid, reproj, lx_right, loctave, langle, ldesc, l_has_obs,
                           margin, direction, check_orientation):
    n, m = len(kps), len(loctave)
    out = np.zeros(max(n, 1), np.int32)
    sf = _c(scale_factors, np.float32)
    a = [_c(g6, np.float64), _c(kps, KP_DTYPE), _c(desc, np.uint8), _c(x_right, np.float32), _c(occupied, np.uint8)]
    b = [_c(valid, np.uint8), _c(reproj, np.float32), _c(lx_right, np.float32), _c(loctave, np.int32), _c(langle, np.float32),
         _c(ldesc, np.uint8), _c(l_has_obs, np.uint8)]
    num = lib().oracle_match_current_and_last(*[_p(v) for v in a], n, _p(sf), len(sf), *[_p(v) for v in b], m, margin, direction,
                                              int(check_orientation), _p(out))
    return out[:n].copy(), num


def brute_force_match(desc1, angle1, desc2, angle2, valid2, lowe_ratio, check_orientation):
    n1, n2 = len(desc1), len(desc2)
    out = np.zeros(max(n1, 1), np.int32)
    v = [_c(desc1, np.uint8), _c(angle1, np.float32)]
    w = [_c(desc2, np.uint8), _c(angle2, np.float32), _c(valid2, np.uint8)]
    num = lib().oracle_brute_force_match(_p(v[0]), _p(v[1]), n1, _p(w[0]), _p(w[1]), _p(w[2]), n2, lowe_ratio, int(check_orientation), _p(out))
    return out[:n1].copy(), num


def fast_atan2(y, x):
    """cv::fastAtan2 (degrees, f32) as the oracle restates it (oracle/cv_restated.hpp fast_atan2f_deg)"""
    return float(lib().oracle_fast_atan2(float(y), float(x)))


def f_cos_sin(a):
    """D2: (float)cos((double)a), (float)sin((double)a) with this machine's libm (lsd_restated.hpp f_cos / f_sin)"""
    a = np.ascontiguousarray(a, np.float32)
    c = np.zeros(a.shape, np.float32); s = np.zeros(a.shape, np.float32)
    lib().oracle_f_cos_sin(_p(a), a.size, _p(c), _p(s))
    return c, s


# ---- line front-end (oracle/line_oracle.cpp)
KL_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"), ("response", "<f4"),
                     ("size", "<f4"), ("startPointX", "<f4"), ("startPointY", "<f4"), ("endPointX", "<f4"), ("endPointY", "<f4"),
                     ("sPointInOctaveX", "<f4"), ("sPointInOctaveY", "<f4"), ("ePointInOctaveX", "<f4"), ("ePointInOctaveY", "<f4"),
                     ("lineLength", "<f4"), ("numOfPixels", "<i4")])


class LineOracle:
    """one extract_LSD_LBD run of the oracle with its stage outputs"""

    def __init__(self, img, stable_order=False):
        """stable_order=False (default): seeds visited in the order std::sort leaves, as OpenCV's lsd.cpp does it (a reference built with this C++
        library); True: bin descending then row-major, the library's PLP_SEED_ORDER_STABLE mode (definition D1 of rounds 1-3)"""
        img = np.ascontiguousarray(img, np.uint8)
        self.shape = img.shape
        L = lib()
        h = L.oracle_line_extract(_p(img), img.shape[0], img.shape[1], img.strides[0], int(stable_order))
        try:
            n = L.oracle_line_count(h, 0); na = L.oracle_line_count(h, 1); nr = L.oracle_line_count(h, 2)
            self.keylsd = np.zeros(n, KL_DTYPE); self.lbd = np.zeros((n, 32), np.uint8); self.linefn = np.zeros((n, 3), np.float64)
            L.oracle_line_get(h, 0, _p(self.keylsd), _p(self.lbd), _p(self.linefn), None)
            self.all_kl = np.zeros(na, KL_DTYPE); self.all_lbd = np.zeros((na, 32), np.uint8); self.all_desc_f = np.zeros((na, 72), np.float32)
            L.oracle_line_get(h, 1, _p(self.all_kl), _p(self.all_lbd), None, _p(self.all_desc_f))
            self.raw = np.zeros((nr, 4), np.float32)
            L.oracle_line_raw(h, _p(self.raw))
            r, c = C.c_int(), C.c_int()
            L.oracle_line_scaled_size(h, C.byref(r), C.byref(c))
            self.scaled = np.zeros((r.value, c.value), np.uint8)
            L.oracle_line_scaled(h, _p(self.scaled))
            self.order = np.zeros((r.value - 1) * (c.value - 1), np.int32)
            L.oracle_line_order(h, _p(self.order))
            self.dx = np.zeros(img.shape, np.int16); self.dy = np.zeros(img.shape, np.int16)
            if na:
                L.oracle_line_sobel(h, _p(self.dx), _p(self.dy))
        finally:
            L.oracle_line_free(h)


def resize_linear_exact_u8(src, fx, fy):
    src = np.ascontiguousarray(src, np.uint8)
    dh, dw = int(np.rint(src.shape[0] * fy)), int(np.rint(src.shape[1] * fx))
    dst = np.zeros((dh, dw), np.uint8)
    lib().oracle_resize_linear_exact_u8(_p(src), src.shape[0], src.shape[1], fx, fy, _p(dst))
    return dst


def match_frame_and_landmarks_line(kl, lbd, kp_octave, occupied, sf_lsd, lm_valid, lm_sp, lm_ep, lm_level, lm_desc, lm_has_obs, margin, ratio):
    n, m = len(kl), len(lm_level)
    out = np.zeros(max(n, 1), np.int32)
    a = [_c(kl, KL_DTYPE), _c(lbd, np.uint8), _c(kp_octave, np.int32), _c(occupied, np.uint8)]
    b = [_c(sf_lsd, np.float32), _c(lm_valid, np.uint8), _c(lm_sp, np.float32), _c(lm_ep, np.float32), _c(lm_level, np.int32),
         _c(lm_desc, np.uint8), _c(lm_has_obs, np.uint8)]
    num = lib().oracle_match_frame_and_landmarks_line(*[_p(v) for v in a], n, *[_p(v) for v in b], m, margin, ratio, _p(out))
    return out[:n].copy(), num


def match_current_and_last_line(kl, lbd, xr_pair, occupied, sf_lsd, num_levels_lsd, valid, sp, ep, lxr_sp, lxr_ep, loctave, ldesc, l_has_obs,
                                margin, direction, is_rgbd):
    n, m = len(kl), len(loctave)
    out = np.zeros(max(n, 1), np.int32)
    a = [_c(kl, KL_DTYPE), _c(lbd, np.uint8), _c(xr_pair, np.float32), _c(occupied, np.uint8)]
    sf = _c(sf_lsd, np.float32)
    b = [_c(valid, np.uint8), _c(sp, np.float32), _c(ep, np.float32), _c(lxr_sp, np.float32), _c(lxr_ep, np.float32), _c(loctave, np.int32),
         _c(ldesc, np.uint8), _c(l_has_obs, np.uint8)]
    num = lib().oracle_match_current_and_last_line(*[_p(v) for v in a], n, _p(sf), num_levels_lsd, *[_p(v) for v in b], m, margin, direction,
                                                   int(is_rgbd), _p(out))
    return out[:n].copy(), num


def stereo_compute(orb_left, orb_right, kl, kr, dl, dr, fxb, tb):
    """orb_left / orb_right: OrbOracle objects that just extracted the left / right image"""
    kl = _c(kl, KP_DTYPE); kr = _c(kr, KP_DTYPE); dl = _c(dl, np.uint8); dr = _c(dr, np.uint8)
    xr = np.zeros(max(len(kl), 1), np.float32); dp = np.zeros(max(len(kl), 1), np.float32)
    lib().oracle_stereo_compute(orb_left.h, orb_right.h, _p(kl), len(kl), _p(kr), len(kr), _p(dl), _p(dr), fxb, tb, _p(xr), _p(dp))
    return xr[:len(kl)].copy(), dp[:len(kl)].copy()


def lbd_match_1nn(q, t):
    q = _c(q, np.uint8).reshape(-1, 32); t = _c(t, np.uint8).reshape(-1, 32)
    idx = np.zeros(max(len(q), 1), np.int32); dist = np.zeros(max(len(q), 1), np.int32)
    lib().oracle_lbd_match_1nn(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx[:len(q)].copy(), dist[:len(q)].copy()


def match_bow(q_desc, q_angle, q_node, q_valid, t_desc, t_angle, t_node, t_skip, ratio, check_orientation):
    m, n = len(q_desc), len(t_desc)
    out = np.zeros(max(n, 1), np.int32)
    a = [_c(q_desc, np.uint8), _c(q_angle, np.float32), _c(q_node, np.int32), _c(q_valid, np.uint8)]
    b = [_c(t_desc, np.uint8), _c(t_angle, np.float32), _c(t_node, np.int32), _c(t_skip, np.uint8)]
    num = lib().oracle_match_bow(*[_p(v) for v in a], m, *[_p(v) for v in b], n, ratio, int(check_orientation), _p(out))
    return out[:n].copy(), num


def fuse_search(g6, kps, desc, x_right, sf, inv_sigma, lm_valid, reproj_d, lm_x_right, pred_level, lm_desc, margin):
    n, m = len(kps), len(pred_level)
    out = np.zeros(max(m, 1), np.int32)
    a = [_c(g6, np.float64), _c(kps, KP_DTYPE), _c(desc, np.uint8), _c(x_right, np.float32)]
    b = [_c(sf, np.float32), _c(inv_sigma, np.float32), _c(lm_valid, np.uint8), _c(reproj_d, np.float64), _c(lm_x_right, np.float32),
         _c(pred_level, np.uint32), _c(lm_desc, np.uint8)]
    lib().oracle_fuse_search(*[_p(v) for v in a], n, *[_p(v) for v in b], m, margin, _p(out))
    return out[:m].copy()


def match_area(g6, kps1, desc1, kps2, desc2, prev_pts, margin, ratio, check_orientation):
    n1, n2 = len(kps1), len(kps2)
    out = np.zeros(max(n1, 1), np.int32)
    pp = _c(prev_pts, np.float32).copy()
    v = [_c(g6, np.float64), _c(kps1, KP_DTYPE), _c(desc1, np.uint8)]
    w = [_c(kps2, KP_DTYPE), _c(desc2, np.uint8)]
    num = lib().oracle_match_area(_p(v[0]), _p(v[1]), _p(v[2]), n1, _p(w[0]), _p(w[1]), n2, _p(pp), int(margin), ratio, int(check_orientation), _p(out))
    return out[:n1].copy(), pp, num


def _call(name, args, restype=None):
    """generic ctypes call: numpy arrays -> pointers, python ints/floats by argtype inference"""
    fn = getattr(lib(), name)
    fn.restype = restype
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(C.c_void_p(a.ctypes.data))
        elif isinstance(a, float):
            conv.append(C.c_float(a))
        elif isinstance(a, tuple):      # ("u", v): unsigned, ("z", v): size_t
            conv.append(C.c_size_t(a[1]) if a[0] == "z" else C.c_uint(a[1]))
        else:
            conv.append(C.c_int(int(a)))
    return fn(*conv)


def match_frame_and_keyframe(g6, kps, desc, occupied, sf, valid, reproj, pred, langle, ldesc, margin, thr, check):
    n, m = len(kps), len(pred)
    out = np.zeros(max(n, 1), np.int32)
    keep = [_c(g6, np.float64), _c(kps, KP_DTYPE), _c(desc, np.uint8), _c(occupied, np.uint8), _c(sf, np.float32), _c(valid, np.uint8),
            _c(reproj, np.float32), _c(pred, np.uint32), _c(langle, np.float32), _c(ldesc, np.uint8)]
    num = _call("oracle_match_frame_and_keyframe", keep[:4] + [n] + keep[4:] + [m, float(margin), ("u", thr), int(check), out], C.c_uint)
    return out[:n].copy(), num


def match_frame_and_keyframe_line(kl, lbd, occupied, sf_lsd, valid, sp, ep, pred, ldesc, margin, thr):
    n, m = len(kl), len(pred)
    out = np.zeros(max(n, 1), np.int32)
    keep = [_c(kl, KL_DTYPE), _c(lbd, np.uint8), _c(occupied, np.uint8), _c(sf_lsd, np.float32), _c(valid, np.uint8), _c(sp, np.float32),
            _c(ep, np.float32), _c(pred, np.uint32), _c(ldesc, np.uint8)]
    num = _call("oracle_match_frame_and_keyframe_line", keep[:3] + [n] + keep[3:] + [m, float(margin), ("u", thr), out], C.c_uint)
    return out[:n].copy(), num


def match_by_sim3(g6, kps, desc, occupied, sf, valid, reproj, pred, ldesc, margin):
    n, m = len(kps), len(pred)
    out = np.zeros(max(n, 1), np.int32)
    keep = [_c(g6, np.float64), _c(kps, KP_DTYPE), _c(desc, np.uint8), _c(occupied, np.uint8), _c(sf, np.float32), _c(valid, np.uint8),
            _c(reproj, np.float32), _c(pred, np.uint32), _c(ldesc, np.uint8)]
    num = _call("oracle_match_by_sim3", keep[:4] + [n] + keep[4:] + [m, float(margin), out], C.c_uint)
    return out[:n].copy(), num


def project_best(g6, kps, desc, sf, valid, reproj_d, pred, ldesc, margin, thr, signed_level):
    n, m = len(kps), len(pred)
    out = np.zeros(max(m, 1), np.int32)
    keep = [_c(g6, np.float64), _c(kps, KP_DTYPE), _c(desc, np.uint8), _c(sf, np.float32), _c(valid, np.uint8), _c(reproj_d, np.float64),
            _c(pred, np.uint32), _c(ldesc, np.uint8)]
    _call("oracle_project_best", keep[:3] + [n] + keep[3:] + [m, float(margin), ("u", thr), int(signed_level), out])
    return out[:m].copy()


def cross_check(idx2_of_1, idx1_of_2):
    a, b = _c(idx2_of_1, np.int32), _c(idx1_of_2, np.int32)
    out = np.zeros(max(len(a), 1), np.int32)
    num = _call("oracle_cross_check", [a, len(a), b, out], C.c_uint)
    return out[:len(a)].copy(), num


def fuse_search_line(kl, lbd, sf_lsd, inv_sigma_lsd, valid, sp_d, ep_d, pred, ldesc, margin):
    n, m = len(kl), len(pred)
    out = np.zeros(max(m, 1), np.int32)
    keep = [_c(kl, KL_DTYPE), _c(lbd, np.uint8), _c(sf_lsd, np.float32), _c(inv_sigma_lsd, np.float32), _c(valid, np.uint8), _c(sp_d, np.float64),
            _c(ep_d, np.float64), _c(pred, np.uint32), _c(ldesc, np.uint8)]
    _call("oracle_fuse_search_line", keep[:2] + [n] + keep[2:] + [m, float(margin), out])
    return out[:m].copy()


def match_for_triangulation(q_desc, q_angle, q_node, q_has_lm, q_x_right, q_octave, q_bearing, t_desc, t_angle, t_node, t_has_lm, t_x_right,
                            t_bearing, sf, E_12, epipole, check):
    m, n = len(q_desc), len(t_desc)
    out = np.zeros(max(m, 1), np.int32)
    a = [_c(q_desc, np.uint8), _c(q_angle, np.float32), _c(q_node, np.int32), _c(q_has_lm, np.uint8), _c(q_x_right, np.float32),
         _c(q_octave, np.int32), _c(q_bearing, np.float64)]
    b = [_c(t_desc, np.uint8), _c(t_angle, np.float32), _c(t_node, np.int32), _c(t_has_lm, np.uint8), _c(t_x_right, np.float32),
         _c(t_bearing, np.float64)]
    c = [_c(sf, np.float32), _c(E_12, np.float64), _c(epipole, np.float64)]
    num = _call("oracle_match_for_triangulation", a + [m] + b + [n] + c + [int(check), out], C.c_uint)
    return out[:m].copy(), num


# ---- post-extract step (oracle/post_oracle.cpp)
def post_extract(cam10, kps, depth=None, keylines=None, kl_depths=None, kl_x_right=None):
    cam = _c(cam10, np.float64)
    k = _c(kps, KP_DTYPE); n = len(k)
    und = np.zeros(max(n, 1), KP_DTYPE); bear = np.zeros((max(n, 1), 3), np.float64)
    _call("oracle_undistort_keypoints", [cam, k, n, und])
    _call("oracle_bearings", [cam, und, n, bear])
    out = dict(undist_keypts=und[:n].copy(), bearings=bear[:n].copy())
    if depth is not None:
        d = _c(depth, np.float32)
        xr = np.zeros(max(n, 1), np.float32); dep = np.zeros(max(n, 1), np.float32)
        _call("oracle_stereo_from_depth", [cam, d, d.shape[0], d.shape[1], k, und, n, xr, dep])
        out.update(stereo_x_right=xr[:n].copy(), depths=dep[:n].copy())
        if keylines is not None:
            kl = _c(keylines, KL_DTYPE)
            kd = _c(kl_depths, np.float32).copy(); kx = _c(kl_x_right, np.float32).copy()
            _call("oracle_stereo_from_depth_lines", [cam, d, d.shape[0], d.shape[1], kl, len(kl), kd, kx])
            out.update(kl_depths=kd, kl_x_right=kx)
    return out


def landmark_descriptor(descs):
    d = _c(descs, np.uint8).reshape(-1, 32)
    return _call("oracle_landmark_descriptor", [d, len(d)], C.c_int)


# EuRoC MAV stereo calibration as quoted by the reference's example/euroc/EuRoC_stereo.yaml
EUROC = {
    "camera": dict(fx=435.2046959714599, fy=435.2046959714599, cx=367.4517211914062, cy=252.2008514404297, cols=752, rows=480),
    "StereoRectifier.K_left": [458.654, 0.0, 367.215, 0.0, 457.296, 248.375, 0.0, 0.0, 1.0],
    "StereoRectifier.D_left": [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0],
    "StereoRectifier.R_left": [0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127, 0.9999741760894847,
                               0.007055629199258132, -0.008089410156878961, -0.007044357138835809, 0.9999424675829176],
    "StereoRectifier.K_right": [457.587, 0.0, 379.999, 0.0, 456.134, 255.238, 0.0, 0.0, 1],
    "StereoRectifier.D_right": [-0.28368365, 0.07451284, -0.00010473, -3.555907e-05, 0.0],
    "StereoRectifier.R_right": [0.9999633526194376, -0.003625811871560086, 0.007755443660172947, 0.003680398547259526, 0.9999684752771629,
                                -0.007035845251224894, -0.007729688520722713, 0.007064130529506649, 0.999945173484644],
}


def rectify_map(K, D, R, cam, rows, cols):
    """cv::initUndistortRectifyMap(K, D, R, K_rect(float), (cols, rows), CV_32F) -> map_x, map_y"""
    K = _c(K, np.float64); D = _c(D, np.float64); R = _c(R, np.float64)
    Kr = np.array([cam["fx"], 0, cam["cx"], 0, cam["fy"], cam["cy"], 0, 0, 1], np.float64)
    mx = np.zeros((rows, cols), np.float32); my = np.zeros((rows, cols), np.float32)
    rc = _call("oracle_init_undistort_rectify_map", [K, D if D.size else np.zeros(1), int(D.size), R, Kr, rows, cols, mx, my], C.c_int)
    if rc != 0:
        raise ValueError("K_rect * R is singular")
    return mx, my


def remap_linear(src, map_x, map_y):
    src = _c(src, np.uint8); map_x = _c(map_x, np.float32); map_y = _c(map_y, np.float32)
    dst = np.zeros(map_x.shape, np.uint8)
    _call("oracle_remap_linear", [src, src.shape[0], src.shape[1], ("z", src.shape[1]), map_x, map_y, map_x.shape[0], map_x.shape[1], dst])
    return dst


def random_vocab(rng, k, L, p_leaf=0.08, p_dup=0.1, p_stop=0.05, k_jitter=0):
    """A DBoW2-shaped tree in m_nodes order (the k children of a node are created together, then each is expanded):
    returns parents, is_leaf, descs, weights.  Some leaves sit above level L, some siblings share a descriptor (ties),
    some words have weight 0 (stopped)."""
    parents, level = [-1], [0]
    todo = [0]
    while todo:
        p = todo.pop()
        if level[p] == L or (p != 0 and rng.uniform() < p_leaf):
            continue
        kk = max(1, k + int(rng.integers(-k_jitter, k_jitter + 1)))
        first = len(parents)
        parents += [p] * kk; level += [level[p] + 1] * kk
        todo += list(range(first + kk - 1, first - 1, -1))
    n = len(parents)
    parents = np.array(parents)
    is_leaf = np.bincount(parents[1:], minlength=n) == 0
    descs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for i in range(2, n):
        if parents[i] == parents[i - 1] and rng.uniform() < p_dup:
            descs[i] = descs[i - 1]
    weights = np.where(is_leaf, rng.uniform(0.1, 12.0, n), 0.0)
    weights[is_leaf & (rng.uniform(size=n) < p_stop)] = 0.0
    return parents, is_leaf, descs, weights


def bow_transform(child_offset, children, node_desc, node_weight, node_word, L, desc, levelsup, accumulate, norm):
    """DBoW2 transform(features, BowVector, FeatureVector, levelsup) on the flat tree -> (word_id, node_id, bow_word, bow_value, fv_node, fv_feat)"""
    desc = _c(desc, np.uint8).reshape(-1, 32)
    n = len(desc)
    word = np.zeros(max(n, 1), np.uint32); node = np.zeros(max(n, 1), np.uint32)
    bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64); fn = np.zeros(max(n, 1), np.uint32); ff = np.zeros(max(n, 1), np.uint32)
    cnt = np.zeros(2, np.int32)
    _call("oracle_bow_transform", [len(node_weight), int(L), _c(child_offset, np.int32), _c(children, np.int32), _c(node_desc, np.uint8),
                                   _c(node_weight, np.float64), _c(node_word, np.uint32), desc if n else np.zeros(32, np.uint8), n, int(levelsup),
                                   int(accumulate), int(norm), word, node, bw, bv, cnt[0:1], fn, ff, cnt[1:2]])
    return word[:n], node[:n], bw[:cnt[0]], bv[:cnt[0]], fn[:cnt[1]], ff[:cnt[1]]


# TUM-VI stereo calibration as quoted by the reference's example/tum_vi/TUM_VI_stereo.yaml (equidistant fisheye lenses)
TUM_VI = {
    "camera": dict(fx=61.75453410721205, fy=61.75453410721205, cx=240.22941720459062, cy=255.73235402091632, cols=512, rows=512),
    "StereoRectifier.model": "fisheye",
    "StereoRectifier.K_left": [190.97847715128717, 0.0, 254.93170605935475, 0.0, 190.9733070521226, 256.8974428996504, 0.0, 0.0, 1.0],
    "StereoRectifier.D_left": [0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182],
    "StereoRectifier.R_left": [0.9997641946925044, 0.01925271884177015, 0.010044293307535757, -0.01901185247371587, 0.9995418997803748,
                               -0.02354867403818772, -0.010493068014919314, 0.02335216051329943, 0.99967223234568],
    "StereoRectifier.K_right": [190.44236969414825, 0.0, 252.59949716835982, 0.0, 190.4344384721956, 254.91723064636983, 0.0, 0.0, 1.0],
    "StereoRectifier.D_right": [0.0034003170790442797, 0.001766278153469831, -0.00266312569781606, 0.0003299517423931039],
    "StereoRectifier.R_right": [0.9997411981023351, 0.01955199401713946, 0.011629976219300583, -0.019819377433695273, 0.9995311538731381,
                                0.02333805294307984, -0.011168218078479885, -0.023562511898925578, 0.9996599816627474],
}


def fisheye_rectify_map(K, D4, R, cam, rows, cols):
    """cv::fisheye::initUndistortRectifyMap(K, D, R, K_rect(float), (cols, rows), CV_32F) -> map_x, map_y"""
    K = _c(K, np.float64); D4 = _c(D4, np.float64); R = _c(R, np.float64)
    Kr = np.array([cam["fx"], 0, cam["cx"], 0, cam["fy"], cam["cy"], 0, 0, 1], np.float64)
    mx = np.zeros((rows, cols), np.float32); my = np.zeros((rows, cols), np.float32)
    if _call("oracle_fisheye_rectify_map", [K, D4, R, Kr, rows, cols, mx, my], C.c_int) != 0:
        raise ValueError("K_rect * R is singular")
    return mx, my


# ---- reference build only (oracle/_ref/libplpref2.so, oracle/ref_driver2.cpp): entry points whose oracle counterpart has another shape
def _ref2_lib():
    with reference() as L:
        return L


def ref_line_extract(img, fx=520.0, fy=521.0, cx=None, cy=None, cap=4096):
    """feature::LineFeatureTracker(camera).extract_LSD_LBD of the reference build: (key lines, LBD, line functions)"""
    img = np.ascontiguousarray(img, np.uint8)
    L = _ref2_lib()
    kl = np.zeros(cap, KL_DTYPE); lbd = np.zeros((cap, 32), np.uint8); fn = np.zeros((cap, 3), np.float64)
    L.ref_line_extract.restype = C.c_int
    L.ref_line_extract.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_void_p] * 3 + [C.c_int]
    n = L.ref_line_extract(_p(img), img.shape[0], img.shape[1], fx, fy, img.shape[1] / 2 if cx is None else cx, img.shape[0] / 2 if cy is None else cy,
                           _p(kl), _p(lbd), _p(fn), cap)
    assert n >= 0
    return kl[:n].copy(), lbd[:n].copy(), fn[:n].copy()


def ref_lsd_keylines(img, cap=8192):
    """LSDDetectorC::detect as line_extractor.cc calls it: every key line, before the length-60 filter"""
    img = np.ascontiguousarray(img, np.uint8)
    L = _ref2_lib()
    kl = np.zeros(cap, KL_DTYPE)
    L.ref_lsd_keylines.restype = C.c_int
    L.ref_lsd_keylines.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = L.ref_lsd_keylines(_p(img), img.shape[0], img.shape[1], _p(kl), cap)
    assert n >= 0
    return kl[:n].copy()


def ref_lbd(img, keylines):
    """BinaryDescriptor::compute of the reference build on given key lines: (binary 32 B, float 72)"""
    img = np.ascontiguousarray(img, np.uint8); kl = _c(keylines, KL_DTYPE)
    L = _ref2_lib()
    out = np.zeros((len(kl), 32), np.uint8); o72 = np.zeros((len(kl), 72), np.float32)
    L.ref_lbd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    if len(kl):
        L.ref_lbd(_p(img), img.shape[0], img.shape[1], _p(kl), len(kl), _p(out), _p(o72))
    return out, o72


def lbd(img, keylines):
    """the oracle's LBD on given key lines"""
    img = np.ascontiguousarray(img, np.uint8); kl = _c(keylines, KL_DTYPE)
    out = np.zeros((len(kl), 32), np.uint8); o72 = np.zeros((len(kl), 72), np.float32)
    if len(kl):
        _load().oracle_lbd(_p(img), img.shape[0], img.shape[1], _p(kl), len(kl), _p(out), _p(o72))
    return out, o72


def ref_stereo_compute(levels_left, levels_right, kl, kr, dl, dr, scale_factors, inv_scale_factors, fxb, tb):
    """match::stereo(...).compute of the reference build; levels_*: lists of the pyramid level images"""
    L = _ref2_lib()
    nl = len(levels_left)
    ll = [np.ascontiguousarray(a, np.uint8) for a in levels_left]; lr = [np.ascontiguousarray(a, np.uint8) for a in levels_right]
    pl = (C.c_void_p * nl)(*[a.ctypes.data for a in ll]); pr = (C.c_void_p * nl)(*[a.ctypes.data for a in lr])
    rows = np.array([a.shape[0] for a in ll], np.int32); cols = np.array([a.shape[1] for a in ll], np.int32)
    kl = _c(kl, KP_DTYPE); kr = _c(kr, KP_DTYPE); dl = _c(dl, np.uint8); dr = _c(dr, np.uint8)
    sf = _c(scale_factors, np.float32); isf = _c(inv_scale_factors, np.float32)
    xr = np.zeros(max(len(kl), 1), np.float32); dp = np.zeros(max(len(kl), 1), np.float32)
    L.ref_stereo_compute.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.ref_stereo_compute(pl, pr, _p(rows), _p(cols), nl, _p(kl), len(kl), _p(kr), len(kr), _p(dl), _p(dr), _p(sf), _p(isf), fxb, tb, _p(xr), _p(dp))
    return xr[:len(kl)].copy(), dp[:len(kl)].copy()


def ref_match_keyframes_mutually(g6, kps1, desc1, kps2, desc2, sf, lm1_valid, reproj_1in2, pred_1in2, lm2_valid, reproj_2in1, pred_2in1, margin):
    """projection::match_keyframes_mutually of the reference build, complete (both passes + cross check)"""
    L = _ref2_lib()
    n1, n2 = len(kps1), len(kps2)
    out = np.zeros(max(n1, 1), np.int32)
    a = [_c(g6, np.float64), _c(kps1, KP_DTYPE), _c(desc1, np.uint8)]
    b = [_c(kps2, KP_DTYPE), _c(desc2, np.uint8)]
    sf = _c(sf, np.float32)
    c = [_c(lm1_valid, np.uint8), _c(reproj_1in2, np.float64), _c(pred_1in2, np.uint32), _c(lm2_valid, np.uint8), _c(reproj_2in1, np.float64), _c(pred_2in1, np.uint32)]
    L.ref_match_keyframes_mutually.restype = C.c_uint
    L.ref_match_keyframes_mutually.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_float, C.c_void_p]
    num = L.ref_match_keyframes_mutually(_p(a[0]), _p(a[1]), _p(a[2]), n1, _p(b[0]), _p(b[1]), n2, _p(sf), len(sf), *[_p(v) for v in c], margin, _p(out))
    return out[:n1].copy(), num


def ref_robust_match_frame_and_keyframe(desc1, angle1, desc2, angle2, valid2, lowe_ratio, check_orientation):
    """robust::match_frame_and_keyframe of the reference build with an all-inlier essential solver (= its brute_force_match)"""
    L = _ref2_lib()
    n1, n2 = len(desc1), len(desc2)
    out = np.zeros(max(n1, 1), np.int32)
    v = [_c(desc1, np.uint8), _c(angle1, np.float32), _c(desc2, np.uint8), _c(angle2, np.float32), _c(valid2, np.uint8)]
    L.ref_robust_match_frame_and_keyframe.restype = C.c_uint
    L.ref_robust_match_frame_and_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    num = L.ref_robust_match_frame_and_keyframe(_p(v[0]), _p(v[1]), n1, _p(v[2]), _p(v[3]), _p(v[4]), n2, lowe_ratio, int(check_orientation), _p(out))
    return out[:n1].copy(), num


def std_sort_entries(entries):
    """this machine's std::sort on uint32 entries compared by bits 20..29, larger first (lsd.cpp's comparator on the gradient bin, pixel index as payload)"""
    e = np.ascontiguousarray(entries, np.uint32).copy()
    _load().oracle_std_sort_entries(_p(e), C.c_long(e.size))
    return e


def std_introsort_loop_entries(entries, depth_limit=-1):
    """libstdc++'s std::__introsort_loop on such entries (None when the oracle was built against another C++ library)"""
    e = np.ascontiguousarray(entries, np.uint32).copy()
    ok = _load().oracle_std_introsort_loop_entries(_p(e), C.c_long(e.size), int(depth_limit))
    return e if ok else None


def index_sort_by_size(sizes):
    """the reference's index_sort_by_size (std::sort of the bin indices by size, descending) with this machine's C++ library"""
    sz = np.ascontiguousarray(sizes, np.int32)
    idx = np.zeros(len(sz), np.uint32)
    _load().oracle_index_sort_by_size(_p(sz), len(sz), _p(idx))
    return idx


def angle_checker_last_tie():
    """1 if the oracle's last orientation check met equally full bins at the cut (definition D3: the reference's unstable std::sort decides)"""
    return int(_load().oracle_angle_checker_last_tie())
