"""The reference's real per-frame call pattern, run and CHECKED: two host threads per frame, ORB in one and LSD + LBD in the other
(`/root/reference/src/PLPSLAM/data/frame.cc:691-694, 1143-1147`: `std::thread(&frame::extract_orb ...)`, `std::thread(&frame::extract_line ...)`,
two `join`s), through the single-frame host-pointer entries `plp_orb_extract` || `plp_line_extract` (the latter runs `k_lsd_grow_mw`, the
several-waves region grower, beside the ORB kernels of the other thread), optionally with a third thread that keeps `plp_match_host` busy to widen
co-residency.  Every result of every pair is compared with the CPU oracle, every call must return a clean status.

Shared by tests/test_gpu_concurrent_single_frame.py (>= 2 000 pairs) and tools/soak_concurrent_pairs.py (the long form, log under profiles/).
"""
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import match_cases as MC
import oracle_lib as O


class Expected:
    """oracle results of one frame: ORB (K key points) and the line front end in both seed orders"""

    def __init__(self, img, K):
        self.img = np.ascontiguousarray(img)
        self.kps, self.desc = O.OrbOracle(K).extract(self.img)
        self.lines = {}
        for stable in (False, True):
            lo = O.LineOracle(self.img, stable_order=stable)
            self.lines[stable] = (np.asarray(lo.keylsd), np.asarray(lo.lbd), np.asarray(lo.linefn))


class MatchLoad(threading.Thread):
    """third thread: one landmark-projection problem after another through plp_match_host, each compared with the oracle"""

    def __init__(self, plp, seed=7):
        super().__init__(daemon=True)
        self.plp, self.stop_flag, self.calls, self.error = plp, threading.Event(), 0, None
        rng = np.random.default_rng(seed)
        self.grid = plp.make_grid(640, 480)
        self.sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
        self.problems = []
        for _ in range(4):
            t, q = MC.random_problem(rng, 600, 800, n_words=8)
            want, wn = O.match_frame_and_landmarks(O.grid6(self.grid), t["t_kps"], t["t_desc"], t["t_x_right"], t["t_occupied"], self.sf, q["q_valid"], q["q_reproj"],
                                                   q["q_x_right"], q["q_level"], q["q_desc"], q["q_has_obs"], 12.0, 0.8)
            self.problems.append(({**t, **q}, want, wn))

    def run(self):
        try:
            mt = self.plp.matcher(0.8, True)
            while not self.stop_flag.is_set():
                f, want, wn = self.problems[self.calls % len(self.problems)]
                got, gn = mt.match_host(self.plp.MODE_LANDMARKS, 600, 800, f, margin=12.0, scale_factors=self.sf, grid=self.grid)
                if gn[0] != wn or not np.array_equal(got[0], want):
                    raise AssertionError(f"plp_match_host call {self.calls}: matches differ from the oracle")
                self.calls += 1
        except BaseException as e:      # noqa: BLE001 -- reported by the caller
            self.error = e


def run_pairs(plp, expected, n_pairs, K, stable, with_matcher=True, ex=None, lt=None, deadline_s=None, log=None):
    """`n_pairs` frames, each extracted by two threads at once; returns (pairs done, matcher calls).  Raises on the first difference."""
    ex = ex or plp.orb_extractor(K)
    lt = lt or plp.LineFeatureTracker()
    lt.set_seed_order(plp.SEED_ORDER_STABLE if stable else plp.SEED_ORDER_LIBSTDCXX)
    load = MatchLoad(plp) if with_matcher else None
    if load:
        load.start()
    pool = ThreadPoolExecutor(2)
    done, t0 = 0, time.perf_counter()
    try:
        for i in range(n_pairs):
            e = expected[i % len(expected)]
            fo = pool.submit(ex.extract, e.img)                 # thread 1: plp_orb_extract
            fl = pool.submit(lt.extract_LSD_LBD, e.img)         # thread 2: plp_line_extract (raises PlpError on any status bit)
            kps, desc = fo.result()
            kl, lbd, fn = fl.result()
            assert np.array_equal(kps, e.kps) and np.array_equal(desc, e.desc), f"pair {i}: ORB differs from the oracle beside a line extraction"
            wkl, wlbd, wfn = e.lines[stable]
            assert len(kl) == len(wkl) and np.array_equal(kl, wkl), f"pair {i}: key lines differ from the oracle beside an ORB extraction"
            assert np.array_equal(lbd, wlbd) and np.array_equal(fn, wfn), f"pair {i}: LBD / line functions differ from the oracle"
            if load and load.error:
                raise load.error
            done += 1
            if log and done % 1000 == 0:
                log(f"  {done} pairs, {load.calls if load else 0} matcher calls, {time.perf_counter() - t0:.0f} s")
            if deadline_s is not None and time.perf_counter() - t0 > deadline_s:
                break
    finally:
        if load:
            load.stop_flag.set()
            load.join(30)
        pool.shutdown(wait=True)
    if load and load.error:
        raise load.error
    return done, (load.calls if load else 0)
