#!/usr/bin/env python3
"""Long form of tests/test_gpu_concurrent_single_frame.py: the reference's per-frame pattern (`data/frame.cc:691-694, 1143-1147`: ORB and lines of one
frame in two host threads) through `plp_orb_extract` || `plp_line_extract`, with a third thread on `plp_match_host`, every result compared with the
CPU oracle, for a number of pairs or a time budget.  Alternates the seed order every block of pairs.
    python tools/soak_concurrent_pairs.py [--pairs 30000] [--minutes 12] [--frames 64] [--seed 1234]"""
import argparse, importlib, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import concurrent_pairs as CP
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=30000); ap.add_argument("--minutes", type=float, default=12.0)
    ap.add_argument("--frames", type=int, default=64); ap.add_argument("--seed", type=int, default=1234); ap.add_argument("--block", type=int, default=2500)
    a = ap.parse_args()
    K = 1000
    t0 = time.time()
    expected = [CP.Expected(f, K) for f in synth.replay(a.seed, a.frames, 480, 640)]
    print(f"oracle results of {a.frames} frames (ORB K = {K}, lines in both seed orders): {time.time() - t0:.0f} s", flush=True)
    ex, lt = plp.orb_extractor(K), plp.LineFeatureTracker()
    t_end = time.time() + 60 * a.minutes
    total = calls = 0
    stable = False
    while total < a.pairs and time.time() < t_end:
        n, c = CP.run_pairs(plp, expected, min(a.block, a.pairs - total), K, stable, with_matcher=True, ex=ex, lt=lt, deadline_s=max(1.0, t_end - time.time()))
        total += n; calls += c
        print(f"{'stable' if stable else 'std::sort'} seed order: {n} pairs (plp_orb_extract || plp_line_extract), {c} plp_match_host calls beside them: all equal to the oracle, status clean; "
              f"{total} pairs so far, {time.time() - t0:.0f} s", flush=True)
        stable = not stable
    print(f"TOTAL {total} concurrent pairs, {calls} matcher calls, 0 mismatches, 0 status bits")


if __name__ == "__main__":
    main()
