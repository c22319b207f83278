import sys, numpy as np
sys.path[:0] = ["tests", "."]
import oracle_lib as O
from plp import plp
from test_index_models import _seed_entries
r = np.random.default_rng(0)
def want(e, d): return O.std_introsort_loop_entries(e, d)
fails = []
for n in [65, 66, 70, 80, 90, 100, 110, 127, 128, 129, 130, 150, 200, 257, 300]:
    for kind in range(8):
        e = _seed_entries(r, n, kind)
        for d in (1, 2, 3, 4, -1):
            g = plp.seed_introsort_debug(e, d); w = want(e, d)
            g2 = plp.seed_introsort_debug(e, d)
            if not np.array_equal(g, w):
                bad = np.nonzero(g != w)[0]
                fails.append((n, kind, d, int(bad[0]), int(bad[-1]), len(bad), int((g == 0).sum()), "rerun_same" if np.array_equal(g, g2) else "rerun_differs"))
print("task-regime fails:", len(fails))
for f in fails[:80]: print(f)
# smallest failing case in full
for n in range(65, 200):
    e = _seed_entries(np.random.default_rng(n), n, 1)
    g = plp.seed_introsort_debug(e, -1); w = want(e, -1)
    if not np.array_equal(g, w):
        print("first failing n", n)
        print("in ", (e >> 20).tolist())
        print("got", [(int(x >> 20), int(x & 0xfffff)) for x in g])
        print("exp", [(int(x >> 20), int(x & 0xfffff)) for x in w])
        for d in range(1, 8):
            print(d, np.array_equal(plp.seed_introsort_debug(e, d), want(e, d)))
        break
