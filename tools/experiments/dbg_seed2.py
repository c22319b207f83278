import os, sys, numpy as np
sys.path[:0] = ["tests", "."]
import oracle_lib as O
from plp import plp
from test_index_models import _seed_entries
def want(e, d): return O.std_introsort_loop_entries(e, d)
cases = [(108, 1), (150, 2), (300, 7), (1000, 1), (3000, 1)]
print("lib", plp.LIB_PATH)
for flags in ("0", "1"):
    os.environ["PLP_SEED_SORT_DBG"] = flags
    for n, kind in cases:
        e = _seed_entries(np.random.default_rng(n), n, kind)
        w = want(e, -1)
        bad = 0
        for rep in range(30):
            g = plp.seed_introsort_debug(e, -1)
            bad += not np.array_equal(g, w)
        print("flags", flags, "n", n, "kind", kind, "failures of 30:", bad, flush=True)
if os.environ.get("PLP_FRONT_LIB"):
    sys.exit(0)
os.environ["PLP_SEED_SORT_DBG"] = "0"; os.environ["PLP_SEED_SORT_DBG_FILE"] = "/tmp/ss_log.bin"
n, kind = 300, 7
e = _seed_entries(np.random.default_rng(n), n, kind)
w = want(e, -1)
for rep in range(200):
    g = plp.seed_introsort_debug(e, -1)
    if not np.array_equal(g, w):
        log = np.fromfile("/tmp/ss_log.bin", np.int32)
        k = log[1]
        print("failing run, tasks:", k)
        recs = log[2:2 + 6 * k].reshape(k, 6)
        print(recs.tolist())
        bad = np.nonzero(g != w)[0]
        print("bad range", bad[0], bad[-1], len(bad))
        np.save("gpurun_out/r04a/fail_e.npy", e); np.save("gpurun_out/r04a/fail_g.npy", g); np.save("gpurun_out/r04a/fail_log.npy", recs)
        break
