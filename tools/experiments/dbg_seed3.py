import os, sys, numpy as np
sys.path[:0] = ["tests", "."]
import oracle_lib as O
from plp import plp
from test_index_models import _seed_entries
def want(e, d): return O.std_introsort_loop_entries(e, d)
for n, kind in [(300, 7), (1000, 1), (108, 1), (3000, 2), (20000, 1)]:
    e = _seed_entries(np.random.default_rng(n), n, kind)
    w = want(e, -1)
    nfail = sum(not np.array_equal(plp.seed_introsort_debug(e, -1), w) for rep in range(100))
    print(n, kind, "failures of 100:", nfail, flush=True)
