"""cycles of the exact seed sort's phases on the seed array of a real frame (debug entry, one workgroup)"""
import os, sys, time, numpy as np
sys.path[:0] = ["tests", "."]
import oracle_lib as O
from plp import plp, synth
from PIL import Image
os.environ["PLP_SEED_SORT_DBG"] = "0"; os.environ["PLP_SEED_SORT_DBG_FILE"] = "/tmp/ss_log.bin"
frames = [np.asarray(Image.open("tests/golden/equirect2_640x480.png")), synth.replay(1234, 1, 480, 640)[0]]
names = ["G_partitions", "win_load", "wg_levels", "wave_tasks", "lanes", "store", "n_windows", "n_G", "task_busy_sum", "task_max", "reg_cycles", "reg_calls", "reg_parts", "sub_cycles(incl reg)", "sub_calls", "sub_parts", "G_A", "G_S", "G_X", "G_B", "G_C", "G_n", "", ""]
for f in frames:
    s = O.LineOracle(f, False).scaled.astype(np.int64)
    DA, BC = s[1:, 1:] - s[:-1, :-1], s[:-1, 1:] - s[1:, :-1]
    norm = np.sqrt(((DA + BC) ** 2 + (DA - BC) ** 2) / 4.0)
    rho = 2.0 / np.sin(np.pi * 22.5 / 180)
    bins = (norm * (1023.0 / norm[norm > rho].max())).astype(np.int64).ravel()
    e = (bins.astype(np.uint32) << np.uint32(20)) | ((norm > rho).ravel().astype(np.uint32) << np.uint32(19)) | np.arange(bins.size, dtype=np.uint32)
    skip = int(bins[(norm > rho).ravel()].min())
    w = plp.model_seed_introsort(e, -1, skip)
    for rep in range(3):
        t0 = time.perf_counter(); g, nl = plp.seed_introsort_debug(e, -1, skip, return_live=True); dt = time.perf_counter() - t0
    assert np.array_equal(g[:nl], w[:nl]), 'live part differs from the model'
    print('n_live', nl, 'of', e.size)
    print('skip key', skip)
    log = np.fromfile("/tmp/ss_log.bin", np.int32)
    t = log[2 + 6 * 4000:].view(np.int64)
    print({k: int(v) for k, v in zip(names, t)}, "tasks", int(log[1]), "total cycles (100 MHz clock64 ticks?)", int(t[:6].sum()))
