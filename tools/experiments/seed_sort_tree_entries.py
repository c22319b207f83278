import sys, numpy as np
sys.path[:0] = ["/root/repo/tests", "/root/repo"]
import importlib
import oracle_lib as O
synth = importlib.import_module("structure-plp-slam_amd.synth")
frames = synth.replay(1234, 64, 480, 640)
out = []
rho = 2.0 / np.sin(np.pi * 22.5 / 180)
for f in frames[::8]:
    s = O.LineOracle(f).scaled.astype(np.int64)
    sh, sw = s.shape
    DA = s[1:, 1:] - s[:-1, :-1]; BC = s[:-1, 1:] - s[1:, :-1]
    gx = DA + BC; gy = DA - BC
    g2 = gx * gx + gy * gy
    norm = np.sqrt(g2 / 4.0)
    mx = norm.max()
    bins = (norm * (1023.0 / mx)).astype(np.int64)
    defined = ~(norm <= rho)
    yy, xx = np.mgrid[0:sh-1, 0:sw-1]
    pix = yy * sw + xx
    e = (pix | (defined.astype(np.int64) << 19) | (bins << 20)).astype(np.uint32).ravel()
    out.append(e)
    g = 0
    while np.sqrt(g / 4.0) <= rho: g += 1
    skip = int(np.sqrt(g / 4.0) * (1023.0 / mx))
    print(len(e), "defined", int(defined.sum()), "skip_key", skip, "max bin of undefined", int(bins[~defined].max()), "min bin defined", int(bins[defined].min()))
    out.append(np.array([skip], np.uint32))
np.concatenate(out).tofile("/tmp/ss/entries.bin")
