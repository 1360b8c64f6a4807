"""Eigen-decomposition residuals over a sweep of sizes that cross every size-dependent switch of the
dense path (QL leaves n > 64, two-stage n > 256, panel groups of 4, factored D&C levels n >= 4096,
two-level panel-QR exchange n > ~7000) -- development probe.  python tools/size_sweep.py [large]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
rng = np.random.default_rng(3)
sizes = [2, 3, 17, 63, 64, 65, 66, 127, 129, 255, 256, 257, 258, 300, 321, 449, 512, 513, 577, 640, 705, 1000,
         1023, 1025, 2047, 2049, 3000, 4095, 4096, 4097, 5003, 6912, 6913, 7169, 9001]
# round 4: the stage-2 back-transform as chains (n > 8000), four panels per trailing update (trailing matrix >= 12 800
# rows, i.e. n >= 13 121), pairs below (>= 10 752)
if len(sys.argv) > 1 and sys.argv[1] == "large":
    sizes = [7999, 8000, 8001, 8033, 8191, 10751, 11009, 11073, 13055, 13121, 13185, 13377, 13441, 14001]
# round 6: the panel chain on one stream below 6 144 trailing rows (every n crosses it; the first panels of n = 6 2xx start
# just above), eight panels per merged back-transform block at every size, the dense path at the sizes where the block
# Lanczos takes over for Neig << N
if len(sys.argv) > 1 and sys.argv[1] == "r6":
    sizes = [6143, 6207, 6208, 6209, 6273, 6337, 12863, 12929, 16383, 16384, 16385, 20011]
bad = 0
for n in sizes:
    p = 4
    X, _ = synth(n, p, 1000 + n)
    sd = X.std(0, ddof=1) if n > 1 else np.ones(p)
    Xs = (X - X.mean(0)) / np.where(sd > 0, sd, 1.0)
    K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
    for neig, trunc in ((None, -1.0), (max(1, n // 20), -1.0), (None, 0.01)):
        eo = ops.bEigen(K, neig, trunc)
        k = eo.lastkeeper
        Q = eo.vectors
        lam = eo.values[:k]
        R = ops.gemm(False, False, K, Q).to_numpy() - Q.to_numpy() * lam
        G = ops.gemm(True, False, Q, Q).to_numpy()
        res = np.abs(R).max() / max(abs(lam[0]), 1e-300)
        orth = np.abs(G - np.eye(k)).max()
        tr = abs(eo.values.sum() - n) / n if neig is None else 0.0
        ok = res < 1e-11 and orth < 1e-10 and tr < 1e-11
        bad += not ok
        if not ok or (neig is None and trunc < 0):
            print(f"n={n:5d} Neig={neig} trunc={trunc}: kept {k:5d} resid {res:.1e} orth {orth:.1e} trace {tr:.1e}{'' if ok else '   <-- CHECK'}")
print("problems:", bad)
