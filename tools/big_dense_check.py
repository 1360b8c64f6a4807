"""Dense path above the sizes the resident kernels cover (n > 32768: bulge chasing by wavefront launches,
panel QR by column steps for the first panels): truncated eigenpairs, residual / orthogonality / trace
(development probe).  python tools/big_dense_check.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 34000
p = 8
ctx = bk.Context(0)
X, _ = synth(n, p, 77)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
os.environ["BIGKRLS_EIGK"] = "dense"
t0 = time.perf_counter(); eo = ops.bEigen(K, None, 0.001); ctx.sync(); dt = time.perf_counter() - t0
k = eo.lastkeeper
Q = eo.vectors
lam = eo.values[:k]
R = ops.gemm(False, False, K, Q).to_numpy() - Q.to_numpy() * lam
G = ops.gemm(True, False, Q, Q).to_numpy()
print(f"N={n}: {dt:.2f} s, kept {k}, max|K q - lam q|/lam1 = {np.abs(R).max()/lam[0]:.2e}, "
      f"max|Q'Q - I| = {np.abs(G - np.eye(k)).max():.2e}, |sum(lam) - N|/N = {abs(eo.values.sum() - n)/n:.2e}")
