import sys, time, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n, p = int(sys.argv[1]), int(sys.argv[2])
trunc = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ctx = bk.Context(0)
X, y = synth(n, p, 102)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
for rep in range(2):
    ctx.sync(); t0 = time.perf_counter()
    eo = ops.bEigen(K, n, trunc)
    ctx.sync(); dt = time.perf_counter() - t0
    print(f"N={n} trunc={trunc} lastkeeper={eo.lastkeeper} eigen {dt:.3f}s")
Q = eo.vectors
G = ops.bCrossProd(Q).to_numpy()
print(" orth", np.abs(G - np.eye(Q.ncol)).max())
Kh = K.to_numpy(); Qh = Q.to_numpy()
print(" resid", np.abs(Kh @ Qh - Qh * eo.values[:Q.ncol]).max() / eo.values[0])
