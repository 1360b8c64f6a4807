"""Run the same eigendecomposition twice and demand bitwise identical results (race detector)."""
import sys, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n, p = int(sys.argv[1]), int(sys.argv[2])
ctx = bk.Context(0)
X, y = synth(n, p, 7)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
res = []
for rep in range(3):
    eo = ops.bEigen(K, n, 0.001)
    res.append((eo.values.copy(), eo.vectors.to_numpy()))
    junk = ctx.torch.rand((2000, 2000), device=ctx.device)  # perturb the allocator / timing
for i in (1, 2):
    dv = np.abs(res[i][0] - res[0][0]).max() / res[0][0][0]
    dq = np.abs(np.abs(res[i][1]) - np.abs(res[0][1])).max()
    print(f"N={n}: run {i} vs 0: max rel eigenvalue diff {dv:.3e}, eigenvector diff {dq:.3e}", "BITWISE" if dv == 0 and dq == 0 else "DIFFERENT")
