"""Block-Lanczos top-k path vs the dense path (development probe)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n, p, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = bk.Context(0)
X, y = synth(n, p, 104)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
os.environ["BIGKRLS_EIGK"] = "dense"
t0 = time.perf_counter(); d = ops.bEigen(K, k, 0.001); ctx.sync(); td = time.perf_counter() - t0
os.environ["BIGKRLS_EIGK"] = "krylov"
t0 = time.perf_counter(); a = ops.bEigen(K, k, 0.001); ctx.sync(); ta = time.perf_counter() - t0
t0 = time.perf_counter(); a = ops.bEigen(K, k, 0.001); ctx.sync(); ta2 = time.perf_counter() - t0
print(f"N={n} Neig={k}: dense {td:.3f}s  krylov {ta:.3f}s (2nd {ta2:.3f}s)  lastkeeper {d.lastkeeper} / {a.lastkeeper}")
print("  max rel eigenvalue diff:", np.max(np.abs(a.values - d.values)) / d.values[0])
Qd, Qa = d.vectors.to_numpy(), a.vectors.to_numpy()
nv = min(Qd.shape[1], Qa.shape[1])
G = Qd[:, :nv].T @ Qa[:, :nv]
print("  ortho of krylov Q:", np.max(np.abs(Qa.T @ Qa - np.eye(Qa.shape[1]))))
# projector difference on the kept subspace (sign/rotation invariant)
yv = (y - y.mean()) / y.std(ddof=1)
w = 1.0 / (d.values[:nv] + 1.0)
cd = Qd[:, :nv] @ (w * (Qd[:, :nv].T @ yv)); ca = Qa[:, :nv] @ (w * (Qa[:, :nv].T @ yv))
print("  rel diff of c = Q(w o Q'y):", np.max(np.abs(cd - ca)) / np.max(np.abs(cd)))
