"""Inner-blocked panel QR (BIGKRLS_PQ=blocked) against the per-column one: eigenvalues, residual,
orthogonality, time, over sizes with 1 .. 79 workgroups per panel (development probe).
python tools/pq_check.py [sizes...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
sizes = [int(a) for a in sys.argv[1:]] or [300, 700, 1283, 3000, 7700, 20000]
ctx = bk.Context(0)
for n in sizes:
    p = 6 if n < 10000 else 20
    X, _ = synth(n, p, 1000 + n)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
    res = {}
    for mode in ("resident", "blocked", "resident", "blocked"):
        os.environ["BIGKRLS_PQ"] = mode
        t0 = time.perf_counter(); eo = ops.bEigen(K, None, 0.001); ctx.sync(); dt = time.perf_counter() - t0
        k = eo.lastkeeper; Q = eo.vectors; lam = eo.values[:k]
        R = ops.gemm(False, False, K, Q).to_numpy() - Q.to_numpy() * lam
        G = ops.gemm(True, False, Q, Q).to_numpy()
        print(f"PQ={mode:8s} N={n}: {dt*1e3:8.1f} ms, kept {k}, resid {np.abs(R).max()/lam[0]:.2e}, orth {np.abs(G-np.eye(k)).max():.2e}, "
              f"trace err {abs(eo.values.sum()-n)/n:.2e}", flush=True)
        res[mode] = eo.values.copy()
    print("   max |d_blocked - d_resident| / d1 =", np.abs(res["blocked"] - res["resident"]).max() / res["resident"][0], flush=True)
