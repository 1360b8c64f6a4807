"""Aggregated trailing update (two panels per pass over A22, stage 1) against one update per panel
(BIGKRLS_S1AGG=0): eigenvalues, residual, orthogonality, time (development probe).
python tools/agg_check.py [N] [P]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = bk.Context(0)
X, _ = synth(n, p, 103)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
res = {}
for mode in ("0", "1", "0", "1"):
    os.environ["BIGKRLS_S1AGG"] = mode
    t0 = time.perf_counter(); eo = ops.bEigen(K, None, 0.001); ctx.sync(); dt = time.perf_counter() - t0
    k = eo.lastkeeper; Q = eo.vectors; lam = eo.values[:k]
    R = ops.gemm(False, False, K, Q).to_numpy() - Q.to_numpy() * lam
    G = ops.gemm(True, False, Q, Q).to_numpy()
    print(f"S1AGG={mode} N={n}: {dt*1e3:.1f} ms, kept {k}, resid {np.abs(R).max()/lam[0]:.2e}, orth {np.abs(G-np.eye(k)).max():.2e}, "
          f"trace err {abs(eo.values.sum()-n)/n:.2e}", flush=True)
    res[mode] = eo.values.copy()
print("max |d_agg - d_plain| / d1 =", np.abs(res["1"] - res["0"]).max() / res["0"][0])
