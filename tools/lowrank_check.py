"""Numerically low-rank kernels (P = 2: the spectrum falls below rounding long before Neig) through the block Lanczos and
the dense path (development tool, round 6): quality of the kept pairs, and the same as a fit.
python tools/lowrank_check.py [N] [P] [NEIG]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, p, neig = (sys.argv[1:4] + ["20000", "2", "512"][len(sys.argv) - 1:])[:3]
child = r'''
import sys, os, time
sys.path.insert(0, %r)
import numpy as np
import bigkrls_amd as bk
import bigkrls_amd._lib as L
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n, p, neig = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X, y = synth(n, p, 7 + n %% 97)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
ctx = bk.Context(0)
K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
eo = ops.bEigen(K, neig, 0.001)
d = np.asarray(eo.values); k = eo.lastkeeper
Qh = eo.vectors.to_numpy()
KQh = ops.gemm(False, False, K, eo.vectors).to_numpy()
r = np.linalg.norm(KQh - Qh * d[:k], axis=0) / d[0]
print("eigen: kept %%d of %%d; residual/theta1 max %%.2e (pair %%d), orth %%.2e; theta[k-1]/theta1 %%.2e theta[last]/theta1 %%.2e" %% (k, neig, r.max(), int(r.argmax()), float(np.max(np.abs(Qh.T @ Qh - np.eye(k)))), d[k-1]/d[0], d[-1]/d[0]), flush=True)
np.save(sys.argv[4], d)
try:
    out = bk.bigKRLS(y, X, Neig=neig, ctx=ctx, noisy=False)
    print("fit: ok, lambda %%.12g kept %%d |c| %%.12g counters %%s" %% (out["lambda"], out["lastkeeper"], float(np.linalg.norm(out["coeffs"])), ctx.counters()), flush=True)
except L.BigKRLSError as e:
    print("fit: ERROR", e, ctx.counters(), flush=True)
''' % ROOT
import numpy as np
vals = {}
for tag, env in (("krylov", {"BIGKRLS_EIGK": "krylov"}), ("dense", {"BIGKRLS_EIGK": "dense"})):
    e = dict(os.environ); e.update(env); e["BIGKRLS_VERBOSE"] = "1"
    out = "/tmp/lowrank_%s.npy" % tag
    r = subprocess.run([sys.executable, "-c", child, n, p, neig, out], env=e, capture_output=True, text=True)
    keep = [l for l in (r.stdout + r.stderr).splitlines() if ("Lanczos" in l or l.startswith("eigen:") or l.startswith("fit:") or "rror" in l or "left to" in l)]
    seen = []
    for l in keep:
        if l not in seen: seen.append(l)
    print("---- %s" % tag); print("\n".join(seen[:40]), flush=True)
    if os.path.exists(out): vals[tag] = np.load(out); os.remove(out)
if len(vals) == 2:
    a, b = vals["krylov"], vals["dense"]
    print("max |theta(krylov) - theta(dense)| / theta_1 = %.2e" % float(np.max(np.abs(a - b)) / b[0]))
