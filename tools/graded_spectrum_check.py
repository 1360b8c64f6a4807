"""The dense eigensolver on small symmetric matrices whose spectrum falls far below rounding (development tool, round 6):
A = U diag(s) U' with s graded over `decades` decades; residual and orthogonality of all returned pairs.
python tools/graded_spectrum_check.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd import ops
ctx = bk.Context(0)
rng = np.random.default_rng(5)
for n in (300, 512, 640, 1024, 2048):
    for decades in (6, 12, 20, 40):
        U, _ = np.linalg.qr(rng.standard_normal((n, n)))
        s = 7.6e3 * 10.0 ** (-np.linspace(0.0, decades, n))
        A = (U * s) @ U.T
        A = 0.5 * (A + A.T)
        Ad = ctx.from_numpy(A)
        for neig in (None, n // 2):
            eo = ops.bEigen(Ad, neig, -1.0)
            d = np.asarray(eo.values); Q = eo.vectors.to_numpy(); k = Q.shape[1]
            res = float(np.max(np.linalg.norm(A @ Q - Q * d[:k], axis=0)) / d[0])
            orth = float(np.max(np.abs(Q.T @ Q - np.eye(k))))
            verr = float(np.max(np.abs(d - s[:len(d)])) / s[0])
            flag = "" if (res < 1e-12 and orth < 1e-12) else "   <-- BAD"
            print("n=%5d decades=%2d neig=%s: resid/theta1 %.2e orth %.2e values %.2e%s" % (n, decades, str(neig), res, orth, verr, flag), flush=True)
