"""Block Lanczos over a spread of shapes (development tool, round 6): for each (N, P, Neig) the eigenpairs of the Gaussian
kernel with the library's defaults and with full checks only (BIGKRLS_KRY_NOEST=1, fresh process each), their residual
against K, orthogonality, and the difference of the two sets of eigenvalues; BIGKRLS_VERBOSE lines show where the
estimate on the compressed projected problem replaced a check.   python tools/kry_sweep.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(20000, 10, 128), (24000, 4, 256), (30000, 3, 512), (40000, 15, 300), (60000, 30, 400), (36000, 8, 768),
          (50000, 2, 512), (17000, 10, 60)]
child = r'''
import sys, os, time
sys.path.insert(0, %r)
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n, p, neig = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X, _ = synth(n, p, 7 + n %% 97)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
ctx = bk.Context(0)
K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
t0 = time.perf_counter(); eo = ops.bEigen(K, neig, 0.001); ctx.sync(); dt = time.perf_counter() - t0
t0 = time.perf_counter(); eo = ops.bEigen(K, neig, 0.001); ctx.sync(); dt = time.perf_counter() - t0
d = np.asarray(eo.values)
Q = eo.vectors
k = eo.lastkeeper
Qh = Q.to_numpy()
KQh = ops.gemm(False, False, K, Q).to_numpy()
res = float(np.max(np.linalg.norm(KQh - Qh * d[:k], axis=0)) / d[0])
orth = float(np.max(np.abs(Qh.T @ Qh - np.eye(k))))
np.save(sys.argv[4], d)
print("n=%%d p=%%d neig=%%d kept=%%d  %%.3f s  resid/theta1 %%.2e  orth %%.2e  counters %%s" %% (n, p, neig, k, dt, res, orth, ctx.counters()), flush=True)
''' % ROOT
import numpy as np
for n, p, neig in SHAPES:
    vals = []
    for tag, env in (("default", {}), ("full_checks_only", {"BIGKRLS_KRY_NOEST": "1"})):
        e = dict(os.environ)
        e.update(env)
        e["BIGKRLS_VERBOSE"] = "1"
        out = "/tmp/kry_sweep_%s.npy" % tag
        r = subprocess.run([sys.executable, "-c", child, str(n), str(p), str(neig), out], env=e, capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).splitlines() if "Lanczos" in l or l.startswith("n=") or "Error" in l or "error" in l]
        # (the second decomposition's lines repeat the first's)
        seen = []
        for l in lines:
            if l not in seen:
                seen.append(l)
        print("---- %s" % tag)
        print("\n".join(seen), flush=True)
        if os.path.exists(out):
            vals.append(np.load(out))
            os.remove(out)
    if len(vals) == 2:
        print("==== n=%d neig=%d: max |theta(default) - theta(full checks)| / theta_1 = %.2e" % (n, neig, float(np.max(np.abs(vals[0] - vals[1])) / vals[0][0])), flush=True)
