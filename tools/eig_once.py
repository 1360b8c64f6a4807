"""bEigen of the G(N,P,103) kernel three times (for rocprofv3): python tools/eig_once.py N P"""
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
for rep in range(3):
    t0 = time.perf_counter(); eo = ops.bEigen(K, None, 0.001); ctx.sync()
    print(f"rep {rep}: {1e3*(time.perf_counter()-t0):.1f} ms kept {eo.lastkeeper}", flush=True)
