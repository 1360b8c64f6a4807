"""Run one eigendecomposition with the PQ_TIMING debug build of the library (development probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bigkrls_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbigkrls_hip_pqt.so")
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n = int(sys.argv[1])
ctx = bk.Context(0)
X, y = synth(n, 20, 7)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
K = ops.bGaussKernel(ctx.from_numpy(Xs), 20.0)
eo = ops.bEigen(K, n, 0.001); ctx.sync()
print("done", eo.lastkeeper)
