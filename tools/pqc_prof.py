"""Clock ticks per phase of pq_chol (CholeskyQR2 + Householder reconstruction panel kernel), summed over the launches
of one decomposition: needs the -DBK_PQC_PROF build of the library (tools/_ab/libbigkrls_pqcprof.so). Development tool.
python tools/pqc_prof.py N P"""
import sys, os, ctypes as C
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bigkrls_amd._lib as L
L.LIB_PATH = os.path.join(root, "tools", "_ab", "libbigkrls_pqcprof.so")
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n, p = int(sys.argv[1]), int(sys.argv[2])
ctx = bk.Context(0)
X, _ = synth(n, p, 103)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
lib = L.load()
buf = (C.c_ulonglong * 16)()
ops.bEigen(K, None, 0.001); ctx.sync()
lib.bk_pqc_prof_get(buf, 1)
ops.bEigen(K, None, 0.001); ctx.sync()
lib.bk_pqc_prof_get(buf, 1)
names = ["load / between stages", "Gram (x2)", "all-reduce (x2)", "Cholesky (x2)", "broadcast of Q1", "LU", "R product", "substitution (x3)", "stores"]
panels = (n - 64) // 64
tot = sum(buf[i] for i in range(9))
print(f"n={n}: {panels} panels, {tot * 0.01 / panels:.1f} us per panel in workgroup 0")
for i, nm in enumerate(names):
    print(f"  {nm:24s} {buf[i] * 0.01 / panels:8.1f} us per panel")
