import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 3:
    import bigkrls_amd._lib as L
    L.LIB_PATH = os.path.abspath(sys.argv[3])      # A/B: another build of the library
import bigkrls_amd as bk
from bigkrls_amd import ops
n, p = int(sys.argv[1]), int(sys.argv[2])
ctx = bk.Context(0)
rng = np.random.default_rng(0)
X = ctx.from_numpy(rng.standard_normal((n, p)))
K = ops.bGaussKernel(X, float(p)); ctx.sync()
ctx.set_profile(True)
for _ in range(5):
    K = ops.bGaussKernel(X, float(p))
ms, fl, cnt = ctx.get_profile("kernel_block")
ms /= cnt; fl /= cnt
print(f"N={n} P={p}: {ms*1e3:.1f} us  {fl/ms/1e9:.2f} TFLOP/s  write {8.0*n*n/ms/1e6:.0f} GB/s")
