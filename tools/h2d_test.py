"""Where does the occasional 10-70 ms 'h2d' phase of a fit come from? (development probe)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
X, y = synth(20000, 20, 103)
orig = ctx.from_numpy
def timed(a):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    aa = np.ascontiguousarray(np.asarray(a, dtype=np.float64).T if np.asarray(a).ndim > 1 else np.asarray(a, dtype=np.float64)[None, :])
    t1 = time.perf_counter()
    r = orig(a)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"   from_numpy{np.asarray(a).shape}: pre-sync+transpose {1e3*(t1-t0):.2f} ms, copy {1e3*(t2-t1):.2f} ms")
    return r
ctx.from_numpy = timed
for rep in range(4):
    T = {}
    t0 = time.perf_counter()
    out = bk.bigKRLS(y, X, ctx=ctx, timings=T)
    print(f"rep {rep}: total {time.perf_counter()-t0:.3f}s h2d phase {T['h2d']*1e3:.2f} ms")
    del out
