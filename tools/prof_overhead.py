"""Overhead of the library's HIP-event sampling on the fit (development probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
X, y = synth(20000, 20, 103)
out = bk.bigKRLS(y, X, ctx=ctx); del out
for mode in (False, True, False, True):
    ctx.set_profile(mode)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); out = bk.bigKRLS(y, X, ctx=ctx); ctx.sync(); dt = time.perf_counter() - t0
        best = min(best, dt); del out
    print("profile", mode, "best %.4f s" % best)
ctx.set_profile(False)
