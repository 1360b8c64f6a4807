import os, sys, subprocess
child = r'''
import sys, os, time
sys.path.insert(0, %r)
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
X, y = synth(20000, 20, 104)
ctx = bk.Context(0)
best = 1e9
for rep in range(4):
    T = {}
    t0 = time.perf_counter(); out = bk.bigKRLS(y, X, ctx=ctx, timings=T); ctx.sync(); dt = time.perf_counter() - t0
    if rep: best = min(best, dt)
    del out
print("AGG_MIN=%%s best %%.4f s" %% (os.environ.get("BIGKRLS_S1AGG_MIN"), best))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for v in ("14848", "12800", "10752", "16896", "8704"):
        subprocess.run([sys.executable, "-c", child], env=dict(os.environ, BIGKRLS_S1AGG_MIN=v), check=False)
