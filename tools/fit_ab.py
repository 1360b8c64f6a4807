"""A/B timing of two builds of the library in the same GPU session (development tool):
python tools/fit_ab.py N P libA.so libB.so  -- alternates fresh processes, prints the best of 3 fits each."""
import sys, os, subprocess
n, p = sys.argv[1], sys.argv[2]
libs = sys.argv[3:]
child = r'''
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else %r)
import bigkrls_amd._lib as L
L.LIB_PATH = sys.argv[1]
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n, p = int(sys.argv[2]), int(sys.argv[3])
X, y = synth(n, p, 104)
ctx = bk.Context(0)
best = 1e9
for rep in range(4):
    T = {}
    t0 = time.perf_counter(); out = bk.bigKRLS(y, X, ctx=ctx, timings=T); ctx.sync(); dt = time.perf_counter() - t0
    if rep: best = min(best, dt)
    del out
print("%%s best %%.4f s (eigen %%.4f)" %% (os.path.basename(sys.argv[1]), best, T["eigen"]))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for lib in libs:
        subprocess.run([sys.executable, "-c", child, os.path.abspath(lib), n, p], check=False)
