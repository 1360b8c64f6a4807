import sys, time, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
for (n, p, neig, seed, reps) in [(20000, 20, None, 103, 40), (5000, 10, None, 102, 60), (50000, 20, 512, 104, 12)]:
    X, y = synth(n, p, seed)
    ts = []
    for r in range(reps):
        t0 = time.perf_counter(); out = bk.bigKRLS(y, X, Neig=neig, ctx=ctx); ctx.sync(); ts.append(time.perf_counter() - t0); del out
    ts = np.array(ts[2:])
    print(f"N={n} P={p} Neig={neig}: {len(ts)} fits  min {ts.min():.4f}  median {np.median(ts):.4f}  max {ts.max():.4f}  mean {ts.mean():.4f} s")
