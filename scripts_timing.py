import sys, time, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from oracle import krls_oracle as orc
n, p = int(sys.argv[1]), int(sys.argv[2])
X, y = orc.synth(n, p, 103)
ctx = bk.Context(0)
for rep in range(2):
    T = {}
    t0 = time.perf_counter()
    out = bk.bigKRLS(y, X, ctx=ctx, timings=T)
    ctx.sync()
    print(f"rep{rep} N={n} P={p} total {time.perf_counter()-t0:.3f}s lastkeeper={out['lastkeeper']} lambda={out['lambda']:.6f} R2={out['R2']:.4f}")
    print("  ", {k: round(v, 4) for k, v in T.items()})
    del out
