"""C4 through the row-block path (world = 1 here; one process per GPU under torchrun): development probe."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bigkrls_amd as bk
from bigkrls_amd import dist as bkdist
from bigkrls_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ctx = bk.Context(0)
X, y = synth(n, 20, 104)
for rep in range(2):
    T = {}
    t0 = time.perf_counter()
    out = bkdist.bigKRLS_dist(y, X, Neig=512, ctx=ctx, timings=T, keep_outputs=False)
    ctx.sync()
    print(f"rep{rep} dist path N={n}: {time.perf_counter()-t0:.3f}s lastkeeper={out['lastkeeper']} lambda={out['lambda']:.6f}",
          {k: round(v, 3) for k, v in T.items()})
    del out
