import sys, time, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from oracle import krls_oracle as orc
n, p = int(sys.argv[1]), int(sys.argv[2])
X, y = orc.synth(n, p, 103)
ctx = bk.Context(0)
T = {}
out = bk.bigKRLS(y, X, ctx=ctx, timings=T)
ctx.sync()
print({k: round(v, 4) for k, v in T.items()})
