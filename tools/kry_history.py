import sys, time, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n, p, neig, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
X, y = synth(n, p, seed)
ctx = bk.Context(0)
out = bk.bigKRLS(y, X, Neig=neig, ctx=ctx, derivative=False, vcov_est=False)
print("lastkeeper", out["lastkeeper"])
