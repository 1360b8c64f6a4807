"""One C4-shaped fit (N=50000, P=20, Neig=512) for rocprofv3 (development probe)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ctx = bk.Context(0)
X, y = synth(n, 20, 104)
T = {}
out = bk.bigKRLS(y, X, Neig=512, ctx=ctx, timings=T)
print({k: round(v, 3) for k, v in T.items()})
