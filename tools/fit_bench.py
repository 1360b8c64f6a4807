"""Time one bigKRLS() fit at an arbitrary config (development tool): python tools/fit_bench.py N P [Neig] [seed]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n, p = int(sys.argv[1]), int(sys.argv[2])
neig = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 104
X, y = synth(n, p, seed)
ctx = bk.Context(0, own_stream=bool(os.environ.get('BIGKRLS_S1_GRAPH') or os.environ.get('KNOB_AB_OWN_STREAM')))  # (a stream capture needs a stream of its own)
for rep in range(2):
    T = {}
    t0 = time.perf_counter()
    out = bk.bigKRLS(y, X, Neig=neig, ctx=ctx, timings=T)
    ctx.sync()
    print(f"rep{rep} N={n} P={p} Neig={neig} total {time.perf_counter()-t0:.3f}s lastkeeper={out['lastkeeper']} "
          f"lambda={out['lambda']:.6f} R2={out['R2']:.4f} Neff={out['Neffective']:.1f}")
    print("  ", {k: round(v, 4) for k, v in T.items()})
    d = out["K.eigenvalues"]
    print("   eig[0..2]", d[:3], "eig[-1]", d[-1], "sum", d.sum())
    import torch
    print("   torch mem GB", torch.cuda.memory_allocated() / 1e9, "ws GB", ctx.workspace_bytes() / 1e9)
    del out
