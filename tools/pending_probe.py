"""Sporadic 10-100 ms stalls at the first device synchronisation of a fit, early in a process's life:
are they tied to the fit count or to the time since the process started? (development probe)"""
import gc, os, sys, time
T0 = time.perf_counter()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
X, y = synth(20000, 20, 103)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "sleep":
    x = torch.zeros(10, device=ctx.device); torch.cuda.synchronize()
    time.sleep(6.0)
if mode == "nogc":
    gc.collect(); gc.freeze()
big = torch.randn(4096, 4096, device=ctx.device) if mode == "busy" else None
for rep in range(12):
    T = {}
    t0 = time.perf_counter()
    out = bk.bigKRLS(y, X, ctx=ctx, timings=T)
    t1 = time.perf_counter()
    del out
    if mode == "idle":
        time.sleep(0.25)
    if mode == "busy":   # keep the GPU busy while the host prepares the next fit
        for _ in range(6):
            big @ big
    print(f"{mode} rep {rep}: at {t0-T0:.2f}s fit {t1-t0:.4f}s wall {T['wall']:.4f} h2d {1e3*T['h2d']:.2f} ms")
