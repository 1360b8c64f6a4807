"""Soak run (development): many fits of mixed sizes back to back on one context, eigen-only decompositions in between,
with a per-fit time log; reports any fit slower than 1.5x the median of its size (stalls, watchdog retries)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
cases = [(20000, 20, None, 103), (5000, 10, None, 102), (12000, 8, None, 7), (16384, 6, 160, 12), (50000, 20, 512, 104),
         (3000, 5, None, 9), (7777, 12, None, 3)]
data = {c: synth(c[0], c[1], c[3]) for c in cases}
times = {c: [] for c in cases}
lam = {}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t_all = time.perf_counter()
for r in range(rounds):
    for c in cases:
        X, y = data[c]
        t0 = time.perf_counter()
        out = bk.bigKRLS(y, X, Neig=c[2], ctx=ctx)
        ctx.sync()
        times[c].append(time.perf_counter() - t0)
        if c in lam:
            assert out["lambda"] == lam[c], (c, out["lambda"], lam[c])      # bitwise reproducible
        lam[c] = out["lambda"]
        del out
print(f"{rounds} rounds in {time.perf_counter() - t_all:.1f} s")
bad = 0
for c in cases:
    t = np.array(times[c][1:])
    med = float(np.median(t))
    slow = int((t > 1.5 * med).sum())
    bad += slow
    print(f"N={c[0]:6d} P={c[1]:2d} Neig={c[2]}: median {med:.4f} s  max {t.max():.4f} s  slow fits {slow}")
print("OK" if bad == 0 else f"{bad} slow fits")
