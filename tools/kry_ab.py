"""Same-box A/B of the block-Lanczos knobs (development tool): fits of one Neig << N configuration in fresh processes,
alternating the environments below; prints the best of 3 fits after a warm-up fit and the eigen phase.
python tools/kry_ab.py N P NEIG [rounds]"""
import os
import subprocess
import sys

n, p, neig = sys.argv[1:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [
    ("default", {}),
    ("full_checks_only", {"BIGKRLS_KRY_NOEST": "1"}),
    ("cgs_twice_against_all", {"BIGKRLS_KRY_CGS": "2"}),
    ("with_sample_check", {"BIGKRLS_KRY_SAMPLE": "1"}),
    ("round5_behaviour", {"BIGKRLS_KRY_CGS": "2", "BIGKRLS_KRY_SAMPLE": "1"}),
]
child = r'''
import sys, os, time
sys.path.insert(0, %r)
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n, p, neig = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
X, y = synth(n, p, 104)
ctx = bk.Context(0)
best, eig = 1e9, 0.0
for rep in range(4):
    T = {}
    t0 = time.perf_counter(); out = bk.bigKRLS(y, X, ctx=ctx, Neig=neig, timings=T, noisy=False); ctx.sync(); dt = time.perf_counter() - t0
    if rep and dt < best: best, eig = dt, T["eigen"]
    co = np.asarray(out["coeffs"]).copy(); lam = out["lambda"]; keep = out["lastkeeper"]
    del out
print("%%-24s best %%.4f s (eigen %%.4f) lambda %%.12g kept %%d |c| %%.12g counters %%s" %% (sys.argv[1], best, eig, lam, keep, float(np.linalg.norm(co)), ctx.counters()), flush=True)
''' % ROOT
for rnd in range(rounds):
    for name, env in VARIANTS:
        e = dict(os.environ)
        e.update(env)
        subprocess.run([sys.executable, "-c", child, name, n, p, neig], env=e, check=False)
