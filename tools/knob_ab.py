"""A/B timing of runtime knobs of the library in the same GPU session (development tool):
python tools/knob_ab.py N P "NAME=VAL,NAME2=VAL2" "..." -- one fresh process per variant and round (two rounds,
alternating), prints the best of 3 fits and its eigen phase. "-" is the variant without any knob."""
import sys, os, subprocess
n, p = sys.argv[1], sys.argv[2]
variants = sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, os, time
sys.path.insert(0, %r)
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n, p = int(sys.argv[1]), int(sys.argv[2])
X, y = synth(n, p, 103)
ctx = bk.Context(0, own_stream=bool(os.environ.get('BIGKRLS_S1_GRAPH') or os.environ.get('KNOB_AB_OWN_STREAM')))  # (a stream capture needs a stream of its own)
best = 1e9; lam = None
for rep in range(4):
    T = {}
    t0 = time.perf_counter(); out = bk.bigKRLS(y, X, ctx=ctx, timings=T); ctx.sync(); dt = time.perf_counter() - t0
    if rep and dt < best: best, eig, kb, dv = dt, T["eigen"], T["kernel"], T["derivatives"]
    lam = out["lambda"]; del out
print("%%-40s best %%.4f s (eigen %%.4f, kernel %%.3f ms, derivatives %%.3f ms) lambda %%.12g" %% (sys.argv[3], best, eig, 1e3 * kb, 1e3 * dv, lam), flush=True)
''' % root
for rnd in range(2):
    for v in variants:
        env = dict(os.environ)
        if v != "-":
            for kv in v.split(","):
                k, val = kv.split("=")
                env[k] = val
        subprocess.run([sys.executable, "-c", child, n, p, v], check=False, env=env)
