"""Fits on the boundary shapes of the block-Lanczos path against the same fits with the dense decomposition
(development tool, round 6): N = 16384 exactly / odd N, Neig = N/8, Neig = 1, 2, 127, 129, narrow and wide kernels,
repeated observations, binary columns.   python tools/fit_kry_sweep.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bigkrls_amd as bk
import bigkrls_amd._lib as L
from bigkrls_amd.synth import synth

ctx = bk.Context(0)


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


CASES = [
    ("N = 16384, Neig = N/8", 16384, 5, dict(Neig=2048)),
    ("odd N, Neig = 1", 16385, 7, dict(Neig=1)),
    ("Neig = 2", 16390, 3, dict(Neig=2)),
    ("Neig = 127", 17001, 12, dict(Neig=127)),
    ("Neig = 129", 17001, 12, dict(Neig=129)),
    ("narrow kernel (sigma = P/4)", 20000, 6, dict(Neig=400, sigma=1.5)),
    ("wide kernel (sigma = 50 P)", 20000, 6, dict(Neig=100, sigma=300.0)),
    ("eigtrunc = 0 (all Neig pairs kept)", 18000, 9, dict(Neig=300, eigtrunc=0.0)),
    ("repeated observations", 18000, 4, dict(Neig=64), "dup"),
    ("binary columns", 18000, 5, dict(Neig=200), "bin"),
    ("which.derivatives subset", 18000, 8, dict(Neig=150, which_derivatives=[2, 5])),
]
bad = 0
for case in CASES:
    name, n, p, kw = case[:4]
    X, y = synth(n, p, 11 + n % 13)
    if len(case) > 4 and case[4] == "dup":
        X[1000:1500] = X[:500]
        y[1000:1500] = y[:500]
    if len(case) > 4 and case[4] == "bin":
        X[:, 1] = (X[:, 1] > 0.3).astype(float)
        X[:, 3] = (X[:, 3] > -0.2).astype(float)
    res = {}
    for mode in ("default", "dense"):
        if mode == "dense":
            os.environ["BIGKRLS_EIGK"] = "dense"
        else:
            os.environ.pop("BIGKRLS_EIGK", None)
        T = {}
        try:
            out = bk.bigKRLS(y, X, ctx=ctx, noisy=False, timings=T, **kw)
            res[mode] = dict(lam=out["lambda"], keep=out["lastkeeper"], c=np.array(out["coeffs"]), yh=np.array(out["yfitted"]),
                             d=np.array(out["derivatives"]), avg=np.array(out["avgderivatives"]), var=np.array(out["var.avgderivatives"]),
                             ev=np.array(out["K.eigenvalues"]), t=T.get("eigen", 0.0))
            del out
        except L.BigKRLSError as e:
            res[mode] = str(e)
    os.environ.pop("BIGKRLS_EIGK", None)
    a, b = res["default"], res["dense"]
    if isinstance(a, str) or isinstance(b, str):
        same = isinstance(a, str) and isinstance(b, str)
        print("%-40s default: %s | dense: %s %s" % (name, a if isinstance(a, str) else "ok", b if isinstance(b, str) else "ok", "" if same else "  <-- DIFFERENT OUTCOME"), flush=True)
        bad += 0 if same else 1
        continue
    errs = dict(lam=rel(a["lam"], b["lam"]), c=rel(a["c"], b["c"]), yh=rel(a["yh"], b["yh"]), d=rel(a["d"], b["d"]), avg=rel(a["avg"], b["avg"]),
                var=rel(a["var"], b["var"]), ev=rel(a["ev"], b["ev"]))
    ok = a["keep"] == b["keep"] and all(v < 1e-6 for v in errs.values())
    bad += 0 if ok else 1
    print("%-40s kept %d / %d  eigen %.3f s vs %.3f s  %s%s" % (name, a["keep"], b["keep"], a["t"], b["t"],
          " ".join("%s %.1e" % kv for kv in errs.items()), "" if ok else "   <-- MISMATCH"), flush=True)
print("counters", ctx.counters(), "mismatches", bad)
