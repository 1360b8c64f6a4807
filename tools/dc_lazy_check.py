"""Divide & conquer with the top levels kept factored vs every level formed (BIGKRLS_DC=explicit):
eigenpair residuals, orthogonality and timing (development probe).  python tools/dc_lazy_check.py N P"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 3:
    import numpy as np
    import bigkrls_amd as bk
    from bigkrls_amd import ops
    from bigkrls_amd.synth import synth
    n, p = int(sys.argv[1]), int(sys.argv[2])
    ctx = bk.Context(0)
    X, y = synth(n, p, 7)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
    for neig, trunc in ((None, 0.001), (n // 16, 0.0), (None, 1e-9)):
        eo = ops.bEigen(K, neig, trunc); ctx.sync()
        t0 = time.perf_counter(); eo = ops.bEigen(K, neig, trunc); ctx.sync(); dt = time.perf_counter() - t0
        Q = eo.vectors
        KQ = ops.gemm(False, False, K, Q).to_numpy()
        Qh = Q.to_numpy()
        lam = eo.values[: eo.lastkeeper]
        res = np.abs(KQ - Qh * lam).max() / lam[0]
        orth = np.abs(Qh.T @ Qh - np.eye(eo.lastkeeper)).max()
        print(f"  {sys.argv[3]:9s} N={n} Neig={neig} trunc={trunc}: kept {eo.lastkeeper:5d}  {dt*1e3:7.1f} ms  "
              f"max|K q - lam q|/lam1 = {res:.2e}  max|Q'Q - I| = {orth:.2e}  lam[:2]={lam[:2]} lam[-1]={lam[-1]:.6e}")
else:
    for mode in ("factored", "explicit"):
        env = dict(os.environ)
        if mode == "explicit":
            env["BIGKRLS_DC"] = "explicit"
        subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1], sys.argv[2], mode], env=env)
