"""Stage-2 back-transform: compact-WY MFMA kernel vs the reflector-by-reflector kernel (BIGKRLS_BT2=seq),
all eigenvectors and a truncated set (development probe).  python tools/bt2_ab.py N"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 2:
    import numpy as np
    import bigkrls_amd as bk
    from bigkrls_amd import ops
    from bigkrls_amd.synth import synth
    n = int(sys.argv[1])
    ctx = bk.Context(0)
    X, y = synth(n, 8, 7)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    K = ops.bGaussKernel(ctx.from_numpy(Xs), 8.0)
    for neig, trunc in ((None, 0.001), (None, -1.0)):
        eo = ops.bEigen(K, neig, trunc); ctx.sync()
        t0 = time.perf_counter(); eo = ops.bEigen(K, neig, trunc); ctx.sync(); dt = time.perf_counter() - t0
        Q = eo.vectors
        lam = eo.values[: eo.lastkeeper]
        R = ops.gemm(False, False, K, Q).to_numpy() - Q.to_numpy() * lam
        G = ops.gemm(True, False, Q, Q).to_numpy()
        print(f"  {sys.argv[2]:4s} N={n} kept {eo.lastkeeper:6d}: {dt*1e3:8.1f} ms  max|K q - lam q|/lam1 = {np.abs(R).max()/lam[0]:.2e}"
              f"  max|Q'Q - I| = {np.abs(G - np.eye(eo.lastkeeper)).max():.2e}")
else:
    for mode in ("wy", "seq"):
        env = dict(os.environ)
        if mode == "seq":
            env["BIGKRLS_BT2"] = "seq"
        subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1], mode], env=env)
