"""Stress the dense eigensolver on hard spectra (development probe): eigenvalue error vs LAPACK,
orthogonality and residual of the computed eigenvectors."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bigkrls_amd as bk
from bigkrls_amd import ops
ctx = bk.Context(0)
rng = np.random.default_rng(0)

def check(name, A):
    n = A.shape[0]
    A = (A + A.T) / 2
    eo = ops.bEigen(ctx.from_numpy(A), n, -1.0)       # threshold < 0: keep every eigenvector
    vals, Q = eo.values, eo.vectors.to_numpy()
    ref = np.linalg.eigvalsh(A)[::-1]
    sc = max(abs(ref).max(), 1e-300)
    ev = np.max(np.abs(vals - ref)) / sc
    orth = np.max(np.abs(Q.T @ Q - np.eye(Q.shape[1])))
    res = np.max(np.abs(A @ Q - Q * vals[:Q.shape[1]])) / sc
    flag = "" if (ev < 1e-12 and orth < 1e-11 and res < 1e-11) else "   <-- CHECK"
    print(f"{name:34s} n={n:5d} nv={Q.shape[1]:5d}  eigval {ev:.1e}  orth {orth:.1e}  resid {res:.1e}{flag}")

def with_spectrum(d):
    n = len(d)
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Qm * d) @ Qm.T

for n in (300, 777, 1500):
    check("graded 1e0..1e-16", with_spectrum(np.logspace(0, -16, n)))
    check("two tight clusters", with_spectrum(np.r_[1 + 1e-13 * rng.standard_normal(n // 2), 2 + 1e-13 * rng.standard_normal(n - n // 2)]))
    check("all equal (identity)", np.eye(n))
    check("rank 3 + noise 1e-14", with_spectrum(np.r_[[5.0, 3.0, 1.0], 1e-14 * rng.random(n - 3)]))
    check("uniform random spectrum", with_spectrum(rng.random(n)))
    T = np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    check("Wilkinson W+", T)
    X = rng.standard_normal((n, 2))
    K = np.exp(-((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / 2.0)
    check("Gaussian kernel P=2 (singular)", K)
    check("negative definite graded", -with_spectrum(np.logspace(0, -10, n)))
