"""Synthetic inputs G(N, P, seed) for bench.py and examples:
X ~ N(0,1), beta_j = j/||(1..P)||, y = sin(X beta) + 0.25 eps, numpy PCG64 stream so
that both machines generate identical data without R."""
import numpy as np


def synth(n: int, p: int, seed: int, binary_last: bool = False):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    beta = np.arange(1, p + 1, dtype=np.float64)
    beta /= np.linalg.norm(beta)
    if binary_last:
        X[:, p - 1] = (X[:, p - 1] > 0.12345).astype(np.float64)
    y = np.sin(X @ beta) + 0.25 * rng.standard_normal(n)
    return X, y
