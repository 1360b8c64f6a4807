"""Worker for tests/test_dist_gloo.py: runs bigKRLS_dist on CPU under gloo with a
numpy test double for the local kernels, and checks every rank's outputs against the
single-process oracle. (Test infrastructure: this file may import the oracle.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import krls_oracle as orc  # noqa: E402
from bigkrls_amd import dist as bkdist  # noqa: E402


class NumpyBackend:
    """Same interface as bigkrls_amd.dist.HipBackend on CPU tensors ((ncol, nrow) layout)."""

    def from_numpy(self, a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            a = a[:, None]
        return torch.from_numpy(np.ascontiguousarray(a.T))

    def empty(self, nrow, ncol):
        return torch.zeros((ncol, nrow), dtype=torch.float64)

    def mm(self, ta, tb, A, B, alpha=1.0, beta=0.0, out=None):
        a, b = A.numpy().T, B.numpy().T                      # column-major views
        r = alpha * ((a.T if ta else a) @ (b.T if tb else b))
        if out is not None and beta != 0.0:
            r = r + beta * out.numpy().T
        res = torch.from_numpy(np.ascontiguousarray(r.T))
        if out is not None:
            out.copy_(res)
            return out
        return res

    def dense_eig_top(self, T, k):
        w, V = np.linalg.eigh(T)
        w, V = w[::-1], V[:, ::-1]
        return w.copy(), torch.from_numpy(np.ascontiguousarray(V[:, :k].T))

    def kernel_cols(self, X, sigma, c0, c1, out):
        Xn = X.numpy().T
        out.copy_(torch.from_numpy(orc.temp_kernel_literal(Xn[c0:c1], Xn, sigma)))
        for c in range(c0, c1):
            out[c - c0, c] = 1.0

    def eigen(self, K, neig, eigtrunc, rank=0, world=1):
        eo = orc.b_eigen(K.numpy().T, neig, eigtrunc)
        V = np.ascontiguousarray(eo.vectors.T)           # (lastkeeper, n)
        if world > 1:                                     # same contract as bigkrls_dev_eigen_part
            nv = V.shape[0]
            c0, c1 = nv * rank // world, nv * (rank + 1) // world
            V[:c0] = 0.0
            V[c1:] = 0.0
        return eo.values, eo.lastkeeper, torch.from_numpy(V), torch.from_numpy(eo.values.copy())[None, :]

    # ---- test double of the per-panel stage-1 entry points (bigkrls_dev_s1_*) and eigen_resume ----
    def s1_open(self, n):
        self._n = n
        self._W = np.zeros((n, n))
        self._V, self._T = {}, {}

    def s1_strip_from(self, A, lc, w, k, n, strip):
        strip.copy_(A[lc:lc + w, k:n])

    def s1_panel(self, n, k, strip):
        import scipy.linalg as sla
        b = bkdist.S1_B
        W = self._W
        W[k:, k:k + b] = strip.numpy().T
        P = W[k + b:, k:k + b]
        (qr, tau), _ = sla.qr(P, mode="raw")
        m = P.shape[0]
        ncol = min(b, m)
        V = np.tril(qr, -1)[:, :b]
        V[np.arange(ncol), np.arange(ncol)] = 1.0
        V[:, ncol:] = 0.0
        tau = np.concatenate([tau, np.zeros(b - tau.size)])
        T = np.zeros((b, b))                     # dlarft, forward columnwise
        for j in range(b):
            T[j, j] = tau[j]
            if j > 0:
                T[:j, j] = -tau[j] * (T[:j, :j] @ (V[:, :j].T @ V[:, j]))
        W[k + b:, k:k + b] = np.triu(qr)[:, :b] if m >= b else qr
        if m >= b:
            W[k + 2 * b:, k:k + b] = V[b:, :]
            W[k + b:k + 2 * b, k:k + b] = np.triu(qr[:b, :b]) + np.tril(V[:b, :b], -1)
        self._V[k], self._T[k] = V, T

    def s1_av(self, n, k, A, la0, ncols, Ysend):
        if ncols > 0:
            b = bkdist.S1_B
            Acols = A.numpy()[la0:la0 + ncols, k + b:n]            # (ncols, m): own columns, transposed
            Ysend[:, la0:la0 + ncols] = torch.from_numpy((Acols @ self._V[k]).T.copy())

    def s1_panel_begin(self, n, k, strip):          # (no streams on the host: the look-ahead call is the plain one)
        self.s1_panel(n, k, strip)

    def s1_thin(self, n, k, Y):
        V, T = self._V[k], self._T[k]
        Yt = Y.numpy().T @ T
        S = T.T @ (V.T @ Yt)
        self._Z = Yt - 0.5 * V @ S

    def s1_update_cols(self, n, k, A, la0, ncols, row0):
        b = bkdist.S1_B
        V, Z = self._V[k], self._Z
        if ncols > 0:
            An = A.numpy()
            An[la0:la0 + ncols, k + b:n] -= (V @ Z[row0:row0 + ncols].T + Z @ V[row0:row0 + ncols].T).T

    def s1_update(self, n, k, Y, A, la0, ncols, row0):
        self.s1_thin(n, k, Y)
        self.s1_update_cols(n, k, A, la0, ncols, row0)

    def s1_put(self, n, k, strip, ncols):
        self._W[k:, k:k + ncols] = strip.numpy().T

    def eigen_resume(self, n, neig, eigtrunc, rank, world):
        """Eigenpairs of the band matrix (LAPACK) back-transformed with the stored block reflectors;
        only this rank's slice of the kept columns, zeros elsewhere (the contract of the library)."""
        b = bkdist.S1_B
        W = self._W
        B = np.zeros((n, n))
        for d in range(b + 1):
            idx = np.arange(n - d)
            B[idx + d, idx] = W[idx + d, idx]
            B[idx, idx + d] = W[idx + d, idx]
        w, Z = np.linalg.eigh(B)
        w, Z = w[::-1][:neig].copy(), Z[:, ::-1][:, :neig].copy()
        for k in sorted(self._V, reverse=True):                     # Q = Q_0 Q_1 ... Q_last
            V, T = self._V[k], self._T[k]
            Z[k + b:] -= V @ (T @ (V.T @ Z[k + b:]))
        keep = np.nonzero(w >= eigtrunc * w[0])[0]
        nv = int(keep.max()) + 1
        Q = np.ascontiguousarray(Z[:, :nv].T)
        if world > 1:
            c0, c1 = nv * rank // world, nv * (rank + 1) // world
            Q[:c0] = 0.0
            Q[c1:] = 0.0
        return w, nv, torch.from_numpy(Q), torch.from_numpy(w.copy())[None, :]

    def qty_rows(self, Q, r0, r1, y):
        return torch.from_numpy((Q.numpy()[:, r0:r1] @ y.numpy().ravel()[r0:r1])[None, :].copy())

    def solveforc_rows(self, Q, r0, r1, d, a, lam, want_c):
        k = Q.shape[0]
        w = 1.0 / (d.numpy().ravel()[:k] + lam)
        Qr = Q.numpy()[:, r0:r1].T
        c = Qr @ (w * a.numpy().ravel())
        g = (Qr * Qr) @ w
        return float(np.sum((c / g) ** 2)), (torch.from_numpy(c[None, :].copy()) if want_c else None)

    def gemv_t(self, Kcols, x):
        return torch.from_numpy((Kcols.numpy() @ x.numpy().ravel())[None, :].copy())

    def vcov_cols(self, Q, wv, r0, r1):
        Qn = Q.numpy().T
        k = Qn.shape[1]
        V = (Qn * np.asarray(wv)[:k]) @ Qn[r0:r1].T
        return torch.from_numpy(np.ascontiguousarray(V.T))

    def deriv_rows(self, Kcols, r0, X, isbin, c, sigma):
        # evaluate the O(N^2) identities on the full problem and keep this block's rows
        nb, n = Kcols.shape
        raise_if = None
        Kfull = self._Kfull
        Xn = X.numpy().T
        D, S = _deriv_full(Xn, Kfull, c.numpy().ravel(), sigma)
        return torch.from_numpy(np.ascontiguousarray(D[r0:r0 + nb].T)), \
            torch.from_numpy(np.ascontiguousarray(S[r0:r0 + nb].T))

    def deriv_var(self, Q, wv, S, scale):
        Qn = Q.numpy().T
        k = Qn.shape[1]
        T = Qn.T @ S.numpy().T
        return np.asarray(scale) * np.sum(np.asarray(wv)[:k, None] * T * T, axis=0)

    def sync(self):
        pass


def _deriv_full(X, K, c, sigma):
    n, p = X.shape
    one = np.ones(n)
    K1, Kc = K @ one, K @ c
    D = np.empty((n, p)); S = np.empty((n, p))
    for j in range(p):
        x = X[:, j]
        if np.unique(x).size == 2:
            z0, z1 = x.min(), x.max()
            sd = 1.0 / (z1 - z0); phi = -((z1 - z0) ** 2) / sigma
            E, Ei = np.exp(phi), np.exp(-phi)
            b = (x == z1).astype(float); hi = b == 1
            Kb, Kbc = K @ b, K @ (b * c)
            S1 = np.where(hi, Kb, K1 - Kb); O1 = np.where(hi, K1 - Kb, Kb)
            Sc = np.where(hi, Kbc, Kc - Kbc); Oc = np.where(hi, Kc - Kbc, Kbc)
            D[:, j] = sd * np.where(hi, 1.0, -1.0) * ((1 - E) * Sc + (1 - Ei) * Oc)
            S[:, j] = np.where(hi, S1 + Ei * O1, E * S1 + O1) - np.where(hi, E * S1 + O1, S1 + Ei * O1)
        else:
            D[:, j] = (-2.0 / sigma) * (x * Kc - K @ (x * c))
            S[:, j] = x * K1 - K @ x
    return D, S


def main_krylov():
    """eigen_krylov_dist (sharded K B_j products + all-gather per step) against LAPACK on the full K."""
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, p, neig = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    X, y = orc.synth(n, p, 78)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    be = NumpyBackend()
    nb, parts = bkdist.partition(n, world)
    r0, r1 = parts[rank]
    Xd = be.from_numpy(Xs)
    Kcols = be.empty(n, r1 - r0)
    be.kernel_cols(Xd, float(p), r0, r1, Kcols)
    vals, lastkeeper, Q, dvals = bkdist.eigen_krylov_dist(be, torch, dist, Kcols, n, rank, world, neig, 0.001,
                                                          block=32, tol=1e-10)
    K = orc.gauss_kernel_literal(Xs, float(p))
    w, V = np.linalg.eigh(K)
    w, V = w[::-1], V[:, ::-1]
    assert np.max(np.abs(vals - w[:neig])) <= 1e-10 * w[0], (rank, np.max(np.abs(vals - w[:neig])) / w[0])
    assert lastkeeper == int(np.max(np.nonzero(w[:neig] >= 0.001 * w[0])[0])) + 1
    Qn = Q.numpy().T
    assert np.max(np.abs(Qn.T @ Qn - np.eye(lastkeeper))) < 1e-10
    ys = (y - y.mean()) / y.std(ddof=1)
    wt = 1.0 / (w[:lastkeeper] + 0.7)
    c_ref = V[:, :lastkeeper] @ (wt * (V[:, :lastkeeper].T @ ys))
    c_got = Qn @ (wt * (Qn.T @ ys))
    assert np.max(np.abs(c_ref - c_got)) <= 1e-8 * np.max(np.abs(c_ref))
    print("OK", flush=True)
    dist.destroy_process_group()


def main_dense():
    """eigen_dense_dist (stage 1 partitioned by column blocks: broadcast of the panel strip, all-gather of
    A22 V per panel, all-gather of the eigenvector column blocks) against LAPACK on the full K."""
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, p, neig = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    X, y = orc.synth(n, p, 79)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    be = NumpyBackend()
    nb, parts = bkdist.partition(n, world, bkdist.S1_B)
    r0, r1 = parts[rank]
    Kcols = be.empty(n, r1 - r0)
    if r1 > r0:
        be.kernel_cols(be.from_numpy(Xs), float(p), r0, r1, Kcols)
    vals, lastkeeper, Q, dvals = bkdist.eigen_dense_dist(be, torch, dist, Kcols.clone(), n, rank, world, nb, neig, 0.001)
    K = orc.gauss_kernel_literal(Xs, float(p))
    w, V = np.linalg.eigh(K)
    w = w[::-1]
    assert vals.shape == (neig,)
    assert np.max(np.abs(vals - w[:neig])) <= 1e-12 * w[0], (rank, np.max(np.abs(vals - w[:neig])) / w[0])
    assert lastkeeper == int(np.max(np.nonzero(w[:neig] >= 0.001 * w[0])[0])) + 1
    Qn = Q.numpy().T
    assert Qn.shape == (n, lastkeeper)
    assert np.max(np.abs(Qn.T @ Qn - np.eye(lastkeeper))) < 1e-11
    assert np.max(np.abs(K @ Qn - Qn * vals[:lastkeeper])) <= 1e-11 * w[0]
    print("OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if sys.argv[1] == "krylov":
        return main_krylov()
    if sys.argv[1] == "dense":
        return main_dense()
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, p = int(sys.argv[1]), int(sys.argv[2])
    binary = bool(int(sys.argv[3]))
    X, y = orc.synth(n, p, 77, binary_last=binary)
    be = NumpyBackend()
    # the test double needs the full K for its derivative identities: capture it from the all-gather
    orig_eigen = be.eigen
    def eigen_capture(K, neig, eigtrunc, rank=0, world=1):
        be._Kfull = K.numpy().T.copy()
        return orig_eigen(K, neig, eigtrunc, rank, world)
    be.eigen = eigen_capture
    # (the sharded eigensolvers never gather K: give the derivative double its full K directly)
    Xs_ = (X - X.mean(0)) / X.std(0, ddof=1)
    be._Kfull = orc.gauss_kernel_literal(Xs_, float(p))
    tr = []
    out = bkdist.bigKRLS_dist(y, X, backend=be, trace=tr)
    ref_tr = orc.LambdaTrace(0, 0)
    ref = orc.fit(y, X, literal=False, trace=ref_tr)

    def rel(a, b):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

    assert out["lastkeeper"] == ref["lastkeeper"]
    assert len(tr) == len(ref_tr.probes)
    assert abs(out["lambda"] - ref["lambda"]) <= 1e-10 * ref["lambda"]
    for k in ["coeffs", "yfitted", "derivatives", "var.avgderivatives", "avgderivatives"]:
        assert rel(out[k], ref[k]) < 1e-9, (rank, k, rel(out[k], ref[k]))
    for k in ["R2", "R2AME", "Looe", "Neffective", "sigmasq"]:
        assert abs(out[k] - ref[k]) <= 1e-9 * abs(ref[k]), (rank, k)
    r0, r1 = out["rows"]
    nb, parts = bkdist.partition(n, world, bkdist.S1_B if n >= bkdist.DENSE_DIST_MIN_N else 1)
    assert (r0, r1) == parts[rank]
    assert rel(out["K.cols"].numpy().T, ref["K"][:, r0:r1]) < 1e-12
    assert rel(out["vcov.est.c.cols"].numpy().T, ref["vcov.est.c"][:, r0:r1]) < 1e-9
    assert rel(out["vcov.est.fitted.cols"].numpy().T, ref["vcov.est.fitted"][:, r0:r1]) < 1e-9
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} OK")


if __name__ == "__main__":
    main()
