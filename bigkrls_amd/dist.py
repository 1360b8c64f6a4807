"""Row-block partitioned bigKRLS() over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" on CPU for the tests).

Partition: rank r owns rows [r0, r1) of K, Q and V. K is
symmetric, so the row block is stored as the contiguous column block K[:, r0:r1].

  phase            local work                         exchange
  kernel build     K[:, r0:r1] (fp64 MFMA + exp)       all-gather of the column blocks
  eigen            reduction + divide&conquer          all-reduce (sum) of Q: every rank
                   replicated (not yet distributed     back-transforms its own slice of the
                   -- section 8(f))                     eigenvector columns, zeros elsewhere
  lambda search    Q[r0:r1,:]: a_r = Q_r' y_r          all-reduce a (K doubles) once,
                   per probe c_r, g_r, Le_r             all-reduce of one scalar per probe
  coefficients     c_r                                  all-gather c (N doubles)
  fitted values    yhat_r = K[:, r0:r1]' c              all-gather (N doubles)
  V, V_yhat        column blocks Q W Q[r0:r1,:]'        none (kept sharded)
  derivatives      D_r, S_r from K[:, r0:r1]            all-gather D, S (N x P')
  var(avg deriv)   replicated skinny GEMM Q'S           none

The numeric kernels sit behind a small backend object so that the orchestration
and every collective can be exercised on CPU (gloo, world_size 2) with a test
double; `HipBackend` is the product path and has no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import time
from typing import Dict, Optional

import numpy as np

from . import _lib, ops
from .api import BigKRLS, _cor, _sd, _var
from .device import Context, DeviceMatrix


def partition(n: int, world: int):
    """Equal blocks of nb = ceil(n/world) rows; the last ranks may be short or empty."""
    nb = (n + world - 1) // world
    return nb, [(min(r * nb, n), min((r + 1) * nb, n)) for r in range(world)]


class HipBackend:
    """Local compute on one MI355X through the C ABI. Tensors are torch float64 CUDA
    tensors of shape (ncol, nrow) == column-major (nrow x ncol)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.torch = ctx.torch
        self.device = ctx.device

    def from_numpy(self, a):
        return self.ctx.from_numpy(a).t

    def empty(self, nrow, ncol):
        return self.ctx.empty(nrow, ncol).t

    def _dm(self, t):
        return DeviceMatrix(self.ctx, t)

    def kernel_cols(self, X, sigma, c0, c1, out):
        n, p = X.shape[1], X.shape[0]
        Xd = self._dm(X)
        _lib.call("bigkrls_dev_kernel_block", self.ctx.handle, Xd.ptr, n, n, Xd.col_ptr(0, c0),
                  c1 - c0, n, p, float(sigma), C.c_void_p(out.data_ptr()), n, c0)

    def eigen(self, K, neig, eigtrunc, rank=0, world=1):
        """Replicated reduction + divide & conquer; this rank's slice of the eigenvector columns
        back-transformed, zeros elsewhere (the caller all-reduces Q)."""
        eo = ops.bEigen(self._dm(K), neig, eigtrunc, part=(rank, world) if world > 1 else None)
        return eo.values, eo.lastkeeper, eo.vectors.t, eo.values_dev.t

    def qty_rows(self, Q, r0, r1, y):
        k, n = Q.shape
        a = self.torch.zeros((1, k), dtype=self.torch.float64, device=self.device)
        if r1 > r0:
            _lib.call("bigkrls_dev_qty", self.ctx.handle, C.c_void_p(Q.data_ptr() + 8 * r0), r1 - r0,
                      k, n, C.c_void_p(y.data_ptr() + 8 * r0), C.c_void_p(a.data_ptr()))
        return a

    def solveforc_rows(self, Q, r0, r1, d, a, lam, want_c):
        k, n = Q.shape
        if r1 <= r0:
            return 0.0, (self.torch.zeros((1, 0), dtype=self.torch.float64, device=self.device) if want_c else None)
        c = self.torch.empty((1, r1 - r0), dtype=self.torch.float64, device=self.device) if want_c else None
        le = C.c_double()
        _lib.call("bigkrls_dev_solveforc", self.ctx.handle, C.c_void_p(Q.data_ptr() + 8 * r0), r1 - r0, k, n,
                  C.c_void_p(d.data_ptr()), C.c_void_p(a.data_ptr()), float(lam),
                  C.c_void_p(c.data_ptr()) if want_c else None, C.byref(le))
        return float(le.value), c

    def gemv_t(self, Kcols, x):
        nb, n = Kcols.shape
        out = self.torch.empty((1, nb), dtype=self.torch.float64, device=self.device)
        if nb > 0:
            _lib.call("bigkrls_dev_gemv", self.ctx.handle, 1, n, nb, 1.0, C.c_void_p(Kcols.data_ptr()), n,
                      C.c_void_p(x.data_ptr()), 0.0, C.c_void_p(out.data_ptr()))
        return out

    def vcov_cols(self, Q, wv, r0, r1):
        """V[:, r0:r1] = (Q diag(wv)) Q[r0:r1, :]'."""
        k, n = Q.shape
        m = ops.bMultDiag(self._dm(Q), wv)
        out = self.torch.empty((r1 - r0, n), dtype=self.torch.float64, device=self.device)
        if r1 > r0:
            _lib.call("bigkrls_dev_gemm", self.ctx.handle, 0, 1, n, r1 - r0, k, 1.0, m.ptr, n,
                      C.c_void_p(Q.data_ptr() + 8 * r0), n, 0.0, C.c_void_p(out.data_ptr()), n)
        return out

    def deriv_rows(self, Kcols, r0, X, isbin, c, sigma):
        nb, n = Kcols.shape
        p = X.shape[0]
        D = self.torch.empty((p, nb), dtype=self.torch.float64, device=self.device)
        S = self.torch.empty((p, nb), dtype=self.torch.float64, device=self.device)
        if nb > 0:
            isb = np.ascontiguousarray(np.asarray(isbin).astype(np.int32))
            _lib.call("bigkrls_dev_deriv_rows", self.ctx.handle, C.c_void_p(Kcols.data_ptr()), n, nb, n, r0,
                      C.c_void_p(X.data_ptr()), p, n, C.c_void_p(isb.ctypes.data), C.c_void_p(c.data_ptr()),
                      float(sigma), C.c_void_p(D.data_ptr()), nb, C.c_void_p(S.data_ptr()), nb)
        return D, S

    def deriv_var(self, Q, wv, S, scale):
        k, n = Q.shape
        p = S.shape[0]
        dwv = self.ctx.from_numpy(np.asarray(wv, dtype=np.float64)[:k])
        sc = np.ascontiguousarray(scale, dtype=np.float64)
        var = np.empty(p)
        _lib.call("bigkrls_dev_deriv_var", self.ctx.handle, C.c_void_p(Q.data_ptr()), n, k, n, dwv.ptr,
                  C.c_void_p(S.data_ptr()), p, n, C.c_void_p(sc.ctypes.data), C.c_void_p(var.ctypes.data))
        return var

    def sync(self):
        self.ctx.sync()


def _all_gather_cols(torch, dist, local, nb, n_total, world):
    """local: (nb_r, n) rows block (padded to nb) -> (n_total, n)."""
    if local.shape[0] < nb:
        pad = torch.zeros((nb - local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    full = torch.empty((nb * world, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous())
    return full[:n_total]


def _all_gather_vec(torch, dist, local, nb, n_total, world):
    """local: (rows, nb_r) slices of per-row vectors/matrices -> (rows, n_total)."""
    rows = local.shape[0]
    if local.shape[1] < nb:
        pad = torch.zeros((rows, nb - local.shape[1]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=1)
    full = torch.empty((world, rows, nb), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous().unsqueeze(0))
    return full.permute(1, 0, 2).reshape(rows, world * nb)[:, :n_total].contiguous()


def bigKRLS_dist(y, X, sigma=None, derivative=True, which_derivatives=None, Neig=None, eigtrunc=None,
                 lambda_=None, L=None, U=None, ctx: Optional[Context] = None, backend=None,
                 timings: Optional[Dict[str, float]] = None, trace=None, keep_outputs=True) -> BigKRLS:
    """bigKRLS() with the kernel build, lambda search, coefficient, variance and
    marginal-effects passes partitioned over the ranks of the default process group.
    Every rank returns the same small outputs; N x N outputs stay sharded
    (`K.cols`, `vcov.est.c.cols`, `vcov.est.fitted.cols` hold this rank's column block)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if backend is None:
        backend = HipBackend(ctx or Context())
    Xh = np.array(X, dtype=np.float64)
    yh = np.array(y, dtype=np.float64).ravel()
    n, p = Xh.shape
    if Xh.std(axis=0, ddof=1).min() == 0:
        raise ValueError("The following columns in X are constant and must be removed")
    if n != yh.shape[0]:
        raise ValueError("nrow(X) not equal to number of elements in y.")
    Neig = min(n, int(Neig)) if Neig is not None else n
    if eigtrunc is None:
        eigtrunc = 0.001 if n > 3000 else 0.0
    sigma = float(p) if sigma is None else float(sigma)
    X_init_sd = Xh.std(axis=0, ddof=1)
    y_init_sd, y_init_mean = _sd(yh), float(yh.mean())
    Xs = (Xh - Xh.mean(axis=0)) / X_init_sd
    ys = (yh - y_init_mean) / y_init_sd
    nb, parts = partition(n, world)
    r0, r1 = parts[rank]
    T = timings if timings is not None else {}
    t_last = [time.perf_counter()]

    def mark(name):
        backend.sync()
        now = time.perf_counter()
        T[name] = now - t_last[0]
        t_last[0] = now

    t_start = time.perf_counter()
    Xd = backend.from_numpy(Xs)
    yd = backend.from_numpy(ys)
    # ---- step 1: kernel, column blocks + all-gather -------------------------------
    Kpad = backend.empty(n, nb * world)          # (nb*world, n): column c at Kpad[c]
    Kloc = Kpad[rank * nb: rank * nb + (r1 - r0)]
    if r1 > r0:
        backend.kernel_cols(Xd, sigma, r0, r1, Kloc)
    mark("kernel")
    if world > 1:
        dist.all_gather_into_tensor(Kpad, Kpad[rank * nb:(rank + 1) * nb].clone())
    K = Kpad[:n]
    Kcols = K[r0:r1]
    mark("kernel_allgather")
    # ---- step 2: eigen (replicated) -------------------------------------------------
    vals, lastkeeper, Q, dvals = backend.eigen(K, Neig, eigtrunc, rank, world)
    if world > 1:
        # each rank back-transformed its own eigenvector columns (zeros elsewhere): sum = Q.
        # This is the RCCL exchange north_star names for the eigenvector back-transform.
        dist.all_reduce(Q, op=dist.ReduceOp.SUM)
    mark("eigen")
    # ---- step 3: lambda search on row blocks of Q ------------------------------------
    a = backend.qty_rows(Q, r0, r1, yd)
    if world > 1:
        dist.all_reduce(a)

    def loo(lam):
        le, _ = backend.solveforc_rows(Q, r0, r1, dvals, a, lam, False)
        if world > 1:
            t = torch.tensor([le], dtype=torch.float64, device=a.device)
            dist.all_reduce(t)
            le = float(t.item())
        return le

    if lambda_ is None:
        class _E:  # minimal Eigenobject for bLambdaSearch's bounds
            values = vals
        class _Y:
            nrow = n
        lambda_ = ops.bLambdaSearch(L=L, U=U, y=_Y, Eigenobject=_E, trace=trace, loo=loo)
    mark("lambda")
    # ---- step 4: coefficients, fitted values, variances -------------------------------
    le_loc, c_loc = backend.solveforc_rows(Q, r0, r1, dvals, a, lambda_, True)
    if world > 1:
        t = torch.tensor([le_loc], dtype=torch.float64, device=a.device)
        dist.all_reduce(t)
        Le = float(t.item())
        c_full = _all_gather_vec(torch, dist, c_loc, nb, n, world)
    else:
        Le, c_full = le_loc, c_loc
    yhat_loc = backend.gemv_t(Kcols, c_full)
    yhat_full = _all_gather_vec(torch, dist, yhat_loc, nb, n, world) if world > 1 else yhat_loc
    coeffs = c_full.cpu().numpy().ravel()
    yfitted = yhat_full.cpu().numpy().ravel()
    mark("coeffs")
    resid = ys - yfitted
    sigmasq = float(resid @ resid) / n
    wv = sigmasq * (vals[:lastkeeper] + lambda_) ** -2.0
    Vcols = backend.vcov_cols(Q, wv, r0, r1)
    Vyhat_cols = backend.vcov_cols(Q, wv * vals[:lastkeeper] ** 2, r0, r1)
    mark("vcov")
    w = BigKRLS()
    # ---- step 5: marginal effects -------------------------------------------------------
    if derivative:
        cols = list(range(p)) if which_derivatives is None else [int(i) - 1 for i in which_derivatives]
        Xe_h = Xs[:, cols]
        Xe = Xd if which_derivatives is None else backend.from_numpy(Xe_h)
        isb = ops.binary_columns(Xe_h)
        D_loc, S_loc = backend.deriv_rows(Kcols, r0, Xe, isb, c_full, sigma)
        if world > 1:
            D_full = _all_gather_vec(torch, dist, D_loc, nb, n, world)
            S_full = _all_gather_vec(torch, dist, S_loc, nb, n, world)
        else:
            D_full, S_full = D_loc, S_loc
        var = backend.deriv_var(Q, wv, S_full, ops.deriv_scales(Xe_h, isb, sigma))
        derivmat = D_full.cpu().numpy().T.copy()
        mark("derivatives")
        w["derivatives.std"] = derivmat.copy()
        w["var.avgderivatives.std"] = var.copy()
        w["R2AME"] = _cor(yh, Xe_h @ derivmat.mean(axis=0)) ** 2
        derivmat = y_init_sd * derivmat
        for i in range(derivmat.shape[1]):
            derivmat[:, i] /= X_init_sd[i]
        w["avgderivatives"] = derivmat.mean(axis=0)[None, :]
        w["var.avgderivatives"] = ((y_init_sd / X_init_sd[cols]) ** 2 * var)[None, :]
        w["derivatives"] = derivmat
    w["K.eigenvalues"] = vals
    w["lastkeeper"] = lastkeeper
    w["Neffective"] = n - float(np.sum(vals / (vals + lambda_)))
    w["coeffs"] = coeffs
    w["y"] = yh
    w["X"] = Xh
    w["sigma"] = sigma
    w["lambda"] = float(lambda_)
    w["yfitted.std"] = yfitted.copy()
    yf = yfitted * y_init_sd + y_init_mean
    w["yfitted"] = yf
    w["R2"] = 1 - (_var(yh - yf) / (y_init_sd ** 2))
    w["Le"] = Le
    w["Looe"] = Le * y_init_sd
    w["sigmasq"] = sigmasq
    w["rows"] = (r0, r1)
    if keep_outputs:
        w["K.cols"] = Kcols
        w["vcov.est.c.cols"] = Vcols * (y_init_sd ** 2)
        w["vcov.est.fitted.cols"] = Vyhat_cols * (y_init_sd ** 2)
    backend.sync()
    T["wall"] = time.perf_counter() - t_start
    return w
