"""Row-block partitioned bigKRLS() over the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" on CPU for the tests).

Partition: rank r owns rows [r0, r1) of K, Q and V. K is
symmetric, so the row block is stored as the contiguous column block K[:, r0:r1].

  phase            local work                         exchange
  kernel build     K[:, r0:r1] (fp64 MFMA + exp)       all-gather of the column blocks
  eigen (Neig<<N)  block Lanczos: rows r0:r1 of K B_j   all-gather of an N x 128 block per step;
                   (K never leaves its row blocks)      orthogonalisation / Ritz problem replicated
  eigen (dense)    reduction + divide&conquer          all-reduce (sum) of Q: every rank
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
import os
import time
from typing import Dict, Optional

import numpy as np

from . import _lib, ops
from .api import BigKRLS, _cor, _sd, _var
from .device import Context, DeviceMatrix


def _host(backend, t):
    """numpy copy (same shape, C order) of a backend tensor; HBM tensors come down through the
    context's pinned staging buffer (device.py), CPU tensors (the gloo tests) are viewed."""
    if t.is_cuda:
        return backend.ctx.download(t.contiguous())
    return t.contiguous().numpy()


S1_B = 64          # panel width of the dense reduction (S2_B in csrc/eigen_2stage.inc)
DENSE_DIST_MIN_N = 257   # below, the dense eigensolver stays replicated (one-stage path in the library)


def _torch_dist():
    """(torch, torch.distributed) as bigKRLS_dist uses them. tools/dist_world2_one_gpu.py replaces this
    with an object that stages device tensors through the host, to drive WORLD_SIZE > 1 on one GPU."""
    import torch
    import torch.distributed as dist
    return torch, dist


def partition(n: int, world: int, align: int = 1):
    """Equal blocks of nb = ceil(n/world) rows, rounded up to a multiple of `align`; the last ranks
    may be short or empty."""
    nb = (n + world - 1) // world
    nb = (nb + align - 1) // align * align
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

    def mm(self, ta, tb, A, B, alpha=1.0, beta=0.0, out=None):
        """out = alpha op(A) op(B) + beta out on column-major matrices held as (ncol, nrow) tensors."""
        # a (ncol, nrow) tensor is the column-major nrow x ncol matrix: rows = shape[1], cols = shape[0]
        am, ak = (A.shape[0], A.shape[1]) if ta else (A.shape[1], A.shape[0])
        bk2, bn = (B.shape[1], B.shape[0]) if not tb else (B.shape[0], B.shape[1])
        assert ak == bk2, (A.shape, B.shape, ta, tb)
        assert A.is_contiguous() and B.is_contiguous()
        if out is None:
            out = self.torch.empty((bn, am), dtype=self.torch.float64, device=self.device)
        assert out.shape == (bn, am) and out.is_contiguous()
        _lib.call("bigkrls_dev_gemm", self.ctx.handle, int(ta), int(tb), am, bn, ak, float(alpha),
                  C.c_void_p(A.data_ptr()), A.shape[1], C.c_void_p(B.data_ptr()), B.shape[1], float(beta),
                  C.c_void_p(out.data_ptr()), out.shape[1])
        return out

    # ---- dense eigensolver with stage 1 partitioned by column blocks -------------
    def s1_open(self, n):
        _lib.call("bigkrls_dev_s1_open", self.ctx.handle, n)

    def s1_strip_from(self, A, lc, w, k, n, strip):
        """strip ((w, n-k) tensor) = rows k..n of the local columns lc..lc+w of A ((ncl, n) tensor)."""
        _lib.call("bigkrls_dev_copy_matrix", self.ctx.handle, C.c_void_p(A.data_ptr() + 8 * (lc * n + k)),
                  n - k, w, n, C.c_void_p(strip.data_ptr()), n - k)

    def s1_panel(self, n, k, strip):
        _lib.call("bigkrls_dev_s1_panel", self.ctx.handle, n, k, C.c_void_p(strip.data_ptr()))

    def s1_av(self, n, k, A, la0, ncols, Ysend):
        """Rows la0.. of Ysend ((b, nb) tensor == nb x b column-major) = A22[:, own]' V."""
        if ncols > 0:
            _lib.call("bigkrls_dev_s1_av", self.ctx.handle, n, k,
                      C.c_void_p(A.data_ptr() + 8 * (la0 * n + k + S1_B)), n, ncols,
                      C.c_void_p(Ysend.data_ptr() + 8 * la0), Ysend.shape[1])

    def s1_update(self, n, k, Y, A, la0, ncols, row0):
        _lib.call("bigkrls_dev_s1_update", self.ctx.handle, n, k, C.c_void_p(Y.data_ptr()),
                  C.c_void_p(A.data_ptr() + 8 * (la0 * n + k + S1_B)) if ncols > 0 else None, n, ncols, row0)

    def s1_put(self, n, k, strip, ncols):
        _lib.call("bigkrls_dev_s1_put", self.ctx.handle, n, k, C.c_void_p(strip.data_ptr()), ncols)

    def s1_panel_begin(self, n, k, strip):
        """s1_panel on the look-ahead stream (returns at once; the next s1_av / s1_thin waits for it)."""
        _lib.call("bigkrls_dev_s1_panel_begin", self.ctx.handle, n, k, C.c_void_p(strip.data_ptr()))

    def s1_thin(self, n, k, Y):
        _lib.call("bigkrls_dev_s1_thin", self.ctx.handle, n, k, C.c_void_p(Y.data_ptr()))

    def s1_update_cols(self, n, k, A, la0, ncols, row0):
        if ncols > 0:
            _lib.call("bigkrls_dev_s1_update_cols", self.ctx.handle, n, k,
                      C.c_void_p(A.data_ptr() + 8 * (la0 * n + k + S1_B)), n, ncols, row0)

    def eigen_resume(self, n, neig, eigtrunc, rank, world):
        """Stage 2, divide & conquer and this rank's slice of the back-transform. Returns (values
        host, lastkeeper, Q tensor (lastkeeper, n) whose rows outside the slice are zero, values tensor)."""
        vals = self.ctx.empty(neig, 1)
        vecs = self.ctx.empty(n, neig)
        nv = C.c_int64(0)
        _lib.call("bigkrls_dev_eigen_resume", self.ctx.handle, n, neig, vals.ptr, neig, float(eigtrunc), vecs.ptr, n,
                  C.byref(nv), int(rank), int(world))
        k = int(nv.value)
        return vals.to_numpy().ravel(), k, vecs.t[:k], vals.t

    def random_block(self, b, n, seed):
        """(b, n) tensor of uniform values in [-0.5, 0.5) that depend on the position and the seed only."""
        W = self.torch.empty((b, n), dtype=self.torch.float64, device=self.device)
        _lib.call("bigkrls_dev_fill_random", self.ctx.handle, C.c_void_p(W.data_ptr()), b * n, int(seed) & 0xFFFFFFFF)
        return W

    def cholqr2(self, W, R_dev=None):
        """Orthonormalise the rows of the (b, n) tensor W (the columns of the n x b block) on the device, in
        place: (W, R host (b, b) upper triangular with W_in = W_out R, ok). R_dev: optional (b, b) device tensor
        that receives R in the library's column-major layout."""
        b, n = W.shape
        assert W.is_contiguous()
        tmp = self.torch.empty_like(W)
        R = np.empty((b, b), dtype=np.float64, order="F")
        brk = C.c_int32(0)
        _lib.call("bigkrls_dev_cholqr2", self.ctx.handle, C.c_void_p(W.data_ptr()), C.c_void_p(tmp.data_ptr()), n, b,
                  R.ctypes.data_as(C.c_void_p), C.byref(brk),
                  C.c_void_p(R_dev.data_ptr()) if R_dev is not None else None)
        return W, (None if brk.value else np.ascontiguousarray(R)), brk.value == 0

    def projected_eig_top(self, A_blocks, beta_blocks, steps, k):
        """Eigenvalues (descending, host) and top-k eigenvectors (tensor (k, m)) of the block-tridiagonal projected
        matrix assembled on the device from the (maxsteps, b, b) block tensors."""
        b = A_blocks.shape[1]
        m = steps * b
        Tm = self.ctx.empty(m, m)
        _lib.call("bigkrls_dev_lanczos_projected", self.ctx.handle, C.c_void_p(A_blocks.data_ptr()),
                  C.c_void_p(beta_blocks.data_ptr()), steps, b, Tm.ptr)
        eo = ops.bEigen(Tm, m, -1.0)
        return eo.values, eo.vectors.t[:k]

    def dense_eig_top(self, T, k):
        """All eigenvalues (descending, host) and the top-k eigenvectors (tensor (k, m)) of the dense
        symmetric T given as a host array."""
        eo = ops.bEigen(self.ctx.from_numpy(np.asfortranarray(T)), T.shape[0], -1.0)
        return eo.values, eo.vectors.t[:k]

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


def _chol_upper_and_inverse(G):
    """Upper Cholesky factor R (G = R'R) and R^-1 of a small SPD matrix with plain numpy vector
    operations. (A threaded LAPACK call on a 128 x 128 matrix costs ~10 ms on a 128-core host,
    which would dominate a Lanczos step.) Returns (None, None) on breakdown."""
    b = G.shape[0]
    R = np.zeros((b, b))
    A = G.copy()
    for j in range(b):
        d = A[j, j]
        if not (d > 0.0) or not np.isfinite(d):
            return None, None
        rjj = math.sqrt(d)
        R[j, j] = rjj
        if j + 1 < b:
            row = A[j, j + 1:] / rjj
            R[j, j + 1:] = row
            A[j + 1:, j + 1:] -= np.outer(row, row)
    Rinv = np.zeros((b, b))
    eye = np.eye(b)
    for i in range(b - 1, -1, -1):             # row i of R^-1 from the rows below it (R Rinv = I)
        Rinv[i, :] = (eye[i, :] - R[i, i + 1:] @ Rinv[i + 1:, :]) / R[i, i]
    return R, Rinv


def eigen_krylov_dist(backend, torch, dist, Kcols, n, rank, world, neig, eigtrunc, block=128, tol=1e-10,
                      seed=20240229):
    """Top-`neig` eigenpairs of K for Neig << N without ever forming K on one GPU (SURVEY 8(e),
    "Eigen, partial"): block Lanczos with full re-orthogonalisation, the K B_j products sharded by
    row block -- rank r multiplies its own rows K[r0:r1, :] (stored as the column block
    Kcols = K[:, r0:r1]) and one all-gather of an N x 128 block per step assembles K B_j; the
    orthogonalisation, the Cholesky QR and the projected eigenproblem are replicated (identical,
    deterministic arithmetic on every rank). Same algorithm and stopping rule as the single-GPU
    `eigen_krylov` in csrc/eigen.hip. Returns (values[neig] host, lastkeeper, Q tensor (lastkeeper, n),
    values tensor (1, neig))."""
    b = int(block)
    nb, parts = partition(n, world)
    r0, r1 = parts[rank]
    maxdim = min(n // 2 // b * b, max(16 * neig, 4096) // b * b)
    maxsteps = maxdim // b

    def k_times(Bj):                       # (b, n) tensor == n x b column-major  ->  K Bj, same layout
        cols = Bj.shape[0]
        Wloc = torch.zeros((cols, nb), dtype=torch.float64, device=Bj.device)
        if r1 > r0:
            Wloc[:, : r1 - r0] = backend.mm(True, False, Kcols, Bj)     # (K[:, r0:r1])' Bj  = rows r0:r1 of K Bj
        if not dist.is_initialized():
            return Wloc[:, :n].contiguous()
        full = torch.empty((world * cols, nb), dtype=torch.float64, device=Bj.device)
        dist.all_gather_into_tensor(full, Wloc.contiguous())
        return full.view(world, cols, nb).permute(1, 0, 2).reshape(cols, world * nb)[:, :n].contiguous()

    def cholqr2(W, R_dev=None):
        """Orthonormalise the columns of W (tensor (b, n)); returns (Q, R host upper, ok). The HIP backend does it
        on the device (the library's Gram product + register-tile Cholesky / inverse); a backend without
        `cholqr2` (the numpy double of the gloo tests) goes through its products and a host Cholesky."""
        if hasattr(backend, "cholqr2"):
            return backend.cholqr2(W.contiguous(), R_dev)
        Racc = None
        for _ in range(2):
            G = _host(backend, backend.mm(True, False, W, W)).T         # b x b
            G = 0.5 * (G + G.T)
            R, Rinv = _chol_upper_and_inverse(G)                     # G = R'R (no threaded LAPACK: b is 128)
            if R is None:
                return W, None, False
            W = backend.mm(False, False, W, backend.from_numpy(Rinv))
            Racc = R if Racc is None else R @ Racc
        return W, Racc, True

    def agree_min(values):
        """Element-wise minimum of a few host scalars over the ranks (control decisions only)."""
        if not dist.is_initialized():
            return [float(v) for v in values]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=Kcols.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return [float(v) for v in t.tolist()]

    import time as _t
    _prof = {} if os.environ.get("BIGKRLS_VERBOSE") else None

    def _tick(name, t0):
        if _prof is not None:
            backend.sync()
            _prof[name] = _prof.get(name, 0.0) + (_t.perf_counter() - t0)

    _t0 = _t.perf_counter()
    if hasattr(backend, "random_block"):                 # on the device: the single-GPU library's start block
        W0 = backend.random_block(b, n, seed)
    else:
        rng = np.random.default_rng(seed)
        W0 = backend.from_numpy(rng.random((n, b)) - 0.5)
    Bj, _, ok = cholqr2(W0)
    if not agree_min([1.0 if ok else 0.0])[0] > 0.5:
        raise RuntimeError("eigen_krylov_dist: start block is rank deficient")
    Ball = torch.empty((maxdim, n), dtype=torch.float64, device=Bj.device)
    Ball[:b] = Bj
    _tick("start block", _t0)
    Ablk, Bblk = [], []
    # with the HIP backend the blocks of the projected matrix stay on the device (as in csrc/eigen.hip): no per-step
    # read-back of A_j, no host assembly / upload of T at a check
    dev_blocks = hasattr(backend, "projected_eig_top")
    if dev_blocks:
        A_dev = torch.zeros((maxsteps, b, b), dtype=torch.float64, device=Bj.device)
        beta_dev = torch.zeros((maxsteps, b, b), dtype=torch.float64, device=Bj.device)
    steps, converged, Y, theta = 0, False, None, None
    # the check schedule of csrc/eigen.hip (sizes only, so that every rank and every run decides alike): the first
    # check at a subspace of 4 neig columns, or of 2 neig where a step costs more than a check
    check_is_cheap = 12e-6 * 4.0 * neig < 2.0 * float(n) * float(n) * b / 50e12 / max(world, 1)
    next_check = max(2, ((2 if check_is_cheap else 4) * neig + b - 1) // b)
    while True:
        _t0 = _t.perf_counter()
        W = k_times(Ball[steps * b:(steps + 1) * b])
        _tick("K*B", _t0); _t0 = _t.perf_counter()
        dim = (steps + 1) * b
        Bv = Ball[:dim]
        Aj = None
        for pas in range(2):                                          # classical Gram-Schmidt, twice
            Cc = backend.mm(True, False, Bv, W)                       # dim x b
            if pas == 0:
                if dev_blocks:
                    A_dev[steps].copy_(Cc[:, steps * b:(steps + 1) * b])     # (column, row) = column-major A_j
                else:
                    Aj = _host(backend, Cc[:, steps * b:(steps + 1) * b]).T.copy()
            W = backend.mm(False, False, Bv, Cc, alpha=-1.0, beta=1.0, out=W)
        _tick("cgs2", _t0); _t0 = _t.perf_counter()
        W, R, ok = cholqr2(W, beta_dev[steps] if dev_blocks else None)
        _tick("cholqr2", _t0); _t0 = _t.perf_counter()
        if not dev_blocks:
            Ablk.append(0.5 * (Aj + Aj.T))
        steps += 1
        # Every branch below is taken on values agreed by all ranks (a last-bit difference between
        # replicas must never let one rank leave the loop while the others enter the next
        # all-gather): the breakdown flag is the minimum over the ranks.
        ok = bool(agree_min([1.0 if ok else 0.0])[0] > 0.5)
        last = (not ok) or steps >= maxsteps
        if ok:
            Bblk.append(R)
        if last or steps >= next_check:
            m = steps * b
            if dev_blocks:
                theta, Y = backend.projected_eig_top(A_dev, beta_dev, steps, neig)
            else:
                T = np.zeros((m, m))
                for j in range(steps):
                    T[j * b:(j + 1) * b, j * b:(j + 1) * b] = Ablk[j]
                    if j + 1 < steps:
                        T[(j + 1) * b:(j + 2) * b, j * b:(j + 1) * b] = Bblk[j]
                        T[j * b:(j + 1) * b, (j + 1) * b:(j + 2) * b] = Bblk[j].T
                theta, Y = backend.dense_eig_top(T, neig)             # Y: (neig, m)
            worst = 0.0
            if ok:
                Ylast = _host(backend, Y[:, m - b:]).T                     # b x neig
                worst = float(np.max(np.linalg.norm(Bblk[-1] @ Ylast, axis=0)))
            # agreed values: the largest residual and the smallest theta_1 over the ranks
            neg_worst, theta1 = agree_min([-worst, abs(float(theta[0]))])
            worst = -neg_worst
            if worst <= tol * theta1 or last:
                converged = worst <= tol * theta1
                break
            # distance to the tolerance at the collapse rate of the worst residual (x25 - x45 per step once the
            # subspace reaches the neig-th eigenvalue; csrc/eigen.hip has the measurements)
            gain = 40.0 if check_is_cheap else 15.0
            inc = int(math.ceil(math.log(worst / (tol * theta1)) / math.log(gain))) if worst > 0.0 else 1
            next_check = steps + max(1, min(inc, max(2, steps // 2)))
            _tick("check", _t0); _t0 = _t.perf_counter()
        Ball[steps * b:(steps + 1) * b] = W
    if not converged:
        raise RuntimeError("eigen_krylov_dist: not converged within the subspace limit")
    dim = steps * b
    _t0 = _t.perf_counter()
    Q = backend.mm(False, False, Ball[:dim], Y)                       # n x neig
    # The Ritz pairs of T are verified against K on the block that converges last (the smallest min(neig, b)
    # Ritz values: one more sharded product); only if the true residuals are not at the estimated level are all
    # pairs refined by a Rayleigh-Ritz step against K (as csrc/eigen.hip; BIGKRLS_KRY_REFINE=1 forces it).
    refine = os.environ.get("BIGKRLS_KRY_REFINE") == "1"
    vals = np.asarray(theta[:neig], dtype=np.float64)
    if not refine:
        bs = min(neig, b)
        Qs = Q[neig - bs:].contiguous()
        Rs = k_times(Qs)
        Rs = backend.mm(False, False, Qs, backend.from_numpy(np.diag(vals[neig - bs:])), alpha=-1.0, beta=1.0, out=Rs)
        r2 = np.diag(_host(backend, backend.mm(True, False, Rs, Rs)))
        rmax = float(np.sqrt(max(float(np.max(r2)), 0.0)))
        neg_r, theta1 = agree_min([-rmax, abs(float(vals[0]))])
        refine = not (-neg_r <= 10.0 * tol * theta1)
    if refine:
        KQ = k_times(Q)
        H = _host(backend, backend.mm(True, False, Q, KQ)).T
        H = 0.5 * (H + H.T)
        hv, Zr = backend.dense_eig_top(H, neig)
        vals = np.asarray(hv[:neig], dtype=np.float64)
    lastkeeper = int(np.max(np.nonzero(vals >= eigtrunc * vals[0])[0])) + 1
    Qf = backend.mm(False, False, Q, Zr[:lastkeeper]) if refine else Q[:lastkeeper].contiguous()
    _tick("Ritz vectors + verification", _t0)
    if _prof is not None and rank == 0:
        print("[bigkrls] eigen_krylov_dist steps=%d dim=%d" % (steps, dim), {kk: round(v, 3) for kk, v in _prof.items()}, flush=True)
    return vals, lastkeeper, Qf, backend.from_numpy(vals[:, None])


def eigen_dense_dist(backend, torch, dist, A, n, rank, world, nb, neig, eigtrunc):
    """Dense symmetric eigendecomposition with stage 1 (dense -> band, 4/3 N^3 flops) partitioned by
    column blocks over the ranks ("Eigen, dense tridiagonalisation").

    `A`: this rank's column block K[:, c0:c1] as an (ncl, n) tensor, c0 = rank * nb, nb a multiple of
    64; it is overwritten. Per 64-column panel: one broadcast of the panel strip from its owner, the
    replicated panel QR, this rank's rows of Y = A22 V (A22 symmetric: its own columns, transposed),
    one all-gather of Y (N x 64), the replicated thin products and the update of the own columns.
    The reduced matrix (band + reflectors) ends up replicated; stage 2 and the divide & conquer are
    replicated (latency-bound, no flops to share), the back-transform is split by eigenvector column
    and assembled with an all-gather of the column blocks -- the RCCL exchange north_star names.
    Returns (values host (neig), lastkeeper, Q tensor (lastkeeper, n), values tensor)."""
    b = S1_B
    assert nb % b == 0
    c0 = min(rank * nb, n)
    ncl = A.shape[0]
    dev = A.device
    backend.s1_open(n)
    sbuf = torch.empty(b * n, dtype=torch.float64, device=dev)
    Ysend = torch.zeros((b, nb), dtype=torch.float64, device=dev)
    Yrecv = torch.empty((world * b, nb), dtype=torch.float64, device=dev) if dist.is_initialized() else None

    def has_panel(k):
        return k + b < n and n - k - b > 1

    def bcast_strip(k, w):
        """Rows k..n of the global columns k..k+w (inside one owner's block) on every rank."""
        owner = k // nb
        strip = sbuf[: w * (n - k)].view(w, n - k)
        if owner == rank:
            backend.s1_strip_from(A, k - c0, w, k, n, strip)
        if dist.is_initialized():
            dist.broadcast(strip, src=owner)
        return strip

    import time as _t
    _prof = {} if os.environ.get("BIGKRLS_VERBOSE") else None

    def _tick(name, t0):
        if _prof is not None:
            backend.sync()
            _prof[name] = _prof.get(name, 0.0) + (_t.perf_counter() - t0)
        return _t.perf_counter()

    # Look-ahead: the columns of the NEXT panel are updated first (by their owner), its strip is broadcast and its
    # factorisation started on the look-ahead stream, and only then do the ranks update the rest of their columns --
    # the latency-bound panel QR runs beside the throughput-bound update instead of after it.
    k = 0
    if has_panel(0):
        _t0 = _t.perf_counter()
        backend.s1_panel(n, 0, bcast_strip(0, b))
        _t0 = _tick("panel QR + T", _t0)
    while has_panel(k):
        m = n - k - b
        _t0 = _t.perf_counter()
        la0 = min(max(k + b - c0, 0), ncl)           # first own column inside the trailing matrix
        nact = ncl - la0
        backend.s1_av(n, k, A, la0, nact, Ysend)     # (waits for the factorisation of panel k)
        _t0 = _tick("A22 V", _t0)
        if dist.is_initialized():
            dist.all_gather_into_tensor(Yrecv, Ysend)
            Yfull = Yrecv.view(world, b, nb).permute(1, 0, 2).reshape(b, world * nb)
        else:
            Yfull = Ysend
        Y = Yfull[:, k + b: n].contiguous()          # m x 64, column-major
        _t0 = _tick("all-gather Y", _t0)
        backend.s1_thin(n, k, Y)
        row0 = (c0 + la0) - (k + b) if nact > 0 else 0
        nxt = k + b
        first = 0                                     # own columns already updated before the look-ahead
        if has_panel(nxt):
            if nxt // nb == rank:                     # the next panel's columns are the first active ones of their owner
                first = min(b, nact)
                backend.s1_update_cols(n, k, A, la0, first, row0)
            _t0 = _tick("thin products + next panel's columns", _t0)
            strip = bcast_strip(nxt, b)
            _t0 = _tick("strip", _t0)
            backend.s1_panel_begin(n, nxt, strip)
        backend.s1_update_cols(n, k, A, la0 + first, nact - first, row0 + first)
        _t0 = _tick("update (beside the next panel QR)", _t0)
        k += b
    if _prof is not None and rank == 0:
        print("[bigkrls] eigen_dense_dist stage 1:", {kk: round(v, 3) for kk, v in _prof.items()}, flush=True)
    while k < n:                                      # what is left of the trailing matrix: not panels
        owner_end = min((k // nb + 1) * nb, n)
        w = min(b, owner_end - k)
        strip = bcast_strip(k, w)
        backend.s1_put(n, k, strip, w)
        k += w
    vals, lastkeeper, Qpart, dvals = backend.eigen_resume(n, neig, eigtrunc, rank, world)
    if not dist.is_initialized():
        return vals, lastkeeper, Qpart, dvals
    # all-gather of the back-transformed column blocks (rank r holds columns nv r / world .. nv (r+1) / world)
    cuts = [lastkeeper * r // world for r in range(world + 1)]
    pmax = max(cuts[r + 1] - cuts[r] for r in range(world))
    send = torch.zeros((pmax, n), dtype=torch.float64, device=dev)
    mine = cuts[rank + 1] - cuts[rank]
    if mine > 0:
        send[:mine] = Qpart[cuts[rank]: cuts[rank + 1]]
    recv = torch.empty((world * pmax, n), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send)
    Q = torch.cat([recv[r * pmax: r * pmax + (cuts[r + 1] - cuts[r])] for r in range(world)], dim=0).contiguous()
    return vals, lastkeeper, Q, dvals


def bigKRLS_dist(y, X, sigma=None, derivative=True, which_derivatives=None, Neig=None, eigtrunc=None,
                 lambda_=None, L=None, U=None, ctx: Optional[Context] = None, backend=None,
                 timings: Optional[Dict[str, float]] = None, trace=None, keep_outputs=True,
                 eigen_mode: Optional[str] = None) -> BigKRLS:
    """bigKRLS() with the kernel build, lambda search, coefficient, variance and
    marginal-effects passes partitioned over the ranks of the default process group.
    Every rank returns the same small outputs; N x N outputs stay sharded
    (`K.cols`, `vcov.est.c.cols`, `vcov.est.fitted.cols` hold this rank's column block).
    `eigen_mode`: None (block Lanczos with sharded products when N >= 16384 and Neig <= N/8, like
    the single-GPU library; otherwise the dense path with stage 1 partitioned by column blocks),
    "krylov" / "dense" to force either, "replicated" for the dense decomposition replicated on every
    rank (K all-gathered; what tiny problems, n <= 256, always use)."""
    torch, dist = _torch_dist()

    # (every collective below runs whenever a process group exists, also one of size 1: a
    #  single-GPU run under an RCCL group then exercises exactly the calls of a multi-GPU one)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if backend is None:
        backend = HipBackend(ctx or Context())
    Xh = np.array(X, dtype=np.float64, order="F")
    yh = np.array(y, dtype=np.float64).ravel()
    n, p = Xh.shape
    X_init_sd = Xh.std(axis=0, ddof=1)
    if X_init_sd.min() == 0:
        raise ValueError("The following columns in X are constant and must be removed")
    if n != yh.shape[0]:
        raise ValueError("nrow(X) not equal to number of elements in y.")
    Neig = min(n, int(Neig)) if Neig is not None else n
    if eigtrunc is None:
        eigtrunc = 0.001 if n > 3000 else 0.0
    sigma = float(p) if sigma is None else float(sigma)
    y_init_sd, y_init_mean = _sd(yh), float(yh.mean())
    Xs = (Xh - Xh.mean(axis=0)) / X_init_sd
    ys = (yh - y_init_mean) / y_init_sd
    Neig_eff = min(int(Neig), n)
    use_krylov = (eigen_mode == "krylov") or (eigen_mode is None and Neig_eff * 8 <= n and n >= 16384)
    # dense: stage 1 partitioned by column blocks (64-column panels must not straddle two ranks);
    # tiny problems (and eigen_mode="replicated") keep the replicated decomposition
    dense_sharded = not use_krylov and eigen_mode != "replicated" and n >= DENSE_DIST_MIN_N
    nb, parts = partition(n, world, S1_B if dense_sharded else 1)
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
    if dense_sharded or use_krylov:
        Kpad = None                              # K stays sharded: only the own column block exists
        Kloc = backend.empty(n, r1 - r0)
    else:
        Kpad = backend.empty(n, nb * world)      # (nb*world, n): column c at Kpad[c]
        Kloc = Kpad[rank * nb: rank * nb + (r1 - r0)]
    if r1 > r0:
        backend.kernel_cols(Xd, sigma, r0, r1, Kloc)
    mark("kernel")
    if dense_sharded:
        # K is never gathered: every rank keeps (and later reuses) only its own column block, and the
        # reduction works on a copy of it
        Kcols = Kloc
        mark("kernel_allgather")
        vals, lastkeeper, Q, dvals = eigen_dense_dist(backend, torch, dist, Kloc.clone(), n, rank, world, nb,
                                                       Neig_eff, eigtrunc)
        K = None
        mark("eigen")
    elif use_krylov:
        # Neig << N: K stays sharded (no all-gather of K); block Lanczos with sharded K B_j products
        Kcols = Kloc
        mark("kernel_allgather")
        vals, lastkeeper, Q, dvals = eigen_krylov_dist(backend, torch, dist, Kcols, n, rank, world, Neig_eff,
                                                       eigtrunc)
        K = None
        mark("eigen")
    else:
        if dist.is_initialized():
            dist.all_gather_into_tensor(Kpad, Kpad[rank * nb:(rank + 1) * nb].clone())
        K = Kpad[:n]
        Kcols = K[r0:r1]
        mark("kernel_allgather")
        # ---- step 2: eigen (replicated) -------------------------------------------------
        vals, lastkeeper, Q, dvals = backend.eigen(K, Neig, eigtrunc, rank, world)
        if dist.is_initialized():
            # each rank back-transformed its own eigenvector columns (zeros elsewhere): sum = Q.
            # This is the RCCL exchange north_star names for the eigenvector back-transform.
            dist.all_reduce(Q, op=dist.ReduceOp.SUM)
        mark("eigen")
    # ---- step 3: lambda search on row blocks of Q ------------------------------------
    a = backend.qty_rows(Q, r0, r1, yd)
    if dist.is_initialized():
        dist.all_reduce(a)

    def loo(lam):
        le, _ = backend.solveforc_rows(Q, r0, r1, dvals, a, lam, False)
        if dist.is_initialized():
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
    if dist.is_initialized():
        t = torch.tensor([le_loc], dtype=torch.float64, device=a.device)
        dist.all_reduce(t)
        Le = float(t.item())
        c_full = _all_gather_vec(torch, dist, c_loc, nb, n, world)
    else:
        Le, c_full = le_loc, c_loc
    yhat_loc = backend.gemv_t(Kcols, c_full)
    yhat_full = _all_gather_vec(torch, dist, yhat_loc, nb, n, world) if dist.is_initialized() else yhat_loc
    coeffs = _host(backend, c_full).ravel()
    yfitted = _host(backend, yhat_full).ravel()
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
        if dist.is_initialized():
            D_full = _all_gather_vec(torch, dist, D_loc, nb, n, world)
            S_full = _all_gather_vec(torch, dist, S_loc, nb, n, world)
        else:
            D_full, S_full = D_loc, S_loc
        var = backend.deriv_var(Q, wv, S_full, ops.deriv_scales(Xe_h, isb, sigma))
        derivmat = _host(backend, D_full).T.copy()
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
