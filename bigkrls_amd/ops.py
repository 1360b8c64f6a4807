"""Host-side mirror of the reference's `b*` helpers (R/bigKRLS_Rcpp_functions.R).

Same names, same argument meaning and the same error behaviour as the R
wrappers, but every matrix is a `DeviceMatrix` in HBM and every numeric step is a
HIP kernel reached through the C ABI (include/bigkrls.h).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .device import Context, DeviceMatrix

GOLDEN = 0.381966  # R/bigKRLS_Rcpp_functions.R:38-39


def _hptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


# ---------------------------------------------------------------------------
# kernels   (R/bigKRLS_Rcpp_functions.R:201-227)
# ---------------------------------------------------------------------------
def bGaussKernel(X: DeviceMatrix, bandwidth: Optional[float] = None,
                 cols: Optional[Tuple[int, int]] = None) -> DeviceMatrix:
    """K[i,j] = exp(-||x_i-x_j||^2/bandwidth)  (bGaussKernel, :201-210 -> BigGaussKernel,
    src/gauss_kernel.cpp:32-42).  `cols=(c0,c1)` builds only the column block
    K[:, c0:c1] (== the row block, K is symmetric) for the multi-GPU partition."""
    ctx = X.ctx
    n, p = X.nrow, X.ncol
    bandwidth = float(p) if bandwidth is None else float(bandwidth)
    c0, c1 = (0, n) if cols is None else cols
    out = ctx.empty(n, c1 - c0)
    _lib.call("bigkrls_dev_kernel_block", ctx.handle, X.ptr, n, X.ld, X.col_ptr(0, c0), c1 - c0,
              X.ld, p, bandwidth, out.ptr, out.ld, c0)
    return out


def bNeffective(X: DeviceMatrix) -> float:
    """Effective sample size as a function of the mean absolute pairwise correlation of the
    rows of X (bNeffective, R/bigKRLS_Rcpp_functions.R:212-217 -> BigNeffective,
    src/Neffective.cpp:13-76)."""
    out = np.zeros(1, dtype=np.float64)
    _lib.call("bigkrls_dev_neffective", X.ctx.handle, X.ptr, X.nrow, X.ld, X.ncol, _hptr(out))
    return float(out[0])


def bTempKernel(X_new: DeviceMatrix, X_old: DeviceMatrix, sigma: float) -> DeviceMatrix:
    """out[i,j] = exp(-||new_i - old_j||^2/sigma)  (bTempKernel, :219-227)."""
    ctx = X_new.ctx
    if X_new.ncol != X_old.ncol:
        raise ValueError("bTempKernel: column counts differ")
    out = ctx.empty(X_new.nrow, X_old.nrow)
    _lib.call("bigkrls_dev_kernel_block", ctx.handle, X_new.ptr, X_new.nrow, X_new.ld, X_old.ptr,
              X_old.nrow, X_old.ld, X_new.ncol, float(sigma), out.ptr, out.ld, -1)
    return out


# ---------------------------------------------------------------------------
# eigen   (R/bigKRLS_Rcpp_functions.R:173-199)
# ---------------------------------------------------------------------------
@dataclass
class Eigenobject:
    values: np.ndarray          # all Neig eigenvalues, descending (host)
    lastkeeper: int             # number of kept pairs
    vectors: DeviceMatrix       # N x lastkeeper
    values_dev: DeviceMatrix    # the same eigenvalues on the device


def bEigen(A: DeviceMatrix, Neig: Optional[int] = None, eigtrunc: float = 0.0,
           part: Optional[Tuple[int, int]] = None) -> Eigenobject:
    """bEigen (:173-199): all Neig eigenvalues are kept, eigenvectors only for
    1..lastkeeper = max(which(values >= eigtrunc*values[1])) (:190).  The sign flip
    at :186 is immaterial (quirk Q9) and not reproduced.  `part=(rank, world)` (multi-GPU):
    only this rank's slice of the kept eigenvector columns is back-transformed, the others
    are zeros (sum over ranks = Q)."""
    ctx = A.ctx
    n = A.nrow
    if A.ncol != n:
        raise ValueError("bEigen: matrix must be square")
    Neig = n if Neig is None else int(Neig)
    if not (1 <= Neig <= n):
        raise ValueError("bEigen: Neig out of range")
    vals = ctx.empty(Neig, 1)
    vecs = ctx.empty(n, Neig)
    nv = C.c_int64(0)
    if part is None:
        _lib.call("bigkrls_dev_eigen", ctx.handle, A.ptr, n, A.ld, Neig, vals.ptr, Neig,
                  float(eigtrunc), vecs.ptr, vecs.ld, C.byref(nv))
    else:
        _lib.call("bigkrls_dev_eigen_part", ctx.handle, A.ptr, n, A.ld, Neig, vals.ptr, Neig,
                  float(eigtrunc), vecs.ptr, vecs.ld, C.byref(nv), int(part[0]), int(part[1]))
    values = vals.to_numpy().ravel()
    lastkeeper = int(nv.value)
    return Eigenobject(values=values, lastkeeper=lastkeeper, vectors=vecs.cols(0, lastkeeper),
                       values_dev=vals)


# ---------------------------------------------------------------------------
# solveforc / lambda search   (R/bigKRLS_Rcpp_functions.R:5-95)
# ---------------------------------------------------------------------------
class _SolveState:
    """a = Q'y hoisted out of the probes; cached per (Eigenobject, y)."""

    def __init__(self, eig: Eigenobject, y: DeviceMatrix):
        ctx = y.ctx
        Q = eig.vectors
        self.a = ctx.empty(Q.ncol, 1)
        _lib.call("bigkrls_dev_qty", ctx.handle, Q.ptr, Q.nrow, Q.ncol, Q.ld, y.ptr, self.a.ptr)


def _state(eig: Eigenobject, y: DeviceMatrix) -> _SolveState:
    """The cached a = Q'y is reused only for the very same tensor object in the very same state:
    the state keeps a reference to y's tensor (so its address cannot be recycled for another
    array while the cache lives) and torch's in-place version counter (bumped by every torch
    write). Writes torch cannot see (a raw kernel through the C ABI) must go through a new
    DeviceMatrix or `forget_solve_state`."""
    key = "_solve_state"
    st = getattr(eig, key, None)
    if st is None or st.y_tensor is not y.t or st.y_version != y.t._version:
        st = _SolveState(eig, y)
        st.y_tensor = y.t
        st.y_version = y.t._version
        setattr(eig, key, st)
    return st


def forget_solve_state(eig: Eigenobject) -> None:
    """Drop the cached Q'y of an Eigenobject (after y was overwritten behind torch's back)."""
    if hasattr(eig, "_solve_state"):
        delattr(eig, "_solve_state")


def bSolveForc(y: DeviceMatrix, Eigenobject: Eigenobject, lambda_: float):
    """bSolveForc (:84-90) -> BigSolveForc (src/solveforc.cpp:67-78).
    Returns dict(Le=..., coeffs=DeviceMatrix N x 1)."""
    ctx = y.ctx
    Q = Eigenobject.vectors
    st = _state(Eigenobject, y)
    c = ctx.empty(Q.nrow, 1)
    le = C.c_double()
    _lib.call("bigkrls_dev_solveforc", ctx.handle, Q.ptr, Q.nrow, Q.ncol, Q.ld,
              Eigenobject.values_dev.ptr, st.a.ptr, float(lambda_), c.ptr, C.byref(le))
    return {"Le": float(le.value), "coeffs": c}


def bLooLoss(y: DeviceMatrix, Eigenobject: Eigenobject, lambda_: float) -> float:
    """bLooLoss (:92-95)."""
    ctx = y.ctx
    Q = Eigenobject.vectors
    st = _state(Eigenobject, y)
    le = C.c_double()
    _lib.call("bigkrls_dev_solveforc", ctx.handle, Q.ptr, Q.nrow, Q.ncol, Q.ld,
              Eigenobject.values_dev.ptr, st.a.ptr, float(lambda_), None, C.byref(le))
    return float(le.value)


def lambda_bounds(values: np.ndarray, n: int) -> Tuple[float, float]:
    """The U and L loops of bLambdaSearch (:16-36) (host, uses ALL Neig values, quirk Q5)."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    L = C.c_double()
    U = C.c_double()
    _lib.call("bigkrls_lambda_bounds", _hptr(v), v.size, int(n), C.byref(L), C.byref(U))
    return float(L.value), float(U.value)


def bLambdaSearch(L=None, U=None, y: DeviceMatrix = None, Eigenobject: Eigenobject = None,
                  tol=None, noisy=False, trace: Optional[List[Tuple[float, float]]] = None,
                  loo=None) -> float:
    """bLambdaSearch (:5-82), statement for statement.  `loo(lambda)` may be
    supplied to evaluate the leave-one-out loss elsewhere (the multi-GPU path sums
    row-block partials with an all-reduce); by default it is bLooLoss."""
    if np.isnan(Eigenobject.values).any():
        raise ValueError("Missing eigenvalues prevent bigKRLS from obtaining the regularization "
                         "parameter lambda.\n\tCheck for repeated observations (or other perfect "
                         "linear combinations in X).")
    n = y.nrow
    if tol is None:
        tol = 1e-3 * n
    else:
        if not (np.isscalar(tol) and tol > 0):
            raise ValueError("tol must be a positive scalar")
    if U is None or L is None:
        L0, U0 = lambda_bounds(Eigenobject.values, n)
        U = U0 if U is None else U
        L = L0 if L is None else L
    if not (np.isscalar(U) and U > 0):
        raise ValueError("U must be a positive scalar")
    if not (np.isscalar(L) and L >= 0):
        raise ValueError("L must be a non-negative scalar")
    if loo is None:
        def loo(lam):
            return bLooLoss(y=y, Eigenobject=Eigenobject, lambda_=lam)

    def probe(lam):
        s = loo(lam)
        if trace is not None:
            trace.append((float(lam), float(s)))
        return s

    X1 = L + GOLDEN * (U - L)
    X2 = U - GOLDEN * (U - L)
    S1 = probe(X1)
    S2 = probe(X2)
    it = 0
    while abs(S1 - S2) > tol:
        if S1 < S2:
            U = X2
            X2 = X1
            X1 = L + GOLDEN * (U - L)
            S2 = S1
            S1 = probe(X1)
        else:
            L = X1
            X1 = X2
            X2 = U - GOLDEN * (U - L)
            S1 = S2
            S2 = probe(X2)
        it += 1
        if it > 10000 or not np.isfinite(S1 + S2):
            raise RuntimeError("bLambdaSearch: golden-section search did not terminate")
        if noisy:
            print(f"L: {L:.3f} X1: {X1:.3f} X2: {X2:.3f} U: {U:.3f} S1: {S1:.3f} S2: {S2:.3f}")
    return float(X1 if S1 < S2 else X2)


# ---------------------------------------------------------------------------
# multdiag / cross-products   (R/bigKRLS_Rcpp_functions.R:159-171, 229-258)
# ---------------------------------------------------------------------------
def bMultDiag(X: DeviceMatrix, v) -> DeviceMatrix:
    """out[:,i] = X[:,i]*v[i] (bMultDiag :159-171 -> src/multdiag.cpp:13-24)."""
    ctx = X.ctx
    v = np.asarray(v, dtype=np.float64).ravel()
    if v.size < X.ncol:
        raise ValueError("bMultDiag: diag shorter than ncol(X)")
    dv = ctx.from_numpy(v[: X.ncol])
    out = ctx.empty(X.nrow, X.ncol)
    _lib.call("bigkrls_dev_multdiag", ctx.handle, X.ptr, X.nrow, X.ncol, X.ld, dv.ptr, out.ptr, out.ld)
    return out


def gemm(ta: bool, tb: bool, A: DeviceMatrix, B: DeviceMatrix, alpha: float = 1.0,
         out: Optional[DeviceMatrix] = None, beta: float = 0.0) -> DeviceMatrix:
    ctx = A.ctx
    m = A.ncol if ta else A.nrow
    k = A.nrow if ta else A.ncol
    kb = B.ncol if tb else B.nrow
    n = B.nrow if tb else B.ncol
    if k != kb:
        raise ValueError(f"gemm: inner dimensions differ ({k} vs {kb})")
    if out is None:
        out = ctx.empty(m, n)
    _lib.call("bigkrls_dev_gemm", ctx.handle, int(ta), int(tb), m, n, k, float(alpha), A.ptr, A.ld,
              B.ptr, B.ld, float(beta), out.ptr, out.ld)
    return out


def bCrossProd(X: DeviceMatrix, Y: Optional[DeviceMatrix] = None) -> DeviceMatrix:
    """X'Y or X'X (bCrossProd :229-243 -> src/crossprod.cpp:13-48)."""
    return gemm(True, False, X, X if Y is None else Y)


def bTCrossProd(X: DeviceMatrix, Y: Optional[DeviceMatrix] = None) -> DeviceMatrix:
    """XY' or XX' (bTCrossProd :245-258 -> src/crossprod.cpp:51-85)."""
    return gemm(False, True, X, X if Y is None else Y)


def matvec(A: DeviceMatrix, x: DeviceMatrix, trans: bool = False) -> DeviceMatrix:
    """A %*% x for a single column x (bigalgebra dgemv, R/bigKRLS.R:291,601)."""
    ctx = A.ctx
    out = ctx.empty(A.ncol if trans else A.nrow, 1)
    _lib.call("bigkrls_dev_gemv", ctx.handle, int(trans), A.nrow, A.ncol, 1.0, A.ptr, A.ld, x.ptr,
              0.0, out.ptr)
    return out


# ---------------------------------------------------------------------------
# derivatives   (R/bigKRLS_Rcpp_functions.R:260-270)
# ---------------------------------------------------------------------------
def binary_columns(Xhost: np.ndarray) -> np.ndarray:
    """src/bigderiv_v3.cpp:28-31: a column with exactly two distinct values."""
    out = np.zeros(Xhost.shape[1], dtype=bool)
    for j in range(Xhost.shape[1]):
        col = Xhost[:, j]
        if col.size:
            lo, hi = col.min(), col.max()          # exactly two distinct values <=> everything is lo or hi
            out[j] = bool(lo != hi and np.all((col == lo) | (col == hi)))
    return out


def deriv_scales(Xhost: np.ndarray, is_binary: np.ndarray, sigma: float) -> np.ndarray:
    n, p = Xhost.shape
    sc = np.empty(p)
    for j in range(p):
        if is_binary[j]:
            sd = 1.0 / (Xhost[:, j].max() - Xhost[:, j].min())      # src/bigderiv_v3.cpp:36
            sc[j] = 2.0 * sd * sd / (float(n) ** 2)                  # :85
        else:
            sc[j] = 4.0 / (sigma * sigma * float(n) ** 2)            # :105
    return sc


def bDerivatives(X: DeviceMatrix, sigma: float, K: DeviceMatrix, coeffs: DeviceMatrix,
                 eig: Eigenobject, wv: np.ndarray, Xhost: np.ndarray):
    """bDerivatives (:260-270) -> BigDerivMat (src/bigderiv_v3.cpp:113-132) in its
    O(N^2) form.  `vcovmatc` is represented by its factors (eig.vectors, wv):
    V = Q diag(wv) Q'.  Returns dict(derivatives=DeviceMatrix N x P, varavgderiv=ndarray P)."""
    ctx = X.ctx
    n, p = X.nrow, X.ncol
    isb = binary_columns(Xhost)
    isb32 = np.ascontiguousarray(isb.astype(np.int32))
    D = ctx.empty(n, p)
    S = ctx.empty(n, p)
    _lib.call("bigkrls_dev_deriv_rows", ctx.handle, K.ptr, n, n, K.ld, 0, X.ptr, p, X.ld,
              _hptr(isb32), coeffs.ptr, float(sigma), D.ptr, D.ld, S.ptr, S.ld)
    scale = np.ascontiguousarray(deriv_scales(Xhost, isb, sigma))
    var = np.empty(p)
    Q = eig.vectors
    dwv = ctx.from_numpy(np.asarray(wv, dtype=np.float64)[: Q.ncol])
    _lib.call("bigkrls_dev_deriv_var", ctx.handle, Q.ptr, n, Q.ncol, Q.ld, dwv.ptr, S.ptr, p, S.ld,
              _hptr(scale), _hptr(var))
    return {"derivatives": D, "varavgderiv": var}
