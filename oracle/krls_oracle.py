"""CPU oracle for the bigKRLS fit / predict / cross-validation hot path.

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement (numpy + the LAPACK /
ARPACK that ship inside scipy) of the reference's algorithm; it is the parity
checker for the HIP path and the `cpu_baseline` leg of bench.py.  Nothing under
`bigkrls_amd/` may import it.  Only `tests/`, `__graft_entry__.smoke()` and
bench.py's `cpu_baseline` leg do.

Pinning status (all checked in tests/test_oracle.py against this file):
(i) the reference's own tests pin one kernel column of `mtcars`
(tests/testthat/test_basic_usage.R:71-108, one-sided tol 0.01) and (ii) one predicted
proportion, 0.6875 (:64-67): reproduced to 7e-15 and exactly; (iii) the reference's
only end-to-end known answer -- the six average marginal effects of the N=500, P=6
(binary last column), eigtrunc=0.01 fit printed in examples/numeric_convergence.md:40-46
-- is reproduced to all 7 printed significant figures by both the literal and the
O(N^2) restatement, from inputs regenerated with a restatement of R's random stream
(oracle/r_rng.py: set.seed scrambling, Mersenne-Twister, inversion rnorm, itself
checked against R's well-known first draws). (iii) pins kernel -> eigen -> truncation
-> lambda search -> coefficients -> continuous and binary derivatives -> rescaling.
Individual eigenvalues, lambda and c are not printed anywhere in the reference tree.
The reference cannot be built here (no R, Rcpp, RcppArmadillo, bigmemory) so there is
no `oracle/_ref`.

Where the arithmetic lives in third-party code that is absent from
/root/reference (Armadillo via RcppArmadillo -> LAPACK dsyevd / BLAS dgemm;
ARPACK via arma::eigs_sym; versions unpinned by DESCRIPTION:9-11) the stand-ins
are scipy.linalg.eigh(driver="evd") (= dsyevd, what arma::eig_sym's default "dc"
method calls) and scipy.sparse.linalg.eigsh(which="LM", tol=0) (= ARPACK IRLM,
what arma::eigs_sym is).

Every function cites the reference file:line it follows.  `*_literal` functions
keep the reference's loop structure and its avoidable O(N^3) terms; `*_fast`
functions use the O(N^2) identities the HIP path uses (documented in DESIGN.md)
and are asserted equal to the literal ones in tests/test_oracle.py.

All matrices are float64; "column-major" is irrelevant at this level (numpy
arrays, any layout).
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as sla

GOLDEN = 0.381966  # R/bigKRLS_Rcpp_functions.R:38-39 (hard-coded, not (3-sqrt5)/2)
DBL_EPS = np.finfo(np.float64).eps  # .Machine$double.eps, R/bigKRLS_Rcpp_functions.R:28


# --------------------------------------------------------------------------
# synthetic inputs G(N, P, seed)  
# --------------------------------------------------------------------------
def synth(n: int, p: int, seed: int, binary_last: bool = False):
    """G(N,P,seed): X ~ N(0,1), beta_j = j/||1..P||, y = sin(X beta) + 0.25 eps."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    beta = np.arange(1, p + 1, dtype=np.float64)
    beta /= np.linalg.norm(beta)
    if binary_last:
        X[:, p - 1] = (X[:, p - 1] > 0.12345).astype(np.float64)
    y = np.sin(X @ beta) + 0.25 * rng.standard_normal(n)
    return X, y


# --------------------------------------------------------------------------
# R-level statistics helpers
# --------------------------------------------------------------------------
def r_sd(v: np.ndarray) -> float:
    """R's sd(): n-1 denominator (biganalytics::colsd, R/bigKRLS.R:179,248)."""
    return float(np.std(v, ddof=1))


def r_var(v: np.ndarray) -> float:
    return float(np.var(v, ddof=1))


def r_cor(a: np.ndarray, b: np.ndarray) -> float:
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    a0 = a - a.mean()
    b0 = b - b.mean()
    return float((a0 @ b0) / math.sqrt((a0 @ a0) * (b0 @ b0)))


def standardize(X: np.ndarray, y: np.ndarray):
    """R/bigKRLS.R:248-254: column-wise (x-mean)/sd with n-1 sd; same for y."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).ravel()
    xm = X.mean(axis=0)
    xs = X.std(axis=0, ddof=1)
    ym = float(y.mean())
    ys = r_sd(y)
    return (X - xm) / xs, (y - ym) / ys, xm, xs, ym, ys


def is_binary_column(col: np.ndarray) -> bool:
    """R/bigKRLS.R:242 and src/bigderiv_v3.cpp:28-31: exactly two unique values."""
    return np.unique(col).size == 2


# --------------------------------------------------------------------------
# a1/a2  kernels
# --------------------------------------------------------------------------
def gauss_kernel_literal(X: np.ndarray, sigma: float) -> np.ndarray:
    """src/gauss_kernel.cpp:13-30: upper-triangle loop, exp(-sum((xi-xj)^2)/sigma),
    mirrored.  The inner j loop is vectorised per row i (same arithmetic per
    element: difference, square, sum over columns, divide, exp)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = X.shape[0]
    K = np.zeros((n, n))
    for i in range(n):
        diff = X[i] - X[i:]                      # X.row(i) - X.row(j), j >= i
        sim = np.exp(-1.0 * np.sum(diff * diff, axis=1) / sigma)
        K[i:, i] = sim                           # out(j,i)
        K[i, i:] = sim                           # out(i,j)
    return K


def temp_kernel_literal(A: np.ndarray, B: np.ndarray, sigma: float) -> np.ndarray:
    """src/temp_kernel.cpp:13-30: out(i,j) = exp(-||A_i - B_j||^2 / sigma), U x V."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    out = np.empty((A.shape[0], B.shape[0]))
    for i in range(A.shape[0]):
        diff = A[i] - B
        out[i] = np.exp(-1.0 * np.sum(diff * diff, axis=1) / sigma)
    return out


# --------------------------------------------------------------------------
# a3  eigen
# --------------------------------------------------------------------------
@dataclass
class EigenObject:
    values: np.ndarray       # all Neig values, descending (R keeps all of them, Q5)
    lastkeeper: int          # 1-based count of kept pairs
    vectors: np.ndarray      # N x lastkeeper


def big_eigen_literal(A: np.ndarray, neig: int):
    """src/eigen.cpp:13-30.  Neig < N -> eigs_sym (ARPACK, largest magnitude) on
    sp_mat(A); else eig_sym (dsyevd).  Then flip to descending order (:28-29)."""
    n = A.shape[0]
    if neig < n:
        import scipy.sparse.linalg as ssl
        vals, vecs = ssl.eigsh(A, k=int(neig), which="LM", tol=0)
        order = np.argsort(vals)                 # eigs_sym returns ascending
        vals, vecs = vals[order], vecs[:, order]
    else:
        vals, vecs = sla.eigh(A, driver="evd")   # ascending
    return vals[::-1].copy(), vecs[:, ::-1].copy()


def b_eigen(A: np.ndarray, neig: Optional[int] = None, eigtrunc: float = 0.0) -> EigenObject:
    """R/bigKRLS_Rcpp_functions.R:173-199 (bEigen)."""
    n = A.shape[0]
    neig = n if neig is None else int(neig)
    vals, vecs = big_eigen_literal(A, neig)
    vecs = -1.0 * vecs                           # :186
    keep = np.nonzero(vals >= eigtrunc * vals[0])[0]
    lastkeeper = int(keep.max()) + 1             # :190  max(which(...)), 1-based
    return EigenObject(values=vals, lastkeeper=lastkeeper, vectors=vecs[:, :lastkeeper].copy())


# --------------------------------------------------------------------------
# a4  solveforc
# --------------------------------------------------------------------------
def solveforc_literal(Q: np.ndarray, eigenvalues: np.ndarray, y: np.ndarray, lam: float):
    """src/solveforc.cpp:13-65.  Row i of the lower triangle of
    G^-1 = Q diag(1/(d+lambda)) Q' is formed as a (1 x K)(K x (i+1)) product on a
    growing view of Q' (:41-42); Ginv_diag, coeffs accumulated as at :44-46
    (Q3: span(0,i-1) at i=0 is empty; sum(ginv * y(0..i)) is the dot product).
    Only the first K = ncol(Q) eigenvalues are used (Q1)."""
    Q = np.asarray(Q, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).ravel()
    n, k = Q.shape
    w = 1.0 / (np.asarray(eigenvalues, dtype=np.float64)[:k] + lam)
    Qt = np.ascontiguousarray(Q.T)               # Eigenvectors = trans(Eigenvectors) (:34)
    ginv_diag = np.zeros(n)
    coeffs = np.zeros(n)
    for i in range(n):
        ginv = (Qt[:, i] * w) @ Qt[:, : i + 1]   # (:42)
        ginv_diag[i] = ginv[i]
        if i > 0:
            coeffs[:i] += ginv[:i] * y[i]        # (:45)
        coeffs[i] += float(ginv @ y[: i + 1])    # (:46)
    le = float(np.sum((coeffs / ginv_diag) ** 2))  # (:56-58)
    return le, coeffs


def solveforc_fast(Q: np.ndarray, eigenvalues: np.ndarray, y: np.ndarray, lam: float):
    """Identity used by the HIP path: c = Q (w o Q'y), g_i = sum_k Q_ik^2 w_k."""
    n, k = Q.shape
    w = 1.0 / (np.asarray(eigenvalues, dtype=np.float64)[:k] + lam)
    a = Q.T @ np.asarray(y, dtype=np.float64).ravel()
    c = Q @ (w * a)
    g = (Q * Q) @ w
    return float(np.sum((c / g) ** 2)), c


# --------------------------------------------------------------------------
# a5  lambda search
# --------------------------------------------------------------------------
@dataclass
class LambdaTrace:
    L0: float
    U0: float
    probes: List[Tuple[float, float]] = field(default_factory=list)  # (lambda, Le)
    n_l_iters: int = 0
    n_u_iters: int = 0


def lambda_bounds(values: np.ndarray, n: int, L=None, U=None, trace: Optional[LambdaTrace] = None):
    """R/bigKRLS_Rcpp_functions.R:16-36: U counts down from n while
    sum(d/(d+U)) < 1; L counts up from eps by 0.05 while sum(d/(d+L)) > q with
    q = which.min(abs(d - max(d)/1000)) (1-based).  All Neig values are used (Q5)."""
    d = np.asarray(values, dtype=np.float64)
    nu = nl = 0
    if U is None:
        U = float(n)
        while np.sum(d / (d + U)) < 1:
            U -= 1
            nu += 1
    if L is None:
        L = float(DBL_EPS)
        q = int(np.argmin(np.abs(d - d.max() / 1000.0))) + 1
        while np.sum(d / (d + L)) > q:
            L += 0.05
            nl += 1
    if trace is not None:
        trace.n_l_iters, trace.n_u_iters = nl, nu
    return float(L), float(U)


def lambda_search(eig: EigenObject, y: np.ndarray, L=None, U=None, tol=None,
                  solver=None, trace: Optional[LambdaTrace] = None) -> float:
    """R/bigKRLS_Rcpp_functions.R:5-82 (bLambdaSearch).  `tol` defaults to
    1e-3*n (:11-12); bigKRLS() never forwards its own `tol` (R/bigKRLS.R:274-275)."""
    if np.isnan(eig.values).any():
        raise ValueError("Missing eigenvalues prevent bigKRLS from obtaining the regularization parameter lambda.")
    solver = solver or solveforc_fast
    y = np.asarray(y, dtype=np.float64).ravel()
    n = y.shape[0]
    if tol is None:
        tol = 1e-3 * n
    L, U = lambda_bounds(eig.values, n, L, U, trace)
    if trace is not None:
        trace.L0, trace.U0 = L, U

    def loo(lam):
        le, _ = solver(eig.vectors, eig.values, y, lam)
        if trace is not None:
            trace.probes.append((float(lam), float(le)))
        return le

    X1 = L + GOLDEN * (U - L)
    X2 = U - GOLDEN * (U - L)
    S1 = loo(X1)
    S2 = loo(X2)
    while abs(S1 - S2) > tol:
        if S1 < S2:
            U = X2
            X2 = X1
            X1 = L + GOLDEN * (U - L)
            S2 = S1
            S1 = loo(X1)
        else:
            L = X1
            X1 = X2
            X2 = U - GOLDEN * (U - L)
            S1 = S2
            S2 = loo(X2)
    return float(X1 if S1 < S2 else X2)


# --------------------------------------------------------------------------
# a7/a8  multdiag, crossprods
# --------------------------------------------------------------------------
def multdiag(A: np.ndarray, v: np.ndarray) -> np.ndarray:
    """src/multdiag.cpp:13-24: out[:,i] = A[:,i]*v[i] for i < ncol(A) (v may be longer)."""
    return A * np.asarray(v, dtype=np.float64).ravel()[: A.shape[1]]


def crossprod(A, B=None):
    """src/crossprod.cpp:13-48: A'B or A'A."""
    return A.T @ (A if B is None else B)


def tcrossprod(A, B=None):
    """src/crossprod.cpp:51-85: AB' or AA'."""
    return A @ (A if B is None else B).T


# --------------------------------------------------------------------------
# a9  derivatives
# --------------------------------------------------------------------------
def derivmat_literal(X: np.ndarray, K: np.ndarray, V: np.ndarray, coeffs: np.ndarray, sigma: float):
    """src/bigderiv_v3.cpp:13-111, literal (N x N temporaries, N^3 products).
    Returns (Derivatives N x P, VarAvgDerivatives P)."""
    X = np.asarray(X, dtype=np.float64)
    n, p = X.shape
    c = np.asarray(coeffs, dtype=np.float64).ravel()
    D = np.full((n, p), -1.0)
    var = np.full(p, -1.0)
    for j in range(p):
        xj = X[:, j]
        if np.unique(xj).size == 2:                      # :28-31
            z0, z1 = xj.min(), xj.max()                   # :34-35
            sdxj = 1.0 / (z1 - z0)                        # :36
            phi = -1.0 / (sdxj ** 2 * sigma)              # :37
            kt_rs = np.zeros(n)
            kc_rs = np.zeros(n)
            adj_t = np.zeros((n, n))
            adj_c = np.zeros((n, n))
            for i in range(n):                            # BIG LOOP #1 (:50-78)
                c1 = 1 if xj[i] == z0 else 0
                both_max = ((xj + xj[i]) == 2 * z1).astype(np.float64)
                both_min = ((xj + xj[i]) == 2 * z0).astype(np.float64)
                first_greater = (xj[i] > xj).astype(np.float64)
                second_greater = (xj[i] < xj).astype(np.float64)
                adj_t_local = both_min - first_greater
                adj_c_local = both_max - second_greater
                adj_t[i, :] = adj_t_local + first_greater - second_greater
                adj_c[i, :] = adj_c_local - first_greater + second_greater
                kt_rs[i] = float(np.exp(adj_t_local * phi) @ K[:, i])   # Q2: dot product
                kc_rs[i] = float(np.exp(adj_c_local * phi) @ K[:, i])
                c2 = np.exp((-2 * (both_max + both_min) + 1) * (z1 - z0) ** 2 / sigma)
                D[i, j] = float((sdxj * (-1.0) ** c1 * (1 - c2) * K[:, i]) @ c)  # Q1
            Vt = V.T
            a_t = np.sum((np.exp(adj_t * phi) * K) @ Vt, axis=0)        # :82-84
            a_c = np.sum((np.exp(adj_c * phi) * K) @ Vt, axis=0)
            vcv_sum = float(np.sum(a_t * kt_rs + a_c * kc_rs - 2 * a_t * kc_rs))
            var[j] = 2 * sdxj ** 2 * vcv_sum / n ** 2                   # :85
        else:
            differences = xj[:, None] - xj[None, :]       # differences.col(i) = X.col(j) - X(i,j) (:95)
            Lm = differences * K                          # :102
            D[:, j] = (-2.0 / sigma) * (Lm @ c)           # :103
            var[j] = (1.0 / n ** 2) * (-2.0 / sigma) ** 2 * float(np.sum(Lm.T @ V @ Lm))  # :105
    return D, var


def derivmat_fast(X: np.ndarray, K: np.ndarray, c: np.ndarray, sigma: float,
                  Q: np.ndarray, wv: np.ndarray):
    """O(N^2) restatement of src/bigderiv_v3.cpp used by the HIP path.

    V = Q diag(wv) Q' is never formed: s'Vs = sum_k wv_k (q_k's)^2.
    Continuous column: s = x o (K1) - K x ;  D = (-2/sigma)(x o (Kc) - K(x o c)).
    Binary column (b = [x == z1], values z0 < z1, phi = -(z1-z0)^2/sigma, E=e^phi):
      row in group g in {0,1}; same-group sums  S1 = K[b==g] 1, Sc = K[b==g] c;
      other-group sums O1, Oc.
      D_i   = sd * (+1 if x_i=z1 else -1) * ((1-E) Sc_i + (1-1/E) Oc_i)
      x_i=z0: KT_rs = u_T = E*S1 + O1,      KC_rs = u_C = S1 + O1/E
      x_i=z1: KT_rs = u_T = S1 + O1/E,      KC_rs = u_C = E*S1 + O1
      (K symmetric => the column sums u_T,u_C equal the row sums KT_rs,KC_rs), so
      var   = 2 sd^2/N^2 (KT'V KT + KC'V KC - 2 KT'V KC) = 2 sd^2/N^2 (KT-KC)'V(KT-KC)
    """
    X = np.asarray(X, dtype=np.float64)
    n, p = X.shape
    c = np.asarray(c, dtype=np.float64).ravel()
    one = np.ones(n)
    D = np.empty((n, p))
    var = np.empty(p)
    K1 = K @ one
    Kc = K @ c

    def vquad(a, b):
        return float(np.sum(wv * (Q.T @ a) * (Q.T @ b)))

    for j in range(p):
        x = X[:, j]
        if np.unique(x).size == 2:
            z0, z1 = x.min(), x.max()
            sd = 1.0 / (z1 - z0)
            phi = -((z1 - z0) ** 2) / sigma
            E = math.exp(phi)
            Einv = math.exp(-phi)
            b = (x == z1).astype(np.float64)
            Kb = K @ b
            Kbc = K @ (b * c)
            hi = b == 1.0
            S1 = np.where(hi, Kb, K1 - Kb)
            O1 = np.where(hi, K1 - Kb, Kb)
            Sc = np.where(hi, Kbc, Kc - Kbc)
            Oc = np.where(hi, Kc - Kbc, Kbc)
            sign = np.where(hi, 1.0, -1.0)
            D[:, j] = sd * sign * ((1 - E) * Sc + (1 - Einv) * Oc)
            kt = np.where(hi, S1 + Einv * O1, E * S1 + O1)
            kc = np.where(hi, E * S1 + O1, S1 + Einv * O1)
            vs = vquad(kt - kc, kt - kc)
            var[j] = 2 * sd ** 2 * vs / n ** 2
        else:
            s = x * K1 - K @ x
            D[:, j] = (-2.0 / sigma) * (x * Kc - K @ (x * c))
            var[j] = (4.0 / (sigma ** 2 * n ** 2)) * vquad(s, s)
    return D, var


# --------------------------------------------------------------------------
# the fit (R/bigKRLS.R:97-516, numeric part)
# --------------------------------------------------------------------------
def fit(y, X, sigma=None, derivative=True, which_derivatives: Optional[Sequence[int]] = None,
        vcov_est=True, neig=None, eigtrunc=None, lam=None, L=None, U=None,
        literal=True, return_squares=True, timings: Optional[Dict[str, float]] = None,
        trace: Optional[LambdaTrace] = None) -> Dict[str, object]:
    """Numeric restatement of bigKRLS() (R/bigKRLS.R:175-470).  `which_derivatives`
    is 1-based like R.  literal=True keeps the reference's O(N^3) terms
    (solveforc row loop, V_yhat = K'(VK), L'VL); literal=False uses the identities."""
    T = timings if timings is not None else {}
    X = np.array(X, dtype=np.float64)
    y = np.array(y, dtype=np.float64).ravel()
    n, p = X.shape
    if np.isnan(X).any():
        raise ValueError("the following columns in X contain missing data, which must be removed")
    x_init_sd = X.std(axis=0, ddof=1)                              # :179
    if x_init_sd.min() == 0:
        raise ValueError("The following columns in X are constant and must be removed")
    if n != y.shape[0]:
        raise ValueError("nrow(X) not equal to number of elements in y.")
    if np.isnan(y).any():
        raise ValueError("y contains missing data.")
    if r_sd(y) == 0:
        raise ValueError("y is a constant.")
    neig = min(n, int(neig)) if neig is not None else n            # :194
    if eigtrunc is None:
        eigtrunc = 0.001 if n > 3000 else 0.0                      # :195-201
    elif eigtrunc < 0 or eigtrunc > 1:
        raise ValueError("eigtrunc must be between 0 (no truncation) and 1 (keep largest only).")
    if which_derivatives is not None:
        if not derivative:
            raise ValueError("which.derivative requires derivative = TRUE")
        assert all(1 <= w <= p for w in which_derivatives)
    if derivative and not vcov_est:
        raise ValueError("vcov.est is needed to get derivatives (derivative==TRUE requires vcov.est=TRUE).")
    sigma = float(p) if sigma is None else float(sigma)            # :230
    x_is_binary = np.array([np.unique(X[:, j]).size == 2 for j in range(p)])  # :242 (raw X)

    y_init = y.copy()
    y_init_sd = r_sd(y_init)                                       # :248
    y_init_mean = float(y_init.mean())
    Xs, ys, _, _, _, _ = standardize(X, y)                         # :251-254

    t0 = time.perf_counter()
    K = gauss_kernel_literal(Xs, sigma)                            # Step 1 (:262)
    T["kernel"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    eig = b_eigen(K, neig, eigtrunc)                               # Step 2 (:266)
    T["eigen"] = time.perf_counter() - t0

    solver = solveforc_literal if literal else solveforc_fast
    t0 = time.perf_counter()
    if lam is None:
        lam = lambda_search(eig, ys, L=L, U=U, solver=solver, trace=trace)   # Step 3 (:274)
    T["lambda"] = time.perf_counter() - t0

    w: Dict[str, object] = {}
    w["K.eigenvalues"] = eig.values
    w["lastkeeper"] = eig.lastkeeper
    w["Neffective"] = n - float(np.sum(eig.values / (eig.values + lam)))      # :280 (all Neig, Q5)

    t0 = time.perf_counter()
    le, coeffs = solver(eig.vectors, eig.values, ys, lam)          # Step 4 (:286)
    yfitted = K @ coeffs                                           # :291
    T["coeffs"] = time.perf_counter() - t0

    V = Vyhat = None
    sigmasq = None
    if vcov_est:
        t0 = time.perf_counter()
        resid = ys - yfitted
        sigmasq = float(resid @ resid) / n                         # :294
        wv = sigmasq * (eig.values[: eig.lastkeeper] + lam) ** -2.0
        m = multdiag(eig.vectors, sigmasq * (eig.values + lam) ** -2.0)  # :299
        V = tcrossprod(m, eig.vectors)                             # :301
        T["vcov_c"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if literal:
            Vyhat = crossprod(K, V @ K)                            # :307 (4N^3)
        else:
            dd = eig.values[: eig.lastkeeper]
            Vyhat = (eig.vectors * (wv * dd * dd)) @ eig.vectors.T
        T["vcov_fitted"] = time.perf_counter() - t0

    if derivative:
        t0 = time.perf_counter()
        cols = list(range(p)) if which_derivatives is None else [int(i) - 1 for i in which_derivatives]
        X_est = Xs[:, cols]                                        # :326
        if literal:
            derivmat, varavg = derivmat_literal(X_est, K, V, coeffs, sigma)   # :329
        else:
            derivmat, varavg = derivmat_fast(X_est, K, coeffs, sigma, eig.vectors, wv)
        T["derivatives"] = time.perf_counter() - t0
        w["derivatives.std"] = derivmat.copy()                     # raw standardised (for Q6-free parity)
        w["var.avgderivatives.std"] = varavg.copy()
        yhat_ame = X_est @ derivmat.mean(axis=0)                   # :390
        w["R2AME"] = r_cor(y_init, yhat_ame) ** 2                  # :392
        derivmat = y_init_sd * derivmat                            # :394
        for i in range(derivmat.shape[1]):
            derivmat[:, i] = derivmat[:, i] / x_init_sd[i]         # :395-397 (Q6: index i, not cols[i])
        w["avgderivatives"] = derivmat.mean(axis=0)[None, :]       # :400
        w["var.avgderivatives"] = ((y_init_sd / x_init_sd[cols]) ** 2 * varavg)[None, :]  # :403-407
        w["derivatives"] = derivmat

    w["coeffs"] = coeffs
    w["y"] = y_init
    w["X"] = X
    w["sigma"] = sigma
    w["lambda"] = float(lam)
    w["binaryindicator"] = x_is_binary
    w["which.derivatives"] = None if which_derivatives is None else list(which_derivatives)
    w["yfitted.std"] = yfitted.copy()
    yf = yfitted * y_init_sd + y_init_mean                         # :428
    w["yfitted"] = yf
    w["R2"] = 1 - (r_var(y_init - yf) / (y_init_sd ** 2))          # :429
    w["Looe"] = le * y_init_sd                                     # :430
    w["Le"] = le
    w["sigmasq"] = sigmasq
    if return_squares:
        w["K"] = K
        if vcov_est:
            w["vcov.est.c"] = (y_init_sd ** 2) * V                 # :438
            w["vcov.est.fitted"] = (y_init_sd ** 2) * Vyhat        # :445
    elif vcov_est:
        w["vcov.c.diag.std"] = np.diag(V).copy()
        w["vcov.fitted.diag.std"] = np.diag(Vyhat).copy()
    w["derivative.call"] = derivative
    w["_eig"] = eig
    return w


# --------------------------------------------------------------------------
# predict (R/bigKRLS.R:547-637)
# --------------------------------------------------------------------------
def neffective_literal(X: np.ndarray) -> float:
    """xBigNeffective, src/Neffective.cpp:13-65: rows de-meaned and normalised (:29-44), r = sum over
    i > j of |z_i . z_j| (:52-55; `abs` of a double), Neffective = N (1 - 2r/N^2) + 1 (:61-64)."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    Z = X - X.mean(axis=1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        Z = Z / np.sqrt((Z * Z).sum(axis=1, keepdims=True))
    r = 0.0
    for i0 in range(0, n, 1024):                      # blocks of rows of the strict lower triangle
        G = np.abs(Z[i0:i0 + 1024] @ Z.T)
        rows = np.arange(i0, min(n, i0 + 1024))[:, None]
        r += float(np.where(np.arange(n)[None, :] < rows, G, 0.0).sum())
    return n * (1.0 - 2.0 * r / float(n) ** 2) + 1.0


def summary_tables(obj: Dict[str, object], degrees: str = "Neffective", probs=(0.05, 0.25, 0.5, 0.75, 0.95)):
    """The numbers of summary.bigKRLS (R/bigKRLS.R:666-757): est, se (rescaled by N/n unless
    degrees == "Neffective", :722-724), t, p = 2 pt(|t|, n - p, lower=FALSE) (:726), and the
    percentiles of the pointwise marginal effects (:741-742, R quantile type 7)."""
    from scipy import stats
    X = np.asarray(obj["X"], dtype=np.float64)
    N, p = X.shape
    n = float(N)
    if degrees == "Neffective":
        n = float(obj["Neffective"])
    elif degrees == "acf":
        Xs = (X - X.mean(axis=0)) / X.std(axis=0, ddof=1)
        n = neffective_literal(Xs)
    est = np.asarray(obj["avgderivatives"], dtype=np.float64).ravel()
    se = np.sqrt(np.asarray(obj["var.avgderivatives"], dtype=np.float64).ravel())
    if degrees != "Neffective":
        se = se * N / n
    tval = est / se
    pval = 2.0 * stats.t.sf(np.abs(tval), n - p)
    D = np.asarray(obj["derivatives"], dtype=np.float64)
    return {"ttests": np.column_stack([est, se, tval, pval]),
            "percentiles": np.quantile(D, list(probs), axis=0).T, "n": n}


def predict(obj: Dict[str, object], newdata: np.ndarray, se_pred=False, correct_se=True):
    X = np.asarray(obj["X"], dtype=np.float64)
    newdata = np.array(newdata, dtype=np.float64)
    if X.shape[1] != newdata.shape[1]:
        raise ValueError("ncol(newdata) differs from ncol(X) from fitted bigKRLS object")
    xm = X.mean(axis=0)                                            # :590
    xs = X.std(axis=0, ddof=1)                                     # :591
    Xs = (X - xm) / xs                                             # :593-594
    nd = (newdata - xm) / xs                                       # :596-597
    newdataK = temp_kernel_literal(nd, Xs, float(obj["sigma"]))    # :599
    ypred = newdataK @ np.asarray(obj["coeffs"])                   # :601
    vcov_pred = se = None
    yv = np.asarray(obj["y"])
    if se_pred:
        if obj.get("vcov.est.c") is None:
            raise ValueError("recompute bigKRLS object with bigKRLS(,vcov.est=TRUE) to compute standard errors")
        vy = r_var(yv)
        vraw = np.asarray(obj["vcov.est.c"]) * (1.0 / vy)
        vcov_pred = vy * tcrossprod(newdataK @ vraw, newdataK)     # :608
        if correct_se and obj.get("Neffective") is not None:
            vcov_pred = math.sqrt(X.shape[0] / float(obj["Neffective"])) * vcov_pred  # :610-611 (Q10)
        se = np.sqrt(np.diag(vcov_pred))                           # :613
    ypred = ypred * r_sd(yv) + float(yv.mean())                    # :621
    return {"predicted": ypred, "se.pred": se, "vcov.est.pred": vcov_pred,
            "newdata": newdata, "newdataK": newdataK}


# --------------------------------------------------------------------------
# cross-validation statistics with explicit index sets (R/bigKRLS.R:1146-1336)
# --------------------------------------------------------------------------
def crossvalidate_split(y, X, train_idx: Sequence[int], test_idx: Sequence[int], **fit_args):
    """ptesting branch (R/bigKRLS.R:1172-1226) with the index sets supplied
    explicitly (0-based) instead of R's sample()."""
    y = np.asarray(y, dtype=np.float64).ravel()
    X = np.asarray(X, dtype=np.float64)
    tr = np.asarray(train_idx)
    te = np.asarray(test_idx)
    marginals = fit_args.get("derivative", True)
    trained = fit(y[tr], X[tr], **fit_args)
    tested = predict(trained, X[te])
    ytest = y[te]
    out = {"trained": trained, "tested": tested}
    out["pseudoR2_is"] = trained["R2"]
    out["pseudoR2_oos"] = r_cor(tested["predicted"], ytest) ** 2                     # :1195
    out["MSE_oos"] = float(np.mean((tested["predicted"] - ytest) ** 2))             # :1196
    out["MSE_is"] = float(np.mean((trained["yfitted"] - trained["y"]) ** 2))        # :1197
    if marginals:
        out["pseudoR2AME_is"] = trained["R2AME"]
        delta = np.asarray(trained["avgderivatives"]).ravel()
        out["MSE_AME_is"] = float(np.mean((trained["y"] - trained["X"] @ delta) ** 2))  # :1206
        yhat_ame = X[te] @ delta
        out["pseudoR2AME_oos"] = r_cor(ytest, yhat_ame) ** 2                        # :1212
        out["MSE_AME_oos"] = float(np.mean((ytest - yhat_ame) ** 2))                # :1213
    return out


def crossvalidate_kfolds(y, X, folds: Sequence[int], **fit_args):
    """Kfolds branch (R/bigKRLS.R:1228-1334) with the fold label of every row
    supplied explicitly (1..Kfolds) instead of cut(sample(N))."""
    folds = np.asarray(folds)
    kf = int(folds.max())
    out = {"Kfolds": kf, "R2_is": [], "R2_oos": [], "MSE_is": [], "MSE_oos": [],
           "R2AME_is": [], "R2AME_oos": [], "MSE_AME_is": [], "MSE_AME_oos": []}
    marginals = fit_args.get("derivative", True)
    for k in range(1, kf + 1):
        tr = np.nonzero(folds != k)[0]
        te = np.nonzero(folds == k)[0]
        r = crossvalidate_split(y, X, tr, te, **fit_args)
        out[f"fold_{k}"] = r
        out["R2_is"].append(r["pseudoR2_is"])
        out["R2_oos"].append(r["pseudoR2_oos"])
        out["MSE_is"].append(r["MSE_is"])
        out["MSE_oos"].append(r["MSE_oos"])
        if marginals:
            out["R2AME_is"].append(r["pseudoR2AME_is"])
            out["R2AME_oos"].append(r["pseudoR2AME_oos"])
            out["MSE_AME_is"].append(r["MSE_AME_is"])
            out["MSE_AME_oos"].append(r["MSE_AME_oos"])
    return out
