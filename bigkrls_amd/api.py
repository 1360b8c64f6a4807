"""bigKRLS(), predict(), crossvalidate(): host-side mirror of the reference's R API
(R/bigKRLS.R:97-516, 547-637, 1146-1336) over HIP device buffers.

The control flow, argument names (dots become underscores), defaults, validation
messages and output fields follow the R functions so that the parity tests read
like the reference's own; every N x N object lives in HBM as a DeviceMatrix and
every numeric step is a HIP kernel behind the C ABI (include/bigkrls.h).
"""
from __future__ import annotations

import math
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

import ctypes as C

from . import _lib, ops
from ._lib import i64
from .device import Context, DeviceMatrix, is_device_matrix

_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def _sd(v) -> float:
    return float(np.std(np.asarray(v, dtype=np.float64), ddof=1))


def _var(v) -> float:
    return float(np.var(np.asarray(v, dtype=np.float64), ddof=1))


def _cor(a, b) -> float:
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    a0, b0 = a - a.mean(), b - b.mean()
    return float((a0 @ b0) / math.sqrt((a0 @ a0) * (b0 @ b0)))


class BigKRLS(dict):
    """The list `w` returned by bigKRLS() (R/bigKRLS.R:420-469), class "bigKRLS"."""

    r_class = "bigKRLS"


class BigKRLSPredicted(dict):
    r_class = "bigKRLS_predicted"


class BigKRLSCV(dict):
    """crossvalidate.bigKRLS's output (R/bigKRLS.R:1330-1336), class "bigKRLS_CV"."""

    r_class = "bigKRLS_CV"


def _as_host_matrix(X) -> np.ndarray:
    if is_device_matrix(X):
        return X.to_numpy()
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    return X


def _call_native(name, *args):
    """A library call whose BIGKRLS_EINVAL (the reference's validation errors, with R's message
    text) becomes the ValueError the R-mirroring API raises; BIGKRLS_ENOMEM is retried once after
    torch's cached-but-unused device blocks were released; everything else stays a BigKRLSError."""
    for attempt in (0, 1):
        try:
            _lib.call(name, *args)
            return
        except _lib.BigKRLSError as e:
            if e.code == _lib.EINVAL:
                raise ValueError(str(e).split(": ", 2)[-1]) from None
            if e.code == _lib.ENOMEM and attempt == 0:
                # the library's workspace comes from hipMalloc; blocks that torch's caching allocator
                # holds but does not use are invisible to it: hand them back to the driver and retry once
                import torch
                torch.cuda.empty_cache()
                continue
            raise


def bigKRLS(y=None, X=None, sigma=None, derivative=True, which_derivatives=None, vcov_est=True,
            Neig=None, eigtrunc=None, lambda_=None, L=None, U=None, tol=None,
            model_subfolder_name=None, overwrite_existing=False, Ncores=None,
            acf=False, noisy=None, instructions=True, ctx: Optional[Context] = None,
            timings: Optional[Dict[str, float]] = None,
            trace: Optional[list] = None, comm=None, keep_outputs: bool = True) -> BigKRLS:
    """Kernel-regularised least squares fit (R/bigKRLS.R:97-516).

    The numeric body -- validation of the data, standardisation, the five steps and the rescaling
    (R/bigKRLS.R:175-470) -- is ONE call into the C ABI, `bigkrls_fit` (include/bigkrls.h,
    csrc/fit.hip); this function checks the argument types, allocates the outputs (the reference's
    ownership rule: the caller allocates, native code writes in place) and assembles the list `w`.
    `which_derivatives` is 1-based like R.  `lambda_` is R's `lambda`.  `model_subfolder_name` /
    `overwrite_existing` (R/bigKRLS.R:111-133, :471-504): the fitted object is also written to that
    folder -- never silently into an existing one -- exactly as save_bigKRLS() does (device-resident
    matrices as text, the rest as estimates.RData), and `w["path"]` records where.  `Ncores` (PSOCK
    workers of the derivative loop, :337-363) and `instructions` are accepted and have no effect: the
    marginal effects of all columns are one pass over K on the GPU.  `timings`
    (optional dict) receives per-phase seconds measured with HIP events on the context's stream;
    `trace` (optional list) the (lambda, Le) probes of the golden-section search.
    `comm` (a bigkrls_amd.dist.Comm): the fit runs over the ranks of that communicator -- every process calls
    with the same y, X and arguments -- through `bigkrls_fit_dist`; the N x N outputs are then this rank's
    column blocks `K.cols`, `vcov.est.c.cols`, `vcov.est.fitted.cols` (n x (r1 - r0), rows `w["rows"]`),
    or not kept at all with keep_outputs=False.
    """
    ctx = (comm.ctx if comm is not None else ctx) or default_context()
    if X is None or y is None:
        raise ValueError("y and X are required")
    if model_subfolder_name is not None and not isinstance(model_subfolder_name, str):        # :112
        raise TypeError("model_subfolder_name must be a character string")
    return_big_rectangles = is_device_matrix(X)                                  # :149
    # column-major like R / the C ABI
    Xh = np.array(_as_host_matrix(X), dtype=np.float64, order="F")
    yh = np.ascontiguousarray(np.array(_as_host_matrix(y), dtype=np.float64).ravel())
    n, p = Xh.shape
    return_big_squares = return_big_rectangles or n > 2500                       # :150
    w = BigKRLS()
    w["has.big.matrices"] = bool(return_big_squares or return_big_rectangles)
    noisy = (n > 2000) if noisy is None else bool(noisy)                          # :153
    xlabs = [f"x{i + 1}" for i in range(p)]                                       # :167
    # ---- argument checks the C ABI cannot see (its "unset" is <= 0 / NULL) ----------------------
    if eigtrunc is not None and (not np.isscalar(eigtrunc) or eigtrunc < 0 or eigtrunc > 1):   # :195-201
        raise ValueError("eigtrunc must be between 0 (no truncation) and 1 (keep largest only).")
    if which_derivatives is not None:                                             # :206-215
        if not derivative:
            raise ValueError("which.derivative requires derivative = TRUE")
        which_derivatives = [int(i) for i in which_derivatives]
        if not which_derivatives or not all(1 <= i <= p for i in which_derivatives):
            raise ValueError("which.derivatives must index columns of X")
    if n != yh.shape[0]:
        raise ValueError("nrow(X) not equal to number of elements in y.")
    if lambda_ is not None and not (np.isscalar(lambda_) and lambda_ > 0):        # :225
        raise ValueError("lambda must be a positive scalar")
    if sigma is not None and not (np.isscalar(sigma) and sigma > 0):              # :227
        raise ValueError("sigma must be a positive scalar")
    if tol is not None and not (np.isscalar(tol) and tol > 0):                    # :232-236 (validated, never forwarded: :274-275)
        raise ValueError("tol must be a positive scalar")
    if U is not None and not (np.isscalar(U) and U > 0):
        raise ValueError("U must be a positive scalar")
    if L is not None and not (np.isscalar(L) and L >= 0):
        raise ValueError("L must be a non-negative scalar")
    if Neig is not None and int(Neig) < 1:
        raise ValueError("Neig must be a positive integer")
    neig = min(n, int(Neig)) if Neig is not None else n                           # :194
    pd = 0 if not derivative else (p if which_derivatives is None else len(which_derivatives))

    opt = _lib.FitOptions()
    opt.struct_bytes = C.sizeof(_lib.FitOptions)
    opt.sigma = -1.0 if sigma is None else float(sigma)
    opt.lambda_ = -1.0 if lambda_ is None else float(lambda_)
    opt.L = -1.0 if L is None else float(L)
    opt.U = -1.0 if U is None else float(U)
    opt.eigtrunc = -1.0 if eigtrunc is None else float(eigtrunc)
    opt.neig = neig
    opt.derivative = int(bool(derivative))
    opt.vcov_est = int(bool(vcov_est))
    opt.acf = int(bool(acf))
    which_arr = None
    if which_derivatives is not None:
        which_arr = np.ascontiguousarray(which_derivatives, dtype=np.int64)
        opt.which_derivatives = which_arr.ctypes.data_as(_lib.pi64)
        opt.n_which = which_arr.size

    def hbuf(*shape):
        return np.empty(shape, dtype=np.float64, order="F")

    out = _lib.FitOutputs()
    out.struct_bytes = C.sizeof(_lib.FitOutputs)
    vals, coeffs, yf, yfs = hbuf(neig), hbuf(n), hbuf(n), hbuf(n)
    isbin = np.zeros(p, dtype=np.int32)
    max_trace = 512
    tracebuf = hbuf(2 * max_trace)
    out.eigenvalues, out.coeffs = vals.ctypes.data, coeffs.ctypes.data
    out.yfitted, out.yfitted_std = yf.ctypes.data, yfs.ctypes.data
    out.binaryindicator = isbin.ctypes.data
    out.lambda_trace, out.max_trace = tracebuf.ctypes.data, max_trace
    if derivative:
        D, Dstd = hbuf(n, pd), hbuf(n, pd)
        avg, var, varstd = hbuf(pd), hbuf(pd), hbuf(pd)
        out.derivatives, out.derivatives_std = D.ctypes.data, Dstd.ctypes.data
        out.avgderivatives, out.var_avgderivatives = avg.ctypes.data, var.ctypes.data
        out.var_avgderivatives_std = varstd.ctypes.data
    r0, r1 = 0, n
    if comm is not None:
        a0, a1 = i64(0), i64(0)
        _call_native("bigkrls_fit_dist_rows", comm.handle, n, C.byref(opt), C.byref(a0), C.byref(a1))
        r0, r1 = int(a0.value), int(a1.value)
    ncols = r1 - r0                                                               # columns of K this process holds
    K = vcovmatc = vcovmatyhat = None
    if keep_outputs or comm is None:
        K = ctx.empty(n, max(ncols, 1))                                           # :434
        out.d_K = K.ptr
        if vcov_est:
            vcovmatc, vcovmatyhat = ctx.empty(n, max(ncols, 1)), ctx.empty(n, max(ncols, 1))
            out.d_vcov_c, out.d_vcov_fitted = vcovmatc.ptr, vcovmatyhat.ptr

    t_wall0 = time.perf_counter()
    if comm is None:
        _call_native("bigkrls_fit", ctx.handle, Xh.ctypes.data, yh.ctypes.data, n, p, C.byref(opt), C.byref(out))
    else:
        _call_native("bigkrls_fit_dist", comm.handle, Xh.ctypes.data, yh.ctypes.data, n, p, C.byref(opt), C.byref(out))
    t_native = time.perf_counter() - t_wall0

    if trace is not None:
        for i in range(min(int(out.n_probes), max_trace)):
            trace.append((float(tracebuf[2 * i]), float(tracebuf[2 * i + 1])))
    w["X"] = Xh
    w["K.eigenvalues"] = vals                                                     # :268
    w["lastkeeper"] = int(out.lastkeeper)                                         # :269
    w["Neffective"] = float(out.Neffective)                                       # :280
    if derivative:
        w["derivatives.std"] = Dstd
        w["var.avgderivatives.std"] = varstd
        w["R2AME"] = float(out.R2AME)                                             # :392
    w["Neffective.acf"] = float(out.Neffective_acf) if (acf and p > 2) else None  # :412-416, :431
    w["coeffs"] = coeffs                                                          # :420
    w["y"] = yh
    w["sigma"] = float(out.sigma)
    w["lambda"] = float(out.lambda_)
    w["binaryindicator"] = isbin.astype(bool)
    w["which.derivatives"] = which_derivatives
    w["xlabs"] = xlabs
    w["yfitted.std"] = yfs
    w["yfitted"] = yf                                                             # :428
    w["R2"] = float(out.R2)                                                       # :429
    w["Looe"] = float(out.Looe)                                                   # :430
    w["Le"] = float(out.Le)
    w["sigmasq"] = float(out.sigmasq) if vcov_est else None
    if comm is not None:
        w["rows"] = (r0, r1)
        if keep_outputs:
            cut = (lambda m: m if ncols > 0 else None)
            w["K.cols"] = cut(K)
            w["vcov.est.c.cols"] = cut(vcovmatc) if vcov_est else None
            w["vcov.est.fitted.cols"] = cut(vcovmatyhat) if vcov_est else None
    else:
        w["K"] = K if return_big_squares else K.to_numpy()                        # :434
        if vcov_est:
            w["vcov.est.c"] = vcovmatc if return_big_squares else vcovmatc.to_numpy()          # :438
            w["vcov.est.fitted"] = vcovmatyhat if return_big_squares else vcovmatyhat.to_numpy()   # :445
        else:
            w["vcov.est.c"] = None
            w["vcov.est.fitted"] = None
    w["derivative.call"] = derivative
    if derivative:
        w["avgderivatives"] = avg[None, :]                                        # :400
        w["var.avgderivatives"] = var[None, :]                                    # :403-407
        w["derivatives"] = D
    if timings is not None:
        for name, sec in zip(_lib.PHASES, out.phase_s):
            timings[name] = float(sec)
        timings["native"] = t_native
        timings["wall"] = time.perf_counter() - t_wall0
    w["_ctx"] = ctx
    if model_subfolder_name is not None:                                          # :471-504
        from .persist import save_bigKRLS
        if comm is None:
            save_bigKRLS(w, model_subfolder_name, overwrite_existing=overwrite_existing, noisy=noisy)
        elif comm.rank == 0:
            # every rank holds the same small outputs: ONE rank writes them (several would race for the folder name);
            # the sharded column blocks are not members load_bigKRLS knows and stay on the GPUs
            small = BigKRLS({k: v for k, v in w.items() if not k.endswith(".cols")})
            save_bigKRLS(small, model_subfolder_name, overwrite_existing=overwrite_existing, noisy=noisy)
            w["path"], w["model_subfolder_name"] = small["path"], small["model_subfolder_name"]
    return w


def _betacf(a: float, b: float, x: float) -> float:
    """Continued fraction of the regularised incomplete beta function (modified Lentz)."""
    tiny = 1e-300
    qab, qap, qam = a + b, a + 1.0, a - 1.0
    c, d = 1.0, 1.0 - qab * x / qap
    d = 1.0 / (d if abs(d) > tiny else tiny)
    h = d
    for m in range(1, 500):
        m2 = 2 * m
        aa = m * (b - m) * x / ((qam + m2) * (a + m2))
        d = 1.0 + aa * d
        d = 1.0 / (d if abs(d) > tiny else tiny)
        c = 1.0 + aa / c
        c = c if abs(c) > tiny else tiny
        h *= d * c
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2))
        d = 1.0 + aa * d
        d = 1.0 / (d if abs(d) > tiny else tiny)
        c = 1.0 + aa / c
        c = c if abs(c) > tiny else tiny
        delta = d * c
        h *= delta
        if abs(delta - 1.0) < 1e-16:
            break
    return h


def _pt_upper(t: float, df: float) -> float:
    """P(T > t) for Student's t with df degrees of freedom, t >= 0 (R: pt(t, df, lower.tail=FALSE)):
    0.5 I_x(df/2, 1/2) with x = df / (df + t^2)."""
    if not np.isfinite(t) or not df > 0:
        return float("nan")
    import math
    x = df / (df + t * t)
    a, b = 0.5 * df, 0.5
    if x <= 0.0:
        return 0.0
    if x >= 1.0:
        return 0.5
    lbeta = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
    front = math.exp(lbeta + a * math.log(x) + b * math.log1p(-x))
    if x < (a + 1.0) / (a + b + 2.0):
        ib = front * _betacf(a, b, x) / a
    else:
        ib = 1.0 - front * _betacf(b, a, 1.0 - x) / b
    return 0.5 * ib


def summary(object: BigKRLS, degrees: str = "Neffective", probs=(0.05, 0.25, 0.5, 0.75, 0.95),
            digits: int = 4, labs=None, quiet: bool = False) -> Optional[dict]:
    """summary.bigKRLS (R/bigKRLS.R:666-757): t-tests of the average marginal effects and the
    percentiles of the pointwise marginal effects. Returns {"ttests": (P' x 4), "percentiles":
    (P' x len(probs)), "rownames": [...]}; prints the R text unless `quiet`."""
    if not isinstance(object, BigKRLS):
        raise TypeError("Object not of class 'bigKRLS'")
    if degrees not in ("acf", "Neffective", "N"):                                 # :677
        raise ValueError('degrees must be one of "acf", "Neffective", "N"')
    Xh = np.asarray(object["X"], dtype=np.float64)
    N = n = Xh.shape[0]
    if degrees == "Neffective":                                                   # :679-681
        n = object["Neffective"]
    if degrees == "acf":                                                          # :682-691
        if object.get("Neffective.acf") is None:
            ctx = object.get("_ctx") or default_context()
            Xs = (Xh - Xh.mean(axis=0)) / Xh.std(axis=0, ddof=1)                  # scale(object$X[])
            n = ops.bNeffective(ctx.from_numpy(Xs))
        else:
            n = object["Neffective.acf"]
    say = (lambda *a: None) if quiet else (lambda *a: print(*a))
    say("\n\nMODEL SUMMARY:\n")
    say("lambda:", round(object["lambda"], digits))
    say("N:", N)
    if n != N:
        say("N Effective:", n)
    p = Xh.shape[1]
    say("R2:", round(float(object["R2"]), digits))
    if object.get("derivatives") is None:                                         # :700-703
        say("\nrecompute with bigKRLS(..., derivative = TRUE) for estimates of marginal effects\n")
        return None
    if object.get("R2AME") is not None:
        say("R2AME**:", round(float(object["R2AME"]), digits), "\n")
    if labs is not None:                                                          # :708-714
        if len(labs) != p:
            raise ValueError("length(labs) must equal ncol(X)")
        names = list(labs)
    else:
        names = list(object["xlabs"])
    which = object.get("which.derivatives") or list(range(1, p + 1))              # :716-718
    est = np.asarray(object["avgderivatives"], dtype=np.float64).ravel()          # :720
    se = np.sqrt(np.asarray(object["var.avgderivatives"], dtype=np.float64).ravel())
    if degrees != "Neffective":                                                   # :722-724
        se = se * N / n
    tval = est / se
    pval = np.array([2.0 * _pt_upper(abs(t), n - p) for t in tval])              # :726
    AME = np.column_stack([est, se, tval, pval])
    isbin = np.asarray(object["binaryindicator"], dtype=bool)
    rown = [names[i - 1] + ("*" if isbin[i - 1] else "") for i in which]          # :729-733
    deriv = np.asarray(object["derivatives"], dtype=np.float64).reshape(N, len(which))
    qderiv = np.quantile(deriv, list(probs), axis=0).T                            # R quantile type 7
    say("Average Marginal Effects:\n")
    say("%-12s %12s %12s %12s %12s" % ("", "Estimate", "Std. Error", "t value", "Pr(>|t|)"))
    for nm, row in zip(rown, np.round(AME, digits)):
        say("%-12s %12g %12g %12g %12g" % (nm, *row))
    say("\n\nPercentiles of Marginal Effects:\n")
    say("%-12s " % "" + " ".join("%11s%%" % (100 * q) for q in probs))
    for nm, row in zip(rown, np.round(qderiv, digits)):
        say("%-12s " % nm + " ".join("%12g" % v for v in row))
    if isbin.any():
        say("\n(*) Reported average and percentiles of dy/dx is for discrete change of the dummy "
            "variable from min to max (usually 0 to 1)).\n")
    say("\n(**) Pseudo-R^2 computed using only the Average Marginal Effects.")
    return {"ttests": AME, "percentiles": qderiv, "rownames": rown,
            "colnames": ["Estimate", "Std. Error", "t value", "Pr(>|t|)"], "n": n}


def predict(object: BigKRLS, newdata, se_pred=False, correct_SE=True, ytest=None,
            ctx: Optional[Context] = None) -> BigKRLSPredicted:
    """predict.bigKRLS (R/bigKRLS.R:547-637); the numeric body (:590-621) is ONE call into the
    C ABI, `bigkrls_predict` (include/bigkrls.h, csrc/fit.hip)."""
    if not isinstance(object, BigKRLS):
        raise TypeError("Object not of class 'bigKRLS'")
    if se_pred and object.get("vcov.est.c") is None:
        raise ValueError("recompute bigKRLS object with bigKRLS(,vcov.est=TRUE) to compute standard errors")
    ctx = ctx or object.get("_ctx") or default_context()
    Xh = np.asfortranarray(np.asarray(object["X"], dtype=np.float64))
    bigmatrix_in = is_device_matrix(newdata) or bool(object["has.big.matrices"])   # :582
    nd_init = _as_host_matrix(newdata)
    nd = np.array(nd_init, dtype=np.float64, order="F")
    if Xh.shape[1] != nd.shape[1]:
        raise ValueError("ncol(newdata) differs from ncol(X) from fitted bigKRLS object")
    n, p = Xh.shape
    u = nd.shape[0]
    yv = np.ascontiguousarray(np.asarray(object["y"], dtype=np.float64).ravel())
    coeffs = np.ascontiguousarray(np.asarray(object["coeffs"], dtype=np.float64).ravel())
    ypred = np.empty(u)
    newdataK = ctx.empty(u, n)
    se = vcov_est_pred = Vd = None
    neff = -1.0
    if se_pred:
        V = object["vcov.est.c"]
        Vd = V if is_device_matrix(V) else ctx.from_numpy(np.asarray(V))
        vcov_est_pred = ctx.empty(u, u)
        se = np.empty(u)
        if correct_SE and object.get("Neffective") is not None:                   # :610-611
            neff = float(object["Neffective"])
    _call_native("bigkrls_predict", ctx.handle, Xh.ctypes.data, n, p, yv.ctypes.data, coeffs.ctypes.data,
                 float(object["sigma"]), nd.ctypes.data, u, Vd.ptr if se_pred else None, neff,
                 ypred.ctypes.data, se.ctypes.data if se_pred else None, newdataK.ptr,
                 vcov_est_pred.ptr if se_pred else None)
    if not bigmatrix_in:                                                          # :623-626
        vcov_est_pred = None if vcov_est_pred is None else vcov_est_pred.to_numpy()
        newdataK = newdataK.to_numpy()
    out = BigKRLSPredicted(predicted=ypred, newdata=nd_init, newdataK=newdataK, ytest=ytest)
    out["se.pred"] = se
    out["vcov.est.pred"] = vcov_est_pred
    out["has.big.matrices"] = bigmatrix_in
    return out


def _run_folds(jobs, contexts, fit_fn, predict_fn):
    """Run independent (train, test) jobs, one worker thread per context (== per GPU), and return
    the results in job order. Fold k goes to context k mod G: a fixed assignment, and because every
    kernel is deterministic the result of a fold does not depend on which GPU computed it. The
    native calls release the GIL (ctypes), a context serves one thread at a time (include/bigkrls.h,
    "Threading"), and HIP's current device is per thread -- no process is spawned, so this is safe
    in a process that has already initialised the GPU."""
    import threading
    results = [None] * len(jobs)
    errors = []

    def worker(slot):
        cx = contexts[slot]
        try:
            if hasattr(cx, "torch"):
                cx.torch.cuda.set_device(cx.device_index)          # thread-local current device
            for j in range(slot, len(jobs), len(contexts)):
                results[j] = jobs[j](cx, fit_fn, predict_fn)
        except BaseException as e:                                 # re-raised in the caller's thread
            errors.append(e)

    if len(contexts) == 1:
        worker(0)
    else:
        threads = [threading.Thread(target=worker, args=(g,), name=f"bigkrls-fold-gpu{g}") for g in range(len(contexts))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return results


def _fold_contexts(ctx, devices, folds_per_device=1):
    """One context per requested GPU. `devices`: None (the given / default context only), "all",
    or a list of device indices (or of ready-made contexts). `folds_per_device` = 2 adds a second context
    with a stream of its own on every device (two folds side by side per GPU)."""
    if folds_per_device not in (1, 2):
        # measured (tools/cv_concurrency.py, N = 4000): two at a time 25 ms per fit against 41 one after the other;
        # three 37 (an odd split of the GPU between the persistent kernels), four 22-28: the second context is where
        # the gain is, and every further one costs its own N x N workspaces
        raise ValueError("folds_per_device must be 1 or 2")
    if folds_per_device == 2:
        base = _fold_contexts(ctx, devices)
        return base + [Context(c.device_index, own_stream=True) for c in base]
    if devices is None:
        return [ctx or default_context()]
    import torch
    if devices == "all":
        devices = list(range(torch.cuda.device_count()))
    devices = list(devices)
    if not devices:
        raise ValueError("devices must name at least one GPU")
    out = []
    for d in devices:
        if not isinstance(d, (int, np.integer)):                   # a ready-made context
            out.append(d)
        elif ctx is not None and ctx.device_index == int(d) and ctx not in out:
            out.append(ctx)
        else:
            out.append(Context(int(d)))
    return out


def crossvalidate(y, X, seed=None, Kfolds=None, ptesting=None, train_idx=None, folds=None,
                  ctx: Optional[Context] = None, devices=None, folds_per_device=1, **fit_args) -> BigKRLSCV:
    """crossvalidate.bigKRLS (R/bigKRLS.R:1146-1336).

    R partitions with set.seed(seed); sample() (:1168,1179,1232), a stream that
    cannot be reproduced without R, so the partition can be supplied explicitly:
    `train_idx` (0-based rows, ptesting branch) or `folds` (label 1..Kfolds per
    row, Kfolds branch).  Without them numpy's default_rng(seed) draws one.

    `devices` (None, "all" or a list of GPU indices): the folds of the Kfolds branch are whole,
    independent fits (the loop at :1268-1282), so they run as replicas, one fold per GPU at a time
    (last row): one context and one worker thread per GPU, no data-path
    collective. The statistics are identical to the sequential loop's.
    `folds_per_device` = 2 runs two folds side by side on every GPU (a second context with its own stream and
    worker thread): fold-sized fits are bound by latency chains that leave most of the GPU idle -- eight fits of
    N = 4000 take 0.33 s one after the other and 0.20 s two at a time (`tools/cv_concurrency.py`); results bitwise
    those of the sequential loop (tests/test_gpu_fit.py, two contexts). More than two gain little (see _fold_contexts) and are refused.
    """
    if (Kfolds is None) + (ptesting is None) != 1:
        raise ValueError("Specify either Kfolds or ptesting but not both.")
    Xh = _as_host_matrix(X)
    yh = np.asarray(_as_host_matrix(y), dtype=np.float64).ravel()
    N = Xh.shape[0]
    marginals = fit_args.get("derivative", True)
    rng = np.random.default_rng(seed)

    def one_split(tr, te, cx, fit_fn, predict_fn):
        trained = fit_fn(yh[tr], Xh[tr], ctx=cx, **fit_args)
        tested = predict_fn(trained, Xh[te])
        ytest = yh[te]
        tested["ytest"] = ytest
        r = {"trained": trained, "tested": tested}
        r["pseudoR2_is"] = trained["R2"]
        r["pseudoR2_oos"] = _cor(tested["predicted"], ytest) ** 2                 # :1195
        r["MSE_oos"] = float(np.mean((tested["predicted"] - ytest) ** 2))         # :1196
        r["MSE_is"] = float(np.mean((trained["yfitted"] - trained["y"]) ** 2))    # :1197
        if marginals:
            r["pseudoR2AME_is"] = trained["R2AME"]
            delta = np.asarray(trained["avgderivatives"]).ravel()
            r["MSE_AME_is"] = float(np.mean((trained["y"] - trained["X"] @ delta) ** 2))   # :1206
            yhat_ame = Xh[te] @ delta
            r["pseudoR2AME_oos"] = _cor(ytest, yhat_ame) ** 2                     # :1212
            r["MSE_AME_oos"] = float(np.mean((ytest - yhat_ame) ** 2))            # :1213
        return r

    contexts = _fold_contexts(ctx, devices, folds_per_device)
    if ptesting is not None:
        if ptesting < 0 or ptesting > 100:
            raise ValueError("ptesting, the percentage of data to be used for validation, must be between 0 and 100.")
        Ntesting = int(round(N * ptesting / 100.0))
        Ntraining = N - Ntesting
        if train_idx is None:
            train_idx = rng.choice(N, Ntraining, replace=False)
        tr = np.asarray(train_idx)
        te = np.setdiff1d(np.arange(N), tr)
        out = BigKRLSCV(one_split(tr, te, contexts[0], _cv_fit, _cv_predict))
        out.update(type="crossvalidated", seed=seed, ptesting=ptesting,
                   indices={"train.set": tr, "test.set": te})
        return out

    if not (float(Kfolds) > 0 and float(Kfolds) % 1 == 0):
        raise ValueError("Kfolds must be a positive integer")
    Kfolds = int(Kfolds)
    if folds is None:
        perm = rng.permutation(N)
        folds = np.empty(N, dtype=int)
        folds[perm] = (np.arange(N) * Kfolds // N) + 1
    folds = np.asarray(folds)
    out = BigKRLSCV({"type": "KfoldsCV", "Kfolds": Kfolds, "seed": seed, "folds": folds})
    keys = ["R2_is", "R2_oos", "MSE_is", "MSE_oos"]
    if marginals:
        keys += ["R2AME_is", "R2AME_oos", "MSE_AME_is", "MSE_AME_oos"]
    for k in keys:
        out[k] = []

    def job(k):
        tr = np.nonzero(folds != k)[0]
        te = np.nonzero(folds == k)[0]
        return lambda cx, fit_fn, predict_fn: one_split(tr, te, cx, fit_fn, predict_fn)

    results = _run_folds([job(k) for k in range(1, Kfolds + 1)], contexts, _cv_fit, _cv_predict)
    out["devices"] = [getattr(c, "device_index", None) for c in contexts]
    for k, r in zip(range(1, Kfolds + 1), results):
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


# the fit / predict the cross-validation driver calls (module-level so that the CPU tests of the fold
# scheduler can substitute doubles that need no GPU)
_cv_fit = bigKRLS
_cv_predict = predict
