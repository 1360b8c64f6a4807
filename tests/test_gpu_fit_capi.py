"""bigkrls_fit() / bigkrls_predict(): the whole path through ONE C-ABI call each (SURVEY.md
section 8(b)(2); the numeric body of R/bigKRLS.R:175-470 and :590-621), driven with plain ctypes --
no bigkrls_amd.api in between -- against the CPU oracle. Tolerance 1e-6 relative (north_star)."""
import ctypes as C

import numpy as np
import pytest

from oracle import krls_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-6


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def native_fit(ctx, X, y, which=None, want_squares=True, **kw):
    """Minimal caller of bigkrls_fit, as an R shim would be: allocate outputs, one call."""
    from bigkrls_amd import _lib
    lib = _lib.load()
    Xf = np.asfortranarray(X, dtype=np.float64)
    yf = np.ascontiguousarray(y, dtype=np.float64)
    n, p = Xf.shape
    opt = _lib.FitOptions()
    opt.struct_bytes = C.sizeof(_lib.FitOptions)
    opt.sigma, opt.lambda_, opt.L, opt.U, opt.eigtrunc = -1.0, -1.0, -1.0, -1.0, -1.0
    opt.neig, opt.derivative, opt.vcov_est, opt.acf = 0, 1, 1, 0
    for k, v in kw.items():
        setattr(opt, k, v)
    if which is not None:
        warr = np.ascontiguousarray(which, dtype=np.int64)
        opt.which_derivatives = warr.ctypes.data_as(_lib.pi64)
        opt.n_which = warr.size
    pd = (len(which) if which is not None else p) if opt.derivative else 0
    neig = n if opt.neig <= 0 else min(n, opt.neig)
    out = _lib.FitOutputs()
    out.struct_bytes = C.sizeof(_lib.FitOutputs)
    host = {"eigenvalues": np.zeros(neig), "coeffs": np.zeros(n), "yfitted": np.zeros(n), "yfitted_std": np.zeros(n),
            "derivatives": np.zeros((n, pd), order="F"), "derivatives_std": np.zeros((n, pd), order="F"),
            "avgderivatives": np.zeros(pd), "var_avgderivatives": np.zeros(pd), "var_avgderivatives_std": np.zeros(pd),
            "lambda_trace": np.zeros(2 * 256)}
    for k, a in host.items():
        setattr(out, k, a.ctypes.data)
    isbin = np.zeros(p, dtype=np.int32)
    out.binaryindicator = isbin.ctypes.data
    out.max_trace = 256
    dev = {}
    if want_squares:
        for k in ("d_K", "d_vcov_c", "d_vcov_fitted"):
            dev[k] = ctx.empty(n, n)
            setattr(out, k, dev[k].ptr)
    st = lib.bigkrls_fit(ctx.handle, Xf.ctypes.data, yf.ctypes.data, n, p, C.byref(opt), C.byref(out))
    return st, out, host, isbin, dev


def test_bigkrls_fit_single_call_vs_oracle(ctx):
    X, y = orc.synth(600, 5, 71, binary_last=True)
    tr = orc.LambdaTrace(0, 0)
    ref = orc.fit(y, X, literal=True, trace=tr)
    st, out, host, isbin, dev = native_fit(ctx, X, y)
    assert st == 0
    assert out.lastkeeper == ref["lastkeeper"] and out.neig == 600 and out.n_deriv == 5
    assert list(isbin) == [0, 0, 0, 0, 1]
    assert rel(host["eigenvalues"], ref["K.eigenvalues"]) < 1e-11
    assert abs(out.lambda_ - ref["lambda"]) <= TOL * ref["lambda"]
    assert out.n_probes == len(tr.probes)
    for i, (lam, le) in enumerate(tr.probes):                      # quirk Q8: identical probe sequence
        assert abs(host["lambda_trace"][2 * i] - lam) <= 1e-12 * lam
        assert abs(host["lambda_trace"][2 * i + 1] - le) <= 1e-8 * le
    assert rel(host["coeffs"], ref["coeffs"]) < TOL
    assert rel(host["yfitted"], ref["yfitted"]) < TOL
    assert rel(host["yfitted_std"], ref["yfitted.std"]) < TOL
    assert rel(host["derivatives"], ref["derivatives"]) < TOL
    assert rel(host["derivatives_std"], ref["derivatives.std"]) < TOL
    assert rel(host["avgderivatives"], np.ravel(ref["avgderivatives"])) < TOL
    assert rel(host["var_avgderivatives"], np.ravel(ref["var.avgderivatives"])) < TOL
    assert rel(host["var_avgderivatives_std"], ref["var.avgderivatives.std"]) < TOL
    for name, key in [("R2", "R2"), ("R2AME", "R2AME"), ("Looe", "Looe"), ("Le", "Le"), ("Neffective", "Neffective"),
                      ("sigmasq", "sigmasq")]:
        assert abs(getattr(out, name) - ref[key]) <= TOL * abs(ref[key]), name
    assert abs(out.y_mean - y.mean()) < 1e-14 and abs(out.y_sd - y.std(ddof=1)) < 1e-14
    assert rel(dev["d_K"].to_numpy(), ref["K"]) < 1e-12
    assert rel(dev["d_vcov_c"].to_numpy(), ref["vcov.est.c"]) < TOL
    assert rel(dev["d_vcov_fitted"].to_numpy(), ref["vcov.est.fitted"]) < TOL
    assert all(t >= 0 for t in out.phase_s) and sum(out.phase_s) > 0
    # the same call without device outputs: K lives in the library's workspace, V is never formed
    st2, out2, host2, _, _ = native_fit(ctx, X, y, want_squares=False)
    assert st2 == 0 and out2.lambda_ == out.lambda_
    assert np.array_equal(host2["coeffs"], host["coeffs"]) and np.array_equal(host2["derivatives"], host["derivatives"])
    assert np.array_equal(host2["var_avgderivatives"], host["var_avgderivatives"])


def test_bigkrls_fit_options(ctx):
    """Neig < N, eigtrunc, user lambda, which.derivatives (quirk Q6), no derivatives, acf."""
    X, y = orc.synth(400, 5, 72)
    X[:, 1] *= 5.0
    ref = orc.fit(y, X, neig=60, eigtrunc=0.01, which_derivatives=[2, 4], literal=False)
    st, out, host, _, _ = native_fit(ctx, X, y, which=[2, 4], neig=60, eigtrunc=0.01, want_squares=False)
    assert st == 0 and out.neig == 60 and out.n_deriv == 2 and out.lastkeeper == ref["lastkeeper"]
    assert abs(out.lambda_ - ref["lambda"]) <= TOL * ref["lambda"]
    assert rel(host["derivatives"], ref["derivatives"]) < TOL
    assert rel(host["var_avgderivatives"], np.ravel(ref["var.avgderivatives"])) < TOL
    ref2 = orc.fit(y, X, lam=0.25, derivative=False, literal=False)
    st, out, host, _, _ = native_fit(ctx, X, y, lambda_=0.25, derivative=0, want_squares=False)
    assert st == 0 and out.lambda_ == 0.25 and out.n_probes == 0 and out.n_deriv == 0 and np.isnan(out.R2AME)
    assert rel(host["coeffs"], ref2["coeffs"]) < TOL and abs(out.R2 - ref2["R2"]) < TOL
    st, out, _, _, _ = native_fit(ctx, X, y, acf=1, want_squares=False)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    assert st == 0 and abs(out.Neffective_acf - orc.neffective_literal(Xs)) < 1e-9 * 400


def test_bigkrls_fit_validation_messages(ctx):
    """The reference's validation block (R/bigKRLS.R:183-240): BIGKRLS_EINVAL + R's message text."""
    from bigkrls_amd import _lib
    lib = _lib.load()
    X, y = orc.synth(60, 3, 73)
    cases = []
    Xc = X.copy(); Xc[:, 2] = 1.5
    cases.append((Xc, y, {}, b"constant and must be removed: 3"))
    Xn = X.copy(); Xn[5, 1] = np.nan
    cases.append((Xn, y, {}, b"contain missing data, which must be removed: 2"))
    yn = y.copy(); yn[0] = np.nan
    cases.append((X, yn, {}, b"y contains missing data."))
    cases.append((X, np.full(60, 2.0), {}, b"y is a constant."))
    cases.append((X, y, {"eigtrunc": 1.5}, b"eigtrunc must be between 0"))
    cases.append((X, y, {"vcov_est": 0}, b"vcov.est is needed to get derivatives"))
    for Xi, yi, kw, msg in cases:
        st, *_ = native_fit(ctx, Xi, yi, want_squares=False, **kw)
        assert st == _lib.EINVAL and msg in lib.bigkrls_last_error(), (msg, lib.bigkrls_last_error())
    st, *_ = native_fit(ctx, X, y, which=[4], want_squares=False)
    assert st == _lib.EINVAL and b"which.derivatives must index columns of X" in lib.bigkrls_last_error()


def test_bigkrls_predict_single_call_vs_oracle(ctx):
    from bigkrls_amd import _lib
    lib = _lib.load()
    X, y = orc.synth(500, 4, 74)
    Xtr, ytr, Xte = X[:430], y[:430], X[430:]
    ref = orc.fit(ytr, Xtr, literal=False)
    pr = orc.predict(ref, Xte, se_pred=True)
    st, out, host, _, dev = native_fit(ctx, Xtr, ytr)
    assert st == 0
    n, p, u = 430, 4, 70
    Xf, nd = np.asfortranarray(Xtr), np.asfortranarray(Xte)
    ytr_c = np.ascontiguousarray(ytr)
    pred, se = np.zeros(u), np.zeros(u)
    dKn, dVp = ctx.empty(u, n), ctx.empty(u, u)
    st = lib.bigkrls_predict(ctx.handle, Xf.ctypes.data, n, p, ytr_c.ctypes.data, host["coeffs"].ctypes.data,
                             out.sigma, nd.ctypes.data, u, dev["d_vcov_c"].ptr, out.Neffective,
                             pred.ctypes.data, se.ctypes.data, dKn.ptr, dVp.ptr)
    assert st == 0, lib.bigkrls_last_error()
    assert rel(pred, pr["predicted"]) < TOL and rel(se, pr["se.pred"]) < TOL
    assert rel(dKn.to_numpy(), pr["newdataK"]) < 1e-12
    assert rel(dVp.to_numpy(), pr["vcov.est.pred"]) < TOL
    # mean only, nothing kept on the device; and the uncorrected SE (correct.SE = FALSE)
    pred2 = np.zeros(u)
    st = lib.bigkrls_predict(ctx.handle, Xf.ctypes.data, n, p, ytr_c.ctypes.data, host["coeffs"].ctypes.data,
                             out.sigma, nd.ctypes.data, u, None, -1.0, pred2.ctypes.data, None, None, None)
    assert st == 0 and np.array_equal(pred2, pred)
    se_raw = np.zeros(u)
    st = lib.bigkrls_predict(ctx.handle, Xf.ctypes.data, n, p, ytr_c.ctypes.data, host["coeffs"].ctypes.data,
                             out.sigma, nd.ctypes.data, u, dev["d_vcov_c"].ptr, -1.0, pred2.ctypes.data,
                             se_raw.ctypes.data, None, None)
    pr_raw = orc.predict(ref, Xte, se_pred=True, correct_se=False)
    assert st == 0 and rel(se_raw, pr_raw["se.pred"]) < TOL
    st = lib.bigkrls_predict(ctx.handle, Xf.ctypes.data, n, p, ytr_c.ctypes.data, host["coeffs"].ctypes.data,
                             out.sigma, nd.ctypes.data, u, None, -1.0, pred2.ctypes.data, se_raw.ctypes.data, None, None)
    assert st == _lib.EINVAL and b"vcov.est=TRUE" in lib.bigkrls_last_error()


def test_plain_c_caller_of_the_abi(ctx):
    """tests/capi/fit_example.c (C99, gcc, no Python in the call path apart from loading it): context,
    device buffers, one bigkrls_fit, one bigkrls_predict -- against the oracle. Built by
    __graft_entry__.build() (gcc cannot be spawned from a process that holds the GPU)."""
    import os
    from bigkrls_amd import _lib
    _lib.load()
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "capi", "libfit_example.so")
    assert os.path.exists(path), "run python -c 'import __graft_entry__ as g; g.build()' first"
    ex = C.CDLL(path)
    ex.capi_fit_example.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    n, p, u = 350, 4, 25
    X, y = orc.synth(n, p, 75)
    ref = orc.fit(y, X, derivative=False, literal=False)
    Xf, yc = np.asfortranarray(X), np.ascontiguousarray(y)
    coeffs, res = np.zeros(n), np.zeros(8)
    st = ex.capi_fit_example(Xf.ctypes.data, yc.ctypes.data, n, p, u, coeffs.ctypes.data, res.ctypes.data)
    assert st == 0, _lib.load().bigkrls_last_error()
    assert abs(res[0] - ref["lambda"]) <= TOL * ref["lambda"]
    assert abs(res[1] - ref["R2"]) <= TOL and abs(res[2] - ref["Le"]) <= TOL * ref["Le"]
    assert int(res[3]) == ref["lastkeeper"]
    assert rel(coeffs, ref["coeffs"]) < TOL
    assert abs(res[4] - np.mean(ref["yfitted"][:u])) <= TOL * abs(np.mean(ref["yfitted"][:u])) + 1e-9
    assert res[5] == 1.0 and abs(res[6] - ref["K"][1, 0]) < 1e-13
