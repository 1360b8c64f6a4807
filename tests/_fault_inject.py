"""The in-call recovery paths of the eigensolver, driven by the fault-injection hooks of the TEST build of the
library (tests/capi/libbigkrls_hip_fault.so, -DBK_FAULT_INJECT; the shipped library has no such hooks).
Run as a script by tests/test_gpu_configs.py: the library path is per process.
BIGKRLS_FAULT=watchdog: the first attempt reports a fired persistent-kernel watchdog after stage 1; the call must
redo the decomposition with the per-step kernels and succeed. BIGKRLS_FAULT=noconv: the block Lanczos reports
non-convergence; the same call must fall through to the dense path (no user-visible switch, like the reference's
eigs_sym branch, src/eigen.cpp:18-22). BIGKRLS_FAULT=dc_lag: the stream lags behind the host inside the divide &
conquer; the result must not change by a bit. BIGKRLS_FAULT=eig_garbage[_always]: a decomposition that comes back
wrong without an error is caught by the fit's check against K and redone once (then an error)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np

import bigkrls_amd._lib as L
L.LIB_PATH = os.path.join(HERE, "capi", "libbigkrls_hip_fault.so")
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def quality(K, vectors, values):
    Kh, Q = K.to_numpy(), vectors.to_numpy()
    d = np.asarray(values)[:Q.shape[1]]
    return (float(np.max(np.abs(Kh @ Q - Q * d)) / abs(d[0])), float(np.max(np.abs(Q.T @ Q - np.eye(Q.shape[1])))))


ctx = bk.Context(0)
n, p = 3000, 5
X, _ = synth(n, p, 9)
K = ops.bGaussKernel(ctx.from_numpy((X - X.mean(0)) / X.std(0, ddof=1)), float(p))
good = ops.bEigen(K, 40, -1.0)
os.environ["BIGKRLS_FAULT"] = "watchdog"
again = ops.bEigen(K, 40, -1.0)
assert rel(again.values, good.values) < 1e-12
res, orth = quality(K, again.vectors, again.values)
assert res < 1e-11 and orth < 1e-11, (res, orth)
# BIGKRLS_FAULT=watchdog_bt2: the watchdog word of the persistent stage-2 back-transform reads as fired after the
# decomposition: the call must redo it with per-wavefront launches
os.environ["BIGKRLS_FAULT"] = "watchdog_bt2"
third = ops.bEigen(K, 40, -1.0)
assert rel(third.values, good.values) < 1e-12
res, orth = quality(K, third.vectors, third.values)
assert res < 1e-11 and orth < 1e-11, (res, orth)
# BIGKRLS_FAULT=dc_lag: the stream runs 2 ms behind the host inside the divide & conquer (a spin kernel ahead of each
# level's descriptor upload) while the host churns its heap: every host vector that is the source of an asynchronous
# copy must stay alive and untouched until the level is synchronised (one of them used to be freed early: harmless on
# this runtime, which copies a pageable source before hipMemcpyAsync returns, but not promised). All eigenvectors:
# the explicit merges.
os.environ["BIGKRLS_FAULT"] = ""
full = ops.bEigen(K, None, -1.0)
os.environ["BIGKRLS_FAULT"] = "dc_lag"
for _ in range(2):
    lag = ops.bEigen(K, None, -1.0)
    assert np.array_equal(np.asarray(lag.values), np.asarray(full.values))
    assert np.array_equal(lag.vectors.to_numpy(), full.vectors.to_numpy())
# BIGKRLS_FAULT=eig_garbage: the first decomposition of a fit comes back WRONG without an error (one kept eigenvector
# scaled by 1.001). The fit checks every decomposition against K itself (trace; two +-1 combinations of ALL kept pairs)
# and redoes it once:
# the result must be the undisturbed one; =eig_garbage_always: the redone one is wrong too -> an error, not a result.
Xf, yf = synth(1500, 5, 77)
os.environ["BIGKRLS_FAULT"] = ""
ref = bk.bigKRLS(yf, Xf, ctx=ctx, eigtrunc=0.001, noisy=False)
os.environ["BIGKRLS_FAULT"] = "eig_garbage"
healed = bk.bigKRLS(yf, Xf, ctx=ctx, eigtrunc=0.001, noisy=False)
assert healed["lastkeeper"] == ref["lastkeeper"] and healed["lambda"] == ref["lambda"]
assert np.array_equal(healed["coeffs"], ref["coeffs"]) and np.array_equal(healed["derivatives"], ref["derivatives"])
os.environ["BIGKRLS_FAULT"] = "eig_garbage_always"
try:
    bk.bigKRLS(yf, Xf, ctx=ctx, eigtrunc=0.001, noisy=False)
    raise AssertionError("a decomposition that fails the check against K twice must be an error")
except L.BigKRLSError as e:
    assert e.code == L.EHIP and "the check against K" in str(e) and "also after the decomposition was redone" in str(e), str(e)
# BIGKRLS_FAULT=eig_swap: two kept eigenvectors come back exchanged -- norms and eigenvalues right, the pairing wrong.
# |Q r|^2 = k still holds, so only the comparison with K Q r sees it: on one GPU that half of the check is deferred to the
# fit's single pass over K (marginal effects + fitted values + K [u_1 u_2]); the fit must then redo everything from the
# decomposition on and return the undisturbed result; =eig_swap_always: an error.
os.environ["BIGKRLS_FAULT"] = "eig_swap"
before = ctx.counters()
healed2 = bk.bigKRLS(yf, Xf, ctx=ctx, eigtrunc=0.001, noisy=False)
assert ctx.counters()["redone"] == before["redone"] + 1, (before, ctx.counters())
assert healed2["lastkeeper"] == ref["lastkeeper"] and healed2["lambda"] == ref["lambda"]
assert np.array_equal(healed2["coeffs"], ref["coeffs"]) and np.array_equal(healed2["derivatives"], ref["derivatives"])
assert np.array_equal(healed2["yfitted"], ref["yfitted"])
os.environ["BIGKRLS_FAULT"] = "eig_swap_always"
try:
    bk.bigKRLS(yf, Xf, ctx=ctx, eigtrunc=0.001, noisy=False)
    raise AssertionError("a decomposition that fails the deferred check against K twice must be an error")
except L.BigKRLSError as e:
    assert e.code == L.EHIP and "the check against K" in str(e) and "also after the decomposition was redone" in str(e), str(e)
# ... and without marginal effects the check is not deferred (its own product K [u_1 u_2]): caught and healed as well
os.environ["BIGKRLS_FAULT"] = ""
ref_nd = bk.bigKRLS(yf, Xf, ctx=ctx, eigtrunc=0.001, noisy=False, derivative=False)
os.environ["BIGKRLS_FAULT"] = "eig_swap"
healed3 = bk.bigKRLS(yf, Xf, ctx=ctx, eigtrunc=0.001, noisy=False, derivative=False)
assert healed3["lambda"] == ref_nd["lambda"] and np.array_equal(healed3["coeffs"], ref_nd["coeffs"])
# BIGKRLS_FAULT=kry_swap: the same with the block Lanczos as the fit's decomposition (Neig << N). Its own sample check
# of the last block of Ritz pairs against K is left out of a fit's first attempt (the fit checks ALL kept pairs), so this
# fault is seen by the fit's check only; the redo runs with the sample check and must return the undisturbed bits.
Xk, yk = synth(16384, 5, 78)
os.environ["BIGKRLS_FAULT"] = ""
ref_k = bk.bigKRLS(yk, Xk, ctx=ctx, Neig=64, eigtrunc=0.001, noisy=False)
for deriv in (True, False):
    os.environ["BIGKRLS_FAULT"] = ""
    ref_kd = ref_k if deriv else bk.bigKRLS(yk, Xk, ctx=ctx, Neig=64, eigtrunc=0.001, noisy=False, derivative=False)
    os.environ["BIGKRLS_FAULT"] = "kry_swap"
    before = ctx.counters()
    healed_k = bk.bigKRLS(yk, Xk, ctx=ctx, Neig=64, eigtrunc=0.001, noisy=False, derivative=deriv)
    assert ctx.counters()["redone"] == before["redone"] + 1, (deriv, before, ctx.counters())
    assert healed_k["lastkeeper"] == ref_kd["lastkeeper"] and healed_k["lambda"] == ref_kd["lambda"]
    assert np.array_equal(healed_k["coeffs"], ref_kd["coeffs"]) and np.array_equal(healed_k["yfitted"], ref_kd["yfitted"])
    if deriv:
        assert np.array_equal(healed_k["derivatives"], ref_k["derivatives"])
os.environ["BIGKRLS_FAULT"] = "noconv"
n2 = 16384                                                   # the size at which Lanczos is chosen by default
X2, _ = synth(n2, p, 10)
K2 = ops.bGaussKernel(ctx.from_numpy((X2 - X2.mean(0)) / X2.std(0, ddof=1)), float(p))
fb = ops.bEigen(K2, 64, -1.0)
# ... and when the dense workspace (4 n^2 doubles) does not fit, the non-convergence is reported as such, with what the
# iteration saw at its checks, instead of an out-of-memory error from inside the dense path
os.environ["BIGKRLS_FAULT_NOFIT"] = "1"
ctx.release_workspace()          # (the workspace the dense fallback just grew counts as held otherwise)
try:
    ops.bEigen(K2, 64, -1.0)
    raise AssertionError("a block Lanczos that does not converge must not succeed when the dense path does not fit")
except L.BigKRLSError as e:
    assert e.code == L.ENOCONV, e
    assert "dense fallback does not fit" in str(e) and "check steps=" in str(e), str(e)
del os.environ["BIGKRLS_FAULT_NOFIT"]
del os.environ["BIGKRLS_FAULT"]
kr = ops.bEigen(K2, 64, -1.0)
assert rel(fb.values, kr.values) < 1e-10
res, orth = quality(K2, fb.vectors, fb.values)
assert res < 1e-11 and orth < 1e-11, (res, orth)
print("fault injection OK")
