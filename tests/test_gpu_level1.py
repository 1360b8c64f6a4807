"""Level-1 C-ABI entry points (host pointers) against the CPU oracle.

Each test is the GPU counterpart of one .Call routine of the reference
(src/RcppExports.cpp:147-160).  Tolerances are relative fp64 and written per test;
north_star asks for 1e-6 on c, yhat, lambda and the derivatives."""
import ctypes as C

import numpy as np
import pytest

from oracle import krls_oracle as orc

pytestmark = pytest.mark.gpu


def F(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def P(a):
    return C.c_void_p(a.ctypes.data)


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def check(lib, status):
    assert status == 0, lib.bigkrls_last_error().decode()


@pytest.mark.parametrize("n,p", [(1, 1), (7, 3), (129, 5), (500, 5), (1000, 20), (777, 50), (300, 150), (513, 129)])
def test_gauss_kernel(lib, n, p):
    X, y = orc.synth(n, p, 1)
    if n > 1:
        Xs, _, _, _, _, _ = orc.standardize(X, y)
    else:
        Xs = X
    Xf = F(Xs)
    out = F(np.zeros((n, n)))
    check(lib, lib.bigkrls_gauss_kernel(P(Xf), n, p, float(p), P(out)))
    ref = orc.gauss_kernel_literal(Xs, float(p))
    assert np.max(np.abs(out - ref)) < 1e-13
    assert np.all(np.diag(out) == 1.0)
    assert np.array_equal(out, out.T)          # bitwise symmetric like the reference's mirror


@pytest.mark.parametrize("band_rows", [8, 5, 1])
def test_kernel_builds_band_major_tile_order_with_several_bands(band_rows):
    """The kernel builds' XCD-aware tile order with band heights of a few tile rows (the default, ~1 MB of X rows, gives a
    single band at every size a CPU oracle reaches): tests/_kb_bands.py in a process of its own, every entry of
    bGaussKernel / bTempKernel against the literal formula (src/gauss_kernel.cpp:13-30, src/temp_kernel.cpp:13-30)."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_kb_bands.py"),
                        str(band_rows)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "band-major tile maps OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_gauss_kernel_mtcars_golden(lib):
    """The reference's own golden vector (tests/testthat/test_basic_usage.R:71-108)."""
    import csv, os
    here = os.path.dirname(__file__)
    rows = list(csv.reader(open(os.path.join(here, "golden", "mtcars.csv"))))
    names = [r[0] for r in rows[1:]]
    M = np.array([[float(v) for v in r[1:]] for r in rows[1:]])
    X = M[:, 1:]
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    out = F(np.zeros((32, 32)))
    Xf = F(Xs)                                  # keep the buffer alive across the call
    check(lib, lib.bigkrls_gauss_kernel(P(Xf), 32, 10, 10.0, P(out)))
    gold = {r[0]: float(r[1]) for r in list(csv.reader(open(os.path.join(here, "golden", "mtcars_corolla_kernel.csv"))))[1:]}
    s = out[:, names.index("Toyota Corolla")]
    diff = np.array([s[i] - gold[nm] for i, nm in enumerate(names)])
    assert diff.max() < 0.01                    # the reference's own (one-sided) criterion
    assert np.abs(diff).max() < 1e-12           # and what it actually achieves


@pytest.mark.parametrize("u,v,p", [(1, 1, 1), (50, 500, 5), (333, 129, 20), (130, 260, 3), (200, 77, 140)])
def test_temp_kernel(lib, u, v, p):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((u, p))
    B = rng.standard_normal((v, p))
    out = F(np.zeros((u, v)))
    Af, Bf = F(A), F(B)
    check(lib, lib.bigkrls_temp_kernel(P(Af), u, P(Bf), v, p, 2.5, P(out)))
    assert np.max(np.abs(out - orc.temp_kernel_literal(A, B, 2.5))) < 1e-13


@pytest.mark.parametrize("shape", [(300, 40, 17), (129, 257, 130), (5, 3, 2), (1000, 64, 64)])
def test_crossprods(lib, shape):
    n, ak, bk = shape
    rng = np.random.default_rng(3)
    A = F(rng.standard_normal((n, ak)))
    B = F(rng.standard_normal((n, bk)))
    out = F(np.zeros((ak, bk)))
    check(lib, lib.bigkrls_crossprod(P(A), n, ak, P(B), bk, P(out)))
    assert relerr(out, A.T @ B) < 1e-13
    out = F(np.zeros((ak, ak)))
    check(lib, lib.bigkrls_xtx(P(A), n, ak, P(out)))
    assert relerr(out, A.T @ A) < 1e-13
    A2 = F(rng.standard_normal((ak, n)))      # an x k
    B2 = F(rng.standard_normal((bk, n)))      # bn x k
    out = F(np.zeros((ak, bk)))
    check(lib, lib.bigkrls_tcrossprod(P(A2), ak, n, P(B2), bk, P(out)))
    assert relerr(out, A2 @ B2.T) < 1e-13
    out = F(np.zeros((ak, ak)))
    check(lib, lib.bigkrls_xxt(P(A2), ak, n, P(out)))
    assert relerr(out, A2 @ A2.T) < 1e-13


def test_gemm_asymmetric_layout(lib, ctx):
    """A = I against an asymmetric B catches a transposed accumulator map."""
    import bigkrls_amd as bk
    n = 200
    B = np.arange(n * n, dtype=np.float64).reshape(n, n) / 7.0
    I = np.eye(n)
    for ta in (False, True):
        for tb in (False, True):
            Bd = ctx.from_numpy(B.T if tb else B)
            out = bk.ops.gemm(ta, tb, ctx.from_numpy(I), Bd).to_numpy()
            assert np.array_equal(out, B), (ta, tb)


def test_multdiag(lib):
    rng = np.random.default_rng(2)
    A = F(rng.standard_normal((300, 70)))
    d = rng.standard_normal(90)                # longer than ncol, only the first 70 used
    out = F(np.zeros((300, 70)))
    check(lib, lib.bigkrls_multdiag(P(A), 300, 70, P(d), P(out)))
    assert np.array_equal(out, orc.multdiag(A, d))


@pytest.mark.parametrize("n,p,trunc", [(200, 3, 0.0), (500, 5, 0.0), (600, 6, 0.01)])
def test_solveforc_matches_literal_row_loop(lib, n, p, trunc):
    X, y = orc.synth(n, p, 7)
    Xs, ys, *_ = orc.standardize(X, y)
    K = orc.gauss_kernel_literal(Xs, float(p))
    eig = orc.b_eigen(K, n, trunc)
    Q = F(eig.vectors)
    k = Q.shape[1]
    for lam in (0.05, 0.7, 13.0):
        le_ref, c_ref = orc.solveforc_literal(eig.vectors, eig.values, ys, lam)
        le = C.c_double()
        c = np.zeros(n)
        vals = np.ascontiguousarray(eig.values)
        yc = np.ascontiguousarray(ys)
        check(lib, lib.bigkrls_solveforc(P(Q), n, k, P(vals), vals.size, P(yc),
                                         lam, C.byref(le), P(c)))
        assert relerr(c, c_ref) < 1e-9
        assert abs(le.value - le_ref) / le_ref < 1e-9


@pytest.fixture(params=["1stage", "2stage"])
def eig_path(request, monkeypatch):
    """Both tridiagonalisation paths: one-stage (symv) and two-stage (band + bulge chasing;
    the default above 256 rows). The library reads BIGKRLS_EIG at every call."""
    if request.param == "1stage":
        monkeypatch.setenv("BIGKRLS_EIG", "1stage")
    else:
        monkeypatch.delenv("BIGKRLS_EIG", raising=False)
    return request.param


@pytest.mark.parametrize("n,p", [(1, 1), (2, 1), (3, 2), (64, 3), (65, 3), (200, 2), (257, 3), (258, 3),
                                 (321, 4), (385, 2), (500, 5), (1000, 10), (1283, 6)])
def test_eigen_full(lib, eig_path, n, p):
    X, y = orc.synth(max(n, 2), p, 9)
    X = X[:n]
    K = orc.gauss_kernel_literal(X, float(p))
    vals = np.zeros(n)
    vecs = F(np.zeros((n, n)))
    Kf = F(K)
    check(lib, lib.bigkrls_eigen(P(Kf), n, n, P(vals), P(vecs)))
    ref_vals, _ = orc.big_eigen_literal(K, n)
    scale = np.abs(ref_vals).max()
    assert np.max(np.abs(vals - ref_vals)) / scale < 1e-12
    assert np.all(np.diff(vals) <= 0)
    assert np.max(np.abs(vecs.T @ vecs - np.eye(n))) < 1e-11
    assert np.max(np.abs(K @ vecs - vecs * vals)) / scale < 1e-11


@pytest.mark.parametrize("n,p,neig", [(300, 4, 10), (500, 5, 50), (400, 3, 399), (900, 7, 64)])
def test_eigen_partial(lib, eig_path, n, p, neig):
    """Neig < N takes the reference's eigs_sym branch (src/eigen.cpp:18-22)."""
    X, y = orc.synth(n, p, 10)
    K = orc.gauss_kernel_literal(X, float(p))
    vals = np.zeros(neig)
    vecs = F(np.zeros((n, neig)))
    Kf = F(K)
    check(lib, lib.bigkrls_eigen(P(Kf), n, neig, P(vals), P(vecs)))
    ref_vals = np.linalg.eigvalsh(K)[::-1][:neig]
    scale = ref_vals[0]
    assert np.max(np.abs(vals - ref_vals)) / scale < 1e-12
    assert np.max(np.abs(vecs.T @ vecs - np.eye(neig))) < 1e-11
    assert np.max(np.abs(K @ vecs - vecs * vals)) / scale < 1e-11


@pytest.mark.parametrize("n", [37, 300])
def test_eigen_diagonal_and_degenerate(lib, eig_path, n):
    """All-deflated merges (diagonal input) and exactly repeated eigenvalues."""
    d = np.linspace(1, 2, n)
    vals = np.zeros(n)
    vecs = F(np.zeros((n, n)))
    Df = F(np.diag(d))
    check(lib, lib.bigkrls_eigen(P(Df), n, n, P(vals), P(vecs)))
    assert np.allclose(vals, d[::-1], rtol=0, atol=1e-15)
    assert np.max(np.abs(vecs.T @ vecs - np.eye(n))) < 1e-13
    # rank-2 projector plus identity: eigenvalues {3,3,1,...,1}
    rng = np.random.default_rng(0)
    Qr, _ = np.linalg.qr(rng.standard_normal((n, 2)))
    A = np.eye(n) + 2 * Qr @ Qr.T
    Af = F(A)
    check(lib, lib.bigkrls_eigen(P(Af), n, n, P(vals), P(vecs)))
    assert np.allclose(vals[:2], 3, atol=1e-13) and np.allclose(vals[2:], 1, atol=1e-13)
    assert np.max(np.abs(vecs.T @ vecs - np.eye(n))) < 1e-12
    assert np.max(np.abs(A @ vecs - vecs * vals)) < 1e-12


@pytest.mark.parametrize("n,p,binary", [(150, 3, False), (300, 4, True), (257, 6, True)])
def test_derivmat_matches_literal(lib, n, p, binary):
    """BigDerivMat drop-in vs the literal N^3 restatement of src/bigderiv_v3.cpp."""
    X, y = orc.synth(n, p, 21, binary_last=binary)
    if binary:
        X[:, 0] = (X[:, 0] > -0.3).astype(float) * 3.0 + 1.0     # a second binary column, other coding
    w = orc.fit(y, X, literal=True)
    Xs, ys, *_ = orc.standardize(X, y)
    K = w["K"]
    V = w["vcov.est.c"] / orc.r_sd(y) ** 2
    D_ref, var_ref = orc.derivmat_literal(Xs, K, V, w["coeffs"], float(p))
    D = F(np.zeros((n, p)))
    var = np.zeros(p)
    Xf, Kf, Vf, cf = F(Xs), F(K), F(V), np.ascontiguousarray(w["coeffs"])
    check(lib, lib.bigkrls_derivmat(P(Xf), n, p, P(Kf), P(Vf), P(D), P(var), P(cf), float(p)))
    assert relerr(D, D_ref) < 1e-10
    assert np.max(np.abs(var - var_ref) / np.abs(var_ref)) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("n,p", [(4200, 16), (4500, 20), (4301, 23)])
def test_deriv_rows_on_the_48_wide_tile(n, p):
    """The one pass over K for all columns, K [1, c, x_j, x_j o c | b_j, b_j o c] (src/bigderiv_v3.cpp:90-106 in O(N^2)),
    on the 128 x 48 tile that 33..48 operand columns (P = 16..23) take at n >= 4096: against the same products in
    numpy (the generic 128 x 64 kernel covers every other column count: test_derivmat_matches_literal, the C2 fit)."""
    import bigkrls_amd as bk
    from bigkrls_amd import ops, _lib
    ctx = bk.Context(0)
    rng = np.random.default_rng(31)
    X = rng.standard_normal((n, p))
    X[:, p - 1] = (X[:, p - 1] > 0.12345).astype(float)              # a binary column: first differences (:31-87)
    c = rng.standard_normal(n) / n
    sigma = float(p)
    dX = ctx.from_numpy(X)
    K = ops.bGaussKernel(dX, sigma)
    Kh = K.to_numpy()
    dc = ctx.from_numpy(c.reshape(n, 1))
    isb = np.zeros(p, dtype=np.int32)
    isb[p - 1] = 1
    D, S = ctx.empty(n, p), ctx.empty(n, p)
    _lib.call("bigkrls_dev_deriv_rows", ctx.handle, K.ptr, n, n, K.ld, 0, dX.ptr, p, dX.ld, isb.ctypes.data, dc.ptr, sigma,
              D.ptr, D.ld, S.ptr, S.ld)
    Dh, Sh = D.to_numpy(), S.to_numpy()
    K1, Kc = Kh.sum(axis=1), Kh @ c
    for j in range(p - 1):                                           # continuous columns (:103, :105)
        x = X[:, j]
        d_ref = (-2.0 / sigma) * (x * Kc - Kh @ (x * c))
        s_ref = x * K1 - Kh @ x
        assert np.max(np.abs(Dh[:, j] - d_ref)) < 1e-12 * max(1.0, np.max(np.abs(d_ref)))
        assert np.max(np.abs(Sh[:, j] - s_ref)) < 1e-11 * max(1.0, np.max(np.abs(s_ref)))
    # the binary column: the group sums of :60-71
    b = X[:, p - 1]
    hi = b == b.max()
    sd = 1.0 / (b.max() - b.min())
    phi = -1.0 / (sd * sd * sigma)
    E, Einv = np.exp(phi), np.exp(-phi)
    Kb, Kbc = Kh @ hi.astype(float), Kh @ (hi * c)
    Sc = np.where(hi, Kbc, Kc - Kbc)
    Oc = np.where(hi, Kc - Kbc, Kbc)
    d_ref = sd * np.where(hi, 1.0, -1.0) * ((1.0 - E) * Sc + (1.0 - Einv) * Oc)
    assert np.max(np.abs(Dh[:, p - 1] - d_ref)) < 1e-12 * max(1.0, np.max(np.abs(d_ref)))


def test_error_reporting(lib):
    st = lib.bigkrls_gauss_kernel(None, 10, 2, 1.0, None)
    assert st == 1 and b"gauss_kernel" in lib.bigkrls_last_error()
    X = F(np.ones((4, 2)))
    out = F(np.zeros((4, 4)))
    st = lib.bigkrls_gauss_kernel(P(X), 4, 2, -1.0, P(out))
    assert st == 1 and b"sigma" in lib.bigkrls_last_error()


@pytest.mark.gpu
def test_eigen_is_run_to_run_deterministic(eig_path):
    """Two decompositions of the same matrix must agree bit for bit (no cross-block races, no atomics)."""
    import bigkrls_amd as bk
    from bigkrls_amd import ops
    ctx = bk.Context(0)
    rng = np.random.default_rng(5)
    n, p = 3000, 6
    X = rng.standard_normal((n, p))
    K = ops.bGaussKernel(ctx.from_numpy(X), float(p))
    a = ops.bEigen(K, n, 0.0)
    va, qa = a.values.copy(), a.vectors.to_numpy()
    ctx.torch.rand((1500, 1500), device=ctx.device)
    b = ops.bEigen(K, n, 0.0)
    assert np.array_equal(va, b.values)
    assert np.array_equal(qa, b.vectors.to_numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("own_stream", [False, True])
def test_eigen_stage1_as_a_captured_graph_is_bitwise_the_plain_loop(own_stream, monkeypatch):
    """BIGKRLS_S1_GRAPH=-1 (an experiment switch, off by default): from the third decomposition of one size on a context
    (n >= 8 192) the stage-1 panel loop is replayed as a captured hipGraph (csrc/eigen.hip, stage1_run): decompositions 1-2
    run the plain loop, 3 captures + instantiates + replays, 4-5 replay the cached executable graph -- all five bit for bit
    the same, on a context of the default stream (captured on a stream of the context's own) and on an own-stream
    context; a size in between (another n) drops the cached graph."""
    monkeypatch.setenv("BIGKRLS_S1_GRAPH", "-1")
    import bigkrls_amd as bk
    from bigkrls_amd import ops
    ctx = bk.Context(0, own_stream=own_stream)
    rng = np.random.default_rng(17)
    n, p = 8448, 6
    X = rng.standard_normal((n, p))
    K = ops.bGaussKernel(ctx.from_numpy(X), float(p))
    first = ops.bEigen(K, n, 0.001)
    v0, q0 = first.values.copy(), first.vectors.to_numpy()
    assert abs(v0.sum() - n) < 1e-9 * n
    for rep in range(4):
        e = ops.bEigen(K, n, 0.001)
        assert e.lastkeeper == first.lastkeeper, rep
        assert np.array_equal(v0, e.values), rep
        assert np.array_equal(q0, e.vectors.to_numpy()), rep
    # another size in between, then the first one again (a fresh count: plain, plain, capture)
    X2 = rng.standard_normal((8320, p))
    K2 = ops.bGaussKernel(ctx.from_numpy(X2), float(p))
    e2 = ops.bEigen(K2, 8320, 0.001)
    assert abs(e2.values.sum() - 8320) < 1e-9 * 8320
    for rep in range(3):
        e = ops.bEigen(K, n, 0.001)
        assert np.array_equal(v0, e.values), rep
    assert ctx.counters() == {"redone": 0, "replayed": 0, "replica_diff": 0}


@pytest.mark.gpu
@pytest.mark.parametrize("bc,pq", [("wavefront", "steps"), ("persistent", "steps"), ("resident", "resident"),
                                   ("lds", "resident")])
@pytest.mark.parametrize("n", [513, 1283])
def test_eigen_two_stage_kernel_variants(lib, monkeypatch, bc, pq, n):
    """The fallback kernels of the two-stage path (one launch per bulge-chasing wavefront / per panel
    column, and the flag-synchronised persistent bulge chasing) must stay correct: they serve the
    sizes the location-resident kernels cannot (n > 32768, panels of > 80 workgroups); "resident" is the
    register-window bulge chasing (bc_regwin, the default), "lds" the LDS-window kernel it replaced."""
    monkeypatch.delenv("BIGKRLS_EIG", raising=False)
    monkeypatch.setenv("BIGKRLS_BC", bc)
    if pq == "steps":
        monkeypatch.setenv("BIGKRLS_PQ", "steps")
    else:
        monkeypatch.delenv("BIGKRLS_PQ", raising=False)
    X, y = orc.synth(n, 5, 21)
    K = orc.gauss_kernel_literal(X, 5.0)
    vals = np.zeros(n)
    vecs = F(np.zeros((n, n)))
    Kf = F(K)
    check(lib, lib.bigkrls_eigen(P(Kf), n, n, P(vals), P(vecs)))
    ref_vals, _ = orc.big_eigen_literal(K, n)
    scale = np.abs(ref_vals).max()
    assert np.max(np.abs(vals - ref_vals)) / scale < 1e-12
    assert np.max(np.abs(vecs.T @ vecs - np.eye(n))) < 1e-11
    assert np.max(np.abs(K @ vecs - vecs * vals)) / scale < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["graded", "clusters", "identity", "rank3", "wilkinson", "negdef"])
@pytest.mark.parametrize("n", [300, 777])
def test_eigen_hard_spectra(lib, kind, n):
    """Spectra that break naive divide & conquer / secular solvers (tight clusters, 16 orders of
    grading, heavy deflation, Wilkinson's matrix, indefinite input): eigenvalues vs LAPACK,
    orthogonality and residuals at rounding level."""
    rng = np.random.default_rng(n)

    def with_spectrum(d):
        Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
        return (Qm * d) @ Qm.T

    if kind == "graded":
        A = with_spectrum(np.logspace(0, -16, n))
    elif kind == "clusters":
        A = with_spectrum(np.r_[1 + 1e-13 * rng.standard_normal(n // 2), 2 + 1e-13 * rng.standard_normal(n - n // 2)])
    elif kind == "identity":
        A = np.eye(n)
    elif kind == "rank3":
        A = with_spectrum(np.r_[[5.0, 3.0, 1.0], 1e-14 * rng.random(n - 3)])
    elif kind == "wilkinson":
        A = np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    else:
        A = -with_spectrum(np.logspace(0, -10, n))
    A = (A + A.T) / 2
    vals = np.zeros(n)
    vecs = F(np.zeros((n, n)))
    Af = F(A)
    check(lib, lib.bigkrls_eigen(P(Af), n, n, P(vals), P(vecs)))
    ref = np.linalg.eigvalsh(A)[::-1]
    scale = max(np.abs(ref).max(), 1e-300)
    assert np.max(np.abs(vals - ref)) / scale < 1e-12
    assert np.max(np.abs(vecs.T @ vecs - np.eye(n))) < 1e-11
    assert np.max(np.abs(A @ vecs - vecs * vals)) / scale < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("n,p,neig,trunc", [(4608, 6, 128, 0.0), (5000, 10, 300, 0.001)])
def test_eigen_block_lanczos_matches_dense_and_arpack(lib, monkeypatch, n, p, neig, trunc):
    """Neig << N (the reference's eigs_sym branch, src/eigen.cpp:18-22): the block-Lanczos path
    (forced here; by default it takes over at N >= 16384, Neig <= N/8) against the dense path of
    the same library and against ARPACK (the oracle's stand-in for eigs_sym)."""
    import bigkrls_amd as bk
    from bigkrls_amd import ops
    ctx = bk.Context(0)
    X, y = orc.synth(n, p, 31)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
    monkeypatch.setenv("BIGKRLS_EIGK", "dense")
    d = ops.bEigen(K, neig, trunc)
    monkeypatch.setenv("BIGKRLS_EIGK", "krylov")
    a = ops.bEigen(K, neig, trunc)
    assert a.lastkeeper == d.lastkeeper
    assert np.max(np.abs(a.values - d.values)) <= 1e-11 * d.values[0]
    Qa, Qd = a.vectors.to_numpy(), d.vectors.to_numpy()
    assert np.max(np.abs(Qa.T @ Qa - np.eye(Qa.shape[1]))) < 1e-11
    ys = (y - y.mean()) / y.std(ddof=1)
    w = 1.0 / (d.values[:d.lastkeeper] + 0.5)
    cd, ca = Qd @ (w * (Qd.T @ ys)), Qa @ (w * (Qa.T @ ys))       # rotation/sign invariant
    assert np.max(np.abs(cd - ca)) <= 1e-8 * np.max(np.abs(cd))
    ref = orc.b_eigen(K.to_numpy(), neig, trunc)                   # ARPACK
    assert ref.lastkeeper == a.lastkeeper
    assert np.max(np.abs(a.values - ref.values)) <= 1e-9 * ref.values[0]


@pytest.mark.gpu
@pytest.mark.parametrize("p,neig", [(2, 512), (2, 1536), (3, 384), (1, 256)])
def test_eigen_block_lanczos_on_numerically_low_rank_kernels(lib, monkeypatch, p, neig):
    """Kernels of one to three columns fall below rounding long before Neig eigenvalues (numerical rank ~20 / ~150 /
    ~600 at N = 17 000): the block Krylov space becomes invariant to working precision after a few steps. The library's
    default choice (N >= 16384, Neig <= N/8: block Lanczos) must then stop and decide on true residuals instead of
    normalising rounding noise into basis vectors (P = 2, 3), re-orthogonalise the ill-conditioned blocks on the way,
    fill up with random blocks when the invariant subspace has fewer than Neig columns (P = 2, Neig = 1536),
    and hand a breakdown before the subspace has Neig columns to the dense path (P = 1) -- the same pairs as the dense
    path in every case (the reference's eigs_sym branch has no such restriction, src/eigen.cpp:18-22)."""
    import bigkrls_amd as bk
    from bigkrls_amd import ops
    ctx = bk.Context(0)
    n = 17000
    X, y = orc.synth(n, p, 41)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
    monkeypatch.setenv("BIGKRLS_EIGK", "dense")
    d = ops.bEigen(K, neig, 0.001)
    monkeypatch.delenv("BIGKRLS_EIGK")
    a = ops.bEigen(K, neig, 0.001)
    assert a.lastkeeper == d.lastkeeper
    assert np.max(np.abs(np.asarray(a.values) - np.asarray(d.values))) <= 1e-10 * d.values[0]
    Qa = a.vectors.to_numpy()
    k = a.lastkeeper
    assert np.max(np.abs(Qa.T @ Qa - np.eye(k))) < 1e-11
    KQ = ops.gemm(False, False, K, a.vectors).to_numpy()
    assert np.max(np.linalg.norm(KQ - Qa * np.asarray(a.values)[:k], axis=0)) <= 1e-9 * a.values[0]
    # ... and as a fit: the same coefficients as with the dense decomposition
    fa = bk.bigKRLS(y, X, Neig=neig, ctx=ctx, noisy=False, derivative=False)
    monkeypatch.setenv("BIGKRLS_EIGK", "dense")
    fd = bk.bigKRLS(y, X, Neig=neig, ctx=ctx, noisy=False, derivative=False)
    assert fa["lastkeeper"] == fd["lastkeeper"]
    assert abs(fa["lambda"] - fd["lambda"]) <= 1e-8 * abs(fd["lambda"])
    assert np.max(np.abs(fa["coeffs"] - fd["coeffs"])) <= 1e-7 * np.max(np.abs(fd["coeffs"]))
    assert ctx.counters()["redone"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("nbytes", [1, 8, 65535, 65536 * 2 + 8, (16 << 20) * 2, (16 << 20) * 5 + 24])
def test_c_abi_host_copies_round_trip(lib, ctx, nbytes):
    """bigkrls_h2d / bigkrls_d2h with caller (pageable) memory: from one byte to 80 MB + 24, byte for byte."""
    import ctypes as C
    rng = np.random.default_rng(nbytes % 1000)
    src = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    back = np.zeros(nbytes, dtype=np.uint8)
    dptr = C.c_void_p()
    check(lib, lib.bigkrls_dev_alloc(ctx.handle, nbytes, C.byref(dptr)))
    try:
        check(lib, lib.bigkrls_h2d(ctx.handle, dptr, src.ctypes.data_as(C.c_void_p), nbytes))
        check(lib, lib.bigkrls_d2h(ctx.handle, back.ctypes.data_as(C.c_void_p), dptr, nbytes))
    finally:
        check(lib, lib.bigkrls_dev_free(ctx.handle, dptr))
    assert np.array_equal(back, src)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,order", [((1000, 7), "C"), ((1000, 7), "F"), ((1, 1), "C"), ((5, 0), "C"),
                                         ((3000, 3001), "C")])
def test_host_device_transfers_round_trip(ctx, monkeypatch, shape, order):
    """from_numpy / to_numpy go through the context's pinned staging buffer; with a small
    buffer the transfer is cut into pieces that straddle columns (the pipelined path that
    N x N matrices above 128 MB take)."""
    from bigkrls_amd import device
    if shape[0] * shape[1] > 1 << 20:
        monkeypatch.setattr(device, "_STAGE_MAX", 1 << 20)
        monkeypatch.setattr(ctx, "_stage", None)
        monkeypatch.setattr(ctx, "_stage_np", None)
    rng = np.random.default_rng(5)
    a = np.asarray(rng.standard_normal(shape), order=order)
    d = ctx.from_numpy(a)
    assert d.shape == shape
    back = d.to_numpy()
    assert back.shape == shape and np.array_equal(back, a)
    if a.size:
        sl = ctx.from_numpy(a[::2, 1:])              # neither C- nor F-contiguous
        assert np.array_equal(sl.to_numpy(), a[::2, 1:])
        assert np.array_equal(d.cols(1, 3).to_numpy(), a[:, 1:3])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,neig", [("clusters", 1100, 100), ("wilkinson", 1501, 150), ("identity", 640, 64),
                                         ("rank3", 900, 20), ("graded", 1024, 128), ("kernel", 4300, 200)])
def test_eigen_divide_conquer_factored_top_levels(ctx, monkeypatch, kind, n, neig):
    """Few eigenvectors wanted: the top levels of the divide & conquer stay factored (boundary rows
    propagated, secular-vector blocks stashed, operators applied to the kept columns at the end).
    Same eigenvalues as with every level formed, orthonormal vectors, rounding-level residuals --
    including spectra whose merges deflate with rotations (clusters, Wilkinson) or entirely
    (identity)."""
    from bigkrls_amd import ops
    rng = np.random.default_rng(n)

    def with_spectrum(d):
        Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
        return (Qm * d) @ Qm.T

    if kind == "clusters":
        A = with_spectrum(np.r_[1 + 1e-13 * rng.standard_normal(n // 2), 2 + 1e-13 * rng.standard_normal(n - n // 2)])
    elif kind == "wilkinson":
        A = np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    elif kind == "identity":
        A = np.eye(n)
    elif kind == "rank3":
        A = with_spectrum(np.r_[[5.0, 3.0, 1.0], 1e-14 * rng.random(n - 3)])
    elif kind == "graded":
        A = with_spectrum(np.logspace(0, -16, n))
    else:
        X, _ = orc.synth(n, 6, 41)
        A = orc.gauss_kernel_literal((X - X.mean(0)) / X.std(0, ddof=1), 6.0)
    A = (A + A.T) / 2
    Ad = ctx.from_numpy(A)
    monkeypatch.setenv("BIGKRLS_EIGK", "dense")
    monkeypatch.setenv("BIGKRLS_DC", "explicit")
    e = ops.bEigen(Ad, neig, -1.0)
    monkeypatch.setenv("BIGKRLS_DC", "factored")
    f = ops.bEigen(Ad, neig, -1.0)
    ref = np.linalg.eigvalsh(A)[::-1][:neig]
    scale = max(np.abs(ref).max(), 1e-300)
    assert f.lastkeeper == e.lastkeeper == neig
    assert np.max(np.abs(f.values - e.values)) / scale < 1e-13
    assert np.max(np.abs(f.values[:neig] - ref)) / scale < 1e-12
    Q = f.vectors.to_numpy()
    assert np.max(np.abs(Q.T @ Q - np.eye(neig))) < 1e-11
    assert np.max(np.abs(A @ Q - Q * f.values[:neig])) / scale < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("n", [700, 1283])
def test_eigen_back_transform_variants(lib, monkeypatch, n):
    """Back-transforms: the compact-WY MFMA tasks of stage 2 in one persistent launch and the merged block
    reflectors of stage 1 (defaults) against the reflector-by-reflector kernel (BIGKRLS_BT2=seq) and the
    panel-by-panel loop (BIGKRLS_BT1=panel): same eigenvectors up to rounding; against the same tasks as one launch
    per anti-diagonal (BIGKRLS_BT2=wavefront) and as chains of tasks (BIGKRLS_BT2=chain): bitwise."""
    monkeypatch.delenv("BIGKRLS_EIG", raising=False)
    X, y = orc.synth(n, 4, 23)
    K = orc.gauss_kernel_literal(X, 4.0)
    Kf = F(K)
    out = {}
    for mode in ("wy", "seq", "wavefront", "chain", "chain3"):
        monkeypatch.delenv("BIGKRLS_BT2_SEG", raising=False)
        if mode in ("chain", "chain3"):                    # chains along the sweep groups (the default above n = 8000)
            monkeypatch.setenv("BIGKRLS_BT2", "chain")
            if mode == "chain3":                           # three groups per ticket: every rotation of the LDS window,
                monkeypatch.setenv("BIGKRLS_BT2_SEG", "3") # segment ends inside and at the end of the chains
            monkeypatch.delenv("BIGKRLS_BT1", raising=False)
            monkeypatch.delenv("BIGKRLS_S1", raising=False)
        elif mode == "seq":
            monkeypatch.setenv("BIGKRLS_BT2", "seq")
            monkeypatch.setenv("BIGKRLS_BT1", "panel")
            monkeypatch.setenv("BIGKRLS_S1", "gemm")      # stage 1's small products as separate GEMMs
        elif mode == "wavefront":                          # the same compact-WY tasks, one launch per anti-diagonal
            monkeypatch.setenv("BIGKRLS_BT2", "wavefront")
            monkeypatch.delenv("BIGKRLS_BT1", raising=False)
            monkeypatch.delenv("BIGKRLS_S1", raising=False)
        else:
            monkeypatch.delenv("BIGKRLS_BT2", raising=False)
            monkeypatch.delenv("BIGKRLS_BT1", raising=False)
            monkeypatch.delenv("BIGKRLS_S1", raising=False)
        vals = np.zeros(n)
        vecs = F(np.zeros((n, n)))
        check(lib, lib.bigkrls_eigen(P(Kf), n, n, P(vals), P(vecs)))
        assert np.max(np.abs(vecs.T @ vecs - np.eye(n))) < 1e-11
        assert np.max(np.abs(K @ vecs - vecs * vals)) / vals[0] < 1e-11
        out[mode] = (vals, vecs)
    assert np.max(np.abs(out["wy"][0] - out["seq"][0])) / out["wy"][0][0] < 1e-13
    # the persistent launch (ticket order) and the per-anti-diagonal launches run the same tasks on the same data
    assert np.array_equal(out["wy"][0], out["wavefront"][0]) and np.array_equal(out["wy"][1], out["wavefront"][1])
    # ... and so do the chains, which keep the rows two consecutive tasks share in LDS
    for mode in ("chain", "chain3"):
        assert np.array_equal(out["wy"][1], out[mode][1]), mode
    # well separated top of the spectrum: the vectors themselves agree
    assert np.max(np.abs(np.abs(out["wy"][1][:, :5]) - np.abs(out["seq"][1][:, :5]))) < 1e-9


@pytest.mark.gpu
def test_eigen_large_panels_two_level_exchange(ctx):
    """n = 7700: the first stage-1 panels span 30 workgroups, above the threshold where the
    register-resident panel QR switches from the all-to-all of partial sums to the two-level exchange
    (group leaders, then group sums). Top eigenpairs: residual, orthogonality, trace."""
    from bigkrls_amd import ops
    n, p, k = 7700, 5, 48
    X, _ = orc.synth(n, p, 61)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    K = ops.bGaussKernel(ctx.from_numpy(Xs), float(p))
    eo = ops.bEigen(K, None, 0.0)                       # all eigenvalues ...
    assert abs(eo.values.sum() - n) / n < 1e-12         # ... sum to trace(K) = n
    top = ops.bEigen(K, k, -1.0)
    Q = top.vectors
    lam = top.values[:k]
    assert np.max(np.abs(lam - eo.values[:k])) / lam[0] < 1e-13
    R = ops.gemm(False, False, K, Q).to_numpy() - Q.to_numpy() * lam
    G = ops.gemm(True, False, Q, Q).to_numpy()
    assert np.max(np.abs(R)) / lam[0] < 1e-12
    assert np.max(np.abs(G - np.eye(k))) < 1e-11


# --------------------------------------------------------------------------------------------
# the device-side pieces of the block Lanczos that the row-block path drives through the C ABI
# --------------------------------------------------------------------------------------------
def test_dev_fill_random_is_a_function_of_index_and_seed(ctx):
    from bigkrls_amd import _lib
    a, b = ctx.empty(5000, 128), ctx.empty(5000, 128)
    for m in (a, b):
        _lib.call("bigkrls_dev_fill_random", ctx.handle, m.ptr, 5000 * 128, 20240229)
    c = ctx.empty(5000, 128)
    _lib.call("bigkrls_dev_fill_random", ctx.handle, c.ptr, 5000 * 128, 7)
    ha, hb, hc = a.to_numpy(), b.to_numpy(), c.to_numpy()
    assert np.array_equal(ha, hb) and not np.array_equal(ha, hc)
    assert ha.min() >= -0.5 and ha.max() < 0.5 and abs(ha.mean()) < 2e-3
    assert np.linalg.matrix_rank(ha[:300, :128]) == 128


@pytest.mark.parametrize("n,b", [(5000, 128), (1237, 128), (4096, 64), (300, 17)])
def test_dev_cholqr2_orthonormalises_and_reports_breakdown(ctx, n, b):
    """W_in = W_out R with W_out'W_out = I to rounding and R upper triangular (Cholesky-QR twice, Gram product on
    the MFMA GEMM, factorisation and inverse in the register-tile kernel); a pivot that is not positive sets the flag
    (columns that are dependent only to rounding can slip through with a tiny positive pivot, as in any Cholesky)."""
    from bigkrls_amd import _lib
    rng = np.random.default_rng(n + b)
    W0 = rng.standard_normal((n, b)) @ np.diag(np.logspace(0, -3, b)) + 0.05 * rng.standard_normal((n, 1))
    W, tmp, Rd = ctx.from_numpy(F(W0)), ctx.empty(n, b), ctx.empty(b, b)
    R = np.zeros((b, b), order="F")
    brk = C.c_int32(-1)
    _lib.call("bigkrls_dev_cholqr2", ctx.handle, W.ptr, tmp.ptr, n, b, R.ctypes.data_as(C.c_void_p), C.byref(brk), Rd.ptr)
    Q = W.to_numpy()
    assert brk.value == 0
    assert np.abs(Q.T @ Q - np.eye(b)).max() < 1e-13
    assert np.abs(np.tril(R, -1)).max() == 0.0 and np.all(np.diag(R) > 0)
    assert np.abs(Q @ R - W0).max() < 1e-12 * np.abs(W0).max()
    assert np.array_equal(Rd.to_numpy(), R)
    W1 = W0.copy()
    W1[:, b // 2] = 0.0                            # a zero column: the pivot is not positive
    W = ctx.from_numpy(F(W1))
    _lib.call("bigkrls_dev_cholqr2", ctx.handle, W.ptr, tmp.ptr, n, b, R.ctypes.data_as(C.c_void_p), C.byref(brk), None)
    assert brk.value == 1


def test_dev_lanczos_projected_assembles_the_block_tridiagonal_matrix(ctx):
    from bigkrls_amd import _lib
    rng = np.random.default_rng(3)
    steps, b = 5, 32
    A = rng.standard_normal((steps, b, b))
    Bt = np.triu(rng.standard_normal((steps, b, b)))
    # device layout: blocks one after the other, each column-major
    Ad = ctx.from_numpy(F(np.concatenate([A[j].T.reshape(-1) for j in range(steps)])[:, None]))
    Bd = ctx.from_numpy(F(np.concatenate([Bt[j].T.reshape(-1) for j in range(steps)])[:, None]))
    m = steps * b
    Td = ctx.empty(m, m)
    _lib.call("bigkrls_dev_lanczos_projected", ctx.handle, Ad.ptr, Bd.ptr, steps, b, Td.ptr)
    ref = np.zeros((m, m))
    for j in range(steps):
        ref[j * b:(j + 1) * b, j * b:(j + 1) * b] = 0.5 * (A[j] + A[j].T)
        if j + 1 < steps:
            ref[(j + 1) * b:(j + 2) * b, j * b:(j + 1) * b] = Bt[j]
            ref[j * b:(j + 1) * b, (j + 1) * b:(j + 2) * b] = Bt[j].T
    assert np.array_equal(Td.to_numpy(), ref)
