"""bigKRLS()/predict()/crossvalidate() on the GPU against the CPU oracle.

Tolerance: north_star asks for 1e-6 relative fp64 on coefficients, fitted values,
lambda and pointwise derivatives; written as TOL below."""
import os

import numpy as np
import pytest

from oracle import krls_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-6


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def assert_fit_parity(out, ref, tol=TOL, squares=True):
    assert out["lastkeeper"] == ref["lastkeeper"]
    assert rel(out["K.eigenvalues"], ref["K.eigenvalues"]) < 1e-11
    assert abs(out["lambda"] - ref["lambda"]) <= tol * abs(ref["lambda"])
    for k in ["coeffs", "yfitted", "derivatives", "avgderivatives", "var.avgderivatives",
              "derivatives.std", "var.avgderivatives.std"]:
        if k in ref:
            assert rel(out[k], ref[k]) < tol, k
    for k in ["R2", "R2AME", "Looe", "Neffective", "sigmasq"]:
        if k in ref and ref[k] is not None:
            assert abs(out[k] - ref[k]) <= tol * max(abs(ref[k]), 1e-12), k
    if squares:
        for k in ["K", "vcov.est.c", "vcov.est.fitted"]:
            o = out[k].to_numpy() if hasattr(out[k], "to_numpy") else out[k]
            assert rel(o, ref[k]) < tol, k


def test_fit_c1_config(ctx):
    """BASELINE.json configs[0]: N=500 P=5 synthetic sin(X beta), full eigen, literal oracle."""
    import bigkrls_amd as bk
    X, y = orc.synth(500, 5, 101)
    tr_ref = orc.LambdaTrace(0, 0)
    ref = orc.fit(y, X, literal=True, trace=tr_ref)
    tr = []
    out = bk.bigKRLS(y, X, ctx=ctx, trace=tr)
    assert_fit_parity(out, ref)
    # quirk Q8: identical golden-section branch sequence
    assert len(tr) == len(tr_ref.probes)
    for (l1, s1), (l2, s2) in zip(tr, tr_ref.probes):
        assert abs(l1 - l2) <= 1e-12 * abs(l2) and abs(s1 - s2) <= 1e-8 * abs(s2)


def test_fit_binary_column_truncated(ctx):
    """Mirrors examples/numeric_convergence.md: N=500, P=6, one binary column, eigtrunc 0.01."""
    import bigkrls_amd as bk
    X, y = orc.synth(500, 6, 2018, binary_last=True)
    ref = orc.fit(y, X, eigtrunc=0.01, literal=True)
    out = bk.bigKRLS(y, X, eigtrunc=0.01, ctx=ctx)
    assert list(out["binaryindicator"]) == [False] * 5 + [True]
    assert_fit_parity(out, ref)


def test_fit_neig_partial(ctx):
    import bigkrls_amd as bk
    X, y = orc.synth(500, 5, 33)
    ref = orc.fit(y, X, neig=50, literal=False)
    out = bk.bigKRLS(y, X, Neig=50, ctx=ctx)
    assert out["K.eigenvalues"].shape == (50,)
    assert_fit_parity(out, ref)


def test_fit_which_derivatives_q6(ctx):
    """Quirk Q6: rescaling uses X.init.sd[1..P'] not the selected columns."""
    import bigkrls_amd as bk
    X, y = orc.synth(300, 5, 44)
    X[:, 1] *= 7.0
    ref = orc.fit(y, X, which_derivatives=[2, 4], literal=True)
    out = bk.bigKRLS(y, X, which_derivatives=[2, 4], ctx=ctx)
    assert out["derivatives"].shape == (300, 2)
    assert_fit_parity(out, ref)
    # a which.derivatives list longer than ncol(X) (repeats are legal in R): X.init.sd[i] is NA for i > p
    # (R/bigKRLS.R:395-397), the standardised derivatives and the correctly subset variances stay finite
    X2, y2 = orc.synth(200, 2, 45)
    rep = bk.bigKRLS(y2, X2, which_derivatives=[1, 1, 2], ctx=ctx)
    one = bk.bigKRLS(y2, X2, ctx=ctx)
    assert rep["derivatives"].shape == (200, 3) and np.isnan(rep["derivatives"][:, 2]).all()
    assert np.isnan(rep["avgderivatives"][0, 2]) and np.isfinite(rep["derivatives"][:, :2]).all()
    assert np.allclose(rep["derivatives.std"][:, [0, 2]], one["derivatives.std"], rtol=1e-9, atol=1e-12)
    assert np.allclose(rep["var.avgderivatives"][0, [1, 2]], one["var.avgderivatives"][0], rtol=1e-8)


def test_fit_user_lambda_and_no_derivative(ctx):
    import bigkrls_amd as bk
    X, y = orc.synth(257, 3, 45)
    ref = orc.fit(y, X, lam=0.3, derivative=False, literal=True)
    out = bk.bigKRLS(y, X, lambda_=0.3, derivative=False, ctx=ctx)
    assert out["lambda"] == 0.3
    assert "derivatives" not in out
    assert_fit_parity(out, ref)


def test_fit_mtcars_reference_test(ctx):
    """The reference's own test (tests/testthat/test_basic_usage.R:48-109): fit with
    eigtrunc=0, predict with hp:=200, proportion 0.6875, Corolla kernel column."""
    import csv, os
    import bigkrls_amd as bk
    here = os.path.dirname(__file__)
    rows = list(csv.reader(open(os.path.join(here, "golden", "mtcars.csv"))))
    names = [r[0] for r in rows[1:]]
    M = np.array([[float(v) for v in r[1:]] for r in rows[1:]])
    y, X = M[:, 0], M[:, 1:]
    out = bk.bigKRLS(y, X, eigtrunc=0, ctx=ctx)
    Xnew = X.copy()
    Xnew[:, 2] = 200.0
    fc = bk.predict(out, Xnew)
    assert np.mean(fc["predicted"] < y) == 0.6875
    gold = {r[0]: float(r[1]) for r in list(csv.reader(open(os.path.join(here, "golden", "mtcars_corolla_kernel.csv"))))[1:]}
    s = np.asarray(out["K"])[:, names.index("Toyota Corolla")]
    assert max(s[i] - gold[nm] for i, nm in enumerate(names)) < 0.01
    ref = orc.fit(y, X, eigtrunc=0, literal=True)
    assert_fit_parity(out, ref)


def test_predict_with_se(ctx):
    import bigkrls_amd as bk
    X, y = orc.synth(400, 4, 46)
    ref = orc.fit(y[:350], X[:350], literal=False)
    out = bk.bigKRLS(y[:350], X[:350], ctx=ctx)
    pr = orc.predict(ref, X[350:], se_pred=True)
    po = bk.predict(out, X[350:], se_pred=True)
    assert rel(po["predicted"], pr["predicted"]) < TOL
    assert rel(po["se.pred"], pr["se.pred"]) < TOL
    assert rel(po["newdataK"], pr["newdataK"]) < 1e-12
    assert rel(po["vcov.est.pred"], pr["vcov.est.pred"]) < TOL


def test_crossvalidate_explicit_indices(ctx):
    import bigkrls_amd as bk
    X, y = orc.synth(300, 3, 47)
    rng = np.random.default_rng(0)
    tr = np.sort(rng.choice(300, 240, replace=False))
    te = np.setdiff1d(np.arange(300), tr)
    ref = orc.crossvalidate_split(y, X, tr, te, literal=False)
    out = bk.crossvalidate(y, X, seed=1, ptesting=20, train_idx=tr, ctx=ctx)
    for k in ["pseudoR2_is", "pseudoR2_oos", "MSE_is", "MSE_oos", "pseudoR2AME_is",
              "pseudoR2AME_oos", "MSE_AME_is", "MSE_AME_oos"]:
        assert abs(out[k] - ref[k]) <= TOL * abs(ref[k]), k
    folds = (np.arange(300) % 3) + 1
    refk = orc.crossvalidate_kfolds(y, X, folds, literal=False)
    outk = bk.crossvalidate(y, X, seed=1, Kfolds=3, folds=folds, ctx=ctx)
    for k in ["R2_is", "R2_oos", "MSE_is", "MSE_oos", "R2AME_is", "R2AME_oos", "MSE_AME_is", "MSE_AME_oos"]:
        assert rel(outk[k], refk[k]) < TOL, k


def test_validation_errors(ctx):
    import bigkrls_amd as bk
    X, y = orc.synth(50, 3, 1)
    Xc = X.copy(); Xc[:, 1] = 2.0
    with pytest.raises(ValueError, match="constant"):
        bk.bigKRLS(y, Xc, ctx=ctx)
    Xn = X.copy(); Xn[3, 0] = np.nan
    with pytest.raises(ValueError, match="missing data"):
        bk.bigKRLS(y, Xn, ctx=ctx)
    with pytest.raises(ValueError, match="nrow"):
        bk.bigKRLS(y[:-1], X, ctx=ctx)
    with pytest.raises(ValueError, match="y is a constant"):
        bk.bigKRLS(np.ones(50), X, ctx=ctx)
    with pytest.raises(ValueError, match="eigtrunc"):
        bk.bigKRLS(y, X, eigtrunc=2, ctx=ctx)
    with pytest.raises(ValueError, match="vcov.est is needed"):
        bk.bigKRLS(y, X, vcov_est=False, ctx=ctx)
    with pytest.raises(ValueError, match="which.derivative"):
        bk.bigKRLS(y, X, derivative=False, which_derivatives=[1], ctx=ctx)


@pytest.mark.parametrize("path", ["1stage", "2stage"])
def test_fit_medium_n2000(ctx, monkeypatch, path):
    """Bigger than one tile grid / several reflector panels; default eigtrunc path off (n<=3000).
    Both tridiagonalisation paths must give the same fit."""
    import bigkrls_amd as bk
    if path == "1stage":
        monkeypatch.setenv("BIGKRLS_EIG", "1stage")
    else:
        monkeypatch.delenv("BIGKRLS_EIG", raising=False)
    X, y = orc.synth(2000, 10, 102)
    ref = orc.fit(y, X, literal=False)
    T = {}
    out = bk.bigKRLS(y, X, ctx=ctx, timings=T)
    print("timings N=2000:", {k: round(v, 4) for k, v in T.items()})
    assert_fit_parity(out, ref)


def test_dist_path_world1_rccl_communicator(ctx):
    """bigkrls_fit_dist on one GPU through a one-rank RCCL communicator (ncclCommInitRank inside the library, no
    process group): the row-block code path with every collective issued -- against the oracle, and the sharded
    N x N outputs (this rank's column blocks = everything) against the oracle's matrices."""
    import bigkrls_amd as bk
    from bigkrls_amd import dist as bkdist
    X, y = orc.synth(700, 5, 48, binary_last=True)
    ref = orc.fit(y, X, literal=False)
    comm = bkdist.get_comm(ctx, "rccl")
    assert (comm.world, comm.rank, comm.kind) == (1, 0, "rccl")
    tr = []
    out = bkdist.bigKRLS_dist(y, X, comm=comm, trace=tr)
    assert out["rows"] == (0, 700)
    assert_fit_parity(out, ref, squares=False)
    assert len(tr) > 5 and abs(tr[-1][0] - out["lambda"]) < 0.5 * out["lambda"]
    assert rel(out["K.cols"].to_numpy(), ref["K"]) < 1e-12
    assert rel(out["vcov.est.c.cols"].to_numpy(), ref["vcov.est.c"]) < TOL
    assert rel(out["vcov.est.fitted.cols"].to_numpy(), ref["vcov.est.fitted"]) < TOL
    # the same through the callback table (host-staged collectives, one rank): identical small outputs
    cb = bkdist.bigKRLS_dist(y, X, ctx=ctx, collectives="host")
    for k in ("coeffs", "derivatives", "var.avgderivatives", "K.eigenvalues"):
        assert np.array_equal(np.asarray(cb[k]), np.asarray(out[k])), k
    assert cb["lambda"] == out["lambda"]
    lean = bkdist.bigKRLS_dist(y, X, comm=comm, keep_outputs=False)      # nothing N x N handed back
    assert "K.cols" not in lean and np.array_equal(lean["coeffs"], out["coeffs"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,p", [(1, 3), (2, 3), (33, 4), (500, 5), (1000, 20), (2049, 7), (777, 130)])
def test_neffective_matches_oracle(n, p):
    """bigkrls_neffective (Level 1, host pointers) and ops.bNeffective (device) vs the literal oracle."""
    import ctypes as C
    import bigkrls_amd as bk
    from bigkrls_amd import _lib, ops
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p)) + 0.3 * rng.standard_normal((n, 1))
    ref = orc.neffective_literal(X)
    Xf = np.asfortranarray(X)
    out = np.zeros(1)
    _lib.call("bigkrls_neffective", Xf.ctypes.data_as(C.c_void_p), n, p, out.ctypes.data_as(C.c_void_p))
    assert abs(out[0] - ref) <= 1e-10 * abs(ref), (out[0], ref)
    ctx = bk.Context(0)
    got = ops.bNeffective(ctx.from_numpy(X))
    assert got == out[0]                      # same kernel, deterministic reduction


@pytest.mark.gpu
def test_acf_and_summary_match_oracle():
    """bigKRLS(acf=TRUE) stores Neffective.acf of the standardised X (R/bigKRLS.R:412-416); summary()
    reproduces the t-tests and percentiles of summary.bigKRLS for all three `degrees` choices."""
    import bigkrls_amd as bk
    X, y = orc.synth(400, 4, 77, binary_last=True)
    out = bk.bigKRLS(y, X, acf=True)
    Xs = (X - X.mean(axis=0)) / X.std(axis=0, ddof=1)
    assert abs(out["Neffective.acf"] - orc.neffective_literal(Xs)) < 1e-9 * 400
    ref_fit = orc.fit(y, X, literal=False)
    for deg in ("Neffective", "N", "acf"):
        s = bk.summary(out, degrees=deg, quiet=True)
        r = orc.summary_tables(ref_fit, degrees=deg)
        assert s["rownames"] == ["x1", "x2", "x3", "x4*"]
        assert np.allclose(s["ttests"][:, :3], r["ttests"][:, :3], rtol=1e-6, atol=0)
        assert np.allclose(s["ttests"][:, 3], r["ttests"][:, 3], rtol=1e-6, atol=1e-300)
        assert np.allclose(s["percentiles"], r["percentiles"], rtol=1e-6, atol=1e-12)
    assert bk.bigKRLS(y, X[:, :2], acf=True)["Neffective.acf"] is None     # acf <- acf & p > 2 (:192)


@pytest.mark.gpu
@pytest.mark.parametrize("binary", [False, True])
def test_save_load_round_trip(tmp_path, binary):
    """save.bigKRLS / load.bigKRLS round trip (the reference pins exactly this:
    tests/testthat/test_basic_usage.R:53-61, :123-128): device-resident matrices come back equal,
    and the reloaded object predicts identically."""
    import bigkrls_amd as bk
    X, y = orc.synth(2600, 3, 5)               # n > 2500: K and the variance matrices stay on the device
    out = bk.bigKRLS(y, X, eigtrunc=0.01)
    folder = bk.save_bigKRLS(out, str(tmp_path / "model"), noisy=False, binary=binary)
    assert os.path.exists(os.path.join(folder, "K.npy" if binary else "K.txt"))
    back = bk.load_bigKRLS(folder, noisy=False)
    for k in ("coeffs", "yfitted", "derivatives", "avgderivatives", "var.avgderivatives", "X", "y"):
        assert np.array_equal(np.asarray(back[k]), np.asarray(out[k])), k
    for k in ("lambda", "R2", "Neffective", "sigma", "Looe", "lastkeeper"):
        assert back[k] == out[k], k
    for k in ("K", "vcov.est.c", "vcov.est.fitted"):
        assert np.array_equal(back[k].to_numpy(), out[k].to_numpy()), k
    p0 = bk.predict(out, X[:20] + 0.05, se_pred=True)
    p1 = bk.predict(back, X[:20] + 0.05, se_pred=True)
    assert np.array_equal(p0["predicted"], p1["predicted"]) and np.array_equal(p0["se.pred"], p1["se.pred"])
    again = bk.save_bigKRLS(out, str(tmp_path / "model"), noisy=False, binary=binary)   # existing folder is not reused
    assert again != folder and os.path.isdir(again)
    assert os.path.exists(os.path.join(folder, "estimates.RData"))
    if not binary:
        # bigKRLS(..., model_subfolder_name=) saves as it goes (R/bigKRLS.R:111-133, :471-504); an existing folder
        # is kept unless overwrite.existing
        out2 = bk.bigKRLS(y, X, eigtrunc=0.01, model_subfolder_name=str(tmp_path / "model"), noisy=False)
        assert out2["path"] == str(tmp_path / "model2") and os.path.exists(os.path.join(out2["path"], "K.txt"))
        back2 = bk.load_bigKRLS(out2["path"], noisy=False)
        assert np.array_equal(back2["coeffs"], out["coeffs"]) and back2["lambda"] == out["lambda"]
        assert np.array_equal(back2["vcov.est.fitted"].to_numpy(), out["vcov.est.fitted"].to_numpy())
        out3 = bk.bigKRLS(y, X, eigtrunc=0.01, model_subfolder_name=str(tmp_path / "model"), overwrite_existing=True,
                          noisy=False)
        assert out3["path"] == str(tmp_path / "model")


@pytest.mark.gpu
def test_eigen_column_partition_sums_to_full():
    """bigkrls_dev_eigen_part: the slices back-transformed by the ranks of a multi-GPU run are
    disjoint column blocks whose sum is the eigenvector matrix of the single-GPU call."""
    import bigkrls_amd as bk
    from bigkrls_amd import ops
    ctx = bk.Context(0)
    X, y = orc.synth(700, 4, 11)
    Xs = (X - X.mean(0)) / X.std(0, ddof=1)
    K = ops.bGaussKernel(ctx.from_numpy(Xs), 4.0)
    full = ops.bEigen(K, 700, 0.001)
    Qf = full.vectors.to_numpy()
    acc = np.zeros_like(Qf)
    for r in range(3):
        part = ops.bEigen(K, 700, 0.001, part=(r, 3))
        assert part.lastkeeper == full.lastkeeper and np.array_equal(part.values, full.values)
        Qp = part.vectors.to_numpy()
        nv = full.lastkeeper
        c0, c1 = nv * r // 3, nv * (r + 1) // 3
        assert not Qp[:, :c0].any() and not Qp[:, c1:].any()
        acc += Qp
    assert np.array_equal(acc, Qf)


@pytest.mark.gpu
def test_dist_path_sharded_block_lanczos_world1(ctx):
    """bigkrls_fit_dist with the block Lanczos whose K B_j products are sharded by row block (SURVEY 8(e) "Eigen,
    partial"), one rank: the same fit as bigkrls_fit for Neig << N."""
    import bigkrls_amd as bk
    from bigkrls_amd import dist as bkdist
    X, y = orc.synth(4500, 6, 52)
    one = bk.bigKRLS(y, X, Neig=96, ctx=ctx)
    out = bkdist.bigKRLS_dist(y, X, Neig=96, ctx=ctx, eigen_mode="krylov")
    assert out["lastkeeper"] == one["lastkeeper"]
    assert abs(out["lambda"] - one["lambda"]) <= 1e-8 * one["lambda"]
    assert rel(out["K.eigenvalues"], one["K.eigenvalues"]) < 1e-9
    for k in ("coeffs", "yfitted", "derivatives", "var.avgderivatives"):
        assert rel(out[k], one[k]) < TOL, k


@pytest.mark.gpu
def test_crossvalidate_fold_parallel_contexts(ctx):
    """Fold-parallel crossvalidate (SURVEY 8(e) "replicas only": whole folds, one per GPU at a time;
    the reference's loop is R/bigKRLS.R:1268-1282). On the one GPU of this box: (i) devices=[0] goes
    through the worker path and equals the sequential loop and the oracle; (ii) two contexts with
    their own streams on device 0, one worker thread each -- the multi-GPU code path with both
    "GPUs" being the same device -- give the same statistics, fold for fold."""
    import bigkrls_amd as bk
    X, y = orc.synth(600, 4, 49, binary_last=True)
    folds = (np.arange(600) % 4) + 1
    ref = orc.crossvalidate_kfolds(y, X, folds, literal=False)
    seq = bk.crossvalidate(y, X, Kfolds=4, folds=folds, ctx=ctx)
    one = bk.crossvalidate(y, X, Kfolds=4, folds=folds, ctx=ctx, devices=[0])
    two = bk.crossvalidate(y, X, Kfolds=4, folds=folds,
                           devices=[bk.Context(0, own_stream=True), bk.Context(0, own_stream=True)])
    assert one["devices"] == [0] and two["devices"] == [0, 0]
    for k in ["R2_is", "R2_oos", "MSE_is", "MSE_oos", "R2AME_is", "R2AME_oos", "MSE_AME_is", "MSE_AME_oos"]:
        assert rel(seq[k], ref[k]) < TOL, k
        assert one[k] == seq[k], k
        assert two[k] == seq[k], k                      # deterministic kernels: bitwise the same on any context
    c0 = two["fold_1"]["trained"]["_ctx"]
    assert two["fold_3"]["trained"]["_ctx"] is c0 and two["fold_2"]["trained"]["_ctx"] is not c0
    # (iii) the same thing asked for by number: two folds at a time on the one device
    pair = bk.crossvalidate(y, X, Kfolds=4, folds=folds, ctx=ctx, folds_per_device=2)
    assert pair["devices"] == [0, 0]
    for k in ["R2_is", "R2_oos", "MSE_is", "MSE_oos", "R2AME_is", "R2AME_oos"]:
        assert pair[k] == seq[k], k
    with pytest.raises(ValueError):
        bk.crossvalidate(y, X, Kfolds=4, folds=folds, ctx=ctx, folds_per_device=3)


@pytest.mark.gpu
def test_dist_dense_sharded_eigen_world1_hip_backend(ctx):
    """bigkrls_fit_dist with the dense eigensolver whose stage 1 is partitioned by column blocks (SURVEY 8(e) "Eigen,
    dense"; one rank: the broadcast / all-gather per panel, the symmetric update of the diagonal block, the split
    back-transform), Neig = N and Neig < N, and the replicated decomposition: against bigKRLS() on the same GPU."""
    import bigkrls_amd as bk
    from bigkrls_amd import dist as bkdist
    n, p = 1500, 6
    X, y = orc.synth(n, p, 53)
    one = bk.bigKRLS(y, X, ctx=ctx)
    for mode, neig in (("dense", None), ("dense", 100), ("replicated", None)):
        kw = dict(Neig=neig) if neig else {}
        ref1 = one if neig is None else bk.bigKRLS(y, X, ctx=ctx, **kw)
        out = bkdist.bigKRLS_dist(y, X, ctx=ctx, eigen_mode=mode, **kw)
        assert out["lastkeeper"] == ref1["lastkeeper"] and abs(out["lambda"] - ref1["lambda"]) <= 1e-9 * ref1["lambda"]
        for k in ("coeffs", "yfitted", "derivatives", "var.avgderivatives", "K.eigenvalues"):
            assert rel(out[k], ref1[k]) < 1e-8, (mode, neig, k)
        assert rel(out["vcov.est.c.cols"].to_numpy(), ref1["vcov.est.c"]) < 1e-8


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_dist_paths_under_a_world1_rccl_group(ctx):
    """Under a torch.distributed `nccl` group of size 1: the communicator is built from the group (unique id from
    rank 0, ncclCommInitRank in the library) and every collective of the fit -- broadcast of the panel strips,
    all-gather of A22 V, of the eigenvector column blocks, of the Lanczos blocks, the all-reduces of the lambda
    search -- is issued by the library through RCCL. Dense-sharded, block-Lanczos and replicated eigen paths."""
    import socket
    import torch
    import torch.distributed as dist
    import bigkrls_amd as bk
    from bigkrls_amd import dist as bkdist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                            device_id=torch.device("cuda", ctx.device_index))
    try:
        X, y = orc.synth(900, 5, 54, binary_last=True)
        ref = orc.fit(y, X, literal=False)
        out = bkdist.bigKRLS_dist(y, X, ctx=ctx)                          # dense, stage 1 by column blocks
        assert_fit_parity(out, ref, squares=False)
        X2, y2 = orc.synth(4500, 6, 52)
        one = bk.bigKRLS(y2, X2, Neig=96, ctx=ctx)
        kr = bkdist.bigKRLS_dist(y2, X2, Neig=96, ctx=ctx, eigen_mode="krylov")
        assert kr["lastkeeper"] == one["lastkeeper"] and abs(kr["lambda"] - one["lambda"]) <= 1e-8 * one["lambda"]
        assert rel(kr["coeffs"], one["coeffs"]) < TOL and rel(kr["derivatives"], one["derivatives"]) < TOL
        rp = bkdist.bigKRLS_dist(y, X, ctx=ctx, eigen_mode="replicated")   # K all-gathered, Q by all-reduce
        assert_fit_parity(rp, ref, squares=False)
    finally:
        bkdist.release_comms()
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("bt2", ["default", "chain"])
def test_two_contexts_fit_concurrently_on_one_gpu(monkeypatch, bt2):
    """Two contexts with their own streams on one device, one host thread each, fitting at the same
    time (include/bigkrls.h "Threading": one context per thread). n = 5000: the persistent kernels of
    both decompositions (24 + 79 workgroups each) are co-resident. Results must be bitwise those of the
    same fits run one after the other. `chain`: the stage-2 back-transform as chains of tasks (the default only
    above n = 8000), whose ticket scheme must not need co-residency either."""
    import threading
    import bigkrls_amd as bk
    if bt2 == "chain":
        monkeypatch.setenv("BIGKRLS_BT2", "chain")
    else:
        monkeypatch.delenv("BIGKRLS_BT2", raising=False)
    data = [orc.synth(5000, 6, 60 + i) for i in range(2)]
    ctxs = [bk.Context(0, own_stream=True) for _ in range(2)]
    seq = [bk.bigKRLS(y, X, ctx=c) for (X, y), c in zip(data, ctxs)]
    par = [None, None]
    errs = []

    def work(i):
        try:
            ctxs[i].torch.cuda.set_device(0)
            par[i] = bk.bigKRLS(data[i][1], data[i][0], ctx=ctxs[i])
        except BaseException as e:          # surfaced below
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for s, p in zip(seq, par):
        assert p["lambda"] == s["lambda"] and p["lastkeeper"] == s["lastkeeper"]
        for k in ("coeffs", "yfitted", "derivatives", "var.avgderivatives", "K.eigenvalues"):
            assert np.array_equal(p[k], s[k]), k
