"""Host logic of the fold-parallel crossvalidate (last row: "replicas only --
one fold per GPU"; the reference's fold loop is R/bigKRLS.R:1268-1282): the scheduler that hands
whole folds to one worker thread per context, with CPU doubles for the fit and the prediction (the
oracle's, so the statistics are the real ones). No GPU needed."""
import threading
import time

import numpy as np

from oracle import krls_oracle as orc


class FakeContext:
    def __init__(self, idx):
        self.device_index = idx
        self.jobs = []


def _doubles(monkeypatch, delay=0.0):
    from bigkrls_amd import api
    seen = []

    def fit(y, X, ctx=None, **kw):
        ctx.jobs.append(len(y))
        seen.append((threading.current_thread().name, ctx.device_index))
        time.sleep(delay)
        out = api.BigKRLS(orc.fit(y, X, literal=False, return_squares=False, **kw))
        out["_ctx"] = ctx
        return out

    def predict(obj, newdata):
        return api.BigKRLSPredicted(orc.predict(obj, newdata))

    monkeypatch.setattr(api, "_cv_fit", fit)
    monkeypatch.setattr(api, "_cv_predict", predict)
    return seen


def test_folds_on_two_contexts_equal_the_sequential_loop(monkeypatch):
    import bigkrls_amd as bk
    seen = _doubles(monkeypatch, delay=0.05)
    X, y = orc.synth(120, 3, 5)
    folds = (np.arange(120) % 5) + 1
    a, b = FakeContext(0), FakeContext(1)
    par = bk.crossvalidate(y, X, Kfolds=5, folds=folds, devices=[a, b])
    seq = bk.crossvalidate(y, X, Kfolds=5, folds=folds, devices=[FakeContext(0)])
    ref = orc.crossvalidate_kfolds(y, X, folds, literal=False)
    for k in ["R2_is", "R2_oos", "MSE_is", "MSE_oos", "R2AME_is", "R2AME_oos", "MSE_AME_is", "MSE_AME_oos"]:
        assert par[k] == seq[k]                                     # identical, in fold order
        assert np.allclose(par[k], ref[k], rtol=1e-12, atol=0)
    assert par["devices"] == [0, 1]
    # fold k ran on context (k - 1) mod 2, each context in its own thread, both were used
    assert len(a.jobs) == 3 and len(b.jobs) == 2
    by_dev = {}
    for name, dev in seen[:5]:
        by_dev.setdefault(dev, set()).add(name)
    assert all(len(v) == 1 for v in by_dev.values()) and by_dev[0] != by_dev[1]
    for k in range(1, 6):
        assert par[f"fold_{k}"]["trained"]["_ctx"] is (a if (k - 1) % 2 == 0 else b)


def test_fold_errors_surface_in_the_caller(monkeypatch):
    import pytest
    import bigkrls_amd as bk
    from bigkrls_amd import api
    _doubles(monkeypatch)

    def bad_fit(y, X, ctx=None, **kw):
        if ctx.device_index == 1:
            raise RuntimeError("device 1 failed")
        return api.BigKRLS(orc.fit(y, X, literal=False, return_squares=False))

    monkeypatch.setattr(api, "_cv_fit", bad_fit)
    X, y = orc.synth(60, 2, 6)
    with pytest.raises(RuntimeError, match="device 1 failed"):
        bk.crossvalidate(y, X, Kfolds=4, devices=[FakeContext(0), FakeContext(1)], seed=3)


def test_folds_per_device_is_one_or_two():
    """crossvalidate(folds_per_device=...): a second context per device, never more (three decompositions side by
    side can keep each other's persistent kernels off the GPU)."""
    import pytest
    from bigkrls_amd import api
    a = FakeContext(0)
    assert api._fold_contexts(a, [a]) == [a]
    for bad in (0, 3, 8):
        with pytest.raises(ValueError):
            api._fold_contexts(a, [a], folds_per_device=bad)
