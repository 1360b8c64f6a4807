"""Generates the committed golden vectors from the CPU oracle (run here, in the build
container: `python tests/golden/make_golden.py`). The oracle itself is pinned to the
reference's own golden vector (mtcars) in tests/test_oracle.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import krls_oracle as orc  # noqa: E402


def dump(name, X, y, w, extra=None):
    d = {"X": X, "y": y, "lastkeeper": w["lastkeeper"], "lambda": w["lambda"], "Le": w["Le"],
         "R2": w["R2"], "Neffective": w["Neffective"], "sigmasq": w["sigmasq"]}
    for k in ["coeffs", "yfitted", "derivatives", "var.avgderivatives", "avgderivatives", "K.eigenvalues"]:
        if k in w:
            d[k.replace(".", "_")] = np.asarray(w[k])
    d["K_row0"] = w["K"][0].copy()
    d["vcov_c_diag"] = np.diag(w["vcov.est.c"]).copy()
    d["vcov_fitted_diag"] = np.diag(w["vcov.est.fitted"]).copy()
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, {k: np.asarray(v).shape for k, v in d.items()})


if __name__ == "__main__":
    X, y = orc.synth(500, 5, 101)
    tr = orc.LambdaTrace(0, 0)
    w = orc.fit(y, X, literal=True, trace=tr)
    dump("c1_n500_p5.npz", X, y, w, {"probes": np.array(tr.probes)})
    X, y = orc.synth(500, 6, 2018, binary_last=True)
    w = orc.fit(y, X, eigtrunc=0.01, literal=True)
    dump("n500_p6_binary_trunc01.npz", X, y, w)
    X, y = orc.synth(32, 4, 32)
    w = orc.fit(y, X, literal=True)
    pr = orc.predict(w, X[:8] + 0.1, se_pred=True)
    dump("n32_p4.npz", X, y, w, {"pred": pr["predicted"], "se_pred": pr["se.pred"]})
