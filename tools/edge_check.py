"""Small and degenerate fits (lastkeeper = 1, n below / at / above a tile edge) against the oracle: the variance
matrices come from the mirrored lower-triangle product (development check). With lastkeeper = 1 the leave-one-out loss does not
depend on lambda at all (c_i / Ginv_ii = q'y / q_i), so the golden-section search is a coin toss between its two first
probes and lambda -- with it V -- may differ from the oracle's at such a case (n = 257, eigtrunc = 0.5): not an error."""
import sys, numpy as np
sys.path.insert(0, '.')
import bigkrls_amd as bk
from oracle import krls_oracle as orc
ctx = bk.Context(0)
for (n, p, et) in [(130, 3, 0.9), (257, 4, 0.5), (500, 5, 0.99), (64, 2, 0.0), (129, 3, 0.0)]:
    X, y = orc.synth(n, p, 5)
    out = bk.bigKRLS(y, X, eigtrunc=et, ctx=ctx)
    ref = orc.fit(y, X, eigtrunc=et)
    V = out["vcov.est.c"].to_numpy() if hasattr(out["vcov.est.c"], "to_numpy") else np.asarray(out["vcov.est.c"])
    Vr = np.asarray(ref["vcov.est.c"])
    Vf = out["vcov.est.fitted"].to_numpy() if hasattr(out["vcov.est.fitted"], "to_numpy") else np.asarray(out["vcov.est.fitted"])
    print(n, p, et, "lastkeeper", out["lastkeeper"], ref["lastkeeper"],
          "vcov.c rel", float(np.abs(V - Vr).max() / np.abs(Vr).max()),
          "vcov.fitted rel", float(np.abs(Vf - np.asarray(ref["vcov.est.fitted"])).max() / np.abs(np.asarray(ref["vcov.est.fitted"])).max()),
          "sym", float(np.abs(V - V.T).max()), "lambda", out["lambda"], ref["lambda"], "coeffs", float(np.abs(out["coeffs"].ravel() - np.asarray(ref["coeffs"]).ravel()).max()))
