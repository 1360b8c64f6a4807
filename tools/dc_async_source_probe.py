"""When does the runtime read a pageable host vector that is the source of hipMemcpyAsync -- when the call returns, or
when the copy executes? Test build of the library, BIGKRLS_FAULT=dc_gd_clobber: a spin kernel keeps the stream 2 ms
behind, and the descriptor vector of each level's batched product in the divide & conquer is zeroed right after the
launch. Result on this runtime (profiles/r05/r05b_dc_async_source_probe.log): the decomposition does not change by a
bit -- the copy was taken at the call, as tools/pageable_h2d_probe.hip shows at every size.
    python tools/dc_async_source_probe.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import bigkrls_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "tests", "capi", "libbigkrls_hip_fault.so")
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
n, p = 3000, 5
X, _ = synth(n, p, 9)
K = ops.bGaussKernel(ctx.from_numpy((X - X.mean(0)) / X.std(0, ddof=1)), float(p))
good = ops.bEigen(K, None, -1.0)
for mode in ("dc_lag", "dc_gd_clobber"):
    os.environ["BIGKRLS_FAULT"] = mode
    try:
        e = ops.bEigen(K, None, -1.0)
        dv = float(np.max(np.abs(np.asarray(e.values) - np.asarray(good.values))) / abs(good.values[0]))
        Q = e.vectors.to_numpy()
        orth = float(np.max(np.abs(Q.T @ Q - np.eye(Q.shape[1]))))
        print(f"BIGKRLS_FAULT={mode}: eigenvalues differ from the undisturbed run by {dv:.3e} (relative), |Q'Q - I| = {orth:.3e}, "
              f"kept {e.lastkeeper} vs {good.lastkeeper}")
    except Exception as ex:
        print(f"BIGKRLS_FAULT={mode}: {type(ex).__name__}: {str(ex)[:200]}")
