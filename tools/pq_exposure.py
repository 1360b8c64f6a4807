"""How much of stage 1 is the panel QR's? (development experiment) Runs the C3-size fit with the normal library and
with the experiment build whose panel QR is a memset (tools/_ab/libbigkrls_pqstub.so, -DBK_PQ_STUB; result garbage,
finite), printing the library's per-phase wall-clock (BIGKRLS_VERBOSE)."""
import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, os, time
sys.path.insert(0, %r)
import bigkrls_amd._lib as L
L.LIB_PATH = sys.argv[1]
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n, p = int(sys.argv[2]), int(sys.argv[3])
X, y = synth(n, p, 103)
import numpy as np
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
ctx = bk.Context(0)
from bigkrls_amd import ops
Xd = ctx.from_numpy(np.asfortranarray(Xs))
K = ops.bGaussKernel(Xd)
for rep in range(3):
    if rep == 2: os.environ["BIGKRLS_VERBOSE"] = "1"
    Kc = K.copy()
    ctx.sync(); t0 = time.perf_counter()
    try:
        ops.bEigen(Kc, eigtrunc=0.001)
    except Exception as e:
        print("eigen raised:", str(e)[:100])
    ctx.sync(); print("%%s rep %%d eigen %%.4f s" %% (os.path.basename(sys.argv[1]), rep, time.perf_counter() - t0), flush=True)
    del Kc
''' % root
n, p = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("20000", "20")
for lib, env in (("bigkrls_amd/libbigkrls_hip.so", {}), ("tools/_ab/libbigkrls_pqstub.so", {"BIGKRLS_PQ_STUB": "1"})):
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, "-c", child, os.path.join(root, lib), n, p], env=e, check=False)
