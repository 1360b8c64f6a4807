"""Where one middle location of bc_resident (stage 2, band -> tridiagonal) spends its sweep: shader clocks per phase,
from a -DBK_BC_PROF build (tools/build_bc_prof.sh). python tools/bc_prof.py [N]"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
import bigkrls_amd._lib as L
L.LIB_PATH = os.path.join(HERE, "libbigkrls_bcprof.so")
import bigkrls_amd as bk
from bigkrls_amd import ops
from bigkrls_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = bk.Context(0)
X, _ = synth(n, 8, 5)
Xs = (X - X.mean(0)) / X.std(0, ddof=1)
K = ops.bGaussKernel(ctx.from_numpy(Xs), 8.0)
lib = L.load()
acc = (ctypes.c_longlong * 8)()
ops.bEigen(K, None, 0.001)          # warm-up
lib.bigkrls_debug_bc_prof(acc, 1)
ops.bEigen(K, None, 0.001)
lib.bigkrls_debug_bc_prof(acc, 0)
names = ["wait for the reflector (+ its barrier)", "partial products p, y", "wait for the entering column (+ barrier)",
         "reflector generation, q, sends (+ barrier)", "u, z, window update",
         "new last row / first column export (bc_regwin)", "-", "end-of-sweep barrier, loop"]
a = np.array(list(acc), dtype=float)
sweeps = n - 2 - (n // 64 // 3) * 64          # sweeps the profiled location (a third of the way down) takes part in
tot = a.sum() - a[6] - (a[5] if os.environ.get('BIGKRLS_BC_PROF_FLIGHT') else 0)
print(f"N={n}: location {n // 64 // 3} of {n // 64}, ~{sweeps} sweeps, {tot / sweeps:.0f} clocks per sweep "
      f"({tot / sweeps / 2.1e3:.2f} us at 2.1 GHz)")
for k, (nm, v) in enumerate(zip(names, a)):
    if v > 0 and k != 6:
        print(f"  {nm:46s} {v / sweeps:8.0f} clocks  {100 * v / tot:5.1f} %")
if a[6] > 0:
    print(f"  reflector: send -> receipt at the next location {10.0 * a[6] / sweeps:8.0f} ns per sweep (100 MHz stamps)")
if os.environ.get("BIGKRLS_BC_PROF_FLIGHT"):
    print(f"  column:    send -> receipt at the previous location {10.0 * a[5] / sweeps:8.0f} ns per sweep")
