"""Timeline of bc_regwin at locations 0..3 and two in the middle over a few sweeps (100 MHz stamps written inside the
kernel; -DBK_BC_PROF build, tools/build_bc_prof.sh). python tools/bc_trace.py [N]"""
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
ops.bEigen(K, None, 0.001)
ops.bEigen(K, None, 0.001)
LOCS, SW, EV = 6, 12, 10
buf = (ctypes.c_ulonglong * (LOCS * SW * EV))()
lib.bigkrls_debug_bc_trace(buf, LOCS * SW * EV)
a = np.array(list(buf), dtype=np.int64).reshape(LOCS, SW, EV)
names = ["top", "v in", "bar1", "prod+bar", "await col", "col in", "v' ready", "cu ready", "bar4", "end"]
t0 = a[0, 2, 0]
nloc = (n - 2) // 64 + 1
labels = ["0", "1", "2", "3", str(nloc // 3), str(nloc // 3 + 1)]
print("times in ns relative to location 0's sweep-2002 top; one row per (location, sweep)")
print("loc sweep " + " ".join(f"{nm:>9s}" for nm in names))
for li in range(LOCS):
    for sw in range(2, 8):
        row = a[li, sw]
        print(f"{labels[li]:>3s} {2000 + sw:5d} " + " ".join(f"{10 * (int(v) - int(t0)):9d}" if v else f"{'-':>9s}" for v in row))
for li in range(LOCS):
    per = np.diff(a[li, 2:10, 0]) * 10.0
    print(f"location {labels[li]}: sweep period {per.mean():.0f} ns; v in -> v' ready {10 * np.mean(a[li, 2:10, 6] - a[li, 2:10, 1]):.0f} ns; "
          f"col in -> v' ready {10 * np.mean(a[li, 2:10, 6] - a[li, 2:10, 5]):.0f} ns; v in -> cu ready {10 * np.mean(a[li, 2:10, 7] - a[li, 2:10, 1]):.0f} ns; "
          f"await col {10 * np.mean(a[li, 2:10, 5] - a[li, 2:10, 4]):.0f} ns; top -> v in {10 * np.mean(a[li, 2:10, 1] - a[li, 2:10, 0]):.0f} ns")
for li in range(LOCS - 1):
    if labels[li + 1] == str(int(labels[li]) + 1):
        print(f"hop {labels[li]} -> {labels[li + 1]}: v' ready at {labels[li]} -> v in at {labels[li + 1]}: {10 * np.mean(a[li + 1, 2:10, 1] - a[li, 2:10, 6]):.0f} ns; "
              f"cu ready at {labels[li + 1]} (sweep s) -> col in at {labels[li]} (sweep s+1): {10 * np.mean(a[li, 3:10, 5] - a[li + 1, 2:9, 7]):.0f} ns")
