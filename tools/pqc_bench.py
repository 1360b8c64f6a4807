"""pq_chol (CholeskyQR2 + Householder reconstruction panel kernel) phase by phase: one build of the library per phase
(tools/pqc_bench.sh build), each timed on a random m x 64 panel in its own process; prints cumulative and per-phase
times. Development tool. python tools/pqc_bench.py [m ...]"""
import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, os, ctypes as C
sys.path.insert(0, %r)
import bigkrls_amd._lib as L
L.LIB_PATH = sys.argv[1]
import bigkrls_amd as bk
ctx = bk.Context(0)
lib = L.load()
lib.bk_pqc_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p]
lib.bk_pqc_bench.restype = C.c_int
out = []
for m in sys.argv[2:]:
    us = C.c_double()
    rc = lib.bk_pqc_bench(ctx.handle, int(m), 30, 0, C.byref(us), None)
    out.append("%%.1f" %% us.value)
print(" ".join(out))
''' % root
names = {1: "load", 2: "Gram 0", 3: "all-reduce 0", 4: "Cholesky 0", 5: "substitution 0", 6: "Gram 1", 7: "all-reduce 1",
         8: "Cholesky 1", 9: "substitution 1", 10: "broadcast Q1", 11: "LU", 12: "R product", 13: "substitution 2",
         0: "stores (whole kernel)"}
ms = sys.argv[1:] or ["512", "5120", "12000", "20000"]
print("m:".ljust(26) + "".join(f"{m:>16s}" for m in ms))
prev = [0.0] * len(ms)
for ph in list(range(1, 14)) + [0]:
    lib = os.path.join(root, "tools", "_ab", "pqc", f"libbigkrls_stop{ph}.so")
    r = subprocess.run([sys.executable, "-c", child, lib] + ms, capture_output=True, text=True)
    try:
        vals = [float(x) for x in r.stdout.strip().split()[-len(ms):]]
    except Exception:
        print(names[ph], "failed:", r.stdout[-200:], r.stderr[-300:]); continue
    print(f"after {names[ph]:20s}" + "".join(f"{v:8.1f} (+{v - p:5.1f})" for v, p in zip(vals, prev)))
    prev = vals
