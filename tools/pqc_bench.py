"""pq_chol (CholeskyQR2 + Householder reconstruction panel kernel) phase by phase: needs the -DBK_PQC_PROF build of the
library (tools/_ab/libbigkrls_pqcprof.so), whose kernel returns after a given phase; prints cumulative and
per-phase times for several panel heights. Development tool. python tools/pqc_bench.py [m ...]"""
import sys, os, ctypes as C
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bigkrls_amd._lib as L
L.LIB_PATH = os.path.join(root, "tools", "_ab", "libbigkrls_pqcprof.so")
import bigkrls_amd as bk
ctx = bk.Context(0)
lib = L.load()
lib.bk_pqc_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p]
lib.bk_pqc_bench.restype = C.c_int
names = {1: "load", 2: "Gram 0", 3: "all-reduce 0", 4: "Cholesky 0", 5: "substitution 0", 6: "Gram 1", 7: "all-reduce 1",
         8: "Cholesky 1", 9: "substitution 1", 10: "broadcast Q1", 11: "LU", 12: "R product", 13: "substitution 2",
         0: "stores (whole kernel)"}
ms = [int(a) for a in sys.argv[1:]] or [512, 4096, 10240, 20000]
for m in ms:
    prev = 0.0
    print(f"m = {m} ({(m + 255) // 256} workgroups)")
    for ph in list(range(1, 14)) + [0]:
        us = C.c_double()
        rc = lib.bk_pqc_bench(ctx.handle, m, 20, ph, C.byref(us), None)
        print(f"  after {names[ph]:22s} {us.value:8.1f} us   (+{us.value - prev:6.1f})" + (f"   status {rc}" if rc else ""))
        prev = us.value
