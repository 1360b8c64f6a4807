"""Two contexts (own streams) fitting N x N problems concurrently on ONE GPU from two host threads:
at N = 20000 the persistent kernels of the two decompositions cannot all be co-resident, so the
watchdog / in-process retry path runs under real contention (development probe).
python tools/contention_check.py [N]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
data = [synth(n, 10, 70 + i) for i in range(2)]
ctxs = [bk.Context(0, own_stream=True) for _ in range(2)]
t0 = time.perf_counter()
seq = [bk.bigKRLS(y, X, ctx=c, derivative=False) for (X, y), c in zip(data, ctxs)]
print(f"sequential: {time.perf_counter()-t0:.2f} s", flush=True)
par, errs = [None, None], []
def work(i):
    try:
        ctxs[i].torch.cuda.set_device(0)
        par[i] = bk.bigKRLS(data[i][1], data[i][0], ctx=ctxs[i], derivative=False)
    except BaseException as e:
        errs.append(repr(e))
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
print(f"concurrent: {time.perf_counter()-t0:.2f} s, errors: {errs}", flush=True)
for s, p in zip(seq, par):
    if p is not None:
        print("lambda equal:", p["lambda"] == s["lambda"], "max rel diff coeffs:",
              float(np.max(np.abs(p["coeffs"] - s["coeffs"])) / np.max(np.abs(s["coeffs"]))), flush=True)
