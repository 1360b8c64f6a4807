"""Whole fits side by side on ONE GPU (own context, stream and host thread each): wall-clock of W fits of N x P run
on 1, 2, 4 ... contexts -- what crossvalidate(folds_per_device=...) gains for fold-sized problems (development probe).
python tools/cv_concurrency.py [N] [P] [W]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W = int(sys.argv[3]) if len(sys.argv) > 3 else 8
data = [synth(n, p, 300 + i) for i in range(W)]
for nctx in (1, 2, 3, 4, 6, 8):
    if nctx > W:
        break
    ctxs = [bk.Context(0, own_stream=True) for _ in range(nctx)]
    for c in ctxs:                       # warm-up: workspaces
        bk.bigKRLS(data[0][1], data[0][0], ctx=c)
    res = [None] * W
    def work(i):
        ctxs[i].torch.cuda.set_device(0)
        for j in range(i, W, nctx):
            res[j] = bk.bigKRLS(data[j][1], data[j][0], ctx=ctxs[i])["lambda"]
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nctx)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"N={n} P={p}: {W} fits on {nctx} context(s): {dt*1e3:7.1f} ms  ({dt/W*1e3:6.1f} ms per fit)  lambda[0]={res[0]:.12g}", flush=True)
    del ctxs
