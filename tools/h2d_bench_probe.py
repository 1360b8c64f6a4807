"""Where do the sporadic 70 ms 'h2d' phases of back-to-back fits come from? (development probe)"""
import gc, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
from bigkrls_amd.device import DeviceMatrix
ctx = bk.Context(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
X, y = synth(20000, 20, 103)
gc_log = []
def gc_cb(phase, info):
    if phase == "start":
        gc_cb.t0 = time.perf_counter()
    else:
        gc_log.append((info["generation"], info["collected"], 1e3 * (time.perf_counter() - gc_cb.t0)))
gc.callbacks.append(gc_cb)
def timed(a):
    t0 = time.perf_counter()
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    h = torch.from_numpy(np.ascontiguousarray(a.T))
    t1 = time.perf_counter()
    ms0 = torch.cuda.memory_stats()
    t = torch.empty(h.shape, dtype=torch.float64, device=ctx.device)
    t15 = time.perf_counter()
    if mode == "presync":
        torch.cuda.synchronize()
    t16 = time.perf_counter()
    t.copy_(h)
    t2 = time.perf_counter()
    ms1 = torch.cuda.memory_stats()
    if t2 - t0 > 2e-3 and a.shape[0] > 250:
        print(f"   slow from_numpy{a.shape}: prep {1e3*(t1-t0):.2f} ms, empty {1e3*(t15-t1):.2f} ms, sync {1e3*(t16-t15):.2f} ms, "
              f"copy_ {1e3*(t2-t16):.2f} ms; device allocs {ms1['num_device_alloc']-ms0['num_device_alloc']} "
              f"frees {ms1['num_device_free']-ms0['num_device_free']} reserved {ms1['reserved_bytes.all.current']/2**30:.1f} GiB")
    return DeviceMatrix(ctx, t)
ctx.from_numpy = timed
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    T = {}
    gc_log.clear()
    if mode == "sync":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = bk.bigKRLS(y, X, ctx=ctx, timings=T)
    lk = out["lastkeeper"]
    del out
    if mode == "collect":
        gc.collect()
    print(f"{mode} rep {rep}: total {time.perf_counter()-t0:.4f}s h2d {T['h2d']*1e3:.2f} ms wall {T['wall']:.4f} gc {[g for g in gc_log if g[2] > 1.0]}")
