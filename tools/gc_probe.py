"""How many GC-tracked objects does one fit leave behind / how many collections does it trigger? (development probe)"""
import gc, os, sys, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bigkrls_amd as bk
from bigkrls_amd.synth import synth
ctx = bk.Context(0)
X, y = synth(20000, 20, 103)
out = bk.bigKRLS(y, X, ctx=ctx); del out
def stats():
    return [s["collections"] for s in gc.get_stats()]
for rep in range(3):
    s0 = stats(); n0 = len(gc.get_objects())
    out = bk.bigKRLS(y, X, ctx=ctx)
    del out
    s1 = stats(); n1 = len(gc.get_objects())
    print("collections gen0/1/2 during fit:", [b - a for a, b in zip(s0, s1)], "tracked objects", n0, "->", n1)
gc.collect()
before = collections.Counter(type(o).__name__ for o in gc.get_objects())
gc.disable()
out = bk.bigKRLS(y, X, ctx=ctx)
del out
after = collections.Counter(type(o).__name__ for o in gc.get_objects())
gc.enable()
diff = {k: after[k] - before.get(k, 0) for k in after if after[k] - before.get(k, 0) > 5}
print("net new tracked objects by type after one fit (gc disabled):", sorted(diff.items(), key=lambda kv: -kv[1])[:15])
t0 = time.perf_counter(); gc.collect(); print("full collection ms", 1e3 * (time.perf_counter() - t0))
gc.freeze()
t0 = time.perf_counter(); gc.collect(); print("full collection after gc.freeze() ms", 1e3 * (time.perf_counter() - t0))
