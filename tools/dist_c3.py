"""C3-shaped fit through the row-block path (bigkrls_amd.dist) on one GPU, with or without a process
group (development probe): python tools/dist_c3.py [N] [P] [--group]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bigkrls_amd as bk
from bigkrls_amd import dist as bkdist
from bigkrls_amd.synth import synth
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 20000
p = int(args[1]) if len(args) > 1 else 20
if "--group" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)),
                            device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))))
ctx = bk.Context(int(os.environ.get("LOCAL_RANK", 0)))
X, y = synth(n, p, 103)
for rep in range(3):
    T = {}
    t0 = time.perf_counter()
    out = bkdist.bigKRLS_dist(y, X, ctx=ctx, timings=T, keep_outputs=False)
    ctx.sync()
    print(f"rep{rep} dist N={n}: {time.perf_counter()-t0:.3f} s lastkeeper={out['lastkeeper']} lambda={out['lambda']:.6f}", {k: round(v, 4) for k, v in T.items()}, flush=True)
T = {}
one = bk.bigKRLS(y, X, ctx=ctx, timings=T)
print("single:", {k: round(v, 4) for k, v in T.items()}, "lambda", one["lambda"])
print("rel diff coeffs", float(np.max(np.abs(out["coeffs"] - one["coeffs"])) / np.max(np.abs(one["coeffs"]))))
