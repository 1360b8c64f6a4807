"""The row-block path (bigkrls_amd.dist) with the HIP backend at WORLD_SIZE > 1 on ONE GPU.

RCCL refuses two ranks on one device, and this pool hands out single-GPU boxes, so the multi-rank code of
the HIP backend (column blocks with offsets, the partitioned stage 1, the split back-transform) is driven
here through a gloo group: every rank is its own process with its own context on device 0, collectives on
device tensors are staged through host memory by the shim below (test only; the product path uses RCCL).
Checks every rank's result against the single-process fit.

    python tests/_dist_world_gpu.py [N] [P] [WORLD] [--krylov NEIG]

The launcher never touches the GPU: it only starts the rank processes. tests/conftest.py starts the launchers at
session start (before the pytest process itself initialises the GPU) and tests/test_gpu_dist_world.py collects them.
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def launcher():
    args = [a for a in sys.argv[1:]]
    pos = [a for a in args if not a.startswith("--") and a.isdigit()]
    world = int(pos[2]) if len(pos) > 2 else 2
    port = str(29600 + os.getpid() % 200)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, BIGKRLS_DIST_WORKER="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + args, env=env))
    rc = 0
    for p in procs:
        rc |= p.wait()
    sys.exit(rc)


class HostStagedDist:
    """torch.distributed with device tensors staged through the host (gloo)."""

    def __init__(self, dist, torch):
        self._d, self._t = dist, torch
        self.ReduceOp = dist.ReduceOp

    def is_initialized(self):
        return self._d.is_initialized()

    def get_rank(self):
        return self._d.get_rank()

    def get_world_size(self):
        return self._d.get_world_size()

    def barrier(self):
        self._d.barrier()

    def broadcast(self, t, src=0):
        h = t.detach().cpu().contiguous()
        self._d.broadcast(h, src=src)
        t.copy_(h.view(t.shape))

    def all_reduce(self, t, op=None):
        h = t.detach().cpu().contiguous()
        self._d.all_reduce(h, op=op if op is not None else self._d.ReduceOp.SUM)
        t.copy_(h.view(t.shape))

    def all_gather_into_tensor(self, out, inp):
        hi = inp.detach().cpu().contiguous()
        ho = self._t.empty(out.shape, dtype=out.dtype)
        self._d.all_gather_into_tensor(ho, hi)
        out.copy_(ho)


def worker():
    sys.path.insert(0, ROOT)
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    import bigkrls_amd as bk
    from bigkrls_amd import dist as bkdist
    from bigkrls_amd.synth import synth

    pos = [a for a in sys.argv[1:] if not a.startswith("--") and a.isdigit()]
    n = int(pos[0]) if pos else 3000
    p = int(pos[1]) if len(pos) > 1 else 8
    neig = None
    if "--krylov" in sys.argv:
        neig = int(sys.argv[sys.argv.index("--krylov") + 1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    staged = HostStagedDist(dist, torch)
    bkdist._torch_dist = lambda: (torch, staged)          # see bigkrls_amd/dist.py
    ctx = bk.Context(0)
    X, y = synth(n, p, 103)
    kw = dict(Neig=neig) if neig else {}
    T = {}
    t0 = time.perf_counter()
    out = bkdist.bigKRLS_dist(y, X, ctx=ctx, timings=T, keep_outputs=True, **kw)
    ctx.sync()
    dt = time.perf_counter() - t0
    one = bk.bigKRLS(y, X, ctx=ctx, **kw)

    def rel(a, b):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

    checks = {
        "lambda": rel(out["lambda"], one["lambda"]),
        "coeffs": rel(out["coeffs"], one["coeffs"]),
        "yfitted": rel(out["yfitted"], one["yfitted"]),
        "avgderivatives": rel(out["avgderivatives"], one["avgderivatives"]),
        "var.avgderivatives": rel(out["var.avgderivatives"], one["var.avgderivatives"]),
        "K.eigenvalues[:lastkeeper]": rel(np.asarray(out["K.eigenvalues"])[: out["lastkeeper"]],
                                          np.asarray(one["K.eigenvalues"])[: one["lastkeeper"]]),
    }
    ok = out["lastkeeper"] == one["lastkeeper"] and all(v < 1e-7 for v in checks.values())
    print(f"rank {rank}/{world} N={n} P={p} neig={neig}: {dt:.2f} s lastkeeper {out['lastkeeper']} vs {one['lastkeeper']} "
          + " ".join(f"{k}={v:.1e}" for k, v in checks.items()) + (" OK" if ok else " MISMATCH"), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    if os.environ.get("BIGKRLS_DIST_WORKER") == "1":
        worker()
    else:
        launcher()
