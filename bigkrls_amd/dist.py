"""Multi-GPU bigKRLS(): one process per GPU, the partitioned fit and every collective INSIDE the library
(`bigkrls_fit_dist`, csrc/fit.hip + csrc/dist.hip). This module only builds the rank object
(`Comm`, a `bigkrls_comm*`) from the process group the caller already has and calls the fit:

  * RCCL (the product path): rank 0 asks the library for a unique id (`bigkrls_comm_unique_id` = ncclGetUniqueId),
    the id travels through torch.distributed's object broadcast, every rank calls `bigkrls_comm_create`
    (= ncclCommInitRank). Without a process group the communicator has one rank: the same calls on one GPU.
  * host-staged callbacks (tests): `bigkrls_comm_create_callbacks` with collectives that copy the device buffer to
    the host, run the gloo collective of the default process group and copy back -- RCCL refuses two ranks on one
    device, this lets WORLD_SIZE > 1 share ONE GPU (tests/_dist_world_gpu.py) and drives the callback table on CPU
    (tests/test_dist_gloo.py, host buffers, no GPU).

Partition (the library's): rank r owns rows [r0, r1) of K, Q and V; K is symmetric, so the row block is stored as the
contiguous column block K[:, r0:r1) and K is never gathered. Kernel build: no exchange. Eigen: block Lanczos with
sharded K B_j products and one all-gather of an N x 128 block per step (Neig << N), or the dense path with stage 1
partitioned by column blocks (one broadcast of the panel strip and one all-gather of A22 V per panel, all-gather of
the back-transformed eigenvector column blocks). Lambda search: one all-reduce of Q'y, one scalar per probe.
Coefficients, fitted values, marginal effects: one all-gather each. Variance matrices: kept sharded.

The reference's parallel path is a PSOCK cluster over the derivative columns (R/bigKRLS.R:337-363).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _lib
from .api import BigKRLS, bigKRLS
from .device import Context


def _torch_dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def trace_hash(a) -> int:
    """The 64-bit hash of csrc/trace.hip (sum over the elements of a position-dependent mix of their bits, modulo
    2^64) of a float64 array: equal to what the library logs for the same bytes on the device."""
    b = np.ascontiguousarray(a, dtype=np.float64).reshape(-1).view(np.uint64)
    with np.errstate(over="ignore"):
        x = b ^ (np.arange(b.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
        return int(x.sum(dtype=np.uint64))


def partition(n: int, world: int, align: int = 1):
    """The library's row blocks (csrc/dist.hip, dist_partition): nb = ceil(n / world) rounded up to a multiple of
    `align`; the last ranks may be short or empty."""
    nb = (n + world - 1) // world
    nb = (nb + align - 1) // align * align
    return nb, [(min(r * nb, n), min((r + 1) * nb, n)) for r in range(world)]


class Comm:
    """One rank of a multi-GPU job: a `bigkrls_comm*` and whatever must stay alive with it."""

    def __init__(self, ctx: Optional[Context], handle, world: int, rank: int, kind: str, keep=None):
        self.ctx, self.handle, self.world, self.rank, self.kind, self._keep = ctx, handle, world, rank, kind, keep

    def rank_count(self):
        """(rank, nranks) as the library's rank object reports them (`bigkrls_comm_rank`)."""
        r, w = C.c_int32(-1), C.c_int32(-1)
        _lib.call("bigkrls_comm_rank", self.handle, C.byref(r), C.byref(w))
        return int(r.value), int(w.value)

    def close(self):
        if self.handle is not None:
            _lib.call("bigkrls_comm_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_rccl(ctx: Context) -> Comm:
    """RCCL communicator over the ranks of the default torch.distributed group (one rank without a group)."""
    torch, dist = _torch_dist()
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    uid = C.create_string_buffer(128)
    if rank == 0:
        _lib.call("bigkrls_comm_unique_id", uid)
    if world > 1:
        box = [uid.raw if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = C.create_string_buffer(box[0], 128)
    h = C.c_void_p()
    with ctx.on_stream():
        _lib.call("bigkrls_comm_create", ctx.handle, world, rank, uid, C.byref(h))
    return Comm(ctx, h, world, rank, "rccl")


def comm_callbacks(ctx: Optional[Context], group_ops=None) -> Comm:
    """Communicator whose collectives are callbacks into this process: the buffer (device memory with a context,
    host memory without one) is staged through a numpy array and reduced / gathered / broadcast by `group_ops`
    (default: torch.distributed on CPU tensors, i.e. the gloo group the caller initialised)."""
    torch, dist = _torch_dist()
    ops = group_ops or dist
    world = ops.get_world_size() if ops.is_initialized() else 1
    rank = ops.get_rank() if ops.is_initialized() else 0
    lib = _lib.load()

    def fetch(ptr, count):
        h = np.empty(count, dtype=np.float64)
        if ctx is not None:
            _lib.check(lib.bigkrls_d2h(ctx.handle, h.ctypes.data, ptr, 8 * count))
        else:
            C.memmove(h.ctypes.data, ptr, 8 * count)
        return h

    def store(ptr, h):
        h = np.ascontiguousarray(h, dtype=np.float64)
        if ctx is not None:
            _lib.check(lib.bigkrls_h2d(ctx.handle, ptr, h.ctypes.data, 8 * h.size))
        else:
            C.memmove(ptr, h.ctypes.data, 8 * h.size)

    # BIGKRLS_TRACE_DIR (diagnostics; the library logs device-side hashes of the same buffers, csrc/trace.hip): the
    # hash of every buffer as it arrives on the host and as it leaves it, one line per collective
    trace_dir = os.environ.get("BIGKRLS_TRACE_DIR")
    trace_state = {"seq": 0, "f": None}

    def trace(kind, count, h_in, h_out):
        if not trace_dir:
            return
        if trace_state["f"] is None:
            trace_state["f"] = open(os.path.join(trace_dir, f"pid{os.getpid()}.pytrace"), "a")
        trace_state["f"].write(f"{trace_state['seq']} {kind} {count} {trace_hash(h_in):016x} {trace_hash(h_out):016x}\n")
        trace_state["f"].flush()
        trace_state["seq"] += 1

    def guard(fn):
        def wrapped(*a):
            try:
                fn(*a)
                return 0
            except Exception as e:      # an exception must not unwind through the C caller
                print("collective callback failed:", repr(e), flush=True)
                return 1
        return wrapped

    @guard
    def all_reduce(user, buf, count, op):
        h = fetch(buf, count)
        t = torch.from_numpy(h.copy() if trace_dir else h)
        if world > 1:
            ops.all_reduce(t, op=ops.ReduceOp.MIN if op == 1 else ops.ReduceOp.SUM)
        trace("ar", count, h, t.numpy())
        store(buf, t.numpy())

    @guard
    def all_gather(user, send, recv, count):
        t = torch.from_numpy(fetch(send, count))
        out = torch.empty(world * count, dtype=torch.float64)
        if world > 1:
            ops.all_gather_into_tensor(out, t)
        else:
            out.copy_(t)
        trace("ag", count, t.numpy(), out.numpy())
        store(recv, out.numpy())

    @guard
    def broadcast(user, buf, count, root):
        h = fetch(buf, count)
        t = torch.from_numpy(h.copy() if trace_dir else h)
        if world > 1:
            ops.broadcast(t, src=root)
        trace("bc", count, h, t.numpy())
        store(buf, t.numpy())

    table = _lib.Collectives()
    table.struct_bytes = C.sizeof(_lib.Collectives)
    table.user = None
    fns = (_lib.ALL_REDUCE_FN(all_reduce), _lib.ALL_GATHER_FN(all_gather), _lib.BROADCAST_FN(broadcast))
    table.all_reduce, table.all_gather, table.broadcast = fns
    h = C.c_void_p()
    _lib.call("bigkrls_comm_create_callbacks", ctx.handle if ctx is not None else None, world, rank, C.byref(table),
              C.byref(h))
    return Comm(ctx, h, world, rank, "callbacks", keep=(table, fns))


_COMMS: Dict[tuple, Comm] = {}


def get_comm(ctx: Context, collectives: Optional[str] = None) -> Comm:
    """The (cached) communicator of this context for the current default process group. `collectives`: "rccl",
    "host" (callbacks staged through the host), or None = BIGKRLS_DIST_COLLECTIVES, else "host" under a gloo group
    and "rccl" otherwise."""
    torch, dist = _torch_dist()
    kind = collectives or os.environ.get("BIGKRLS_DIST_COLLECTIVES")
    if kind is None:
        kind = "host" if (dist.is_initialized() and dist.get_backend() == "gloo") else "rccl"
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    key = (id(ctx), kind, world, rank, dist.is_initialized())
    c = _COMMS.get(key)
    if c is None or c.handle is None:
        c = comm_rccl(ctx) if kind == "rccl" else comm_callbacks(ctx)
        _COMMS[key] = c
    return c


def release_comms() -> None:
    """Destroy the cached communicators (before the process group they were built on goes away)."""
    for c in list(_COMMS.values()):
        c.close()
    _COMMS.clear()


# a communicator must go before the context it points at: at interpreter exit, before the module globals are torn down
import atexit  # noqa: E402

atexit.register(release_comms)


def bigKRLS_dist(y, X, sigma=None, derivative=True, which_derivatives=None, Neig=None, eigtrunc=None,
                 lambda_=None, L=None, U=None, ctx: Optional[Context] = None, comm: Optional[Comm] = None,
                 timings: Optional[Dict[str, float]] = None, trace=None, keep_outputs=True,
                 eigen_mode: Optional[str] = None, collectives: Optional[str] = None) -> BigKRLS:
    """bigKRLS() over the ranks of the default process group (every rank calls with the same y, X): one call into
    `bigkrls_fit_dist`. Every rank returns the same small outputs; N x N outputs stay sharded (`K.cols`,
    `vcov.est.c.cols`, `vcov.est.fitted.cols` hold this rank's column block, `rows` its row range).
    `eigen_mode`: None (the library's choice: block Lanczos with sharded products when N >= 16384 and Neig <= N/8,
    otherwise the dense path with stage 1 partitioned by column blocks), "krylov" / "dense" to force either,
    "replicated" for the decomposition replicated on every rank (what tiny problems, n <= 256, always use)."""
    from .api import default_context
    if comm is None:
        comm = get_comm(ctx or default_context(), collectives)
    saved = os.environ.get("BIGKRLS_DIST_EIGEN")
    try:
        if eigen_mode is not None:
            os.environ["BIGKRLS_DIST_EIGEN"] = eigen_mode
        return bigKRLS(y, X, sigma=sigma, derivative=derivative, which_derivatives=which_derivatives, Neig=Neig,
                       eigtrunc=eigtrunc, lambda_=lambda_, L=L, U=U, timings=timings, trace=trace, comm=comm,
                       keep_outputs=keep_outputs, noisy=False)
    finally:
        if eigen_mode is not None:
            if saved is None:
                os.environ.pop("BIGKRLS_DIST_EIGEN", None)
            else:
                os.environ["BIGKRLS_DIST_EIGEN"] = saved
