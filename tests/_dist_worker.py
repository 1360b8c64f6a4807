"""Worker for tests/test_dist_gloo.py (CPU, gloo): the rank object of the multi-GPU fit over caller-supplied
collectives -- bigkrls_comm_create_callbacks, bigkrls_comm_check, bigkrls_comm_rank, bigkrls_fit_dist_rows
(include/bigkrls.h) -- on host buffers, no GPU. The collectives are the ones bigkrls_amd.dist installs for its
host-staged mode (torch.distributed on CPU tensors); every rank checks what the library handed back."""
import ctypes as C
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bigkrls_amd import _lib, dist as bkdist  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = bkdist.comm_callbacks(None)                       # no context: host memory
    r, w = C.c_int32(-1), C.c_int32(-1)
    _lib.call("bigkrls_comm_rank", comm.handle, C.byref(r), C.byref(w))
    assert (r.value, w.value) == (rank, world)
    # ---- every collective once, on a buffer laid out as bigkrls_comm_check documents ---------------------------
    count = 5
    buf = np.zeros((4 + world) * count)
    base = np.arange(count, dtype=np.float64)
    buf[0:count] = base + 10.0 * rank                        # all-reduce (sum)
    buf[count:2 * count] = base * (1.0 if rank % 2 else -1.0) + rank   # all-reduce (min)
    buf[2 * count:3 * count] = 100.0 * rank + base           # all-gather
    buf[3 * count:4 * count] = 7.0 * rank + base             # broadcast from the last rank
    _lib.call("bigkrls_comm_check", comm.handle, buf.ctypes.data_as(C.c_void_p), count)
    want_sum = world * base + 10.0 * sum(range(world))
    want_min = np.min([base * (1.0 if q % 2 else -1.0) + q for q in range(world)], axis=0)
    want_gather = np.concatenate([100.0 * q + base for q in range(world)])
    assert np.array_equal(buf[0:count], want_sum), buf[0:count]
    assert np.array_equal(buf[count:2 * count], want_min), buf[count:2 * count]
    assert np.array_equal(buf[4 * count:], want_gather)
    assert np.array_equal(buf[3 * count:4 * count], 7.0 * (world - 1) + base)
    assert np.array_equal(buf[2 * count:3 * count], 100.0 * rank + base)          # the send part is untouched
    # ---- the rows a rank owns: the library's plan against the documented rule -----------------------------------
    for n, neig in ((101, 0), (257, 0), (1000, 0), (20000, 0), (50000, 512), (100000, 1024), (130, 0), (64 * world + 1, 0)):
        opt = _lib.FitOptions()
        opt.struct_bytes = C.sizeof(_lib.FitOptions)
        opt.neig = neig
        r0, r1 = C.c_int64(-1), C.c_int64(-1)
        _lib.call("bigkrls_fit_dist_rows", comm.handle, n, C.byref(opt), C.byref(r0), C.byref(r1))
        ne = min(n, neig) if neig > 0 else n
        krylov = ne * 8 <= n and n >= 16384
        align = 1 if (krylov or n <= 256) else 64             # dense path: 64-column panels must not straddle ranks
        nb, parts = bkdist.partition(n, world, align)
        assert (r0.value, r1.value) == parts[rank], (n, neig, r0.value, r1.value, parts[rank])
        if align == 64:
            assert r0.value % 64 == 0
    # all ranks together cover 0..n without overlap (checked through the group)
    import torch
    t = torch.tensor([r1.value - r0.value], dtype=torch.int64)
    dist.all_reduce(t)
    assert int(t.item()) == 64 * world + 1
    # ---- a failing callback comes back as an error code, not as an exception through C --------------------------
    table = _lib.Collectives()
    table.struct_bytes = C.sizeof(_lib.Collectives)
    bad_ar = _lib.ALL_REDUCE_FN(lambda user, b, c, op: 3)
    ok_ag = _lib.ALL_GATHER_FN(lambda user, s, rcv, c: 0)
    ok_bc = _lib.BROADCAST_FN(lambda user, b, c, root: 0)
    table.all_reduce, table.all_gather, table.broadcast = bad_ar, ok_ag, ok_bc
    h = C.c_void_p()
    _lib.call("bigkrls_comm_create_callbacks", None, world, rank, C.byref(table), C.byref(h))
    try:
        _lib.call("bigkrls_comm_check", h, buf.ctypes.data_as(C.c_void_p), count)
        raise AssertionError("a failing collective must fail the call")
    except _lib.BigKRLSError as e:
        assert e.code == _lib.EHIP and "all_reduce returned 3" in str(e)
    _lib.call("bigkrls_comm_destroy", h)
    table.struct_bytes = 8
    try:
        _lib.call("bigkrls_comm_create_callbacks", None, world, rank, C.byref(table), C.byref(h))
        raise AssertionError("a table of the wrong size must be refused")
    except _lib.BigKRLSError as e:
        assert e.code == _lib.EINVAL
    comm.close()
    dist.barrier()
    print(f"rank {rank}/{world} OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
