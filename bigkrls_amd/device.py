"""Device-buffer handles that replace bigmemory's big.matrix in the host layer.

The reference passes `big.matrix@address` external pointers to its native code
(R/bigKRLS_Rcpp_functions.R:86,168,184,207,225); here a `DeviceMatrix` is a
column-major float64 buffer in HBM whose raw pointer goes to the C ABI.  PyTorch
is used only as the allocator / stream provider (and, in dist.py, for RCCL).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib

# Host <-> HBM transfers go through a pinned staging buffer owned by the context, never through
# the caller's pageable memory: for copies above ~1 MB the HIP runtime pins the user pages in
# place, and when numpy later unmaps that memory (free of a multi-MB array) the kernel driver
# evicts and restores the process's GPU queues -- measured as sporadic 50-100 ms stalls at the
# first synchronisation of the next fit (tools/pending_probe.py).
_STAGE_MIN = 1 << 20     # doubles (8 MB)
_STAGE_MAX = 1 << 24     # doubles (128 MB): larger transfers are pipelined through two halves


def _column_major_filler(a: np.ndarray):
    """fill(dst, off, m): write elements [off, off+m) of the column-major flattening of the 2-D
    array `a` into `dst`, without a full-size temporary whatever the layout of `a`."""
    nrow = a.shape[0]
    if a.flags.f_contiguous:
        src = a.reshape(-1, order="F")          # a view: already in column-major order

        def fill(dst, off, m):
            dst[:] = src[off:off + m]
        return fill

    def fill(dst, off, m):                      # whole and partial columns of the piece
        c0, r0 = divmod(off, nrow)
        pos = 0
        while pos < m:
            take = min(nrow - r0, m - pos)
            if r0 == 0 and take == nrow:
                nc = (m - pos) // nrow
                dst[pos:pos + nc * nrow].reshape(nc, nrow)[...] = a[:, c0:c0 + nc].T
                c0 += nc
                pos += nc * nrow
            else:
                dst[pos:pos + take] = a[r0:r0 + take, c0]
                pos += take
                r0 += take
                if r0 == nrow:
                    r0, c0 = 0, c0 + 1
    return fill


class Context:
    """One GPU, one HIP stream, one workspace pool.

    The stream is torch's current stream of the device at construction (so tensors made elsewhere
    on that stream are ordered with the library's kernels) or, with `own_stream=True`, a new stream
    of this context (several contexts on one device, e.g. one per worker thread). Every allocation
    and transfer of this context is issued with that stream as torch's current one, whatever the
    caller's current stream or thread is: the caching allocator then orders re-use of a freed block
    against the stream the library actually launches on."""

    def __init__(self, device: Optional[int] = None, own_stream: bool = False):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("bigkrls_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.torch = torch
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        # (the process-wide current device is left as the caller had it: a context for GPU 3 must not redirect
        #  the caller's later default_context() or plain torch code to GPU 3)
        with torch.cuda.device(self.device_index):
            self.stream = torch.cuda.Stream(self.device) if own_stream else torch.cuda.current_stream(self.device)
            h = C.c_void_p()
            _lib.call("bigkrls_ctx_create_on_stream", self.device_index, C.c_void_p(self.stream.cuda_stream), C.byref(h))
        self.handle = h
        self._events = []
        self._stage = None
        self._stage_np = None

    def close(self):
        if self.handle is not None:
            self.release_events()
            _lib.load().bigkrls_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.call("bigkrls_ctx_sync", self.handle)

    def release_workspace(self):
        _lib.call("bigkrls_ctx_release_workspace", self.handle)

    def workspace_bytes(self) -> int:
        return int(_lib.load().bigkrls_ctx_workspace_bytes(self.handle))

    def set_profile(self, enable: bool):
        _lib.call("bigkrls_ctx_set_profile", self.handle, int(bool(enable)))

    def get_profile(self, name: str):
        """(total_ms, total_work, launches) of the HIP-event samples of kernel `name`."""
        ms, work, n = C.c_double(), C.c_double(), C.c_int64()
        _lib.call("bigkrls_ctx_get_profile", self.handle, name.encode(), C.byref(ms), C.byref(work),
                  C.byref(n))
        return float(ms.value), float(work.value), int(n.value)

    def counters(self) -> dict:
        """How often this context took a recovery path since it was created (all 0 in a healthy session):
        decompositions redone after the fit's check against K failed, decompositions replayed with per-step
        launches (watchdog / ranks that disagreed), multi-GPU fits whose replicated eigenvalues differed from rank 0's."""
        out = (C.c_int64 * 3)()
        _lib.call("bigkrls_ctx_get_counters", self.handle, out)
        return {"redone": int(out[0]), "replayed": int(out[1]), "replica_diff": int(out[2])}

    # ---- allocation ---------------------------------------------------------
    def on_stream(self):
        """`with ctx.on_stream():` -- torch work inside is issued on the context's stream."""
        return self.torch.cuda.stream(self.stream)

    def empty(self, nrow: int, ncol: int = 1) -> "DeviceMatrix":
        with self.on_stream():
            t = self.torch.empty((int(ncol), int(nrow)), dtype=self.torch.float64, device=self.device)
        return DeviceMatrix(self, t)

    def zeros(self, nrow: int, ncol: int = 1) -> "DeviceMatrix":
        with self.on_stream():
            t = self.torch.zeros((int(ncol), int(nrow)), dtype=self.torch.float64, device=self.device)
        return DeviceMatrix(self, t)

    # ---- transfers ------------------------------------------------------------
    def _staging(self, n: int):
        want = min(max(int(n), _STAGE_MIN), _STAGE_MAX)
        if self._stage is None or self._stage.numel() < want:
            self._stage = self._stage_np = None
            self._stage = self.torch.empty(want, dtype=self.torch.float64, pin_memory=True)
            self._stage_np = self._stage.numpy()
        return self._stage, self._stage_np

    def _pieces(self, n: int):
        """(offset, length, staging offset) of the pieces an n-element transfer is cut into."""
        stage, _ = self._staging(n)
        if n <= stage.numel():
            return [(0, n, 0)]
        half = stage.numel() // 2
        return [(off, min(half, n - off), (i % 2) * half) for i, off in enumerate(range(0, n, half))]

    def upload_into(self, flat_dev, fill):
        """Fill the 1-D device tensor `flat_dev`; `fill(dst, off, m)` writes elements [off, off+m)
        of the source into the host array `dst` (a slice of the pinned staging buffer)."""
        n = int(flat_dev.numel())
        if n == 0:
            return
        pieces = self._pieces(n)
        stage, snp = self._stage, self._stage_np
        done = [None, None]
        with self.on_stream():
            for i, (off, m, so) in enumerate(pieces):
                if done[i % 2] is not None:
                    done[i % 2].synchronize()       # this half's previous copy has left the host
                fill(snp[so:so + m], off, m)
                flat_dev[off:off + m].copy_(stage[so:so + m], non_blocking=True)
                done[i % 2] = self.torch.cuda.Event()
                done[i % 2].record()
            for e in done:
                if e is not None:
                    e.synchronize()

    def download(self, t) -> np.ndarray:
        """Host copy (C order, same shape) of a contiguous device tensor."""
        flat = t.reshape(-1)
        n = int(flat.numel())
        out = np.empty(n, dtype=np.float64)
        if n == 0:
            return out.reshape(tuple(t.shape))
        pieces = self._pieces(n)
        stage, snp = self._stage, self._stage_np
        events = []
        with self.on_stream():
            for i, (off, m, so) in enumerate(pieces):
                if i >= 2:                          # the half is free once its previous piece is on the host
                    poff, pm, pso = pieces[i - 2]
                    events[i - 2].synchronize()
                    out[poff:poff + pm] = snp[pso:pso + pm]
                stage[so:so + m].copy_(flat[off:off + m], non_blocking=True)
                e = self.torch.cuda.Event()
                e.record()
                events.append(e)
            for i in range(max(0, len(pieces) - 2), len(pieces)):
                poff, pm, pso = pieces[i]
                events[i].synchronize()
                out[poff:poff + pm] = snp[pso:pso + pm]
        return out.reshape(tuple(t.shape))

    def from_numpy(self, a: np.ndarray) -> "DeviceMatrix":
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            a = a[:, None]
        nrow, ncol = a.shape
        # (ncol, nrow) C-contiguous == (nrow, ncol) column-major
        with self.on_stream():
            t = self.torch.empty((ncol, nrow), dtype=self.torch.float64, device=self.device)
        fill = _column_major_filler(a)
        self.upload_into(t.view(-1), fill)
        return DeviceMatrix(self, t)

    # ---- HIP-event timing on the context's stream ----------------------------
    def event(self):
        e = C.c_void_p()
        _lib.call("bigkrls_event_create", C.byref(e))
        self._events.append(e)
        _lib.call("bigkrls_event_record", self.handle, e)
        return e

    def release_events(self, events=None):
        """Destroy HIP events made by `event()` (all of them when `events` is None)."""
        for e in list(self._events if events is None else events):
            _lib.call("bigkrls_event_destroy", e)
            if e in self._events:
                self._events.remove(e)

    @staticmethod
    def elapsed_ms(e0, e1) -> float:
        ms = C.c_double()
        _lib.call("bigkrls_event_elapsed_ms", e0, e1, C.byref(ms))
        return float(ms.value)


class DeviceMatrix:
    """Column-major float64 matrix resident in HBM (nrow x ncol, ld == nrow)."""

    def __init__(self, ctx: Context, tensor):
        assert tensor.dim() == 2 and tensor.is_contiguous()
        self.ctx = ctx
        self.t = tensor  # shape (ncol, nrow)

    @property
    def nrow(self) -> int:
        return int(self.t.shape[1])

    @property
    def ncol(self) -> int:
        return int(self.t.shape[0])

    @property
    def shape(self):
        return (self.nrow, self.ncol)

    @property
    def ld(self) -> int:
        return self.nrow

    @property
    def ptr(self) -> C.c_void_p:
        return C.c_void_p(self.t.data_ptr())

    def col_ptr(self, col: int, row: int = 0) -> C.c_void_p:
        return C.c_void_p(self.t.data_ptr() + 8 * (int(col) * self.nrow + int(row)))

    def cols(self, c0: int, c1: int) -> "DeviceMatrix":
        """View of columns [c0, c1) (contiguous in column-major storage)."""
        return DeviceMatrix(self.ctx, self.t[c0:c1])

    def to_numpy(self) -> np.ndarray:
        # (ncol, nrow) C order is (nrow, ncol) column-major: return the Fortran-ordered view
        with self.ctx.on_stream():
            t = self.t.contiguous()
        return self.ctx.download(t).T

    def __getitem__(self, idx):  # R's `K[]` idiom: materialise on the host
        return self.to_numpy()[idx]

    def copy(self) -> "DeviceMatrix":
        with self.ctx.on_stream():
            return DeviceMatrix(self.ctx, self.t.clone())

    def scale_(self, alpha: float) -> "DeviceMatrix":
        _lib.call("bigkrls_dev_scale", self.ctx.handle, self.nrow * self.ncol, float(alpha), self.ptr)
        return self

    def diag(self) -> np.ndarray:
        out = self.ctx.empty(self.nrow, 1)
        _lib.call("bigkrls_dev_diag", self.ctx.handle, self.ptr, self.nrow, self.ld, out.ptr)
        return out.to_numpy().ravel()


def is_device_matrix(x) -> bool:
    return isinstance(x, DeviceMatrix)
