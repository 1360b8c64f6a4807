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


class Context:
    """One GPU, one HIP stream (torch's current stream), one workspace pool."""

    def __init__(self, device: Optional[int] = None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("bigkrls_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.torch = torch
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        torch.cuda.set_device(self.device_index)
        self.device = torch.device("cuda", self.device_index)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        _lib.call("bigkrls_ctx_create_on_stream", self.device_index, C.c_void_p(stream), C.byref(h))
        self.handle = h
        self._events = []

    def close(self):
        if self.handle is not None:
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

    # ---- allocation ---------------------------------------------------------
    def empty(self, nrow: int, ncol: int = 1) -> "DeviceMatrix":
        t = self.torch.empty((int(ncol), int(nrow)), dtype=self.torch.float64, device=self.device)
        return DeviceMatrix(self, t)

    def zeros(self, nrow: int, ncol: int = 1) -> "DeviceMatrix":
        t = self.torch.zeros((int(ncol), int(nrow)), dtype=self.torch.float64, device=self.device)
        return DeviceMatrix(self, t)

    def from_numpy(self, a: np.ndarray) -> "DeviceMatrix":
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            a = a[:, None]
        # (ncol, nrow) C-contiguous == (nrow, ncol) column-major
        t = self.torch.from_numpy(np.ascontiguousarray(a.T)).to(self.device)
        return DeviceMatrix(self, t)

    # ---- HIP-event timing on the context's stream ----------------------------
    def event(self):
        e = C.c_void_p()
        _lib.call("bigkrls_event_create", C.byref(e))
        self._events.append(e)
        _lib.call("bigkrls_event_record", self.handle, e)
        return e

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
        return self.t.cpu().numpy().T.copy()

    def __getitem__(self, idx):  # R's `K[]` idiom: materialise on the host
        return self.to_numpy()[idx]

    def copy(self) -> "DeviceMatrix":
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
