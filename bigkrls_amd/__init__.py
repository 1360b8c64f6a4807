"""bigkrls_amd: MI355X-native (gfx950) Kernel-Regularised Least Squares.

Drop-in for the hot path of rdrr1990/bigKRLS: `bigKRLS()`, `predict()`,
`crossvalidate()`, `summary()` and the `b*` helper layer over hand-written HIP kernels reached
through the C ABI in include/bigkrls.h (bigkrls_amd/libbigkrls_hip.so).
Importing the package does not touch the GPU; the first call does, and fails
loudly if the shared library or a HIP device is missing.
"""
from ._lib import BigKRLSError, LIB_PATH  # noqa: F401
from .api import BigKRLS, BigKRLSPredicted, bigKRLS, crossvalidate, predict, summary  # noqa: F401
from .device import Context, DeviceMatrix  # noqa: F401
from .persist import load_bigKRLS, save_bigKRLS  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["bigKRLS", "predict", "crossvalidate", "summary", "save_bigKRLS", "load_bigKRLS", "Context", "DeviceMatrix", "ops",
           "BigKRLS", "BigKRLSPredicted", "BigKRLSError", "LIB_PATH"]
