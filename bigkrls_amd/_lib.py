"""ctypes binding of libbigkrls_hip.so (the C ABI declared in include/bigkrls.h).

There is no fallback: if the shared library is missing or a call fails, an
exception is raised.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbigkrls_hip.so")

OK, EINVAL, ENODEVICE, EHIP, ENOMEM, ENOCONV = 0, 1, 2, 3, 4, 5
_CODES = {1: "EINVAL", 2: "ENODEVICE", 3: "EHIP", 4: "ENOMEM", 5: "ENOCONV"}


class BigKRLSError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libbigkrls_hip: {_CODES.get(code, code)}: {msg}")
        self.code = code


_lib: Optional[C.CDLL] = None

i64 = C.c_int64
f64 = C.c_double
vp = C.c_void_p
i32 = C.c_int32
pf64 = C.POINTER(C.c_double)
pi64 = C.POINTER(C.c_int64)
pi32 = C.POINTER(C.c_int32)

class FitOptions(C.Structure):
    """bigkrls_fit_options (include/bigkrls.h)."""
    _fields_ = [("struct_bytes", i64), ("sigma", f64), ("lambda_", f64), ("L", f64), ("U", f64),
                ("eigtrunc", f64), ("neig", i64), ("derivative", i32), ("vcov_est", i32), ("acf", i32),
                ("reserved", i32), ("which_derivatives", pi64), ("n_which", i64)]


class FitOutputs(C.Structure):
    """bigkrls_fit_outputs (include/bigkrls.h)."""
    _fields_ = [("struct_bytes", i64),
                ("eigenvalues", vp), ("coeffs", vp), ("yfitted", vp), ("yfitted_std", vp),
                ("derivatives", vp), ("derivatives_std", vp), ("avgderivatives", vp),
                ("var_avgderivatives", vp), ("var_avgderivatives_std", vp), ("binaryindicator", vp),
                ("lambda_trace", vp), ("max_trace", i64),
                ("d_K", vp), ("d_vcov_c", vp), ("d_vcov_fitted", vp),
                ("lastkeeper", i64), ("neig", i64), ("n_deriv", i64), ("n_probes", i64),
                ("sigma", f64), ("lambda_", f64), ("Le", f64), ("Looe", f64), ("sigmasq", f64),
                ("R2", f64), ("R2AME", f64), ("Neffective", f64), ("Neffective_acf", f64),
                ("y_mean", f64), ("y_sd", f64), ("phase_s", f64 * 8)]


PHASES = ("h2d", "kernel", "eigen", "lambda", "coeffs", "vcov_c", "vcov_fitted", "derivatives")

# name -> argtypes (every function returns int unless listed in _RESTYPES)
SIGNATURES = {
    "bigkrls_version": [],
    "bigkrls_last_error": [],
    "bigkrls_device_count": [C.POINTER(C.c_int)],
    "bigkrls_ctx_create": [C.c_int, C.POINTER(vp)],
    "bigkrls_ctx_create_on_stream": [C.c_int, vp, C.POINTER(vp)],
    "bigkrls_ctx_destroy": [vp],
    "bigkrls_ctx_sync": [vp],
    "bigkrls_ctx_stream": [vp],
    "bigkrls_ctx_workspace_bytes": [vp],
    "bigkrls_ctx_release_workspace": [vp],
    "bigkrls_ctx_set_profile": [vp, C.c_int],
    "bigkrls_ctx_get_profile": [vp, C.c_char_p, pf64, pf64, pi64],
    "bigkrls_ctx_get_counters": [vp, pi64],
    "bigkrls_dev_alloc": [vp, i64, C.POINTER(vp)],
    "bigkrls_dev_free": [vp, vp],
    "bigkrls_h2d": [vp, vp, vp, i64],
    "bigkrls_d2h": [vp, vp, vp, i64],
    "bigkrls_d2d": [vp, vp, vp, i64],
    "bigkrls_event_create": [C.POINTER(vp)],
    "bigkrls_event_destroy": [vp],
    "bigkrls_event_record": [vp, vp],
    "bigkrls_event_elapsed_ms": [vp, vp, pf64],
    # level 1
    "bigkrls_gauss_kernel": [vp, i64, i64, f64, vp],
    "bigkrls_temp_kernel": [vp, i64, vp, i64, i64, f64, vp],
    "bigkrls_eigen": [vp, i64, i64, vp, vp],
    "bigkrls_solveforc": [vp, i64, i64, vp, i64, vp, f64, pf64, vp],
    "bigkrls_multdiag": [vp, i64, i64, vp, vp],
    "bigkrls_crossprod": [vp, i64, i64, vp, i64, vp],
    "bigkrls_xtx": [vp, i64, i64, vp],
    "bigkrls_tcrossprod": [vp, i64, i64, vp, i64, vp],
    "bigkrls_xxt": [vp, i64, i64, vp],
    "bigkrls_derivmat": [vp, i64, i64, vp, vp, vp, vp, vp, f64],
    "bigkrls_neffective": [vp, i64, i64, vp],
    # level 2
    "bigkrls_dev_kernel_block": [vp, vp, i64, i64, vp, i64, i64, i64, f64, vp, i64, i64],
    "bigkrls_dev_gemm": [vp, C.c_int, C.c_int, i64, i64, i64, f64, vp, i64, vp, i64, f64, vp, i64],
    "bigkrls_dev_multdiag": [vp, vp, i64, i64, i64, vp, vp, i64],
    "bigkrls_dev_eigen": [vp, vp, i64, i64, i64, vp, i64, f64, vp, i64, pi64],
    "bigkrls_dev_eigen_part": [vp, vp, i64, i64, i64, vp, i64, f64, vp, i64, pi64, i32, i32],
    "bigkrls_dev_fill_random": [vp, vp, i64, C.c_uint32],
    "bigkrls_dev_cholqr2": [vp, vp, vp, i64, i64, vp, pi32, vp],
    "bigkrls_dev_lanczos_projected": [vp, vp, vp, i64, i64, vp],
    "bigkrls_dev_copy_matrix": [vp, vp, i64, i64, i64, vp, i64],
    "bigkrls_dev_qty": [vp, vp, i64, i64, i64, vp, vp],
    "bigkrls_dev_solveforc": [vp, vp, i64, i64, i64, vp, vp, f64, vp, pf64],
    "bigkrls_dev_lambda_search": [vp, vp, i64, i64, i64, vp, vp, vp, i64, f64, f64, f64,
                                  pf64, pi64, vp, i64],
    "bigkrls_lambda_bounds": [vp, i64, i64, pf64, pf64],
    "bigkrls_dev_deriv_rows": [vp, vp, i64, i64, i64, i64, vp, i64, i64, vp, vp, f64, vp, i64,
                               vp, i64],
    "bigkrls_dev_deriv_var": [vp, vp, i64, i64, i64, vp, vp, i64, i64, vp, vp],
    "bigkrls_dev_gemv": [vp, C.c_int, i64, i64, f64, vp, i64, vp, f64, vp],
    "bigkrls_dev_dot": [vp, i64, vp, vp, pf64],
    "bigkrls_dev_diag": [vp, vp, i64, i64, vp],
    "bigkrls_dev_scale": [vp, i64, f64, vp],
    "bigkrls_dev_neffective": [vp, vp, i64, i64, i64, vp],
    # level 2, whole path
    "bigkrls_fit": [vp, vp, vp, i64, i64, C.POINTER(FitOptions), C.POINTER(FitOutputs)],
    "bigkrls_predict": [vp, vp, i64, i64, vp, vp, f64, vp, i64, vp, f64, vp, vp, vp, vp],
    # multi-GPU
    "bigkrls_comm_unique_id": [vp],
    "bigkrls_comm_create": [vp, i32, i32, vp, C.POINTER(vp)],
    "bigkrls_comm_create_callbacks": [vp, i32, i32, vp, C.POINTER(vp)],
    "bigkrls_comm_destroy": [vp],
    "bigkrls_comm_forget_context": [vp],
    "bigkrls_comm_rank": [vp, pi32, pi32],
    "bigkrls_comm_check": [vp, vp, i64],
    "bigkrls_fit_dist_rows": [vp, i64, C.POINTER(FitOptions), pi64, pi64],
    "bigkrls_fit_dist": [vp, vp, vp, i64, i64, C.POINTER(FitOptions), C.POINTER(FitOutputs)],
}

ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, i64, i32)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp, i64)
BROADCAST_FN = C.CFUNCTYPE(C.c_int, vp, vp, i64, i32)


class Collectives(C.Structure):
    """bigkrls_collectives (include/bigkrls.h)."""
    _fields_ = [("struct_bytes", i64), ("user", vp), ("all_reduce", ALL_REDUCE_FN), ("all_gather", ALL_GATHER_FN),
                ("broadcast", BROADCAST_FN)]
_RESTYPES = {
    "bigkrls_last_error": C.c_char_p,
    "bigkrls_ctx_stream": vp,
    "bigkrls_ctx_workspace_bytes": i64,
}


def load() -> C.CDLL:
    """Load the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C bigkrls_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as the system one). If this
    # library pulled in the system runtime first, a later `import torch` would load a second
    # HIP runtime into the process and neither would see the other's streams or memory.
    # Importing torch first makes the dynamic loader bind us to the runtime torch uses.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != OK:
        lib = load()
        msg = lib.bigkrls_last_error()
        raise BigKRLSError(status, msg.decode() if msg else "")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args))
