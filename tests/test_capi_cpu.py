"""CPU-side checks of the drop-in boundary: the shared library loads, exports every
symbol include/bigkrls.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "bigkrls.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bigkrls_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_every_reference_entry_point():
    names = header_functions()
    # one Level-1 drop-in per .Call routine of src/RcppExports.cpp:147-160 (BigNeffective is out of scope)
    for n in ["gauss_kernel", "temp_kernel", "eigen", "solveforc", "multdiag", "crossprod", "xtx",
              "tcrossprod", "xxt", "derivmat"]:
        assert f"bigkrls_{n}" in names


def test_library_exports_every_declared_symbol():
    from bigkrls_amd import _lib
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/bigkrls.h but not exported"
    # and the ctypes table covers the header exactly
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_lambda_bounds_host_helper_matches_r_loops():
    """bigkrls_lambda_bounds is host-only: the U/L loops of bLambdaSearch (:16-36)."""
    from bigkrls_amd import ops
    from oracle import krls_oracle as orc
    rng = np.random.default_rng(0)
    for n in (50, 400, 3000):
        d = np.sort(rng.gamma(0.3, 2.0, size=n))[::-1] * n / 10
        d[-3:] = [1e-9, 1e-12, -1e-15]
        L, U = ops.lambda_bounds(d, n)
        L0, U0 = orc.lambda_bounds(d, n)
        assert (L, U) == (L0, U0)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bigkrls_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    st = lib.bigkrls_ctx_create(0, C.byref(h))
    assert st == _lib.ENODEVICE
    assert b"no CPU fallback" in lib.bigkrls_last_error() or b"device" in lib.bigkrls_last_error()
    X = np.asfortranarray(np.ones((4, 2)))
    out = np.asfortranarray(np.zeros((4, 4)))
    st = lib.bigkrls_gauss_kernel(C.c_void_p(X.ctypes.data), 4, 2, 1.0, C.c_void_p(out.ctypes.data))
    assert st != 0 and not out.any()
    import bigkrls_amd as bk
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bk.Context(0)
    with pytest.raises((RuntimeError, _lib.BigKRLSError)):
        bk.bigKRLS(np.arange(10.0), np.random.default_rng(0).standard_normal((10, 2)))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bigkrls_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "krls_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_partition_covers_rows_once():
    from bigkrls_amd.dist import partition
    for n in (1, 7, 8, 9, 20000, 50001):
        for world in (1, 2, 3, 8):
            nb, parts = partition(n, world)
            assert parts[0][0] == 0 and parts[-1][1] == n and nb * world >= n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert all(b - a <= nb for a, b in parts)


def test_summary_student_t_tail_matches_scipy():
    """Host logic of summary(): pt(|t|, df, lower.tail=FALSE) (R/bigKRLS.R:726) is restated with an
    incomplete-beta continued fraction; check it against scipy's Student t survival function."""
    from scipy import stats
    from bigkrls_amd import api
    for df in (0.7, 3.0, 17.5, 396.2, 19752.1, 1e6):
        for t in (0.0, 1e-3, 0.5, 1.96, 5.0, 12.0, 40.0):
            a, b = api._pt_upper(t, df), stats.t.sf(t, df)
            assert abs(a - b) <= 1e-7 * b + 1e-300, (t, df, a, b)


def test_column_major_filler_matches_fortran_flattening():
    """Host side of Context.from_numpy: pieces of the column-major flattening of a 2-D array of any
    layout (C, Fortran, strided view), as the pinned staging buffer receives them."""
    from bigkrls_amd.device import _column_major_filler
    rng = np.random.default_rng(0)
    for shape in [(7, 5), (1, 9), (9, 1), (64, 3), (5, 0)]:
        for order in "CF":
            a = np.asarray(rng.random(shape), order=order)
            ref = a.reshape(-1, order="F")
            fill = _column_major_filler(a)
            for _ in range(40):
                if ref.size == 0:
                    break
                off = int(rng.integers(0, ref.size))
                m = int(rng.integers(1, ref.size - off + 1))
                dst = np.full(m, -1.0)
                fill(dst, off, m)
                assert np.array_equal(dst, ref[off:off + m]), (shape, order, off, m)
    view = rng.random((12, 10))[::2, 1:7]
    dst = np.empty(view.size)
    _column_major_filler(view)(dst, 0, view.size)
    assert np.array_equal(dst, view.reshape(-1, order="F"))


def test_fit_struct_layout_matches_the_header(tmp_path):
    """bigkrls_fit_options / bigkrls_fit_outputs as ctypes sees them vs as a C compiler lays them
    out from include/bigkrls.h (sizes and the offsets of the first scalar output and phase_s)."""
    import subprocess
    from bigkrls_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bigkrls.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(bigkrls_fit_options), '
                   'sizeof(bigkrls_fit_outputs), offsetof(bigkrls_fit_outputs, lastkeeper), '
                   'offsetof(bigkrls_fit_outputs, phase_s), offsetof(bigkrls_fit_options, which_derivatives));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(t) for t in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(_lib.FitOptions), C.sizeof(_lib.FitOutputs), _lib.FitOutputs.lastkeeper.offset,
            _lib.FitOutputs.phase_s.offset, _lib.FitOptions.which_derivatives.offset]
    assert got == want


def test_plain_c_caller_builds_and_fails_loudly_without_gpu():
    """tests/capi/fit_example.c, a C99 caller of bigkrls_fit / bigkrls_predict compiled with gcc against
    include/bigkrls.h (what an R shim's body does): it must build, link and -- without a GPU --
    return BIGKRLS_ENODEVICE instead of computing anything."""
    import subprocess
    import torch
    from bigkrls_amd import _lib
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "capi")], check=True, capture_output=True)
    _lib.load()
    ex = C.CDLL(os.path.join(ROOT, "tests", "capi", "libfit_example.so"))
    ex.capi_fit_example.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    if torch.cuda.is_available():
        pytest.skip("GPU present: exercised by tests/test_gpu_fit_capi.py")
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((40, 3)))
    y = np.ascontiguousarray(X[:, 0] + 0.1)
    coeffs, res = np.zeros(40), np.zeros(8)
    st = ex.capi_fit_example(X.ctypes.data, y.ctypes.data, 40, 3, 5, coeffs.ctypes.data, res.ctypes.data)
    assert st == _lib.ENODEVICE and not coeffs.any() and not res.any()


def test_missing_rccl_is_an_error_code_not_a_crash():
    """include/bigkrls.h: without librccl, bigkrls_comm_unique_id / bigkrls_comm_create return BIGKRLS_ENODEVICE with
    a message (the not-found branch used to call dlerror() twice and pass its second, NULL, result to std::string).
    A fresh process: the loader's choice is made once per process."""
    code = (
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from bigkrls_amd import _lib\n"
        "lib = _lib.load()\n"
        "uid = C.create_string_buffer(128)\n"
        "st = lib.bigkrls_comm_unique_id(uid)\n"
        "msg = lib.bigkrls_last_error().decode()\n"
        "assert st == _lib.ENODEVICE, st\n"
        "assert 'librccl not found' in msg and 'no_such_rccl' in msg, msg\n"
        "print('OK')\n")
    env = dict(os.environ, BIGKRLS_RCCL_LIB="/nonexistent/no_such_rccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
