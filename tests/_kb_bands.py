"""The XCD-aware band-major tile order of the one-wave-per-tile kernel builds (csrc/gemm.hip, band_major_lower /
band_major_rect) with SEVERAL bands: the band height is read once per process (BIGKRLS_KB_R, default ~1 MB of X rows =
thousands of rows), so the multi-band maps -- full bands, a partial last band, the triangular corner of every band, a
band height that does not divide the tile count -- are driven here in a process of their own with a height of a few
tile rows, every entry of the result against the literal kernel formula. Run by tests/test_gpu_level1.py."""
import os
import sys

os.environ["BIGKRLS_KB_R"] = sys.argv[1] if len(sys.argv) > 1 else "8"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ctypes as C

import numpy as np

from bigkrls_amd import _lib
from oracle import krls_oracle as orc

lib = _lib.load()
F = np.asfortranarray
P = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(12)
for n, p in ((1000, 20), (1337, 5), (32 * 17, 33), (257, 3), (2048, 12)):
    X = rng.standard_normal((n, p))
    Xf, out = F(X), F(np.full((n, n), np.nan))
    assert lib.bigkrls_gauss_kernel(P(Xf), n, p, float(p), P(out)) == 0, lib.bigkrls_last_error()
    ref = orc.gauss_kernel_literal(X, float(p))
    assert np.isfinite(out).all(), (n, p, "a tile was never written")
    assert np.max(np.abs(out - ref)) < 1e-13, (n, p)
    assert np.array_equal(out, out.T)
for u, v, p in ((700, 1100, 7), (1025, 300, 20), (33, 2000, 4), (960, 960, 31)):
    A, B = rng.standard_normal((u, p)), rng.standard_normal((v, p))
    Af, Bf, out = F(A), F(B), F(np.full((u, v), np.nan))
    assert lib.bigkrls_temp_kernel(P(Af), u, P(Bf), v, p, 2.5, P(out)) == 0, lib.bigkrls_last_error()
    assert np.isfinite(out).all(), (u, v, p, "a tile was never written")
    assert np.max(np.abs(out - orc.temp_kernel_literal(A, B, 2.5))) < 1e-13, (u, v, p)
print("band-major tile maps OK")
