"""save.bigKRLS / load.bigKRLS (R/bigKRLS.R:901-945, 960-1020; make_path / bSave / bLoad,
R/bigKRLS_Rcpp_functions.R:272-379).

The reference writes every big.matrix member of the object as `<member>.txt` (bigmemory's
`write.big.matrix`: comma separated, one matrix row per line, no header) and everything else
into `estimates.RData` as a list called `bigKRLS_out` (`object` for a cross-validation, whose folds go to
`fold_k/trained` and `fold_k/tested` sub-folders); `load.bigKRLS` loads `estimates.RData` and then the
matrices named in bLoad (:336-343) that exist as text files. Here:

  * device-resident matrices (the counterparts of big.matrix members: K, vcov.est.c,
    vcov.est.fitted, and X / derivatives when they were returned on the device) go to
    `<member>.txt` in exactly that text layout (17 significant digits), or -- `binary=True` --
    to `<member>.npy` (column-major float64), because an N x N text dump is what dominates the
    reference's save time at large N;
  * the base-R part goes to `estimates.RData` in R's own serialisation format (bigkrls_amd/rdata.py: XDR
    version 2, gzip, one pairlist entry named `bigKRLS_out` / `object` with its class attribute): version 2 is what
    every R since 2.3.0 loads, so a folder written here is one R's load.bigKRLS() can open. The other direction: the
    reader takes version-2 files (R < 3.5, `save(..., version = 2)` -- what the r-shim's save.bigKRLS writes) and
    the version-3 / `RDX3` files of a plain `save()` under R >= 3.5, with the ALTREP classes base R uses for plain
    data expanded (compact sequences, wrap_*, deferred strings; any other ALTREP class is refused by name). R is
    absent from the image: version 2 is pinned by the one R-written file of the reference tree
    (tests/test_rdata.py), version 3 only by hand-assembled streams, neither by an R session.
  * a fit that ran over several ranks (`comm=`) is saved by rank 0 only, without the N x N matrices: they stay
    sharded on the GPUs (`K.cols`, ...), and load.bigKRLS has no member to load them into.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from . import rdata
from .api import BigKRLS, BigKRLSCV, BigKRLSPredicted, default_context
from .device import Context, is_device_matrix


def write_big_matrix_text(host: np.ndarray, path: str, digits: int = 17) -> None:
    """The text layout of bigmemory's `write.big.matrix(x, filename)` as bSave calls it
    (R/bigKRLS_Rcpp_functions.R:306-307; defaults sep = ",", no row or column names): one matrix row
    per line, values joined by commas, every value printed the way a C++ ostream with
    `precision(16)` prints a double (printf's %.16g; bigmemory's `ttos`), NA for missing values,
    "\n" line ends. bigmemory is a third-party dependency absent from /root/reference (4.5.19 in the
    reference's recorded session, examples/numeric_convergence.md:68-95); the layout is pinned by the
    hand-written fixture tests/golden/write_big_matrix_3x4.txt. `digits=16` reproduces bigmemory's
    files byte for byte; the default 17 keeps the same layout but round-trips every double exactly
    (16 digits lose the last bit of some values, which is why the reference's own reload test only
    asks for max(K - K2) < 1e-6, tests/testthat/test_basic_usage.R:123-128)."""
    host = np.atleast_2d(np.asarray(host, dtype=np.float64))
    if not np.isnan(host).any():
        np.savetxt(path, host, delimiter=",", fmt=f"%.{int(digits)}g", newline="\n")
        return
    spec = f".{int(digits)}g"
    with open(path, "w", newline="\n") as f:
        for row in host:
            f.write(",".join("NA" if np.isnan(v) else format(v, spec) for v in row) + "\n")


def read_big_matrix_text(path: str) -> np.ndarray:
    """`read.big.matrix(path, type = "double")` as bLoad calls it (R/bigKRLS_Rcpp_functions.R:359-360):
    comma separated, no header, NA -> NaN."""
    try:
        return np.loadtxt(path, delimiter=",", dtype=np.float64, ndmin=2)
    except ValueError:      # NA tokens: the slower tolerant reader
        with open(path) as f:
            ncol = f.readline().count(",") + 1
        m = np.genfromtxt(path, delimiter=",", dtype=np.float64, missing_values="NA", filling_values=np.nan)
        return np.asarray(m).reshape(-1, ncol)     # a single column or a single row is still a matrix

_BIGKRLS_MATRICES = ["K", "X", "derivatives", "vcov.est.c", "vcov.est.fitted"]             # bLoad :337
_PREDICTED_MATRICES = ["predicted", "se.pred", "vcov.est.pred", "newdata", "newdataK", "ytest"]   # :340
_CLASSES = {"bigKRLS": BigKRLS, "bigKRLS_predicted": BigKRLSPredicted, "bigKRLS_CV": BigKRLSCV}


def _make_path(object, model_subfolder_name: str, overwrite_existing: bool, noisy: bool = True) -> str:
    """make_path (R/bigKRLS_Rcpp_functions.R:272-297): never silently reuse a folder; records `path` and
    `model_subfolder_name` in the object."""
    folder = model_subfolder_name
    if os.path.isdir(folder) and not overwrite_existing:
        i = 1
        while os.path.exists(f"{model_subfolder_name}{i}"):
            i += 1
        folder = f"{model_subfolder_name}{i}"
        print(f"A subfolder named {model_subfolder_name} exists. Your output will be saved to {folder} instead. "
              "To turn off this safeguard, set save_bigKRLS(..., overwrite_existing=True) next time.\n")
    os.makedirs(folder, exist_ok=True)
    if noisy:
        print("Saving model estimates to:\n\n", folder, "\n")
    object["path"] = os.path.abspath(folder)
    object["model_subfolder_name"] = folder
    return folder


def _r_value(v):
    """values R has no type for (None inside lists is fine: NULL)"""
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        return v.item()
    return v


def _bsave(object, noisy: bool, binary: bool, digits: int) -> None:
    """bSave (R/bigKRLS_Rcpp_functions.R:300-328)."""
    folder = object["model_subfolder_name"]
    base, nbm = type(object)(), 0
    for name, val in object.items():
        if name == "_ctx":                                          # the Python handle of the HIP context
            continue
        if is_device_matrix(val):                                   # :302-309
            path = os.path.join(folder, name + (".npy" if binary else ".txt"))
            if noisy:
                print("\twriting", path, "...")
            host = val.to_numpy()
            if binary:
                np.save(path, np.asfortranarray(host))
            else:
                write_big_matrix_text(host, path, digits=digits)
            nbm += 1
        else:
            base[name] = _r_value(val)
    if noisy:
        print(f"\n{nbm} matrices saved as big matrices" +
              (" (nothing device-resident in this object).\n" if nbm == 0 else
               ", use load_bigKRLS() on the entire directory to reconstruct the outputted object.\n"))
    rdata.save_rdata(os.path.join(folder, "estimates.RData"), {"bigKRLS_out": base})     # :322-323


def save_bigKRLS(object, model_subfolder_name: str, overwrite_existing: bool = False, noisy: bool = True,
                 binary: bool = False, digits: int = 17) -> str:
    if not isinstance(object, (BigKRLS, BigKRLSPredicted, BigKRLSCV)):
        raise TypeError("Object not a bigKRLS class.")
    if not isinstance(model_subfolder_name, str):
        raise TypeError("model_subfolder_name must be a character string")
    folder = _make_path(object, model_subfolder_name, overwrite_existing, noisy)
    if not isinstance(object, BigKRLSCV):
        _bsave(object, noisy, binary, digits)
        return folder
    # a cross-validation: every fold's trained / tested objects in their own sub-folders (R/bigKRLS.R:916-932)
    top = BigKRLSCV()
    for name, val in object.items():
        if name.startswith("fold_"):
            for kind in ("trained", "tested"):
                _make_path(val[kind], os.path.join(folder, name, kind), True, noisy)
                _bsave(val[kind], noisy, binary, digits)
        elif name in ("trained", "tested"):      # the ptesting branch holds its one split at the top level
            _make_path(val, os.path.join(folder, name), True, noisy)
            _bsave(val, noisy, binary, digits)
        elif name == "indices":
            top[name] = {k: np.asarray(v) + 1 for k, v in val.items()}      # R's row numbers are 1-based
        else:
            top[name] = _r_value(val)
    top["dir"] = sorted(os.path.relpath(os.path.join(dp, f), folder)
                        for dp, _, fs in os.walk(folder) for f in fs)
    rdata.save_rdata(os.path.join(folder, "estimates.RData"), {"object": top})
    return folder


def _from_r(tree):
    """to_python + the conventions of this package's objects: length-1 numeric / logical vectors are scalars,
    named lists with a class become that class."""
    v = rdata.to_python(tree)

    def conv(x, top=False):
        if isinstance(x, dict):
            cls = _CLASSES.get(x.pop("__class__", None), dict)
            out = cls()
            for k, e in x.items():
                out[k] = conv(e)
            return out
        if isinstance(x, np.ndarray) and x.ndim == 1 and x.size == 1:
            return x[0].item()
        if isinstance(x, list) and len(x) == 1 and isinstance(x[0], str):
            return x[0]
        return x
    return conv(v)


_VECTOR_FIELDS = {"xlabs", "which.derivatives", "coeffs", "y", "yfitted", "yfitted.std", "K.eigenvalues",
                  "binaryindicator", "predicted", "se.pred", "ytest", "devices", "dir", "folds",
                  "R2_is", "R2_oos", "MSE_is", "MSE_oos", "R2AME_is", "R2AME_oos", "MSE_AME_is", "MSE_AME_oos"}


def _bload(obj, path: str, noisy: bool, ctx: Optional[Context], to_device: bool):
    """bLoad (R/bigKRLS_Rcpp_functions.R:330-379)."""
    for k in list(obj.keys()):            # vectors of length 1 that are vectors by meaning
        if k in _VECTOR_FIELDS and obj[k] is not None and not isinstance(obj[k], (list, np.ndarray)):
            obj[k] = [obj[k]] if isinstance(obj[k], str) else np.atleast_1d(obj[k])
    if isinstance(obj.get("which.derivatives"), np.ndarray):
        obj["which.derivatives"] = [int(i) for i in obj["which.derivatives"]]
    files = os.listdir(path)
    matrices = _BIGKRLS_MATRICES if isinstance(obj, BigKRLS) else _PREDICTED_MATRICES        # :336-343
    for name in matrices:
        txt, npy = name + ".txt", name + ".npy"
        if txt not in files and npy not in files:
            if name not in obj and noisy:
                print("NOTE:", name, "not found in .RData or in big matrix file,", txt, ".\n")
            continue
        if noisy:
            print("\tReading from", npy if npy in files else txt)
        host = np.load(os.path.join(path, npy)) if npy in files else read_big_matrix_text(os.path.join(path, txt))
        if to_device:
            ctx = ctx or default_context()
            obj[name] = ctx.from_numpy(host)
        else:
            obj[name] = host
    if to_device and any(is_device_matrix(v) for v in obj.values()):
        obj["_ctx"] = ctx
    return obj


def load_bigKRLS(path: str, noisy: bool = True, ctx: Optional[Context] = None, to_device: bool = True):
    files = os.listdir(path)
    match = [f for f in files if f.lower() == "estimates.rdata"]                             # :968-975
    if not match:
        raise FileNotFoundError(
            "estimates.RData not found. Check the path to the output folder.\n\n"
            "Note: load_bigKRLS() anticipates the convention used by save_bigKRLS: estimates.RData stores the "
            "base R objects in a list called bigKRLS_out, big matrices stored as text files named like they are in "
            "bigKRLS objects (object$K becomes K.txt, etc.).")
    loaded = rdata.load_rdata(os.path.join(path, match[0]))
    if "bigKRLS_out" in loaded:
        return _bload(_from_r(loaded["bigKRLS_out"]), path, noisy, ctx, to_device)
    if "object" not in loaded:
        raise ValueError("estimates.RData holds neither `bigKRLS_out` nor `object`")
    obj = _from_r(loaded["object"])
    if not isinstance(obj, BigKRLSCV):
        raise ValueError("`object` in estimates.RData is not of class bigKRLS_CV")
    _bload_cv_fields(obj)
    for rel in sorted(d for d in obj.get("dir", []) if os.path.basename(d).lower() == "estimates.rdata"
                      and os.path.dirname(d)):
        parts = os.path.normpath(os.path.dirname(rel)).split(os.sep)
        sub = os.path.join(path, os.path.dirname(rel))
        part = rdata.load_rdata(os.path.join(sub, os.path.basename(rel)))["bigKRLS_out"]
        loaded_part = _bload(_from_r(part), sub, noisy, ctx, to_device)
        if len(parts) == 1:
            obj[parts[0]] = loaded_part
        else:
            obj.setdefault(parts[-2], {})[parts[-1]] = loaded_part
    if isinstance(obj.get("indices"), dict):
        obj["indices"] = {k: np.asarray(v) - 1 for k, v in obj["indices"].items() if k != "__class__"}
    return obj


def _bload_cv_fields(obj) -> None:
    for k in list(obj.keys()):
        if k in _VECTOR_FIELDS and obj[k] is not None and not isinstance(obj[k], (list, np.ndarray)):
            obj[k] = [obj[k]] if isinstance(obj[k], str) else np.atleast_1d(obj[k])
