"""save.bigKRLS / load.bigKRLS (R/bigKRLS.R:901-945, 960-1020; bSave / bLoad,
R/bigKRLS_Rcpp_functions.R:300-379).

The reference writes every big.matrix member of the object as `<member>.txt` (bigmemory's
`write.big.matrix`: comma separated, one matrix row per line, no header) and everything else
into `estimates.RData`; `load.bigKRLS` reads `estimates.RData` and then the matrices named in
bLoad (:336-343) that exist as text files. Here:

  * device-resident matrices (the counterparts of big.matrix members: K, vcov.est.c,
    vcov.est.fitted, and X / derivatives when they were returned on the device) go to
    `<member>.txt` in exactly that text layout (17 significant digits), or -- `binary=True` --
    to `<member>.npy` (column-major float64), because an N x N text dump is what dominates the
    reference's save time at large N;
  * the base-R part goes to `estimates.npz` (+ a small `estimates.json` for scalars, strings and
    None). R's serialisation format (`estimates.RData`) cannot be produced or checked in this
    image (no R); a maintainer of the R package keeps `save()` for that part.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import numpy as np

from .api import BigKRLS, BigKRLSPredicted, default_context
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
        return np.atleast_2d(np.loadtxt(path, delimiter=",", dtype=np.float64))
    except ValueError:      # NA tokens: the slower tolerant reader
        return np.atleast_2d(np.genfromtxt(path, delimiter=",", dtype=np.float64, missing_values="NA",
                                           filling_values=np.nan))

_BIGKRLS_MATRICES = ["K", "X", "derivatives", "vcov.est.c", "vcov.est.fitted"]             # bLoad :337
_PREDICTED_MATRICES = ["predicted", "se.pred", "vcov.est.pred", "newdata", "newdataK", "ytest"]   # :340


def _make_path(folder: str, overwrite_existing: bool) -> str:
    """make_path (R/bigKRLS_Rcpp_functions.R:272-297): never silently reuse a folder."""
    if os.path.exists(folder) and not overwrite_existing:
        i = 1
        base = folder
        while os.path.exists(folder):
            folder = f"{base}{i}"
            i += 1
        print(f"a folder named {base} exists; output will be saved to {folder} instead "
              "(pass overwrite_existing=True to reuse it)")
    os.makedirs(folder, exist_ok=True)
    return folder


def save_bigKRLS(object, model_subfolder_name: str, overwrite_existing: bool = False, noisy: bool = True,
                 binary: bool = False, digits: int = 17) -> str:
    if not isinstance(object, (BigKRLS, BigKRLSPredicted)):
        raise TypeError("Object not a bigKRLS class.")
    if not isinstance(model_subfolder_name, str):
        raise TypeError("model_subfolder_name must be a character string")
    folder = _make_path(model_subfolder_name, overwrite_existing)
    arrays, meta, nbm = {}, {"r_class": object.r_class, "none": [], "scalars": {}, "strings": {}}, 0
    for name, val in object.items():
        if name == "_ctx":
            continue
        if is_device_matrix(val):                                   # bSave :302-309
            path = os.path.join(folder, name + (".npy" if binary else ".txt"))
            if noisy:
                print("\twriting", path, "...")
            host = val.to_numpy()
            if binary:
                np.save(path, np.asfortranarray(host))
            else:
                write_big_matrix_text(host, path, digits=digits)
            nbm += 1
        elif val is None:
            meta["none"].append(name)
        elif isinstance(val, str):
            meta["strings"][name] = val
        elif isinstance(val, (bool, int, float, np.floating, np.integer, np.bool_)):
            meta["scalars"][name] = val.item() if hasattr(val, "item") else val
        else:
            arrays[name] = np.asarray(val)
    meta["model_subfolder_name"] = folder
    np.savez_compressed(os.path.join(folder, "estimates.npz"), **arrays)
    with open(os.path.join(folder, "estimates.json"), "w") as f:
        json.dump(meta, f)
    if noisy:
        print(f"\n{nbm} matrices saved as big matrices" +
              (" (nothing device-resident in this object).\n" if nbm == 0 else
               ", use load_bigKRLS() on the entire directory to reconstruct the outputted object.\n"))
    return folder


def load_bigKRLS(path: str, noisy: bool = True, ctx: Optional[Context] = None, to_device: bool = True):
    files = os.listdir(path)
    if "estimates.npz" not in files or "estimates.json" not in files:
        raise FileNotFoundError(
            "estimates.npz / estimates.json not found. Check the path to the output folder.\n\n"
            "Note: load_bigKRLS() anticipates the convention used by save_bigKRLS: the base objects in "
            "estimates.npz + estimates.json, big matrices stored as text files named like they are in bigKRLS "
            "objects (object$K becomes K.txt, etc.).")
    with open(os.path.join(path, "estimates.json")) as f:
        meta = json.load(f)
    obj = BigKRLS() if meta["r_class"] == "bigKRLS" else BigKRLSPredicted()
    with np.load(os.path.join(path, "estimates.npz"), allow_pickle=False) as z:
        for k in z.files:
            obj[k] = z[k]
    for k in meta["none"]:
        obj[k] = None
    obj.update(meta["strings"])
    obj.update(meta["scalars"])
    if isinstance(obj.get("which.derivatives"), np.ndarray):
        obj["which.derivatives"] = [int(i) for i in obj["which.derivatives"]]
    if isinstance(obj.get("xlabs"), np.ndarray):
        obj["xlabs"] = [str(s) for s in obj["xlabs"]]
    matrices = _BIGKRLS_MATRICES if meta["r_class"] == "bigKRLS" else _PREDICTED_MATRICES   # bLoad :336-343
    for name in matrices:
        txt, npy = name + ".txt", name + ".npy"
        if txt not in files and npy not in files:
            if name not in obj and noisy:
                print("NOTE:", name, "not found in estimates.npz or in big matrix file,", txt, ".\n")
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
