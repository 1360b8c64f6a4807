"""The R-side boundary artefacts (r-shim/) against the C ABI they bind (include/bigkrls.h).

R and Rcpp are absent from the image, so the shim cannot be compiled here; what can be checked is that it
stays in step with the header: every bigkrls_* call in r-shim/src/bigkrls_shim.cpp names a declared function
and passes as many arguments as its prototype takes, the eleven .Call routines of the reference
(src/RcppExports.cpp:147-160) are all exported with the reference's arities, and every shim routine the R
host functions (r-shim/R/bigKRLS_gpu.R) call exists with that many parameters."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bigkrls.h")
SHIM = os.path.join(ROOT, "r-shim", "src", "bigkrls_shim.cpp")
RHOST = os.path.join(ROOT, "r-shim", "R", "bigKRLS_gpu.R")

# .Call routines registered by the reference and their arities (src/RcppExports.cpp:147-160)
REFERENCE_CALLS = {"BigNeffective": 1, "BigDerivMat": 7, "BigCrossProd": 3, "BigXtX": 2, "BigTCrossProd": 3,
                   "BigXXt": 2, "BigEigen": 4, "BigGaussKernel": 3, "BigMultDiag": 3, "BigSolveForc": 4,
                   "BigTempKernel": 4}


def _strip_comments(src: str, hash_comments: bool = False) -> str:
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    if hash_comments:
        src = re.sub(r"#[^\n]*", " ", src)
    return src


def _split_args(text: str):
    """top-level comma split of the text between a call's parentheses"""
    args, depth, cur, in_str = [], 0, "", None
    for ch in text:
        if in_str:
            cur += ch
            if ch == in_str:
                in_str = None
            continue
        if ch in "\"'":
            in_str = ch
            cur += ch
        elif ch in "([{":
            depth += 1
            cur += ch
        elif ch in ")]}":
            depth -= 1
            cur += ch
        elif ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def _calls(src: str, pattern: str):
    """(name, [args]) for every `name(` matching pattern, with balanced parentheses"""
    out = []
    for m in re.finditer(pattern + r"\s*\(", src):
        name = m.group(1)
        i, depth = m.end(), 1
        while depth and i < len(src):
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        out.append((name, _split_args(src[m.end():i - 1])))
    return out


def _header_prototypes():
    src = _strip_comments(open(HEADER).read())
    protos = {}
    for name, args in _calls(src, r"\b(bigkrls_\w+)"):
        if name in ("bigkrls_fit_options", "bigkrls_fit_outputs", "bigkrls_ctx"):
            continue
        protos[name] = 0 if args == ["void"] else len(args)
    return protos


def _shim_exports():
    src = open(SHIM).read()
    exports = {}
    for m in re.finditer(r"//\s*\[\[Rcpp::export\]\]\s*\n([^\n(]*?)\b(\w+)\s*\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        exports[m.group(2)] = len(_split_args(src[m.end():i - 1]))
    return exports


def test_every_abi_call_in_the_shim_matches_a_declared_prototype():
    protos = _header_prototypes()
    assert len(protos) > 50 and protos["bigkrls_gauss_kernel"] == 5 and protos["bigkrls_fit"] == 7
    src = _strip_comments(open(SHIM).read())
    calls = [(n, a) for n, a in _calls(src, r"\b(bigkrls_\w+)") if n not in ("bigkrls_fit_options", "bigkrls_fit_outputs")]
    assert len(calls) >= 16
    for name, args in calls:
        assert name in protos, f"{name} is not declared in include/bigkrls.h"
        assert len(args) == protos[name], f"{name}: shim passes {len(args)} arguments, header declares {protos[name]}"
    # the structs are filled member by member in the header's order: the positional initialiser of the options
    # must have as many fields as the struct
    hdr = _strip_comments(open(HEADER).read())
    body = re.search(r"typedef struct bigkrls_fit_options \{(.*?)\} bigkrls_fit_options;", hdr, flags=re.S).group(1)
    n_fields = sum(f.count(",") + 1 for f in body.split(";") if f.strip())    # `double L, U;` declares two
    init = re.search(r"bigkrls_fit_options o = \{(.*?)\};", src, flags=re.S).group(1)
    assert len(_split_args(init)) == n_fields == 13
    for member in re.findall(r"\br\.(\w+)\s*=", src):
        assert re.search(r"\b" + member + r"\b", hdr), f"bigkrls_fit_outputs has no member {member}"


def test_the_reference_call_routines_are_all_exported_with_their_arities():
    exports = _shim_exports()
    for name, arity in REFERENCE_CALLS.items():
        assert exports.get(name) == arity, f"{name}: exported with {exports.get(name)} parameters, reference has {arity}"


def test_r_host_functions_call_existing_shim_routines():
    exports = _shim_exports()
    rsrc = _strip_comments(open(RHOST).read().replace("//", "  "), hash_comments=True)
    level2 = [n for n in exports if n not in REFERENCE_CALLS]
    assert {"DevContext", "DevMatrix", "DevToHost", "HostToDev", "BigKRLSFit", "BigKRLSPredict"} <= set(level2)
    seen = set()
    for name, args in _calls(rsrc, r"\b(" + "|".join(level2) + r")"):
        assert len(args) == exports[name], f"{name}: R passes {len(args)} arguments, the shim takes {exports[name]}"
        seen.add(name)
    assert {"DevContext", "DevMatrix", "DevToHost", "BigKRLSFit", "BigKRLSPredict"} <= seen
    # balanced brackets: the file cannot be parsed by R here, at least it is not truncated or mis-nested
    stack = []
    for ch in re.sub(r"\"[^\"\n]*\"|'[^'\n]*'", "", rsrc):
        if ch in "([{":
            stack.append(ch)
        elif ch in ")]}":
            assert stack and "([{".index(stack.pop()) == ")]}".index(ch)
    assert not stack
    # the reference's argument list of bigKRLS() (R/bigKRLS.R:97-103), in order
    sig = re.search(r"bigKRLS <- function\((.*?)\)\s*\{", open(RHOST).read(), flags=re.S).group(1)
    names = [a.split("=")[0].strip() for a in _split_args(sig)]
    assert names[:18] == ["y", "X", "sigma", "derivative", "which.derivatives", "vcov.est", "Neig", "eigtrunc",
                          "lambda", "L", "U", "tol", "model_subfolder_name", "overwrite.existing", "Ncores", "acf",
                          "noisy", "instructions"]
