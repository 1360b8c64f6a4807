"""The R-side boundary artefacts (r-shim/) against the C ABI they bind (include/bigkrls.h).

R and Rcpp are absent from the image, so the shim cannot be built into the R package here. What is checked: the shim
is TYPE-CHECKED by g++ (-fsyntax-only -Wall -Werror) against include/bigkrls.h and a minimal mock of the Rcpp /
bigmemory declarations it uses (tests/rshim_mock/), and it stays in step with the header: every bigkrls_* call in r-shim/src/bigkrls_shim.cpp names a declared function
and passes as many arguments as its prototype takes, the eleven .Call routines of the reference
(src/RcppExports.cpp:147-160) are all exported with the reference's arities, and every shim routine the R
host functions (r-shim/R/bigKRLS_gpu.R) call exists with that many parameters."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bigkrls.h")
SHIM = os.path.join(ROOT, "r-shim", "src", "bigkrls_shim.cpp")
RHOST = os.path.join(ROOT, "r-shim", "R", "bigKRLS_gpu.R")
RMETHODS = os.path.join(ROOT, "r-shim", "R", "bigKRLS_gpu_methods.R")
MOCK = os.path.join(ROOT, "tests", "rshim_mock")

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
    rsrc = _strip_comments((open(RHOST).read() + "\n" + open(RMETHODS).read()).replace("//", "  "), hash_comments=True)
    level2 = [n for n in exports if n not in REFERENCE_CALLS]
    assert {"DevContext", "DevMatrix", "DevToHost", "HostToDev", "BigKRLSFit", "BigKRLSPredict"} <= set(level2)
    seen = set()
    for name, args in _calls(rsrc, r"\b(" + "|".join(level2) + r")"):
        assert len(args) == exports[name], f"{name}: R passes {len(args)} arguments, the shim takes {exports[name]}"
        seen.add(name)
    assert {"DevContext", "DevMatrix", "DevToHost", "BigKRLSFit", "BigKRLSPredict", "NeffectiveHost"} <= seen
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


def _syntax_check(path, extra=()):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
           "-I" + os.path.join(ROOT, "include"), "-I" + MOCK, *extra, path]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300)


def test_shim_type_checks_against_the_header_and_the_rcpp_mock(tmp_path):
    r = _syntax_check(SHIM)
    assert r.returncode == 0, r.stderr[-4000:]
    # the check has teeth: a call with a wrong argument type, and `delete` of an opaque handle, are both rejected
    bad = tmp_path / "bad.cpp"
    bad.write_text('#include <Rcpp.h>\n#include "bigkrls.h"\n'
                   "void f(SEXP c, Rcpp::NumericMatrix X) { bigkrls_ctx* p = nullptr; bigkrls_ctx_create(X, &p); }\n")
    assert _syntax_check(str(bad)).returncode != 0
    bad.write_text('#include <Rcpp.h>\n#include "bigkrls.h"\nvoid f(bigkrls_ctx* p) { delete p; }\n')
    assert _syntax_check(str(bad)).returncode != 0
    # no XPtr over an opaque ABI type (its default finaliser would delete an incomplete type), and every handle the
    # shim creates is released by the ABI's own function through a registered finaliser
    src = _strip_comments(open(SHIM).read())
    assert not re.search(r"XPtr<\s*bigkrls_", src)
    for fin, release in (("ctx_finalizer", "bigkrls_ctx_destroy"), ("comm_finalizer", "bigkrls_comm_destroy"),
                         ("dev_finalizer", "bigkrls_dev_free")):
        body = re.search(r"static void " + fin + r"\(SEXP s\) \{(.*?)\n\}", src, flags=re.S)
        assert body and release in body.group(1), fin
        assert re.search(r"R_RegisterCFinalizerEx\(s, " + fin, src), fin


def test_r_methods_keep_the_reference_signatures_and_fields():
    """summary.bigKRLS (R/bigKRLS.R:706-707), crossvalidate.bigKRLS (:1146), summary.bigKRLS_CV (:760) and what
    predict.bigKRLS returns (:628-631)."""
    msrc = open(RMETHODS).read()

    def signature(name):
        m = re.search(re.escape(name) + r" <- function\((.*?)\)\s*\{", msrc, flags=re.S)
        assert m, name
        return [a.split("=")[0].strip() for a in _split_args(m.group(1))]

    assert signature("summary.bigKRLS") == ["object", "degrees", "probs", "digits", "labs", "..."]
    assert signature("crossvalidate.bigKRLS")[:6] == ["y", "X", "seed", "Kfolds", "ptesting", "estimates_subfolder"]
    assert signature("crossvalidate.bigKRLS")[-1] == "..."
    assert signature("summary.bigKRLS_CV") == ["object", "..."]
    for field in ("R2_is", "R2_oos", "MSE_is", "MSE_oos", "R2AME_is", "R2AME_oos", "MSE_AME_is", "MSE_AME_oos",
                  "pseudoR2_is", "pseudoR2_oos", "pseudoR2AME_is", "pseudoR2AME_oos", "ptesting", "indices", "folds",
                  "Kfolds", "seed", "ttests", "percentiles"):
        assert field in msrc, field
    hsrc = open(RHOST).read()
    pred = hsrc[hsrc.index("predict.bigKRLS <- function"):hsrc.index("# ---- save / load")]
    for field in ("predicted", "se.pred", "vcov.est.pred", "newdata", "newdataK", "has.big.matrices", "ytest"):
        assert re.search(r"\b" + re.escape(field) + r"\s*=", pred), field
    assert "vcov.est.pred = NULL" not in pred and "newdataK = NULL" not in pred
    assert 'version = 2' in hsrc                      # files every R reads, and the format rdata.py writes
    # brackets of the methods file balance (R cannot parse it here)
    stack = []
    for ch in re.sub(r"\"[^\"\n]*\"|'[^'\n]*'", "", _strip_comments(msrc.replace("//", "  "), hash_comments=True)):
        if ch in "([{":
            stack.append(ch)
        elif ch in ")]}":
            assert stack and "([{".index(stack.pop()) == ")]}".index(ch)
    assert not stack
