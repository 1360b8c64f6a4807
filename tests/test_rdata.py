"""R's serialisation format as bigkrls_amd/rdata.py writes and reads it (estimates.RData of save.bigKRLS /
load.bigKRLS, R/bigKRLS.R:495, R/bigKRLS_Rcpp_functions.R:322-323, :349).

Pinned by the one R-written file in the reference tree: build/vignette.rds (saveRDS of a data.frame by
R 3.3.3), kept as data in tests/golden/r_serialize_v2_vignette_index.rds. The reader must parse it and the
writer must reproduce its stream byte for byte. Persistence of whole objects (no GPU: host-resident members
only) goes through the same code."""
import gzip
import os

import numpy as np
import pytest

from bigkrls_amd import rdata

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "r_serialize_v2_vignette_index.rds")


def test_reader_parses_the_r_written_file():
    o = rdata.unserialize(open(FIXTURE, "rb").read())
    assert o.kind == "list" and [k for k, _ in o.attrs] == ["names", "row.names", "class"]
    assert o.attr("class").values == ["data.frame"]
    assert o.attr("names").values == ["File", "Title", "PDF", "R", "Depends", "Keywords"]
    assert o.attr("row.names").values.tolist() == [rdata.NA_INT, -1]          # R's compact row names c(NA, -1)
    py = rdata.to_python(o)
    assert py["File"] == ["bigKRLS_basics.Rmd"] and py["PDF"] == ["bigKRLS_basics.html"]
    assert py["Depends"] == {"Depends": []}                                  # list(Depends = character(0))


def test_writer_reproduces_the_r_written_stream_byte_for_byte():
    raw = gzip.decompress(open(FIXTURE, "rb").read())
    assert rdata.serialize(rdata.unserialize(raw)) == raw
    # ... and from plain Python values
    df = rdata.RVec("list", [rdata.from_python(["bigKRLS_basics.Rmd"]), rdata.from_python(["bigKRLS_basics"]),
                             rdata.from_python(["bigKRLS_basics.html"]), rdata.from_python(["bigKRLS_basics.R"]),
                             rdata.RVec("list", [rdata.RVec("str", [])], [("names", rdata.RVec("str", ["Depends"]))]),
                             rdata.RVec("list", [rdata.RVec("str", [])], [("names", rdata.RVec("str", ["Keywords"]))])],
                    [("names", rdata.RVec("str", ["File", "Title", "PDF", "R", "Depends", "Keywords"])),
                     ("row.names", rdata.RVec("int", np.array([rdata.NA_INT, -1]))),
                     ("class", rdata.RVec("str", ["data.frame"]))])
    assert rdata.serialize(df) == raw


def test_rdata_round_trip_of_every_value_kind(tmp_path):
    rng = np.random.default_rng(3)
    obj = {"real": rng.standard_normal(5), "matrix": rng.standard_normal((4, 3)), "int": np.arange(4, dtype=np.int64),
           "lgl": np.array([True, False, True]), "scalar": 2.5, "count": 7, "flag": False, "text": "x1",
           "labels": ["a", "bé", None], "null": None, "nested": {"u": 1.0, "v": ["w"]},
           "tiny": np.array([5e-324, -0.0, np.inf, np.nan])}
    path = str(tmp_path / "e.RData")
    rdata.save_rdata(path, {"bigKRLS_out": obj})
    assert open(path, "rb").read(2) == b"\x1f\x8b"                            # save()'s default: gzip
    assert gzip.decompress(open(path, "rb").read())[:7] == b"RDX2\nX\n"
    back = rdata.to_python(rdata.load_rdata(path)["bigKRLS_out"])
    assert list(back.keys()) == list(obj.keys())
    assert np.array_equal(back["real"], obj["real"]) and np.array_equal(back["matrix"], obj["matrix"])
    assert back["int"].tolist() == [0, 1, 2, 3] and back["lgl"].tolist() == [True, False, True]
    assert back["scalar"].tolist() == [2.5] and back["count"].tolist() == [7] and back["flag"].tolist() == [False]
    assert back["text"] == ["x1"] and back["labels"] == ["a", "bé", None] and back["null"] is None
    assert back["nested"]["u"].tolist() == [1.0] and back["nested"]["v"] == ["w"]
    assert np.array_equal(back["tiny"], obj["tiny"], equal_nan=True) and np.signbit(back["tiny"][1])


def test_unsupported_streams_are_refused():
    with pytest.raises(ValueError):
        rdata.unserialize(b"A\n2\n")                                          # ascii format
    with pytest.raises(ValueError):
        rdata.unserialize(b"X\n" + (2).to_bytes(4, "big") * 3 + (3).to_bytes(4, "big"))   # a closure


def test_save_load_bigkrls_objects_on_the_host(tmp_path):
    """save_bigKRLS / load_bigKRLS with host-resident members only (n <= 2500 objects hold base matrices,
    R/bigKRLS.R:150), the folder safeguard of make_path (R/bigKRLS_Rcpp_functions.R:272-297) and a
    cross-validation object with its per-fold sub-folders (R/bigKRLS.R:916-932)."""
    import bigkrls_amd as bk
    from bigkrls_amd.api import BigKRLS, BigKRLSCV, BigKRLSPredicted
    rng = np.random.default_rng(0)

    def fit_like(n):
        return BigKRLS({"coeffs": rng.random(n), "X": rng.random((n, 2)), "y": rng.random(n), "lambda": 0.25,
                        "lastkeeper": 5, "xlabs": ["x1", "x2"], "which.derivatives": None, "R2": 0.5,
                        "binaryindicator": np.array([False, True]), "Neffective.acf": None, "derivative.call": True,
                        "avgderivatives": rng.random((1, 2)), "K": rng.random((n, n)), "has.big.matrices": False})

    w = fit_like(7)
    os.chdir(tmp_path)
    folder = bk.save_bigKRLS(w, "model", noisy=False)
    assert folder == "model" and sorted(os.listdir(folder)) == ["estimates.RData"]
    back = bk.load_bigKRLS(folder, noisy=False, to_device=False)
    assert type(back) is BigKRLS and back["path"] == os.path.abspath("model")
    for k, v in w.items():
        assert (v is None and back[k] is None) or np.array_equal(np.asarray(v), np.asarray(back[k])), k
    assert isinstance(back["lambda"], float) and isinstance(back["lastkeeper"], int) and back["xlabs"] == ["x1", "x2"]
    # an existing folder is not reused unless asked for
    assert bk.save_bigKRLS(w, "model", noisy=False) == "model1"
    assert bk.save_bigKRLS(w, "model", noisy=False) == "model2"
    assert bk.save_bigKRLS(w, "model", overwrite_existing=True, noisy=False) == "model"
    # a single-column derivative matrix written as text keeps its shape (read.big.matrix returns n x 1)
    from bigkrls_amd.persist import read_big_matrix_text, write_big_matrix_text
    col = rng.random((7, 1))
    write_big_matrix_text(col, "col.txt")
    assert read_big_matrix_text("col.txt").shape == (7, 1)
    col[3, 0] = np.nan
    write_big_matrix_text(col, "col.txt")
    assert read_big_matrix_text("col.txt").shape == (7, 1)
    write_big_matrix_text(col.T, "row.txt")
    assert read_big_matrix_text("row.txt").shape == (1, 7)
    # cross-validation object
    cv = BigKRLSCV({"type": "KfoldsCV", "Kfolds": 2, "seed": 1, "folds": np.array([1, 2, 1, 2, 1, 2, 1]),
                    "R2_oos": [0.5, 0.6], "MSE_oos": [1.0, 2.0]})
    for k in (1, 2):
        cv[f"fold_{k}"] = {"trained": fit_like(4), "pseudoR2_oos": 0.5,
                           "tested": BigKRLSPredicted({"predicted": rng.random(3), "se.pred": None, "ytest": rng.random(3)})}
    f = bk.save_bigKRLS(cv, "cvout", noisy=False)
    assert os.path.exists(os.path.join(f, "fold_2", "tested", "estimates.RData"))
    cb = bk.load_bigKRLS(f, noisy=False, to_device=False)
    assert type(cb) is BigKRLSCV and cb["Kfolds"] == 2 and np.array_equal(cb["R2_oos"], [0.5, 0.6])
    assert np.array_equal(cb["folds"], cv["folds"])
    for k in (1, 2):
        assert np.array_equal(cb[f"fold_{k}"]["trained"]["coeffs"], cv[f"fold_{k}"]["trained"]["coeffs"])
        assert np.array_equal(cb[f"fold_{k}"]["tested"]["predicted"], cv[f"fold_{k}"]["tested"]["predicted"])
        assert type(cb[f"fold_{k}"]["tested"]) is BigKRLSPredicted
    with pytest.raises(FileNotFoundError):
        bk.load_bigKRLS(str(tmp_path), noisy=False)
