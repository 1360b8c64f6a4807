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


def _v3_stream(body: bytes, rdata_magic: bool) -> bytes:
    """Header of a version-3 stream as R >= 3.5 writes it: 'X\\n', version 3, writer 4.3.1, min reader 3.5.0, the
    native encoding's name (R Internals, 'Serialization Formats')."""
    import struct
    head = (b"RDX3\n" if rdata_magic else b"") + b"X\n" + struct.pack(">iii", 3, 0x00040301, 0x00030500)
    return head + struct.pack(">i", 5) + b"UTF-8" + body


def test_reader_accepts_version_3_rdata_and_expands_altrep():
    """What `save()` of a current R (>= 3.5) emits and the version-2 writer never does: the RDX3 magic, the encoding
    in the header, ALTREP items. Streams assembled by hand from the documented layout (no R here): a compact integer
    sequence 1:5 (what `folds` or row names become), a wrap_real around a numeric vector, a deferred as.character."""
    import struct
    w = rdata._Writer.__new__(rdata._Writer)

    def flags(v):
        return struct.pack(">I", v)

    def sym(name):
        b = name.encode()
        return flags(rdata.SYMSXP) + flags(rdata.CHARSXP | rdata.GP_ASCII) + struct.pack(">i", len(b)) + b

    def real(vals):
        return flags(rdata.REALSXP) + struct.pack(">i", len(vals)) + np.asarray(vals, dtype=">f8").tobytes()

    def intv(vals):
        return flags(rdata.INTSXP) + struct.pack(">i", len(vals)) + np.asarray(vals, dtype=">i4").tobytes()

    nil = flags(rdata.NILVALUE_SXP)

    def info(cls, pkg, typ):
        cell = flags(rdata.LISTSXP)
        return cell + sym(cls) + cell + sym(pkg) + cell + intv([typ]) + nil

    def altrep(cls, typ, state):
        return flags(rdata.ALTREP_SXP) + info(cls, "base", typ) + state + nil

    seq = altrep("compact_intseq", rdata.INTSXP, real([5, 1, 1]))
    # the state of wrap_* is CONS(x, metadata), a DOTTED pair (wrapper_Serialized_state, src/main/altclasses.c): one
    # LISTSXP cell whose CDR is the integer vector itself -- no terminating NULL
    wrapped = altrep("wrap_real", rdata.REALSXP, flags(rdata.LISTSXP) + real([0.5, 2.25]) + intv([0, 0]))
    # (the second and third items refer back to the symbols 'base' etc. only in R's own output; separate streams
    #  here keep the reference table out of the way)
    o = rdata.unserialize(_v3_stream(seq, False))
    assert o.kind == "int" and o.values.tolist() == [1, 2, 3, 4, 5]
    o = rdata.unserialize(_v3_stream(wrapped, False))
    assert o.kind == "real" and o.values.tolist() == [0.5, 2.25]
    # deferred_string: CONS(arg, scipen), dotted likewise (deferred_string_Serialized_state)
    deferred = altrep("deferred_string", rdata.STRSXP, flags(rdata.LISTSXP) + intv([3, 10]) + intv([0]))
    o = rdata.unserialize(_v3_stream(deferred, False))
    assert o.kind == "str" and o.values == ["3", "10"]
    # (the layouts an earlier version of this test pinned -- a VECSXP(2) / a proper two-cell pairlist -- still parse)
    o = rdata.unserialize(_v3_stream(altrep("wrap_real", rdata.REALSXP, flags(rdata.VECSXP) + struct.pack(">i", 2) +
                                            real([0.5, 2.25]) + intv([0, 0])), False))
    assert o.values.tolist() == [0.5, 2.25]
    # a dotted pair outside ALTREP keeps its tail
    pl = rdata.unserialize(_v3_stream(flags(rdata.LISTSXP) + real([1.0]) + intv([7]), False))
    assert isinstance(pl, rdata.Pairlist) and pl[0][1].values.tolist() == [1.0] and pl.dotted_tail.values.tolist() == [7]
    # an .RData file: RDX3 magic + a tagged pairlist
    body = flags(rdata.LISTSXP | rdata.HAS_TAG) + sym("folds") + seq + nil
    pl = rdata.unserialize(gzip.compress(_v3_stream(body, True)))
    assert isinstance(pl, rdata.Pairlist) and pl[0][0] == "folds" and pl[0][1].values.tolist() == [1, 2, 3, 4, 5]
    # an ALTREP class this reader does not know is refused by name, with the way out
    with pytest.raises(ValueError, match="mmap_real.*version = 2"):
        rdata.unserialize(_v3_stream(altrep("mmap_real", rdata.REALSXP, real([1.0])), False))
    # a version-2 stream written here still loads when labelled as what a current R calls it
    v2 = rdata.serialize([rdata.from_python(np.arange(3.0))], rdata_names=["x"])
    assert v2[:5] == b"RDX2\n"


def test_missing_values_are_written_as_r_na_real(tmp_path):
    """Quirk Q6's `NA` entries (R/bigKRLS.R:425-431) reach the host side as NaN; in the file they are R's NA_real_
    (a NaN with low word 1954), which R prints as NA, not NaN. A plain serialize() keeps NaN bits as they are."""
    path = str(tmp_path / "na.RData")
    v = np.array([1.0, np.nan, 3.0])
    rdata.save_rdata(path, {"x": v})
    raw = gzip.decompress(open(path, "rb").read())
    assert raw.count(rdata._NA_REAL_BYTES) == 1
    back = rdata.load_rdata(path)["x"].values
    assert np.isnan(back[1]) and back[[0, 2]].tolist() == [1.0, 3.0]
    assert back[1:2].astype(">f8").tobytes() == rdata._NA_REAL_BYTES          # the payload survives the round trip
    assert rdata._NA_REAL_BYTES not in rdata.serialize(rdata.from_python(v))   # nan_as_na is save_rdata's choice
    assert np.isnan(rdata.NA_REAL)
