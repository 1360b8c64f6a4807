"""R's serialisation format (XDR; written as version 2, read as version 2 or 3) -- enough of it to write and read the `estimates.RData` that
save.bigKRLS / load.bigKRLS exchange (R/bigKRLS.R:901-945, 960-1020; `save(bigKRLS_out, file = ...)` at
R/bigKRLS.R:495 and R/bigKRLS_Rcpp_functions.R:311, `load()` at :349).

R itself is absent from this image. The format is R's documented one ("R Internals", section "Serialization
Formats"): a stream starts with the format line "X\\n", three big-endian ints (format version 2, the writing R
version, the oldest R that can read it) and then one item; an `.RData` file prefixes the magic "RDX2\\n" and
its item is a pairlist whose tags are the saved objects' names; both are gzip-compressed by default. An item
is a flags word -- SEXPTYPE in bits 0-7, is-object 0x100, has-attributes 0x200, has-tag 0x400, the `gp`
field from bit 12 -- followed by the type's payload; symbols are written once and referred to afterwards
(REFSXP, index << 8 | 0xff); attributes follow a vector's payload as a pairlist. The one file in this format
that the reference tree holds, build/vignette.rds (a data.frame written by R 3.3.3), is kept as
tests/golden/r_serialize_v2_vignette_index.rds: the reader must parse it and the writer must reproduce its
stream byte for byte (tests/test_rdata.py).

Version 3 (the default of `save()` / `saveRDS()` since R 3.5; magic "RDX3\n") differs for data in two ways, both
handled by the reader: the header also carries the native encoding's name, and vectors may arrive as ALTREP items
(type 238: a class description, a state, attributes) -- the compact integer / real sequences (`1:n`), the `wrap_*`
classes (a vector plus sortedness metadata) and deferred `as.character(<numbers>)` are expanded, any other ALTREP
class is refused by name. The writer always emits version 2, which every R since 2.3.0 loads; no R session was
available here, so files written by R >= 3.5 are covered by hand-assembled version-3 streams only.

Objects: R vectors map to `RVec(kind, values, attrs)` with kind in {"lgl", "int", "real", "str", "list"};
NULL is None. `from_python` / `to_python` convert between that tree and plain Python (dict = named list,
numpy arrays with `dim`, str / list of str = character vectors).
"""
from __future__ import annotations

import gzip
import io
import struct
from typing import Any, List, Optional, Tuple

import numpy as np

NILSXP, SYMSXP, LISTSXP, CHARSXP, LGLSXP, INTSXP, REALSXP, STRSXP, VECSXP = 0, 1, 2, 9, 10, 13, 14, 16, 19
REFSXP, NILVALUE_SXP, ALTREP_SXP = 255, 254, 238
IS_OBJECT, HAS_ATTR, HAS_TAG = 0x100, 0x200, 0x400
GP_ASCII, GP_UTF8 = 0x40 << 12, 0x08 << 12
NA_INT = -2 ** 31
_NA_REAL_BYTES = struct.pack(">II", 0x7FF00000, 1954)         # R's NA_real_: a NaN whose low word is 1954
NA_REAL = float(np.frombuffer(_NA_REAL_BYTES, dtype=">f8")[0])   # (the payload survives copies and byte swaps)
R_3_3_3, R_2_3_0 = 0x00030303, 0x00020300

_KIND = {LGLSXP: "lgl", INTSXP: "int", REALSXP: "real", STRSXP: "str", VECSXP: "list"}
_TYPE = {v: k for k, v in _KIND.items()}


class RVec:
    """An R vector: values (numpy array for lgl / int / real -- NA as NA_INT / NaN --, list of str-or-None for
    str, list of RVec-or-None for list) and its attributes as an ordered list of (name, value)."""

    def __init__(self, kind: str, values, attrs: Optional[List[Tuple[str, Any]]] = None):
        self.kind, self.values, self.attrs = kind, values, list(attrs or [])

    def attr(self, name: str):
        for k, v in self.attrs:
            if k == name:
                return v
        return None

    @property
    def is_object(self) -> bool:
        return self.attr("class") is not None


class Pairlist(list):
    """[(tag-or-None, value), ...]; `dotted_tail` holds the CDR of the last cell when it is not NULL (a dotted pair:
    what R's ALTREP classes serialise their state as, CONS(x, metadata))"""
    dotted_tail = None


# ---------------------------------------------------------------------------------------------------- writer
class _Writer:
    def __init__(self, out: io.BytesIO, nan_as_na: bool = False):
        self.out, self.symbols, self.nan_as_na = out, {}, nan_as_na

    def i32(self, v: int):
        self.out.write(struct.pack(">i", v))

    def flags(self, v: int):
        self.out.write(struct.pack(">I", v))

    def charsxp(self, s: Optional[str]):
        if s is None:
            self.flags(CHARSXP)
            self.i32(-1)
            return
        b = s.encode("utf-8")
        self.flags(CHARSXP | (GP_ASCII if len(b) == len(s) and s.isascii() else GP_UTF8))
        self.i32(len(b))
        self.out.write(b)

    def symbol(self, name: str):
        if name in self.symbols:
            self.flags((self.symbols[name] << 8) | REFSXP)
            return
        self.symbols[name] = len(self.symbols) + 1
        self.flags(SYMSXP)
        self.charsxp(name)

    def pairlist(self, items):
        for tag, value in items:
            self.flags(LISTSXP | (HAS_TAG if tag is not None else 0))
            if tag is not None:
                self.symbol(tag)
            self.item(value)
        self.flags(NILVALUE_SXP)

    def item(self, obj):
        if obj is None:
            self.flags(NILVALUE_SXP)
            return
        if isinstance(obj, Pairlist):
            self.pairlist(obj)
            return
        assert isinstance(obj, RVec), type(obj)
        f = _TYPE[obj.kind] | (HAS_ATTR if obj.attrs else 0) | (IS_OBJECT if obj.is_object else 0)
        self.flags(f)
        n = len(obj.values)
        self.i32(n)
        if obj.kind in ("lgl", "int"):
            self.out.write(np.asarray(obj.values, dtype=">i4").tobytes())
        elif obj.kind == "real":
            v = np.asarray(obj.values, dtype=np.float64)
            if self.nan_as_na and np.isnan(v).any():
                # a missing value of the host side (Python has no NA) is R's NA_real_, not an IEEE NaN: is.na() is
                # TRUE for both, but R prints and summarises them differently
                b = v.astype(">f8").view(">u8").copy()
                b[np.isnan(v)] = np.frombuffer(_NA_REAL_BYTES, dtype=">u8")[0]
                self.out.write(b.tobytes())
            else:
                self.out.write(v.astype(">f8").tobytes())
        elif obj.kind == "str":
            for s in obj.values:
                self.charsxp(s)
        else:
            for e in obj.values:
                self.item(e)
        if obj.attrs:
            self.pairlist(obj.attrs)


def serialize(obj, rdata_names: Optional[List[str]] = None, nan_as_na: bool = False) -> bytes:
    """The uncompressed stream of `saveRDS(obj)`, or -- with rdata_names -- of `save(<names>)` where obj is the
    list of the saved values. nan_as_na: NaN entries of numeric vectors are written as R's NA_real_."""
    out = io.BytesIO()
    if rdata_names is not None:
        out.write(b"RDX2\n")
    out.write(b"X\n")
    w = _Writer(out, nan_as_na)
    w.i32(2)
    w.i32(R_3_3_3)
    w.i32(R_2_3_0)
    if rdata_names is not None:
        w.pairlist(list(zip(rdata_names, obj)))
    else:
        w.item(obj)
    return out.getvalue()


# ---------------------------------------------------------------------------------------------------- reader
class _Reader:
    def __init__(self, data: bytes):
        self.d, self.p, self.refs = data, 0, []

    def take(self, n: int) -> bytes:
        b = self.d[self.p:self.p + n]
        if len(b) != n:
            raise ValueError("R serialisation stream ends early")
        self.p += n
        return b

    def i32(self) -> int:
        return struct.unpack(">i", self.take(4))[0]

    def charsxp(self, flags: int) -> Optional[str]:
        n = self.i32()
        if n == -1:
            return None
        b = self.take(n)
        return b.decode("latin-1" if flags & (0x04 << 12) else "utf-8")

    def item(self, flags: Optional[int] = None):
        if flags is None:
            flags = struct.unpack(">I", self.take(4))[0]
        t = flags & 0xFF
        if t == NILVALUE_SXP or t == NILSXP:
            return None
        if t == REFSXP:
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == SYMSXP:
            inner = struct.unpack(">I", self.take(4))[0]
            if inner & 0xFF != CHARSXP:
                raise ValueError("symbol without a print name")
            name = _Symbol(self.charsxp(inner))
            self.refs.append(name)
            return name
        if t == LISTSXP:
            items = Pairlist()
            while True:
                if flags & HAS_ATTR:
                    self.item()                    # attributes of a pairlist cell: not used by save()
                tag = self.item() if flags & HAS_TAG else None
                items.append((str(tag) if tag is not None else None, self.item()))
                flags = struct.unpack(">I", self.take(4))[0]
                if flags & 0xFF in (NILVALUE_SXP, NILSXP):
                    return items
                if flags & 0xFF != LISTSXP:
                    # a dotted pair: the CDR is an ordinary item whose flags have just been read (R writes the CDR of
                    # a cell by a tail call of WriteItem, src/main/serialize.c) -- the state of wrap_* and
                    # deferred_string ALTREP objects is CONS(x, <integer metadata>)
                    items.dotted_tail = self.item(flags)
                    return items
        if t == CHARSXP:
            return self.charsxp(flags)
        if t == ALTREP_SXP:
            return self.altrep(flags)
        if t not in _KIND:
            raise ValueError(f"R object of type {t} is not supported (closures, environments, ... are not data)")
        n = self.i32()
        if t in (LGLSXP, INTSXP):
            values = np.frombuffer(self.take(4 * n), dtype=">i4").astype(np.int32)
        elif t == REALSXP:
            values = np.frombuffer(self.take(8 * n), dtype=">f8").astype(np.float64)
        elif t == STRSXP:
            values = [self.item() for _ in range(n)]
        else:
            values = [self.item() for _ in range(n)]
        attrs = self.item() if flags & HAS_ATTR else []
        return RVec(_KIND[t], values, [(k, v) for k, v in (attrs or [])])


class _Symbol(str):
    pass


def _altrep(self, flags: int):
    """An ALTREP item (serialisation version 3): info = pairlist(class symbol, package symbol, base type), state,
    attributes. The classes base R uses for plain data are expanded into ordinary vectors."""
    info, state, attrs = self.item(), self.item(), self.item()
    cls = str(info[0][1]) if isinstance(info, Pairlist) and info else "?"
    attrs = [(k, v) for k, v in (attrs or [])]
    if cls in ("compact_intseq", "compact_realseq"):
        n, start, step = (float(x) for x in state.values[:3])
        seq = start + step * np.arange(int(n), dtype=np.float64)
        if cls == "compact_intseq":
            return RVec("int", seq.astype(np.int32), attrs)
        return RVec("real", seq, attrs)
    if cls.startswith("wrap_"):                       # state = CONS(x, metadata) (a dotted pair): the wrapped vector itself
        inner = state.values[0] if isinstance(state, RVec) else state[0][1]
        if isinstance(inner, RVec):
            return RVec(inner.kind, inner.values, inner.attrs + attrs)
        return inner
    if cls == "deferred_string":                      # state = CONS(arg, scipen): as.character(<integer or real vector>), not yet expanded
        arg = state[0][1] if isinstance(state, Pairlist) else state
        if isinstance(arg, RVec) and arg.kind == "int":
            vals = [None if int(v) == NA_INT else str(int(v)) for v in arg.values]
            return RVec("str", vals, attrs)
        if isinstance(arg, RVec) and arg.kind == "real":
            vals = [None if np.isnan(v) else (str(int(v)) if float(v).is_integer() and abs(v) < 1e15 else repr(float(v)))
                    for v in arg.values]
            return RVec("str", vals, attrs)
    raise ValueError(f"ALTREP class '{cls}' is not supported by this reader: in R, re-save the object with "
                     "save(..., version = 2), which expands it")


_Reader.altrep = _altrep


def unserialize(data: bytes):
    """Inverse of serialize(): returns the object of an .rds stream, or a Pairlist [(name, value), ...] for an
    .RData stream. Accepts gzip-compressed input."""
    if data[:2] == b"\x1f\x8b":
        data = gzip.decompress(data)
    rdata = data[:5] in (b"RDX2\n", b"RDX3\n")      # save() of R < 3.5 (or version = 2) / of R >= 3.5
    if data[:3] == b"RDA" or data[:3] == b"RDB":
        raise ValueError("ASCII / native-binary .RData files are not supported: save(..., ascii = FALSE) writes XDR")
    if rdata:
        data = data[5:]
    if data[:2] != b"X\n":
        raise ValueError("not an XDR R serialisation stream (only format 'X' is supported)")
    r = _Reader(data)
    r.p = 2
    version = r.i32()
    r.i32()
    r.i32()
    if version == 3:                      # R >= 3.5 also records the native encoding
        r.take(r.i32())
    elif version != 2:
        raise ValueError(f"R serialisation version {version} is not supported")
    obj = r.item()
    if rdata and not isinstance(obj, Pairlist):
        raise ValueError(".RData stream does not hold a pairlist")
    return obj


# ------------------------------------------------------------------------------------ Python <-> R objects
def from_python(v, r_class: Optional[str] = None):
    """dict -> named list; numpy float / int / bool arrays -> numeric / integer / logical vectors (2-D: with `dim`,
    column-major); float / int / bool -> vectors of length 1; str, list of str -> character; None -> NULL;
    any other list / tuple -> unnamed list."""
    if v is None or isinstance(v, (RVec, Pairlist)):
        return v
    if isinstance(v, dict):
        keys = [k for k in v.keys()]
        attrs = [("names", RVec("str", [str(k) for k in keys]))]
        cls = r_class or getattr(v, "r_class", None)
        if cls:
            attrs.append(("class", RVec("str", [cls])))
        return RVec("list", [from_python(v[k]) for k in keys], attrs)
    if isinstance(v, str):
        return RVec("str", [v])
    if isinstance(v, (bool, np.bool_)):
        return RVec("lgl", np.array([int(v)], dtype=np.int32))
    if isinstance(v, (int, np.integer)):
        return RVec("int", np.array([int(v)], dtype=np.int32))
    if isinstance(v, (float, np.floating)):
        return RVec("real", np.array([float(v)]))
    if isinstance(v, (list, tuple)) and all(isinstance(s, str) or s is None for s in v) and len(v) > 0:
        return RVec("str", list(v))
    if isinstance(v, (list, tuple)) and not all(isinstance(e, (int, float, bool, np.number)) for e in v):
        return RVec("list", [from_python(e) for e in v])
    a = np.asarray(v)
    attrs = []
    if a.ndim >= 2:
        attrs.append(("dim", RVec("int", np.array(a.shape, dtype=np.int32))))
    flat = a.reshape(-1, order="F")
    if a.dtype == np.bool_:
        return RVec("lgl", flat.astype(np.int32), attrs)
    if np.issubdtype(a.dtype, np.integer):
        return RVec("int", flat.astype(np.int32), attrs)
    if a.dtype.kind in "US":
        return RVec("str", [str(s) for s in flat], attrs)
    return RVec("real", flat.astype(np.float64), attrs)


def to_python(o):
    """RVec tree -> dict (named lists; the class, if any, under key "__class__") / numpy arrays (shaped by `dim`) /
    list of str / list; vectors keep their length (R has no scalars)."""
    if o is None:
        return None
    if isinstance(o, Pairlist):
        return {k: to_python(v) for k, v in o}
    if o.kind == "list":
        names = o.attr("names")
        vals = [to_python(e) for e in o.values]
        if names is None:
            return vals
        d = dict(zip(names.values, vals))
        cls = o.attr("class")
        if cls is not None:
            d["__class__"] = cls.values[0]
        return d
    if o.kind == "str":
        return list(o.values)
    a = np.array(o.values)
    if o.kind == "lgl":
        a = a.astype(bool) if not (a == NA_INT).any() else a
    dim = o.attr("dim")
    if dim is not None:
        a = a.reshape(tuple(int(x) for x in dim.values), order="F")
    return a


def save_rdata(path: str, objects: dict, compress: bool = True, nan_as_na: bool = True) -> None:
    """`save(<names>, file = path, version = 2)` for the objects of the dict (values: anything from_python accepts).
    NaN in numeric vectors goes out as NA_real_ (the host side has no NA of its own: the `NA` entries R produces
    through quirk Q6, R/bigKRLS.R:425-431 with which.derivatives, arrive here as NaN)."""
    names = list(objects.keys())
    stream = serialize([from_python(objects[k]) for k in names], rdata_names=names, nan_as_na=nan_as_na)
    with open(path, "wb") as f:
        f.write(gzip.compress(stream, mtime=0) if compress else stream)


def load_rdata(path: str) -> dict:
    """`load(path)`: {name: RVec tree}."""
    with open(path, "rb") as f:
        pl = unserialize(f.read())
    if not isinstance(pl, Pairlist):
        raise ValueError(f"{path} is not an .RData file")
    return {k: v for k, v in pl}
