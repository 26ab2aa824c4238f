"""Protobuf text-format (prototxt) reader and binary caffemodel codec.

The reference ships its networks and solver settings as Caffe text protos
(reference: models/deploy.prototxt, models/train_val.prototxt,
train/*/train_val.prototxt, train/*/solver.prototxt) and loads weights from a
binary ``NetParameter`` (reference: scripts/fcn_object_detector.py:317).
``protoc`` and ``caffe.proto`` are not available, so this module reads the text
format generically (no schema) and hand-rolls the small part of the wire format
a caffemodel uses.

Text grammar handled (everything the reference files use):
  * ``key: value`` scalars (numbers, quoted strings with ' or ", enum idents,
    true/false), ``key { ... }`` and ``key: { ... }`` sub-messages
    (train/bounding_box/train_val.prototxt:15 uses the colon form),
  * ``#`` comments to end of line, repeated keys.
"""
from __future__ import annotations

import re
import struct
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np


class Msg:
    """A schemaless protobuf message: field name -> list of values in file order."""

    __slots__ = ("fields",)

    def __init__(self) -> None:
        self.fields: Dict[str, List[Any]] = {}

    def add(self, key: str, value: Any) -> None:
        self.fields.setdefault(key, []).append(value)

    def getall(self, key: str) -> List[Any]:
        return self.fields.get(key, [])

    def get(self, key: str, default: Any = None) -> Any:
        v = self.fields.get(key)
        return v[-1] if v else default

    def has(self, key: str) -> bool:
        return key in self.fields

    def __contains__(self, key: str) -> bool:
        return key in self.fields

    def __repr__(self) -> str:
        return "Msg(%r)" % (self.fields,)


_TOKEN = re.compile(
    r"""\s*(?:
        (?P<comment>\#[^\n]*)            |
        (?P<brace>[{}])                  |
        (?P<colon>:)                     |
        (?P<str>"(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*') |
        (?P<atom>[^\s{}:#"']+)
    )""",
    re.VERBOSE,
)


def _tokens(text: str) -> Iterator[Tuple[str, str]]:
    pos, n = 0, len(text)
    while pos < n:
        m = _TOKEN.match(text, pos)
        if m is None:
            if text[pos:].strip() == "":
                return
            raise ValueError("prototxt: cannot tokenise at offset %d: %r" % (pos, text[pos:pos + 30]))
        pos = m.end()
        kind = m.lastgroup
        if kind == "comment":
            continue
        yield kind, m.group(kind)


def _scalar(kind: str, tok: str) -> Any:
    if kind == "str":
        body = tok[1:-1]
        return body.encode("latin-1", "backslashreplace").decode("unicode_escape") if "\\" in body else body
    low = tok.lower()
    if low == "true":
        return True
    if low == "false":
        return False
    try:
        return int(tok, 0)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok  # enum identifier (MAX, PROD, TRAIN, LMDB, ADAM ...)


def parse_text(text: str) -> Msg:
    """Parse prototxt text into a :class:`Msg` tree."""
    toks = list(_tokens(text))
    i = 0

    def message(depth: int) -> Msg:
        nonlocal i
        msg = Msg()
        while i < len(toks):
            kind, tok = toks[i]
            if kind == "brace" and tok == "}":
                if depth == 0:
                    raise ValueError("prototxt: unbalanced '}'")
                i += 1
                return msg
            if kind != "atom":
                raise ValueError("prototxt: expected field name, got %r" % tok)
            key = tok
            i += 1
            if i >= len(toks):
                raise ValueError("prototxt: dangling field %r" % key)
            kind, tok = toks[i]
            if kind == "colon":
                i += 1
                if i >= len(toks):
                    raise ValueError("prototxt: missing value for %r" % key)
                kind, tok = toks[i]
            if kind == "brace" and tok == "{":
                i += 1
                msg.add(key, message(depth + 1))
            elif kind in ("str", "atom"):
                i += 1
                msg.add(key, _scalar(kind, tok))
            else:
                raise ValueError("prototxt: bad value %r for field %r" % (tok, key))
        if depth != 0:
            raise ValueError("prototxt: missing '}'")
        return msg

    return message(0)


def parse_file(path: str) -> Msg:
    with open(path, "r") as f:
        return parse_text(f.read())


# ----------------------------------------------------------------------------
# Binary NetParameter (".caffemodel") — the subset that carries weights.
#
#   NetParameter { string name = 1; repeated V1LayerParameter layers = 2; repeated LayerParameter layer = 100; }
#   LayerParameter { string name = 1; string type = 2; repeated BlobProto blobs = 7; }
#   V1LayerParameter { string name = 4; LayerType type = 5; repeated BlobProto blobs = 6; }
#   BlobProto { BlobShape shape = 7; repeated float data = 5 [packed]; repeated double double_data = 8 [packed];
#               int32 num=1, channels=2, height=3, width=4 (legacy 4-d shape);
#               NVCaffe: Type raw_data_type = 10; bytes raw_data = 12 }
#   BlobShape { repeated int64 dim = 1 [packed]; }
# Field numbers restated from the public BVLC caffe.proto (not vendored by the
# reference); unknown fields are skipped on read.
# ----------------------------------------------------------------------------

def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("caffemodel: varint too long")


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf: bytes) -> Iterator[Tuple[int, int, Any]]:
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("caffemodel: unsupported wire type %d" % wt)
        if pos > n:
            raise ValueError("caffemodel: truncated field %d" % num)
        yield num, wt, v


# NVCaffe's BlobProto extension (0.16+; restated from the public NVIDIA/caffe caffe.proto): enum Type and raw byte storage
_NV_TYPE_DTYPE = {0: "<f8", 1: "<f4", 2: "<f2", 3: "<i4", 4: "<u4"}      # DOUBLE, FLOAT, FLOAT16, INT, UINT


def _packed_or_single(chunks: List[np.ndarray], wt: int, v: Any, dtype: str, single_wt: int) -> None:
    if wt == 2 or wt == single_wt:      # packed run, or one un-packed element (both are legal encodings of a repeated scalar)
        chunks.append(np.frombuffer(bytes(v), dtype=dtype))


def _decode_blob(buf: bytes) -> np.ndarray:
    """BlobProto -> float32 array.  `data` (5, float) / `double_data` (8, double, what a double-precision Caffe build
    writes) / NVCaffe `raw_data` (12, bytes, element type in `raw_data_type` = 10); shape from BlobShape (7) or the legacy
    num / channels / height / width (1..4) of V1-era files; diff fields (6, 9, 11, 13) are ignored as Caffe ignores them
    when copying trained layers."""
    dims: List[int] = []
    legacy = {}
    f32: List[np.ndarray] = []
    f64: List[np.ndarray] = []
    raw: Optional[bytes] = None
    raw_type = 1
    for num, wt, v in _fields(buf):
        if num == 7 and wt == 2:  # BlobShape
            for n2, wt2, v2 in _fields(v):
                if n2 == 1 and wt2 == 2:
                    p = 0
                    while p < len(v2):
                        d, p = _varint(v2, p)
                        dims.append(d)
                elif n2 == 1 and wt2 == 0:
                    dims.append(v2)
        elif num == 5:
            _packed_or_single(f32, wt, v, "<f4", 5)
        elif num == 8:
            _packed_or_single(f64, wt, v, "<f8", 1)
        elif num == 10 and wt == 0:
            raw_type = int(v)
        elif num == 12 and wt == 2:
            raw = bytes(v)
        elif num in (1, 2, 3, 4) and wt == 0:
            legacy[num] = v
    if f32:
        data = np.concatenate(f32)
    elif f64:
        data = np.concatenate(f64).astype(np.float32)
    elif raw is not None:
        if raw_type not in _NV_TYPE_DTYPE:
            raise ValueError("caffemodel: raw_data of unknown raw_data_type %d" % raw_type)
        data = np.frombuffer(raw, dtype=_NV_TYPE_DTYPE[raw_type]).astype(np.float32)
    else:
        data = np.zeros(0, np.float32)
    if not dims and legacy:
        dims = [legacy.get(k, 1) for k in (1, 2, 3, 4)]
    if dims and int(np.prod(dims)) == data.size:
        data = data.reshape(dims)
    return np.array(data, dtype=np.float32)


def read_caffemodel(path: str) -> Dict[str, List[np.ndarray]]:
    """Return ``{layer_name: [blob0, blob1, ...]}`` from a binary NetParameter.

    Both generations of the format are read: ``layer`` (field 100, LayerParameter: name = 1, blobs = 7) and the V1
    ``layers`` (field 2, V1LayerParameter: name = 4, blobs = 6) that model-zoo files such as
    VGG_ILSVRC_16_layers.caffemodel - the fine-tune source of the reference's train/bounding_box/train.sh:12-15 - are
    stored in.  Caffe upgrades V1 nets on load (UpgradeV1Net) and then copies by layer name; the names survive unchanged."""
    with open(path, "rb") as f:
        buf = f.read()
    out: Dict[str, List[np.ndarray]] = {}
    for num, wt, v in _fields(buf):
        if wt != 2 or num not in (100, 2):
            continue
        name_field, blob_field = (1, 7) if num == 100 else (4, 6)
        name: Optional[str] = None
        blobs: List[np.ndarray] = []
        for n2, wt2, v2 in _fields(v):
            if n2 == name_field and wt2 == 2:
                name = bytes(v2).decode("utf-8")
            elif n2 == blob_field and wt2 == 2:
                blobs.append(_decode_blob(v2))
            elif num == 2 and n2 == 1 and wt2 == 2:
                raise ValueError("%s: V0-format layer (V1LayerParameter.layer = 1) - upgrade the file with Caffe's upgrade_net_proto_binary" % path)
        if name is not None and blobs:
            out[name] = blobs
    return out


def copy_trained_layers(path: str, params_host: Dict[str, List[np.ndarray]], set_params, log=None) -> List[str]:
    """Net::CopyTrainedLayersFrom: copy the blobs of every layer of `path` whose NAME the net has; source layers the net
    lacks are ignored (logged, as Caffe does), layers of the net the file lacks keep their values.  Blob counts and
    element counts must fit.  A file that holds no parameter blobs at all is an error, and a file none of whose layers
    match is reported loudly: `--weights` would otherwise train from the fillers while claiming to fine-tune."""
    import sys
    if log is None:
        log = lambda m: sys.stderr.write(m + "\n")      # noqa: E731
    blobs = read_caffemodel(path)
    if not blobs:
        raise ValueError("%s: no layer with parameter blobs found (not a binary NetParameter, or a format this reader does not know)" % path)
    copied: List[str] = []
    for lname, arrs in blobs.items():
        want = params_host.get(lname)
        if want is None:
            log("Ignoring source layer %s" % lname)
            continue
        if len(arrs) != len(want):
            raise ValueError("Incompatible number of blobs for layer %s: %s has %d, the net needs %d" % (lname, path, len(arrs), len(want)))
        for a, w in zip(arrs, want):
            if a.size != w.size:
                raise ValueError("Cannot copy param of layer %s: %s holds %s, the net needs %s" % (lname, path, a.shape, w.shape))
        set_params(lname, [a.reshape(w.shape) for a, w in zip(arrs, want)])
        copied.append(lname)
    if not copied:
        log("WARNING: %s matched NONE of the net's %d parameter layers (%d source layers ignored): every layer keeps its filler values"
            % (path, len(params_host), len(blobs)))
    return copied


def _ld(num: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def write_caffemodel(path: str, layers: List[Tuple[str, str, List[np.ndarray]]], net_name: str = "") -> None:
    """Write ``[(layer_name, layer_type, [blobs])]`` as a binary NetParameter."""
    out = bytearray()
    if net_name:
        out += _ld(1, net_name.encode("utf-8"))
    for lname, ltype, blobs in layers:
        lay = bytearray()
        lay += _ld(1, lname.encode("utf-8"))
        lay += _ld(2, ltype.encode("utf-8"))
        for b in blobs:
            arr = np.ascontiguousarray(b, dtype="<f4")
            shape = b"".join(_enc_varint(int(d)) for d in arr.shape)
            blob = _ld(7, _ld(1, shape)) + _ld(5, arr.tobytes())
            lay += _ld(7, blob)
        out += _ld(100, bytes(lay))
    with open(path, "wb") as f:
        f.write(bytes(out))


def pack_solverstate(it: int, history: List[np.ndarray], learned_net: str = "") -> bytes:
    """SolverState { int32 iter = 1; string learned_net = 2; repeated BlobProto history = 3; } (public caffe.proto)."""
    out = bytearray()
    out += _enc_varint((1 << 3) | 0) + _enc_varint(it)
    if learned_net:
        out += _ld(2, learned_net.encode("utf-8"))
    for h in history:
        arr = np.ascontiguousarray(h, dtype="<f4")
        shape = b"".join(_enc_varint(int(d)) for d in arr.shape)
        out += _ld(3, _ld(7, _ld(1, shape)) + _ld(5, arr.tobytes()))
    return bytes(out)


def unpack_solverstate(buf: bytes, with_learned_net: bool = False):
    it, hist, learned = 0, [], ""
    for num, wt, v in _fields(buf):
        if num == 1 and wt == 0:
            it = v
        elif num == 2 and wt == 2:
            learned = bytes(v).decode("utf-8")
        elif num == 3 and wt == 2:
            hist.append(_decode_blob(v))
    return (it, hist, learned) if with_learned_net else (it, hist)
