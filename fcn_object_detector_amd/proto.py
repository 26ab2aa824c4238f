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
#   NetParameter { string name = 1; repeated LayerParameter layer = 100; }
#   LayerParameter { string name = 1; string type = 2; repeated BlobProto blobs = 7; }
#   BlobProto { BlobShape shape = 7; repeated float data = 5 [packed];
#               int32 num=1, channels=2, height=3, width=4 (legacy 4-d shape) }
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


def _decode_blob(buf: bytes) -> np.ndarray:
    dims: List[int] = []
    legacy = {}
    chunks: List[np.ndarray] = []
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
        elif num == 5 and wt == 2:
            chunks.append(np.frombuffer(v, dtype="<f4"))
        elif num == 5 and wt == 5:
            chunks.append(np.frombuffer(v, dtype="<f4"))
        elif num in (1, 2, 3, 4) and wt == 0:
            legacy[num] = v
    data = np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
    if not dims and legacy:
        dims = [legacy.get(k, 1) for k in (1, 2, 3, 4)]
    if dims and int(np.prod(dims)) == data.size:
        data = data.reshape(dims)
    return np.array(data, dtype=np.float32)


def read_caffemodel(path: str) -> Dict[str, List[np.ndarray]]:
    """Return ``{layer_name: [blob0, blob1, ...]}`` from a binary NetParameter."""
    with open(path, "rb") as f:
        buf = f.read()
    out: Dict[str, List[np.ndarray]] = {}
    for num, wt, v in _fields(buf):
        if num != 100 or wt != 2:  # V2 'layer'; V1 'layers'=2 is not produced by the reference's Caffe
            continue
        name: Optional[str] = None
        blobs: List[np.ndarray] = []
        for n2, wt2, v2 in _fields(v):
            if n2 == 1 and wt2 == 2:
                name = bytes(v2).decode("utf-8")
            elif n2 == 7 and wt2 == 2:
                blobs.append(_decode_blob(v2))
        if name is not None and blobs:
            out[name] = blobs
    return out


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
