"""The .caffemodel / .solverstate codec against an INDEPENDENT implementation of the protobuf wire format.

`proto.read_caffemodel` / `write_caffemodel` used to be tested only against each other.  Here the files are produced and
parsed by the google.protobuf runtime from a descriptor built with the field numbers of the public caffe.proto (BVLC, plus
NVCaffe's raw_data extension) - both generations of the format: V2 `layer` (100) and the V1 `layers` (2) that the model-zoo
file of the reference's fine-tune script is stored in (reference: train/bounding_box/train.sh:12-15, `--weights
VGG_ILSVRC_16_layers.caffemodel`).  CPU only.
"""
import struct

import numpy as np
import pytest

from fcn_object_detector_amd import proto

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

T = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=T.LABEL_OPTIONAL, type_name=None, packed=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed


@pytest.fixture(scope="module")
def caffe_pb():
    """Message classes for the subset of caffe.proto a weights file uses (public field numbers)."""
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "caffe_subset.proto", "caffe_subset", "proto2"
    shape = fd.message_type.add()
    shape.name = "BlobShape"
    _field(shape, "dim", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    blob = fd.message_type.add()
    blob.name = "BlobProto"
    _field(blob, "shape", 7, T.TYPE_MESSAGE, type_name=".caffe_subset.BlobShape")
    _field(blob, "data", 5, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    _field(blob, "diff", 6, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    _field(blob, "double_data", 8, T.TYPE_DOUBLE, T.LABEL_REPEATED, packed=True)
    _field(blob, "double_diff", 9, T.TYPE_DOUBLE, T.LABEL_REPEATED, packed=True)
    _field(blob, "raw_data_type", 10, T.TYPE_INT32)      # NVCaffe: enum Type { DOUBLE = 0; FLOAT = 1; FLOAT16 = 2; .. }
    _field(blob, "raw_data", 12, T.TYPE_BYTES)
    for i, nm in enumerate(("num", "channels", "height", "width"), start=1):
        _field(blob, nm, i, T.TYPE_INT32)
    unpacked = fd.message_type.add()      # a writer that does not pack repeated scalars (legal for a proto2 reader to meet)
    unpacked.name = "BlobProtoUnpacked"
    _field(unpacked, "shape", 7, T.TYPE_MESSAGE, type_name=".caffe_subset.BlobShape")
    _field(unpacked, "data", 5, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=False)
    lay = fd.message_type.add()
    lay.name = "LayerParameter"
    _field(lay, "name", 1, T.TYPE_STRING)
    _field(lay, "type", 2, T.TYPE_STRING)
    _field(lay, "bottom", 3, T.TYPE_STRING, T.LABEL_REPEATED)
    _field(lay, "top", 4, T.TYPE_STRING, T.LABEL_REPEATED)
    _field(lay, "phase", 10, T.TYPE_INT32)
    _field(lay, "blobs", 7, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".caffe_subset.BlobProto")
    v1 = fd.message_type.add()
    v1.name = "V1LayerParameter"
    _field(v1, "bottom", 2, T.TYPE_STRING, T.LABEL_REPEATED)
    _field(v1, "top", 3, T.TYPE_STRING, T.LABEL_REPEATED)
    _field(v1, "name", 4, T.TYPE_STRING)
    _field(v1, "type", 5, T.TYPE_INT32)                   # enum LayerType (CONVOLUTION = 4, RELU = 18, ..)
    _field(v1, "blobs", 6, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".caffe_subset.BlobProto")
    _field(v1, "blobs_lr", 7, T.TYPE_FLOAT, T.LABEL_REPEATED)
    _field(v1, "weight_decay", 8, T.TYPE_FLOAT, T.LABEL_REPEATED)
    net = fd.message_type.add()
    net.name = "NetParameter"
    _field(net, "name", 1, T.TYPE_STRING)
    _field(net, "layers", 2, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".caffe_subset.V1LayerParameter")
    _field(net, "input", 3, T.TYPE_STRING, T.LABEL_REPEATED)
    _field(net, "input_dim", 4, T.TYPE_INT32, T.LABEL_REPEATED)
    _field(net, "layer", 100, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".caffe_subset.LayerParameter")
    st = fd.message_type.add()
    st.name = "SolverState"
    _field(st, "iter", 1, T.TYPE_INT32)
    _field(st, "learned_net", 2, T.TYPE_STRING)
    _field(st, "history", 3, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".caffe_subset.BlobProto")
    _field(st, "current_step", 4, T.TYPE_INT32)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    names = ("BlobShape", "BlobProto", "BlobProtoUnpacked", "LayerParameter", "V1LayerParameter", "NetParameter", "SolverState")
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe_subset." + n)) for n in names}


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


def test_reads_v2_file_written_by_protobuf_runtime(caffe_pb, tmp_path):
    rng = np.random.default_rng(0)
    w, b = _rand(rng, 6, 3, 3, 3), _rand(rng, 6)
    net = caffe_pb["NetParameter"]()
    net.name = "v2"
    relu = net.layer.add()
    relu.name, relu.type = "relu1", "ReLU"                 # a layer without blobs must not appear
    lay = net.layer.add()
    lay.name, lay.type = "conv1", "Convolution"
    lay.bottom.append("data")
    lay.top.append("conv1")
    for arr in (w, b):
        bl = lay.blobs.add()
        bl.shape.dim.extend(arr.shape)
        bl.data.extend(arr.ravel().tolist())
        bl.diff.extend([0.0] * arr.size)                    # snapshots written with snapshot_diff carry these: ignored
    path = tmp_path / "v2.caffemodel"
    path.write_bytes(net.SerializeToString())
    got = proto.read_caffemodel(str(path))
    assert list(got) == ["conv1"]
    assert got["conv1"][0].shape == (6, 3, 3, 3) and np.array_equal(got["conv1"][0], w) and np.array_equal(got["conv1"][1], b)


def test_reads_v1_model_zoo_style_file(caffe_pb, tmp_path):
    """VGG_ILSVRC_16_layers.caffemodel style: `layers` (2), V1LayerParameter name = 4 / blobs = 6, legacy 4-d dims, the bias
    stored as 1 x 1 x 1 x N and an inner-product weight as 1 x 1 x N x K."""
    rng = np.random.default_rng(1)
    w, b, fc = _rand(rng, 8, 3, 3, 3), _rand(rng, 8), _rand(rng, 5, 72)
    net = caffe_pb["NetParameter"]()
    net.name = "VGG_ILSVRC_16_layers"
    net.input.append("data")
    net.input_dim.extend([10, 3, 224, 224])
    conv = net.layers.add()
    conv.name, conv.type = "conv1_1", 4
    conv.bottom.append("data")
    conv.top.append("conv1_1")
    conv.blobs_lr.extend([1, 2])
    for arr, dims in ((w, w.shape), (b, (1, 1, 1, 8))):
        bl = conv.blobs.add()
        bl.num, bl.channels, bl.height, bl.width = dims
        bl.data.extend(arr.ravel().tolist())
    relu = net.layers.add()
    relu.name, relu.type = "relu1_1", 18
    ip = net.layers.add()
    ip.name, ip.type = "fc8", 14
    bl = ip.blobs.add()
    bl.num, bl.channels, bl.height, bl.width = 1, 1, 5, 72
    bl.data.extend(fc.ravel().tolist())
    path = tmp_path / "v1.caffemodel"
    path.write_bytes(net.SerializeToString())
    got = proto.read_caffemodel(str(path))
    assert list(got) == ["conv1_1", "fc8"]
    assert got["conv1_1"][0].shape == (8, 3, 3, 3) and np.array_equal(got["conv1_1"][0], w)
    assert got["conv1_1"][1].shape == (1, 1, 1, 8) and np.array_equal(got["conv1_1"][1].ravel(), b)
    assert got["fc8"][0].shape == (1, 1, 5, 72) and np.array_equal(got["fc8"][0].reshape(5, 72), fc)


def test_reads_double_raw_and_unpacked_blobs(caffe_pb, tmp_path):
    rng = np.random.default_rng(2)
    a, b, c, d = _rand(rng, 4, 2, 1, 1), _rand(rng, 4), _rand(rng, 3, 3), _rand(rng, 7)
    net = caffe_pb["NetParameter"]()
    lay = net.layer.add()
    lay.name, lay.type = "dbl", "Convolution"
    bl = lay.blobs.add()                                    # a double-precision Caffe build: double_data
    bl.shape.dim.extend(a.shape)
    bl.double_data.extend(a.astype(np.float64).ravel().tolist())
    bl = lay.blobs.add()                                    # NVCaffe raw storage, float
    bl.shape.dim.extend(b.shape)
    bl.raw_data_type, bl.raw_data = 1, b.astype("<f4").tobytes()
    lay2 = net.layer.add()
    lay2.name, lay2.type = "half", "InnerProduct"
    bl = lay2.blobs.add()                                   # NVCaffe raw storage, FLOAT16
    bl.shape.dim.extend(c.shape)
    bl.raw_data_type, bl.raw_data = 2, c.astype("<f2").tobytes()
    raw = net.SerializeToString()
    # an un-packed `data` field, appended by hand as a third layer: tag (100, LEN) around tag (7, LEN) around 7 x (5, I32)
    ub = caffe_pb["BlobProtoUnpacked"]()
    ub.shape.dim.extend(d.shape)
    ub.data.extend(d.tolist())
    ub_bytes = ub.SerializeToString()
    assert ub_bytes.count(struct.pack("<B", (5 << 3) | 5)) >= 7      # really one key per element
    lay3 = proto._ld(1, b"loose") + proto._ld(2, b"Bias") + proto._ld(7, ub_bytes)
    path = tmp_path / "mixed.caffemodel"
    path.write_bytes(raw + proto._ld(100, lay3))
    got = proto.read_caffemodel(str(path))
    assert np.array_equal(got["dbl"][0], a) and got["dbl"][0].shape == a.shape
    assert np.array_equal(got["dbl"][1], b)
    assert np.array_equal(got["half"][0], c.astype(np.float16).astype(np.float32))
    assert np.array_equal(got["loose"][0], d)


def test_written_caffemodel_parses_with_protobuf_runtime(caffe_pb, tmp_path):
    rng = np.random.default_rng(3)
    w, b = _rand(rng, 4, 3, 5, 5), _rand(rng, 4)
    path = tmp_path / "out.caffemodel"
    proto.write_caffemodel(str(path), [("conv", "Convolution", [w, b]), ("up", "Deconvolution", [_rand(rng, 2, 1, 4, 4)])], net_name="n")
    net = caffe_pb["NetParameter"]()
    net.ParseFromString(path.read_bytes())
    assert net.name == "n" and [l.name for l in net.layer] == ["conv", "up"] and [l.type for l in net.layer] == ["Convolution", "Deconvolution"]
    assert list(net.layer[0].blobs[0].shape.dim) == [4, 3, 5, 5]
    assert np.array_equal(np.array(net.layer[0].blobs[0].data, np.float32).reshape(4, 3, 5, 5), w)
    assert np.array_equal(np.array(net.layer[0].blobs[1].data, np.float32), b)
    assert len(net.layers) == 0


def test_solverstate_both_directions(caffe_pb):
    rng = np.random.default_rng(4)
    hist = [_rand(rng, 3, 2), _rand(rng, 5)]
    st = caffe_pb["SolverState"]()
    st.ParseFromString(proto.pack_solverstate(1234, hist, learned_net="snap_iter_1234.caffemodel"))
    assert st.iter == 1234 and st.learned_net == "snap_iter_1234.caffemodel" and len(st.history) == 2
    assert list(st.history[0].shape.dim) == [3, 2] and np.array_equal(np.array(st.history[1].data, np.float32), hist[1])
    st2 = caffe_pb["SolverState"]()
    st2.iter, st2.learned_net, st2.current_step = 77, "x.caffemodel", 3
    for h in hist:
        bl = st2.history.add()
        bl.shape.dim.extend(h.shape)
        bl.data.extend(h.ravel().tolist())
    it, got, learned = proto.unpack_solverstate(st2.SerializeToString(), with_learned_net=True)
    assert it == 77 and learned == "x.caffemodel" and all(np.array_equal(g, h) for g, h in zip(got, hist))


def test_copy_trained_layers_reports_what_it_did(caffe_pb, tmp_path):
    rng = np.random.default_rng(5)
    params = {"conv1_1": [np.zeros((8, 3, 3, 3), np.float32), np.zeros(8, np.float32)], "head": [np.zeros((2, 8, 1, 1), np.float32)]}
    net = caffe_pb["NetParameter"]()
    conv = net.layers.add()
    conv.name, conv.type = "conv1_1", 4
    w, b = _rand(rng, 8, 3, 3, 3), _rand(rng, 8)
    for arr, dims in ((w, w.shape), (b, (1, 1, 1, 8))):
        bl = conv.blobs.add()
        bl.num, bl.channels, bl.height, bl.width = dims
        bl.data.extend(arr.ravel().tolist())
    other = net.layers.add()
    other.name, other.type = "fc8", 14
    bl = other.blobs.add()
    bl.num, bl.channels, bl.height, bl.width = 1, 1, 2, 2
    bl.data.extend([1, 2, 3, 4])
    path = tmp_path / "zoo.caffemodel"
    path.write_bytes(net.SerializeToString())
    seen, log = {}, []
    copied = proto.copy_trained_layers(str(path), params, lambda k, v: seen.__setitem__(k, v), log=log.append)
    assert copied == ["conv1_1"] and np.array_equal(seen["conv1_1"][0], w) and seen["conv1_1"][1].shape == (8,)
    assert log == ["Ignoring source layer fc8"]
    # nothing matches: loud, not silent
    log.clear()
    assert proto.copy_trained_layers(str(path), {"unrelated": [np.zeros(3, np.float32)]}, lambda k, v: None, log=log.append) == []
    assert any("matched NONE" in m for m in log)
    # blob count / size mismatch and an unreadable file are errors
    with pytest.raises(ValueError, match="Incompatible number of blobs"):
        proto.copy_trained_layers(str(path), {"conv1_1": [np.zeros((8, 3, 3, 3), np.float32)]}, lambda k, v: None, log=log.append)
    with pytest.raises(ValueError, match="Cannot copy param"):
        proto.copy_trained_layers(str(path), {"conv1_1": [np.zeros((8, 3, 3, 3), np.float32), np.zeros(9, np.float32)]}, lambda k, v: None,
                                  log=log.append)
    empty = tmp_path / "empty.caffemodel"
    empty.write_bytes(proto._ld(1, b"just a name"))
    with pytest.raises(ValueError, match="no layer with parameter blobs"):
        proto.copy_trained_layers(str(empty), params, lambda k, v: None, log=log.append)
