"""f16 activations / weights with f32 accumulation (BASELINE configs[4]: v_mfma_f32_32x32x16_f16 path), kernel level (-m gpu).

Tolerance: the inputs of the oracle are the SAME f16-rounded values, products of halves are exact in f32 and the
accumulation is f32 on both sides, so with FCN_CONV_OUT_F32 only the summation order differs (1e-5); with an f16 output
the final rounding adds half an f16 ulp (2^-11 = 4.9e-4 relative): 1e-3."""
import ctypes as C

import numpy as np
import pytest

from conftest import CFG_STREAM0, rel_err
from fcn_object_detector_amd import lib as L
from fcn_object_detector_amd.engine import DeviceBuffer
from gpu_util import conv_desc, dev_from, dev_to
from oracle import caffe_ref as R

pytestmark = pytest.mark.gpu

CASES = [  # cin, cout, k, stride, pad, h, w, n
    (3, 64, 7, 2, 3, 61, 45, 1),        # conv1: Cin 3 padded to 8
    (64, 192, 3, 1, 1, 23, 19, 2),
    (16, 32, 5, 1, 2, 17, 28, 1),
    (24, 64, 5, 1, 2, 14, 14, 2),       # taps straddle chunk boundaries
    (192, 48, 1, 1, 0, 9, 11, 1),
    (112, 33, 3, 1, 1, 12, 7, 1),
    (832, 384, 1, 1, 0, 7, 7, 1),
    (1024, 4, 1, 1, 0, 28, 28, 1),
]


def r8(c):
    return (c + 7) // 8 * 8


def f16_conv(x, wt, b, pad, stride, flags, out_f32, y_cstride=None, y_coffset=0):
    n, cin, h, w = x.shape
    cout, _, k, _ = wt.shape
    ci8 = r8(cin)
    oh, ow = R.conv_out(h, k, pad, stride), R.conv_out(w, k, pad, stride)
    xh = np.zeros((n, h, w, ci8), np.float16)
    xh[..., :cin] = x.transpose(0, 2, 3, 1)
    wh = np.zeros((cout, k, k, ci8), np.float16)
    wh[..., :cin] = wt.transpose(0, 2, 3, 1)
    ycs = y_cstride or r8(cout)
    yd = dev_from(np.full((n, oh, ow, ycs), -7.0, np.float32 if out_f32 else np.float16))
    xd, wd, bd = dev_from(xh), dev_from(wh), dev_from(b)
    d = conv_desc(xd, wd, bd, yd, n, h, w, ci8, ci8, cout, k, pad, stride, oh, ow, ycs, y_coffset,
                  flags | L.CONV_F16 | (L.CONV_OUT_F32 if out_f32 else 0))
    L.call("fcn_conv2d_fwd_f32", C.byref(d), None)
    y = dev_to(yd, (n, oh, ow, ycs), np.float32 if out_f32 else np.float16)
    return y


# conv_stream_f16 (configurations CFG_STREAM0 ..): persistent workgroups, 256-pixel tiles x 128 / 64 channels, 128 x 192
# (slab rows, slab buffers, pixels per tile) per configuration, in order: +0/+1 take 3x3 and 5x5 launches, +2/+3 1x1 launches, +4/+5 mixed 1x1 + 3x3,
# +6 all three, +7 .. +10 the 128-pixel tiles
_STREAM_SHAPES = [(304, 2, 256), (304, 2, 256), (256, 3, 256), (256, 4, 256), (288, 3, 256), (288, 3, 256), (304, 3, 256),
                  (160, 2, 128), (160, 3, 128), (160, 3, 128), (160, 4, 128)]
STREAM_CFGS = [CFG_STREAM0 + i for i in range(len(_STREAM_SHAPES))]
_STREAM_SHAPE = dict(zip(STREAM_CFGS, _STREAM_SHAPES))


@pytest.mark.parametrize("cfg", [None, "2", "5", "8", "10", "13", "14", "15", "23", "24", "26", "29"])
@pytest.mark.parametrize("case", CASES)
def test_f16_conv_matches_oracle_on_the_same_rounded_inputs(gpu, monkeypatch, case, cfg):
    if cfg is None:
        monkeypatch.delenv("FCN_CONV_CFG", raising=False)
    else:
        monkeypatch.setenv("FCN_CONV_CFG", cfg)
    cin, cout, k, s, p, h, w, n = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float16).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = np.maximum(R.conv2d(x, wt, b, p, s), 0)
    y32 = f16_conv(x, wt, b, p, s, L.CONV_RELU, True)
    assert rel_err(y32[..., :cout].transpose(0, 3, 1, 2), ref) < 1e-5
    assert np.all(y32[..., cout:] == -7.0)
    y16 = f16_conv(x, wt, b, p, s, L.CONV_RELU, False, r8(cout) + 8, 8)
    got = y16[..., 8:8 + cout].astype(np.float32).transpose(0, 3, 1, 2)
    assert rel_err(got, ref) < 1e-3
    assert np.array_equal(got, ref.astype(np.float16).astype(np.float32)) or np.abs(got - ref).max() <= np.abs(ref).max() * 2.0 ** -10
    assert np.all(y16[..., :8] == np.float16(-7.0)) and np.all(y16[..., 8 + cout:] == np.float16(-7.0))


def _stream_problem(rng, cin, cout, k, pad, h, w, n, relu=True, bias=True, y_cstride=None, y_coffset=0, xd=None, x=None):
    """One half-float problem for the streaming kernel: device operands, descriptor, reference output (f32 of the rounded operands)."""
    ci8 = r8(cin)
    if x is None:
        x = rng.standard_normal((n, cin, h, w)).astype(np.float16).astype(np.float32)
        xh = np.zeros((n, h, w, ci8), np.float16)
        xh[..., :cin] = x.transpose(0, 2, 3, 1)
        xd = dev_from(xh)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if bias else None
    wh = np.zeros((cout, k, k, ci8), np.float16)
    wh[..., :cin] = wt.transpose(0, 2, 3, 1)
    oh, ow = R.conv_out(h, k, pad, 1), R.conv_out(w, k, pad, 1)
    ycs = y_cstride or cout
    yd = dev_from(np.full((n, oh, ow, ycs), -7.0, np.float16))
    wd, bd = dev_from(wh), (dev_from(b) if bias else None)
    d = conv_desc(xd, wd, bd, yd, n, h, w, ci8, ci8, cout, k, pad, 1, oh, ow, ycs, y_coffset, L.CONV_F16 | (L.CONV_RELU if relu else 0))
    ref = R.conv2d(x, wt, b if bias else np.zeros(cout, np.float32), pad, 1)
    if relu:
        ref = np.maximum(ref, 0)
    return dict(desc=d, keep=[xd, wd, bd, yd], yd=yd, ref=ref, shape=(n, oh, ow, ycs), cout=cout, coff=y_coffset, x=x, xd=xd)


def _stream_takes(cfg, k, w):
    """A tile's 256 (128) pixels in padded raster order, plus the taps of a filter row, must fit the configuration's slab; 1x1 filters
    need three slab buffers; padded image rows hold at least 16 entries (conv_fwd.hip plan_tiles_cfg)."""
    rows_max, bufs, bm = _STREAM_SHAPE[cfg]
    pad = (k - 1) // 2
    rows = bm - 1 + 2 * pad * ((bm - 2 + w) // w) + 2 * pad + 1
    return rows <= rows_max and (k > 1 or bufs >= 3) and w + 2 * pad >= 16


def _run_stream_group(probs, cfg):
    lib = L.load()
    arr = (L.ConvDesc * len(probs))(*[q["desc"] for q in probs])
    ws = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(len(probs))), zero=False)
    grp = L.ConvGroup()
    L.call("fcn_conv2d_group_prepare", arr, len(probs), ws.ptr, cfg, C.byref(grp))
    assert grp.cfg == cfg
    L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
    L.call("fcn_device_sync")
    for q in probs:
        yfull = dev_to(q["yd"], q["shape"], np.float16)
        got = yfull[..., q["coff"]:q["coff"] + q["cout"]].astype(np.float32).transpose(0, 3, 1, 2)
        assert rel_err(got, q["ref"]) < 1e-3
        assert np.abs(got - q["ref"]).max() <= max(np.abs(q["ref"]).max(), 1.0) * 2.0 ** -10      # half an f16 ulp of the largest value
        assert np.all(yfull[..., :q["coff"]] == np.float16(-7.0)) and np.all(yfull[..., q["coff"] + q["cout"]:] == np.float16(-7.0))
    L.call("fcn_conv2d_group_release", ws.ptr)
    return grp


@pytest.mark.parametrize("cfg", STREAM_CFGS)
@pytest.mark.parametrize("case", [  # cin, cout, k, pad, h, w, n
    (96, 208, 3, 1, 28, 28, 3),       # several row tiles with a ragged last one, two column tiles, a 32-channel tail chunk (96 = 64 + 32)
    (480, 304, 1, 0, 28, 28, 2),      # a long 1x1 walk (8 chunks, the last one 32 channels), three column tiles with a ragged last one
    (48, 64, 5, 2, 28, 28, 2),        # 5x5 on 48 channels: five filter rows of one short chunk, ten image rows per tile
    (16, 32, 5, 2, 20, 40, 1),        # 5x5 on 16 channels, a single ragged tile, image rows of 40
    (64, 192, 3, 1, 40, 36, 5),       # more tiles than a small grid would hold per workgroup: the persistent walk, tiles across images
    (192, 16, 1, 0, 6, 17, 1),        # M = 102 < one tile, Cout = 16
    (192, 16, 1, 0, 9, 11, 1),        # image rows of 11: refused
    (64, 64, 1, 0, 33, 31, 2),        # one chunk per tile (conv2/3x3_reduce): every chunk is a tile's first and last
    (128, 256, 3, 1, 56, 56, 1),      # image rows of 56: a tile covers 4.6 of them
    (32, 96, 5, 2, 28, 28, 3),        # packed taps: two taps of a 32-channel blob per chunk (three chunks per filter row, the last one half empty)
    (16, 48, 5, 2, 28, 28, 3),        # packed taps: four taps of a 16-channel blob per chunk (4 + 1), several tiles across images
    (32, 64, 3, 1, 19, 23, 2),        # packed taps on a 3x3 filter (2 + 1 taps), odd image extents
    (16, 24, 3, 1, 28, 28, 1),        # 3x3 on 16 channels would be ONE chunk per slab: stays unpacked (plan_tiles_cfg)
])
def test_stream_kernel_matches_oracle(gpu, case, cfg):
    cin, cout, k, pad, h, w, n = case
    rng = np.random.default_rng(hash(case) % 2**32)
    if not _stream_takes(cfg, k, w):
        q = _stream_problem(rng, cin, cout, k, pad, h, w, n)
        ws = DeviceBuffer(int(L.load().fcn_conv2d_group_workspace_bytes(1)), zero=False)
        assert L.load().fcn_conv2d_group_prepare((L.ConvDesc * 1)(q["desc"]), 1, ws.ptr, cfg, C.byref(L.ConvGroup())) != 0
        return
    grp = _run_stream_group([_stream_problem(rng, cin, cout, k, pad, h, w, n)], cfg)
    assert grp.total_tiles >= 1


@pytest.mark.parametrize("cfg", STREAM_CFGS)
def test_stream_kernel_group_slices_and_flags(gpu, cfg):
    """Several problems in one persistent launch: an inception module's three 1x1 convolutions on one input (one without bias, one
    without ReLU), its 3x3 + 5x5 level, and a 3x3 beside a 1x1 - different K and filter sizes per problem, outputs as channel slices
    of a wider buffer."""
    rng = np.random.default_rng(17)
    n, h, w = 3, 28, 28
    ran = 0
    if _stream_takes(cfg, 1, w):
        first = _stream_problem(rng, 192, 64, 1, 0, h, w, n, y_cstride=256, y_coffset=0)
        _run_stream_group([first, _stream_problem(rng, 192, 96, 1, 0, h, w, n, bias=False, x=first["x"], xd=first["xd"]),
                           _stream_problem(rng, 192, 16, 1, 0, h, w, n, relu=False, x=first["x"], xd=first["xd"])], cfg)
        ran += 1
    if _stream_takes(cfg, 3, w) and _stream_takes(cfg, 5, w):
        grp = _run_stream_group([_stream_problem(rng, 96, 128, 3, 1, h, w, n, y_cstride=256, y_coffset=64),
                                 _stream_problem(rng, 16, 32, 5, 2, h, w, n, y_cstride=256, y_coffset=192)], cfg)
        assert grp.n == 2
        ran += 1
    if _stream_takes(cfg, 3, w) and _stream_takes(cfg, 1, w):
        _run_stream_group([_stream_problem(rng, 96, 128, 3, 1, h, w, n, y_cstride=256, y_coffset=64),
                           _stream_problem(rng, 192, 32, 1, 0, h, w, n, y_cstride=256, y_coffset=224)], cfg)
        ran += 1
    if _stream_takes(cfg, 1, w) and _stream_takes(cfg, 3, w) and _stream_takes(cfg, 5, w):      # a whole inception level on 28-wide images
        grp = _run_stream_group([_stream_problem(rng, 96, 208, 3, 1, h, w, n, y_cstride=320, y_coffset=0),
                                 _stream_problem(rng, 16, 48, 5, 2, h, w, n, y_cstride=320, y_coffset=208),
                                 _stream_problem(rng, 480, 64, 1, 0, h, w, n, y_cstride=320, y_coffset=256)], cfg)
        assert grp.n == 3
        ran += 1
    assert ran >= 1


def test_stream_kernel_refuses_what_it_does_not_cover(gpu):
    rng = np.random.default_rng(3)
    lib = L.load()
    ws = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(1)), zero=False)
    grp = L.ConvGroup()
    q = _stream_problem(rng, 64, 33, 1, 0, 8, 8, 1, y_cstride=40)        # Cout not a multiple of 8
    assert lib.fcn_conv2d_group_prepare((L.ConvDesc * 1)(q["desc"]), 1, ws.ptr, CFG_STREAM0 + 2, C.byref(grp)) != 0
    q = _stream_problem(rng, 8, 64, 7, 3, 40, 40, 1)                      # 7x7 filters
    assert lib.fcn_conv2d_group_prepare((L.ConvDesc * 1)(q["desc"]), 1, ws.ptr, CFG_STREAM0, C.byref(grp)) != 0
    q = _stream_problem(rng, 64, 64, 3, 1, 20, 8, 1)                      # image rows of 8:  the padded slab of a tile does not fit
    assert lib.fcn_conv2d_group_prepare((L.ConvDesc * 1)(q["desc"]), 1, ws.ptr, CFG_STREAM0, C.byref(grp)) != 0
    q = _stream_problem(rng, 64, 64, 1, 0, 8, 8, 1)                       # 1x1 filters need three slab buffers
    assert lib.fcn_conv2d_group_prepare((L.ConvDesc * 1)(q["desc"]), 1, ws.ptr, CFG_STREAM0, C.byref(grp)) != 0
    q["desc"].flags |= L.CONV_OUT_F32                                     # float32 output (the detection heads)
    assert lib.fcn_conv2d_group_prepare((L.ConvDesc * 1)(q["desc"]), 1, ws.ptr, CFG_STREAM0 + 2, C.byref(grp)) != 0
    x = dev_from(np.zeros((1, 8, 8, 64), np.float32))                     # float32 problems never take it
    wt, yd = dev_from(np.zeros((64, 1, 1, 64), np.float32)), dev_from(np.zeros((1, 8, 8, 64), np.float32))
    d = conv_desc(x, wt, None, yd, 1, 8, 8, 64, 64, 64, 1, 0, 1, 8, 8, 64, 0, 0)
    assert lib.fcn_conv2d_group_prepare((L.ConvDesc * 1)(d), 1, ws.ptr, CFG_STREAM0 + 2, C.byref(grp)) != 0


def test_f16_group_with_fused_pool_and_sigmoid_head(gpu):
    """An f16 group launch: two convolutions + a fused MAX pooling (8 channels per 16-byte item) + the f32 sigmoid output."""
    rng = np.random.default_rng(4)
    n, h, w, cin = 2, 12, 9, 32
    x = rng.standard_normal((n, cin, h, w)).astype(np.float16).astype(np.float32)
    xh = np.ascontiguousarray(x.transpose(0, 2, 3, 1)).astype(np.float16)
    xd = dev_from(xh)
    descs, keep, refs = [], [], []
    for cout, k, pad, sig in ((40, 1, 0, False), (6, 3, 1, True)):
        wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float16).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        wd, bd = dev_from(np.ascontiguousarray(wt.transpose(0, 2, 3, 1)).astype(np.float16)), dev_from(b)
        lin = R.conv2d(x, wt, b, pad, 1)
        if sig:
            yd = dev_from(np.zeros((n, h, w, 8), np.float32))
            y2d = dev_from(np.zeros((n, h, w, 8), np.float32))
            d = conv_desc(xd, wd, bd, yd, n, h, w, cin, cin, cout, k, pad, 1, h, w, 8, 0, L.CONV_F16 | L.CONV_OUT_F32 | L.CONV_SIGMOID2, 0.0,
                          y2d, 8, 0)
            refs.append((y2d, 8, cout, np.float32, R.sigmoid(lin), 1e-5))
        else:
            yd = dev_from(np.zeros((n, h, w, 40), np.float16))
            d = conv_desc(xd, wd, bd, yd, n, h, w, cin, cin, cout, k, pad, 1, h, w, 40, 0, L.CONV_F16 | L.CONV_RELU)
            refs.append((yd, 40, cout, np.float16, np.maximum(lin, 0), 1e-3))
        keep += [wd, bd, yd]
        descs.append(d)
    pool_ref = R.max_pool(x, 3, 1, 1)
    pd_ = dev_from(np.zeros((n, h, w, cin), np.float16))
    pdsc = L.PoolDesc()
    pdsc.x, pdsc.y, pdsc.idx = xd.ptr, pd_.ptr, None
    pdsc.N, pdsc.H, pdsc.W, pdsc.C, pdsc.x_cstride, pdsc.k, pdsc.stride, pdsc.pad = n, h, w, cin, cin, 3, 1, 1
    pdsc.OH, pdsc.OW, pdsc.y_cstride, pdsc.y_coffset, pdsc.f16 = h, w, cin, 0, 1
    arr, parr = (L.ConvDesc * 2)(*descs), (L.PoolDesc * 1)(pdsc)
    ws = DeviceBuffer(int(L.load().fcn_conv2d_group_workspace_bytes(2)), zero=False)
    grp = L.ConvGroup()
    L.call("fcn_conv2d_group_prepare_fused", arr, 2, parr, 1, ws.ptr, 5, C.byref(grp))
    L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
    for yd, cs, cout, dt, ref, tol in refs:
        got = dev_to(yd, (n, h, w, cs), dt)[..., :cout].astype(np.float32).transpose(0, 3, 1, 2)
        assert rel_err(got, ref) < tol
    assert np.array_equal(dev_to(pd_, (n, h, w, cin), np.float16).astype(np.float32).transpose(0, 3, 1, 2), pool_ref)
    # an f32 pooling cannot ride in an f16 group, nor an f16 problem beside an f32 one
    pdsc.f16 = 0
    assert L.load().fcn_conv2d_group_prepare_fused(arr, 2, (L.PoolDesc * 1)(pdsc), 1, ws.ptr, 5, C.byref(grp)) == 1
    descs[1].flags = L.CONV_RELU
    assert L.load().fcn_conv2d_group_prepare_fused((L.ConvDesc * 2)(*descs), 2, None, 0, ws.ptr, 5, C.byref(grp)) != 0


def _round16(a):
    return a.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("shape", [(2, 72, 28, 28), (1, 64, 56, 56), (3, 40, 30, 45), (1, 528, 28, 28), (2, 16, 37, 9)])
def test_f16_lds_pooling_3x3_s1(gpu, shape):
    """The LDS-staged 3x3 / stride 1 / pad 1 MAX pooling (the inception poolings at batch 32) against the oracle, bit for bit:
    NEGATIVE values included (pixels outside the image are replaced by the nearest pixel inside it, never by zeros), channel
    counts that are not multiples of 64, image extents that do not divide into tiles, output as a slice of a wider blob."""
    n, c, h, w = shape
    rng = np.random.default_rng(c * h + w)
    x = _round16(rng.standard_normal((n, c, h, w)) * 3 - 1.0)
    xh = np.ascontiguousarray(x.transpose(0, 2, 3, 1)).astype(np.float16)
    xd = dev_from(xh)
    ref = R.max_pool(x, 3, 1, 1)
    ycs, yco = c + 24, 16
    yd = dev_from(np.full((n, h, w, ycs), -3.0, np.float16))
    L.call("fcn_maxpool_fwd_f16", xd.ptr, yd.ptr, n, h, w, c, c, 3, 1, 1, h, w, ycs, yco, None)
    y = dev_to(yd, (n, h, w, ycs), np.float16)
    assert np.array_equal(y[..., yco:yco + c].astype(np.float32).transpose(0, 3, 1, 2), ref)
    assert np.all(y[..., :yco] == np.float16(-3.0)) and np.all(y[..., yco + c:] == np.float16(-3.0))


def test_f16_pointwise_kernels(gpu):
    rng = np.random.default_rng(6)
    x = _round16(rng.standard_normal((2, 24, 9, 7)) * 3)
    xh = np.zeros((2, 9, 7, 24), np.float16)
    xh[...] = x.transpose(0, 2, 3, 1)
    xd = dev_from(xh)
    for k, s, p in ((3, 2, 0), (3, 1, 1), (2, 2, 0)):
        ref = R.max_pool(x, k, s, p)
        oh, ow = ref.shape[2:]
        yd = dev_from(np.full((2, oh, ow, 32), -3.0, np.float16))
        L.call("fcn_maxpool_fwd_f16", xd.ptr, yd.ptr, 2, 9, 7, 24, 24, k, s, p, oh, ow, 32, 8, None)
        y = dev_to(yd, (2, oh, ow, 32), np.float16)
        assert np.array_equal(y[..., 8:].astype(np.float32).transpose(0, 3, 1, 2), ref)
        assert np.all(y[..., :8] == np.float16(-3.0))
    yd = dev_from(np.zeros((2, 9, 7, 24), np.float16))
    L.call("fcn_lrn_fwd_f16", xd.ptr, yd.ptr, 2 * 9 * 7, 24, 24, 24, 5, 1e-4, 0.75, 1.0, None)
    ref = R.lrn_across(x * 20, 5, 1e-4, 0.75, 1.0) / 20 if False else R.lrn_across(x, 5, 1e-4, 0.75, 1.0)
    assert rel_err(dev_to(yd, (2, 9, 7, 24), np.float16).astype(np.float32).transpose(0, 3, 1, 2), ref) < 1e-3
    # layout converters
    src = rng.standard_normal((2, 5, 6, 7)).astype(np.float32)
    sd, hd = dev_from(src), dev_from(np.zeros((2, 6, 7, 16), np.float16))
    L.call("fcn_nchw_f32_to_nhwc_f16", sd.ptr, hd.ptr, 2, 5, 6, 7, 16, 8, 0.5, None)
    h = dev_to(hd, (2, 6, 7, 16), np.float16)
    assert np.array_equal(h[..., 8:13], (src + np.float32(0.5)).astype(np.float16).transpose(0, 2, 3, 1))
    back = dev_from(np.zeros((2, 5, 6, 7), np.float32))
    L.call("fcn_nhwc_f16_to_nchw_f32", hd.ptr, back.ptr, 2, 5, 6, 7, 16, 8, None)
    assert np.array_equal(dev_to(back, (2, 5, 6, 7)), h[..., 8:13].astype(np.float32).transpose(0, 3, 1, 2))


def test_f16_engine_forward_of_the_detectnet_deploy_net(gpu):
    """Engine(dtype="f16") on models/deploy.prototxt's graph at reduced size against (a) the f32 oracle, at fp16 accuracy,
    and (b) an oracle run that rounds weights and activations to halves at the same points, tightly."""
    from fcn_object_detector_amd import models, proto
    from fcn_object_detector_amd.engine import Engine
    from fcn_object_detector_amd.netspec import NetSpec, fill_params
    from oracle.net_ref import RefNet
    msg = proto.parse_text(models.googlenet_detectnet_deploy(2, 96, 128, 3))
    spec = NetSpec(msg, "TEST")
    spec.infer()
    params = fill_params(spec, seed=21)
    eng = Engine(NetSpec(msg, "TEST"), params={k: [a.copy() for a in v] for k, v in params.items()}, device=0, autotune=False, dtype="f16")
    # the image is half too: un-shifted pixels + two constant-1 channels that carry the folded Power(-127) (engine._half_inputs)
    assert eng.blobs["data"].esize == 2 and eng.blobs["data"].cstride == 8 and eng._half_inputs == {"data": ("transformed_data", -127.0)}
    assert eng.blobs["conv1/7x7_s2"].esize == 2 and eng.blobs["inception_3a/output"].esize == 2
    assert eng.blobs["coverage"].esize == 4 and eng.blobs["bboxes"].esize == 4 and eng.blobs["cvg/classifier"].esize == 4
    x = np.random.default_rng(1).random((2, 3, 96, 128), dtype=np.float32)
    eng.host_array("data")[...] = x
    out = eng.forward()
    ref = RefNet(msg, "TEST", params)
    ref.blobs["data"] = x
    rb = ref.forward()
    for name in ("coverage", "bboxes"):
        assert rel_err(out[name], rb[name]) < 2e-2, name                      # fp16 storage of 60+ layers vs pure f32
    # (b) same rounding points: the weights of every layer are halves, the image and every internal activation are rounded to
    # half (the shifted image is not: the shift lives in the first layer's filters, exact to 2^-22)
    p16 = {k: [_round16(v[0])] + [a.copy() for a in v[1:]] for k, v in params.items()}
    ref16 = RefNet(msg, "TEST", p16)
    ref16.blobs["data"] = _round16(x)
    ref16.round_activations = lambda name, a: a if name in ("data", "transformed_data", "coverage", "bboxes", "cvg/classifier") else _round16(a)
    rb16 = ref16.forward()
    for name in ("coverage", "bboxes"):
        assert rel_err(out[name], rb16[name]) < 3e-3, name
    assert rel_err(eng.read_blob("inception_4a/output"), rb16["inception_4a/output"]) < 3e-3
    # reading the Power top back adds the shift that the device copy no longer carries
    assert np.abs(eng.read_blob("transformed_data") - (x + np.float32(-127.0))).max() <= 2.0 ** -11 + 1e-5
    # the first layer alone, against float64 on the same half operands: the folded shift is exact to ~1e-5 of the bias-sized term
    w16 = _round16(params["conv1/7x7_s2"][0]).astype(np.float64)
    from oracle import caffe_ref as R
    want = R.relu(R.conv2d((_round16(x).astype(np.float64) - 127.0), w16, params["conv1/7x7_s2"][1].astype(np.float64), 3, 2))
    got = eng.read_blob("conv1/7x7_s2")
    assert np.abs(got - want).max() <= np.abs(want).max() * 2.0 ** -10
    # FCN_F16_IMAGE=0 keeps the float32 image and first layer
    import os
    os.environ["FCN_F16_IMAGE"] = "0"
    try:
        eng32 = Engine(NetSpec(msg, "TEST"), params={k: [a.copy() for a in v] for k, v in params.items()}, device=0, autotune=False, dtype="f16")
        assert eng32.blobs["data"].esize == 4 and not eng32._half_inputs
        eng32.host_array("data")[...] = x
        o32 = eng32.forward()
        for name in ("coverage", "bboxes"):
            assert rel_err(o32[name], rb[name]) < 2e-2
        eng32.close()
    finally:
        del os.environ["FCN_F16_IMAGE"]
    eng.close()


@pytest.mark.parametrize("n,h,w,cout,relu", [(2, 448, 448, 64, 1), (5, 100, 130, 48, 0), (1, 64, 64, 64, 1), (11, 40, 200, 40, 1)])
def test_f16_first_layer_kernel_matches_oracle(gpu, monkeypatch, n, h, w, cout, relu):
    """conv_first7_f16_kernel (configuration 30 on half-float problems: 7x7 / stride 2 / pad 3 on 8-half pixels), chosen by the
    built-in heuristic: more tiles than workgroups (the persistent loop, both patch buffers), partial tiles, fewer than 64 channels."""
    monkeypatch.delenv("FCN_CONV_CFG", raising=False)
    rng = np.random.default_rng(77)
    x = rng.standard_normal((n, 3, h, w)).astype(np.float16).astype(np.float32)
    wt = (rng.standard_normal((cout, 3, 7, 7)) / np.sqrt(147)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = R.conv2d(x, wt, b, 3, 2)
    if relu:
        ref = np.maximum(ref, 0)
    y16 = f16_conv(x, wt, b, 3, 2, L.CONV_RELU if relu else 0, False, r8(cout) + 8, 8)
    got = y16[..., 8:8 + cout].astype(np.float32).transpose(0, 3, 1, 2)
    assert rel_err(got, ref) < 1e-3
    assert np.abs(got - ref).max() <= np.abs(ref).max() * 2.0 ** -10
    assert np.all(y16[..., :8] == np.float16(-7.0)) and np.all(y16[..., 8 + cout:] == np.float16(-7.0))
    # the group interface prepares the same configuration for this problem
    lib = L.load()
    xd = dev_from(np.zeros((n, h, w, 8), np.float16))
    wd = dev_from(np.zeros((cout, 7, 7, 8), np.float16))
    oh, ow = R.conv_out(h, 7, 3, 2), R.conv_out(w, 7, 3, 2)
    yd = dev_from(np.zeros((n, oh, ow, r8(cout)), np.float16))
    d = conv_desc(xd, wd, None, yd, n, h, w, 8, 8, cout, 7, 3, 2, oh, ow, r8(cout), 0, L.CONV_F16)
    ws = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(1)), zero=False)
    grp = L.ConvGroup()
    L.call("fcn_conv2d_group_prepare", (L.ConvDesc * 1)(d), 1, ws.ptr, -1, C.byref(grp))
    assert grp.cfg == lib.fcn_conv2d_first_layer_config()
    L.call("fcn_conv2d_group_release", ws.ptr)


@pytest.mark.parametrize("n,h,w,cout,relu", [(2, 448, 448, 64, 1), (5, 100, 130, 48, 0), (1, 64, 64, 64, 1), (11, 40, 201, 40, 1), (3, 13, 9, 64, 1)])
def test_f16_first_layer_with_constant_channels(gpu, monkeypatch, n, h, w, cout, relu):
    """FCN_CONV_IMAGE_ONES (conv_first7_f16x4_kernel): an 8-half pixel image whose channels 3 and 4 are the constant 1 - the folded Power
    shift of the f16 engine.  The kernel multiplies b, g, r only and adds the two constant channels' filters as per-tap constants,
    for every pixel over exactly the taps that lie inside the image (odd extents, images smaller than a tile, more tiles than
    workgroups).  Against float64 convolution of the same half operands, and against the 8-half kernel (FCN_FIRST7_X4=0 is read once
    per process, so the comparison goes through the flag)."""
    monkeypatch.delenv("FCN_CONV_CFG", raising=False)
    rng = np.random.default_rng(n * 1000 + h)
    x = np.zeros((n, 8, h, w), np.float32)
    x[:, :3] = rng.random((n, 3, h, w)).astype(np.float16)
    x[:, 3:5] = 1.0
    wt = np.zeros((cout, 8, 7, 7), np.float32)
    wt[:, :3] = (rng.standard_normal((cout, 3, 7, 7)) / np.sqrt(147)).astype(np.float16)
    term = -127.0 * wt[:, :3].astype(np.float64).sum(1)                      # what Engine._packed_weight folds: hi + lo halves
    wt[:, 3] = term.astype(np.float16)
    wt[:, 4] = (term - wt[:, 3].astype(np.float64)).astype(np.float16)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = R.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 3, 2)
    if relu:
        ref = np.maximum(ref, 0)
    oh, ow = R.conv_out(h, 7, 3, 2), R.conv_out(w, 7, 3, 2)
    xd = dev_from(np.ascontiguousarray(x.transpose(0, 2, 3, 1)).astype(np.float16))
    wd = dev_from(np.ascontiguousarray(wt.transpose(0, 2, 3, 1)).astype(np.float16))
    bd = dev_from(b)
    outs = []
    for flag in (L.CONV_IMAGE_ONES, 0):
        yd = dev_from(np.full((n, oh, ow, r8(cout) + 8), -7.0, np.float16))
        d = conv_desc(xd, wd, bd, yd, n, h, w, 8, 8, cout, 7, 3, 2, oh, ow, r8(cout) + 8, 8, L.CONV_F16 | flag | (L.CONV_RELU if relu else 0))
        L.call("fcn_conv2d_fwd_f32", C.byref(d), None)
        L.call("fcn_device_sync")
        y16 = dev_to(yd, (n, oh, ow, r8(cout) + 8), np.float16)
        got = y16[..., 8:8 + cout].astype(np.float64).transpose(0, 3, 1, 2)
        scale = max(np.abs(ref).max(), 1.0)
        assert np.abs(got - ref).max() <= scale * 2.0 ** -10, flag              # half an f16 ulp of the largest value
        assert np.all(y16[..., :8] == np.float16(-7.0)) and np.all(y16[..., 8 + cout:] == np.float16(-7.0))
        outs.append(got)
    assert np.abs(outs[0] - outs[1]).max() <= max(np.abs(ref).max(), 1.0) * 2.0 ** -10
    # the flag is refused where it cannot hold
    bad = conv_desc(xd, wd, bd, yd, n, h, w, 8, 8, cout, 7, 3, 2, oh, ow, r8(cout) + 8, 8, L.CONV_IMAGE_ONES)
    assert L.load().fcn_conv2d_fwd_f32(C.byref(bad), None) != 0


@pytest.mark.parametrize("lrn_first", [0, 1])
@pytest.mark.parametrize("k,s,p,h,w,c", [(3, 2, 0, 28, 28, 64), (3, 2, 0, 15, 21, 8), (3, 1, 1, 9, 7, 16), (3, 2, 1, 10, 11, 40), (3, 2, 0, 17, 9, 192)])
def test_f16_maxpool_lrn_single_pass_equals_the_two_launches(gpu, lrn_first, k, s, p, h, w, c):
    """fcn_maxpool_lrn5_fwd_f16 against fcn_maxpool_fwd_f16 + fcn_lrn_fwd_f16 in the same order: bit for bit (the maximum of
    halves is exact, every normalised value is rounded to a half before it is compared), and against the oracle at half precision."""
    rng = np.random.default_rng(17)
    x = (rng.standard_normal((2, c, h, w)) * 30).astype(np.float16)
    oh, ow = R.pool_out(h, k, p, s), R.pool_out(w, k, p, s)
    xd = dev_from(np.ascontiguousarray(x.transpose(0, 2, 3, 1)))
    yd = dev_from(np.zeros((2, oh, ow, c), np.float16))
    L.call("fcn_maxpool_lrn5_fwd_f16", xd.ptr, yd.ptr, 2, h, w, c, c, k, s, p, oh, ow, c, lrn_first, 1e-4, 0.75, 1.0, None)
    y = dev_to(yd, (2, oh, ow, c), np.float16)
    mh, mw = (h, w) if lrn_first else (oh, ow)
    md = dev_from(np.zeros((2, mh, mw, c), np.float16))
    zd = dev_from(np.zeros((2, oh, ow, c), np.float16))
    if lrn_first:
        L.call("fcn_lrn_fwd_f16", xd.ptr, md.ptr, 2 * h * w, c, c, c, 5, 1e-4, 0.75, 1.0, None)
        L.call("fcn_maxpool_fwd_f16", md.ptr, zd.ptr, 2, h, w, c, c, k, s, p, oh, ow, c, 0, None)
    else:
        L.call("fcn_maxpool_fwd_f16", xd.ptr, md.ptr, 2, h, w, c, c, k, s, p, oh, ow, c, 0, None)
        L.call("fcn_lrn_fwd_f16", md.ptr, zd.ptr, 2 * oh * ow, c, c, c, 5, 1e-4, 0.75, 1.0, None)
    assert np.array_equal(y, dev_to(zd, (2, oh, ow, c), np.float16))
    x32 = x.astype(np.float32)
    ref = R.max_pool(R.lrn_across(x32, 5, 1e-4, 0.75, 1.0), k, s, p) if lrn_first else R.lrn_across(R.max_pool(x32, k, s, p), 5, 1e-4, 0.75, 1.0)
    assert rel_err(y.astype(np.float32).transpose(0, 3, 1, 2), ref) < 1e-3


@pytest.mark.parametrize("lrn_first", [0, 1])
@pytest.mark.parametrize("n,h,w,c", [(8, 112, 112, 64), (8, 57, 61, 192), (16, 45, 52, 96)])
def test_f16_pool_lrn_lds_patch_kernel(gpu, lrn_first, n, h, w, c):
    """The LDS-patch form of the fused 3x3 / stride 2 pooling + LRN (what large half-float blobs take: pool1 -> norm1 and
    norm2 -> pool2 at batch 32) against the two stand-alone launches, bit for bit - ceil-mode windows that hang over the image
    edge included (odd extents), negative values included."""
    rng = np.random.default_rng(n * h + c)
    x = (rng.standard_normal((n, h, w, c)) * 30 - 5).astype(np.float16)
    oh, ow = R.pool_out(h, 3, 0, 2), R.pool_out(w, 3, 0, 2)
    xd = dev_from(x)
    yd = dev_from(np.zeros((n, oh, ow, c), np.float16))
    L.call("fcn_maxpool_lrn5_fwd_f16", xd.ptr, yd.ptr, n, h, w, c, c, 3, 2, 0, oh, ow, c, lrn_first, 1e-4, 0.75, 1.0, None)
    mh, mw = (h, w) if lrn_first else (oh, ow)
    md = dev_from(np.zeros((n, mh, mw, c), np.float16))
    zd = dev_from(np.zeros((n, oh, ow, c), np.float16))
    if lrn_first:
        L.call("fcn_lrn_fwd_f16", xd.ptr, md.ptr, n * h * w, c, c, c, 5, 1e-4, 0.75, 1.0, None)
        L.call("fcn_maxpool_fwd_f16", md.ptr, zd.ptr, n, h, w, c, c, 3, 2, 0, oh, ow, c, 0, None)
    else:
        L.call("fcn_maxpool_fwd_f16", xd.ptr, md.ptr, n, h, w, c, c, 3, 2, 0, oh, ow, c, 0, None)
        L.call("fcn_lrn_fwd_f16", md.ptr, zd.ptr, n * oh * ow, c, c, c, 5, 1e-4, 0.75, 1.0, None)
    a, b = dev_to(yd, (n, oh, ow, c), np.float16), dev_to(zd, (n, oh, ow, c), np.float16)
    # Equal bit for bit except where a normalised value's float32 sits within ~1e-7 of the half-way point between two halves:
    # the two kernels round the last float32 bit of s^-0.75 x differently there (measured: 18 of 878 592 elements, one f16 ulp each,
    # the single-pass value being the correctly rounded one against float64).  Bound: < 1e-4 of the elements, one ulp.
    diff = a != b
    assert diff.mean() < 1e-4
    if diff.any():
        a32, b32 = a[diff].astype(np.float32), b[diff].astype(np.float32)
        assert np.all(np.abs(a32 - b32) <= np.spacing(np.maximum(np.abs(a), np.abs(b))[diff]).astype(np.float32))
    x32 = x.astype(np.float32).transpose(0, 3, 1, 2)
    ref = R.max_pool(R.lrn_across(x32, 5, 1e-4, 0.75, 1.0), 3, 2, 0) if lrn_first else R.lrn_across(R.max_pool(x32, 3, 2, 0), 5, 1e-4, 0.75, 1.0)
    assert rel_err(a.astype(np.float32).transpose(0, 3, 1, 2), ref) < 1e-3


@pytest.mark.parametrize("n,h,w,relu,ycs,yco", [(8, 224, 224, 1, 64, 0), (3, 57, 61, 1, 64, 0), (2, 45, 52, 0, 96, 32)])
def test_f16_pool_lrn_conv1x1_single_pass(gpu, n, h, w, relu, ycs, yco):
    """fcn_maxpool_lrn5_conv1x1_fwd_f16 (pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce + ReLU at batch 32 as one launch): against
    the library's own pool + LRN launch (whose halves are the convolution's exact inputs) followed by an f32 matrix product of the same
    halves - only the summation order and the one final rounding differ - and against the oracle's three layers."""
    rng = np.random.default_rng(n * h + w)
    c = co = 64
    x = np.maximum(rng.standard_normal((n, h, w, c)) * 30, 0).astype(np.float16)
    wt = (rng.standard_normal((co, c)) * 0.1).astype(np.float16)
    b = rng.standard_normal(co).astype(np.float32)
    oh, ow = R.pool_out(h, 3, 0, 2), R.pool_out(w, 3, 0, 2)
    xd, wd, bd = dev_from(x), dev_from(wt), dev_from(b)
    yd = dev_from(np.full((n, oh, ow, ycs), 7.0, np.float16))
    L.call("fcn_maxpool_lrn5_conv1x1_fwd_f16", xd.ptr, n, h, w, c, c, 3, 2, 0, oh, ow, 1e-4, 0.75, 1.0, wd.ptr, bd.ptr, co, relu, yd.ptr, ycs, yco, None)
    full = dev_to(yd, (n, oh, ow, ycs), np.float16)
    y = full[..., yco:yco + co].astype(np.float32)
    md = dev_from(np.zeros((n, oh, ow, c), np.float16))
    L.call("fcn_maxpool_lrn5_fwd_f16", xd.ptr, md.ptr, n, h, w, c, c, 3, 2, 0, oh, ow, c, 0, 1e-4, 0.75, 1.0, None)
    mid = dev_to(md, (n, oh, ow, c), np.float16).astype(np.float32)
    two = mid.reshape(-1, c) @ wt.astype(np.float32).T + b
    if relu:
        two = np.maximum(two, 0)
    two = two.reshape(n, oh, ow, co)
    # one rounding to half of a float32 sum: half an f16 ulp (2^-11) of the value + the summation order
    assert np.all(np.abs(y - two) <= 6e-4 * np.abs(two) + 1e-3 * np.sqrt(np.mean(two * two)))
    keep = np.ones(ycs, bool)
    keep[yco:yco + co] = False
    assert np.all(full[..., keep] == np.float16(7.0))
    x32 = x.astype(np.float32).transpose(0, 3, 1, 2)
    ref = R.conv2d(R.lrn_across(R.max_pool(x32, 3, 2, 0), 5, 1e-4, 0.75, 1.0), wt.astype(np.float32).reshape(co, c, 1, 1), b, 0, 1)
    if relu:
        ref = np.maximum(ref, 0)
    assert rel_err(y.transpose(0, 3, 1, 2), ref) < 2e-3      # (the normalised halves carry half an ulp each into a 64-term sum)
    with pytest.raises(L.FcnError):
        L.call("fcn_maxpool_lrn5_conv1x1_fwd_f16", xd.ptr, n, h, w, c, c, 3, 1, 1, oh, ow, 1e-4, 0.75, 1.0, wd.ptr, bd.ptr, co, relu, yd.ptr, ycs, yco, None)
