"""GPU parity of every HIP kernel family against the CPU oracle, through the C ABI (run with -m gpu).
Tolerance: 1e-3 relative (north_star) is the bar; the f32 matrix-core path is expected within 1e-4."""
import ctypes as C

import numpy as np
import pytest

from conftest import CFG_DOT1X1, CFG_FIRST7, N_TILE_CFGS, elem_err, rel_err
from fcn_object_detector_amd import lib as L
from fcn_object_detector_amd.engine import DeviceBuffer
from gpu_util import conv_desc, dev_from, dev_to, nchw, nhwc, pack_ohwi
from oracle import caffe_ref as R

pytestmark = pytest.mark.gpu
TOL = 1e-4


def run_conv(x, w, b, pad, stride, flags=0, y_cstride=None, y_coffset=0, x_cstride=None, want_y2=False):
    n, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    oh, ow = R.conv_out(h, k, pad, stride), R.conv_out(wd, k, pad, stride)
    cin4 = (cin + 3) // 4 * 4
    xcs = x_cstride or cin4
    ycs = y_cstride or cout
    xd = dev_from(nhwc(x, xcs))
    wdv = dev_from(pack_ohwi(w))
    bd = dev_from(b) if b is not None else None
    yd = dev_from(np.full((n, oh, ow, ycs), -7.0, np.float32))
    y2d = dev_from(np.zeros((n, oh, ow, cout), np.float32)) if want_y2 else None
    d = conv_desc(xd, wdv, bd, yd, n, h, wd, cin4, xcs, cout, k, pad, stride, oh, ow, ycs, y_coffset, flags, 0.0,
                  y2d, cout if want_y2 else 0, 0)
    L.call("fcn_conv2d_fwd_f32", C.byref(d), None)
    yfull = dev_to(yd, (n, oh, ow, ycs))
    y = nchw(yfull, cout, y_coffset)
    if want_y2:
        return y, nchw(dev_to(y2d, (n, oh, ow, cout)), cout), yfull
    return y, yfull


CONV_CASES = [
    # cin, cout, k, stride, pad, h, w, n
    (3, 64, 7, 2, 3, 61, 45, 1),       # conv1 geometry (Cin 3 padded to 4), odd sizes
    (64, 64, 1, 1, 0, 28, 28, 1),
    (64, 192, 3, 1, 1, 23, 19, 1),
    (16, 32, 5, 1, 2, 17, 28, 1),      # Cin 16: several taps per 32-wide k chunk
    (24, 64, 5, 1, 2, 14, 14, 2),      # Cin 24: taps straddle chunk boundaries
    (192, 48, 1, 1, 0, 9, 11, 1),
    (112, 33, 3, 1, 1, 12, 7, 1),      # Cout not a multiple of 32
    (832, 384, 1, 1, 0, 7, 7, 1),
    (144, 288, 3, 1, 1, 28, 28, 1),    # real inception_4d/3x3
    (1024, 4, 1, 1, 0, 28, 28, 1),     # coverage head
    (8, 8, 3, 2, 0, 15, 15, 3),        # stride 2 without padding, batch 3
]


@pytest.mark.parametrize("cfg", [str(i) for i in range(N_TILE_CFGS)] + [None])      # every tile configuration, the split-role ones (23..31) included
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_oracle(gpu, monkeypatch, case, cfg):
    cin, cout, k, s, p, h, w, n = case
    if cfg is None:
        monkeypatch.delenv("FCN_CONV_CFG", raising=False)
    else:
        monkeypatch.setenv("FCN_CONV_CFG", cfg)
    rng = np.random.default_rng(hash(case) % 2**32)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y, _ = run_conv(x, wt, b, p, s)
    ref = R.conv2d(x, wt, b, p, s)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < TOL


def test_conv_asymmetric_identity_layout(gpu):
    """A = I check with an asymmetric weight matrix: catches a transposed C/D write."""
    cin = cout = 64
    x = np.zeros((1, cin, 8, 8), np.float32)
    for i in range(64):
        x[0, i, i // 8, i % 8] = 1.0                    # pixel i carries channel i
    wt = (np.arange(cout * cin, dtype=np.float32).reshape(cout, cin, 1, 1) % 251) / 16.0
    y, _ = run_conv(x, wt, None, 0, 1)
    assert np.array_equal(y.reshape(cout, 64), wt.reshape(cout, cin))   # y[o, pixel i] = W[o, i]


def test_conv_epilogue_relu_slice(gpu):
    rng = np.random.default_rng(5)
    x = rng.random((1, 3, 20, 20)).astype(np.float32) - np.float32(127.0)      # Power(shift=-127)'d input, zero padding after it
    wt = rng.standard_normal((40, 3, 7, 7)).astype(np.float32) * 0.05
    b = rng.standard_normal(40).astype(np.float32)
    y, yfull = run_conv(x, wt, b, 3, 2, flags=L.CONV_RELU, y_cstride=96, y_coffset=24)
    ref = R.relu(R.conv2d(x, wt, b, 3, 2))
    assert rel_err(y, ref) < TOL
    assert np.all(yfull[..., :24] == -7.0) and np.all(yfull[..., 64:] == -7.0)      # neighbours of the slice untouched


def test_conv_sigmoid_second_output(gpu):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 64, 6, 6)).astype(np.float32)
    wt = rng.standard_normal((4, 64, 1, 1)).astype(np.float32) * 0.2
    b = np.zeros(4, np.float32)
    y, y2, _ = run_conv(x, wt, b, 0, 1, flags=L.CONV_SIGMOID2, want_y2=True)
    ref = R.conv2d(x, wt, b, 0, 1)
    assert rel_err(y, ref) < TOL and np.abs(y2 - R.sigmoid(ref)).max() < 1e-5


@pytest.mark.parametrize("cfg", ["0", "5", "14", None])
def test_conv_epilogue_relu_mask(gpu, monkeypatch, cfg):
    """FCN_CONV_MASK: the result is zeroed where a second tensor (the activation of the layer below, a channel slice of a
    wider buffer) is not positive -- the ReLU backward folded into the data-gradient pass that writes the gradient last."""
    if cfg is None:
        monkeypatch.delenv("FCN_CONV_CFG", raising=False)
    else:
        monkeypatch.setenv("FCN_CONV_CFG", cfg)
    rng = np.random.default_rng(12)
    n, cin, cout, h, w = 2, 48, 40, 13, 9
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32)
    act = np.maximum(rng.standard_normal((n, cout, h, w)), 0).astype(np.float32)       # about half exactly zero
    act_wide = np.full((n, h, w, 72), 5.0, np.float32)
    act_wide[..., 16:16 + cout] = act.transpose(0, 2, 3, 1)
    xd, wd, ad = dev_from(nhwc(x)), dev_from(pack_ohwi(wt)), dev_from(act_wide)
    yd = dev_from(np.full((n, h, w, cout), -7.0, np.float32))
    d = conv_desc(xd, wd, None, yd, n, h, w, cin, cin, cout, 3, 1, 1, h, w, cout, 0, L.CONV_MASK, 0.0, ad, 72, 16)
    L.call("fcn_conv2d_fwd_f32", C.byref(d), None)
    y = nchw(dev_to(yd, (n, h, w, cout)), cout)
    ref = R.conv2d(x, wt, None, 1, 1) * (act > 0)
    assert rel_err(y, ref) < TOL
    assert np.all(y[act <= 0] == 0.0)
    # mask and the second sigmoid output share the y2 slot: asking for both is refused
    d.flags = L.CONV_MASK | L.CONV_SIGMOID2
    assert L.load().fcn_conv2d_fwd_f32(C.byref(d), None) != 0
    d.flags, d.y2 = L.CONV_MASK, None
    assert L.load().fcn_conv2d_fwd_f32(C.byref(d), None) != 0


def test_conv_group_launch(gpu):
    """Four independent problems (an inception module's branch entries) in one launch."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal((1, 192, 14, 14)).astype(np.float32)
    xd = dev_from(nhwc(x))
    couts, ks = [64, 96, 16, 32], [1, 1, 1, 3]
    ws = [(rng.standard_normal((co, 192, k, k)) * 0.05).astype(np.float32) for co, k in zip(couts, ks)]
    bs = [rng.standard_normal(co).astype(np.float32) for co in couts]
    total = sum(couts)
    yd = dev_from(np.zeros((1, 14, 14, total), np.float32))
    keep, descs, off = [], [], 0
    for wt, b, co, k in zip(ws, bs, couts, ks):
        wd, bd = dev_from(pack_ohwi(wt)), dev_from(b)
        keep += [wd, bd]
        descs.append(conv_desc(xd, wd, bd, yd, 1, 14, 14, 192, 192, co, k, k // 2, 1, 14, 14, total, off, L.CONV_RELU))
        off += co
    arr = (L.ConvDesc * 4)(*descs)
    lib = L.load()
    wsd = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(4)), zero=False)
    grp = L.ConvGroup()
    L.call("fcn_conv2d_group_prepare", arr, 4, wsd.ptr, -1, C.byref(grp))
    L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
    y = nchw(dev_to(yd, (1, 14, 14, total)), total)
    ref = np.concatenate([R.relu(R.conv2d(x, wt, b, k // 2, 1)) for wt, b, k in zip(ws, bs, ks)], axis=1)
    assert rel_err(y, ref) < TOL


def test_a_reused_workspace_address_gets_a_fresh_plan(gpu):
    """The library keeps the host copy of a prepared launch group keyed by its device workspace.  Two prepares on the SAME address
    (an engine closed, another one's workspace handed the address again): the second plan must be the one that runs, a handle of
    the first plan must be refused when it no longer describes what is registered, and a released workspace launches nothing
    (what the round-1 abort under rocprofv3 was - and was not - about)."""
    rng = np.random.default_rng(21)
    lib = L.load()
    x = rng.standard_normal((1, 64, 12, 12)).astype(np.float32)
    xd = dev_from(nhwc(x))
    wsd = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(2)), zero=False)

    def plan(couts, k):
        keep, descs, refs = [], [], []
        for co in couts:
            wt = (rng.standard_normal((co, 64, k, k)) * 0.05).astype(np.float32)
            b = rng.standard_normal(co).astype(np.float32)
            wd, bd, yd = dev_from(pack_ohwi(wt)), dev_from(b), dev_from(np.zeros((1, 12, 12, co), np.float32))
            keep += [wd, bd, yd]
            descs.append(conv_desc(xd, wd, bd, yd, 1, 12, 12, 64, 64, co, k, k // 2, 1, 12, 12, co, 0, L.CONV_RELU))
            refs.append((yd, co, R.relu(R.conv2d(x, wt, b, k // 2, 1))))
        grp = L.ConvGroup()
        L.call("fcn_conv2d_group_prepare", (L.ConvDesc * len(descs))(*descs), len(descs), wsd.ptr, -1, C.byref(grp))
        return grp, refs, keep

    def check(grp, refs):
        L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
        for yd, co, ref in refs:
            assert rel_err(nchw(dev_to(yd, (1, 12, 12, co)), co), ref) < TOL

    first, refs1, keep1 = plan([32, 48], 1)
    check(first, refs1)
    second, refs2, keep2 = plan([40], 3)                     # same workspace address, another plan (one problem, 3x3)
    assert second.d_probs == first.d_probs
    check(second, refs2)                                     # the second plan is what runs
    assert lib.fcn_conv2d_fwd_group_f32(C.byref(first), None) != 0      # the first handle (n = 2) no longer matches the registry (n = 1)
    L.call("fcn_conv2d_group_release", wsd.ptr)
    assert lib.fcn_conv2d_fwd_group_f32(C.byref(second), None) != 0     # released: nothing is registered behind the address
    third, refs3, keep3 = plan([32, 48], 1)                  # and a prepare after the release registers a fresh plan again
    check(third, refs3)
    L.call("fcn_conv2d_group_release", wsd.ptr)


@pytest.mark.parametrize("cfg", [23, 21, 18])      # 32 x 32 tiles: 588 / 744 workgroups
@pytest.mark.parametrize("hw", [56, 61])
def test_conv_group_many_rounds_of_workgroups(gpu, cfg, hw):
    """A group whose tiles need more than two rounds of workgroups over the chip's CUs and whose problems differ in chunks
    per tile (given shortest first: prepare() sorts them): conv_fwd_group deals the rounds in a snake so that no CU keeps
    drawing the long tiles - every tile must still be computed exactly once, whatever the permutation."""
    rng = np.random.default_rng(88)
    x = rng.standard_normal((1, 64, hw, hw)).astype(np.float32)
    xd = dev_from(nhwc(x))
    couts, ks, cins = [64, 32, 96], [1, 5, 3], [64, 16, 32]      # 1x1 on 64, 5x5 on the first 16, 3x3 on the first 32 channels
    ws = [(rng.standard_normal((co, ci, k, k)) * 0.05).astype(np.float32) for co, k, ci in zip(couts, ks, cins)]
    bs = [rng.standard_normal(co).astype(np.float32) for co in couts]
    total = sum(couts)
    yd = dev_from(np.full((1, hw, hw, total), -3.0, np.float32))
    keep, descs, off = [], [], 0
    for wt, b, co, k, ci in zip(ws, bs, couts, ks, cins):
        wd, bd = dev_from(pack_ohwi(wt)), dev_from(b)
        keep += [wd, bd]
        descs.append(conv_desc(xd, wd, bd, yd, 1, hw, hw, ci, 64, co, k, k // 2, 1, hw, hw, total, off, L.CONV_RELU))
        off += co
    arr = (L.ConvDesc * 3)(*descs)
    lib = L.load()
    wsd = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(3)), zero=False)
    grp = L.ConvGroup()
    L.call("fcn_conv2d_group_prepare", arr, 3, wsd.ptr, cfg, C.byref(grp))
    assert grp.total_tiles > 512
    L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
    y = nchw(dev_to(yd, (1, hw, hw, total)), total)
    ref = np.concatenate([R.relu(R.conv2d(x[:, :ci], wt, b, k // 2, 1)) for wt, b, k, ci in zip(ws, bs, ks, cins)], axis=1)
    assert rel_err(y, ref) < TOL
    L.call("fcn_conv2d_group_release", wsd.ptr)


def test_conv_rejects_bad_arguments(gpu):
    lib = L.load()
    xd = dev_from(np.zeros(64, np.float32))
    d = conv_desc(xd, xd, None, xd, 1, 4, 4, 3, 3, 1, 1, 0, 1, 4, 4, 1)
    assert lib.fcn_conv2d_fwd_f32(C.byref(d), None) == 2 and b"multiples of 4" in lib.fcn_last_error_string()
    d = conv_desc(xd, xd, None, xd, 1, 4, 4, 4, 4, 1, 3, 1, 1, 5, 4, 1)
    assert lib.fcn_conv2d_fwd_f32(C.byref(d), None) == 1           # wrong OH


@pytest.mark.parametrize("k,s,p,h,w,c,cs", [(3, 2, 0, 28, 28, 64, 64), (3, 2, 0, 15, 21, 8, 12), (3, 1, 1, 9, 7, 12, 12),
                                            (2, 2, 0, 8, 6, 5, 8), (3, 2, 1, 10, 11, 7, 8)])
def test_maxpool_matches_oracle(gpu, k, s, p, h, w, c, cs):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, c, h, w)).astype(np.float32)
    x[0, :, 0, 0] = x[0, :, 0, 1]                         # ties: first maximum must win
    oh, ow = R.pool_out(h, k, p, s), R.pool_out(w, k, p, s)
    xd = dev_from(nhwc(x, cs))
    yd = dev_from(np.zeros((2, oh, ow, cs + 4), np.float32))
    idd = dev_from(np.zeros((2, oh, ow, c), np.int32))
    off = 4 if c % 4 == 0 else 1
    L.call("fcn_maxpool_fwd_f32", xd.ptr, yd.ptr, idd.ptr, 2, h, w, c, cs, k, s, p, oh, ow, cs + 4, off, None)
    y = nchw(dev_to(yd, (2, oh, ow, cs + 4)), c, off)
    idx = dev_to(idd, (2, oh, ow, c), np.int32).transpose(0, 3, 1, 2)
    ref, ridx = R.max_pool(x, k, s, p, return_index=True)
    assert np.array_equal(y, ref) and np.array_equal(idx, ridx)


def test_avepool_matches_oracle(gpu):
    rng = np.random.default_rng(10)
    x = rng.standard_normal((1, 6, 56, 56)).astype(np.float32)
    for k, s, p in [(56, 56, 0), (28, 28, 0), (14, 14, 0), (8, 8, 0), (3, 2, 1)]:
        oh = R.pool_out(56, k, p, s)
        xd = dev_from(nhwc(x, 8))
        yd = dev_from(np.zeros((1, oh, oh, 6), np.float32))
        L.call("fcn_avepool_fwd_f32", xd.ptr, yd.ptr, 1, 56, 56, 6, 8, k, s, p, oh, oh, 6, 0, None)
        y = nchw(dev_to(yd, (1, oh, oh, 6)), 6)
        assert rel_err(y, R.ave_pool(x, k, s, p)) < 1e-5


@pytest.mark.parametrize("c,ls", [(64, 5), (192, 5), (10, 5), (16, 3)])
def test_lrn_matches_oracle(gpu, c, ls):
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((2, c, 5, 7)) * 40).astype(np.float32)
    cs = (c + 3) // 4 * 4
    xd = dev_from(nhwc(x, cs))
    yd = dev_from(np.zeros((2, 5, 7, cs), np.float32))
    sd = dev_from(np.zeros((2, 5, 7, c), np.float32))
    L.call("fcn_lrn_fwd_f32", xd.ptr, yd.ptr, sd.ptr, 70, c, cs, cs, ls, 1e-4, 0.75, 1.0, None)
    y = nchw(dev_to(yd, (2, 5, 7, cs)), c)
    ref, scale = R.lrn_across(x, ls, 1e-4, 0.75, 1.0, return_scale=True)
    assert rel_err(y, ref) < 1e-5
    assert rel_err(nchw(dev_to(sd, (2, 5, 7, c)), c), scale) < 1e-6


@pytest.mark.parametrize("n,h,w,cout,relu", [(1, 64, 64, 64, 1), (2, 37, 51, 64, 0), (1, 30, 130, 48, 1), (1, 448, 448, 64, 1), (3, 16, 16, 40, 1)])
def test_first_layer_kernel_matches_oracle(gpu, n, h, w, cout, relu):
    """Configuration 30 = conv_first7_kernel (7x7, stride 2, pad 3 on 4-channel pixels: conv1/7x7_s2 of deploy.prototxt),
    output sizes that are not multiples of its 8 x 32 patch included; other shapes must be refused for it."""
    lib = L.load()
    rng = np.random.default_rng(21)
    x = rng.standard_normal((n, 3, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, 3, 7, 7)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    oh, ow = R.conv_out(h, 7, 3, 2), R.conv_out(w, 7, 3, 2)
    xd = dev_from(nhwc(x, 4))
    wd = dev_from(np.ascontiguousarray(np.pad(wt, ((0, 0), (0, 1), (0, 0), (0, 0))).transpose(0, 2, 3, 1)))
    bd = dev_from(b)
    yd = dev_from(np.full((n, oh, ow, cout + 4), 7.0, np.float32))
    d = conv_desc(xd, wd, bd, yd, n, h, w, 4, 4, cout, 7, 3, 2, oh, ow, cout + 4, 4, L.CONV_RELU if relu else 0)
    arr = (L.ConvDesc * 1)(d)
    ws = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(1)), zero=False)
    grp = L.ConvGroup()
    L.call("fcn_conv2d_group_prepare", arr, 1, ws.ptr, 30, C.byref(grp))
    assert grp.cfg == CFG_FIRST7 and grp.total_tiles == n * ((oh + 7) // 8) * ((ow + 31) // 32)
    L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
    full = dev_to(yd, (n, oh, ow, cout + 4))
    ref = R.conv2d(x, wt, b, 3, 2)
    if relu:
        ref = R.relu(ref)
    assert rel_err(nchw(full, cout, 4), ref) < 1e-5
    assert np.all(full[..., :4] == 7.0)      # the channels in front of the slice are not touched
    # auto-selection (no autotune) picks it for this shape; a 3x3 problem is refused for configuration 30
    L.call("fcn_conv2d_group_prepare", arr, 1, ws.ptr, -1, C.byref(grp))
    assert grp.cfg == CFG_FIRST7
    d3 = conv_desc(xd, wd, bd, yd, n, h, w, 4, 4, cout, 3, 1, 1, h, w, cout + 4, 4, 0)
    assert lib.fcn_conv2d_group_prepare((L.ConvDesc * 1)(d3), 1, ws.ptr, 30, C.byref(grp)) != 0
    L.call("fcn_conv2d_group_release", ws.ptr)


@pytest.mark.parametrize("n,h,w,cin,couts", [(1, 28, 28, 1024, (4, 16)), (2, 5, 7, 480, (1, 4)), (1, 3, 3, 36, (8, 8, 8, 3)), (1, 9, 5, 1024, (20,))])
def test_lane_split_1x1_kernel_matches_oracle(gpu, n, h, w, cin, couts):
    """Configuration CFG_DOT1X1 = conv_dot1x1_kernel (the detection heads: narrow 1x1 convolutions, K spread over the lanes of a wave):
    groups of up to four 8-channel slices, pixel counts that are not multiples of 4, K that is not a multiple of 256, the
    sigmoid second output and ReLU; groups it does not take must be refused."""
    lib = L.load()
    rng = np.random.default_rng(31)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    xd = dev_from(nhwc(x, cin))
    keep, descs, refs = [xd], [], []
    for i, cout in enumerate(couts):
        wt = (rng.standard_normal((cout, cin, 1, 1)) * 0.05).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        wd, bd = dev_from(pack_ohwi(wt)), dev_from(b)
        yd = dev_from(np.full((n, h, w, cout + 4), 3.0, np.float32))
        sig = i == 0
        y2d = dev_from(np.zeros((n, h, w, cout), np.float32)) if sig else None
        flags = L.CONV_SIGMOID2 if sig else (L.CONV_RELU if i == 1 else 0)
        descs.append(conv_desc(xd, wd, bd, yd, n, h, w, cin, cin, cout, 1, 0, 1, h, w, cout + 4, 4, flags, 0.0, y2d, cout, 0))
        keep += [wd, bd, yd, y2d]
        ref = R.conv2d(x, wt, b, 0, 1)
        refs.append((yd, y2d, cout, R.relu(ref) if flags == L.CONV_RELU else ref))
    arr = (L.ConvDesc * len(descs))(*descs)
    ws = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(len(descs))), zero=False)
    grp = L.ConvGroup()
    L.call("fcn_conv2d_group_prepare", arr, len(descs), ws.ptr, CFG_DOT1X1, C.byref(grp))
    assert grp.cfg == CFG_DOT1X1 and grp.total_tiles == (n * h * w + 3) // 4
    L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
    for yd, y2d, cout, ref in refs:
        full = dev_to(yd, (n, h, w, cout + 4))
        assert rel_err(nchw(full, cout, 4), ref) < 1e-5
        assert np.all(full[..., :4] == 3.0)
        if y2d is not None:
            assert rel_err(nchw(dev_to(y2d, (n, h, w, cout)), cout), R.sigmoid(ref)) < 1e-5
    # a 3x3 problem, or more than four slices, is refused for this configuration
    d3 = conv_desc(xd, keep[1], keep[2], keep[3], n, h, w, cin, cin, couts[0], 3, 1, 1, h, w, couts[0] + 4, 4, 0)
    assert lib.fcn_conv2d_group_prepare((L.ConvDesc * 1)(d3), 1, ws.ptr, CFG_DOT1X1, C.byref(grp)) != 0
    five = (L.ConvDesc * 5)(*([descs[0]] * 5))
    ws5 = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(5)), zero=False)
    assert lib.fcn_conv2d_group_prepare(five, 5, ws5.ptr, CFG_DOT1X1, C.byref(grp)) != 0
    L.call("fcn_conv2d_group_release", ws.ptr)


@pytest.mark.parametrize("lrn_first", [0, 1])
@pytest.mark.parametrize("k,s,p,h,w,c,cs", [(3, 2, 0, 28, 28, 64, 64), (3, 2, 0, 15, 21, 8, 12), (3, 1, 1, 9, 7, 12, 12), (3, 2, 1, 10, 11, 40, 40),
                                            (3, 2, 0, 17, 9, 192, 192),
                                            (3, 2, 0, 112, 112, 192, 192), (3, 2, 0, 57, 61, 96, 96), (3, 2, 0, 64, 70, 64, 80)])      # (LRN first: the LDS-patch form)
def test_maxpool_lrn_single_pass_matches_oracle(gpu, monkeypatch, lrn_first, k, s, p, h, w, c, cs):
    """(the large LRN-first shapes run twice: the default single pass and the opt-in LDS-patch form, FCN_LRN_POOL_LDS=1 - same bits)
    fcn_maxpool_lrn5_fwd_f32 (pool1 -> norm1 and norm2 -> pool2 of deploy.prototxt as one launch) against the oracle's
    two layers, and against the library's own two launches (bit for bit when the pooling comes first)."""
    rng = np.random.default_rng(13)
    x = (rng.standard_normal((2, c, h, w)) * 30).astype(np.float32)
    oh, ow = R.pool_out(h, k, p, s), R.pool_out(w, k, p, s)
    xd = dev_from(nhwc(x, cs))
    yd = dev_from(np.zeros((2, oh, ow, cs), np.float32))
    L.call("fcn_maxpool_lrn5_fwd_f32", xd.ptr, yd.ptr, 2, h, w, c, cs, k, s, p, oh, ow, cs, lrn_first, 1e-4, 0.75, 1.0, None)
    y = nchw(dev_to(yd, (2, oh, ow, cs)), c)
    if lrn_first and h * w * c >= 1 << 18:
        monkeypatch.setenv("FCN_LRN_POOL_LDS", "1")
        pd = dev_from(np.zeros((2, oh, ow, cs), np.float32))
        L.call("fcn_maxpool_lrn5_fwd_f32", xd.ptr, pd.ptr, 2, h, w, c, cs, k, s, p, oh, ow, cs, lrn_first, 1e-4, 0.75, 1.0, None)
        monkeypatch.delenv("FCN_LRN_POOL_LDS")
        assert np.array_equal(nchw(dev_to(pd, (2, oh, ow, cs)), c), y)
    ref = R.max_pool(R.lrn_across(x, 5, 1e-4, 0.75, 1.0), k, s, p) if lrn_first else R.lrn_across(R.max_pool(x, k, s, p), 5, 1e-4, 0.75, 1.0)
    assert rel_err(y, ref) < 1e-5
    mh, mw = (h, w) if lrn_first else (oh, ow)
    md = dev_from(np.zeros((2, mh, mw, cs), np.float32))
    zd = dev_from(np.zeros((2, oh, ow, cs), np.float32))
    if lrn_first:
        L.call("fcn_lrn_fwd_f32", xd.ptr, md.ptr, None, 2 * h * w, c, cs, cs, 5, 1e-4, 0.75, 1.0, None)
        L.call("fcn_maxpool_fwd_f32", md.ptr, zd.ptr, None, 2, h, w, c, cs, k, s, p, oh, ow, cs, 0, None)
    else:
        L.call("fcn_maxpool_fwd_f32", xd.ptr, md.ptr, None, 2, h, w, c, cs, k, s, p, oh, ow, cs, 0, None)
        L.call("fcn_lrn_fwd_f32", md.ptr, zd.ptr, None, 2 * oh * ow, c, cs, cs, 5, 1e-4, 0.75, 1.0, None)
    two = nchw(dev_to(zd, (2, oh, ow, cs)), c)
    if lrn_first:      # the single pass normalises with the hardware rsq / sqrt (1 ulp each)
        assert np.allclose(y, two, rtol=1e-6, atol=0)
    else:
        assert np.array_equal(y, two)


@pytest.mark.parametrize("s,p,h,w,relu,bias,ycs,yco", [(2, 0, 224, 224, 1, True, 64, 0), (2, 0, 15, 21, 1, True, 64, 0), (1, 1, 9, 7, 0, True, 80, 16),
                                                       (2, 1, 10, 11, 1, False, 64, 0), (2, 0, 17, 37, 0, True, 128, 64)])
def test_pool_lrn_conv1x1_single_pass_matches_oracle(gpu, s, p, h, w, relu, bias, ycs, yco):
    """fcn_maxpool_lrn5_conv1x1_fwd_f32 (pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce + ReLU of deploy.prototxt:54-104 as one launch)
    against the oracle's three layers, and against the library's own pool + LRN launch followed by its convolution kernel; channels of the
    output pixel outside the convolution's slice stay untouched."""
    rng = np.random.default_rng(29)
    n, c, co, k = 2, 64, 64, 3
    x = np.maximum(rng.standard_normal((n, c, h, w)) * 30, 0).astype(np.float32)      # (a blob behind a ReLU, like conv1's)
    wt = (rng.standard_normal((co, c, 1, 1)) * 0.1).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32) if bias else None
    oh, ow = R.pool_out(h, k, p, s), R.pool_out(w, k, p, s)
    xd = dev_from(nhwc(x, c))
    wd = dev_from(np.ascontiguousarray(wt.reshape(co, c)))
    bd = dev_from(b) if bias else None
    y0 = np.full((n, oh, ow, ycs), 7.0, np.float32)
    yd = dev_from(y0)
    L.call("fcn_maxpool_lrn5_conv1x1_fwd_f32", xd.ptr, n, h, w, c, c, k, s, p, oh, ow, 1e-4, 0.75, 1.0, wd.ptr, bd.ptr if bias else None, co, relu,
           yd.ptr, ycs, yco, None)
    full = dev_to(yd, (n, oh, ow, ycs))
    y = np.ascontiguousarray(full[..., yco:yco + co].transpose(0, 3, 1, 2))
    mid = R.lrn_across(R.max_pool(x, k, s, p), 5, 1e-4, 0.75, 1.0)
    ref = R.conv2d(mid, wt, b, 0, 1)
    if relu:
        ref = np.maximum(ref, 0)
    assert rel_err(y, ref) < 1e-5
    assert elem_err(y, ref, tol=1e-4)[0] <= 1.0
    keep = np.ones(ycs, bool)
    keep[yco:yco + co] = False
    assert np.all(full[..., keep] == 7.0)
    # the library's own two launches: pool + LRN (IEEE square roots there, the hardware ones inside the single pass), then the tiled convolution
    md = dev_from(np.zeros((n, oh, ow, c), np.float32))
    L.call("fcn_maxpool_lrn5_fwd_f32", xd.ptr, md.ptr, n, h, w, c, c, k, s, p, oh, ow, c, 0, 1e-4, 0.75, 1.0, None)
    two, _ = run_conv(nchw(dev_to(md, (n, oh, ow, c)), c), wt, b, 0, 1, flags=L.CONV_RELU if relu else 0)
    assert rel_err(y, two) < 1e-5
    with pytest.raises(L.FcnError):      # other channel counts: the caller runs the layers separately
        L.call("fcn_maxpool_lrn5_conv1x1_fwd_f32", xd.ptr, n, h, w, 32, c, k, s, p, oh, ow, 1e-4, 0.75, 1.0, wd.ptr, None, co, relu, yd.ptr, ycs, yco, None)


def test_layout_roundtrip(gpu):
    rng = np.random.default_rng(12)
    x = rng.standard_normal((3, 37, 9, 13)).astype(np.float32)
    xd = dev_from(x)
    yd = dev_from(np.zeros((3, 9, 13, 48), np.float32))
    L.call("fcn_nchw_to_nhwc_f32", xd.ptr, yd.ptr, 3, 37, 9, 13, 48, 8, -127.0, None)
    y = dev_to(yd, (3, 9, 13, 48))
    assert np.array_equal(y[..., 8:45], x.transpose(0, 2, 3, 1) + np.float32(-127.0))
    L.call("fcn_nchw_to_nhwc_f32", xd.ptr, yd.ptr, 3, 37, 9, 13, 48, 8, 0.0, None)
    y = dev_to(yd, (3, 9, 13, 48))
    assert np.array_equal(y[..., 8:45], x.transpose(0, 2, 3, 1)) and np.all(y[..., :8] == 0) and np.all(y[..., 45:] == 0)
    zd = dev_from(np.zeros_like(x))
    L.call("fcn_nhwc_to_nchw_f32", yd.ptr, zd.ptr, 3, 37, 9, 13, 48, 8, None)
    assert np.array_equal(dev_to(zd, x.shape), x)


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_image_upload_layout_kernel(gpu, c):
    """The lane-per-pixel form of fcn_nchw_to_nhwc_f32 (at most four channels into 4-float pixels: the net's image): the
    blob's channels shifted, the pad channels zero, odd sizes, a batch."""
    rng = np.random.default_rng(40 + c)
    x = rng.standard_normal((2, c, 17, 23)).astype(np.float32)
    xd = dev_from(x)
    yd = dev_from(np.full((2, 17, 23, 4), 7.0, np.float32))
    L.call("fcn_nchw_to_nhwc_f32", xd.ptr, yd.ptr, 2, c, 17, 23, 4, 0, -127.0, None)
    y = dev_to(yd, (2, 17, 23, 4))
    assert np.array_equal(y[..., :c], x.transpose(0, 2, 3, 1) + np.float32(-127.0)) and np.all(y[..., c:] == 0)


def test_several_blobs_nhwc_to_nchw_in_one_launch(gpu):
    """fcn_nhwc_to_nchw_multi_f32: three blobs of different shapes, channel slices of wider buffers, one launch; and its refusals."""
    import ctypes as C
    rng = np.random.default_rng(44)
    shapes = [(1, 4, 28, 28, 4, 0), (1, 16, 28, 28, 16, 0), (2, 5, 7, 9, 12, 3)]      # N, C, H, W, cstride, coffset
    srcs, dsts, want = [], [], []
    arr = (L.LayoutDesc * len(shapes))()
    for d, (n, c, h, w, cs, co) in zip(arr, shapes):
        buf = rng.standard_normal((n, h, w, cs)).astype(np.float32)
        sd, dd = dev_from(buf), dev_from(np.zeros((n, c, h, w), np.float32))
        srcs.append(sd); dsts.append(dd); want.append(buf[..., co:co + c].transpose(0, 3, 1, 2))
        d.src, d.dst, d.N, d.C, d.H, d.W, d.src_cstride, d.src_coffset = sd.ptr, dd.ptr, n, c, h, w, cs, co
    L.call("fcn_nhwc_to_nchw_multi_f32", arr, len(shapes), None)
    for dd, wv in zip(dsts, want):
        assert np.array_equal(dev_to(dd, wv.shape), wv)
    lib = L.load()
    assert lib.fcn_nhwc_to_nchw_multi_f32(arr, 0, None) != 0 and lib.fcn_nhwc_to_nchw_multi_f32(arr, 9, None) != 0
    arr[1].src_cstride = 8                                   # a slice wider than its stride
    assert lib.fcn_nhwc_to_nchw_multi_f32(arr, 3, None) != 0


def test_unary_eltwise_copy(gpu):
    rng = np.random.default_rng(13)
    a = rng.standard_normal(1003).astype(np.float32)
    b = rng.standard_normal(1003).astype(np.float32)
    ad, bd, yd = dev_from(a), dev_from(b), dev_from(np.zeros(1003, np.float32))
    L.call("fcn_relu_fwd_f32", ad.ptr, yd.ptr, 1003, 0.0, None)
    assert np.array_equal(dev_to(yd, (1003,)), np.maximum(a, 0))
    L.call("fcn_relu_fwd_f32", ad.ptr, yd.ptr, 1003, 0.1, None)
    assert np.allclose(dev_to(yd, (1003,)), np.where(a > 0, a, np.float32(0.1) * a))
    L.call("fcn_sigmoid_fwd_f32", ad.ptr, yd.ptr, 1003, None)
    assert np.abs(dev_to(yd, (1003,)) - R.sigmoid(a)).max() < 1e-6
    L.call("fcn_power_fwd_f32", ad.ptr, yd.ptr, 1003, 1.0, 2.0, -127.0, None)
    assert np.array_equal(dev_to(yd, (1003,)), a * np.float32(2) + np.float32(-127))
    for op, ref in [(L.ELT_PROD, a * b), (L.ELT_SUM, np.float32(0.5) * a + np.float32(2) * b), (L.ELT_MAX, np.maximum(a, b))]:
        L.call("fcn_eltwise_fwd_f32", ad.ptr, bd.ptr, yd.ptr, 1003, op, 0.5, 2.0, None)
        assert np.allclose(dev_to(yd, (1003,)), ref, rtol=1e-6, atol=1e-6)
    src = rng.standard_normal((10, 12)).astype(np.float32)
    sd, dd = dev_from(src), dev_from(np.zeros((10, 20), np.float32))
    L.call("fcn_copy_channels_f32", sd.ptr, dd.ptr, 10, 5, 12, 3, 20, 9, None)
    out = dev_to(dd, (10, 20))
    assert np.array_equal(out[:, 9:14], src[:, 3:8]) and np.all(out[:, :9] == 0) and np.all(out[:, 14:] == 0)


@pytest.mark.parametrize("c,k,s,p,h", [(44, 8, 4, 2, 14), (11, 4, 2, 1, 7), (11, 16, 8, 4, 5)])
def test_depthwise_deconv_matches_oracle(gpu, c, k, s, p, h):
    rng = np.random.default_rng(14)
    x = rng.standard_normal((2, c, h, h)).astype(np.float32)
    w = R.bilinear_filler((c, 1, k, k)) * rng.random((c, 1, 1, 1)).astype(np.float32)
    oh = R.deconv_out(h, k, p, s)
    cs = (c + 3) // 4 * 4
    xd, wd = dev_from(nhwc(x, cs)), dev_from(w.reshape(c, k, k))
    yd = dev_from(np.zeros((2, oh, oh, cs), np.float32))
    L.call("fcn_deconv_depthwise_fwd_f32", xd.ptr, wd.ptr, None, yd.ptr, 2, h, h, c, cs, k, s, p, oh, oh, cs, 0, None)
    y = nchw(dev_to(yd, (2, oh, oh, cs)), c)
    assert rel_err(y, R.deconv2d(x, w, None, p, s, group=c)) < 1e-5


@pytest.mark.parametrize("cfg", [2, 5, 8, 13, 23, 26])
@pytest.mark.parametrize("with_idx", [False, True])
def test_maxpool_fused_into_conv_group(gpu, cfg, with_idx):
    """Two MAX poolings ride in a grouped convolution launch (fcn_conv2d_group_prepare_fused): conv outputs unchanged,
    pool outputs / argmax equal to the oracle's (an inception module's 3x3 s1 pool beside its 1x1 convolutions)."""
    rng = np.random.default_rng(31)
    n, h, w, cin = 2, 14, 11, 24
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    x[0, :, 2, 2] = x[0, :, 2, 3]                                   # ties: the first maximum in raster order wins
    xd = dev_from(nhwc(x, cin))
    descs, keep, refs = [], [], []
    for cout, k, pad in ((40, 1, 0), (16, 3, 1)):
        wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        wd, bd, yd = dev_from(pack_ohwi(wt)), dev_from(b), dev_from(np.zeros((n, h, w, cout), np.float32))
        keep += [wd, bd, yd]
        descs.append(conv_desc(xd, wd, bd, yd, n, h, w, cin, cin, cout, k, pad, 1, h, w, cout, 0, L.CONV_RELU))
        refs.append((yd, cout, np.maximum(R.conv2d(x, wt, b, pad, 1), 0)))
    pools, pool_refs = [], []
    for k, s, p, cs, co in ((3, 1, 1, 32, 8), (3, 2, 0, 24, 0)):
        y, idx = R.max_pool(x, k, s, p, return_index=True)
        oh, ow = y.shape[2:]
        yd = dev_from(np.full((n, oh, ow, cs), -7.0, np.float32))
        idd = dev_from(np.zeros((n, oh, ow, cin), np.int32)) if with_idx else None
        d = L.PoolDesc()
        d.x, d.y, d.idx = xd.ptr, yd.ptr, (idd.ptr if idd is not None else None)
        d.N, d.H, d.W, d.C, d.x_cstride, d.k, d.stride, d.pad = n, h, w, cin, cin, k, s, p
        d.OH, d.OW, d.y_cstride, d.y_coffset = oh, ow, cs, co
        pools.append(d)
        pool_refs.append((yd, idd, cs, co, y, idx))
    arr = (L.ConvDesc * 2)(*descs)
    parr = (L.PoolDesc * 2)(*pools)
    ws = DeviceBuffer(int(L.load().fcn_conv2d_group_workspace_bytes(2)), zero=False)
    grp = L.ConvGroup()
    L.call("fcn_conv2d_group_prepare_fused", arr, 2, parr, 2, ws.ptr, cfg, C.byref(grp))
    L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
    for yd, cout, ref in refs:
        assert rel_err(nchw(dev_to(yd, (n, h, w, cout)), cout), ref) < 1e-5
    for yd, idd, cs, co, y, idx in pool_refs:
        oh, ow = y.shape[2:]
        got = dev_to(yd, (n, oh, ow, cs))
        assert np.array_equal(nchw(got, cin, co), y)
        assert np.all(got[..., :co] == -7.0) and np.all(got[..., co + cin:] == -7.0)       # neighbours of the slice untouched
        if idd is not None:
            assert np.array_equal(dev_to(idd, (n, oh, ow, cin), np.int32).transpose(0, 3, 1, 2), idx)
    # misaligned poolings are refused, not silently mis-computed
    pools[0].y_coffset = 2
    bad = (L.PoolDesc * 1)(pools[0])
    assert L.load().fcn_conv2d_group_prepare_fused(arr, 2, bad, 1, ws.ptr, cfg, C.byref(grp)) == 2      # FCN_E_ALIGN


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("cfg", [23, 24, 25, 27])
@pytest.mark.parametrize("hw", [(28, 28), (9, 7)])
def test_narrow_heads_as_the_tail_of_their_producers(gpu, cfg, hw, split):
    """fcn_conv2d_group_attach_tail: the detection heads (cvg/classifier with its sigmoid + bbox/regressor, deploy.prototxt:2363-2410)
    evaluated by the launches that produce their input - three branches of a concat blob, all in the finalising launch or one of them
    in an earlier contribute-only launch - against the oracle's convolutions; a second pair of launches must give the same bits (the
    arrival words are left zero, the partial sums are added in slot order)."""
    rng = np.random.default_rng(31)
    h, w = hw
    n, cin = 1, 64
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    xd = dev_from(nhwc(x))
    couts, ks = [64, 96, 32], [1, 3, 1]
    K = sum(couts)
    ws = [(rng.standard_normal((co, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32) for co, k in zip(couts, ks)]
    bs = [rng.standard_normal(co).astype(np.float32) * 0.1 for co in couts]
    blob = dev_from(np.zeros((n, h, w, K), np.float32))
    keep, descs, off = [], [], 0
    for wt, b, co, k in zip(ws, bs, couts, ks):
        wd, bd = dev_from(pack_ohwi(wt)), dev_from(b)
        keep += [wd, bd]
        descs.append(conv_desc(xd, wd, bd, blob, n, h, w, cin, cin, co, k, k // 2, 1, h, w, K, off, L.CONV_RELU))
        off += co
    hws = [(rng.standard_normal((co, K, 1, 1)) / np.sqrt(K)).astype(np.float32) for co in (4, 16)]
    hbs = [rng.standard_normal(co).astype(np.float32) for co in (4, 16)]
    cvg, sig, box = (dev_from(np.full((n, h, w, c), -3.0, np.float32)) for c in (4, 4, 16))
    tail = L.ConvTail()
    tail.n = 2
    hd = [dev_from(np.ascontiguousarray(hw_.reshape(hw_.shape[0], K))) for hw_ in hws]
    hb = [dev_from(b) for b in hbs]
    tail.heads[0] = conv_desc(blob, hd[0], hb[0], cvg, n, h, w, K, K, 4, 1, 0, 1, h, w, 4, 0, L.CONV_SIGMOID2, 0.0, sig, 4, 0)
    tail.heads[1] = conv_desc(blob, hd[1], hb[1], box, n, h, w, K, K, 16, 1, 0, 1, h, w, 16, 0, 0)
    lib = L.load()
    sb, ab = int(lib.fcn_conv2d_tail_scratch_bytes(C.byref(tail))), int(lib.fcn_conv2d_tail_arrive_bytes(C.byref(tail)))
    assert sb == (K // 32) * n * h * w * 20 * 4 and ab >= 4 * ((n * h * w + 31) // 32)
    scratch = dev_from(np.full(sb // 4, np.nan, np.float32))       # (every slot of a pixel is written before it is read)
    arrive = dev_from(np.zeros(ab // 4, np.uint32))
    tail.scratch, tail.arrive = scratch.ptr, arrive.ptr
    launches = [([0], 0), ([1, 2], 1)] if split else [([0, 1, 2], 1)]
    groups = []
    for idx, fin in launches:
        arr = (L.ConvDesc * len(idx))(*[descs[i] for i in idx])
        wsd = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(len(idx))), zero=False)
        tail.finalize = fin
        L.call("fcn_conv2d_group_attach_tail", wsd.ptr, C.byref(tail))
        grp = L.ConvGroup()
        L.call("fcn_conv2d_group_prepare", arr, len(idx), wsd.ptr, cfg, C.byref(grp))
        assert grp.cfg == cfg
        groups.append((arr, wsd, grp))
    results = []
    for _ in range(2):
        for _, _, grp in groups:
            L.call("fcn_conv2d_fwd_group_f32", C.byref(grp), None)
        results.append([dev_to(d, (n, h, w, c)) for d, c in ((blob, K), (cvg, 4), (sig, 4), (box, 16))])
        assert not dev_to(arrive, (ab // 4,), np.uint32).any()
    for a, b in zip(*results):
        assert np.array_equal(a, b)
    yb, yc, ys, yx = results[0]
    ref = np.concatenate([R.relu(R.conv2d(x, wt, b, k // 2, 1)) for wt, b, k in zip(ws, bs, ks)], axis=1)
    assert rel_err(nchw(yb, K), ref) < TOL
    rc, rx = R.conv2d(ref, hws[0], hbs[0], 0, 1), R.conv2d(ref, hws[1], hbs[1], 0, 1)
    assert rel_err(nchw(yc, 4), rc) < TOL and rel_err(nchw(yx, 16), rx) < TOL
    assert elem_err(nchw(yc, 4), rc, tol=1e-4)[0] <= 1.0 and elem_err(nchw(yx, 16), rx, tol=1e-4)[0] <= 1.0
    assert np.abs(nchw(ys, 4) - R.sigmoid(rc)).max() < 1e-5
    # a configuration without a tail variant is refused while the tail is attached; detached, the plain group runs there
    _, wsd, grp = groups[-1]
    arr = groups[-1][0]
    assert lib.fcn_conv2d_group_prepare(arr, len(launches[-1][0]), wsd.ptr, 5, C.byref(grp)) != 0
    L.call("fcn_conv2d_group_attach_tail", wsd.ptr, None)
    L.call("fcn_conv2d_group_prepare", arr, len(launches[-1][0]), wsd.ptr, 5, C.byref(grp))
    for _, wsd, _ in groups:
        L.call("fcn_conv2d_group_release", wsd.ptr)
