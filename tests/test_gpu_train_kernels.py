"""GPU parity of the backward / loss / solver kernels against the CPU oracle through the C ABI (-m gpu)."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err
from fcn_object_detector_amd import lib as L
from fcn_object_detector_amd.engine import DeviceBuffer
from gpu_util import conv_desc, dev_from, dev_to, nchw, nhwc, pack_ohwi
from oracle import caffe_ref as R

pytestmark = pytest.mark.gpu
TOL = 2e-4


def r4(c):
    return (c + 3) // 4 * 4


WG_CASES = [  # cin, cout, k, stride, pad, h, w, n
    (3, 64, 7, 2, 3, 45, 37, 2),        # conv1 geometry: Cin padded to 4, many pixel splits
    (64, 192, 3, 1, 1, 19, 17, 1),
    (16, 32, 5, 1, 2, 14, 14, 2),
    (24, 64, 5, 1, 2, 9, 12, 1),
    (480, 96, 1, 1, 0, 14, 14, 2),
    (144, 288, 3, 1, 1, 14, 14, 1),
    (1024, 4, 1, 1, 0, 7, 9, 2),        # bbox head: Cout 4
    (1024, 1, 1, 1, 0, 7, 9, 2),        # coverage head of the training net: Cout 1 (gradient buffer padded to 4)
    (112, 33, 3, 1, 1, 8, 6, 1),
]


@pytest.mark.parametrize("wg_cfg", [None, "0", "1", "2", "3", "4"])  # heuristic choice, then every tile shape / the role-split kernel forced
@pytest.mark.parametrize("case", WG_CASES)
def test_wgrad_and_dgrad_match_oracle(gpu, monkeypatch, case, wg_cfg):
    if wg_cfg is None:
        monkeypatch.delenv("FCN_WGRAD_CFG", raising=False)
    else:
        monkeypatch.setenv("FCN_WGRAD_CFG", wg_cfg)
    cin, cout, k, s, p, h, w, n = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    oh, ow = R.conv_out(h, k, p, s), R.conv_out(w, k, p, s)
    dy = rng.standard_normal((n, cout, oh, ow)).astype(np.float32)
    dw_ref, db_ref, dx_ref = R.conv2d_backward(x, wt, dy, p, s, need_dx=(s == 1))
    cin4, co4 = r4(cin), r4(cout)
    xd = dev_from(nhwc(x, cin4))
    dyd = dev_from(nhwc(dy, co4 + 8, 4))                    # gradient lives in a wider buffer at channel offset 4
    d = conv_desc(xd, xd, None, dyd, n, h, w, cin4, cin4, cout, k, p, s, oh, ow, co4 + 8, 4)
    lib = L.load()
    splits = C.c_int(0)
    nfl = int(lib.fcn_conv2d_wgrad_workspace_floats(C.byref(d), C.byref(splits)))
    ws = DeviceBuffer(nfl * 4, zero=False)
    dwd = dev_from(np.full((cout, k, k, cin4), 7.0, np.float32))
    dbd = dev_from(np.zeros(cout, np.float32))
    L.call("fcn_conv2d_wgrad_f32", C.byref(d), dwd.ptr, dbd.ptr, ws.ptr, None)
    dw = dev_to(dwd, (cout, k, k, cin4))
    assert rel_err(dw[..., :cin].transpose(0, 3, 1, 2), dw_ref) < TOL
    assert np.all(dw[..., cin:] == 0)                       # padded input channels get exactly zero gradient
    assert rel_err(dev_to(dbd, (cout,)), db_ref) < TOL
    # run it twice: bit-reproducible
    L.call("fcn_conv2d_wgrad_f32", C.byref(d), dwd.ptr, dbd.ptr, ws.ptr, None)
    assert np.array_equal(dev_to(dwd, (cout, k, k, cin4)), dw)
    if s != 1 or wg_cfg is not None:
        return
    # data gradient = forward kernel on dY with the flipped bank, accumulating into an existing gradient
    wd = dev_from(pack_ohwi(wt))
    wtd = dev_from(np.zeros((cin, k, k, co4), np.float32))
    L.call("fcn_conv_weights_flip_f32", wd.ptr, wtd.ptr, cout, k, k, cin, cin4, co4, None)
    base = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    dxd = dev_from(nhwc(base, cin4))
    dyd2 = dev_from(nhwc(dy, co4))
    dd = conv_desc(dyd2, wtd, None, dxd, n, oh, ow, co4, co4, cin, k, k - 1 - p, 1, h, w, cin4, 0, L.CONV_ACCUM)
    L.call("fcn_conv2d_fwd_f32", C.byref(dd), None)
    dx = nchw(dev_to(dxd, (n, h, w, cin4)), cin)
    assert rel_err(dx, base + dx_ref) < TOL


def test_relu_sigmoid_bwd(gpu):
    rng = np.random.default_rng(1)
    y = np.maximum(rng.standard_normal((2, 10, 5, 7)), 0).astype(np.float32)
    dy = rng.standard_normal(y.shape).astype(np.float32)
    yd, dyd = dev_from(nhwc(y, 12)), dev_from(nhwc(dy, 12))
    L.call("fcn_relu_bwd_f32", dyd.ptr, yd.ptr, dyd.ptr, 70, 10, 12, None)           # in place on the gradient
    assert np.array_equal(nchw(dev_to(dyd, (2, 5, 7, 12)), 10), dy * (y > 0))
    yc, dyc = dev_from(nhwc(y)), dev_from(nhwc(dy))
    L.call("fcn_relu_bwd_f32", dyc.ptr, yc.ptr, dyc.ptr, 70, 10, 10, None)
    assert np.array_equal(nchw(dev_to(dyc, (2, 5, 7, 10)), 10), dy * (y > 0))
    sg = R.sigmoid(rng.standard_normal(300).astype(np.float32))
    g = rng.standard_normal(300).astype(np.float32)
    sd, gd, od = dev_from(sg), dev_from(g), dev_from(np.ones(300, np.float32))
    L.call("fcn_sigmoid_bwd_f32", sd.ptr, gd.ptr, od.ptr, 300, 1, None)
    assert np.allclose(dev_to(od, (300,)), 1 + R.sigmoid_backward(sg, g), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("c,cs_dy,co_dy,cs_dx,co_dx", [(8, 16, 8, 12, 4), (6, 9, 2, 7, 1)])     # 16-byte lanes / scalar fallback
@pytest.mark.parametrize("k,s,p,h,w", [(3, 2, 0, 15, 14), (3, 1, 1, 9, 7), (2, 2, 0, 8, 6), (3, 2, 0, 28, 28)])
def test_maxpool_bwd(gpu, k, s, p, h, w, c, cs_dy, co_dy, cs_dx, co_dx):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, c, h, w)).astype(np.float32)
    x[0, :, 1, 1] = x[0, :, 1, 2]                                        # ties
    y, idx = R.max_pool(x, k, s, p, return_index=True)
    oh, ow = y.shape[2:]
    dy = rng.standard_normal(y.shape).astype(np.float32)
    ref = R.max_pool_backward(dy, idx, x.shape)
    dyd = dev_from(nhwc(dy, cs_dy, co_dy))
    idd = dev_from(np.ascontiguousarray(idx.transpose(0, 2, 3, 1)).astype(np.int32))
    base = rng.standard_normal(x.shape).astype(np.float32)
    dxd = dev_from(nhwc(base, cs_dx, co_dx))
    L.call("fcn_maxpool_bwd_f32", dyd.ptr, idd.ptr, dxd.ptr, 2, h, w, c, cs_dx, co_dx, k, s, p, oh, ow, cs_dy, co_dy, 1, None)
    assert np.allclose(nchw(dev_to(dxd, (2, h, w, cs_dx)), c, co_dx), base + ref, rtol=1e-6, atol=1e-6)
    L.call("fcn_maxpool_bwd_f32", dyd.ptr, idd.ptr, dxd.ptr, 2, h, w, c, cs_dx, co_dx, k, s, p, oh, ow, cs_dy, co_dy, 0, None)
    assert np.allclose(nchw(dev_to(dxd, (2, h, w, cs_dx)), c, co_dx), ref, rtol=1e-6, atol=1e-6)
    # with the ReLU backward of the blob folded in (fcn_maxpool_bwd_mask_f32): the total is zeroed where the activation is not positive
    act = np.maximum(rng.standard_normal(x.shape), 0).astype(np.float32)
    actd = dev_from(nhwc(act, cs_dx, co_dx))
    dxd2 = dev_from(nhwc(base, cs_dx, co_dx))
    L.call("fcn_maxpool_bwd_mask_f32", dyd.ptr, idd.ptr, dxd2.ptr, 2, h, w, c, cs_dx, co_dx, k, s, p, oh, ow, cs_dy, co_dy, 1, actd.ptr, cs_dx, co_dx, None)
    got = nchw(dev_to(dxd2, (2, h, w, cs_dx)), c, co_dx)
    assert np.allclose(got, (base + ref) * (act > 0), rtol=1e-6, atol=1e-6) and np.all(got[act <= 0] == 0)
    assert np.array_equal(dev_to(dxd2, (2, h, w, cs_dx))[..., :co_dx], nhwc(base, cs_dx, co_dx)[..., :co_dx])      # neighbours of the slice untouched


@pytest.mark.parametrize("c,ls,beta", [(64, 5, 0.75), (64, 5, 0.6), (6, 3, 0.75), (10, 5, 0.75)])   # 16-byte fast path / generic kernel
def test_lrn_bwd(gpu, c, ls, beta):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, c, 4, 5)) * 20).astype(np.float32)
    y, scale = R.lrn_across(x, ls, 1e-4, beta, 1.0, return_scale=True)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    ref = R.lrn_across_backward(x, y, scale, dy, ls, 1e-4, beta)
    xd, yd, sd, dyd = dev_from(nhwc(x)), dev_from(nhwc(y)), dev_from(nhwc(scale)), dev_from(nhwc(dy))
    base = rng.standard_normal((2, 4, 5, c)).astype(np.float32)
    dxd = dev_from(base)
    L.call("fcn_lrn_bwd_f32", xd.ptr, yd.ptr, sd.ptr, dyd.ptr, dxd.ptr, 40, c, c, c, ls, 1e-4, beta, 0, None)
    assert rel_err(nchw(dev_to(dxd, (2, 4, 5, c)), c), ref) < 1e-5
    L.call("fcn_lrn_bwd_f32", xd.ptr, yd.ptr, sd.ptr, dyd.ptr, dxd.ptr, 40, c, c, c, ls, 1e-4, beta, 1, None)
    assert rel_err(nchw(dev_to(dxd, (2, 4, 5, c)), c), 2 * ref) < 1e-5


def test_dropout_mask_is_the_oracle_mask(gpu):
    x = np.random.default_rng(4).standard_normal((3, 20, 6, 5)).astype(np.float32) + 3
    xd = dev_from(nhwc(x))
    yd = dev_from(np.zeros((3, 6, 5, 24), np.float32))
    L.call("fcn_dropout_f32", xd.ptr, yd.ptr, 3, 20, 6, 5, 20, 0, 24, 4, 0.4, 12345, 0, None)
    y = nchw(dev_to(yd, (3, 6, 5, 24)), 20, 4)
    mask = R.dropout_mask(x.shape, 0.4, 12345)
    assert np.array_equal(y, x * mask * np.float32(1.0 / (1.0 - 0.4)))
    assert 0.5 < mask.mean() < 0.7
    # a data-parallel shard (images 1..2 of the 3) reproduces its part of the mask through index_offset
    xs = dev_from(nhwc(x[1:]))
    ys = dev_from(np.zeros((2, 6, 5, 24), np.float32))
    L.call("fcn_dropout_f32", xs.ptr, ys.ptr, 2, 20, 6, 5, 20, 0, 24, 4, 0.4, 12345, 20 * 6 * 5, None)
    assert np.array_equal(nchw(dev_to(ys, (2, 6, 5, 24)), 20, 4), y[1:])


def test_losses(gpu):
    rng = np.random.default_rng(5)
    a = rng.standard_normal((8, 4, 28, 28)).astype(np.float32)
    b = rng.standard_normal((8, 4, 28, 28)).astype(np.float32)
    b[0, 0, 0, :5] = a[0, 0, 0, :5]                                       # sign(0) = 0
    ad, bd = dev_from(nhwc(a)), dev_from(nhwc(b))
    dad = dev_from(np.zeros((8, 28, 28, 4), np.float32))
    ld = dev_from(np.zeros(4, np.float32))
    L.call("fcn_loss_f32", 0, ad.ptr, bd.ptr, dad.ptr, ld.ptr, 8 * 784, 4, 4, 8, 2.0, None)
    assert abs(dev_to(ld, (4,))[0] - R.l1_loss(a, b)) < 1e-5 * R.l1_loss(a, b)
    assert np.array_equal(nchw(dev_to(dad, (8, 28, 28, 4)), 4), R.l1_loss_grad(a, b, 2.0))
    L.call("fcn_loss_f32", 1, ad.ptr, bd.ptr, dad.ptr, ld.ptr, 8 * 784, 4, 4, 8, 1.0, None)
    assert abs(dev_to(ld, (4,))[0] - R.euclidean_loss(a, b)) < 1e-5 * R.euclidean_loss(a, b)
    assert np.allclose(nchw(dev_to(dad, (8, 28, 28, 4)), 4), R.euclidean_loss_grad(a, b, 1.0), rtol=1e-6, atol=1e-8)
    # coverage head of the 1-class net: C = 1 inside a 4-wide buffer
    a1, b1 = a[:, :1], b[:, :1]
    ad, bd = dev_from(nhwc(a1, 4)), dev_from(nhwc(b1, 4))
    L.call("fcn_loss_f32", 1, ad.ptr, bd.ptr, None, ld.ptr, 8 * 784, 1, 4, 8, 1.0, None)
    assert abs(dev_to(ld, (4,))[0] - R.euclidean_loss(a1, b1)) < 1e-5 * R.euclidean_loss(a1, b1)


@pytest.mark.parametrize("kind", ["sgd", "adam"])
def test_solver_updates(gpu, kind):
    rng = np.random.default_rng(6)
    sizes, lrm, dcm = [1000, 37, 4096, 8], [1.0, 2.0, 1.0, 0.0], [1.0, 0.0, 1.0, 1.0]
    offs = np.cumsum([0] + [(s + 3) // 4 * 4 for s in sizes])
    total = int(offs[-1])
    w = rng.standard_normal(total).astype(np.float32)
    segs = (L.SolverSeg * 4)(*[L.SolverSeg(int(offs[i]), sizes[i], lrm[i], dcm[i]) for i in range(4)])
    sd = DeviceBuffer(C.sizeof(segs), zero=False)
    L.call("fcn_memcpy_h2d_async", sd.ptr, C.addressof(segs), C.sizeof(segs), None)
    wd, h1, h2 = dev_from(w), dev_from(np.zeros(total, np.float32)), dev_from(np.zeros(total, np.float32))
    wr = [w[offs[i]:offs[i] + sizes[i]].copy() for i in range(4)]
    m = [np.zeros_like(a) for a in wr]
    v = [np.zeros_like(a) for a in wr]
    for t in range(1, 4):
        g = rng.standard_normal(total).astype(np.float32)
        gd = dev_from(g)
        if kind == "sgd":
            L.call("fcn_sgd_update_f32", wd.ptr, gd.ptr, h1.ptr, sd.ptr, 4, 0.01, 0.9, 0.0005, 0.5, None)
        else:
            L.call("fcn_adam_update_f32", wd.ptr, gd.ptr, h1.ptr, h2.ptr, sd.ptr, 4, 0.01, 0.9, 0.999, 1e-8, 0.0005, t, 0.5, None)
        for i in range(4):
            gi = g[offs[i]:offs[i] + sizes[i]] * np.float32(0.5)
            if lrm[i] == 0.0:
                continue
            if kind == "sgd":
                R.sgd_update(wr[i], gi, m[i], 0.01, 0.9, 0.0005, lrm[i], dcm[i])
            else:
                R.adam_update(wr[i], gi, m[i], v[i], 0.01, 0.9, 0.999, 1e-8, 0.0005, lrm[i], dcm[i], t)
    out = dev_to(wd, (total,))
    for i in range(4):
        assert np.allclose(out[offs[i]:offs[i] + sizes[i]], wr[i], rtol=2e-5, atol=1e-6), i


def test_weights_flip_batch_equals_per_layer_flip(gpu):
    rng = np.random.default_rng(8)
    layers = [(64, 3, 24), (33, 1, 100), (16, 5, 8)]              # cout, k, cin
    w_chunks, segs, wt_off, w_off = [], [], 0, 0
    for cout, k, cin in layers:
        cin4, co4 = r4(cin), r4(cout)
        w = rng.standard_normal((cout, k, k, cin4)).astype(np.float32)
        w[..., cin:] = 0
        segs.append(L.FlipSeg(w_off, wt_off, cout, k, k, cin, cin4, co4))
        w_chunks.append(w.reshape(-1))
        w_off += w.size
        wt_off += r4(cin * k * k * co4)
    wd = dev_from(np.concatenate(w_chunks))
    wtd = dev_from(np.full(wt_off, 9.0, np.float32))
    arr = (L.FlipSeg * len(segs))(*segs)
    sd = DeviceBuffer(C.sizeof(arr), zero=False)
    L.call("fcn_memcpy_h2d_async", sd.ptr, C.addressof(arr), C.sizeof(arr), None)
    L.call("fcn_conv_weights_flip_batch_f32", wd.ptr, wtd.ptr, sd.ptr, len(segs), None)
    got = dev_to(wtd, (wt_off,))
    for sg, (cout, k, cin) in zip(segs, layers):
        one = dev_from(np.zeros(cin * k * k * sg.Cout4, np.float32))
        L.call("fcn_conv_weights_flip_f32", wd.ptr + 4 * sg.w_offset, one.ptr, cout, k, k, cin, sg.Cin4, sg.Cout4, None)
        want = dev_to(one, (cin * k * k * sg.Cout4,))
        assert np.array_equal(got[sg.wt_offset:sg.wt_offset + want.size], want)


@pytest.mark.parametrize("wg_cfg", [None, "0", "2", "4"])
def test_wgrad_group_matches_oracle(gpu, monkeypatch, wg_cfg):
    """Four layers' weight gradients in one launch + one grouped reduction (fcn_conv2d_wgrad_group_f32): each equals the
    oracle and the single-layer entry point; bias behind the weights (the solver's layout), elsewhere, or absent."""
    if wg_cfg is None:
        monkeypatch.delenv("FCN_WGRAD_CFG", raising=False)
    else:
        monkeypatch.setenv("FCN_WGRAD_CFG", wg_cfg)
    rng = np.random.default_rng(12)
    n, h, w = 2, 14, 11
    cases = [(64, 48, 1, 0), (24, 40, 3, 1), (16, 8, 5, 2), (64, 33, 1, 0)]          # cin, cout, k, pad
    lib = L.load()
    descs, refs, keep, dws, dbs = [], [], [], [], []
    for i, (cin, cout, k, pad) in enumerate(cases):
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        dy = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        wt = np.zeros((cout, cin, k, k), np.float32)
        dw_ref, db_ref, _ = R.conv2d_backward(x, wt, dy, pad, 1, need_dx=False)
        cin4, co4 = r4(cin), r4(cout)
        xd, dyd = dev_from(nhwc(x, cin4)), dev_from(nhwc(dy, co4 + 4, 4))
        d = conv_desc(xd, xd, None, dyd, n, h, w, cin4, cin4, cout, k, pad, 1, h, w, co4 + 4, 4)
        flat = dev_from(np.full(cout * k * k * cin4 + cout + 8, 3.0, np.float32))
        dws.append(flat.ptr)
        if i == 0:
            dbs.append(flat.ptr + 4 * cout * k * k * cin4)           # right behind the weights: reduced together
        elif i == 1:
            dbs.append(flat.ptr + 4 * (cout * k * k * cin4 + 4))     # somewhere else: its own reduction
        elif i == 2:
            dbs.append(None)
        else:
            dbs.append(flat.ptr + 4 * cout * k * k * cin4)
        keep += [xd, dyd, flat]
        descs.append(d)
        refs.append((flat, cout, k, cin, cin4, dw_ref, db_ref, dbs[-1]))
    arr = (L.ConvDesc * 4)(*descs)
    ws = DeviceBuffer(int(lib.fcn_conv2d_wgrad_group_workspace_floats(arr, 4)) * 4, zero=False)
    pdw = (C.c_void_p * 4)(*dws)
    pdb = (C.c_void_p * 4)(*dbs)
    L.call("fcn_conv2d_wgrad_group_f32", arr, pdw, pdb, 4, ws.ptr, None)
    first = []
    for flat, cout, k, cin, cin4, dw_ref, db_ref, dbp in refs:
        nw = cout * k * k * cin4
        got = dev_to(flat, (nw + cout + 8,))
        first.append(got.copy())
        dw = got[:nw].reshape(cout, k, k, cin4)
        assert rel_err(dw[..., :cin].transpose(0, 3, 1, 2), dw_ref) < TOL
        if dbp is None:
            assert np.all(got[nw:] == 3.0)
        else:
            off = (dbp - flat.ptr) // 4
            assert rel_err(got[off:off + cout], db_ref) < TOL
    L.call("fcn_conv2d_wgrad_group_f32", arr, pdw, pdb, 4, ws.ptr, None)             # bit-reproducible
    for (flat, cout, k, cin, cin4, *_), f in zip(refs, first):
        assert np.array_equal(dev_to(flat, f.shape), f)


def test_wgrad_configuration_named_by_the_caller(gpu, monkeypatch):
    """fcn_conv2d_wgrad_group_cfg_f32 / fcn_conv2d_wgrad_cfg_f32: every configuration the library offers matches the oracle, is
    the launch FCN_WGRAD_CFG would force (bit for bit), and an index outside -1 .. num_configs - 1 is refused."""
    monkeypatch.delenv("FCN_WGRAD_CFG", raising=False)
    rng = np.random.default_rng(5)
    lib = L.load()
    ncfg = int(lib.fcn_conv2d_wgrad_num_configs())
    assert int(lib.fcn_conv2d_wgrad_split_config()) == ncfg - 1
    n, h, w = 2, 13, 9
    cases = [(32, 40, 3, 1), (64, 24, 1, 0)]
    descs, refs, keep, dws, dbs = [], [], [], [], []
    for cin, cout, k, pad in cases:
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        dy = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        dw_ref, db_ref, _ = R.conv2d_backward(x, np.zeros((cout, cin, k, k), np.float32), dy, pad, 1, need_dx=False)
        xd, dyd = dev_from(nhwc(x, cin)), dev_from(nhwc(dy, r4(cout)))
        descs.append(conv_desc(xd, xd, None, dyd, n, h, w, cin, cin, cout, k, pad, 1, h, w, r4(cout), 0))
        flat = dev_from(np.zeros(cout * k * k * cin + cout, np.float32))
        dws.append(flat.ptr)
        dbs.append(flat.ptr + 4 * cout * k * k * cin)
        keep += [xd, dyd, flat]
        refs.append((flat, cout, k, cin, dw_ref, db_ref))
    arr = (L.ConvDesc * 2)(*descs)
    pdw, pdb = (C.c_void_p * 2)(*dws), (C.c_void_p * 2)(*dbs)

    def read():
        return [dev_to(flat, (cout * k * k * cin + cout,)).copy() for flat, cout, k, cin, _, _ in refs]

    for cfg in range(ncfg):
        ws = DeviceBuffer(int(lib.fcn_conv2d_wgrad_group_workspace_floats_cfg(arr, 2, cfg)) * 4, zero=False)
        L.call("fcn_conv2d_wgrad_group_cfg_f32", arr, pdw, pdb, 2, ws.ptr, cfg, None)
        named = read()
        for got, (flat, cout, k, cin, dw_ref, db_ref) in zip(named, refs):
            nw = cout * k * k * cin
            assert rel_err(got[:nw].reshape(cout, k, k, cin).transpose(0, 3, 1, 2), dw_ref) < TOL, cfg
            assert rel_err(got[nw:], db_ref) < TOL, cfg
        monkeypatch.setenv("FCN_WGRAD_CFG", str(cfg))
        L.call("fcn_conv2d_wgrad_group_f32", arr, pdw, pdb, 2, ws.ptr, None)
        monkeypatch.delenv("FCN_WGRAD_CFG")
        for a, b in zip(named, read()):
            assert np.array_equal(a, b), cfg
        # the single-problem entry point with the same configuration
        ws1 = DeviceBuffer(int(lib.fcn_conv2d_wgrad_workspace_floats_cfg(C.byref(descs[0]), cfg, None)) * 4, zero=False)
        L.call("fcn_conv2d_wgrad_cfg_f32", C.byref(descs[0]), dws[0], dbs[0], ws1.ptr, cfg, None)
        flat, cout, k, cin, dw_ref, db_ref = refs[0]
        got = read()[0]
        assert rel_err(got[:cout * k * k * cin].reshape(cout, k, k, cin).transpose(0, 3, 1, 2), dw_ref) < TOL, cfg
        ws.free()
        ws1.free()
    ws = DeviceBuffer(1 << 20, zero=False)
    for bad in (-2, ncfg):
        assert lib.fcn_conv2d_wgrad_group_cfg_f32(arr, pdw, pdb, 2, ws.ptr, bad, None) != 0
        assert lib.fcn_conv2d_wgrad_cfg_f32(C.byref(descs[0]), dws[0], dbs[0], ws.ptr, bad, None) != 0
