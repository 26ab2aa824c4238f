"""Data-parallel training step (-m gpu): RCCL path with a one-rank communicator, and the two-rank semantics
(shard the minibatch, sum gradients, scale by 1/G, identical replicas) with two processes sharing the GPU."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train(rank, world, port, use_rccl, steps, q, devices=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fcn_object_detector_amd import dp, models, proto
    from fcn_object_detector_amd.netspec import NetSpec, fill_params
    from fcn_object_detector_amd.train import SolverParams, TrainEngine
    from test_gpu_train import make_batch
    device = devices[rank] if devices else 0
    cp = dp.ControlPlane(rank, world, "127.0.0.1", port, token="dp:%d" % world, timeout=120)
    if use_rccl:
        comm = dp.RcclComm(cp, device)
    elif world > 1:
        from host_comm import HostComm
        comm = HostComm(cp)
    else:
        comm = None
    total_batch, h, w = 4, 64, 96
    per = total_batch // world
    msg = proto.parse_text(models.googlenet_detectnet_train("m", "L", "x", num_classes=1))
    full = make_batch(np.random.default_rng(5), total_batch, h, w)
    shapes = {k: (per,) + v.shape[1:] for k, v in full.items()}
    spec = NetSpec(msg, "TRAIN")
    spec.infer(shapes)
    params = fill_params(spec, seed=1234)
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params=params, device=device, comm=comm, autotune=False,
                      solver=SolverParams(base_lr=1e-3, momentum=0.9, weight_decay=1e-7))
    drop = eng.blobs["pool5/drop_s1"]
    eng.dropout_index_offset = rank * int(np.prod(drop.shape))
    losses = []
    for it in range(steps):
        batch = make_batch(np.random.default_rng(5 + it), total_batch, h, w)
        for k, v in batch.items():
            eng.host_array(k)[...] = v[rank * per:(rank + 1) * per]
        losses.append(eng.step(seed=50 + it)["loss"])
    out = eng.download_params()
    grads = eng.download_grads()      # the last step's gradient buffer AFTER the all-reduce (summed over the ranks, before the 1/G of the update)
    eng.close()
    cp.close()
    keep = ("conv1/7x7_s2", "inception_4c/3x3", "bbox/regressor")
    q.put((rank, losses, {k: [a.copy() for a in v] for k, v in out.items() if k in keep},
           {k: [a.copy() for a in v] for k, v in grads.items() if k in keep}))


def _run(world, use_rccl, steps=2, devices=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL's intra-node transport needs on this driver
    ps = [ctx.Process(target=_train, args=(r, world, port, use_rccl, steps, q, devices)) for r in range(world)]
    for p in ps:
        p.start()
    import queue
    import time
    got, deadline = [], time.time() + 300
    while len(got) < world:
        try:
            got.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
            assert not dead, "a rank process died with exit code %s" % dead      # fail now, not after the timeout
            assert time.time() < deadline, "ranks did not finish in time"
    res = sorted(got, key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_rccl_single_rank_and_two_rank_semantics(gpu):
    single = _run(1, use_rccl=False)[0]
    rccl1 = _run(1, use_rccl=True)[0]          # RCCL communicator of size 1: bucketed all-reduce on the side stream
    assert rccl1[1] == single[1]
    for k in single[2]:
        for a, b in zip(single[2][k], rccl1[2][k]):
            assert np.array_equal(a, b)
    two = _run(2, use_rccl=False)              # 2 ranks x batch 2 == 1 rank x batch 4
    for k in single[2]:
        for a, b, c in zip(single[2][k], two[0][2][k], two[1][2][k]):
            assert np.array_equal(b, c), k                      # replicas stay identical
            assert rel_err(b, a) < 1e-3, k                      # and equal the undivided batch
    # the mean of the ranks' losses is the loss of the undivided batch
    for it in range(len(single[1])):
        assert abs(0.5 * (two[0][1][it] + two[1][1][it]) - single[1][it]) < 1e-3 * abs(single[1][it])


def test_rccl_two_ranks_on_two_devices(gpu):
    """BASELINE configs[3] in small: two rank processes on two DIFFERENT GPUs, gradients summed by ncclAllReduce (RCCL over
    xGMI) in buckets on the side stream.  Replicas must stay bit-identical and match the undivided batch within 1e-3.
    Skipped on a one-GPU box (the development box); the round-end node has eight."""
    import ctypes
    from fcn_object_detector_amd import lib as L
    n = ctypes.c_int(0)
    L.call("fcn_device_count", ctypes.byref(n))
    if n.value < 2:
        pytest.skip("needs two GPUs (found %d)" % n.value)
    single = _run(1, use_rccl=False)[0]
    two = _run(2, use_rccl=True, devices=(0, 1))
    for k in single[2]:
        for a, b, c in zip(single[2][k], two[0][2][k], two[1][2][k]):
            assert np.array_equal(b, c), k                      # replicas stay identical
            assert rel_err(b, a) < 1e-3, k                      # and equal the undivided batch
    for it in range(len(single[1])):
        assert abs(0.5 * (two[0][1][it] + two[1][1][it]) - single[1][it]) < 1e-3 * abs(single[1][it])
    # the same two ranks on the same two GPUs with the gradients summed through host memory in rank order: with two ranks a sum
    # has one order (a + b = b + a in floating point), so ncclAllReduce over the fixed buckets must give the SAME bits - in the
    # gradient buffer of the last step and in every weight after it
    host = _run(2, use_rccl=False, devices=(0, 1))
    for r in range(2):
        assert two[r][1] == host[r][1]                          # losses, step by step
        for k in host[r][3]:
            for a, b in zip(two[r][3][k], host[r][3][k]):
                assert np.array_equal(a, b), ("gradient", k)
            for a, b in zip(two[r][2][k], host[r][2][k]):
                assert np.array_equal(a, b), ("weights", k)
