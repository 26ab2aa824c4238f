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


def _train(rank, world, port, use_rccl, steps, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fcn_object_detector_amd import dp, models, proto
    from fcn_object_detector_amd.netspec import NetSpec, fill_params
    from fcn_object_detector_amd.train import SolverParams, TrainEngine
    from test_gpu_train import make_batch
    cp = dp.ControlPlane(rank, world, "127.0.0.1", port, token="dp:%d" % world, timeout=120)
    if use_rccl:
        comm = dp.RcclComm(cp, 0)
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
    eng = TrainEngine(NetSpec(msg, "TRAIN"), shapes, params=params, device=0, comm=comm, autotune=False,
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
    eng.close()
    cp.close()
    q.put((rank, losses, {k: [a.copy() for a in v] for k, v in out.items() if k in ("conv1/7x7_s2", "inception_4c/3x3", "bbox/regressor")}))


def _run(world, use_rccl, steps=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_train, args=(r, world, port, use_rccl, steps, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
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
