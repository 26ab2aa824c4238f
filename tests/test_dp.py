"""World-size-2 (and 3) control-plane tests on CPU: rendezvous, barrier, all_gather/max, sharding (no GPU, no torch)."""
import multiprocessing as mp
import os
import socket

from fcn_object_detector_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    cp = dp.ControlPlane(rank, world, "127.0.0.1", port, token="t:%d" % world, timeout=30)
    vals = cp.all_gather({"rank": rank, "sq": rank * rank})
    cp.barrier()
    m = cp.max(10.0 - rank)
    s = cp.sum(rank + 1)
    b = cp.broadcast("id-from-0" if rank == 0 else None)
    lo, hi = dp.shard_range(10, rank, world)
    cp.close()
    q.put((rank, vals, m, s, b, (lo, hi)))


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in range(world))
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    return res


def test_two_ranks():
    res = _run(2)
    for rank, vals, m, s, b, rng in res:
        assert vals == [{"rank": 0, "sq": 0}, {"rank": 1, "sq": 1}]
        assert m == 10.0 and s == 3.0 and b == "id-from-0"
    assert [r[5] for r in res] == [(0, 5), (5, 10)]


def test_three_ranks_uneven_shards():
    res = _run(3)
    assert [r[5] for r in res] == [(0, 4), (4, 7), (7, 10)]
    assert all(r[3] == 6.0 for r in res)


def test_single_rank_is_local():
    cp = dp.ControlPlane(0, 1)
    assert cp.all_gather(5) == [5] and cp.max(2.5) == 2.5
    cp.barrier()
    assert dp.shard_range(7, 0, 1) == (0, 7)
