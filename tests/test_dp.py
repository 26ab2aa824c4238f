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


def _rank0_with_intruder(port, q):
    cp = dp.ControlPlane(0, 2, "127.0.0.1", port, token="t:2", timeout=30, secret="s3cret")
    q.put(("gathered", cp.all_gather("zero")))
    cp.close()


def test_unauthenticated_peers_are_dropped_before_anything_is_parsed():
    """A peer that does not know the job secret (wrong MAC, a pickle bomb, a huge length prefix) never reaches the
    message parser; the real rank 1 still joins afterwards."""
    import pickle
    import struct
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    p0 = ctx.Process(target=_rank0_with_intruder, args=(port, q))
    p0.start()
    deadline = time.time() + 20
    intruded = 0
    payloads = [struct.pack("!I", 1) + b"\0" * 32,                                   # right shape, wrong MAC
                struct.pack("!Q", 1 << 40) + pickle.dumps(("t:2", 1, "x" * 64)),    # the old pickle hello with a giant length
                b"\xff" * 36]
    while intruded < len(payloads) and time.time() < deadline:
        for off in dp.ControlPlane.PORT_OFFSETS:
            try:
                s = socket.create_connection(("127.0.0.1", port + off), timeout=1.0)
            except OSError:
                continue
            s.settimeout(5.0)
            assert len(s.recv(16)) == 16                                             # the nonce
            s.sendall(payloads[intruded])
            try:
                assert s.recv(64) == b""                                             # dropped: no "srv" proof, no data
            except ConnectionResetError:
                pass
            s.close()
            intruded += 1
            break
        else:
            time.sleep(0.1)
    assert intruded == len(payloads)
    cp = dp.ControlPlane(1, 2, "127.0.0.1", port, token="t:2", timeout=30, secret="s3cret")
    assert cp.all_gather("one") == ["zero", "one"]
    cp.close()
    assert q.get(timeout=30) == ("gathered", ["zero", "one"])
    p0.join(30)
    assert p0.exitcode == 0


def test_wire_format_is_json_with_bytes_and_a_size_cap():
    import pytest
    a, b = socket.socketpair()
    key = b"k" * 32
    msg = {"uid": bytes(range(128)), "t": [1.5, None, "x"], "n": 3}
    dp._send(a, msg, key)
    assert dp._recv(b, key) == msg
    dp._send(a, "tampered", key)
    with pytest.raises(ConnectionError, match="authentication"):
        dp._recv(b, b"another key".ljust(32, b"."))
    with pytest.raises(ValueError, match="exceeds"):
        dp._send(a, "x" * (dp.MAX_MESSAGE + 1), key)
    with pytest.raises(TypeError):
        dp._send(a, object(), key)
    a.close()
    b.close()


def test_spawn_ranks_starts_one_process_per_device(tmp_path):
    """What `bench.py --gpus N` and `caffe train --gpu=a,b` do before touching a GPU: N fresh processes that find each other
    through the environment the launcher hands them (no GPU involved here: the rank script only uses the control plane)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "rank.py"
    out = tmp_path / "out.json"
    script.write_text(
        "import json, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from fcn_object_detector_amd import dp\n"
        "assert dp.launched_as_rank() and os.environ['FCN_DP_SECRET'] and os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'\n"
        "cp = dp.ControlPlane()\n"
        "got = cp.all_gather({'rank': dp.env_rank(), 'device': dp.env_device(), 'arg': sys.argv[1]})\n"
        "cp.barrier()\n"
        "if dp.env_rank() == 0:\n"
        "    json.dump(got, open(%r, 'w'))\n"
        "cp.close()\n"
        "sys.exit(3 if sys.argv[1] == 'fail' and dp.env_rank() == 1 else 0)\n" % (ROOT, str(out)))
    code = ("import sys; sys.path.insert(0, %r); from fcn_object_detector_amd import dp; "
            "sys.exit(dp.spawn_ranks(%r, [sys.argv[1]], [5, 2, 7]))" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FCN_DP_SECRET")}
    assert subprocess.run([sys.executable, "-c", code, "hello"], env=env, timeout=120).returncode == 0
    assert json.load(open(out)) == [{"rank": r, "device": d, "arg": "hello"} for r, d in enumerate((5, 2, 7))]
    assert subprocess.run([sys.executable, "-c", code, "fail"], env=env, timeout=120).returncode == 3      # a rank's failure is the job's


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_spawn_ranks_ends_the_job_when_one_rank_dies(tmp_path):
    """A rank that exits non-zero while the others are blocked (here: in a barrier that the dead rank never joins, standing in
    for an ncclAllReduce without a timeout) must not leave the launcher waiting for rank 0: the survivors are terminated and
    the launcher returns the failure promptly."""
    import subprocess
    import sys
    import time
    from conftest import ROOT
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from fcn_object_detector_amd import dp\n"
        "cp = dp.ControlPlane(timeout=300)\n"
        "cp.barrier()\n"
        "if dp.env_rank() == 1:\n"
        "    os._exit(7)\n"           # dies without closing its sockets cleanly
        "time.sleep(600)\n" % ROOT)     # the others: blocked for ten minutes
    code = ("import sys; sys.path.insert(0, %r); from fcn_object_detector_amd import dp; "
            "sys.exit(dp.spawn_ranks(%r, [], [0, 1, 2]))" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FCN_DP_SECRET")}
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, timeout=120)
    assert r.returncode == 7 and time.time() - t0 < 60


def test_spawn_ranks_deadline(tmp_path):
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "rank.py"
    script.write_text("import time\ntime.sleep(600)\n")
    code = ("import sys; sys.path.insert(0, %r); from fcn_object_detector_amd import dp; "
            "sys.exit(dp.spawn_ranks(%r, [], [0, 1]))" % (ROOT, str(script)))
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FCN_DP_SECRET")}, FCN_SPAWN_TIMEOUT="2")
    assert subprocess.run([sys.executable, "-c", code], env=env, timeout=60).returncode == 124


def _worker_env_addr(rank, world, port, master_addr, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR=master_addr, MASTER_PORT=str(port))
    os.environ.pop("FCN_DP_BIND", None)
    cp = dp.ControlPlane(timeout=30)      # address and port from the environment, as under torch.distributed.run
    q.put((rank, cp.all_gather(rank), cp.addr))
    cp.close()


def test_single_node_job_with_master_addr_set_to_the_host_name():
    """`torch.distributed.run --standalone` sets MASTER_ADDR to the host's name (which resolves to 127.0.1.1 or a NIC address);
    rank 0 listens on the loopback interface, so on a single-node job the other ranks must connect there as well."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    name = socket.getfqdn()
    ps = [ctx.Process(target=_worker_env_addr, args=(r, 2, port, name, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in range(2))
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [[0, 1], [0, 1]] and all(r[2] == "127.0.0.1" for r in res)


def test_multi_node_master_addr_needs_an_explicit_bind(monkeypatch):
    import pytest
    monkeypatch.delenv("FCN_DP_BIND", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setenv("MASTER_ADDR", "203.0.113.7")      # (TEST-NET-3: never this host)
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    with pytest.raises(RuntimeError, match="FCN_DP_BIND"):
        dp.ControlPlane(0, 2, timeout=1)


def test_multi_node_job_on_the_master_node_itself_needs_an_explicit_bind(monkeypatch):
    """Rank 0's own node of a two-node job: MASTER_ADDR names THIS host, but LOCAL_WORLD_SIZE (1) differs from WORLD_SIZE (2), so
    the job is not single-node and rank 0 must not bind the loopback interface silently (the remote rank would time out)."""
    import pytest
    monkeypatch.delenv("FCN_DP_BIND", raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    monkeypatch.setenv("MASTER_ADDR", socket.getfqdn())
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    assert dp._single_node_job(2, socket.getfqdn()) is False
    assert dp._single_node_job(1, socket.getfqdn()) is True
    with pytest.raises(RuntimeError, match="FCN_DP_BIND"):
        dp.ControlPlane(0, 2, timeout=1)
