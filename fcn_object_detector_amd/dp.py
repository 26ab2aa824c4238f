"""One-process-per-GPU plumbing: rank discovery and a tiny TCP control plane.

The reference trains and serves on a single GPU (``--gpu=0``, reference: train/train.sh:26;
``device_id`` param, scripts/fcn_object_detector.py:38); running on the 8 GPUs of an MI355X node
is new.  Ranks are launched by ``python -m torch.distributed.run`` (which only sets RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) — the rank processes themselves stay
PyTorch-free so that only ONE HIP runtime (the system ROCm that libfcnhip.so links) lives in the
process.  The data plane (gradient all-reduce) is RCCL inside libfcnhip.so; this module is the
control plane: barrier, small all-gathers (timings, the RCCL unique id) over localhost TCP.
"""
from __future__ import annotations

import hashlib
import hmac
import json
import os
import socket
import struct
import sys
import time
from typing import Any, List, Optional


def env_rank() -> int:
    return int(os.environ.get("RANK", "0"))


def env_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))


def env_world_size() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total: int, rank: int, world: int):
    """Contiguous, balanced split of ``total`` units: the first ``total % world`` ranks get one extra."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def launched_as_rank() -> bool:
    """True inside a rank process (started by spawn_ranks() or by `python -m torch.distributed.run`)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def spawn_ranks(script: str, argv: List[str], devices: List[int], extra_env: Optional[dict] = None) -> int:
    """Start one fresh process per GPU (`script argv...`, rank i on devices[i]) and wait for all of them; returns the
    first non-zero exit code.  MUST be called before the calling process has touched the GPU - it never does afterwards
    either: the launcher only waits.  Children find each other through RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT (what torch.distributed.run would set), FCN_DEVICE (the GPU of the rank) and a per-job FCN_DP_SECRET."""
    import secrets
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    secret = secrets.token_hex(16)
    procs = []
    for rank, dev in enumerate(devices):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(len(devices)), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FCN_DEVICE=str(dev), FCN_DP_SECRET=secret)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL's intra-node transport needs on this driver
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))
    # Watch ALL children: a rank that dies (bad weights file, out of memory, RCCL init error) leaves the others blocked in a
    # collective that has no timeout, and a launcher that waits for rank 0 first would wait forever.  The first non-zero exit -
    # or $FCN_SPAWN_TIMEOUT seconds, if set - ends the job: the ranks still running are terminated (then killed) and the launcher
    # returns non-zero.  Children are only ever signalled, never re-executed.
    deadline = None
    if os.environ.get("FCN_SPAWN_TIMEOUT"):
        deadline = time.time() + float(os.environ["FCN_SPAWN_TIMEOUT"])
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [c for c in codes if c not in (None, 0)]
            if failed:
                rc = failed[0]
                break
            if all(c == 0 for c in codes):
                break
            if deadline is not None and time.time() > deadline:
                rc = 124
                sys.stderr.write("spawn_ranks: ranks still running after FCN_SPAWN_TIMEOUT; ending the job\n")
                break
            time.sleep(0.05)
        if rc != 0:
            alive = [p for p in procs if p.poll() is None]
            for p in alive:
                p.terminate()
            t_end = time.time() + 5.0
            while alive and time.time() < t_end:
                alive = [p for p in alive if p.poll() is None]
                time.sleep(0.05)
            for p in alive:
                p.kill()
            for p in procs:
                p.wait()
    except BaseException:
        for p in procs:      # exactly the processes started here
            if p.poll() is None:
                p.kill()
        raise
    return rc


def env_device() -> int:
    """GPU of this rank: FCN_DEVICE when a launcher of this package set it, else LOCAL_RANK."""
    return int(os.environ.get("FCN_DEVICE", env_local_rank()))


def _single_node_job(world: int, master_addr: str) -> bool:
    """Every rank on this host?  LOCAL_WORLD_SIZE == WORLD_SIZE (torch.distributed.run sets both), or MASTER_ADDR names this host."""
    lws = os.environ.get("LOCAL_WORLD_SIZE")
    if lws is not None:
        # a launcher that sets it has said how many ranks share this host: fewer than WORLD_SIZE is a multi-node job even on the
        # node whose name MASTER_ADDR is (rank 0's own node: the name test below would call that job single-node)
        return int(lws) == int(world)
    if master_addr in ("127.0.0.1", "localhost", "::1"):
        return True
    try:
        mine = {socket.gethostname(), socket.getfqdn()}
        if master_addr in mine:
            return True
        addrs = {ai[4][0] for ai in socket.getaddrinfo(master_addr, None)}
        local = {"127.0.0.1", "127.0.1.1", "::1"}
        for name in mine:
            try:
                local |= {ai[4][0] for ai in socket.getaddrinfo(name, None)}
            except OSError:
                pass
        return bool(addrs & local)
    except OSError:
        return False


MAX_MESSAGE = 1 << 20      # control-plane messages are ranks, timings, error strings and the 128-byte RCCL id
_MAC = 32


def _to_wire(obj: Any) -> Any:
    if isinstance(obj, (bytes, bytearray)):
        return {"__bytes__": bytes(obj).hex()}
    if isinstance(obj, (list, tuple)):
        return [_to_wire(v) for v in obj]
    if isinstance(obj, dict):
        return {str(k): _to_wire(v) for k, v in obj.items()}
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    raise TypeError("control plane: cannot send a %s (only None, bool, int, float, str, bytes, list, dict)" % type(obj).__name__)


def _from_wire(obj: Any) -> Any:
    if isinstance(obj, dict):
        if set(obj) == {"__bytes__"}:
            return bytes.fromhex(obj["__bytes__"])
        return {k: _from_wire(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_from_wire(v) for v in obj]
    return obj


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 16, n - len(buf)))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send(sock: socket.socket, obj: Any, key: bytes) -> None:
    """One message = u32 length | HMAC-SHA256(key, payload) | JSON payload.  Nothing on this wire is ever unpickled."""
    data = json.dumps(_to_wire(obj), separators=(",", ":"), allow_nan=True).encode("utf-8")
    if len(data) > MAX_MESSAGE:
        raise ValueError("control plane: message of %d bytes exceeds the %d-byte limit" % (len(data), MAX_MESSAGE))
    sock.sendall(struct.pack("!I", len(data)) + hmac.new(key, data, hashlib.sha256).digest() + data)


def _recv(sock: socket.socket, key: bytes) -> Any:
    (n,) = struct.unpack("!I", _recv_exact(sock, 4))
    if n > MAX_MESSAGE:
        raise ConnectionError("control plane: peer announced a %d-byte message (limit %d)" % (n, MAX_MESSAGE))
    mac = _recv_exact(sock, _MAC)
    data = _recv_exact(sock, n)
    if not hmac.compare_digest(mac, hmac.new(key, data, hashlib.sha256).digest()):
        raise ConnectionError("control plane: message authentication failed")
    return _from_wire(json.loads(data.decode("utf-8")))


class ControlPlane:
    """Star-topology collectives over TCP on one node: rank 0 serves, the others connect.

    Framing is fixed-size headers + JSON (never pickle).  A connection is authenticated before anything it sends is
    parsed: rank 0 sends a random nonce, the peer answers with its rank and HMAC-SHA256(secret, nonce | rank), rank 0
    proves itself with HMAC(secret, nonce | "srv"); every later message carries its own MAC.  The secret is $FCN_DP_SECRET
    (bench.py and the `caffe` tool generate one per job for the ranks they start); without it the job token is used, which
    is NOT a secret - so rank 0 listens on the loopback interface only unless $FCN_DP_BIND names another address."""

    PORT_OFFSETS = tuple(range(1, 33))

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, addr: Optional[str] = None,
                 base_port: Optional[int] = None, token: Optional[str] = None, timeout: float = 120.0, secret: Optional[str] = None):
        self.rank = env_rank() if rank is None else int(rank)
        self.world = env_world_size() if world is None else int(world)
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        # Rank 0 listens on $FCN_DP_BIND, by default the loopback interface.  `torch.distributed.run --standalone` sets MASTER_ADDR to
        # the host's name, which resolves to 127.0.1.1 or a NIC address where nobody listens: on a single-node job (every rank on
        # this host) the clients therefore connect to the loopback address too.  A multi-node job must name the interface.
        if addr is None and not os.environ.get("FCN_DP_BIND"):
            if _single_node_job(self.world, self.addr):
                self.addr = "127.0.0.1"
            elif self.world > 1 and self.rank == 0:
                raise RuntimeError("control plane: MASTER_ADDR=%s is not this host and FCN_DP_BIND is unset: rank 0 would listen on "
                                   "127.0.0.1 where the other nodes cannot reach it.  Set FCN_DP_BIND to the interface to listen on "
                                   "(and FCN_DP_SECRET to a per-job secret shared by all ranks)." % self.addr)
        self.base_port = int(base_port if base_port is not None else os.environ.get("MASTER_PORT", "29500"))
        self.token = token or os.environ.get("TORCHELASTIC_RUN_ID", "fcn") + ":%d" % self.world
        secret = secret if secret is not None else os.environ.get("FCN_DP_SECRET")
        self.key = hashlib.sha256(("fcn-dp|%s|%s" % (self.token, secret or "")).encode("utf-8")).digest()
        self.timeout = timeout
        self.peers: List[socket.socket] = []
        self.sock: Optional[socket.socket] = None
        if self.world > 1:
            self._connect()

    def _hello_mac(self, nonce: bytes, who: bytes) -> bytes:
        return hmac.new(self.key, nonce + who, hashlib.sha256).digest()

    def _connect(self) -> None:
        deadline = time.time() + self.timeout
        if self.rank == 0:
            bind_addr = os.environ.get("FCN_DP_BIND", "127.0.0.1")
            srv = None
            for off in self.PORT_OFFSETS:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((bind_addr, self.base_port + off))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("control plane: no free port near %d" % self.base_port)
            srv.listen(self.world)
            slots: List[Optional[socket.socket]] = [None] * self.world
            got = 0
            while got < self.world - 1:
                srv.settimeout(max(deadline - time.time(), 0.1))
                conn, _ = srv.accept()
                conn.settimeout(min(self.timeout, 10.0))
                try:
                    nonce = os.urandom(16)
                    conn.sendall(nonce)
                    reply = _recv_exact(conn, 4 + _MAC)      # fixed size: nothing is parsed before the MAC checks out
                    (r,) = struct.unpack("!I", reply[:4])
                    ok = hmac.compare_digest(reply[4:], self._hello_mac(nonce, reply[:4])) and 0 < r < self.world and slots[r] is None
                    if not ok:
                        conn.close()
                        continue
                    conn.sendall(self._hello_mac(nonce, b"srv"))
                except Exception:
                    conn.close()
                    continue
                conn.settimeout(self.timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                slots[r] = conn
                got += 1
            srv.close()
            self.peers = [s for s in slots[1:]]
        else:
            last_err: Optional[Exception] = None
            while time.time() < deadline and self.sock is None:
                for off in self.PORT_OFFSETS:
                    try:
                        s = socket.create_connection((self.addr, self.base_port + off), timeout=2.0)
                        s.settimeout(5.0)
                        nonce = _recv_exact(s, 16)
                        me = struct.pack("!I", self.rank)
                        s.sendall(me + self._hello_mac(nonce, me))
                        if hmac.compare_digest(_recv_exact(s, _MAC), self._hello_mac(nonce, b"srv")):
                            s.settimeout(self.timeout)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self.sock = s
                            break
                        s.close()
                    except Exception as e:  # not up yet / someone else's port / wrong job
                        last_err = e
                if self.sock is None:
                    time.sleep(0.2)
            if self.sock is None:
                raise RuntimeError("control plane: rank %d could not reach rank 0 (%s)" % (self.rank, last_err))

    def all_gather(self, obj: Any) -> List[Any]:
        """Every rank contributes one value (None, bool, int, float, str, bytes, lists / dicts of those); every rank gets
        the list ordered by rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            vals = [obj] + [_recv(p, self.key) for p in self.peers]
            for p in self.peers:
                _send(p, vals, self.key)
            return vals
        _send(self.sock, obj, self.key)
        return _recv(self.sock, self.key)

    def barrier(self) -> None:
        self.all_gather(None)

    def broadcast(self, obj: Any, root: int = 0) -> Any:
        return self.all_gather(obj if self.rank == root else None)[root]

    def max(self, value: float) -> float:
        return max(self.all_gather(float(value)))

    def sum(self, value: float) -> float:
        return sum(self.all_gather(float(value)))

    def close(self) -> None:
        for p in self.peers:
            try:
                p.close()
            except Exception:
                pass
        if self.sock is not None:
            try:
                self.sock.close()
            except Exception:
                pass
        self.peers, self.sock = [], None


class RcclComm:
    """Data plane: RCCL communicator inside libfcnhip.so; the ncclUniqueId travels over the control plane."""

    def __init__(self, cp: ControlPlane, device: int):
        import ctypes as C
        from . import lib as L
        self.cp, self.world, self.rank, self.device = cp, cp.world, cp.rank, device
        L.call("fcn_init", device)
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            L.call("fcn_comm_unique_id", buf)
        uid = cp.broadcast(bytes(buf.raw) if self.rank == 0 else None)
        comm = C.c_void_p()
        L.call("fcn_comm_init", C.byref(comm), C.create_string_buffer(uid, 128), self.world, self.rank)
        self.comm = comm

    def all_reduce_sum(self, ptr: int, count: int, stream) -> None:
        """In-place sum of `count` floats at device address `ptr` over all ranks, enqueued on `stream`."""
        from . import lib as L
        L.call("fcn_comm_allreduce_sum_f32", self.comm, ptr, count, stream)

    def close(self) -> None:
        from . import lib as L
        if self.comm:
            L.load().fcn_comm_destroy(self.comm)
            self.comm = None
