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

import os
import pickle
import socket
import struct
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


def _send(sock: socket.socket, obj: Any) -> None:
    data = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack("!Q", len(data)) + data)


def _recv(sock: socket.socket) -> Any:
    hdr = b""
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        hdr += chunk
    (n,) = struct.unpack("!Q", hdr)
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return pickle.loads(bytes(buf))


class ControlPlane:
    """Star-topology collectives over TCP on one node: rank 0 serves, the others connect."""

    PORT_OFFSETS = tuple(range(1, 33))

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, addr: Optional[str] = None,
                 base_port: Optional[int] = None, token: Optional[str] = None, timeout: float = 120.0):
        self.rank = env_rank() if rank is None else int(rank)
        self.world = env_world_size() if world is None else int(world)
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.base_port = int(base_port if base_port is not None else os.environ.get("MASTER_PORT", "29500"))
        self.token = token or os.environ.get("TORCHELASTIC_RUN_ID", "fcn") + ":%d" % self.world
        self.timeout = timeout
        self.peers: List[socket.socket] = []
        self.sock: Optional[socket.socket] = None
        if self.world > 1:
            self._connect()

    def _connect(self) -> None:
        deadline = time.time() + self.timeout
        if self.rank == 0:
            srv = None
            for off in self.PORT_OFFSETS:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((self.addr if self.addr not in ("localhost",) else "127.0.0.1", self.base_port + off))
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
                conn.settimeout(self.timeout)
                try:
                    hello = _recv(conn)
                except Exception:
                    conn.close()
                    continue
                if not (isinstance(hello, tuple) and len(hello) == 2 and hello[0] == self.token and 0 < hello[1] < self.world
                        and slots[hello[1]] is None):
                    _send(conn, "reject")
                    conn.close()
                    continue
                _send(conn, "ok")
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                slots[hello[1]] = conn
                got += 1
            srv.close()
            self.peers = [s for s in slots[1:]]
        else:
            last_err: Optional[Exception] = None
            while time.time() < deadline and self.sock is None:
                for off in self.PORT_OFFSETS:
                    try:
                        s = socket.create_connection((self.addr, self.base_port + off), timeout=2.0)
                        s.settimeout(self.timeout)
                        _send(s, (self.token, self.rank))
                        if _recv(s) == "ok":
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self.sock = s
                            break
                        s.close()
                    except Exception as e:  # not up yet / someone else's port
                        last_err = e
                if self.sock is None:
                    time.sleep(0.2)
            if self.sock is None:
                raise RuntimeError("control plane: rank %d could not reach rank 0 (%s)" % (self.rank, last_err))

    def all_gather(self, obj: Any) -> List[Any]:
        """Every rank contributes one picklable object; every rank gets the list ordered by rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            vals = [obj] + [_recv(p) for p in self.peers]
            for p in self.peers:
                _send(p, vals)
            return vals
        _send(self.sock, obj)
        return _recv(self.sock)

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
