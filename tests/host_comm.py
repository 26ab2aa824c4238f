"""TEST-ONLY communicator: sums the gradient buffer of several rank processes through host memory.  It lets two ranks that
share the ONE GPU of the test box exercise the data-parallel step end to end (RCCL refuses two ranks on one device); the
product path uses fcn_object_detector_amd.dp.RcclComm.  The buffers travel through files in a per-job scratch directory
(the control plane carries small JSON messages only - ranks, timings, the RCCL id - and is used here for the barriers)."""
import hashlib
import os
import tempfile

import numpy as np

from fcn_object_detector_amd import lib as L


class HostComm:
    def __init__(self, cp):
        self.cp, self.world, self.rank = cp, cp.world, cp.rank
        tag = hashlib.sha1(("%s|%d" % (cp.token, cp.base_port)).encode()).hexdigest()[:12]
        self.dir = os.path.join(tempfile.gettempdir(), "fcn_hostcomm_" + tag)
        os.makedirs(self.dir, exist_ok=True)
        self.calls = 0

    def _path(self, rank):
        return os.path.join(self.dir, "c%d_r%d.npy" % (self.calls, rank))

    def all_reduce_sum(self, ptr, count, stream):
        buf = np.empty(count, np.float32)
        L.call("fcn_memcpy_d2h_async", buf.ctypes.data, ptr, buf.nbytes, stream)
        L.call("fcn_stream_sync", stream)
        np.save(self._path(self.rank), buf)
        self.cp.barrier()                               # every rank's part is on disk
        total = np.load(self._path(0))
        for r in range(1, self.world):
            total += np.load(self._path(r))             # rank order: every replica adds in the same order
        self.cp.barrier()                               # everybody has read
        os.remove(self._path(self.rank))
        self.calls += 1
        self._keep = total
        L.call("fcn_memcpy_h2d_async", ptr, total.ctypes.data, total.nbytes, stream)
        L.call("fcn_stream_sync", stream)
