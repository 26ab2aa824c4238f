"""TEST-ONLY communicator: sums the gradient buffer of several rank processes through host memory over the TCP control
plane.  It lets two ranks that share the ONE GPU of the test box exercise the data-parallel step end to end (RCCL refuses
two ranks on one device); the product path uses fcn_object_detector_amd.dp.RcclComm."""
import numpy as np

from fcn_object_detector_amd import lib as L


class HostComm:
    def __init__(self, cp):
        self.cp, self.world, self.rank = cp, cp.world, cp.rank

    def all_reduce_sum(self, ptr, count, stream):
        buf = np.empty(count, np.float32)
        L.call("fcn_memcpy_d2h_async", buf.ctypes.data, ptr, buf.nbytes, stream)
        L.call("fcn_stream_sync", stream)
        parts = self.cp.all_gather(buf)
        total = parts[0].copy()
        for p in parts[1:]:
            total += p
        self._keep = total
        L.call("fcn_memcpy_h2d_async", ptr, total.ctypes.data, total.nbytes, stream)
        L.call("fcn_stream_sync", stream)
