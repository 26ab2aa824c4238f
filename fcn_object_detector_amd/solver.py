"""Solver front end: what ``caffe train --solver=... [--weights=...|--snapshot=...] --gpu=...`` does
(reference: train/train.sh:25-28, with the solver.prototxt files under models/), on top of TrainEngine.

Mirrors Caffe's Solver: Step() = forward/backward + update, ``display`` lines with the smoothed loss and the
net outputs, ``snapshot`` every N iterations as ``<prefix>_iter_<N>.caffemodel`` + ``.solverstate``, resume from a
``.solverstate``.  Multi-GPU (new: the reference is single-GPU) is data parallel, one process per GPU, gradients
summed with RCCL by TrainEngine.
"""
from __future__ import annotations

import os
import sys
import time
from collections import OrderedDict, deque
from typing import Callable, Dict, List, Optional

import numpy as np

from . import lib as L
from . import proto
from . import pylayer
from .netspec import NetSpec, fill_params
from .train import SolverParams, TrainEngine


def _resolve(path: str, anchor: str) -> str:
    """Caffe resolves paths against the working directory; fall back to the directory of the file that names them."""
    if os.path.isabs(path) or os.path.isfile(path):
        return path
    alt = os.path.join(os.path.dirname(os.path.abspath(anchor)), path)
    return alt if os.path.isfile(alt) else path


def _log(msg: str) -> None:
    sys.stderr.write(time.strftime("I%m%d %H:%M:%S ") + msg + "\n")
    sys.stderr.flush()


class _BlobView(object):
    def __init__(self, eng: TrainEngine, name: str):
        self._eng, self._name = eng, name

    @property
    def data(self) -> np.ndarray:
        return self._eng.read_blob(self._name)

    @property
    def diff(self) -> np.ndarray:
        return self._eng.read_grad(self._name)

    @property
    def shape(self):
        return tuple(self._eng.shapes[self._name])


class _ParamView(object):
    def __init__(self, solver: "Solver", layer: str, index: int):
        self._s, self._layer, self._index = solver, layer, index

    @property
    def data(self) -> np.ndarray:
        return self._s.engine.download_params()[self._layer][self._index]

    @property
    def diff(self) -> np.ndarray:
        return self._s.engine.download_grads()[self._layer][self._index]


class TrainNet(object):
    """``solver.net``: read access to blobs / params of the training net, ``copy_from`` and ``save``."""

    def __init__(self, solver: "Solver"):
        self._s = solver
        eng = solver.engine
        self.blobs = OrderedDict((n, _BlobView(eng, n)) for n in eng.shapes)
        self.params = OrderedDict((l.name, [_ParamView(solver, l.name, i) for i in range(len(eng.params_host[l.name]))])
                                  for l in eng.spec.param_layers())

    def copy_from(self, weights_path: str) -> None:
        """Net::CopyTrainedLayersFrom: by layer name; layers absent from the file keep their filler values."""
        if not os.path.isfile(weights_path):
            raise IOError("weights file not found: %s" % weights_path)
        eng = self._s.engine
        proto.copy_trained_layers(weights_path, eng.params_host, eng.set_params, log=self._s.log)

    def save(self, path: str) -> None:
        self._s.engine.save(path)


class Solver(object):
    def __init__(self, solver_file: str, device: int = 0, comm=None, rank: int = 0, log: Optional[Callable[[str], None]] = _log,
                 autotune: bool = True):
        if not os.path.isfile(str(solver_file)):
            raise IOError("solver file not found: %s" % solver_file)
        self.solver_file = str(solver_file)
        self.param = SolverParams(proto.parse_file(self.solver_file))
        if not self.param.net:
            raise ValueError("%s: solver needs a `net:` (or `train_net:`) entry" % solver_file)
        self.net_file = _resolve(str(self.param.net), self.solver_file)
        if not os.path.isfile(self.net_file):
            raise IOError("net file not found: %s" % self.param.net)
        self.rank, self.comm, self.device = int(rank), comm, int(device)
        self.log = log if (log is not None and self.rank == 0) else (lambda m: None)
        L.call("fcn_init", self.device)
        msg = proto.parse_file(self.net_file)
        spec = NetSpec(msg, "TRAIN")
        self.py_layers, data_shapes = pylayer.setup_python_layers(spec, pylayer.TRAIN)
        spec = NetSpec(msg, "TRAIN")
        spec.infer({**spec.input_shapes, **data_shapes})
        self.engine = TrainEngine(spec, data_shapes, fill_params(spec, seed=0), device=self.device, solver=self.param, comm=comm,
                                  autotune=autotune)
        self.net = TrainNet(self)
        self._losses: deque = deque(maxlen=max(self.param.average_loss, 1))
        self._fed = False
        self._device_label_tops = {}
        for l, inst, bottoms, tops in self.py_layers:
            # a data layer that can render its scenes on the device writes `data` (and HEAD's class mask) straight into HBM
            if getattr(inst, "supports_device_scenes", False):
                inst.bind_device(self.engine, [t.name for t in tops])
                self.engine.device_fed |= set(inst.device_tops)
            # a data layer that can hand over ground-truth boxes gets its label grids generated in HBM (fcn_gen_targets_nhwc)
            if getattr(inst, "supports_device_targets", False) and getattr(inst, "mode", None) == "detectnet" and len(tops) >= 6:
                inst.device_targets = True
                self._device_label_tops[id(inst)] = tuple(t.name for t in tops[1:6])
        self.log("Solver: %s, net %s, %d learnable floats, world %d" % (
            self.param.kind, self.net_file, self.engine.param_count, comm.world if comm is not None else 1))

    @property
    def iter(self) -> int:
        return self.engine.iter

    # ------------------------------------------------------------------ Solver::Step
    def _feed(self) -> None:
        eng = self.engine
        for l, inst, bottoms, tops in self.py_layers:
            inst.reshape(bottoms, tops)
            inst.forward(bottoms, tops)
            label_tops = self._device_label_tops.get(id(inst), ()) if getattr(inst, "device_targets", False) else ()
            for t in tops:
                if tuple(t.shape_) != tuple(eng.shapes[t.name]):
                    raise NotImplementedError("Python layer %s changed the shape of %s" % (l.name, t.name))
                if t.name not in label_tops and t.name not in getattr(inst, "device_tops", ()):
                    eng.host_array(t.name)[...] = t.data
            if label_tops:
                eng.set_targets(inst.last_rects, inst.last_labels, inst.stride, tops=label_tops)

    def _all_device_fed(self) -> bool:
        """True when no top of any Python layer travels through a host array (scenes composed and labels generated in HBM):
        the next batch can then be planned and enqueued while the device is still busy with the current iteration."""
        for l, inst, bottoms, tops in self.py_layers:
            label_tops = self._device_label_tops.get(id(inst), ()) if getattr(inst, "device_targets", False) else ()
            if any(t.name not in label_tops and t.name not in getattr(inst, "device_tops", ()) for t in tops):
                return False
        return bool(self.py_layers)

    def step(self, iters: int = 1, pipeline: bool = False) -> Dict[str, float]:
        """Solver::Step.  pipeline=True (what solve() / `caffe train` use) plans and enqueues the next batch while the device
        runs the current iteration; it is off by default because the input blobs then already hold the NEXT batch when
        step() returns, which a pycaffe caller inspecting net.blobs would not expect."""
        p, out = self.param, {}
        stop = self.iter + int(iters)
        pipelined = pipeline and self._all_device_fed()
        while self.iter < stop:
            it = self.iter
            if not self._fed:
                self._feed()
            self._fed = False
            self.engine.step_begin()
            if pipelined and it + 1 < p.max_iter:
                # the renders of the next batch queue up behind this iteration on the stream; its planning (host RNG and
                # box bookkeeping) overlaps the device work
                self._feed()
                self._fed = True
            out = self.engine.step_end()
            self._losses.append(out["total_loss"])
            if p.display and it % p.display == 0:
                self.log("Iteration %d, loss = %g" % (it, sum(self._losses) / len(self._losses)))
                for j, (name, w) in enumerate(self.engine.loss_blobs.items()):
                    self.log("    Train net output #%d: %s = %g (* %g = %g loss)" % (j, name, out[name], w, w * out[name]))
                self.log("Iteration %d, lr = %g" % (it, p.rate(it)))
            if not np.isfinite(out["total_loss"]):
                raise FloatingPointError("loss is %r at iteration %d" % (out["total_loss"], it))
            if p.snapshot and self.iter % p.snapshot == 0:
                self.snapshot()
        return out

    def solve(self, resume_file: Optional[str] = None) -> None:
        if resume_file:
            self.restore(resume_file)
        self.log("Solving %s" % (self.engine.spec.name or os.path.basename(self.net_file)))
        self.step(max(self.param.max_iter - self.iter, 0), pipeline=True)
        if not (self.param.snapshot and self.iter % self.param.snapshot == 0):
            self.snapshot()
        self.log("Optimization Done.")

    # ------------------------------------------------------------------ snapshots
    def snapshot_filename(self, ext: str) -> str:
        return "%s_iter_%d%s" % (self.param.snapshot_prefix, self.iter, ext)

    def snapshot(self) -> Optional[str]:
        if self.rank != 0:
            return None
        model, state = self.snapshot_filename(".caffemodel"), self.snapshot_filename(".solverstate")
        d = os.path.dirname(model)
        if d:
            os.makedirs(d, exist_ok=True)
        self.log("Snapshotting to binary proto file %s" % model)
        self.engine.save(model)
        self.log("Snapshotting solver state to binary proto file %s" % state)
        with open(state, "wb") as f:
            f.write(proto.pack_solverstate(self.iter, self.engine.download_history(), learned_net=model))
        return model

    def restore(self, state_file: str) -> None:
        if not os.path.isfile(state_file):
            raise IOError("solver state not found: %s" % state_file)
        with open(state_file, "rb") as f:
            it, history, learned = proto.unpack_solverstate(f.read(), with_learned_net=True)
        if learned:
            self.net.copy_from(_resolve(learned, state_file))
        self.engine.upload_history(history)
        self.engine.iter = int(it)
        self.log("Restoring previous solver status from %s (iteration %d)" % (state_file, it))

    def close(self) -> None:
        self.engine.close()


SGDSolver = Solver
AdamSolver = Solver


def get_solver(solver_file: str, **kw) -> Solver:
    return Solver(solver_file, **kw)
