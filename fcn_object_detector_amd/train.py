"""Training step on the device: Net::ForwardBackward + solver update, optionally data-parallel.

Stands in for `caffe train` (reference: train/train.sh:25-28) over the DetectNet training net
(reference: models/train_val.prototxt with the Python data layer's tops, README.md:57-76) and the
solver settings of the reference's solver.prototxt files (SGD with momentum, Adam, fixed / step
learning-rate policy, L2 weight decay, per-blob lr_mult / decay_mult).

Data layout: every blob that receives a gradient has a gradient buffer with the SAME NHWC view
geometry as its activation (so Concat / Slice / Dropout views need no backward kernel); all
parameter gradients live in one flat buffer parallel to `Engine.param_flat`, which is what the
solver kernel updates and what the RCCL all-reduce sums across ranks in one call.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import lib as L
from . import proto
from .engine import Blob, DevView, DeviceBuffer, Engine, Op, PinnedArray, _r4
from .netspec import DATA_TYPES, LOSS_TYPES, Layer, NetSpec, kernel_stride_pad

F32 = np.float32


class SolverParams:
    """The subset of Caffe's SolverParameter the reference's solver.prototxt files use."""

    def __init__(self, msg: Optional[proto.Msg] = None, **kw):
        g = (lambda k, d=None: msg.get(k, d)) if msg is not None else (lambda k, d=None: d)
        self.net = g("net", g("train_net"))
        self.base_lr = float(kw.get("base_lr", g("base_lr", 0.01)))
        self.momentum = float(kw.get("momentum", g("momentum", 0.0)))
        self.momentum2 = float(kw.get("momentum2", g("momentum2", 0.999)))
        self.delta = float(kw.get("delta", g("delta", 1e-8)))
        self.weight_decay = float(kw.get("weight_decay", g("weight_decay", 0.0)))
        self.lr_policy = str(kw.get("lr_policy", g("lr_policy", "fixed")))
        self.gamma = float(kw.get("gamma", g("gamma", 0.1)))
        self.stepsize = int(kw.get("stepsize", g("stepsize", 1)))
        self.max_iter = int(kw.get("max_iter", g("max_iter", 1)))
        self.iter_size = int(kw.get("iter_size", g("iter_size", 1)))
        self.display = int(kw.get("display", g("display", 0)))
        self.average_loss = int(kw.get("average_loss", g("average_loss", 1)))
        self.snapshot = int(kw.get("snapshot", g("snapshot", 0)))
        self.snapshot_prefix = str(kw.get("snapshot_prefix", g("snapshot_prefix", "snapshot")))
        kind = kw.get("solver_type", g("solver_type", g("type", "SGD")))
        self.kind = str(kind).upper()
        if self.kind not in ("SGD", "ADAM"):
            raise NotImplementedError("solver type %s (the reference uses SGD and ADAM)" % self.kind)
        if self.lr_policy not in ("fixed", "step"):
            raise NotImplementedError("lr_policy %s (the reference uses fixed and step)" % self.lr_policy)

    def rate(self, it: int) -> float:
        if self.lr_policy == "fixed":
            return self.base_lr
        return self.base_lr * self.gamma ** (it // self.stepsize)


class TrainEngine(Engine):
    """Engine for the TRAIN phase with backward pass and solver state."""

    def __init__(self, spec: NetSpec, data_shapes: Dict[str, Tuple[int, ...]], params=None, device: int = 0,
                 solver: Optional[SolverParams] = None, comm=None, autotune: bool = True):
        if spec.phase != "TRAIN":
            raise ValueError("TrainEngine needs a TRAIN-phase NetSpec")
        self.solver = solver or SolverParams()
        self.comm = comm                      # None or an object with all_reduce_sum(ptr, count, stream) and .world
        self.grad_blobs: Dict[str, Blob] = {}
        self.bwd_ops: List[Op] = []
        self.iter = 0
        super().__init__(spec, data_shapes, params, device, fuse=True, group_convs=True, autotune=autotune)
        self._alloc_solver_state()
        self._build_backward()

    # ------------------------------------------------------------------ gradient buffers
    def _learns(self, l: Layer) -> bool:
        """True if the solver will move any blob of this layer: Caffe's param_propagate_down (lr_mult != 0)."""
        if l.name not in self.spec.param_shapes:
            return False
        n = len(self.spec.param_shapes[l.name])
        return any((l.lr_mult[i] if i < len(l.lr_mult) else 1.0) != 0.0 for i in range(n))

    def _needs_grad(self) -> set:
        """Blobs downstream of a layer that learns (Caffe's propagate_down): only those carry gradients.  Frozen layers
        (lr_mult 0: conv1_1..conv3_3 of train/bounding_box, every bilinear deconvolution) neither get a weight gradient
        nor pull the backward pass below them."""
        need = set()
        for l in self.spec.layers:
            if l.type in DATA_TYPES:
                continue
            if self._learns(l) or any(b in need for b in l.bottoms):
                need.update(l.tops)
        return need

    def _plan_buffers(self) -> None:
        super()._plan_buffers()
        need = self._needs_grad()
        self.need_grad = need
        roots: Dict[int, DeviceBuffer] = {}      # activation buffer address -> gradient buffer
        for name, b in self.blobs.items():
            if name not in need or len(b.shape) != 4:
                continue
            gb = roots.get(b.buf.ptr)
            if gb is None:
                gb = DeviceBuffer(b.buf.nbytes, zero=True)
                roots[b.buf.ptr] = gb
            g = Blob(name, b.shape)
            g.buf, g.coffset, g.cstride = gb, b.coffset, b.cstride
            self.grad_blobs[name] = g

    def _loss_grad_ptr(self, blob: str) -> Optional[int]:
        g = self.grad_blobs.get(blob)
        if g is None:
            return None
        if g.coffset:
            raise NotImplementedError("loss gradient into a channel slice")
        return g.ptr

    # ------------------------------------------------------------------ solver state
    def _alloc_solver_state(self) -> None:
        n = max(self.param_count, 4)
        self.grad_flat = DeviceBuffer(n * 4, zero=True)
        self.hist = DeviceBuffer(n * 4, zero=True)
        self.hist2 = DeviceBuffer(n * 4, zero=True) if self.solver.kind == "ADAM" else None
        segs = (L.SolverSeg * len(self.param_layout))(*[
            L.SolverSeg(e["offset"], e["count"], e["lr_mult"], e["decay_mult"]) for e in self.param_layout])
        self._segs_host = segs
        self.segs_dev = DeviceBuffer(max(C.sizeof(segs), 16), zero=False)
        L.call("fcn_memcpy_h2d_async", self.segs_dev.ptr, C.addressof(segs), C.sizeof(segs), None)
        L.call("fcn_device_sync")
        # pinned: the loss read-back at the end of step_begin() must not hold the host thread until the step has run (a
        # pageable destination makes the "async" copy synchronous, and the next batch could not be prepared meanwhile)
        self._loss_pinned = {name: PinnedArray((1,)) for name in self.loss_blobs}
        self.loss_host = {name: p.array for name, p in self._loss_pinned.items()}

    def _grad_view(self, layer: str, index: int) -> DevView:
        for e in self.param_layout:
            if e["layer"] == layer and e["index"] == index:
                return DevView(self.grad_flat.ptr + 4 * e["offset"], 4 * e["count"])
        raise KeyError((layer, index))

    # ------------------------------------------------------------------ backward plan
    def _build_backward(self) -> None:
        spec, B, G, lib = self.spec, self.blobs, self.grad_blobs, L.load()
        written: Dict[int, List[Tuple[int, int]]] = {}      # gradient buffer -> channel ranges already holding a gradient

        def state(g: Blob) -> str:
            """'none' | 'full' for the channel range of view g (partial overlap is a planning error)."""
            lo, hi = g.coffset, g.coffset + g.channels
            cov = 0
            for a, b in written.get(g.buf.ptr, []):
                o = min(hi, b) - max(lo, a)
                if o > 0:
                    cov += o
            if cov == 0:
                return "none"
            if cov >= hi - lo:
                return "full"
            raise NotImplementedError("gradient of %s is partially written" % g.name)

        writers: Dict[str, List[object]] = {}      # gradient blob -> what wrote it, in order (a dgrad record or None)

        def mark(g: Blob, writer: object = None) -> None:
            written.setdefault(g.buf.ptr, []).append((g.coffset, g.coffset + g.channels))
            writers.setdefault(g.name, []).append(writer)

        ws_floats = 1
        ops: List[Op] = []
        # Concat outputs all of whose members are convolutions with a fused in-place ReLU: their ReLU backward is one launch
        concat_members: Dict[str, List[str]] = {}
        for child, (parent, _off) in self.alias.items():
            if any(q.type == "Concat" and parent in q.tops for q in self.producers.get(parent, [])):
                concat_members.setdefault(parent, []).append(child)
        concat_relu: Dict[str, str] = {}
        for parent, members in concat_members.items():
            prods = [[q for q in self.producers.get(m, []) if q.type == "Convolution"] for m in members]
            if all(len(pr) == 1 and self._conv_layer_meta.get(pr[0].name, {}).get("relu") for pr in prods) and \
                    sum(B[m].channels for m in members) == B[parent].channels and B[parent].coffset == 0:
                for m in members:
                    concat_relu[m] = parent
        relu_done: set = set()
        dgrad_done: set = set()
        wgrad_done: set = set()
        sibling_reduces: Dict[str, List[Layer]] = {}      # reduce layer -> the reduce layers of its module (ready together)
        # Flipped / transposed filter banks of the data-gradient passes: slices of ONE flat buffer that a single launch
        # refreshes from the current weights at the start of every backward pass (58 launches otherwise).
        flip_layout: Dict[str, int] = {}
        flip_segs: List[L.FlipSeg] = []
        flip_floats = 0
        for l in spec.layers:
            if l.type != "Convolution" or G.get(l.bottoms[0]) is None or G.get(l.tops[0]) is None:
                continue
            k, s_, _pad = kernel_stride_pad(l.sub("convolution_param"))
            if s_ != 1:
                continue
            cin, cout = B[l.bottoms[0]].shape[1], B[l.tops[0]].shape[1]
            flip_layout[l.name] = flip_floats
            flip_segs.append(L.FlipSeg((self.params_dev[l.name][0].ptr - self.param_flat.ptr) // 4, flip_floats, cout, k, k, cin,
                                       _r4(cin), _r4(cout)))
            flip_floats += _r4(cin * k * k * _r4(cout))
        self._flip_flat = DeviceBuffer(max(flip_floats, 4) * 4, zero=True)
        if flip_segs:
            seg_arr = (L.FlipSeg * len(flip_segs))(*flip_segs)
            self._flip_segs_dev = DeviceBuffer(C.sizeof(seg_arr), zero=False)
            L.call("fcn_memcpy_h2d_async", self._flip_segs_dev.ptr, C.addressof(seg_arr), C.sizeof(seg_arr), None)
            L.call("fcn_device_sync")
            ops.append(Op("flip", "%d filter banks" % len(flip_segs), lambda st, n=len(flip_segs): L.check(lib.fcn_conv_weights_flip_batch_f32(
                self.param_flat.ptr, self._flip_flat.ptr, self._flip_segs_dev.ptr, n, st))))
        skip_sigmoid_of = {m["sigmoid_top"]: name for name, m in self._conv_layer_meta.items() if m.get("sigmoid_top")}

        def dgrad_desc(l: Layer, gtop: Blob, gbot: Blob, accumulate: bool) -> Tuple[L.ConvDesc, float]:
            """Data gradient of convolution l = the forward kernel on dY with the flipped / transposed bank."""
            xb, yb = B[l.bottoms[0]], B[l.tops[0]]
            k, s, pad = kernel_stride_pad(l.sub("convolution_param"))
            n, cin, h, w = xb.shape
            _, cout, oh, ow = yb.shape
            if s != 1:
                raise NotImplementedError("data gradient of the strided convolution %s" % l.name)
            cin_dg = _r4(cout)      # the flipped bank reads Cout4 input channels: the gradient view must expose them contiguously
            if gtop.cstride - gtop.coffset < cin_dg:
                raise NotImplementedError("gradient view of %s too narrow for the data-gradient pass" % l.tops[0])
            wt = DevView(self._flip_flat.ptr + 4 * flip_layout[l.name], cin * k * k * cin_dg * 4)
            dd = L.ConvDesc()
            dd.x, dd.w, dd.bias, dd.y = gtop.ptr, wt.ptr, None, gbot.buf.ptr
            dd.N, dd.H, dd.W, dd.Cin, dd.x_cstride = n, oh, ow, cin_dg, gtop.cstride
            dd.Cout, dd.kh, dd.kw, dd.pad, dd.stride, dd.OH, dd.OW = cin, k, k, k - 1 - pad, 1, h, w
            dd.y_cstride, dd.y_coffset = gbot.cstride, gbot.coffset
            dd.flags = L.CONV_ACCUM if accumulate else 0
            self._keep.append(dd)
            return dd, 2.0 * n * cout * oh * ow * cin * k * k

        dgrad_records: List[dict] = []
        relu_ops: Dict[str, Op] = {}       # gradient blob whose ReLU backward is the op (candidates for the fused mask)

        def emit_dgrads(name: str, items: List[Tuple[L.ConvDesc, float]], targets: List[str]) -> dict:
            """One grouped launch for data-gradient passes that write different buffers.  The group is prepared (and
            autotuned) after the whole backward plan is known, because the LAST writer of a gradient may still get the ReLU
            mask of the layer below folded into its epilogue (finish_dgrads)."""
            rec = dict(name=name, descs=[it[0] for it in items], targets=list(targets), grp=L.ConvGroup())
            rec["op"] = Op("dgrad", name, lambda st, g=rec["grp"]: L.check(lib.fcn_conv2d_fwd_group_f32(C.byref(g), st)), sum(it[1] for it in items))
            ops.append(rec["op"])
            dgrad_records.append(rec)
            return rec

        def finish_dgrads() -> None:
            # fold "ReLU backward of blob X" into the last data-gradient pass that writes dX, when that is what wrote it last
            for x, rop in relu_ops.items():
                w = writers.get(x, [])
                rec = w[-1] if w else None
                if not isinstance(rec, dict) or rop not in ops:
                    continue
                if "pool" in rec:        # the last writer is a pooling backward of exactly this blob
                    if rec["pool"] == x:
                        rec["mask"] = (B[x].buf.ptr, B[x].cstride, B[x].coffset)
                        ops.remove(rop)
                    continue
                d = rec["descs"][rec["targets"].index(x)]
                act = B[x]
                d.y2, d.y2_cstride, d.y2_coffset = act.buf.ptr, act.cstride, act.coffset
                d.flags |= L.CONV_MASK
                ops.remove(rop)
            for rec in dgrad_records:
                n_ = len(rec["descs"])
                arr = (L.ConvDesc * n_)(*rec["descs"])
                gws = DeviceBuffer(int(lib.fcn_conv2d_group_workspace_bytes(n_)), zero=False)
                cfg = self._tuned_cfg("dgrad:" + rec["name"], arr, n_, gws) if self.autotune else -1
                L.call("fcn_conv2d_group_prepare", arr, n_, gws.ptr, cfg, C.byref(rec["grp"]))
                self._keep.extend([arr, gws, rec["grp"]])
                self._group_workspaces.append(gws)
                rec["op"].name = "%s [cfg%d %dwg]" % (rec["name"], rec["grp"].cfg, rec["grp"].total_tiles)

        def wgrad_item(l: Layer, gtop: Blob):
            """(descriptor with y = dY of the layer, dW view, db view or None, flops) of a layer that learns."""
            xb, yb = B[l.bottoms[0]], B[l.tops[0]]
            k, s, pad = kernel_stride_pad(l.sub("convolution_param"))
            n, cin, h, w = xb.shape
            _, cout, oh, ow = yb.shape
            if gtop.coffset % 4 or gtop.cstride % 4:
                raise NotImplementedError("gradient view of %s is not 16-byte aligned" % l.tops[0])
            d = L.ConvDesc()
            d.x, d.y = xb.ptr, gtop.buf.ptr
            d.N, d.H, d.W, d.Cin, d.x_cstride = n, h, w, _r4(cin), xb.cstride
            d.Cout, d.kh, d.kw, d.pad, d.stride, d.OH, d.OW = cout, k, k, pad, s, oh, ow
            d.y_cstride, d.y_coffset = gtop.cstride, gtop.coffset
            self._keep.append(d)
            dw = self._grad_view(l.name, 0)
            db = self._grad_view(l.name, 1) if len(self.params_dev[l.name]) > 1 else None
            return d, dw, db, 2.0 * n * cout * oh * ow * cin * k * k

        def emit_wgrads(layers_: List[Layer], gtops: List[Blob]) -> None:
            """Weight (and bias) gradients of layers that are ready together: one launch + one reduction for up to four."""
            nonlocal ws_floats
            todo = [(l_, g_) for l_, g_ in zip(layers_, gtops) if self._learns(l_) and l_.name not in wgrad_done]
            for base in range(0, len(todo), 4):
                chunk = todo[base:base + 4]
                its = [wgrad_item(l_, g_) for l_, g_ in chunk]
                names = [l_.name for l_, _ in chunk]
                sel = {"cfg": -1}      # -1: the library's heuristic; _tune_wgrads() replaces it once the workspace exists
                cfgs = [-1] + (list(range(int(lib.fcn_conv2d_wgrad_num_configs()))) if self.autotune else [])
                if len(its) == 1:
                    d, dw, db, fl = its[0]
                    ws_floats = max([ws_floats] + [int(lib.fcn_conv2d_wgrad_workspace_floats_cfg(C.byref(d), c, None)) for c in cfgs])
                    op = Op("wgrad", names[0], lambda st, d=d, dw=dw, db=db, sel=sel: L.check(lib.fcn_conv2d_wgrad_cfg_f32(
                        C.byref(d), dw.ptr, db.ptr if db else None, self._ws.ptr, sel["cfg"], st)), fl)
                else:
                    arr = (L.ConvDesc * len(its))(*[it[0] for it in its])
                    pdw = (C.c_void_p * len(its))(*[it[1].ptr for it in its])
                    pdb = (C.c_void_p * len(its))(*[(it[2].ptr if it[2] is not None else None) for it in its])
                    ws_floats = max([ws_floats] + [int(lib.fcn_conv2d_wgrad_group_workspace_floats_cfg(arr, len(its), c)) for c in cfgs])
                    self._keep.extend([arr, pdw, pdb])
                    op = Op("wgrad", "+".join(names), lambda st, arr=arr, pdw=pdw, pdb=pdb, m=len(its), sel=sel: L.check(
                        lib.fcn_conv2d_wgrad_group_cfg_f32(arr, pdw, pdb, m, self._ws.ptr, sel["cfg"], st)), sum(it[3] for it in its))
                op.sel = sel
                op.layers = names
                ops.append(op)
                wgrad_done.update(names)

        for l in reversed(spec.layers):
            t = l.type
            if t in DATA_TYPES or t in ("Concat", "Slice") or (t == "ReLU" and l.name in self._fused_relu_layers()):
                if t == "Slice" and l.name in self.copy_slices and any(tp in G for tp in l.tops):
                    raise NotImplementedError("backward through the copied Slice %s" % l.name)
                continue
            if t in ("L1Loss", "EuclideanLoss", "SoftmaxWithLoss"):
                g = G.get(l.bottoms[0])
                if g is None:
                    continue
                if l.bottoms[1] in self.need_grad:
                    raise NotImplementedError("loss layer %s: gradient w.r.t. the second bottom" % l.name)
                if state(g) != "none":
                    raise NotImplementedError("loss gradient would have to accumulate into %s" % l.bottoms[0])
                mark(g)                      # written by the forward loss kernel (da)
                continue
            if t == "Sigmoid" and l.tops[0] in skip_sigmoid_of:
                # fused into the conv epilogue in forward; backward is its own small kernel
                yb, gtop, gbot = B[l.tops[0]], G.get(l.tops[0]), G.get(l.bottoms[0])
                if gtop is None or gbot is None or state(gtop) == "none":
                    continue
                acc = 1 if state(gbot) == "full" else 0
                count = yb.pixels * yb.cstride
                if yb.coffset or gtop.coffset or gbot.coffset or yb.cstride != gbot.cstride:
                    raise NotImplementedError("sigmoid backward on channel slices")
                ops.append(Op("sigmoid_bwd", l.name, lambda st, y=yb, a=gtop, b=gbot, acc=acc, n=count: L.check(
                    lib.fcn_sigmoid_bwd_f32(y.ptr, a.ptr, b.ptr, n, acc, st))))
                mark(gbot)
                continue
            gtop = G.get(l.tops[0]) if l.tops else None
            if gtop is None or state(gtop) == "none":
                continue                     # no gradient reaches this layer
            if t == "Convolution":
                meta = self._conv_layer_meta[l.name]
                xb, yb = B[l.bottoms[0]], B[l.tops[0]]
                p = l.sub("convolution_param")
                k, s, pad = kernel_stride_pad(p)
                n, cin, h, w = xb.shape
                _, cout, oh, ow = yb.shape
                if meta.get("relu") and l.tops[0] not in relu_done:
                    whole = concat_relu.get(l.tops[0])
                    if whole is not None and whole in G and state(G[whole]) == "full":
                        # every member of this Concat is a convolution with an in-place ReLU and the gradient of the whole
                        # concatenation is final: ONE contiguous launch masks all members (an inception module: 4 -> 1)
                        gw, yw = G[whole], B[whole]
                        relu_ops[whole] = Op("relu_bwd", whole, lambda st, g=gw, y=yw: L.check(lib.fcn_relu_bwd_f32(
                            g.ptr, y.ptr, g.ptr, y.pixels, y.channels, y.cstride, st)), 0.0, 12.0 * yw.pixels * yw.channels)
                        ops.append(relu_ops[whole])
                        relu_done.update(concat_members[whole])
                        # the members' data gradients only need this masked gradient and write four different buffers (the
                        # module input and the outputs of the reduce / pool layers): one grouped launch at the top of the
                        # module's backward instead of four scattered ones
                        items, names, targets, tnames = [], [], [], []
                        for m in concat_members[whole]:
                            lm = [q for q in self.producers.get(m, []) if q.type == "Convolution"][0]
                            gb = G.get(lm.bottoms[0])
                            if gb is None or lm.name not in flip_layout or state(gb) != "none" or any(gb.buf.ptr == tb for tb in targets):
                                continue
                            items.append(dgrad_desc(lm, G[m], gb, False))
                            names.append(lm.name)
                            targets.append(gb.buf.ptr)
                            tnames.append(lm.bottoms[0])
                        if len(items) > 1:
                            rec = emit_dgrads("+".join(names), items, tnames)
                            for nm, tn in zip(names, tnames):
                                dgrad_done.add(nm)
                                mark(G[tn], rec)
                        # ... and their weight gradients need nothing else either: one grouped launch
                        mem_layers = [[q for q in self.producers.get(m, []) if q.type == "Convolution"][0] for m in concat_members[whole]]
                        emit_wgrads(mem_layers, [G[m] for m in concat_members[whole]])
                        # the layers feeding the members (3x3_reduce, 5x5_reduce) get their whole gradient from that dgrad
                        # launch: they become ready together too
                        sibs = []
                        for lm in mem_layers:
                            if lm.name not in dgrad_done:
                                continue
                            prods = [q for q in self.producers.get(lm.bottoms[0], []) if q.type == "Convolution"]
                            cons = [q for q in self.consumers.get(lm.bottoms[0], []) if not (q.type in ("ReLU", "Dropout") and q.bottoms == q.tops)]
                            if len(prods) == 1 and len(cons) == 1 and lm.bottoms[0] not in self.alias and lm.bottoms[0] in G:
                                sibs.append(prods[0])
                        if len(sibs) > 1:
                            for q in sibs:
                                sibling_reduces[q.name] = sibs
                    else:
                        rop = Op("relu_bwd", l.name, lambda st, g=gtop, y=yb: L.check(lib.fcn_relu_bwd_f32(
                            g.ptr, y.ptr, g.ptr, y.pixels, y.channels, y.cstride, st)), 0.0, 12.0 * yb.pixels * cout)
                        ops.append(rop)
                        if l.tops[0] not in self.alias:
                            relu_ops[l.tops[0]] = rop
                        relu_done.add(l.tops[0])
                sibs = sibling_reduces.get(l.name)
                if sibs and l.name not in wgrad_done and all(state(G[q.tops[0]]) == "full" for q in sibs):
                    # first of the module's reduce layers to be visited: mask and take the weight gradients of all of them now
                    for q in sibs:
                        if q.tops[0] not in relu_done and self._conv_layer_meta[q.name].get("relu"):
                            gq, yq = G[q.tops[0]], B[q.tops[0]]
                            relu_ops[q.tops[0]] = Op("relu_bwd", q.name, lambda st, g=gq, y=yq: L.check(lib.fcn_relu_bwd_f32(
                                g.ptr, y.ptr, g.ptr, y.pixels, y.channels, y.cstride, st)), 0.0, 12.0 * yq.pixels * yq.channels)
                            ops.append(relu_ops[q.tops[0]])
                            relu_done.add(q.tops[0])
                    emit_wgrads(sibs, [G[q.tops[0]] for q in sibs])
                emit_wgrads([l], [gtop])
                gbot = G.get(l.bottoms[0])
                if gbot is not None and l.name not in dgrad_done:
                    rec = emit_dgrads(l.name, [dgrad_desc(l, gtop, gbot, state(gbot) == "full")], [l.bottoms[0]])
                    mark(gbot, rec)
                continue
            if t == "Eltwise" and str(l.sub("eltwise_param").get("operation", "SUM")) == "SUM":
                p = l.sub("eltwise_param")
                # d(bottom_i) = dY for every bottom (train/fcn_bbox fuse_pool4 / fuse_pool3: skip connections)
                if any(float(c) != 1.0 for c in p.getall("coeff")):
                    raise NotImplementedError("Eltwise SUM backward with coefficients (%s)" % l.name)
                for bn in l.bottoms:
                    gb = G.get(bn)
                    if gb is None:
                        continue
                    if state(gb) == "full":
                        if gtop.coffset or gb.coffset or gb.cstride != gtop.cstride:
                            raise NotImplementedError("Eltwise SUM backward accumulating into a channel slice")
                        ops.append(Op("eltwise_bwd", l.name + ":" + bn, lambda st, a=gtop, b=gb: L.check(lib.fcn_eltwise_fwd_f32(
                            a.ptr, b.ptr, b.ptr, a.pixels * a.cstride, L.ELT_SUM, 1.0, 1.0, st))))
                    else:
                        ops.append(Op("eltwise_bwd", l.name + ":" + bn, lambda st, a=gtop, b=gb: L.check(lib.fcn_copy_channels_f32(
                            a.buf.ptr, b.buf.ptr, a.pixels, a.channels, a.cstride, a.coffset, b.cstride, b.coffset, st))))
                    mark(gb)
                continue
            gbot = G.get(l.bottoms[0]) if l.bottoms else None
            if gbot is None:
                continue
            acc = 1 if state(gbot) == "full" else 0
            pool_writer = None
            if t == "Pooling":
                pp = l.sub("pooling_param")
                if str(pp.get("pool", "MAX")) != "MAX":
                    raise NotImplementedError("backward of AVE pooling (%s)" % l.name)
                xb, yb = B[l.bottoms[0]], B[l.tops[0]]
                n, c, h, w = xb.shape
                _, _, oh, ow = yb.shape
                k, s, pad = kernel_stride_pad(pp)
                idx = self.aux_dev[l.name]
                prec = dict(pool=l.bottoms[0], mask=(None, 0, 0))      # finish_dgrads may fold a ReLU backward into this pass
                ops.append(Op("maxpool_bwd", l.name, lambda st, a=gtop, b=gbot, idx=idx, g=(n, h, w, c), kk=(k, s, pad, oh, ow), acc=acc, r=prec:
                              L.check(lib.fcn_maxpool_bwd_mask_f32(a.buf.ptr, idx.ptr, b.buf.ptr, g[0], g[1], g[2], g[3], b.cstride, b.coffset,
                                                                   kk[0], kk[1], kk[2], kk[3], kk[4], a.cstride, a.coffset, acc, r["mask"][0],
                                                                   r["mask"][1], r["mask"][2], st))))
                pool_writer = prec
            elif t == "LRN":
                xb, yb = B[l.bottoms[0]], B[l.tops[0]]
                p = l.sub("lrn_param")
                ls, al, be = int(p.get("local_size", 5)), float(p.get("alpha", 1.0)), float(p.get("beta", 0.75))
                sc = self.aux_dev[l.name]
                if gtop.coffset or gbot.coffset or xb.coffset or yb.coffset:
                    raise NotImplementedError("LRN backward on channel slices")
                ops.append(Op("lrn_bwd", l.name, lambda st, x=xb, y=yb, sc=sc, a=gtop, b=gbot, q=(ls, al, be), acc=acc: L.check(
                    lib.fcn_lrn_bwd_f32(x.ptr, y.ptr, sc.ptr, a.ptr, b.ptr, x.pixels, x.channels, x.cstride, y.cstride, q[0], q[1], q[2], acc, st))))
            elif t == "Dropout":
                if acc:
                    raise NotImplementedError("dropout backward into an already written gradient")
                ratio = float(l.sub("dropout_param").get("dropout_ratio", 0.5))
                n, c, h, w = B[l.bottoms[0]].shape
                ops.append(Op("dropout_bwd", l.name, lambda st, a=gtop, b=gbot, g=(n, c, h, w), r=ratio: L.check(lib.fcn_dropout_f32(
                    a.buf.ptr, b.buf.ptr, g[0], g[1], g[2], g[3], a.cstride, a.coffset, b.cstride, b.coffset, r, self.dropout_seed,
                    self.dropout_index_offset, st))))
            elif t == "Eltwise":
                p = l.sub("eltwise_param")
                opname = str(p.get("operation", "SUM"))
                if opname != "PROD" or len(l.bottoms) != 2:
                    raise NotImplementedError("backward of Eltwise %s" % l.name)
                if l.bottoms[1] in self.need_grad:
                    raise NotImplementedError("Eltwise PROD backward w.r.t. both bottoms (%s)" % l.name)
                if acc:
                    raise NotImplementedError("Eltwise backward into an already written gradient")
                other = B[l.bottoms[1]]
                count = gtop.pixels * gtop.cstride
                if gtop.coffset or gbot.coffset or other.coffset or other.cstride != gtop.cstride:
                    raise NotImplementedError("Eltwise backward on channel slices")
                ops.append(Op("eltwise_bwd", l.name, lambda st, a=gtop, o=other, b=gbot, n=count: L.check(lib.fcn_eltwise_fwd_f32(
                    a.ptr, o.ptr, b.ptr, n, L.ELT_PROD, 1.0, 1.0, st))))
            elif t == "Deconvolution":
                if any(m != 0.0 for m in l.lr_mult) or not l.lr_mult:
                    raise NotImplementedError("learnable Deconvolution %s (the reference freezes its bilinear upsampling, lr_mult 0)" % l.name)
                p = l.sub("convolution_param")
                k, s, pad = kernel_stride_pad(p)
                xb, yb = B[l.bottoms[0]], B[l.tops[0]]
                n, c, h, w = xb.shape
                _, _, oh, ow = yb.shape
                wdev = self.params_dev[l.name][0].ptr
                ops.append(Op("deconv_bwd", l.name, lambda st, a=gtop, b=gbot, g=(n, h, w, c, k, s, pad, oh, ow), acc=acc, wdev=wdev: L.check(
                    lib.fcn_deconv_depthwise_bwd_f32(a.buf.ptr, wdev, b.ptr, g[0], g[1], g[2], g[3], b.cstride, g[4], g[5], g[6], g[7], g[8],
                                                     a.cstride, a.coffset, acc, st))))
            elif t == "Sigmoid":
                yb = B[l.tops[0]]
                ops.append(Op("sigmoid_bwd", l.name, lambda st, y=yb, a=gtop, b=gbot, acc=acc: L.check(
                    lib.fcn_sigmoid_bwd_f32(y.ptr, a.ptr, b.ptr, y.pixels * y.cstride, acc, st))))
            elif t == "ReLU":
                yb = B[l.tops[0]]
                ops.append(Op("relu_bwd", l.name, lambda st, y=yb, a=gtop, b=gbot: L.check(lib.fcn_relu_bwd_f32(
                    a.ptr, y.ptr, b.ptr, y.pixels, y.channels, y.cstride, st))))
            elif t == "Power":
                continue        # input transform: nothing upstream learns
            else:
                raise NotImplementedError("backward of layer type %s (%s)" % (t, l.name))
            mark(gbot, pool_writer)
        finish_dgrads()
        self._ws = DeviceBuffer(ws_floats * 4, zero=False)
        self.bwd_ops = ops
        if self.autotune:
            self._tune_wgrads()
        self._plan_buckets()

    def _tune_wgrads(self) -> None:
        """Plan-time choice of every weight-gradient launch's configuration (the 64-wide tile shapes and the role-split kernel of
        csrc/train.hip): each is timed on the buffers the step will use, the fastest is kept - and remembered in $FCN_TUNE_CACHE
        beside the forward plan.  Gradient buffers hold garbage until the first real backward pass, which overwrites them."""
        lib = L.load()
        ncfg = int(lib.fcn_conv2d_wgrad_num_configs())
        cache = self._load_tune_cache()
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.call("fcn_event_create", C.byref(e0))
        L.call("fcn_event_create", C.byref(e1))
        dirty = False
        for op in self.bwd_ops:
            if op.kind != "wgrad":
                continue
            key = self._tune_key("wgrad:" + "+".join(op.layers))
            if self._tune_from is not None and key in self._tune_from._chosen_cfgs:
                op.sel["cfg"] = self._chosen_cfgs[key] = self._tune_from._chosen_cfgs[key]
                continue
            if cache is not None and key in cache and 0 <= int(cache[key]) < ncfg:
                op.sel["cfg"] = self._chosen_cfgs[key] = int(cache[key])
                continue
            best, best_ms = -1, 1e30
            timed = []
            for cfg in range(ncfg):
                op.sel["cfg"] = cfg
                for _ in range(2):
                    op.run(self.stream)
                L.call("fcn_event_record", e0, self.stream)
                for _ in range(5):
                    op.run(self.stream)
                L.call("fcn_event_record", e1, self.stream)
                L.call("fcn_event_sync", e1)
                ms = C.c_float()
                L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
                if ms.value < best_ms:
                    best, best_ms = cfg, ms.value
                timed.append((ms.value, cfg))
            # a second look at the contenders within 4 % (as Engine._time_conv_cfgs): four more rounds of five launches each, the minimum counts
            finals = []
            for t1, cfg in sorted(timed)[:3]:
                if t1 > 1.04 * best_ms or len(timed) < 2:
                    break
                op.sel["cfg"] = cfg
                rounds = [t1]
                for _ in range(4):
                    L.call("fcn_event_record", e0, self.stream)
                    for _ in range(5):
                        op.run(self.stream)
                    L.call("fcn_event_record", e1, self.stream)
                    L.call("fcn_event_sync", e1)
                    ms = C.c_float()
                    L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
                    rounds.append(ms.value)
                finals.append((min(rounds), cfg))
            if finals:
                best = min(finals)[1]
            op.sel["cfg"] = self._chosen_cfgs[key] = best
            if cache is not None:
                cache[key] = best
                dirty = True
        if dirty:
            self._save_tune_cache()
        for op in self.bwd_ops:
            if op.kind == "wgrad":
                op.name += " [cfg%d]" % op.sel["cfg"]
        L.call("fcn_event_destroy", e0)
        L.call("fcn_event_destroy", e1)

    def _plan_buckets(self, bucket_floats: int = 1536 * 1024) -> None:
        """Gradient buckets for the overlapped all-reduce: contiguous ranges of the flat gradient buffer (forward layer
        order).  Backward fills the buffer roughly from its end; a bucket is complete once the LAST of its layers' weight
        gradient launches (grouped launches reorder them within a module) has been enqueued; its all-reduce then runs on a
        side stream while the main stream continues with earlier layers."""
        self.buckets: List[dict] = []
        if self.comm is None:
            return
        cur = None
        for e in self.param_layout:
            if cur is None or (e["index"] == 0 and cur["count"] >= bucket_floats):
                cur = dict(offset=e["offset"], count=0, layers=[])
                self.buckets.append(cur)
            cur["count"] = e["offset"] + _r4(e["count"]) - cur["offset"]
            if e["layer"] not in cur["layers"]:
                cur["layers"].append(e["layer"])
        wg_index = {}
        for i, op in enumerate(self.bwd_ops):
            if op.kind == "wgrad":
                for nm in getattr(op, "layers", [op.name]):
                    wg_index[nm] = i
        lib = L.load()
        sp = C.c_void_p()
        L.call("fcn_stream_create", C.byref(sp))
        self.comm_stream = int(sp.value)
        for b in self.buckets:
            done = [wg_index[nm] for nm in b["layers"] if nm in wg_index]
            b["after_op"] = max(done) if done else len(self.bwd_ops) - 1
            for key in ("ready", "done"):
                ev = C.c_void_p()
                L.call("fcn_event_create", C.byref(ev))
                b[key] = ev

    def _fused_relu_layers(self) -> set:
        if not hasattr(self, "_fused_relu_cache"):
            out = set()
            layers = self.spec.layers
            for li, l in enumerate(layers):
                if l.type == "Convolution" and self._conv_layer_meta.get(l.name, {}).get("relu"):
                    top = l.tops[0]
                    for nxt in layers[li + 1:]:
                        if top in nxt.bottoms or top in nxt.tops:
                            if nxt.type == "ReLU":
                                out.add(nxt.name)
                            break
            self._fused_relu_cache = out
        return self._fused_relu_cache

    # ------------------------------------------------------------------ one solver iteration
    # blob names of the DetectNet label tops, in the order DataArgumentationLayer emits them (data_argumentation_layer.py:67-72)
    LABEL_TOPS = ("coverage-label", "bbox-label", "size-block", "obj-block", "coverage-block")

    def set_targets(self, rects: Sequence[Sequence[Sequence[int]]], labels: Sequence[Sequence[int]], stride: int,
                    iou_thresh: float = 0.1, tops: Sequence[str] = LABEL_TOPS) -> None:
        """Stage the ground-truth boxes of the next step; the label blobs are then generated ON THE DEVICE inside step()
        (fcn_gen_targets_nhwc: bounding_box_parameterized_labels of the reference) instead of being uploaded."""
        fg = self.blobs[tops[0]]
        n, c, gy, gx = fg.shape
        if len(rects) != n or len(labels) != n:
            raise ValueError("need boxes for %d images" % n)
        offs = np.zeros(n + 1, np.int32)
        flat_r, flat_l = [], []
        for i, (rs, ls) in enumerate(zip(rects, labels)):
            for r, lab in zip(rs, ls):
                if not 0 <= int(lab) < c:
                    raise IndexError("label %d outside [0, %d)" % (lab, c))
                flat_r.append([int(v) for v in r])
                flat_l.append(int(lab))
            offs[i + 1] = len(flat_r)
        if not hasattr(self, "_tgt"):
            cap = max(64 * n, 256)
            self._tgt = dict(cap=cap, rects=DeviceBuffer(cap * 16, zero=True), labels=DeviceBuffer(cap * 4, zero=True),
                             offs=DeviceBuffer((n + 1) * 4, zero=True))
        if len(flat_r) > self._tgt["cap"]:
            raise ValueError("too many boxes in one batch (%d > %d)" % (len(flat_r), self._tgt["cap"]))
        self._tgt.update(h_rects=np.asarray(flat_r, np.int32).reshape(-1, 4), h_labels=np.asarray(flat_l, np.int32), h_offs=offs,
                         stride=int(stride), thresh=float(iou_thresh), tops=tuple(tops), pending=True)

    def _enqueue_targets(self) -> None:
        t, lib = self._tgt, L.load()
        if t["h_rects"].size:
            L.check(lib.fcn_memcpy_h2d_async(t["rects"].ptr, t["h_rects"].ctypes.data, t["h_rects"].nbytes, self.stream))
            L.check(lib.fcn_memcpy_h2d_async(t["labels"].ptr, t["h_labels"].ctypes.data, t["h_labels"].nbytes, self.stream))
        L.check(lib.fcn_memcpy_h2d_async(t["offs"].ptr, t["h_offs"].ctypes.data, t["h_offs"].nbytes, self.stream))
        fg, bb, sz, ob, cv = (self.blobs[nm] for nm in t["tops"])
        n, c, gy, gx = fg.shape
        for b in (bb, sz, ob, cv):
            if b.coffset or b.cstride != bb.cstride:
                raise NotImplementedError("label blobs must be plain buffers of one geometry")
        L.check(lib.fcn_gen_targets_nhwc(t["rects"].ptr, t["labels"].ptr, t["offs"].ptr, n, c, gy, gx, t["stride"], t["thresh"],
                                         fg.ptr, fg.cstride, bb.ptr, sz.ptr, ob.ptr, cv.ptr, bb.cstride, self.stream))

    def step(self, seed: Optional[int] = None, upload: bool = True) -> Dict[str, float]:
        """Solver::Step for one iteration.  Inputs come from the input blobs' host arrays (upload=True), except label
        blobs staged with set_targets(), which are generated on the device; upload=False reuses what is already in HBM.
        Returns {loss blob: value} plus 'total_loss' = sum of loss_weight * value (what `caffe train` prints; also under
        'loss' when no blob has that name)."""
        self.step_begin(seed, upload)
        return self.step_end()

    def step_begin(self, seed: Optional[int] = None, upload: bool = True) -> None:
        """Enqueue a whole iteration (inputs, targets, forward, backward, all-reduce, update, loss read-back) and return
        without waiting: the caller may prepare the next batch while the device works (step_end() collects the losses)."""
        lib = L.load()
        with self.lock:
            L.call("fcn_init", self.device)
            self.dropout_seed = int(seed if seed is not None else self.iter) & 0xFFFFFFFF
            dev_targets = getattr(self, "_tgt", None) is not None and self._tgt.get("pending")
            fed = set(self.device_fed)      # inputs some producer already wrote in HBM (device scene renderer)
            if upload:
                skip = (set(self._tgt["tops"]) if dev_targets else set()) | fed
                for nm in self.inputs:
                    if nm not in skip:
                        self._enqueue_upload(nm, self.stream)
            if dev_targets:
                self._enqueue_targets()
            world = self.comm.world if self.comm is not None else 1
            side = self._wgrad_stream()
            side_used = False
            for kind, item in self._step_plan():
                if kind == "graph":
                    L.check(lib.fcn_graph_launch(item, self.stream))
                elif kind == "op":
                    item.run(self.stream)
                elif kind == "fork":
                    op, ev0, ev1 = item
                    L.check(lib.fcn_event_record(ev0, self.stream))
                    L.check(lib.fcn_stream_wait_event(side, ev0))
                    op.run(side)
                    L.check(lib.fcn_event_record(ev1, side))
                elif kind == "join":
                    L.check(lib.fcn_stream_wait_event(self.stream, item))
                elif kind == "side":
                    # a weight gradient: nothing later in this step reads it except the update, and its inputs (dY, X) are
                    # final here -> it runs on the second stream beside the data-gradient chain
                    op, ev = item
                    L.check(lib.fcn_event_record(ev, self.stream))
                    L.check(lib.fcn_stream_wait_event(side, ev))
                    op.run(side)
                    side_used = True
                else:
                    for b in item:
                        # this bucket's gradients are final: sum them across ranks on the side stream
                        if side_used:       # final = everything queued so far on BOTH streams
                            if "ready_main" not in b:
                                ev = C.c_void_p()
                                L.call("fcn_event_create", C.byref(ev))
                                b["ready_main"] = ev
                            L.check(lib.fcn_event_record(b["ready_main"], self.stream))
                            L.check(lib.fcn_stream_wait_event(side, b["ready_main"]))
                        L.check(lib.fcn_event_record(b["ready"], side if side_used else self.stream))
                        L.check(lib.fcn_stream_wait_event(self.comm_stream, b["ready"]))
                        if not getattr(self, "comm_dry", False):      # (benchmarks: the same step without the collective)
                            if getattr(self, "_replicas_diverged", False):
                                raise RuntimeError("TrainEngine: a comm_dry step applied un-reduced gradients; the replicas no longer hold the "
                                                   "same weights and this engine must not take real data-parallel steps (bench.py closes it)")
                            self.comm.all_reduce_sum(self.grad_flat.ptr + 4 * b["offset"], b["count"], self.comm_stream)
                        elif world > 1:
                            self._replicas_diverged = True
                        L.check(lib.fcn_event_record(b["done"], self.comm_stream))
            for b in self.buckets:
                L.check(lib.fcn_stream_wait_event(self.stream, b["done"]))
            if side_used:
                L.check(lib.fcn_event_record(self._side_done, side))
                L.check(lib.fcn_stream_wait_event(self.stream, self._side_done))
            self.apply_update(1.0 / (world * self.solver.iter_size))
            for name, arr in self.loss_host.items():
                L.check(lib.fcn_memcpy_d2h_async(arr.ctypes.data, self.blobs[name].buf.ptr, 4, self.stream))
            if getattr(self, "_step_done", None) is None:
                ev = C.c_void_p()
                L.call("fcn_event_create", C.byref(ev))
                self._step_done = ev
            L.check(lib.fcn_event_record(self._step_done, self.stream))
            self._in_flight = (list(self._tgt["tops"]) if dev_targets else []) + list(fed)

    def time_allreduce(self, reps: int = 10) -> Dict[str, float]:
        """The step's gradient all-reduce alone: every bucket back to back on the communication stream, nothing else on
        the GPU (every rank must call this together).  Returns microseconds per full-gradient all-reduce, the bytes summed
        and the bus bandwidth 2 (G-1)/G * bytes / time of the usual collective accounting."""
        if self.comm is None or not self.buckets:
            return {"allreduce_us": 0.0, "bytes": 0, "bus_GBps": 0.0, "buckets": 0}
        with self.lock:
            L.call("fcn_device_sync")
            e0, e1 = C.c_void_p(), C.c_void_p()
            L.call("fcn_event_create", C.byref(e0))
            L.call("fcn_event_create", C.byref(e1))
            scratch = DeviceBuffer(self.grad_flat.nbytes)      # summing zeros: the gradients themselves stay untouched
            for r in range(reps + 2):
                if r == 2:
                    L.call("fcn_event_record", e0, self.comm_stream)
                for b in self.buckets:
                    self.comm.all_reduce_sum(scratch.ptr + 4 * b["offset"], b["count"], self.comm_stream)
            L.call("fcn_event_record", e1, self.comm_stream)
            L.call("fcn_event_sync", e1)
            ms = C.c_float()
            L.call("fcn_event_elapsed_ms", e0, e1, C.byref(ms))
            L.call("fcn_event_destroy", e0)
            L.call("fcn_event_destroy", e1)
            scratch.free()
        nbytes = 4 * sum(b["count"] for b in self.buckets)
        us = ms.value * 1e3 / reps
        g = self.comm.world
        return {"allreduce_us": us, "bytes": nbytes, "buckets": len(self.buckets),
                "bus_GBps": (2.0 * (g - 1) / g * nbytes / (us * 1e-6) / 1e9) if us > 0 else 0.0}

    def step_end(self) -> Dict[str, float]:
        with self.lock:
            L.call("fcn_event_sync", self._step_done)
            for b in self.blobs.values():
                b.host_valid = b.is_input
            for nm in self._in_flight:
                self.blobs[nm].host_valid = False          # generated in HBM, never on the host
            out = {k: float(v[0]) for k, v in self.loss_host.items()}
            out["total_loss"] = float(sum(self.loss_blobs[k] * out[k] for k in self.loss_blobs))
            out.setdefault("loss", out["total_loss"])      # shorthand, unless a blob is itself called "loss" (train/fcn_bbox)
            self.iter += 1
            return out

    # kinds whose launch arguments change from step to step (the dropout seed): they stay ordinary launches
    DYNAMIC_KINDS = ("dropout", "dropout_bwd")

    def _wgrad_stream(self) -> Optional[int]:
        """Second stream for the weight-gradient launches (FCN_WGRAD_STREAM=0 keeps everything on one stream)."""
        import os
        if os.environ.get("FCN_WGRAD_STREAM", "1") == "0":
            return None
        if getattr(self, "_side_stream", None) is None:
            sp, ev = C.c_void_p(), C.c_void_p()
            L.call("fcn_stream_create", C.byref(sp))
            L.call("fcn_event_create", C.byref(ev))
            self._side_stream, self._side_done = int(sp.value), ev
        return self._side_stream

    def _step_plan(self) -> List[Tuple[str, object]]:
        """Forward + backward of one step as [("graph", hipGraphExec) | ("op", Op) | ("reduce", [buckets])].
        Maximal runs of launches with step-invariant arguments are captured once into hipGraphs (a step is ~240
        launches, most of them a few microseconds long: inside a graph the gap between two of them is about half of
        what a stream launch costs); dropout and the points where a gradient bucket goes to RCCL stay outside."""
        if getattr(self, "_plan", None) is not None:
            return self._plan
        import os
        # the first step runs as ordinary launches (code objects load lazily on a kernel's first launch, which must not
        # happen inside a stream capture); graphs are captured from the second step on
        first = not getattr(self, "_warm", False)
        self._warm = True
        use_graph = os.environ.get("FCN_TRAIN_GRAPH", "1") != "0" and os.environ.get("FCN_NO_GRAPH", "0") in ("", "0") and not first
        triggers: Dict[int, List[dict]] = {}
        for b in self.buckets:
            triggers.setdefault(b["after_op"], []).append(b)
        seq: List[Tuple[str, object]] = [("op", op) for op in self.ops]
        for i, op in enumerate(self.bwd_ops):
            seq.append(("op", op))
            if i in triggers:
                seq.append(("reduce", triggers[i]))
        plan: List[Tuple[str, object]] = []
        run: List[Op] = []

        def flush() -> None:
            if not run:
                return
            if use_graph and len(run) > 1:
                L.call("fcn_graph_begin", self.stream)
                try:
                    for op in run:
                        op.run(self.stream)
                finally:
                    g = C.c_void_p()
                    L.call("fcn_graph_end", self.stream, C.byref(g))
                self._step_graphs.append(int(g.value))
                plan.append(("graph", int(g.value)))
            else:
                plan.extend(("op", op) for op in run)
            run.clear()

        self._step_graphs: List[int] = getattr(self, "_step_graphs", [])
        side = self._wgrad_stream() is not None

        def new_event():
            ev = C.c_void_p()
            L.call("fcn_event_create", C.byref(ev))
            self._keep.append(ev)
            return ev
        if side and self.bwd_ops and self.bwd_ops[0].kind == "flip":
            # the flipped filter banks depend on the weights only: refreshed beside the forward pass, joined before backward
            at = len(self.ops)
            flip_evs = (new_event(), new_event())
            seq[at] = ("join", flip_evs[1])
            seq.insert(0, ("fork", (self.bwd_ops[0], flip_evs[0], flip_evs[1])))
        for kind, item in seq:
            if kind == "op" and side and item.kind == "wgrad":
                flush()
                plan.append(("side", (item, new_event())))
                continue
            if kind == "op" and item.kind not in self.DYNAMIC_KINDS:
                run.append(item)
                continue
            flush()
            plan.append((kind, item))
        flush()
        if not first:
            self._plan = plan
        return plan

    def close(self) -> None:
        lib = L.load()
        for g in getattr(self, "_step_graphs", []):
            lib.fcn_graph_destroy(g)
        self._step_graphs = []
        self._plan = None
        if getattr(self, "_side_stream", None):
            lib.fcn_stream_sync(self._side_stream)
            lib.fcn_stream_destroy(self._side_stream)
            self._side_stream = None
        super().close()

    def apply_update(self, grad_scale: float) -> None:
        sp, lib = self.solver, L.load()
        rate = sp.rate(self.iter)
        n = len(self.param_layout)
        if sp.kind == "ADAM":
            L.check(lib.fcn_adam_update_f32(self.param_flat.ptr, self.grad_flat.ptr, self.hist.ptr, self.hist2.ptr, self.segs_dev.ptr, n, rate,
                                            sp.momentum, sp.momentum2, sp.delta, sp.weight_decay, self.iter + 1, grad_scale, self.stream))
        else:
            L.check(lib.fcn_sgd_update_f32(self.param_flat.ptr, self.grad_flat.ptr, self.hist.ptr, self.segs_dev.ptr, n, rate, sp.momentum,
                                           sp.weight_decay, grad_scale, self.stream))

    # ------------------------------------------------------------------ parameters back to Caffe layout
    def download_params(self) -> Dict[str, List[np.ndarray]]:
        """Current parameters as Caffe-layout host arrays (conv: OIHW, bias)."""
        flat = np.empty(max(self.param_count, 4), F32)
        L.call("fcn_memcpy_d2h_async", flat.ctypes.data, self.param_flat.ptr, flat.nbytes, self.stream)
        L.call("fcn_stream_sync", self.stream)
        return self._unpack(flat)

    def download_grads(self) -> Dict[str, List[np.ndarray]]:
        flat = np.empty(max(self.param_count, 4), F32)
        L.call("fcn_memcpy_d2h_async", flat.ctypes.data, self.grad_flat.ptr, flat.nbytes, self.stream)
        L.call("fcn_stream_sync", self.stream)
        return self._unpack(flat)

    def _unpack(self, flat: np.ndarray) -> Dict[str, List[np.ndarray]]:
        out: Dict[str, List[np.ndarray]] = {}
        types = {l.name: l.type for l in self.spec.layers}
        for e in self.param_layout:
            a = flat[e["offset"]:e["offset"] + e["count"]].reshape(e["shape"])
            if e["index"] == 0 and types[e["layer"]] == "Convolution":
                host_shape = self.params_host[e["layer"]][0].shape
                a = a[..., :host_shape[1]].transpose(0, 3, 1, 2)
            elif e["index"] == 0 and types[e["layer"]] == "Deconvolution":
                a = a.reshape(self.params_host[e["layer"]][0].shape)
            out.setdefault(e["layer"], []).append(np.ascontiguousarray(a))
        return out

    def _pack(self, per_layer: Dict[str, List[np.ndarray]]) -> np.ndarray:
        """Inverse of _unpack: Caffe-layout blobs -> the flat device layout (padded input channels stay zero)."""
        flat = np.zeros(max(self.param_count, 4), F32)
        types = {l.name: l.type for l in self.spec.layers}
        for e in self.param_layout:
            a = np.asarray(per_layer[e["layer"]][e["index"]], F32)
            if e["index"] == 0 and types[e["layer"]] == "Convolution":
                co, ci, kh, kw = self.params_host[e["layer"]][0].shape
                d = np.zeros(e["shape"], F32)
                d[..., :ci] = a.reshape(co, ci, kh, kw).transpose(0, 2, 3, 1)
                a = d
            flat[e["offset"]:e["offset"] + e["count"]] = a.reshape(-1)
        return flat

    def download_history(self) -> List[np.ndarray]:
        """Solver history in Caffe's order: one blob per learnable parameter (SGD momentum / Adam m), then Adam's v blobs."""
        out: List[np.ndarray] = []
        for buf in (self.hist, self.hist2):
            if buf is None:
                continue
            flat = np.empty(max(self.param_count, 4), F32)
            L.call("fcn_memcpy_d2h_async", flat.ctypes.data, buf.ptr, flat.nbytes, self.stream)
            L.call("fcn_stream_sync", self.stream)
            per = self._unpack(flat)
            for l in self.spec.param_layers():
                out.extend(per[l.name])
        return out

    def upload_history(self, history: Sequence[np.ndarray]) -> None:
        bufs = [b for b in (self.hist, self.hist2) if b is not None]
        per_buf = sum(len(self.params_host[l.name]) for l in self.spec.param_layers())
        if len(history) != per_buf * len(bufs):
            raise ValueError("solver state holds %d history blobs, this solver needs %d" % (len(history), per_buf * len(bufs)))
        it = iter(history)
        for buf in bufs:
            per = {l.name: [next(it) for _ in self.params_host[l.name]] for l in self.spec.param_layers()}
            flat = self._pack(per)
            L.call("fcn_memcpy_h2d_async", buf.ptr, flat.ctypes.data, flat.nbytes, self.stream)
            L.call("fcn_stream_sync", self.stream)

    def read_grad(self, name: str) -> np.ndarray:
        """NCHW host copy of a blob's gradient (debug / tests)."""
        g = self.grad_blobs[name]
        n, c, h, w = g.shape
        raw = np.empty((n, h, w, g.cstride), F32)
        L.call("fcn_memcpy_d2h_async", raw.ctypes.data, g.buf.ptr, raw.nbytes, self.stream)
        L.call("fcn_stream_sync", self.stream)
        return np.ascontiguousarray(raw[..., g.coffset:g.coffset + c].transpose(0, 3, 1, 2))

    def save(self, path: str) -> None:
        params = self.download_params()
        layers = [(l.name, l.type, params[l.name]) for l in self.spec.param_layers()]
        proto.write_caffemodel(path, layers, self.spec.name)
