"""Host mirror of the reference's inference node, with the arithmetic on the GPU.

Mirrors ``FCNObjectDetector`` (reference: scripts/fcn_object_detector.py) without
ROS: same method names, argument meaning and defaults —

  demean_rgb_image + cv.resize + transpose  (:79-82, :407-413)  -> fcn_preprocess_bgr8
  net.forward()                             (:87)               -> hipGraph replay
  gridbox_to_boxes + vote_boxes             (:337-394)          -> fcn_detect_decode_group
  np.asarray(dtype=int) + resize_detection  (:123-124, :396-405) -> host ints (5 numbers per box)

The node reads blobs ``pool_score`` / ``upscore_pool5_bbox`` at stride 8 and skips the
background channel (:89-90, :360); the shipped models/deploy.prototxt exposes
``coverage`` / ``bboxes`` at stride 16 without a background channel, so the
blob mapping is a parameter (:class:`HeadMapping`) with both presets.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import lib as L
from .engine import DeviceBuffer, Engine, PinnedArray

F32 = np.float32


class HeadMapping:
    """Which blobs hold coverage / boxes, the grid stride, and whether channel 0 is background."""

    def __init__(self, cvg_blob: str, bbox_blob: str, stride: int, skip_background: bool):
        self.cvg_blob, self.bbox_blob, self.stride, self.skip_background = cvg_blob, bbox_blob, int(stride), bool(skip_background)

    @classmethod
    def reference_node(cls) -> "HeadMapping":
        # fcn_object_detector.py:89-90 and :360 (stride = 16/2)
        return cls("pool_score", "upscore_pool5_bbox", 8, True)

    @classmethod
    def detectnet_deploy(cls) -> "HeadMapping":
        # models/deploy.prototxt:2144-2176; boundary_adjustment/boundary_refinement.py:265-302 (stride 16)
        return cls("coverage", "bboxes", 16, False)

    @classmethod
    def auto(cls, blob_names: Sequence[str]) -> "HeadMapping":
        if "pool_score" in blob_names and "upscore_pool5_bbox" in blob_names:
            return cls.reference_node()
        if "coverage" in blob_names and "bboxes" in blob_names:
            return cls.detectnet_deploy()
        raise KeyError("net exposes neither (pool_score, upscore_pool5_bbox) nor (coverage, bboxes)")


class GridDecoder:
    """Device-side gridbox_to_boxes + vote_boxes (cv.groupRectangles) for a fixed grid."""

    def __init__(self, batch: int, num_classes: int, gy: int, gx: int, im_w: int, im_h: int,
                 prob_thresh: float = 0.5, min_boxes: int = 3, eps: float = 0.2, min_height: int = 20,
                 round_mode: int = L.RECT_ROUND_NEAREST_EVEN, max_out: Optional[int] = None):
        self.batch, self.C, self.gy, self.gx = int(batch), int(num_classes), int(gy), int(gx)
        p = L.DetectParams()
        p.num_classes, p.gy, p.gx = self.C, self.gy, self.gx
        p.cell_w, p.cell_h = int(im_w) // self.gx, int(im_h) // self.gy     # :368-369 integer division
        p.prob_thresh, p.group_thresh, p.eps = float(np.float32(prob_thresh)), int(min_boxes), float(eps)
        p.min_height, p.round_mode = int(min_height), int(round_mode)
        p.max_out = int(max_out) if max_out else self.gy * self.gx
        self.params = p
        lib = L.load()
        slots = self.batch * self.C
        self.ws = DeviceBuffer(max(int(lib.fcn_detect_workspace_bytes(C.byref(p), self.batch)), 16), zero=True)
        self.d_rects = DeviceBuffer(slots * p.max_out * 16, zero=True)
        self.d_weights = DeviceBuffer(slots * p.max_out * 4, zero=True)
        self.d_count = DeviceBuffer(slots * 4, zero=True)
        # results land in pinned host memory: the read-back is then truly asynchronous (a pageable destination would hold the
        # host thread until the stream drains, which serialises the frames of a DetectorPipeline)
        self._pinned = [PinnedArray((slots, p.max_out, 4)), PinnedArray((slots, p.max_out)), PinnedArray((slots,))]
        self.h_rects, self.h_weights, self.h_count = (a.array.view(np.int32) for a in self._pinned)

    def launch(self, cvg_ptr: int, cvg_cstride: int, cvg_coffset: int, cvg_image_stride: int,
               box_ptr: int, box_cstride: int, box_coffset: int, box_image_stride: int, stream: Optional[int]) -> None:
        p = self.params
        p.cvg_cstride, p.cvg_coffset, p.box_cstride, p.box_coffset = cvg_cstride, cvg_coffset, box_cstride, box_coffset
        L.call("fcn_detect_decode_group", cvg_ptr, box_ptr, self.batch, cvg_image_stride, box_image_stride, C.byref(p),
               self.ws.ptr, self.d_rects.ptr, self.d_weights.ptr, self.d_count.ptr, stream)

    def fetch_begin(self, stream: Optional[int]) -> None:
        """Enqueue the read-back of the detections behind the decode launch (returns at once)."""
        L.call("fcn_memcpy_d2h_async", self.h_count.ctypes.data, self.d_count.ptr, self.h_count.nbytes, stream)
        L.call("fcn_memcpy_d2h_async", self.h_rects.ctypes.data, self.d_rects.ptr, self.h_rects.nbytes, stream)
        L.call("fcn_memcpy_d2h_async", self.h_weights.ctypes.data, self.d_weights.ptr, self.h_weights.nbytes, stream)

    def fetch(self, stream: Optional[int], begun: bool = False) -> List[Tuple[np.ndarray, np.ndarray]]:
        """Per image: (detections (D,5) float64 [x1,y1,x2,y2,log n] in reference order, labels (D,) int)."""
        if not begun:
            self.fetch_begin(stream)
        L.call("fcn_stream_sync", stream)
        out = []
        for img in range(self.batch):
            dets, labels = [], []
            for c in range(self.C):
                slot = img * self.C + c
                n = int(self.h_count[slot])
                if n < 0:
                    raise RuntimeError("decode: image %d, class %d has more candidate cells than the kernel's 5120 (include/fcnhip.h); "
                                       "raise the detection threshold" % (img, c))
                if n > self.params.max_out:
                    raise RuntimeError("detection overflow: %d clusters > max_out %d" % (n, self.params.max_out))
                for r, wgt in zip(self.h_rects[slot, :n], self.h_weights[slot, :n]):
                    dets.append([float(r[0]), float(r[1]), float(r[2]), float(r[3]), math.log(int(wgt))])   # :347-348
                    labels.append(c)
            out.append((np.asarray(dets, dtype=np.float64).reshape(-1, 5), np.asarray(labels, dtype=np.int64)))
        return out


class ScoreMasks:
    """Device side of what run_detector2 does with its score maps after net.forward() (scripts/fcn_object_detector.py:208-236 and
    create_mask_labels :279-303; csrc/mask.hip): threshold, x 255, cv.resize to the window, uint8 cast, OR into the frame-sized
    probability map, and per (window, class) the bounding rectangle of the largest contour."""

    def __init__(self, n_windows: int, num_classes: int, w: int, h: int, frame_h: int, frame_w: int):
        self.n, self.C, self.w, self.h, self.frame_h, self.frame_w = int(n_windows), int(num_classes), int(w), int(h), int(frame_h), int(frame_w)
        lib = L.load()
        self.maps = self.n * (self.C - 1)
        self.ws = DeviceBuffer(max(int(lib.fcn_score_masks_workspace_bytes(self.n, self.C, self.w, self.h)), 16), zero=False)
        self.d_pmap = DeviceBuffer((self.frame_h * self.frame_w + 3) // 4 * 4, zero=True)
        self.d_out = DeviceBuffer(max(self.maps, 1) * 5 * 4, zero=True)
        self._pinned = [PinnedArray(((self.frame_h * self.frame_w + 3) // 4,)), PinnedArray((max(self.maps, 1) * 5,))]
        self.h_pmap = self._pinned[0].array.view(np.uint8)[:self.frame_h * self.frame_w].reshape(self.frame_h, self.frame_w)
        self.h_out = self._pinned[1].array.view(np.int32).reshape(-1, 5)

    def launch(self, score_ptr: int, H: int, W: int, cstride: int, coffset: int, rects: np.ndarray, prob_thresh: float,
               stream: Optional[int]) -> None:
        rects = np.ascontiguousarray(rects, np.int32).reshape(self.n, 4)
        L.call("fcn_memset_async", self.d_pmap.ptr, 0, self.d_pmap.nbytes, stream)
        L.call("fcn_score_masks", score_ptr, self.n, self.C, H, W, cstride, coffset, rects.ctypes.data, float(np.float32(prob_thresh)),
               self.d_pmap.ptr, self.frame_h, self.frame_w, self.ws.ptr, self.d_out.ptr, stream)
        L.call("fcn_memcpy_d2h_async", self.h_pmap.ctypes.data, self.d_pmap.ptr, self.frame_h * self.frame_w, stream)
        L.call("fcn_memcpy_d2h_async", self.h_out.ctypes.data, self.d_out.ptr, self.maps * 20, stream)
        self._rects = rects

    def fetch(self, stream: Optional[int], padding: int = 10):
        """-> (pmap (frame_h, frame_w) uint8, [(np.array([x, y, w, h]), class index), ...]) in the reference's order (:218-236)."""
        L.call("fcn_stream_sync", stream)
        bboxs = []
        for n in range(self.n):
            x, y = int(self._rects[n, 0]), int(self._rects[n, 1])
            for c in range(1, self.C):
                found, rx, ry, rw, rh = (int(v) for v in self.h_out[n * (self.C - 1) + c - 1])
                if found:
                    bboxs.append((np.array([rx + x - padding, ry + y - padding, rw + 2 * padding, rh + 2 * padding]), c))
        return self.h_pmap.copy(), bboxs


def score_masks_from_maps(feature_maps: np.ndarray, rects, frame_hw: Tuple[int, int], prob_thresh: float = 0.5, padding: int = 10):
    """Run the device kernels of ScoreMasks on host score maps (N, C, H, W) float32 (NCHW, as net.blobs['score'].data)."""
    fm = np.ascontiguousarray(feature_maps, F32)
    n, c, h, w = fm.shape
    rects = np.ascontiguousarray(rects, np.int32).reshape(n, 4)
    L.call("fcn_init", 0)
    nhwc = np.ascontiguousarray(fm.transpose(0, 2, 3, 1))
    d = DeviceBuffer(nhwc.nbytes, zero=False)
    L.call("fcn_memcpy_h2d_async", d.ptr, nhwc.ctypes.data, nhwc.nbytes, None)
    sm = ScoreMasks(n, c, int(rects[0, 2]), int(rects[0, 3]), frame_hw[0], frame_hw[1])
    sm.launch(d.ptr, h, w, c, 0, rects, prob_thresh, None)
    return sm.fetch(None, padding)


def detect_from_maps(cvg: np.ndarray, bbox: np.ndarray, im_w: int, im_h: int, prob_thresh: float = 0.5,
                     min_boxes: int = 3, eps: float = 0.2, min_height: int = 20,
                     round_mode: int = L.RECT_ROUND_NEAREST_EVEN) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Run the device decode+group kernel on host maps: cvg (N,C,gy,gx), bbox (N,4C,gy,gx) float32 (NCHW)."""
    cvg = np.ascontiguousarray(cvg, F32)
    bbox = np.ascontiguousarray(bbox, F32)
    n, c, gy, gx = cvg.shape
    assert bbox.shape == (n, 4 * c, gy, gx)
    L.call("fcn_init", 0)
    cv_nhwc = np.ascontiguousarray(cvg.transpose(0, 2, 3, 1))
    bb_nhwc = np.ascontiguousarray(bbox.transpose(0, 2, 3, 1))
    d_c, d_b = DeviceBuffer(cv_nhwc.nbytes, zero=False), DeviceBuffer(bb_nhwc.nbytes, zero=False)
    L.call("fcn_memcpy_h2d_async", d_c.ptr, cv_nhwc.ctypes.data, cv_nhwc.nbytes, None)
    L.call("fcn_memcpy_h2d_async", d_b.ptr, bb_nhwc.ctypes.data, bb_nhwc.nbytes, None)
    dec = GridDecoder(n, c, gy, gx, im_w, im_h, prob_thresh, min_boxes, eps, min_height, round_mode)
    dec.launch(d_c.ptr, c, 0, gy * gx * c, d_b.ptr, 4 * c, 0, gy * gx * 4 * c, None)
    return dec.fetch(None)


def generate_targets(rects_per_image: Sequence[Sequence[Sequence[int]]], labels_per_image: Sequence[Sequence[int]],
                     im_w: int, im_h: int, stride: int, num_classes: int, iou_thresh: float = 0.1) -> Tuple[np.ndarray, ...]:
    """Device bounding_box_parameterized_labels for a batch: returns (foreground, bbox, size, obj, coverage_block) NCHW."""
    batch = len(rects_per_image)
    gy, gx = int(im_h) // int(stride), int(im_w) // int(stride)      # grid_region: Python-2 integer division (:284)
    offs = np.zeros(batch + 1, np.int32)
    flat_r, flat_l = [], []
    for i, (rs, ls) in enumerate(zip(rects_per_image, labels_per_image)):
        if len(rs) != len(ls):
            raise ValueError("image %d: %d rects but %d labels" % (i, len(rs), len(ls)))
        for r, lab in zip(rs, ls):
            if not 0 <= int(lab) < num_classes:
                raise IndexError("label %d outside [0, %d)" % (lab, num_classes))   # numpy would raise on the write (:107)
            flat_r.append([int(v) for v in r])
            flat_l.append(int(lab))
        offs[i + 1] = len(flat_r)
    rects = np.asarray(flat_r, np.int32).reshape(-1, 4)
    labels = np.asarray(flat_l, np.int32)
    L.call("fcn_init", 0)
    d_r = DeviceBuffer(max(rects.nbytes, 16), zero=False)
    d_l = DeviceBuffer(max(labels.nbytes, 16), zero=False)
    d_o = DeviceBuffer(offs.nbytes, zero=False)
    if rects.size:
        L.call("fcn_memcpy_h2d_async", d_r.ptr, rects.ctypes.data, rects.nbytes, None)
        L.call("fcn_memcpy_h2d_async", d_l.ptr, labels.ctypes.data, labels.nbytes, None)
    L.call("fcn_memcpy_h2d_async", d_o.ptr, offs.ctypes.data, offs.nbytes, None)
    G = gy * gx
    outs = [np.zeros((batch, num_classes, gy, gx), F32)] + [np.zeros((batch, 4 * num_classes, gy, gx), F32) for _ in range(4)]
    devs = [DeviceBuffer(o.nbytes, zero=False) for o in outs]
    L.call("fcn_gen_targets", d_r.ptr, d_l.ptr, d_o.ptr, batch, num_classes, gy, gx, int(stride), float(iou_thresh),
           devs[0].ptr, devs[1].ptr, devs[2].ptr, devs[3].ptr, devs[4].ptr, None)
    for o, d in zip(outs, devs):
        L.call("fcn_memcpy_d2h_async", o.ctypes.data, d.ptr, o.nbytes, None)
    L.call("fcn_device_sync")
    return tuple(outs)


def load_label_manifest(path: Optional[str], num_outputs: int = 0) -> List[str]:
    """Class names for the detections (scripts/fcn_object_detector.py:441-461).  The node expects three fields per line
    (`idx <anything> name`); the data layer writes two (`index label`, data_argumentation_layer.py:181-188) - both are read.
    Without a file the node's fallback names `object_<k-1>` are produced."""
    if path is None or not os.path.isfile(str(path)):
        return ["object_%d" % (i - 1) for i in range(num_outputs)]
    names = []
    with open(path) as f:
        for line in f:
            parts = line.rstrip("\n").split(" ")
            if len(parts) >= 3:
                names.append(parts[2])
            elif len(parts) == 2:
                names.append(parts[1])
    return names


def resize_detection(in_size: Sequence[int], boxes: np.ndarray, net_w: int, net_h: int) -> np.ndarray:
    """fcn_object_detector.py:396-405 — scale columns 0..3 in place on the integer array (truncating)."""
    diffx = float(in_size[1]) / float(net_w)
    diffy = float(in_size[0]) / float(net_h)
    for i in range(len(boxes)):
        boxes[i, 0] = boxes[i, 0] * diffx
        boxes[i, 1] = boxes[i, 1] * diffy
        boxes[i, 2] = boxes[i, 2] * diffx
        boxes[i, 3] = boxes[i, 3] * diffy
    return boxes


def detection_window_roi(im_shape: Sequence[int], stride: int = 2) -> np.ndarray:
    """Window geometry of fcn_object_detector.py:257-277: stride x stride windows in raster order, then the central one; rows are
    (x, y, w, h) int32.  Python-2 integer division throughout (im_x / stride, w / 2)."""
    im_y, im_x = int(im_shape[0]), int(im_shape[1])
    stride = int(stride)
    if stride < 1 or im_x < stride or im_y < stride:
        raise ValueError("detection_window_roi: stride %d does not divide a %d x %d frame into windows" % (stride, im_x, im_y))
    w, h = im_x // stride, im_y // stride
    rects = [(i * w, j * h, w, h) for j in range(stride) for i in range(stride)]
    rects.append((im_x // 2 - w // 2, im_y // 2 - h // 2, w, h))
    return np.asarray(rects, dtype=np.int32)


class FCNObjectDetector:
    """ROS-free mirror of the reference node: ``run_detector(frame_bgr_uint8) -> (boxes (D,5) int, labels (D,) int)``."""

    def __init__(self, engine: Engine, detection_threshold: float = 0.5, min_boxes: int = 3, nms_eps: float = 0.2,
                 mapping: Optional[HeadMapping] = None, round_mode: int = L.RECT_ROUND_NEAREST_EVEN):
        self.engine = engine
        self.prob_thresh, self.min_boxes, self.eps = detection_threshold, min_boxes, nms_eps   # :33-35 defaults
        self.mapping = mapping or HeadMapping.auto(list(engine.blobs))
        data = engine.blobs["data"]
        self.batch, _, self.im_height, self.im_width = data.shape
        cvg, box = engine.blobs[self.mapping.cvg_blob], engine.blobs[self.mapping.bbox_blob]
        skip = 1 if self.mapping.skip_background else 0
        self.num_classes = cvg.channels - skip
        gx, gy = int(self.im_width / self.mapping.stride), int(self.im_height / self.mapping.stride)   # :362-363
        if (gy, gx) != tuple(cvg.shape[2:]):
            raise ValueError("grid %dx%d from stride %d does not match blob %s %s" % (gy, gx, self.mapping.stride,
                                                                                     self.mapping.cvg_blob, cvg.shape))
        self.decoder = GridDecoder(self.batch, self.num_classes, gy, gx, self.im_width, self.im_height, self.prob_thresh,
                                   self.min_boxes, self.eps, 20, round_mode)
        self._cvg_args = (cvg.buf.ptr, cvg.cstride, cvg.coffset + skip, cvg.pixels // self.batch * cvg.cstride)
        self._box_args = (box.buf.ptr, box.cstride, box.coffset, box.pixels // self.batch * box.cstride)
        self._frame_dev: Optional[DeviceBuffer] = None
        self._minmax = DeviceBuffer(32)
        self._minmax_batch_holder: List[DeviceBuffer] = []

    def _half_flag(self, data) -> int:
        """dst_f16 of fcn_preprocess_bgr8_batch / _rois: 0 = float32 blob, 1 = half blob, 3 = the f16 engine's half image (8-half pixels
        whose channels 3 and 4 are the constant 1): the kernel then writes whole pixels in one store."""
        if data.esize != 2:
            return 0
        return 3 if data.name in getattr(self.engine, "_half_inputs", {}) and data.cstride == 8 and data.coffset == 0 else 1

    def run_detector_batch(self, frames: Sequence[np.ndarray]) -> List[Tuple[np.ndarray, np.ndarray]]:
        """BASELINE configs[4] minus the fp16 arithmetic: `batch` frames through pre-processing, ONE forward and ONE fused
        decode + groupRectangles launch ((image, class) per workgroup); per frame the result of run_detector."""
        self.submit_batch(frames)
        return self.collect_batch()

    def submit_batch(self, frames: Sequence[np.ndarray]) -> None:
        """Enqueue a whole batch (upload, pre-processing, forward, decode + grouping, read-back) without waiting."""
        eng = self.engine
        if len(frames) != self.batch:
            raise ValueError("need %d frames (the engine's batch), got %d" % (self.batch, len(frames)))
        if getattr(self, "_outstanding", None) is not None:
            raise RuntimeError("a batch is already in flight: collect_batch() it first")
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        if any(f.ndim != 3 or f.shape[2] != 3 for f in frames):
            raise ValueError("expected BGR uint8 frames")
        with eng.lock:
            L.call("fcn_init", eng.device)
            need = sum((f.nbytes + 15) // 16 * 16 for f in frames)
            if self._frame_dev is None or self._frame_dev.nbytes < need:
                self._frame_dev = DeviceBuffer(need, zero=False)
                self._frame_pinned = PinnedArray(((need + 3) // 4,))      # staging: an async copy needs pinned memory
            stage = self._frame_pinned.array.view(np.uint8)
            data = eng.blobs["data"]
            if len(self._minmax_batch_holder) == 0 or self._minmax_batch_holder[0].nbytes < 32 * len(frames):
                self._minmax_batch_holder[:] = [DeviceBuffer(32 * len(frames))]
            same = all(f.shape == frames[0].shape for f in frames)
            off, offs = 0, []
            for f in frames:
                stage[off:off + f.nbytes] = f.reshape(-1)
                offs.append(off)
                off += f.nbytes if same else (f.nbytes + 15) // 16 * 16
            L.call("fcn_memcpy_h2d_async", self._frame_dev.ptr, stage.ctypes.data, off, eng.stream)
            self._enqueue_batch(self._frame_dev.ptr, [(o, f.shape[0], f.shape[1]) for o, f in zip(offs, frames)], same)
            self._outstanding = [f.shape for f in frames]

    def _enqueue_batch(self, dev_ptr: int, layout: Sequence[Tuple[int, int, int]], same: bool) -> None:
        """Pre-processing of frames already in HBM (byte offset, h, w each), forward, decode launch, read-back: all enqueued."""
        eng = self.engine
        data = eng.blobs["data"]
        if same:      # one camera: the whole batch in three launches
            L.call("fcn_preprocess_bgr8_batch", dev_ptr, len(layout), layout[0][1], layout[0][2], data.ptr, self._half_flag(data),
                   self.im_height, self.im_width, data.cstride, data.upload_shift, self._minmax_batch_holder[0].ptr, eng.stream)
        else:
            for i, (off, h, w) in enumerate(layout):
                L.call("fcn_preprocess_bgr8_batch", dev_ptr + off, 1, h, w, data.ptr + data.esize * i * self.im_height * self.im_width * data.cstride,
                       self._half_flag(data), self.im_height, self.im_width, data.cstride, data.upload_shift, self._minmax.ptr, eng.stream)
        eng.forward_enqueue()
        self.decoder.launch(*self._cvg_args, *self._box_args, eng.stream)
        self.decoder.fetch_begin(eng.stream)
        data.host_valid = False

    def run_detector2(self, frame: np.ndarray, stride: int = 2) -> Tuple[np.ndarray, List[Tuple[np.ndarray, np.ndarray]]]:
        """The node's multi-window path without ROS (run_detector2 :178-211): the frame is demeaned and normalised as a whole,
        cut into stride x stride windows plus the central one (detection_window_roi :257-277), every window resized to the net's
        input, and ALL windows go through ONE batched forward (`net.blobs['data'].reshape(batch_size, ...)`; here the engine's
        batch must equal stride^2 + 1) and one fused decode + groupRectangles launch.  Returns (rects (n, 4) int32 x y w h,
        [(boxes (D, 5) int in FRAME coordinates, labels (D,))] per window); `net.blobs[...]` hold the windows' maps afterwards
        (the node reads 'score' there).  What the node does with its score maps next (cv.resize to the window, findContours,
        publishing) needs OpenCV / ROS and is not part of the tensor path."""
        eng = self.engine
        frame = np.ascontiguousarray(frame, np.uint8)
        if frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("expected a BGR uint8 frame")
        rects = detection_window_roi(frame.shape, stride)
        if len(rects) != self.batch:
            raise ValueError("stride %d makes %d windows; the engine's batch is %d" % (stride, len(rects), self.batch))
        if getattr(self, "_outstanding", None) is not None:
            raise RuntimeError("a batch is already in flight: collect it first")
        h, w, _c = frame.shape
        with eng.lock:
            L.call("fcn_init", eng.device)
            if self._frame_dev is None or self._frame_dev.nbytes < frame.nbytes:
                self._frame_dev = DeviceBuffer(frame.nbytes, zero=False)
                self._frame_pinned = PinnedArray(((frame.nbytes + 3) // 4,))
            stage = self._frame_pinned.array.view(np.uint8)[:frame.nbytes]
            stage[...] = frame.reshape(-1)
            data = eng.blobs["data"]
            L.call("fcn_memcpy_h2d_async", self._frame_dev.ptr, stage.ctypes.data, frame.nbytes, eng.stream)
            L.call("fcn_preprocess_bgr8_rois", self._frame_dev.ptr, h, w, rects.ctypes.data, len(rects), data.ptr, self._half_flag(data),
                   self.im_height, self.im_width, data.cstride, data.upload_shift, self._minmax.ptr, eng.stream)
            eng.forward_enqueue()
            self.decoder.launch(*self._cvg_args, *self._box_args, eng.stream)
            self.decoder.fetch_begin(eng.stream)
            data.host_valid = False
            res = self.decoder.fetch(eng.stream, begun=True)
        out = []
        for (x, y, rw, rh), (dets, labels) in zip(rects.tolist(), res):
            boxes = np.asarray(dets, dtype=np.int64).reshape(-1, 5)
            if len(boxes):
                boxes = resize_detection((rh, rw), boxes, self.im_width, self.im_height)
                boxes[:, 0] += x; boxes[:, 2] += x
                boxes[:, 1] += y; boxes[:, 3] += y
            out.append((boxes, labels))
        return rects, out

    def run_detector2_masks(self, frame: np.ndarray, stride: int = 1, score_blob: str = "score", padding: int = 10):
        """run_detector2 as the node runs it (scripts/fcn_object_detector.py:178-236; it calls detection_window_roi with stride 1): the
        frame is normalised as a whole and cut into stride^2 + 1 windows, ONE batched forward, then - on the device - the score maps
        of `score_blob` (the node reads net.blobs['score']) are thresholded at detection_threshold, scaled to 0..255, resized to their
        windows, OR-ed into the frame-sized probability map the node publishes on /fcn_object_detector/..., and every (window, class)
        map gives the padded bounding rectangle of its largest contour.  -> (pmap (h, w) uint8, [(np.array([x, y, w, h]), class), ...])."""
        eng = self.engine
        frame = np.ascontiguousarray(frame, np.uint8)
        if frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("expected a BGR uint8 frame")
        rects = detection_window_roi(frame.shape, stride)
        if len(rects) != self.batch:
            raise ValueError("stride %d makes %d windows; the engine's batch is %d" % (stride, len(rects), self.batch))
        sb = eng.blobs[score_blob]
        if sb.esize != 4 or len(sb.shape) != 4:
            raise NotImplementedError("score blob %s must be a 4-d float32 blob" % score_blob)
        h, w, _c = frame.shape
        _n, C_, sh, sw = sb.shape
        key = (len(rects), C_, int(rects[0][2]), int(rects[0][3]), h, w)
        if getattr(self, "_score_masks_key", None) != key:
            self._score_masks = ScoreMasks(*key)
            self._score_masks_key = key
        with eng.lock:
            L.call("fcn_init", eng.device)
            if self._frame_dev is None or self._frame_dev.nbytes < frame.nbytes:
                self._frame_dev = DeviceBuffer(frame.nbytes, zero=False)
                self._frame_pinned = PinnedArray(((frame.nbytes + 3) // 4,))
            stage = self._frame_pinned.array.view(np.uint8)[:frame.nbytes]
            stage[...] = frame.reshape(-1)
            data = eng.blobs["data"]
            L.call("fcn_memcpy_h2d_async", self._frame_dev.ptr, stage.ctypes.data, frame.nbytes, eng.stream)
            L.call("fcn_preprocess_bgr8_rois", self._frame_dev.ptr, h, w, rects.ctypes.data, len(rects), data.ptr, self._half_flag(data),
                   self.im_height, self.im_width, data.cstride, data.upload_shift, self._minmax.ptr, eng.stream)
            eng.forward_enqueue()
            for op in eng._lazy_blob_ops.get(score_blob, ()):
                op.run(eng.stream)
            self._score_masks.launch(sb.buf.ptr, sh, sw, sb.cstride, sb.coffset, rects, self.prob_thresh, eng.stream)
            data.host_valid = False
            return self._score_masks.fetch(eng.stream, padding)

    def collect_batch(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        if getattr(self, "_outstanding", None) is None:
            raise RuntimeError("no batch in flight")
        shapes, self._outstanding = self._outstanding, None
        with self.engine.lock:
            res = self.decoder.fetch(self.engine.stream, begun=True)
        out = []
        for shape, (dets, labels) in zip(shapes, res):
            boxes = np.asarray(dets, dtype=np.int64).reshape(-1, 5)
            out.append((resize_detection(shape, boxes, self.im_width, self.im_height) if len(boxes) else boxes, labels))
        return out

    def submit(self, frame: np.ndarray) -> None:
        """Enqueue one frame end to end - upload, pre-processing, forward, decode + grouping, read-back - on the engine's stream
        and return without waiting; collect() delivers the detections.  One frame may be outstanding per detector."""
        eng = self.engine
        if self.batch != 1:
            raise ValueError("run_detector handles one frame; build the engine with batch 1")
        if getattr(self, "_outstanding", None) is not None:
            raise RuntimeError("a frame is already in flight: collect() it first")
        frame = np.ascontiguousarray(frame, np.uint8)
        if frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("expected a BGR uint8 frame")
        h, w, _c = frame.shape
        with eng.lock:
            L.call("fcn_init", eng.device)
            if self._frame_dev is None or self._frame_dev.nbytes < frame.nbytes:
                self._frame_dev = DeviceBuffer(frame.nbytes, zero=False)
                self._frame_pinned = PinnedArray(((frame.nbytes + 3) // 4,))      # staging: an async copy needs pinned memory
            stage = self._frame_pinned.array.view(np.uint8)[:frame.nbytes]
            stage[...] = frame.reshape(-1)
            data = eng.blobs["data"]
            L.call("fcn_memcpy_h2d_async", self._frame_dev.ptr, stage.ctypes.data, frame.nbytes, eng.stream)
            L.call("fcn_preprocess_bgr8_batch", self._frame_dev.ptr, 1, h, w, data.ptr, self._half_flag(data), self.im_height, self.im_width,
                   data.cstride, data.upload_shift, self._minmax.ptr, eng.stream)
            eng.forward_enqueue()
            self.decoder.launch(*self._cvg_args, *self._box_args, eng.stream)
            self.decoder.fetch_begin(eng.stream)
            data.host_valid = False
            self._outstanding = frame.shape

    def collect(self) -> Tuple[np.ndarray, np.ndarray]:
        if getattr(self, "_outstanding", None) is None:
            raise RuntimeError("no frame in flight")
        shape, self._outstanding = self._outstanding, None
        with self.engine.lock:
            dets, labels = self.decoder.fetch(self.engine.stream, begun=True)[0]
        boxes = np.asarray(dets, dtype=np.int64).reshape(-1, 5)        # :123 np.asarray(..., dtype=np.int) truncates
        if not len(boxes):
            return boxes, labels
        return resize_detection(shape, boxes, self.im_width, self.im_height), labels

    def run_detector(self, frame: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        self.submit(frame)
        return self.collect()


class DetectorPipeline:
    """`depth` detectors on replica engines (engine.ForwardPipeline's idea for the whole node: upload, pre-processing,
    forward, decode + grouping and read-back of consecutive camera frames overlap on the device).  Results are those of
    run_detector, in frame order."""

    def __init__(self, make_engine, depth: int = 3, **detector_kw):
        self.detectors: List[FCNObjectDetector] = []
        first = None
        for _ in range(max(int(depth), 1)):
            eng = make_engine(first)            # make_engine(tune_from) -> Engine (batch 1)
            first = first or eng
            self.detectors.append(FCNObjectDetector(eng, **detector_kw))
        self._queue: List[FCNObjectDetector] = []
        self._next = 0

    def submit(self, frame: np.ndarray) -> None:
        if len(self._queue) >= len(self.detectors):
            raise RuntimeError("DetectorPipeline: %d frames already in flight" % len(self._queue))
        d = self.detectors[self._next]
        self._next = (self._next + 1) % len(self.detectors)
        d.submit(frame)
        self._queue.append(d)

    def collect(self) -> Tuple[np.ndarray, np.ndarray]:
        if not self._queue:
            raise RuntimeError("DetectorPipeline: nothing in flight")
        return self._queue.pop(0).collect()

    def run_detector_batches(self, batches) -> List[List[Tuple[np.ndarray, np.ndarray]]]:
        """Batched engines: consecutive batches overlap (the read-back and host-side unpacking of one with the forward of the next)."""
        out, queue, nxt = [], [], 0
        for frames in batches:
            if len(queue) == len(self.detectors):
                out.append(queue.pop(0).collect_batch())
            d = self.detectors[nxt]
            nxt = (nxt + 1) % len(self.detectors)
            d.submit_batch(frames)
            queue.append(d)
        while queue:
            out.append(queue.pop(0).collect_batch())
        return out

    def run_detector_stream(self, frames) -> List[Tuple[np.ndarray, np.ndarray]]:
        out = []
        for f in frames:
            if len(self._queue) == len(self.detectors):
                out.append(self.collect())
            self.submit(f)
        while self._queue:
            out.append(self.collect())
        return out

    def close(self) -> None:
        for d in self.detectors:
            d.engine.close()
