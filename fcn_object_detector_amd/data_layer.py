"""Host mirror of the reference's Caffe Python data layer, with label generation on the GPU.

Mirrors ``DataArgumentationLayer`` (reference: scripts/data_argumentation_layer/data_argumentation_layer.py:14-190)
and the deterministic geometry of ``ArgumentationEngine`` (reference: argumentation_engine.py:24-138, 241-303):
same ``param_str`` (``W,H,stride,num_classes,batch,train.txt``), same dataset-file convention (every 2nd line,
``img mask label x y w h``; labels re-indexed with ``np.unique``; a manifest is written), same six tops.  The per-cell
label tensors come from ``fcn_gen_targets`` (HIP), not from interpreted loops.

Differences that are deliberate and documented:
  * ``imgaug`` (argumentation_engine.py:308-322; an un-pinned, un-vendored submodule) is not available offline: its colour
    sequence is restated from the library's documented operators (plan_color below; pixel work in csrc/augment.hip) with
    this layer's own numpy generator for the parameter draws, so the images are statistically, not bitwise, those of imgaug;
  * the reference reads a hard-coded background JPEG (data_argumentation_layer.py:86); here the background is the
    ``FCN_BACKGROUND`` image if set, else seeded noise;
  * ``train.txt`` may be the keyword ``synthetic[:N]`` — N procedurally textured objects instead of image files
    (needed for benchmarks and tests: there is no dataset in the reference);
  * an optional 7th ``param_str`` field ``detectnet`` makes top[1] the per-cell coverage grid
    (``foreground_labels``, the line the reference has commented out at :107) instead of the full-resolution class
    mask HEAD emits (:113-121) — models/train_val.prototxt's EuclideanLoss needs the grid.
"""
from __future__ import annotations

import math
import os
import random
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .pylayer import Layer

MEAN_BGR = (104.0069879317889, 116.66876761696767, 122.6789143406786)


# ----------------------------------------------------------------------------
# deterministic geometry
# ----------------------------------------------------------------------------

def resize_rects(src_hw: Sequence[int], dst_wh: Sequence[int], rects: Sequence[Sequence[int]]) -> List[Tuple[int, int, int, int]]:
    """Rect part of resize_image_and_labels (argumentation_engine.py:114-138): float32 arithmetic, int() truncation."""
    ratio_x = np.float32(src_hw[1]) / np.float32(dst_wh[0])
    ratio_y = np.float32(src_hw[0]) / np.float32(dst_wh[1])
    out = []
    for r in rects:
        x, y, w, h = (np.float32(v) for v in r)
        xt, yt = x / ratio_x, y / ratio_y
        xb, yb = (x + w) / ratio_x, (y + h) / ratio_y
        out.append((int(xt), int(yt), int(xb - xt), int(yb - yt)))
    return out


def flip_rects(im_hw: Sequence[int], rects: Sequence[Sequence[int]], flip_flag: int) -> List[List[int]]:
    """Rect part of flip_image (argumentation_engine.py:241-267): mirrored corners with the -1 pixel convention."""
    H, W = int(im_hw[0]), int(im_hw[1])
    out = []
    for r in rects:
        p1 = (r[0], r[1])
        p2 = (r[0] + r[2], r[1] + r[3])
        if flip_flag == -1:
            p1, p2 = (W - p1[0] - 1, H - p1[1] - 1), (W - p2[0] - 1, H - p2[1] - 1)
        elif flip_flag == 0:
            p1, p2 = (p1[0], H - p1[1] - 1), (p2[0], H - p2[1] - 1)
        elif flip_flag == 1:
            p1, p2 = (W - p1[0] - 1, p1[1]), (W - p2[0] - 1, p2[1])
        x, y = min(p1[0], p2[0]), min(p1[1], p2[1])
        out.append([max(int(x), 0), max(int(y), 0), int(abs(p2[0] - p1[0])), int(abs(p2[1] - p1[1]))])
    return out


def plan_zoom(im_hw: Sequence[int], rect: Sequence[int], rect_flip: Sequence[int]):
    """The "zoom in" of random_argumentation (argumentation_engine.py:156-172) and crop_image_dimension (:190-236) for a
    scene that holds ONE box: the random draws (``random`` module, reference order) and the integer crop window.

    Returns (window (x, y, w, h) inside the image, new rect) or None when the reference's arithmetic leaves no pixels
    (the reference would fail inside cv.resize; here the sample keeps the uncropped scene).  Python-2 ``/`` on ints is
    floor division (`rect[2]/2`), kept as `//`."""
    H, W = int(im_hw[0]), int(im_hw[1])
    if rect[2] <= 0 or rect[3] <= 0:
        return None
    scale_x = int(math.floor(np.float32(W) / np.float32(rect[2])))
    scale_y = int(math.floor(np.float32(H) / np.float32(rect[3])))
    e1 = random.uniform(1.0, float(np.float32(scale_x / 1.0)))
    e2 = random.uniform(1.0, float(np.float32(scale_y / 1.0)))
    rx, ry, rw, rh = (int(v) for v in rect_flip)
    widths = (int(rw * e1), rw * e2)
    heights = (int(rh * e1), rh * e2)                     # sic: the first factor again (:167)
    x = (rx + rw // 2) - widths[0]
    y = (ry + rh // 2) - heights[0]
    w = widths[1] + widths[0]
    h = heights[1] + heights[0]
    cx, cy = rx + rw / 2.0, ry + rh / 2.0
    shift_x, shift_y = random.randint(0, int(w / 2)), random.randint(0, int(h / 2))
    cx = (cx + shift_x) if random.randint(0, 1) else (cx - shift_x)
    cy = (cy + shift_y) if random.randint(0, 1) else (cy - shift_y)
    nx, ny, nw, nh = int(cx - (w / 2)), int(cy - (h / 2)), int(w), int(h)
    if nx > x:
        nx = x                                            # (the width correction that follows in the reference subtracts |nx - x| = 0)
    if ny > y:
        ny = y
    if nx + nw < x + w:
        nx += (x + w) - (nx + nw)
    if ny + nh < y + h:
        ny += (y + h) - (ny + nh)
    x, y, w, h = nx, ny, nw, nh
    x = 0 if x < 0 else x
    y = 0 if y < 0 else y
    if x > W or y > H:
        return None                                       # the reference's negative-extent branch: an empty crop
    x0, y0 = int(x), int(y)
    x1, y1 = min(int(x + w), W), min(int(y + h), H)       # numpy slicing clips to the image
    if x1 <= x0 or y1 <= y0 or int(x + w) < 0 or int(y + h) < 0:
        return None
    return (x0, y0, x1 - x0, y1 - y0), [int(rx - x), int(ry - y), rw, rh]


def plan_color(rng: np.random.Generator) -> dict:
    """Parameter draws of color_space_argumentation (argumentation_engine.py:308-322): OneOf(GaussianBlur sigma (0, 3),
    AverageBlur k (2, 7), MedianBlur k (3, 7)) -> Sharpen(alpha (0, 1), lightness (0.75, 1.5)) -> Add((-2, 21), per_channel
    0.5) -> Multiply((0.75, 1.25), per_channel 0.5) -> Grayscale(alpha (0, 0.5)), in this order."""
    kind = ("gauss", "box", "median")[int(rng.integers(0, 3))]
    blur = dict(kind=kind)
    if kind == "gauss":
        blur["sigma"] = float(rng.uniform(0.0, 3.0))
    elif kind == "box":
        blur["k"] = int(rng.integers(2, 8))
    else:
        k = int(rng.integers(3, 8))
        blur["k"] = k - 1 if k % 2 == 0 else k            # imgaug: even median kernels are made odd by subtracting one
    sharpen = (float(rng.uniform(0.0, 1.0)), float(rng.uniform(0.75, 1.5)))
    add = [int(v) for v in rng.integers(-2, 22, 3)] if rng.random() < 0.5 else [int(rng.integers(-2, 22))] * 3
    mul = [float(v) for v in rng.uniform(0.75, 1.25, 3)] if rng.random() < 0.5 else [float(rng.uniform(0.75, 1.25))] * 3
    return dict(blur=blur, sharpen=sharpen, add=add, mul=mul, gray=float(rng.uniform(0.0, 0.5)))


def gauss_taps(sigma: float) -> np.ndarray:
    """Half of the symmetric Gaussian kernel (centre first), float32, normalised in float64; radius ceil(3.3 sigma)
    (3.3 sigma is where imgaug's cv2 path cuts the kernel for sigma below 3)."""
    r = max(int(math.ceil(3.3 * sigma)), 1)
    x = np.arange(0, r + 1, dtype=np.float64)
    w = np.exp(-0.5 * x * x / (sigma * sigma))
    w /= w[0] + 2.0 * w[1:].sum()
    return w.astype(np.float32)


GAUSS_MIN_SIGMA = 0.001       # imgaug skips the blur below this


# ----------------------------------------------------------------------------
# dataset
# ----------------------------------------------------------------------------

def parse_param_str(param_str: str) -> dict:
    """data_argumentation_layer.py:25-32 — six positional comma-separated fields (+ the optional mode field)."""
    p = [s.strip() for s in str(param_str).split(",")]
    if len(p) < 6:
        raise ValueError("Parameter string missing or data type is wrong!")
    try:
        out = dict(image_size_x=int(p[0]), image_size_y=int(p[1]), stride=int(p[2]), num_classes=int(p[3]), batch_size=int(p[4]),
                   train_fn=str(p[5]), mode=(p[6].lower() if len(p) > 6 else "mask"))
    except ValueError:
        raise ValueError("Parameter string missing or data type is wrong!")
    if out["mode"] not in ("mask", "detectnet"):
        raise ValueError("7th param_str field must be 'mask' or 'detectnet'")
    return out


def read_data_from_textfile2(train_fn: str, manifest_dir: Optional[str] = "snapshots/labels"):
    """data_argumentation_layer.py:158-190: every 2nd line `img mask label x y w h`; labels -> np.unique inverse."""
    with open(train_fn) as f:
        lines = [ln.rstrip("\n") for ln in f]
    img_paths, mask_imgs, labels, rects = [], [], [], []
    for index in range(0, len(lines), 2):
        parts = lines[index].split()
        if len(parts) < 7:
            continue
        img_paths.append(parts[0])
        mask_imgs.append(parts[1])
        labels.append(int(parts[2]))
        rects.append(np.array([int(float(v)) for v in parts[3:7]], dtype=np.int64))
    label_unique, label_indices = np.unique(np.array(labels), return_inverse=True)
    if manifest_dir:
        try:
            os.makedirs(manifest_dir, exist_ok=True)
            with open(os.path.join(manifest_dir, "labels_" + time.strftime("%Y%m%d%H%M%S") + ".txt"), "w") as f:
                for index, label in enumerate(label_unique):
                    f.write("%s\n" % (str(index + 1) + " " + str(label)))
        except OSError:
            pass
    return np.array(img_paths), np.array(mask_imgs), label_indices, np.array(rects)


def _load_image(path: str) -> Optional[np.ndarray]:
    """BGR uint8 like cv.imread; None when the file or the decoder is missing."""
    if not os.path.isfile(path):
        return None
    try:
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    except Exception:
        return None


class SyntheticObjects:
    """`synthetic[:N]` dataset: N textured blobs with elliptical masks, one class each (labels 0..N-1)."""

    def __init__(self, n: int, seed: int = 0):
        rng = np.random.default_rng(seed)
        self.items = []
        for k in range(n):
            h, w = int(rng.integers(60, 160)), int(rng.integers(60, 160))
            yy, xx = np.mgrid[0:h, 0:w]
            mask = (((yy - h / 2.0) / (h / 2.0)) ** 2 + ((xx - w / 2.0) / (w / 2.0)) ** 2 <= 1.0)
            tex = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            tex[:, :, k % 3] = np.clip(tex[:, :, k % 3].astype(int) + 80, 0, 255).astype(np.uint8)
            self.items.append((tex, (mask * 255).astype(np.uint8), k))


# ----------------------------------------------------------------------------
# the layer
# ----------------------------------------------------------------------------


class DataArgumentationLayer(Layer):
    """Six tops: data (N,3,H,W); top[1] class mask (N,1,H,W) or coverage grid (N,C,gy,gx); bbox-label, size-block, obj-block,
    coverage-block (N,4C,gy,gx)."""

    MAX_PLACEMENT_RETRIES = 100      # ArgumentationEngineMapping.__max_counter
    PLACEMENT_IOU = 0.05             # ArgumentationEngineMapping.__iou_thresh
    supports_device_targets = True   # forward() can stop at the boxes; the solver then builds the label grids in HBM
    supports_device_scenes = True    # bind_device(): scenes are composed and normalised on the device (the only renderer)

    def setup(self, bottom, top):
        if len(bottom) > 0:
            raise Exception("This layer takes no bottom")
        if len(top) < 6:
            raise Exception("Current Implementation needs 6 top blobs")
        cfg = parse_param_str(self.param_str)
        self.__dict__.update(cfg)
        self.randomize = True
        fn = self.train_fn
        if fn.startswith("synthetic"):
            n = int(fn.split(":")[1]) if ":" in fn else max(self.num_classes, 1)
            self.objects = SyntheticObjects(min(n, max(self.num_classes, 1)))
            self.dataset = None
        else:
            if not os.path.isfile(fn):
                raise ValueError("Provide the dataset textfile")
            self.img_paths, self.mask_imgs, self.labels, self.rects = read_data_from_textfile2(fn)
            if len(self.img_paths) == 0:
                raise ValueError("Provide the dataset textfile")
            self.dataset = True
        bg = _load_image(os.environ.get("FCN_BACKGROUND", ""))
        if bg is None:      # the reference hard-codes a JPEG path (:86); without one, a seeded noise image plays the background
            bg = np.random.default_rng(1234).integers(0, 256, (960, 1280, 3), dtype=np.uint8)
        self.background = bg
        self._sources = {}
        self.zoom_augmentation = os.environ.get("FCN_DATA_ZOOM", "1") != "0"     # random_argumentation's is_crop (:156)
        self.color_augmentation = os.environ.get("FCN_DATA_COLOR", "1") != "0"
        self._color_rng = np.random.default_rng(int(os.environ["FCN_DATA_SEED"]) + 1 if "FCN_DATA_SEED" in os.environ else None)
        if self.randomize:
            random.seed(int(os.environ["FCN_DATA_SEED"])) if "FCN_DATA_SEED" in os.environ else random.seed()

    def reshape(self, bottom, top):
        n = self.batch_size
        gx, gy = int(self.image_size_x / self.stride), int(self.image_size_y / self.stride)
        ch = int(self.num_classes * 4)
        top[0].reshape(n, 3, self.image_size_y, self.image_size_x)
        if self.mode == "detectnet":
            top[1].reshape(n, self.num_classes, gy, gx)
        else:
            top[1].reshape(n, 1, self.image_size_y, self.image_size_x)
        for i in range(2, 6):
            top[i].reshape(n, ch, gy, gx)

    # -- scene synthesis: ArgumentationEngineMapping.argument (argumentation_engine.py:651-746) -------------------------
    # The random DECISIONS (plan_scene, host) are separate from the pixel work, which is one kernel launch per sample on
    # the device (DeviceRenderer); oracle/scene_ref.py renders the same plan with numpy for the parity tests.
    SCENE_H, SCENE_W = 480, 640      # the reference composes at the background's 640x480 and resizes to the net input later

    def _source(self, idx: int):
        """(image uint8 HxWx3 BGR, mask uint8 HxW in {0,255}, label, rect) of dataset entry idx, cached on the host."""
        hit = self._sources.get(idx)
        if hit is not None:
            return hit
        if self.dataset is None:
            tex, mask, label = self.objects.items[idx]
            h, w = mask.shape
            item = (tex, mask, label, np.array([0, 0, w, h]))
        else:
            image, mask = _load_image(self.img_paths[idx]), _load_image(self.mask_imgs[idx])
            if image is None or mask is None:
                raise IOError("cannot read %s / %s" % (self.img_paths[idx], self.mask_imgs[idx]))
            item = (image, np.where(mask[:, :, 0] > 0, 255, 0).astype(np.uint8), int(self.labels[idx]), self.rects[idx].copy())
        self._sources[idx] = item
        return item

    def _num_sources(self) -> int:
        return len(self.objects.items) if self.dataset is None else len(self.img_paths)

    def _iou(self, a, b) -> float:
        x, y = max(a[0], b[0]), max(a[1], b[1])
        w, h = min(a[0] + a[2], b[0] + b[2]) - x, min(a[1] + a[3], b[1] + b[3]) - y
        if w < 0 or h < 0:
            return 0.0
        ux, uy = min(a[0], b[0]), min(a[1], b[1])
        uw, uh = max(a[0] + a[2], b[0] + b[2]) - ux, max(a[1] + a[3], b[1] + b[3]) - uy
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.float32(a[2] * a[3]) / np.float32(b[2] * b[3])
            return float(np.float32(w * h) / np.float32(uw * uh) / ratio)

    def plan_scene(self) -> dict:
        """All random draws of one sample, in the reference's order: background crop, number of objects, per object
        (entry, flip, rescale, position with up to 100 retries against the boxes already placed), whole-image flip."""
        im_y, im_x = self.SCENE_H, self.SCENE_W
        bgh, bgw = self.background.shape[:2]
        hh, ww = bgh // 2, bgw // 2
        bx, by = random.randint(0, ww), random.randint(0, hh)
        bx = bx - (bx + ww - bgw) if bx + ww > bgw else bx
        by = by - (by + hh - bgh) if by + hh > bgh else by
        objs, placed, labels = [], [], []
        for _ in range(random.randint(1, 3)):
            idx = random.randint(0, self._num_sources() - 1)
            image, _mask, label, rect = self._source(idx)
            sh, sw = image.shape[:2]
            flip_flag = random.randint(-1, 2)
            if -2 < flip_flag < 2:
                rect = np.array(flip_rects((sh, sw), [rect], flip_flag)[0])
            x, y, w, h = (int(v) for v in rect)
            w, h = max(min(x + w, sw) - x, 0), max(min(y + h, sh) - y, 0)      # numpy slicing clips the crop to the image
            roi = (x, y, w, h)
            if random.randint(0, 1):
                scale = random.uniform(1.0, 2.2)
                w, h = int(w * scale), int(h * scale)

            def draw():
                cx, cy = random.randint(0, im_x - 1), random.randint(0, im_y - 1)
                cx = cx - ((cx + w) - im_x) if cx + w > im_x - 1 else cx
                cy = cy - ((cy + h) - im_y) if cy + h > im_y - 1 else cy
                return cx, cy
            cx, cy = draw()
            nrect = np.array([cx, cy, w, h])
            found = True
            if placed and any(self._iou(b, nrect) > self.PLACEMENT_IOU for b in placed):
                found = False
                for _try in range(self.MAX_PLACEMENT_RETRIES + 1):
                    cx, cy = draw()
                    nrect = np.array([cx, cy, w, h])
                    if not any(self._iou(b, nrect) > self.PLACEMENT_IOU for b in placed):
                        found = True
                        break
            if not found or roi[2] <= 0 or roi[3] <= 0 or w <= 0 or h <= 0:
                continue
            if min(cx + w, im_x) <= max(cx, 0) or min(cy + h, im_y) <= max(cy, 0):
                continue
            objs.append(dict(idx=idx, flip=flip_flag, roi=roi, out=(w, h), pos=(cx, cy), label=label))
            placed.append(nrect)
            labels.append(label)
        rects = [tuple(int(v) for v in r) for r in placed]
        final_flip = random.randint(-1, 2)                        # random_argumentation (argumentation_engine.py:143-188)
        unflipped = rects
        if -2 < final_flip < 2 and rects:
            rects = [tuple(r) for r in flip_rects((im_y, im_x), rects, final_flip)]
        else:
            final_flip = 2
        view = None                                               # the zoom crops the IMAGE; the class mask stays whole (:148-172)
        if self.zoom_augmentation and len(rects) == 1:
            z = plan_zoom((im_y, im_x), unflipped[0], rects[0])
            if z is not None:
                view, rects = z[0], [tuple(z[1])]
        color = plan_color(self._color_rng) if self.color_augmentation else None
        src_hw = (view[3], view[2]) if view else (im_y, im_x)
        rects = resize_rects(src_hw, (self.image_size_x, self.image_size_y), rects)
        return dict(bg_crop=(bx, by, ww, hh), objects=objs, final_flip=final_flip, view=view, color=color, rects=rects, labels=labels)

    def bind_device(self, engine, top_names: Sequence[str]) -> None:
        """Called by the solver: from now on forward() renders `data` (and the class mask of HEAD's mode) straight into the
        engine's input blobs on the device instead of into the host tops."""
        self._renderer = DeviceRenderer(self, engine, top_names[0], top_names[1] if self.mode != "detectnet" else None)
        self.device_tops = {top_names[0]} | ({top_names[1]} if self.mode != "detectnet" else set())

    def forward(self, bottom, top):
        from .detector import generate_targets
        all_rects, all_labels = [], []
        renderer = getattr(self, "_renderer", None)
        if renderer is None:
            raise RuntimeError("DataArgumentationLayer: no engine bound (bind_device): the scenes are composed on the MI355X, there is no "
                               "host renderer in the product")
        for index in range(self.batch_size):
            plan = self.plan_scene()
            renderer.render(index, plan)
            all_rects.append(plan["rects"])
            all_labels.append(plan["labels"])
        renderer.commit()
        self.last_rects, self.last_labels = all_rects, all_labels
        if getattr(self, "device_targets", False):
            return              # the solver hands last_rects to TrainEngine.set_targets: labels are generated in HBM
        fg, bb, sz, ob, cv = generate_targets(all_rects, all_labels, self.image_size_x, self.image_size_y, self.stride, self.num_classes)
        if self.mode == "detectnet":
            top[1].data[...] = fg
        top[2].data[...] = bb
        top[3].data[...] = sz
        top[4].data[...] = ob
        top[5].data[...] = cv

    def backward(self, top, propagate_down, bottom):
        pass


class DeviceRenderer(object):
    """Renders planned scenes on the device (csrc/scene.hip + fcn_preprocess_bgr8): the background and every dataset
    entry are uploaded once and stay in HBM; a sample costs one 64-byte record per object, one compose launch, the
    normalise + resize launch into the engine's `data` blob and, in mask mode, the nearest-neighbour label launch."""

    def __init__(self, layer: DataArgumentationLayer, engine, data_top: str, label_top: Optional[str]):
        import ctypes as C
        from . import lib as L
        from .engine import DeviceBuffer
        self.C, self.L, self.DeviceBuffer = C, L, DeviceBuffer
        self.layer, self.engine = layer, engine
        self.data, self.label = engine.blobs[data_top], (engine.blobs[label_top] if label_top else None)
        L.call("fcn_init", engine.device)
        self.bg = self._upload(layer.background)
        self.entries = {}
        n = layer.batch_size
        h, w = layer.SCENE_H, layer.SCENE_W
        self.scene = [DeviceBuffer(h * w * 3, zero=False) for _ in range(n)]
        self.mask = [DeviceBuffer(h * w, zero=False) for _ in range(n)]
        self.recs_host = [(L.SceneObj * 4)() for _ in range(n)]
        self.recs_dev = [DeviceBuffer(C.sizeof(L.SceneObj) * 4, zero=True) for _ in range(n)]
        self.minmax = DeviceBuffer(64, zero=True)
        self.aug_a = [DeviceBuffer(h * w * 3, zero=False) for _ in range(n)]       # colour augmentation ping-pong buffers
        self.aug_b = [DeviceBuffer(h * w * 3, zero=False) for _ in range(n)]
        self.aug_f32 = DeviceBuffer(h * w * 3 * 4, zero=False)                    # Gaussian blur: row pass result (stream ordered)
        self.final = [None] * n
        # The renders run on their own stream into a staging copy of `data`, so that they overlap the training step that is
        # still reading the blob (conv1's weight gradient needs `data` until the very end of backward); commit() hands the
        # batch over on the engine's stream: wait for the renders, one device copy, the class-mask tops.
        sp = C.c_void_p()
        L.call("fcn_stream_create", C.byref(sp))
        self.side = sp
        self.stage = DeviceBuffer(self._blob_bytes(self.data), zero=True)
        self.ev_rendered, self.ev_taken = C.c_void_p(), C.c_void_p()
        L.call("fcn_event_create", C.byref(self.ev_rendered))
        L.call("fcn_event_create", C.byref(self.ev_taken))
        self._taken_pending = False
        self._batch_open = False

    @staticmethod
    def _blob_bytes(b) -> int:
        n, c, H, W = b.shape
        return n * H * W * b.cstride * 4

    def begin_batch(self) -> None:
        """First render of a batch: the staging buffer and the per-slot scratch must have been consumed by the last commit()."""
        if self._taken_pending:
            self.L.call("fcn_stream_wait_event", self.side, self.ev_taken)
            self._taken_pending = False
        self._batch_open = True

    def commit(self) -> None:
        """Hand the rendered batch to the engine (on the engine's stream, i.e. behind whatever step is still running)."""
        L, lay, st = self.L, self.layer, self.engine.stream
        L.call("fcn_event_record", self.ev_rendered, self.side)
        L.call("fcn_stream_wait_event", st, self.ev_rendered)
        d = self.data
        n, c, H, W = d.shape
        L.call("fcn_memcpy_d2d_async", d.ptr, self.stage.ptr, self._blob_bytes(d), st)
        if self.label is not None:
            lb = self.label
            for index in range(n):
                L.call("fcn_mask_to_label_f32", self.mask[index].ptr, lay.SCENE_H, lay.SCENE_W, lb.ptr + 4 * index * H * W * lb.cstride, H, W,
                       lb.cstride, st)
        L.call("fcn_event_record", self.ev_taken, st)
        self._taken_pending = True
        self._batch_open = False

    def _upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        d = self.DeviceBuffer(max(a.nbytes, 16), zero=False)
        self.L.call("fcn_memcpy_h2d_async", d.ptr, a.ctypes.data, a.nbytes, None)
        self.L.call("fcn_device_sync")
        return d

    def _entry(self, idx: int):
        e = self.entries.get(idx)
        if e is None:
            image, mask, _label, _rect = self.layer._source(idx)
            e = (self._upload(image), self._upload(mask), image.shape[0], image.shape[1])
            self.entries[idx] = e
        return e

    def render(self, index: int, plan: dict) -> None:
        """Enqueue the pixel work of one planned sample on the render stream (commit() makes the batch visible)."""
        L, C, lay, st = self.L, self.C, self.layer, self.side
        if not self._batch_open:
            self.begin_batch()
        recs = self.recs_host[index]
        objs = plan["objects"][:4]
        for i, o in enumerate(objs):
            img, msk, sh, sw = self._entry(o["idx"])
            recs[i] = L.SceneObj(img.ptr, msk.ptr, sh, sw, o["flip"], o["roi"][0], o["roi"][1], o["roi"][2], o["roi"][3], o["out"][0],
                                 o["out"][1], o["pos"][0], o["pos"][1], o["label"] + 1)
        if objs:
            L.call("fcn_memcpy_h2d_async", self.recs_dev[index].ptr, C.addressof(recs), C.sizeof(L.SceneObj) * len(objs), st)
        bx, by, ww, hh = plan["bg_crop"]
        bgh, bgw = lay.background.shape[:2]
        SH, SW = lay.SCENE_H, lay.SCENE_W
        view = plan.get("view")
        vx, vy, vw, vh = view if view else (0, 0, SW, SH)
        compose = [self.bg.ptr, bgh, bgw, bx, by, ww, hh, self.recs_dev[index].ptr, len(objs), plan["final_flip"]]
        if view:            # the zoom crops the image only: the class mask is rendered whole by a second, mask-only launch
            L.call("fcn_compose_scene_view_bgr8", *compose, self.scene[index].ptr, None, SH, SW, vx, vy, vw, vh, st)
            if self.label is not None:
                L.call("fcn_compose_scene_view_bgr8", *compose, None, self.mask[index].ptr, SH, SW, 0, 0, SW, SH, st)
        else:
            L.call("fcn_compose_scene_view_bgr8", *compose, self.scene[index].ptr, self.mask[index].ptr if self.label is not None else None,
                   SH, SW, 0, 0, SW, SH, st)
        img = self.scene[index]
        if plan.get("color"):
            img = self._color(index, img, vh, vw, plan["color"], st)
        self.final[index] = (img, vh, vw)
        d = self.data
        n, c, H, W = d.shape
        L.call("fcn_preprocess_bgr8", img.ptr, vh, vw, self.stage.ptr + 4 * index * H * W * d.cstride, H, W, d.cstride,
               float(getattr(d, "upload_shift", 0.0) or 0.0), self.minmax.ptr, st)

    def _color(self, index: int, img, h: int, w: int, color: dict, st):
        """color_space_argumentation on the device: blur (one of three) then the fused Sharpen/Add/Multiply/Grayscale pass;
        the image ping-pongs between the slot's two scratch buffers.  Returns the buffer holding the result."""
        L, C = self.L, self.C
        a, b2 = self.aug_a[index], self.aug_b[index]
        blur = color["blur"]
        if blur["kind"] == "gauss":
            if blur["sigma"] >= GAUSS_MIN_SIGMA:
                taps = gauss_taps(blur["sigma"])
                L.call("fcn_blur_gauss_bgr8", img.ptr, a.ptr, self.aug_f32.ptr, h, w, taps.ctypes.data, len(taps) - 1, st)
                img = a
        elif blur["kind"] == "box":
            L.call("fcn_blur_box_bgr8", img.ptr, a.ptr, h, w, blur["k"], st)
            img = a
        else:
            L.call("fcn_blur_median_bgr8", img.ptr, a.ptr, h, w, blur["k"], st)
            img = a
        al, light = color["sharpen"]
        ga = np.float32(color["gray"])
        q = L.ColorParams(float(np.float32((1.0 - al) + al * (8.0 + light))), float(np.float32(-al)), (C.c_int32 * 3)(*color["add"]),
                          (C.c_float * 3)(*[float(np.float32(m)) for m in color["mul"]]), float(ga), float(np.float32(1.0) - ga))
        L.call("fcn_color_augment_bgr8", img.ptr, b2.ptr, h, w, C.byref(q), st)
        return b2

    def read_scene(self, index: int):
        """Host copies of the final uint8 image (zoom window, colour augmented) and the class mask of batch slot `index` (tests)."""
        lay = self.layer
        buf, h, w = self.final[index]
        img = np.empty((h, w, 3), np.uint8)
        msk = np.empty((lay.SCENE_H, lay.SCENE_W), np.uint8)
        self.L.call("fcn_memcpy_d2h_async", img.ctypes.data, buf.ptr, img.nbytes, self.side)
        self.L.call("fcn_memcpy_d2h_async", msk.ctypes.data, self.mask[index].ptr, msk.nbytes, self.side)
        self.L.call("fcn_stream_sync", self.side)
        return img, msk
