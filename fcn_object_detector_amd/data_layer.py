"""Host mirror of the reference's Caffe Python data layer, with label generation on the GPU.

Mirrors ``DataArgumentationLayer`` (reference: scripts/data_argumentation_layer/data_argumentation_layer.py:14-190)
and the deterministic geometry of ``ArgumentationEngine`` (reference: argumentation_engine.py:24-138, 241-303):
same ``param_str`` (``W,H,stride,num_classes,batch,train.txt``), same dataset-file convention (every 2nd line,
``img mask label x y w h``; labels re-indexed with ``np.unique``; a manifest is written), same six tops.  The per-cell
label tensors come from ``fcn_gen_targets`` (HIP), not from interpreted loops.

Differences that are deliberate and documented:
  * ``imgaug`` colour augmentation (argumentation_engine.py:308-322) is not available offline and is skipped;
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


def flip_image(img: np.ndarray, flip_flag: int) -> np.ndarray:
    """cv.flip: 0 = around the x axis (vertical), 1 = around the y axis (horizontal), -1 = both."""
    if flip_flag == 0:
        return img[::-1].copy()
    if flip_flag == 1:
        return img[:, ::-1].copy()
    return img[::-1, ::-1].copy()


def demean_rgb_image(im: np.ndarray) -> np.ndarray:
    """argumentation_engine.py:297-303: float32, subtract the BGR mean, min-max normalise the whole image to [0, 1]."""
    im = im.astype(np.float32)
    for c in range(3):
        im[:, :, c] -= np.float32(MEAN_BGR[c])
    return (im - im.min()) / (im.max() - im.min())


def resize_bilinear(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv.resize(img, (W, H)) with the default INTER_LINEAR (the reference's INTER_CUBIC lands in the dst slot, :120)."""
    h, w = img.shape[:2]
    if (h, w) == (H, W):
        return img.copy()

    def coords(n_out, n_in):
        f = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / float(n_out)) - 0.5
        f = f.astype(np.float32)
        s = np.floor(f).astype(np.int64)
        fr = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        fr[lo], s[lo] = 0, 0
        hi = s >= n_in - 1
        fr[hi], s[hi] = 0, n_in - 1
        return s, np.minimum(s + 1, n_in - 1), fr

    x0, x1, fx = coords(W, w)
    y0, y1, fy = coords(H, h)
    src = img.astype(np.float32) if img.dtype != np.float64 else img
    fx = fx[None, :, None] if img.ndim == 3 else fx[None, :]
    fy = fy[:, None, None] if img.ndim == 3 else fy[:, None]
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx
    out = top * (1 - fy) + bot * fy
    return np.rint(out).clip(0, 255).astype(np.uint8) if img.dtype == np.uint8 else out.astype(img.dtype)


def resize_nearest(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv.resize(..., interpolation=INTER_NEAREST): src index = floor(dst * scale)."""
    h, w = img.shape[:2]
    ys = np.minimum((np.arange(H) * (h / float(H))).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(W) * (w / float(W))).astype(np.int64), w - 1)
    return img[ys][:, xs].copy()


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
        self.background = bg
        self._noise = np.random.default_rng(1234)
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

    # -- scene synthesis: ArgumentationEngineMapping.argument (argumentation_engine.py:651-746), vectorised paste -----
    def _object(self):
        if self.dataset is None:
            tex, mask, label = self.objects.items[random.randint(0, len(self.objects.items) - 1)]
            h, w = mask.shape
            return tex.copy(), np.repeat(mask[:, :, None], 3, axis=2), label, np.array([0, 0, w, h])
        idx = random.randint(0, len(self.img_paths) - 1)
        image, mask = _load_image(self.img_paths[idx]), _load_image(self.mask_imgs[idx])
        if image is None or mask is None:
            raise IOError("cannot read %s / %s" % (self.img_paths[idx], self.mask_imgs[idx]))
        mask = np.where(mask > 0, 255, 0).astype(np.uint8)
        return image, mask, int(self.labels[idx]), self.rects[idx].copy()

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

    def synthesize(self, num_proposals: int, im_bg: np.ndarray):
        im_y, im_x = im_bg.shape[:2]
        img_out = im_bg.copy()
        mask_out = np.zeros((im_y, im_x, 1), np.uint8)
        placed: List[np.ndarray] = []
        labels: List[int] = []
        for _ in range(num_proposals):
            image, mask, label, rect = self._object()
            flip_flag = random.randint(-1, 2)
            if -2 < flip_flag < 2:
                rect = np.array(flip_rects(image.shape[:2], [rect], flip_flag)[0])
                image, mask = flip_image(image, flip_flag), flip_image(mask, flip_flag)
            x, y, w, h = (int(v) for v in rect)
            im_roi, im_msk = image[y:y + h, x:x + w].copy(), mask[y:y + h, x:x + w].copy()
            h, w = im_roi.shape[:2]
            if random.randint(0, 1):
                scale = random.uniform(1.0, 2.2)
                w, h = int(w * scale), int(h * scale)
                im_roi, im_msk = resize_bilinear(im_roi, w, h), resize_bilinear(im_msk, w, h)

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
            if not found:
                continue
            x0, y0 = max(cx, 0), max(cy, 0)
            x1, y1 = min(cx + w, im_x), min(cy + h, im_y)
            if x1 <= x0 or y1 <= y0:
                continue
            sel = im_msk[y0 - cy:y1 - cy, x0 - cx:x1 - cx, 0] > 0
            img_out[y0:y1, x0:x1][sel] = im_roi[y0 - cy:y1 - cy, x0 - cx:x1 - cx][sel]
            mask_out[y0:y1, x0:x1, 0][sel] = label + 1
            placed.append(nrect)
            labels.append(label)
        return img_out, mask_out, [tuple(int(v) for v in r) for r in placed], labels

    def _background(self, h: int = 480, w: int = 640) -> np.ndarray:
        if self.background is None:
            return self._noise.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im = self.background
        hh, ww = im.shape[0] // 2, im.shape[1] // 2
        x, y = random.randint(0, ww), random.randint(0, hh)
        x = x - (x + ww - im.shape[1]) if x + ww > im.shape[1] else x
        y = y - (y + hh - im.shape[0]) if y + hh > im.shape[0] else y
        return resize_bilinear(im[y:y + hh, x:x + ww], w, h)

    def make_sample(self):
        """One training sample: (image float32 HxWx3 in [0,1], class mask HxW uint8, rects at net resolution, labels)."""
        bg = self._background()
        img, mask, rects, labels = self.synthesize(random.randint(1, 3), bg)
        flip_flag = random.randint(-1, 2)                       # random_argumentation (argumentation_engine.py:143-188)
        if -2 < flip_flag < 2 and rects:
            rects = [tuple(r) for r in flip_rects(img.shape[:2], rects, flip_flag)]
            img, mask = flip_image(img, flip_flag), flip_image(mask, flip_flag)
        src_hw = img.shape[:2]
        img = demean_rgb_image(img)
        img = resize_bilinear(img, self.image_size_x, self.image_size_y)
        rects = resize_rects(src_hw, (self.image_size_x, self.image_size_y), rects)
        mask = resize_nearest(mask[:, :, 0], self.image_size_x, self.image_size_y)
        return img, mask, rects, labels

    def forward(self, bottom, top):
        from .detector import generate_targets
        all_rects, all_labels = [], []
        for index in range(self.batch_size):
            img, mask, rects, labels = self.make_sample()
            top[0].data[index] = img.transpose((2, 0, 1))
            if self.mode != "detectnet":
                top[1].data[index, 0] = mask
            all_rects.append(rects)
            all_labels.append(labels)
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
