"""ORACLE — TEST INFRASTRUCTURE ONLY.  Parity unpinned (no reference tests/fixtures exist).

CPU restatement of the reference's host-side hot-path logic:

  * ``gridbox_to_boxes``      reference: scripts/fcn_object_detector.py:357-394
  * ``vote_boxes``            reference: scripts/fcn_object_detector.py:337-351
  * ``group_rectangles``      OpenCV 3 ``cv::groupRectangles`` / ``cv::partition`` / ``SimilarRects``
                              (objdetect/cascadedetect.cpp, core/operations.hpp — NOT vendored by the
                              reference; restated from the published algorithm, un-pinned version)
  * ``to_rect``               the cv2 Python->``vector<Rect>`` converter: each float goes through
                              ``saturate_cast<int>(double)`` = cvRound = round-half-to-even
                              (``round_mode='trunc'`` gives the C-cast alternative SURVEY.md row A8 assumed)
  * ``resize_detection``      reference: scripts/fcn_object_detector.py:396-405
  * ``demean_rgb_image`` + bilinear resize   reference: scripts/fcn_object_detector.py:79-82,407-413
  * ``JaccardCoeff`` / ``bounding_box_parameterized_labels``
                              reference: scripts/data_argumentation_layer/argumentation_engine.py:24-109,272-292
  * ``resize_rects`` / ``flip_rects``        reference: argumentation_engine.py:114-138, 241-267

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np


# ---------------------------------------------------------------------------
# A7: gridbox_to_boxes
# ---------------------------------------------------------------------------

def gridbox_to_boxes(net_cvg: np.ndarray, net_boxes: np.ndarray, prob_thresh: float, im_w: int, im_h: int, stride: int):
    """net_cvg (gy,gx) f32, net_boxes (4,gy,gx) f32 -> boxes (M,4) float64 in np.where (row-major) order."""
    grid_sz_x = int(im_w / stride)
    grid_sz_y = int(im_h / stride)
    cell_width = im_w // grid_sz_x
    cell_height = im_h // grid_sz_y
    cvg_val = net_cvg[0:grid_sz_y][0:grid_sz_x]          # the double row-slice of the reference (:371)
    mask = cvg_val >= np.float32(prob_thresh)
    y, x = np.where(mask)
    mx = x * cell_width
    my = y * cell_height
    x1 = net_boxes[0][y, x].astype(np.float32) + mx     # float32 + int64 -> float64
    y1 = net_boxes[1][y, x].astype(np.float32) + my
    x2 = net_boxes[2][y, x].astype(np.float32) + mx
    y2 = net_boxes[3][y, x].astype(np.float32) + my
    boxes = np.transpose(np.vstack((x1, y1, x2, y2))).astype(np.float64).reshape(-1, 4)
    return boxes, mask


# ---------------------------------------------------------------------------
# A8: cv.groupRectangles
# ---------------------------------------------------------------------------

def cv_round(v: float) -> int:
    """cvRound: round half to even (lrint in the default rounding mode)."""
    return int(np.rint(v))


def to_rect(box: Sequence[float], round_mode: str = "nearest_even") -> Tuple[int, int, int, int]:
    if round_mode == "trunc":
        return tuple(int(math.trunc(v)) for v in box)
    return tuple(cv_round(float(v)) for v in box)


def similar_rects(r1, r2, eps: float) -> bool:
    delta = eps * (min(r1[2], r2[2]) + min(r1[3], r2[3])) * 0.5
    return (abs(r1[0] - r2[0]) <= delta and abs(r1[1] - r2[1]) <= delta and
            abs(r1[0] + r1[2] - r2[0] - r2[2]) <= delta and abs(r1[1] + r1[3] - r2[1] - r2[3]) <= delta)


def partition(vec: List[tuple], eps: float) -> Tuple[List[int], int]:
    """cv::partition — literal union-find transcription of the published algorithm (O(N^2))."""
    n = len(vec)
    parent = [-1] * n
    rank = [0] * n
    for i in range(n):
        root = i
        while parent[root] >= 0:
            root = parent[root]
        for j in range(n):
            if i == j or not similar_rects(vec[i], vec[j], eps):
                continue
            root2 = j
            while parent[root2] >= 0:
                root2 = parent[root2]
            if root2 != root:
                if rank[root] > rank[root2]:
                    parent[root2] = root
                else:
                    parent[root] = root2
                    rank[root2] += 1 if rank[root] == rank[root2] else 0
                    root = root2
                k = j
                while parent[k] >= 0:
                    k2 = parent[k]
                    parent[k] = root
                    k = k2
                k = i
                while parent[k] >= 0:
                    k2 = parent[k]
                    parent[k] = root
                    k = k2
    labels = [0] * n
    nclasses = 0
    cls_of_root = {}
    for i in range(n):
        root = i
        while parent[root] >= 0:
            root = parent[root]
        if root not in cls_of_root:
            cls_of_root[root] = nclasses
            nclasses += 1
        labels[i] = cls_of_root[root]
    return labels, nclasses


def partition_fast(rects: np.ndarray, eps: float) -> Tuple[np.ndarray, int]:
    """Same result as :func:`partition` (components numbered by first member) via a vectorised adjacency."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components
    n = len(rects)
    if n == 0:
        return np.zeros(0, np.int64), 0
    r = rects.astype(np.int64)
    x, y, w, h = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    rows, cols = [], []
    for i in range(n):
        delta = eps * (np.minimum(w[i], w) + np.minimum(h[i], h)).astype(np.float64) * 0.5
        sim = ((np.abs(x[i] - x) <= delta) & (np.abs(y[i] - y) <= delta) &
               (np.abs(x[i] + w[i] - x - w) <= delta) & (np.abs(y[i] + h[i] - y - h) <= delta))
        js = np.nonzero(sim)[0]
        rows.extend([i] * len(js))
        cols.extend(js.tolist())
    adj = csr_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n, n))
    ncomp, comp = connected_components(adj, directed=False)
    remap = {}
    labels = np.zeros(n, np.int64)
    for i in range(n):
        c = comp[i]
        if c not in remap:
            remap[c] = len(remap)
        labels[i] = remap[c]
    return labels, ncomp


def group_rectangles(rect_list: List[tuple], group_threshold: int, eps: float, fast: bool = False):
    """cv::groupRectangles(rectList, weights, groupThreshold, eps) -> (rects, weights)."""
    if group_threshold <= 0 or len(rect_list) == 0:
        return list(rect_list), [1] * len(rect_list)
    if fast:
        labels, nclasses = partition_fast(np.asarray(rect_list, np.int64).reshape(-1, 4), eps)
    else:
        labels, nclasses = partition(rect_list, eps)
    rrects = [[0, 0, 0, 0] for _ in range(nclasses)]
    rweights = [0] * nclasses
    for r, cls in zip(rect_list, labels):
        cls = int(cls)
        for k in range(4):
            rrects[cls][k] += int(r[k])
        rweights[cls] += 1
    for i in range(nclasses):
        s = np.float32(1.0) / np.float32(rweights[i])
        rrects[i] = [cv_round(float(np.float32(v) * s)) for v in rrects[i]]     # saturate_cast<int>(r.x * s), float math
    out_r, out_w = [], []
    for i in range(nclasses):
        r1, n1 = rrects[i], rweights[i]
        if n1 <= group_threshold:
            continue
        keep = True
        for j in range(nclasses):
            n2 = rweights[j]
            if j == i or n2 <= group_threshold:
                continue
            r2 = rrects[j]
            dx = cv_round(r2[2] * eps)
            dy = cv_round(r2[3] * eps)
            if (r1[0] >= r2[0] - dx and r1[1] >= r2[1] - dy and r1[0] + r1[2] <= r2[0] + r2[2] + dx and
                    r1[1] + r1[3] <= r2[1] + r2[3] + dy and (n2 > max(3, n1) or n1 < 3)):
                keep = False
                break
        if keep:
            out_r.append(tuple(r1))
            out_w.append(n1)
    return out_r, out_w


def vote_boxes(propose_boxes: np.ndarray, min_boxes: int, eps: float, round_mode: str = "nearest_even",
               min_height: int = 20, fast: bool = False) -> List[list]:
    """reference vote_boxes (:337-351): detections [x, y, w, h, log(n)] as the reference stores them."""
    detections = []
    if not np.asarray(propose_boxes).any():
        return detections
    rects = [to_rect(b, round_mode) for b in np.asarray(propose_boxes).tolist()]
    nboxes, weights = group_rectangles(rects, min_boxes, eps, fast=fast)
    for rect, weight in zip(nboxes, weights):
        if (rect[3] - rect[1]) >= min_height:
            detections.append([rect[0], rect[1], rect[2], rect[3], math.log(weight)])
    return detections


def detect(cvg: np.ndarray, bbox: np.ndarray, im_w: int, im_h: int, stride: int, prob_thresh: float = 0.5,
           min_boxes: int = 3, eps: float = 0.2, round_mode: str = "nearest_even", fast: bool = False):
    """The per-class loop of run_detector (:104-118) for one image: cvg (C,gy,gx), bbox (4C,gy,gx)."""
    boxes, labels = [], []
    for index, p_map in enumerate(cvg):
        prop, _ = gridbox_to_boxes(p_map, bbox[4 * index:4 * index + 4], prob_thresh, im_w, im_h, stride)
        for b in vote_boxes(prop, min_boxes, eps, round_mode, fast=fast):
            boxes.append(b)
            labels.append(index)
    return np.asarray(boxes, dtype=np.float64).reshape(-1, 5), np.asarray(labels, dtype=np.int64)


def resize_detection(in_size, bbox: np.ndarray, net_w: int, net_h: int) -> np.ndarray:
    diffx = float(in_size[1]) / float(net_w)
    diffy = float(in_size[0]) / float(net_h)
    out = bbox
    for i, box in enumerate(bbox):
        out[i, 0] = box[0] * diffx
        out[i, 1] = box[1] * diffy
        out[i, 2] = box[2] * diffx
        out[i, 3] = box[3] * diffy
    return out


# ---------------------------------------------------------------------------
# A6: pre-processing
# ---------------------------------------------------------------------------

def demean_rgb_image(im: np.ndarray, dtype=np.float64) -> np.ndarray:
    im = im.astype(dtype)
    im[:, :, 0] -= dtype(104.0069879317889)
    im[:, :, 1] -= dtype(116.66876761696767)
    im[:, :, 2] -= dtype(122.6789143406786)
    return (im - im.min()) / (im.max() - im.min())


def resize_bilinear_cv(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv.resize(img, (W, H)) with INTER_LINEAR on a float64 HxWxC image (float coefficients, double accumulate)."""
    h, w = img.shape[:2]
    sx_scale, sy_scale = w / float(W), h / float(H)

    def coords(n_out, n_in, scale):
        idx = np.zeros(n_out, np.int64)
        frac = np.zeros(n_out, np.float32)
        for d in range(n_out):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = np.float32(f - np.float32(s))
            if s < 0:
                f, s = np.float32(0), 0
            if s >= n_in - 1:
                f, s = np.float32(0), n_in - 1
            idx[d], frac[d] = s, f
        return idx, frac

    xi, xf = coords(W, w, sx_scale)
    yi, yf = coords(H, h, sy_scale)
    xi1 = np.minimum(xi + 1, w - 1)
    yi1 = np.minimum(yi + 1, h - 1)
    a0 = (np.float32(1) - xf).astype(np.float64)[None, :, None]
    a1 = xf.astype(np.float64)[None, :, None]
    b0 = (np.float32(1) - yf).astype(np.float64)[:, None, None]
    b1 = yf.astype(np.float64)[:, None, None]
    src = img.astype(np.float64)
    r0 = src[yi][:, xi] * a0 + src[yi][:, xi1] * a1
    r1 = src[yi1][:, xi] * a0 + src[yi1][:, xi1] * a1
    return r0 * b0 + r1 * b1


def preprocess_frame(frame_bgr: np.ndarray, W: int, H: int) -> np.ndarray:
    """run_detector :79-82 -> (3,H,W) float32 blob contents."""
    im = demean_rgb_image(frame_bgr, np.float64)
    im = resize_bilinear_cv(im, W, H)
    return im.transpose(2, 0, 1).astype(np.float32)


def detection_window_roi(image: np.ndarray, net_size: Tuple[int, int], stride: int = 2):
    """scripts/fcn_object_detector.py:257-277 - stride x stride windows of the (already demeaned) image in raster order plus a
    central one of the same size, each cv.resize'd to net_size = (W, H) and transposed to CHW; rects are (x, y, w, h).  The
    reference is Python 2: `/` on ints floors (w/2, h/2 of the central crop)."""
    im_y, im_x = image.shape[:2]
    w, h = int(im_x // stride), int(im_y // stride)
    im_rois, rects = [], []
    for j in range(stride):
        for i in range(stride):
            roi = image[j * h:j * h + h, i * w:i * w + w]
            im_rois.append(resize_bilinear_cv(roi, net_size[0], net_size[1]).transpose(2, 0, 1))
            rects.append(np.array([i * w, j * h, w, h]))
    cx, cy = int(im_x // 2) - w // 2, int(im_y // 2) - h // 2
    roi = image[cy:cy + h, cx:cx + w]
    im_rois.append(resize_bilinear_cv(roi, net_size[0], net_size[1]).transpose(2, 0, 1))
    rects.append(np.array([cx, cy, w, h]))
    return im_rois, rects


def run_detector2_inputs(frame_bgr: np.ndarray, W: int, H: int, stride: int = 2):
    """run_detector2 :198-211 up to net.forward(): the whole frame is demeaned and normalised FIRST (its min / max, not a
    window's), then cut into windows; blob.data[...] = in_datum rounds float64 to float32.  -> ((n, 3, H, W) float32, rects)."""
    im = demean_rgb_image(frame_bgr, np.float64)
    im_rois, rects = detection_window_roi(im, (W, H), stride)
    return np.stack(im_rois).astype(np.float32), rects


def window_boxes_to_frame(rect, boxes: np.ndarray, net_w: int, net_h: int) -> np.ndarray:
    """Detections of one window (net coordinates, integer rows x1 y1 x2 y2 score) in frame coordinates: resize_detection
    (:396-405) with the window's size as the input size, then the window's origin added, as run_detector2 moves its boxes by
    rect[0], rect[1] (:232-233; its 10-pixel padding belongs to the mask path and is not applied)."""
    out = np.asarray(boxes, dtype=np.int64).reshape(-1, 5).copy()
    if len(out):
        out = resize_detection((int(rect[3]), int(rect[2])), out, net_w, net_h)
        out[:, 0] += int(rect[0]); out[:, 2] += int(rect[0])
        out[:, 1] += int(rect[1]); out[:, 3] += int(rect[1])
    return out


# ---------------------------------------------------------------------------
# A4: target generation
# ---------------------------------------------------------------------------

def jaccard_iou(a, b):
    """JaccardCoeff.iou (argumentation_engine.py:26-55), a = cell box (floats), b = rect (ints)."""
    x = max(a[0], b[0])
    y = max(a[1], b[1])
    w = min(a[0] + a[2], b[0] + b[2]) - x
    h = min(a[1] + a[3], b[1] + b[3]) - y
    if w < 0 or h < 0:
        return 0
    ux = min(a[0], b[0])
    uy = min(a[1], b[1])
    uw = max(a[0] + a[2], b[0] + b[2]) - ux
    uh = max(a[1] + a[3], b[1] + b[3]) - uy
    with np.errstate(divide="ignore", invalid="ignore"):
        aub = np.float32(uw * uh)
        anb = np.float32(w * h)
        area_ratio = np.float32(a[2] * a[3]) / np.float32(b[2] * b[3])
        score = anb / aub
        score = score / area_ratio
    return score


def grid_region(im_h: int, im_w: int, stride: int) -> np.ndarray:
    gy, gx = im_h // stride, im_w // stride
    boxes = np.zeros((gy, gx, 4))
    for j in range(gy):
        for i in range(gx):
            boxes[j][i] = (i * stride, j * stride, stride, stride)
    return boxes


def bounding_box_parameterized_labels(im_h: int, im_w: int, rects, labels, stride: int, num_classes: int,
                                      iou_thresh: float = 0.1):
    """argumentation_engine.py:69-109 -> (foreground, boxes, size, obj, coverage) float64 arrays."""
    boxes = grid_region(im_h, im_w, stride)
    gy, gx = boxes.shape[:2]
    ch = 4 * num_classes
    fg = np.zeros((num_classes, gy, gx))
    bl = np.zeros((ch, gy, gx))
    sl = np.zeros((ch, gy, gx))
    ol = np.zeros((ch, gy, gx))
    cl = np.zeros((ch, gy, gx))
    for rect, label in zip(rects, labels):
        k = int(label * 4)
        for j in range(gy):
            for i in range(gx):
                t = boxes[j, i]
                if jaccard_iou(t, rect) > iou_thresh:
                    bl[k + 0, j, i] = rect[0] - t[0]
                    bl[k + 1, j, i] = rect[1] - t[1]
                    bl[k + 2, j, i] = (rect[0] + rect[2]) - t[0]
                    bl[k + 3, j, i] = (rect[1] + rect[3]) - t[1]
                    sl[k + 0, j, i] = 1.0 / rect[2]
                    sl[k + 1, j, i] = 1.0 / rect[3]
                    sl[k + 2, j, i] = 1.0 / rect[2]
                    sl[k + 3, j, i] = 1.0 / rect[3]
                    ol[k:k + 4, j, i] = np.float32(t[2] * t[3]) / np.float32(rect[2] * rect[3])
                    cl[k:k + 4, j, i] = 1.0
                    fg[int(label), j, i] = 1.0
    return fg, bl, sl, ol, cl


def resize_rects(src_hw, dst_wh, rects):
    """Rect part of resize_image_and_labels (argumentation_engine.py:114-138): f32 math, int truncation."""
    out = []
    ratio_x = np.float32(src_hw[1]) / np.float32(dst_wh[0])
    ratio_y = np.float32(src_hw[0]) / np.float32(dst_wh[1])
    for rect in rects:
        x, y, w, h = (np.float32(v) for v in rect)
        xt, yt = x / ratio_x, y / ratio_y
        xb, yb = (x + w) / ratio_x, (y + h) / ratio_y
        out.append((int(xt), int(yt), int(xb - xt), int(yb - yt)))
    return out


def flip_rects(im_hw, rects, flip_flag: int):
    """Rect part of flip_image (argumentation_engine.py:241-267)."""
    H, W = im_hw
    out = []
    for rect in rects:
        pt1 = (rect[0], rect[1])
        pt2 = (rect[0] + rect[2], rect[1] + rect[3])
        if flip_flag == -1:
            pt1 = (W - pt1[0] - 1, H - pt1[1] - 1)
            pt2 = (W - pt2[0] - 1, H - pt2[1] - 1)
        elif flip_flag == 0:
            pt1 = (pt1[0], H - pt1[1] - 1)
            pt2 = (pt2[0], H - pt2[1] - 1)
        elif flip_flag == 1:
            pt1 = (W - pt1[0] - 1, pt1[1])
            pt2 = (W - pt2[0] - 1, pt2[1])
        x, y = min(pt1[0], pt2[0]), min(pt1[1], pt2[1])
        w, h = abs(pt2[0] - pt1[0]), abs(pt2[1] - pt1[1])
        out.append([max(x, 0), max(y, 0), w, h])
    return out
