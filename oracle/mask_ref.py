"""ORACLE — TEST INFRASTRUCTURE ONLY.  Parity unpinned: OpenCV is not in /root/reference and not installable here.

CPU restatement of what the node does with its score maps AFTER net.forward() in run_detector2
(reference: scripts/fcn_object_detector.py:208-236) and of create_mask_labels (:279-303):

    feature_maps[feature_maps < prob_thresh] = 0
    for every window (fmaps, rect) and every class index 1 .. C-1:
        feat = (fmaps[index] * 255)                        float32
        feat = cv.resize(feat, (rect[2], rect[3]))         INTER_LINEAR on CV_32F
        feat = feat.astype(np.uint8)
        pmap[y:y+h, x:x+w] |= feat
        r = create_mask_labels(feat)                       bounding rect of the contour with the largest cv.contourArea
        r += (rect[0] - padding, rect[1] - padding, 2 * padding, 2 * padding)

The OpenCV pieces are restated from the library's published algorithms (OpenCV 3.x, imgproc):

  * ``cv::resize`` INTER_LINEAR, CV_32F, one channel (resize.cpp: ``resizeGeneric_`` with HResizeLinear / VResizeLinear over
    float): source coordinate fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx, clamped to the image with fx = 0 at
    either end; weights (1 - fx, fx) as floats; a horizontal pass S[sx] * a0 + S[sx + 1] * a1 and a vertical pass R0 * b0 + R1 * b1,
    every product and sum rounded to float32 (the SSE path multiplies and adds separately: no fused multiply-add).
  * ``cv::findContours(RETR_CCOMP, CHAIN_APPROX_SIMPLE)`` (contours.cpp: Suzuki & Abe 1985 border following): outer borders are
    followed with 8-connectivity starting at a pixel whose left neighbour is 0, scanning the image in raster order; hole borders
    are followed too.  The chain approximation drops collinear points only, so neither the polygon's area nor its bounding
    rectangle changes.  The contours come back in REVERSE order of discovery (cvFindContours links a new contour in front of
    its siblings), the holes of a component after its outer border.
  * ``cv::contourArea`` = |sum(x_i * y_{i+1} - x_{i+1} * y_i)| / 2 over the border's points in double; ``cv::boundingRect`` =
    (min x, min y, max x - min x + 1, max y - min y + 1).

What the selection ``if max_area < a`` (max_area starting at 0) then comes to: the bounding box of the 8-connected component whose
OUTER border polygon has the largest area, provided that area is positive (a single pixel, a one-pixel-wide line: area 0, never
selected); among components of equal area the one discovered LAST in raster order (first in OpenCV's list).  A hole border never
wins: its polygon lies inside its own component's outer polygon (its area is at most that one's, equal only for a one-pixel-wide
ring, where both have the same bounding box, and the outer border precedes its holes in the list).  `find_contours_outer`
below follows the outer borders only, with Suzuki's rules, and the tests check the hole argument on rings.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32
# direction codes of cvFindContours (x right, y DOWN): 0 = E, counter-clockwise on the screen in steps of 45 degrees
DELTAS = ((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))


def resize_linear_f32(src: np.ndarray, W: int, H: int) -> np.ndarray:
    """cv.resize(src, (W, H)) with INTER_LINEAR on one float32 channel (see the module header)."""
    src = np.ascontiguousarray(src, F32)
    h, w = src.shape

    def coords(n_out: int, n_in: int):
        scale = 1.0 / (float(n_out) / float(n_in))      # (resize.cpp: inv_scale = dsize / ssize in double, scale = 1 / inv_scale)
        idx = np.zeros(n_out, np.int64)
        frac = np.zeros(n_out, F32)
        for d in range(n_out):
            f = F32((d + 0.5) * scale - 0.5)
            s = int(math.floor(float(f)))
            f = F32(f - F32(s))
            if s < 0:
                f, s = F32(0), 0
            if s >= n_in - 1:
                f, s = F32(0), n_in - 1
            idx[d], frac[d] = s, f
        return idx, frac

    xi, xf = coords(W, w)
    yi, yf = coords(H, h)
    xi1, yi1 = np.minimum(xi + 1, w - 1), np.minimum(yi + 1, h - 1)
    a0, a1 = (F32(1) - xf)[None, :], xf[None, :]
    b0, b1 = (F32(1) - yf)[:, None], yf[:, None]
    # horizontal pass on the two source rows of every output row, then the vertical pass: float32 products, float32 sums
    r0 = (src[yi][:, xi] * a0).astype(F32) + (src[yi][:, xi1] * a1).astype(F32)
    r1 = (src[yi1][:, xi] * a0).astype(F32) + (src[yi1][:, xi1] * a1).astype(F32)
    return ((r0.astype(F32) * b0).astype(F32) + (r1.astype(F32) * b1).astype(F32)).astype(F32)


def to_uint8(a: np.ndarray) -> np.ndarray:
    """ndarray.astype(np.uint8) of float32 as the C cast numpy performs on x86-64: through a 32-bit integer (truncation towards
    zero), then the low byte.  In range - the node's maps are scores in [0, 1] times 255 - this is plain truncation."""
    return (np.trunc(np.asarray(a, np.float64)).astype(np.int64) & 0xFF).astype(np.uint8)


def follow_outer_border(mask: np.ndarray, y0: int, x0: int) -> List[Tuple[int, int]]:
    """Suzuki border following of the OUTER border that starts at (x0, y0) - a nonzero pixel whose left neighbour is zero and
    that is the first pixel of its component in raster order (cvFindContours / icvFetchContour with is_hole = 0).  Returns the
    border's points (x, y) in following order, every visit listed (no chain approximation)."""
    h, w = mask.shape

    def at(x: int, y: int) -> bool:
        return 0 <= x < w and 0 <= y < h and mask[y, x] != 0

    s_end = s = 4                                   # start looking at the pixel we came from: the left neighbour (code 4 = W)
    while True:                                     # clockwise (decreasing code) for the first nonzero neighbour
        s = (s - 1) & 7
        x1, y1 = x0 + DELTAS[s][0], y0 + DELTAS[s][1]
        if at(x1, y1) or s == s_end:
            break
    if s == s_end:
        return [(x0, y0)]                           # an isolated pixel
    pts = []
    x3, y3 = x0, y0
    while True:
        s_end = s
        while True:                                 # counter-clockwise from the neighbour after the one we came from
            s = (s + 1) & 7
            x4, y4 = x3 + DELTAS[s][0], y3 + DELTAS[s][1]
            if at(x4, y4):
                break
        pts.append((x3, y3))
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return pts


def contour_area2(pts: Sequence[Tuple[int, int]]) -> int:
    """Twice cv.contourArea of a closed polygon: |sum(x_i * y_{i+1} - x_{i+1} * y_i)| (an integer for integer points)."""
    a = 0
    n = len(pts)
    for i in range(n):
        x0, y0 = pts[i - 1]
        x1, y1 = pts[i]
        a += x0 * y1 - x1 * y0
    return abs(a)


def component_starts(mask: np.ndarray) -> List[Tuple[int, int]]:
    """(y, x) of the first pixel in raster order of every 8-connected component, in raster order (= discovery order of the
    outer borders in cvFindContours)."""
    h, w = mask.shape
    seen = np.zeros((h, w), bool)
    out = []
    fg = mask != 0
    for y in range(h):
        xs = np.nonzero(fg[y] & ~seen[y])[0]
        for x in xs:
            if seen[y, x]:
                continue
            out.append((y, int(x)))
            stack = [(y, int(x))]
            seen[y, x] = True
            while stack:
                cy, cx = stack.pop()
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        ny, nx = cy + dy, cx + dx
                        if 0 <= ny < h and 0 <= nx < w and fg[ny, nx] and not seen[ny, nx]:
                            seen[ny, nx] = True
                            stack.append((ny, nx))
    return out


def create_mask_labels(im_mask: np.ndarray) -> Optional[Tuple[int, int, int, int]]:
    """scripts/fcn_object_detector.py:279-303: bounding rect (x, y, w, h) of the contour with the largest area, None without one."""
    mask = np.asarray(im_mask) > 0
    best_area, best = 0, None
    for (y, x) in component_starts(mask):                       # raster order; the LAST of equal areas wins (OpenCV lists it first)
        pts = follow_outer_border(mask, y, x)
        a2 = contour_area2(pts)
        if a2 > 0 and a2 >= best_area:
            xs, ys = [p[0] for p in pts], [p[1] for p in pts]
            best_area, best = a2, (min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1)
    return best


def run_detector2_post(feature_maps: np.ndarray, rects: Sequence[Sequence[int]], frame_hw: Tuple[int, int], prob_thresh: float,
                       padding: int = 10):
    """scripts/fcn_object_detector.py:208-236 -> (pmap (h, w) uint8, [(np.array([x, y, w, h]), class index), ...])."""
    fm = np.array(feature_maps, F32, copy=True)
    fm[fm < F32(prob_thresh)] = 0
    pmap = np.zeros(frame_hw, np.uint8)
    bboxs = []
    for fmaps, rect in zip(fm, rects):
        x, y, w, h = (int(v) for v in rect)
        for index in range(1, fmaps.shape[0]):
            feat = (fmaps[index] * F32(255)).astype(F32)
            feat = to_uint8(resize_linear_f32(feat, w, h))
            pmap[y:y + h, x:x + w] |= feat[0:h, 0:w]
            r = create_mask_labels(feat)
            if r is not None:
                r = np.array(r)
                r[0] += x - padding
                r[1] += y - padding
                r[2] += 2 * padding
                r[3] += 2 * padding
                bboxs.append((r, index))
    return pmap, bboxs
