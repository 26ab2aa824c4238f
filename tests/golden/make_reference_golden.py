#!/usr/bin/env python3
"""Build-container-only: run the reference's OWN numpy functions and write array-only fixtures (tests/golden/reference_*.npz).

The reference (/root/reference, Python 2, read-only) cannot be imported here: its modules import cv2 / imgaug / rospy /
caffe at the top and use Python-2 syntax elsewhere in the same files.  But the functions of SURVEY.md rows A4 / A5 / A6 / A7 /
A9 need nothing except numpy.  This script reads their SOURCE TEXT from /root/reference at run time, takes exactly the
`def` blocks listed in WANTED (found by name, the cited line ranges are asserted), assembles them into a class of the
reference's name (so that `self.__x` name mangling resolves as in the original) and executes them in memory.  Nothing of
that text is written anywhere: the outputs are numpy arrays (inputs + the values the reference computed).

Python-2 semantics the bodies rely on, supplied by the runner (language semantics, not stand-ins for anything the
bodies call): `xrange` (= range), and the classic `/` on two integers (floor division) - every `/` of the extracted
bodies is evaluated by `_py2div`, which floors when BOTH operands are integers (Python ints, numpy integer scalars or
integer arrays) and divides truly otherwise, as Python 2 / numpy-under-Python-2 do.

Two of the functions touch pixels through cv2 on their first statement (`flip_image`: cv.flip, :242;
`resize_image_and_labels`: cv.resize, :120).  cv2 is absent and no stand-in is written for it: from those two only the
rectangle arithmetic is executed - the `for rect in rects` statement of flip_image (:244-266) verbatim, and statements
:122-137 of resize_image_and_labels verbatim with `img` bound to an empty array of the destination size (all they read
of it is `img.shape`).

Usage (this container only; the GPU box has no /root/reference):   python3 tests/golden/make_reference_golden.py
"""
import ast
import os
import re
import sys
import textwrap

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ENGINE = "scripts/data_argumentation_layer/argumentation_engine.py"
NODE = "scripts/fcn_object_detector.py"
TWIN = "scripts/boundary_adjustment/boundary_refinement.py"

# (file, class name the methods live in, {method: (first line, last line) as cited in SURVEY.md section 8a})
WANTED = [
    (ENGINE, "JaccardCoeff", {"iou": (26, 35), "__intersection": (37, 45), "__union": (47, 52), "__area": (54, 55)}),
    (ENGINE, "ArgumentationEngine", {"__init__": (59, 67), "bounding_box_parameterized_labels": (69, 109), "generate_box_labels": (272, 278),
                                    "grid_region": (283, 292), "demean_rgb_image": (297, 303)}),
    (NODE, "FCNObjectDetector", {"gridbox_to_boxes": (357, 394), "resize_detection": (396, 405), "demean_rgb_image": (407, 413)}),
    (TWIN, "BoundaryRefinement", {"gridbox_to_boxes": (265, 302)}),
]


def _is_int(v):
    if isinstance(v, (bool, int, np.integer)):
        return True
    return isinstance(v, np.ndarray) and np.issubdtype(v.dtype, np.integer)


def _py2div(a, b):
    """Python 2's `/`: floor division for two integers, true division otherwise."""
    if _is_int(a) and _is_int(b):
        return np.floor_divide(a, b) if isinstance(a, np.ndarray) or isinstance(b, np.ndarray) else a // b
    return a / b


class _ClassicDivision(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[node.left, node.right], keywords=[]), node)
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            load = ast.parse(ast.unparse(node.target), mode="eval").body      # the same target, in Load context
            call = ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[load, node.value], keywords=[])
            return ast.copy_location(ast.Assign(targets=[node.target], value=call), node)
        return node


def _method_source(lines, cls, name, cited):
    """Source lines of `def name(` in the body of `class cls`, up to (not including) the next statement at that indentation."""
    c0 = [i for i, l in enumerate(lines) if re.match(r"^class %s\b" % re.escape(cls), l)]
    assert len(c0) == 1, "class %s: %d definitions" % (cls, len(c0))
    c1 = next((i for i in range(c0[0] + 1, len(lines)) if re.match(r"^class \w", lines[i])), len(lines))
    pat = re.compile(r"^    def %s\(" % re.escape(name))
    starts = [i for i in range(c0[0], c1) if pat.match(lines[i])]
    assert len(starts) == 1, "method %s: %d definitions" % (name, len(starts))
    i0 = starts[0]
    i1 = i0 + 1
    while i1 < len(lines) and (lines[i1].strip() == "" or lines[i1].startswith("        ")):
        i1 += 1
    while lines[i1 - 1].strip() == "":
        i1 -= 1
    if (i0 + 1, i1) != cited:
        raise AssertionError("method %s.%s spans lines %d-%d, the table says %d-%d (has the reference changed?)" % (cls, name, i0 + 1, i1, cited[0], cited[1]))
    return lines[i0:i1]


def _build_namespace():
    ns = {"np": np, "xrange": range, "_py2div": _py2div, "__name__": "reference_functions"}
    sources = {}
    for path, cls, methods in WANTED:
        with open(os.path.join(REF, path)) as f:
            lines = f.read().split("\n")
        sources[path] = lines
        body = []
        for m, cited in methods.items():
            body += _method_source(lines, cls, m, cited) + [""]
        text = "class %s(object):\n" % cls + "\n".join(body) + "\n"
        tree = _ClassicDivision().visit(ast.parse(text, filename=path))
        ast.fix_missing_locations(tree)
        scope = dict(ns)
        exec(compile(tree, path, "exec"), scope)
        key = cls if path != TWIN else cls + "16"
        ns[key] = scope[cls]
    # the rectangle arithmetic of the two functions whose first statement needs cv2 (see the module docstring)
    eng = sources[ENGINE]

    def stmts(first, last, indent):
        src = textwrap.dedent("\n".join(eng[first - 1:last]))
        tree = _ClassicDivision().visit(ast.parse(src, filename=ENGINE))
        ast.fix_missing_locations(tree)
        return compile(tree, ENGINE, "exec")

    assert eng[240].startswith("    def flip_image(") and eng[243].strip() == "for rect in rects:" and eng[266].strip() == "return im_flip, flip_rects"
    assert eng[113].startswith("    def resize_image_and_labels(") and eng[119].strip().startswith("img = cv.resize(") and eng[136].strip() == "resize_rects.append(rect_resize)"
    flip_loop = stmts(244, 266, 8)
    resize_body = stmts(122, 137, 16)

    def flip_rects(image_shape, rects, flip_flag):
        scope = {"np": np, "image": np.empty(image_shape, np.uint8), "rects": rects, "flip_flag": flip_flag, "flip_rects": [], "_py2div": _py2div}
        exec(flip_loop, scope)
        return scope["flip_rects"]

    def resize_rects(image_shape, net_wh, rects):
        out = []
        for rect in rects:
            scope = {"np": np, "image": np.empty(image_shape, np.uint8), "img": np.empty((net_wh[1], net_wh[0], 3), np.uint8),
                     "rect": rect, "resize_rects": out, "_py2div": _py2div}
            exec(resize_body, scope)
        return out

    ns["flip_rects"], ns["resize_rects"] = flip_rects, resize_rects
    return ns


def _rects(rng, n, h, w, wmax=None):
    out = []
    for _ in range(n):
        rw, rh = int(rng.integers(8, wmax or w // 2)), int(rng.integers(8, wmax or h // 2))
        out.append((int(rng.integers(0, max(w - rw, 1))), int(rng.integers(0, max(h - rh, 1))), rw, rh))
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("make_reference_golden.py runs in the build container only (no %s here)" % REF)
    import warnings
    warnings.simplefilter("ignore")      # `flip_flag is -1` (SyntaxWarning), division by zero in the w*h = 0 case
    ns = _build_namespace()
    rng = np.random.default_rng(20261004)
    out = {}

    # ---- A4: JaccardCoeff.iou and bounding_box_parameterized_labels -------------------------------------------------
    jc = ns["JaccardCoeff"]()
    cells, rects, scores = [], [], []
    for _ in range(400):
        s = int(rng.choice([8, 16, 32]))
        cell = [float(rng.integers(0, 28) * s), float(rng.integers(0, 28) * s), float(s), float(s)]
        rw, rh = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        rect = (int(cell[0]) + int(rng.integers(-rw - 4, s + 4)), int(cell[1]) + int(rng.integers(-rh - 4, s + 4)), rw, rh)
        if rng.random() < 0.15:      # touching edges: a zero-area intersection TUPLE, not the integer 0
            rect = (int(cell[0]) + s, rect[1], rw, rh)
        cells.append(cell)
        rects.append(rect)
        scores.append(float(jc.iou(np.array(cell), rect)))
    out["iou_cells"], out["iou_rects"], out["iou_scores"] = np.array(cells), np.array(rects, np.int64), np.array(scores, np.float64)

    cases = [
        ("kat1", 448, 448, 16, 1, [(100, 120, 80, 60)], [0]),
        ("kat2", 448, 448, 8, 11, [(40, 64, 120, 200), (300, 310, 64, 48)], [3, 10]),              # KAT-2
        ("demo", 224, 224, 16, 1, [(361, 198, 100, 134)], [0]),                                    # the reference's commented demo (:360-365)
        ("overlap", 96, 128, 16, 2, [(10, 10, 60, 50), (30, 20, 70, 60), (0, 0, 128, 96)], [0, 0, 1]),      # later rects overwrite
        ("edges", 64, 96, 16, 1, [(16, 16, 32, 32), (80, 0, 16, 64), (95, 63, 40, 40)], [0, 0, 0]),        # cell-aligned, border, outside
        ("zero_area", 64, 64, 16, 1, [(8, 8, 0, 20), (20, 30, 12, 9)], [0, 0]),                             # w*h == 0: inf / nan, no guard
    ]
    for i in range(6):
        h, w = int(rng.choice([64, 96, 128, 288])), int(rng.choice([64, 128, 160, 288]))
        s = int(rng.choice([8, 16]))
        c = int(rng.integers(1, 4))
        n = int(rng.integers(1, 5))
        cases.append(("rand%d" % i, h, w, s, c, _rects(rng, n, h, w), [int(v) for v in rng.integers(0, c, n)]))
    out["a4_names"] = np.array([c[0] for c in cases])
    for name, h, w, s, c, rs, ls in cases:
        eng = ns["ArgumentationEngine"](w, h, s, c)
        img = np.zeros((h, w, 3), np.uint8)
        got = eng.bounding_box_parameterized_labels(img, rs, ls)
        out["a4_%s_meta" % name] = np.array([h, w, s, c], np.int64)
        out["a4_%s_rects" % name] = np.array(rs, np.int64).reshape(-1, 4)
        out["a4_%s_labels" % name] = np.array(ls, np.int64)
        for key, arr in zip(("fg", "bbox", "size", "obj", "cvg"), got):
            out["a4_%s_%s" % (name, key)] = np.asarray(arr, np.float64)
        grid = eng.grid_region(img, s)
        out["a4_%s_grid" % name] = np.asarray(grid, np.float64)

    # ---- A5: rect arithmetic of resize_image_and_labels / flip_image, demean_rgb_image (float32) -----------------------
    shapes, nets, rin, rout, cnt = [], [], [], [], []
    for _ in range(40):
        h, w = int(rng.integers(40, 700)), int(rng.integers(40, 900))
        net = (int(rng.choice([224, 288, 448])), int(rng.choice([224, 288, 448])))
        rs = _rects(rng, int(rng.integers(1, 5)), h, w)
        got = ns["resize_rects"]((h, w, 3), net, rs)
        shapes.append((h, w))
        nets.append(net)
        cnt.append(len(rs))
        rin += rs
        rout += [tuple(int(v) for v in r) for r in got]
    out["a5_resize_src_hw"], out["a5_resize_net_wh"], out["a5_resize_counts"] = np.array(shapes, np.int64), np.array(nets, np.int64), np.array(cnt, np.int64)
    out["a5_resize_in"], out["a5_resize_out"] = np.array(rin, np.int64), np.array(rout, np.int64)
    shapes, flags, rin, rout, cnt = [], [], [], [], []
    for k in range(45):
        h, w = int(rng.integers(20, 500)), int(rng.integers(20, 500))
        flag = (-1, 0, 1)[k % 3]
        rs = _rects(rng, int(rng.integers(1, 5)), h, w)
        if k % 5 == 0:
            rs.append((w - 3, h - 2, 10, 9))      # sticks out of the image: mirrored corner goes negative, x / y clamp to 0
        got = ns["flip_rects"]((h, w, 3), rs, flag)
        shapes.append((h, w))
        flags.append(flag)
        cnt.append(len(rs))
        rin += rs
        rout += [tuple(int(v) for v in r) for r in got]
    out["a5_flip_hw"], out["a5_flip_flags"], out["a5_flip_counts"] = np.array(shapes, np.int64), np.array(flags, np.int64), np.array(cnt, np.int64)
    out["a5_flip_in"], out["a5_flip_out"] = np.array(rin, np.int64), np.array(rout, np.int64)
    eng = ns["ArgumentationEngine"](448, 448, 16, 1)
    node = ns["FCNObjectDetector"]()
    for i, (h, w) in enumerate(((7, 9), (16, 12), (33, 31))):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out["demean_%d_in" % i] = im
        got32 = eng.demean_rgb_image(im.copy())
        got64 = node.demean_rgb_image(im.copy())
        assert got32.dtype == np.float32 and got64.dtype == np.float64
        out["a5_demean_%d_f32" % i], out["a6_demean_%d_f64" % i] = got32, got64

    # ---- A7 / A9: gridbox_to_boxes (node: stride 8; boundary_refinement twin: stride 16), resize_detection --------------
    for i, (net_w, net_h, thresh, p_fire) in enumerate(((448, 448, 0.5, 0.1), (448, 448, 0.5, 1.0), (288, 288, 0.3, 0.3), (640, 480, 0.5, 0.05), (64, 32, 0.9, 0.0))):
        for tag, cls, stride in (("s8", "FCNObjectDetector", 8), ("s16", "BoundaryRefinement16", 16)):
            det = ns[cls]()
            setattr(det, "_%s__im_width" % cls.rstrip("16"), net_w)
            setattr(det, "_%s__im_height" % cls.rstrip("16"), net_h)
            gy, gx = net_h // stride, net_w // stride
            cvg = (rng.random((gy, gx)) < p_fire).astype(np.float32) * rng.uniform(thresh, 1.0, (gy, gx)).astype(np.float32) + \
                rng.uniform(0, thresh * 0.999, (gy, gx)).astype(np.float32) * (rng.random((gy, gx)) < 0.5)
            cvg = np.minimum(cvg, 1.0).astype(np.float32)
            if i == 1:
                cvg[...] = 0.75      # every cell fires
            bb = rng.normal(0, 40, (4, gy, gx)).astype(np.float32)
            boxes, cvgs, mask = det.gridbox_to_boxes(cvg if stride == 8 else cvg[None], bb, thresh)
            key = "a7_%d_%s" % (i, tag)
            out[key + "_meta"] = np.array([net_w, net_h, stride], np.int64)
            out[key + "_thresh"] = np.array(thresh, np.float64)
            out[key + "_cvg"], out[key + "_bbox"] = cvg, bb
            out[key + "_boxes"] = np.asarray(boxes, np.float64).reshape(-1, 4)
            out[key + "_cvgs"] = np.asarray(cvgs, np.float64).reshape(-1, 3)
            out[key + "_mask"] = np.asarray(mask, np.bool_)
    det = ns["FCNObjectDetector"]()
    det._FCNObjectDetector__im_width, det._FCNObjectDetector__im_height = 448, 448
    for i, in_size in enumerate(((480, 640), (1080, 1920), (448, 448), (300, 517))):
        bbox = np.asarray(rng.integers(-20, 470, (9, 5)), dtype=int)      # np.asarray(object_boxes, dtype=np.int), fcn_object_detector.py:123
        out["a9_%d_in_size" % i], out["a9_%d_in" % i] = np.array(in_size, np.int64), bbox.copy()
        out["a9_%d_out" % i] = np.asarray(det.resize_detection(in_size, bbox.copy()), np.int64)

    path = os.path.join(HERE, "reference_numpy_rows.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %.1f KB" % (path, len(out), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
