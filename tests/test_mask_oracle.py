"""The oracle of run_detector2's post-forward leg (oracle/mask_ref.py) on hand-checkable masks, and against independent
witnesses: scipy.ndimage's labelling / bounding boxes, and Pick's theorem for the area of the followed border
(a lattice polygon's area = enclosed lattice points - boundary steps / 2 - 1)."""
import numpy as np

from oracle import mask_ref as M


def mk(rows):
    return np.array([[255 if ch == "#" else 0 for ch in r] for r in rows], np.uint8)


def test_hand_checkable_masks():
    # 2 x 2 block: polygon through four pixel centres, area 1; bounding rect x, y, w, h
    assert M.create_mask_labels(mk([".....", "..##.", "..##.", "....."])) == (2, 1, 2, 2)
    # a single pixel and a one-pixel-wide line have contours of area 0: `max_area < a` never holds -> None (fcn_object_detector.py:295-301)
    assert M.create_mask_labels(mk([".....", "..#..", "....."])) is None
    assert M.create_mask_labels(mk([".....", ".###.", "....."])) is None
    assert M.create_mask_labels(mk(["....", "...."])) is None
    # 8-connectivity: a diagonal chain hangs the block at the bottom right onto the pixel at the top left
    assert M.create_mask_labels(mk(["#....", ".#...", "..#..", "...##", "...##"])) == (0, 0, 5, 5)
    # a ring: the outer border wins over (ties with) its hole border; same bounding box either way
    ring = mk([".......", ".#####.", ".#...#.", ".#...#.", ".#####.", "......."])
    assert M.create_mask_labels(ring) == (1, 1, 5, 4)
    assert M.contour_area2(M.follow_outer_border(ring > 0, 1, 1)) == 2 * 4 * 3
    # two components of equal area: the one found LAST in raster order is first in OpenCV's list and wins the strict '<'
    two = mk(["##....", "##....", "......", "...##.", "...##."])
    assert M.create_mask_labels(two) == (3, 3, 2, 2)
    # a larger component beats a later, smaller one
    assert M.create_mask_labels(mk(["###...", "###...", "###...", ".....#", "....##"])) == (0, 0, 3, 3)
    # a component nested in another one's hole is a component of its own
    nest = mk(["#######", "#.....#", "#.###.#", "#.###.#", "#.....#", "#######"])
    assert M.component_starts(nest > 0) == [(0, 0), (2, 2)] and M.create_mask_labels(nest) == (0, 0, 7, 6)


def test_border_following_against_scipy_and_pick():
    from scipy import ndimage as ndi
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(120):
        a = (ndi.gaussian_filter(rng.random((24, 31)), 1.2) > 0.5).astype(np.uint8)
        lab, n = ndi.label(a, structure=np.ones((3, 3)))
        starts = M.component_starts(a > 0)
        assert len(starts) == n
        for (y, x) in starts:
            pts = M.follow_outer_border(a > 0, y, x)
            sl = ndi.find_objects((lab == lab[y, x]).astype(int))[0]
            xs, ys = [p[0] for p in pts], [p[1] for p in pts]
            assert (min(ys), max(ys) + 1, min(xs), max(xs) + 1) == (sl[0].start, sl[0].stop, sl[1].start, sl[1].stop)
            filled = int(ndi.binary_fill_holes(lab == lab[y, x]).sum())
            steps = len(pts) if len(pts) > 1 else 0
            assert M.contour_area2(pts) == 2 * filled - steps - 2
            checked += 1
    assert checked > 200


def test_resize_and_cast():
    src = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(M.resize_linear_f32(src, 4, 3), src)                        # identity
    up = M.resize_linear_f32(src, 8, 6)
    assert up.shape == (6, 8) and up[0, 0] == 0 and up[-1, -1] == 11 and np.all(np.diff(up, axis=1) >= 0)
    assert up[0, 1] == np.float32(0.25) and up[0, 2] == np.float32(0.75)               # fx = (dx + 0.5) / 2 - 0.5
    assert M.to_uint8(np.array([0.0, 0.99, 1.0, 254.999, 255.0, 256.0, -1.5], np.float32)).tolist() == [0, 0, 1, 254, 255, 0, 255]


def test_run_detector2_post_small():
    fm = np.zeros((2, 3, 4, 4), np.float32)
    fm[0, 1, 1:3, 1:3] = 0.9           # window 0, class 1: a 2 x 2 blob -> 4 x 4 after the resize to 8 x 8
    fm[1, 2, :, :] = 0.4               # below the threshold: nothing
    rects = [(0, 0, 8, 8), (8, 0, 8, 8)]
    pmap, bboxs = M.run_detector2_post(fm, rects, (8, 16), 0.5)
    assert pmap[:, 8:].max() == 0 and pmap.max() == int(np.float32(0.9) * np.float32(255))
    assert len(bboxs) == 1 and bboxs[0][1] == 1
    x, y, w, h = bboxs[0][0]
    assert (w, h) == (pmap[:, :8].any(0).sum() + 20, pmap[:, :8].any(1).sum() + 20) and x == np.argmax(pmap.any(0)) - 10
