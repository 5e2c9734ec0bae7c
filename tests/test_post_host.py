"""Native host code of the post-processing (csrc/host_db.cpp, host_group.cpp, host_refine.cpp: no
device work, runs without a GPU) against the oracle restatement.  The GPU kernels are checked in
tests/test_gpu_post.py and end-to-end in tests/test_gpu_e2e.py."""
import copy

import numpy as np
import pytest

from conftest import pkg
from oracle import cv_ref as cv
from oracle import postproc_ref as R


def fake_outputs(seed=0, size=512):
    """Plausible network outputs rendered from a synthetic page's ink (no network needed)."""
    p = pkg()
    page = p.synth.text_like_page((size, size), seed, n_blocks=10)
    ink = (page.min(axis=2) < 60)
    mask = cv.dilate((ink * 255).astype(np.uint8), cv.RECT3, 1)
    mask_u8 = (mask.astype(np.float32) / 255 * 0.9 * 255).astype(np.uint8)
    blob = cv.dilate((ink * 255).astype(np.uint8), cv.RECT3, 4)
    prob = (blob / 255.0 * 0.85 + 0.05).astype(np.float32)
    big = cv.dilate((ink * 255).astype(np.uint8), cv.RECT3, 10)
    n, lab, stats = R.connected_components_with_stats(big, 8)
    rng = np.random.RandomState(seed)
    blines = np.array([[x, y, x + w, y + h] for x, y, w, h, a in stats[1:]], np.int32).reshape(-1, 4)
    cls = rng.randint(0, 2, len(blines)).astype(np.int32)
    confs = np.round(rng.uniform(0.5, 1, len(blines)), 3)
    return page, mask_u8, prob, (blines, cls, confs)


def blocks_equal(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert [int(v) for v in x.xyxy] == [int(v) for v in y.xyxy]
        assert np.array_equal(np.asarray(x.lines), np.asarray(y.lines))
        assert (x.language, bool(x.vertical), int(x.angle)) == (y.language, bool(y.vertical), int(y.angle))
        assert float(x.font_size) == pytest.approx(float(y.font_size))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_db_boxes_group_output_match_oracle(seed):
    p = pkg()
    page, mask_u8, prob, blks = fake_outputs(seed)
    H, W = prob.shape
    # --- DB boxes: product host geometry on scipy labels vs the contour-walking oracle
    bitmap = prob > 0.3
    nf, lab_f, st_f = R.connected_components_with_stats(bitmap.astype(np.uint8), 8)
    nb, lab_b, st_b = R.connected_components_with_stats((~bitmap).astype(np.uint8), 4)
    rep = p.postproc.SegRepresenter()
    boxes, scores = rep._page(prob, lab_f, st_f[1:], lab_b, st_b[1:], W, H)
    rboxes, rscores = R.boxes_from_bitmap(prob, bitmap, W, H)
    assert len(boxes) == len(rboxes)
    np.testing.assert_array_equal(boxes, rboxes)
    np.testing.assert_allclose(scores, rscores, rtol=0, atol=1e-6)
    # --- grouping
    lines = boxes[scores > 0.6].astype(np.int32)
    got = p.textblock.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8)
    ref = R.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8)
    blocks_equal(got, ref)


@pytest.mark.parametrize("seed,size,what", [
    (140, 640, "a block of > 16 lines with EQUAL distances: numpy's default argsort is x86-simd-sort there (the product calls it)"),
    (153, 640, "... and the order of the tied lines decides a split: different blocks"),
    (59, 1024, "two hull edges bound rectangles of mathematically equal area (2778): the min-area box must not be picked by rounding noise"),
    (226, 512, "two lines of one text row: |sin(arccos(c))| * len ties mathematically, glibc's acos and numpy's SVML arccos differ in the last bit"),
    (1056, 1024, "... the same tie in a vertical block, cascading into a different split (11 blocks against 12)"),
])
def test_native_host_stages_on_the_findings_of_the_round6_seed_sweep(seed, size, what):
    """The five pages (of 120) on which the first seed sweep of the whole-tail parity test (tests/test_gpu_e2e.py
    `test_tail_seed_sweep_by_hand`) found the product's native host code and the oracle apart -- every one a TIE that the
    reference breaks by an implementation detail of numpy.  DB boxes (product geometry on scipy labels) and grouping against
    the oracle, as in the test above; DESIGN 5."""
    p = pkg()
    page, mask_u8, prob, blks = fake_outputs(seed, size)
    H, W = prob.shape
    bitmap = prob > 0.3
    nf, lab_f, st_f = R.connected_components_with_stats(bitmap.astype(np.uint8), 8)
    nb, lab_b, st_b = R.connected_components_with_stats((~bitmap).astype(np.uint8), 4)
    boxes, scores = p.postproc.SegRepresenter()._page(prob, lab_f, st_f[1:], lab_b, st_b[1:], W, H)
    rboxes, rscores = R.boxes_from_bitmap(prob, bitmap, W, H)
    np.testing.assert_array_equal(boxes, rboxes, err_msg=what)
    np.testing.assert_allclose(scores, rscores, rtol=0, atol=1e-6)
    lines = boxes[scores > 0.6].astype(np.int32)
    blocks_equal(p.textblock.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8),
                 R.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8))


@pytest.mark.parametrize("key", ["map0", "map1"])
def test_db_boxes_short_side_of_exactly_two_pixels(key):
    """`sside < 2` (db_utils.py:146) on contours whose min-area rectangle has a short side of MATHEMATICALLY 2.0: the calipers
    return 2.0 or 1.9999999999999858 by rounding noise (one map each way, found by scripts/gpu_db_stress.py: 2 of 1 800 random
    maps).  Oracle and product compare the float32 the reference sees (cv2.minAreaRect returns a Size2f)."""
    import os
    p = pkg()
    prob = np.load(os.path.join(os.path.dirname(__file__), "golden", "db_short_side_ties.npz"))[key]
    H, W = prob.shape
    bitmap = prob > 0.3
    nf, lab_f, st_f = R.connected_components_with_stats(bitmap.astype(np.uint8), 8)
    nb, lab_b, st_b = R.connected_components_with_stats((~bitmap).astype(np.uint8), 4)
    boxes, scores = p.postproc.SegRepresenter()._page(prob, lab_f, st_f[1:], lab_b, st_b[1:], W, H)
    rboxes, rscores = R.boxes_from_bitmap(prob, bitmap, W, H)
    np.testing.assert_array_equal(boxes, rboxes)
    np.testing.assert_allclose(scores, rscores, rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", ["noise", "holes", "thin", "empty", "full", "cap"])
def test_db_boxes_native_host_geometry_edge_cases(case):
    """`ctd_db_boxes` (csrc/host_db.cpp, host-only: runs without a GPU) against the oracle's
    contour walk + polygon fill on bitmaps that stress it: speckle, nested holes/islands,
    1-px lines (min-side rejection), no foreground, all foreground, more contours than the cap."""
    p = pkg()
    rng = np.random.RandomState(7)
    H, W = 96, 160
    prob = np.full((H, W), 0.05, np.float32)
    cap = 1000
    if case == "noise":
        from scipy import ndimage
        prob = ndimage.uniform_filter(rng.rand(H, W), 3).astype(np.float32)
        prob = (prob - prob.min()) / (prob.max() - prob.min()) * 0.6
    elif case == "holes":
        prob[10:80, 10:120] = 0.9
        prob[20:70, 20:110] = 0.1           # hole
        prob[30:60, 30:100] = 0.8           # island in the hole
        prob[40:50, 40:90] = 0.2            # hole in the island
        prob[43:47, 50:60] = 0.7            # island in that hole
        prob[5:9, 130:150] = 0.95
    elif case == "thin":
        prob[10, 5:100] = 0.9
        prob[20:60, 30] = 0.9
        for i in range(30):
            prob[50 + i, 60 + i] = 0.9      # 8-connected diagonal
        prob[70:73, 100:140] = 0.9
    elif case == "full":
        prob[:] = 0.9
    elif case == "cap":
        prob[::3, ::3] = 0.9                # > 1000 single-pixel contours
        prob[40:60, 40:100] = 0.9
        cap = 50
    bitmap = prob > 0.3
    nf, lab_f, st_f = R.connected_components_with_stats(bitmap.astype(np.uint8), 8)
    nb, lab_b, st_b = R.connected_components_with_stats((~bitmap).astype(np.uint8), 4)
    rep = p.postproc.SegRepresenter(max_candidates=cap)
    boxes, scores = rep._page(prob, lab_f, st_f[1:], lab_b, st_b[1:], W, H)
    rboxes, rscores = R.boxes_from_bitmap(prob, bitmap, W, H, max_candidates=cap)
    np.testing.assert_array_equal(boxes, rboxes)
    np.testing.assert_allclose(scores, rscores, rtol=0, atol=1e-6)
    if case in ("empty", "full"):
        assert (scores > 0).sum() == (1 if case == "full" else 0)
    if case == "holes":
        assert (scores > 0).sum() == 6      # 3 outer borders + 2 hole borders + the separate bar


def test_db_boxes_native_host_geometry_speckle_sweep():
    """Random speckle at several correlation lengths: many nested holes, peninsulas and diagonal
    links -- the cases where 'pixels of the filled contour polygon' is subtle (a hole border's
    polygon excludes foreground that its ring merely surrounds)."""
    from scipy import ndimage
    p = pkg()
    rep = p.postproc.SegRepresenter()
    for seed in range(12):
        rng = np.random.RandomState(100 + seed)
        H, W = 64 + 3 * seed, 100 + 5 * seed
        prob = ndimage.uniform_filter(rng.rand(H, W), 1 + seed % 4).astype(np.float32)
        prob = (prob - prob.min()) / (prob.max() - prob.min()) * (0.55 + 0.01 * (seed % 10))
        bitmap = prob > 0.3
        nf, lab_f, st_f = R.connected_components_with_stats(bitmap.astype(np.uint8), 8)
        nb, lab_b, st_b = R.connected_components_with_stats((~bitmap).astype(np.uint8), 4)
        boxes, scores = rep._page(prob, lab_f, st_f[1:], lab_b, st_b[1:], W, H)
        rboxes, rscores = R.boxes_from_bitmap(prob, bitmap, W, H)
        np.testing.assert_array_equal(boxes, rboxes)
        np.testing.assert_allclose(scores, rscores, rtol=0, atol=1e-6)


def test_group_output_hand_built_scenes():
    p = pkg()
    W = H = 600
    mask = np.full((H, W), 255, np.uint8)

    def quad(x, y, w, h):
        return [[x, y], [x + w, y], [x + w, y + h], [x, y + h]]
    # vertical JA block with 3 columns + a far column that must split off; horizontal EN block;
    # two scattered horizontal lines that merge; one scattered line over empty mask that is dropped
    lines = np.array([quad(300, 50, 20, 200), quad(270, 50, 20, 200), quad(240, 50, 20, 180), quad(120, 60, 20, 150),
                      quad(50, 400, 200, 24), quad(50, 430, 180, 24),
                      quad(400, 500, 100, 20), quad(400, 524, 90, 20), quad(10, 10, 30, 8)], np.int32)
    mask[5:25, 5:50] = 0
    blks = (np.array([[110, 40, 330, 260], [40, 390, 260, 460]], np.int32), np.array([1, 0], np.int32),
            np.array([0.9, 0.8]))
    got = p.textblock.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask)
    ref = R.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask)
    blocks_equal(got, ref)
    assert any(b.vertical for b in got) and any(b.language == "eng" for b in got)
    assert sum(len(b.lines) for b in got) == 8      # the line over empty mask was dropped
    # empty inputs are legal (reference inference.py:166-167)
    assert p.textblock.group_output((np.zeros((0, 4), np.int32), np.zeros(0, np.int32), np.zeros(0)), [], W, H, mask) == []


def test_refine_host_decisions_match_oracle():
    """csrc/host_refine.cpp: the colour pick (np.histogram(bins=255) + get_topk_color), the Otsu threshold
    and the cv2.inRange bounds against the oracle's restatements, on random and degenerate histograms."""
    import ctypes as C
    lib = pkg()._lib.lib()
    rng = np.random.RandomState(3)
    for it in range(300):
        kind = it % 6
        if kind == 0:
            px = rng.randint(0, 256, rng.randint(1, 4000)).astype(np.uint8)
        elif kind == 1:
            px = np.clip(rng.normal(rng.randint(20, 230), rng.uniform(1, 40), rng.randint(1, 5000)), 0, 255).astype(np.uint8)
        elif kind == 2:
            px = np.r_[np.full(rng.randint(1, 900), rng.randint(0, 256)), rng.randint(0, 256, rng.randint(0, 30))].astype(np.uint8)
        elif kind == 3:
            px = np.full(rng.randint(1, 50), rng.randint(0, 256), np.uint8)          # min == max
        elif kind == 4:
            px = np.zeros(0, np.uint8)                                               # nothing selected
        else:
            lo = rng.randint(0, 200)
            px = rng.randint(lo, lo + rng.randint(2, 56), rng.randint(1, 3000)).astype(np.uint8)
        hist = np.bincount(px, minlength=256).astype(np.int64)
        out = np.zeros(3, np.float64)
        n = lib.ctd_topk_colors(hist.ctypes.data, out.ctypes.data)
        counts, edges = np.histogram(px, bins=255)
        ref = R.get_topk_color(edges, counts, k=3, color_var=10)
        assert n == len(ref)
        np.testing.assert_array_equal(out[:n], np.asarray(ref, np.float64))
        if len(px):
            assert lib.ctd_otsu_from_hist(hist.ctypes.data) == cv.otsu_threshold_value(px)
    for lo, hi in [(7.4, 67.4), (7.5, 67.5), (8.5, 68.5), (-12.3, 47.7), (195.0, 255), (-70.2, -10.2), (254.6, 255),
                   (300.0, 360.0), (100.5, 100.4), (0.5, 0.5), (-0.5, 0.49)] + [tuple(rng.uniform(-80, 300, 2)) for _ in range(50)]:
        lb, ub = C.c_int32(), C.c_int32()
        lib.ctd_inrange_bounds(lo, hi, C.byref(lb), C.byref(ub))
        rl, ru = cv.in_range_bounds(lo, hi)
        assert (lb.value > ub.value) == (rl > ru)
        if rl <= ru:
            assert (lb.value, ub.value) == (rl, ru)


def test_oracle_contours_and_cc_against_scipy():
    """Pins what can be pinned of the unpinned half: contour count = 8-connected components +
    enclosed 4-connected background regions; the filled outer contour = component with holes filled."""
    from scipy import ndimage
    rng = np.random.RandomState(1)
    img = (ndimage.uniform_filter(rng.rand(80, 120), 5) > 0.5).astype(np.uint8)
    contours = cv.find_contours(img)
    ncomp = ndimage.label(img, structure=np.ones((3, 3)))[1]
    bg, nbg = ndimage.label(1 - img)
    border = set(np.unique(np.r_[bg[0], bg[-1], bg[:, 0], bg[:, -1]])) - {0}
    assert len(contours) == ncomp + (nbg - len(border))
    lab, n = ndimage.label(img, structure=np.ones((3, 3)))
    sizes = sorted(int(ndimage.binary_fill_holes(lab == l).sum()) for l in range(1, n + 1))
    filled = sorted(int(cv.fill_poly(img.shape, c).sum()) for c in contours)
    for s in sizes:
        assert s in filled


def _db_single_component(pts_mask, prob_val=0.9):
    """Runs `ctd_db_boxes` on a bitmap given as a boolean array; returns (boxes, scores)."""
    p = pkg()
    H, W = pts_mask.shape
    prob = np.where(pts_mask, prob_val, 0.05).astype(np.float32)
    nf, lab_f, st_f = R.connected_components_with_stats(pts_mask.astype(np.uint8), 8)
    nb, lab_b, st_b = R.connected_components_with_stats((~pts_mask).astype(np.uint8), 4)
    return p.postproc.SegRepresenter()._page(prob, lab_f, st_f[1:], lab_b, st_b[1:], W, H)


def test_db_boxes_geometric_invariants_on_random_blobs():
    """Properties that hold whatever OpenCV's exact rounding is: the unclipped box of a solid
    convex blob contains the blob, its area exceeds the blob's, it stays inside the map, and the
    transposed blob gives a box of (nearly) the same area."""
    rng = np.random.RandomState(5)
    for _ in range(25):
        H, W = 80, 96
        m = np.zeros((H, W), bool)
        cy, cx = rng.randint(20, H - 20), rng.randint(20, W - 20)
        ry, rx = rng.randint(3, 12), rng.randint(3, 14)
        yy, xx = np.mgrid[:H, :W]
        ang = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
        v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
        m[(u / rx) ** 2 + (v / ry) ** 2 <= 1] = True
        boxes, scores = _db_single_component(m)
        assert len(boxes) == 1 and scores[0] == pytest.approx(0.9, abs=1e-6)
        box = boxes[0].astype(np.float64)
        assert (box >= 0).all() and (box[:, 0] <= W).all() and (box[:, 1] <= H).all()
        # every blob pixel lies inside the quad (cross products of a consistently oriented quad)
        ys, xs = np.nonzero(m)
        pts = np.stack([xs, ys], 1).astype(np.float64)
        e = np.roll(box, -1, axis=0) - box
        cr = e[:, None, 0] * (pts[None, :, 1] - box[:, None, 1]) - e[:, None, 1] * (pts[None, :, 0] - box[:, None, 0])
        clipped = (box == 0).any() or (box[:, 0] == W).any() or (box[:, 1] == H).any()
        if not clipped:
            assert (cr >= -1e-9).all() or (cr <= 1e-9).all()
        area = 0.5 * abs(np.dot(box[:, 0], np.roll(box[:, 1], -1)) - np.dot(box[:, 1], np.roll(box[:, 0], -1)))
        assert area >= m.sum() or clipped
        # the transposed bitmap gives a box of the same area up to rounding (the rectangle itself may
        # differ: equal-area candidates are tie-broken in hull order, as in OpenCV)
        tb, _ = _db_single_component(np.ascontiguousarray(m.T))
        t = tb[0].astype(np.float64)
        tarea = 0.5 * abs(np.dot(t[:, 0], np.roll(t[:, 1], -1)) - np.dot(t[:, 1], np.roll(t[:, 0], -1)))
        if not clipped:
            assert abs(tarea - area) <= 0.15 * area + 8


def test_oracle_resize_geometry_against_torch_bilinear():
    """Independent check of `cv_ref.resize_linear_u8`'s sampling geometry (half-pixel centres, edge clamp, up- and
    down-scaling without anti-aliasing): torch's float bilinear `interpolate(align_corners=False)` implements the
    same `src = (dst + 0.5) * scale - 0.5` convention in floating point, so the fixed-point result must agree with
    it to within one grey level (the 11-bit coefficient rounding), on every pixel."""
    import torch
    import torch.nn.functional as F
    from oracle import cv_ref
    rng = np.random.RandomState(7)
    for (sh, sw), (dh, dw) in [((37, 53), (74, 106)), ((64, 48), (1024, 768)), ((413, 292), (1024, 724)),
                               ((300, 200), (77, 51)), ((96, 128), (95, 127)), ((50, 50), (50, 131))]:
        img = rng.randint(0, 256, (sh, sw, 3)).astype(np.uint8)
        got = cv_ref.resize_linear_u8(img, (dw, dh)).astype(np.int64)
        t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
        want = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0)
        diff = np.abs(got - np.rint(want.numpy()).astype(np.int64))
        assert diff.max() <= 1, ((sh, sw), (dh, dw), int(diff.max()))
        assert (diff > 0).mean() < 0.2


def test_oracle_round_offset_is_the_minkowski_sum_with_a_disc():
    """Independent check of `cv_ref.clipper_offset_round` (the restated Clipper 6.4.2 JT_ROUND offset behind
    `unclip`, reference db_utils.py:168-174): geometry, not Clipper's code.  For a convex polygon P and distance d
    the exact result is the Minkowski sum P + disc(d): (1) every vertex of the ring lies at distance d from P, up to
    the arc tolerance (chords sag inwards by <= 0.25) and the integer rounding (<= 0.5 * sqrt(2)); (2) its area is
    Steiner's A + L*d + pi*d^2 up to the same two effects along its perimeter; (3) it contains P."""
    from oracle import cv_ref
    rng = np.random.RandomState(11)

    def seg_dist(p, a, b):
        ab, ap = b - a, p - a
        t = np.clip(np.dot(ap, ab) / max(np.dot(ab, ab), 1e-12), 0.0, 1.0)
        return float(np.linalg.norm(ap - t * ab))

    for _ in range(60):
        cx, cy = rng.uniform(100, 900, 2)
        w, h = rng.uniform(6, 300), rng.uniform(4, 120)
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        quad = (np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) / 2) @ R.T + [cx, cy]
        ipts = np.trunc(quad).astype(np.int64)                       # pyclipper's cast
        P = ipts.astype(np.float64)
        A = abs(cv_ref.polygon_area(P.astype(np.float32)))
        Lp = sum(np.linalg.norm(P[i] - P[(i + 1) % 4]) for i in range(4))
        if A < 4:
            continue
        d = float(A * 1.5 / Lp)
        ring = cv_ref.clipper_offset_round(ipts, d).astype(np.float64)
        assert len(ring) >= 8
        slack = 0.25 + 0.5 * 2 ** 0.5 + 1e-6
        for v in ring:
            dist = min(seg_dist(v, P[i], P[(i + 1) % 4]) for i in range(4))
            assert d - slack <= dist <= d + 0.5 * 2 ** 0.5 + 1e-6, (d, dist)
        x, y = ring[:, 0], ring[:, 1]
        area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
        steiner = A + Lp * d + np.pi * d * d
        per = sum(np.linalg.norm(ring[i] - ring[(i + 1) % len(ring)]) for i in range(len(ring)))
        assert abs(area - steiner) <= per * slack, (area, steiner)
        # contains the polygon: every corner of P is on the inner side of every ring edge (ring is convex)
        e, q = np.roll(ring, -1, 0) - ring, P[:, None, :] - ring[None, :, :]
        s = np.sign(e[None, :, 0] * q[:, :, 1] - e[None, :, 1] * q[:, :, 0])
        assert (s >= 0).all() or (s <= 0).all()


def test_pages_lying_back_to_back_become_a_batch_without_a_copy():
    """`TextDetector._as_batch`: slices of one allocation (the pages `_stage` uploads) are viewed, anything else is stacked."""
    import importlib
    import torch
    det = importlib.import_module("comic-text-detector_amd.detector")
    buf = torch.arange(5 * 4 * 6 * 3, dtype=torch.uint8)            # wraps at 256, content irrelevant
    pages = [buf[i * 72:(i + 1) * 72].view(4, 6, 3) for i in range(1, 5)]      # a run that does not start at the allocation
    x = det.TextDetector._as_batch(pages)
    assert x.shape == (4, 4, 6, 3) and x.data_ptr() == pages[0].data_ptr()      # a view: no copy
    assert all(torch.equal(x[i], p) for i, p in enumerate(pages))
    shuffled = [pages[1], pages[0], pages[2]]
    y = det.TextDetector._as_batch(shuffled)
    assert y.data_ptr() != shuffled[0].data_ptr() and all(torch.equal(y[i], p) for i, p in enumerate(shuffled))
    separate = [p.clone() for p in pages]
    z = det.TextDetector._as_batch(separate)
    assert z.data_ptr() != separate[0].data_ptr() and torch.equal(z, x)
    gap = [pages[0], pages[2]]                                         # same allocation, not adjacent
    assert torch.equal(det.TextDetector._as_batch(gap), torch.stack(gap))
