"""Build-container only (skipped where /root/reference is absent): the reference's OWN
post-processing code, run with functional stand-ins for cv2 / shapely / pyclipper / torchvision
(oracle/ref_post_import.py), against the oracle restatement (oracle/postproc_ref.py).  Both sides
share the third-party primitives, so this pins the restatement of the reference's control flow;
the primitives themselves stay unpinned."""
import copy

import numpy as np
import pytest
import torch

from oracle import postproc_ref as R
from oracle import ref_import as RI
from test_gpu_post import random_blks
from test_post_host import fake_outputs

pytestmark = pytest.mark.skipif(not RI.reference_available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    from oracle import ref_post_import as RP
    return RP.load_reference_post()


def same_blocks(ours, theirs):
    assert len(ours) == len(theirs)
    for a, b in zip(ours, theirs):
        assert [int(v) for v in a.xyxy] == [int(v) for v in b.xyxy]
        assert np.array_equal(np.asarray(a.lines), np.asarray(b.lines))
        assert (a.language, bool(a.vertical), int(a.angle), bool(a.merged)) == \
               (b.language, bool(b.vertical), int(b.angle), bool(b.merged))
        assert float(a.font_size) == float(b.font_size) and type(a.font_size) is type(b.font_size)
        for k in ("distance", "vec"):
            u, v = getattr(a, k), getattr(b, k)
            assert (u is None) == (v is None)
            if u is not None:
                np.testing.assert_array_equal(np.asarray(u), np.asarray(v))
        assert float(a.norm) == float(b.norm) and float(a.weight) == float(b.weight)


@pytest.mark.parametrize("rows,frac", [(1008, 0.05), (4032, 0.3)])
def test_nms_restatement_equals_reference_code(ref, rows, frac):
    rng = np.random.RandomState(rows)
    blks = random_blks(rng, 2, rows, frac=frac)
    theirs = ref.YU.non_max_suppression(torch.from_numpy(blks), 0.4, 0.35)
    ours = R.non_max_suppression(blks, 0.4, 0.35)
    for a, b in zip(ours, theirs):
        np.testing.assert_array_equal(a, b.numpy())


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_db_boxes_group_output_refine_equal_reference_code(ref, seed):
    page, mask_u8, prob, blks = fake_outputs(seed, 512)
    H, W = prob.shape
    pred = np.stack([prob, np.zeros_like(prob)])[None]
    # SegDetectorRepresenter.__call__ (db_utils.py:40-69)
    tb, ts = ref.DB.SegDetectorRepresenter(thresh=0.3)(None, pred.copy())
    ob, os_ = R.seg_rep((H, W), pred.copy(), 0.3)
    np.testing.assert_array_equal(ob[0], tb[0])
    np.testing.assert_array_equal(os_[0], ts[0])
    lines = ob[0][os_[0] > 0.6].astype(np.int32)
    # group_output (textblock.py:421-508)
    theirs = ref.TB.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8)
    ours = R.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8)
    same_blocks(ours, theirs)
    # refine_mask / refine_undetected_mask (textmask.py:135-169), both modes
    for mode in (0, 1):
        a = R.refine_mask(page, mask_u8, ours, mode)
        b = ref.TM.refine_mask(page, mask_u8, theirs, refine_mode=mode)
        np.testing.assert_array_equal(a, b)
    m1, m2 = mask_u8.copy(), mask_u8.copy()
    a = R.refine_undetected_mask(page, m1, R.refine_mask(page, mask_u8, ours, 1), ours[: len(ours) // 2], 1)
    b = ref.TM.refine_undetected_mask(page, m2, ref.TM.refine_mask(page, mask_u8, theirs, refine_mode=1),
                                      theirs[: len(theirs) // 2], refine_mode=1)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(m1, m2)


def test_speckle_bitmaps_equal_reference_code(ref):
    """Noisy prob maps: hundreds of tiny contours, nested holes (the DB path's hard cases)."""
    from scipy import ndimage
    for seed in range(4):
        rng = np.random.RandomState(50 + seed)
        H, W = 64 + 8 * seed, 96
        prob = ndimage.uniform_filter(rng.rand(H, W), 1 + seed % 3).astype(np.float32)
        prob = (prob - prob.min()) / (prob.max() - prob.min()) * 0.6
        pred = np.stack([prob, np.zeros_like(prob)])[None]
        tb, ts = ref.DB.SegDetectorRepresenter(thresh=0.3)(None, pred.copy())
        ob, os_ = R.seg_rep((H, W), pred.copy(), 0.3)
        np.testing.assert_array_equal(ob[0], tb[0])
        np.testing.assert_array_equal(os_[0], ts[0])


def letterboxed_case(seed, im_hw=(420, 300), size=512):
    """A page of another aspect ratio + network outputs that are consistent with its letterbox."""
    from oracle import cv_ref as cv
    from test_gpu_e2e import blks_tensor
    page512, mask_u8, prob, blks = fake_outputs(seed, size)
    im_h, im_w = im_hw
    r = min(size / im_h, size / im_w)
    nw, nh = int(round(im_w * r)), int(round(im_h * r))
    page = cv.resize_linear_u8(np.ascontiguousarray(page512[:nh, :nw]), (im_w, im_h))
    mask = mask_u8.astype(np.float32) / 255
    mask[nh:], mask[:, nw:] = 0, 0
    prob = prob.copy()
    prob[nh:], prob[:, nw:] = 0.01, 0.01
    keep = (blks[0][:, 2] <= nw) & (blks[0][:, 3] <= nh)
    blks = tuple(b[keep] for b in blks)
    return page, blks_tensor(blks), mask[None, None], np.stack([prob, np.zeros_like(prob)])[None], (size - nw, size - nh)


@pytest.mark.parametrize("seed,keep", [(0, False), (1, True)])
def test_whole_detector_call_equals_reference_code(ref, seed, keep):
    """`TextDetector.__call__` of the reference (inference.py:141-178: preprocess_img + letterbox,
    postprocess_yolo / postprocess_mask, crop + resize back, group_output, refine_mask, ...) on a
    page that needs a real letterbox, with the network replaced by fixed outputs."""
    from oracle import ref_post_import as RP
    page, blks, mask, lines_map, (dw, dh) = letterboxed_case(seed)
    det = RP.reference_detector(ref, (torch.from_numpy(blks.copy()), torch.from_numpy(mask.copy()),
                                      torch.from_numpy(lines_map.copy())), input_size=(512, 512))
    mode = 1 if keep else 0
    tm, tr, tb = det(page.copy(), refine_mode=mode, keep_undetected_mask=keep)
    # the letterbox geometry the reference derived is the one the outputs were built for
    img_in, ratio, rdw, rdh = ref.INF.preprocess_img(page.copy(), input_size=(512, 512), device="cpu")
    assert (rdw, rdh) == (dw, dh) and tuple(img_in.shape) == (1, 3, 512, 512)
    om, orf, ob = R.detector_tail(page.copy(), blks.copy(), mask.copy(), lines_map.copy(), input_size=(512, 512),
                                  dw=dw, dh=dh, refine_mode=mode, keep_undetected_mask=keep)
    np.testing.assert_array_equal(om, tm)
    np.testing.assert_array_equal(orf, tr)
    same_blocks(ob, tb)
    assert len(tb) > 0 and tr.any()
    # preprocess_img itself: the oracle's letterbox + channel handling (App. C-1: the net sees BGR)
    from oracle import cv_ref as cv
    lb, _, (odw, odh) = cv.letterbox(page, (512, 512))
    assert (odw, odh) == (dw, dh)
    np.testing.assert_array_equal(img_in[0].numpy(), lb.transpose(2, 0, 1).astype(np.float32) / 255)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_refine_mask_adversarial_windows_equal_reference_code(ref, seed):
    """The scenario of tests/test_gpu_e2e.py::test_refine_mask_gpu_merge_stage_on_adversarial_windows
    (noisy colours, blob masks, overlapping / border / thin blocks): oracle vs the reference's code,
    so the GPU merge stage is compared with something that is itself pinned."""
    rng = np.random.RandomState(seed)
    H, W = 384, 512
    page = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    page[:, : W // 2] = (page[:, : W // 2] // 64) * 64
    mask = np.zeros((H, W), np.uint8)
    for _ in range(60):
        y, x = rng.randint(0, H), rng.randint(0, W)
        hh, ww = rng.randint(2, 40), rng.randint(2, 60)
        mask[y: y + hh, x: x + ww] = rng.randint(40, 256)
    boxes = [[0, 0, 90, 70], [60, 40, 220, 160], [200, 100, 330, 230], [W - 120, H - 90, W, H],
             [10, 300, 400, 306], [430, 5, 436, 200], [100, 100, 180, 150]]
    for _ in range(5):
        x1, y1 = rng.randint(0, W - 40), rng.randint(0, H - 40)
        boxes.append([x1, y1, min(W, x1 + rng.randint(12, 200)), min(H, y1 + rng.randint(12, 150))])
    ours = [R.TextBlock(b) for b in boxes]
    theirs = [ref.TB.TextBlock(b) for b in boxes]
    for mode in (0, 1):
        np.testing.assert_array_equal(R.refine_mask(page, mask, ours, mode),
                                      ref.TM.refine_mask(page, mask, theirs, refine_mode=mode))
    m1, m2 = mask.copy(), mask.copy()
    a = R.refine_undetected_mask(page, m1, R.refine_mask(page, mask, ours[:4], 0), ours[:4], 0)
    b = ref.TM.refine_undetected_mask(page, m2, ref.TM.refine_mask(page, mask, theirs[:4], refine_mode=0), theirs[:4],
                                      refine_mode=0)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(m1, m2)


def test_reference_example_page_golden_is_what_the_reference_code_gives(ref):
    """tests/golden/real_page.npz (the reference's real example page, network outputs from its published mask)
    holds the results of the reference's own `TextDetector.__call__`: regenerate and compare."""
    import json
    import os
    from conftest import GOLDEN
    from oracle import annot_ref as A
    from oracle import ref_post_import as RP
    from oracle.gen_golden_real import SIZE, load_fixture
    page, blks, mask_u8, prob, (dw, dh), g = load_fixture(os.path.join(GOLDEN, "real_page.npz"))
    mask_f = (mask_u8.astype(np.float32) / 255)[None, None]
    lines_map = np.stack([prob, np.zeros_like(prob)])[None]
    det = RP.reference_detector(ref, (torch.from_numpy(blks.copy()), torch.from_numpy(mask_f.copy()),
                                      torch.from_numpy(lines_map.copy())), input_size=(SIZE, SIZE))
    m, r, b = det(page.copy(), refine_mode=1, keep_undetected_mask=True)
    np.testing.assert_array_equal(m, g["mask1"])
    np.testing.assert_array_equal(np.packbits(r > 0), g["refined1"])
    rec = json.dumps([t.to_dict() for t in b], ensure_ascii=False, cls=A.NumpyEncoder)
    assert rec.encode("utf8") == g["records1"].tobytes()
    img_in, ratio, rdw, rdh = ref.INF.preprocess_img(page.copy(), input_size=(SIZE, SIZE), device="cpu")
    assert (rdw, rdh) == (dw, dh)
