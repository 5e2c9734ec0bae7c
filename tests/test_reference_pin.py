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
