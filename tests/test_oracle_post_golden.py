"""The oracle's post-processing restatement against golden outputs of the reference's OWN code
(tests/golden/post_seed*.npz, generated in the build container by oracle/gen_golden_post.py with
functional stand-ins for the third-party packages).  Runs anywhere, CPU only."""
import copy
import json
import os

import numpy as np
import pytest

from oracle import annot_ref as A
from oracle import postproc_ref as R
from test_gpu_post import random_blks
from test_post_host import fake_outputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_equals_reference_code_golden(seed):
    g = np.load(os.path.join(GOLD, f"post_seed{seed}.npz"))
    page, mask_u8, prob, blks = fake_outputs(seed, 512)
    H, W = prob.shape
    pred = np.stack([prob, np.zeros_like(prob)])[None]
    boxes, scores = R.seg_rep((H, W), pred, 0.3)
    np.testing.assert_array_equal(boxes[0], g["boxes"])
    np.testing.assert_array_equal(scores[0], g["scores"])
    lines = boxes[0][scores[0] > 0.6].astype(np.int32)
    blk_list = R.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8)
    records = json.dumps([b.to_dict() for b in blk_list], ensure_ascii=False, cls=A.NumpyEncoder)
    assert records == bytes(g["records"]).decode("utf8")
    for mode, key in ((0, "refined_inpaint"), (1, "refined_annot")):
        r = R.refine_mask(page, mask_u8, blk_list, mode)
        assert set(np.unique(r)) <= {0, 255}
        np.testing.assert_array_equal(np.packbits(r > 0), g[key])
    m2 = mask_u8.copy()
    und = R.refine_undetected_mask(page, m2, R.refine_mask(page, mask_u8, blk_list, 1), blk_list[: len(blk_list) // 2], 1)
    np.testing.assert_array_equal(np.packbits(und > 0), g["undetected"])
    np.testing.assert_array_equal(m2, g["mask_after_undetected"])
    yolo = random_blks(np.random.RandomState(100 + seed), 1, 2016, frac=0.1, size=512)
    np.testing.assert_array_equal(R.non_max_suppression(yolo, 0.4, 0.35)[0], g["nms"])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_detector_tail_equals_reference_call_golden(seed):
    """`R.detector_tail` vs the reference's `TextDetector.__call__` (fixed network outputs) on a page
    that needs a letterbox, mask crop and resize back to the page."""
    from test_reference_pin import letterboxed_case
    g = np.load(os.path.join(GOLD, f"post_seed{seed}.npz"))
    page, blks, mask, lines_map, (dw, dh) = letterboxed_case(seed)
    assert [dw, dh] == g["det_dwdh"].tolist()
    m, r, b = R.detector_tail(page, blks, mask, lines_map, input_size=(512, 512), dw=dw, dh=dh,
                              refine_mode=seed % 2, keep_undetected_mask=bool(seed % 2))
    np.testing.assert_array_equal(m, g["det_mask"])
    np.testing.assert_array_equal(np.packbits(r > 0), g["det_refined"])
    assert json.dumps([x.to_dict() for x in b], ensure_ascii=False, cls=A.NumpyEncoder) == bytes(g["det_records"]).decode("utf8")


def test_oracle_tail_on_the_reference_example_page_matches_reference_code_golden():
    """The reference's one real fixture (data/examples/AisazuNihaIrarenai-003.jpg + its published mask as the
    source of the network outputs, oracle/gen_golden_real.py): the oracle tail against what the reference's own
    `TextDetector.__call__` produced for it (masks bit for bit, block records byte for byte)."""
    import json
    import os
    from conftest import GOLDEN
    from oracle import annot_ref as A
    from oracle.gen_golden_real import SIZE, load_fixture
    page, blks, mask_u8, prob, (dw, dh), g = load_fixture(os.path.join(GOLDEN, "real_page.npz"))
    assert page.shape[:2] == tuple(g["shape"]) == (1170, 1654)
    mask_f = (mask_u8.astype(np.float32) / 255)[None, None]
    lines_map = np.stack([prob, np.zeros_like(prob)])[None]
    for keep in (0, 1):
        m, r, b = R.detector_tail(page.copy(), blks.copy(), mask_f.copy(), lines_map.copy(), input_size=(SIZE, SIZE), dw=dw,
                                  dh=dh, refine_mode=keep, keep_undetected_mask=bool(keep))
        np.testing.assert_array_equal(m, g[f"mask{keep}"])
        np.testing.assert_array_equal(np.packbits(r > 0), g[f"refined{keep}"])
        rec = json.dumps([t.to_dict() for t in b], ensure_ascii=False, cls=A.NumpyEncoder)
        assert rec.encode("utf8") == g[f"records{keep}"].tobytes()
        assert len(b) == 16
