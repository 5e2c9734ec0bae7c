"""`ctd_db_boxes_compact` (csrc/host_db.cpp, host only): boxes and scores from the device-compacted
component tables (emulated with numpy here, tests/dbc_emul.py) against the oracle's contour walk +
polygon fill (reference utils/db_utils.py:123-211) -- text-like maps, speckle at several correlation
lengths (nested holes, islands, peninsulas, diagonal links), thin lines, empty / full maps and more
contours than the candidate cap."""
import numpy as np
import pytest

from conftest import pkg
from dbc_emul import boxes_from_tables, dbc_tables
from oracle import postproc_ref as R
from test_post_host import fake_outputs


def check(prob, cap=1000):
    bitmap = prob > 0.3
    H, W = prob.shape
    boxes, scores = boxes_from_tables(pkg(), dbc_tables(prob, bitmap), cap)
    rboxes, rscores = R.boxes_from_bitmap(prob, bitmap, W, H, max_candidates=cap)
    np.testing.assert_array_equal(boxes, rboxes)
    np.testing.assert_allclose(scores, rscores, rtol=0, atol=1e-6)
    return scores


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_text_like(seed):
    _, _, prob, _ = fake_outputs(seed, 384)
    assert (check(prob) > 0.6).sum() > 3


@pytest.mark.parametrize("case", ["holes", "thin", "empty", "full", "cap", "frame"])
def test_edge_cases(case):
    H, W = 96, 160
    prob = np.full((H, W), 0.05, np.float32)
    cap = 1000
    if case == "holes":
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
            prob[50 + i, 60 + i] = 0.9
        prob[70:73, 100:140] = 0.9
    elif case == "full":
        prob[:] = 0.9
    elif case == "cap":
        prob[::3, ::3] = 0.9
        prob[40:60, 40:100] = 0.9
        cap = 50
    elif case == "frame":                   # pockets closed against the page frame are not holes
        prob[0:40, 0:50] = 0.9
        prob[10:30, 0:30] = 0.1
        prob[60:96, 100:160] = 0.9
        prob[70:96, 120:150] = 0.1
        prob[50:58, 0:160] = 0.8
    s = check(prob, cap)
    if case == "holes":
        assert (s > 0).sum() == 6


def test_speckle_sweep():
    from scipy import ndimage
    for seed in range(12):
        rng = np.random.RandomState(100 + seed)
        H, W = 64 + 3 * seed, 100 + 5 * seed
        prob = ndimage.uniform_filter(rng.rand(H, W), 1 + seed % 4).astype(np.float32)
        prob = (prob - prob.min()) / (prob.max() - prob.min()) * 0.62
        check(prob)
