"""-m gpu: the acceptance metric of BASELINE.json's `north_star` measured END TO END -- the product
detector (HIP engine -> native tail) against the oracle (CPU fp32 network -> oracle tail = the
reference's `TextDetector.__call__` restated) on the same pages, with a checkpoint whose maps have real
contours (`synth.make_blob_checkpoint`).  Two engines:

  * exact-fp32 engine (`half=False`, the reference's precision): masks equal up to single-level flips on
    <0.1 % of the pixels, boxes / lines / blocks identical on these pages;
  * fp16-operand MFMA engine (`half=True`, the benchmarked configuration): a different arithmetic than the
    reference's fp32, so thresholded outputs can differ along blob boundaries; the measured IoUs are printed
    (they are also in bench.py's `parity` block) and asserted against the floors written below."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import accept, cv_ref as cv
from oracle import postproc_ref as R
from oracle.net_ref import OracleNet

pytestmark = pytest.mark.gpu

_CK = {}


def blob_ckpt():
    if 0 not in _CK:
        _CK[0] = pkg().synth.make_blob_checkpoint(0)
    return _CK[0]


def oracle_result(ck, page, size):
    lb, ratio, (dw, dh) = cv.letterbox(page, (size, size))
    x = torch.from_numpy(np.ascontiguousarray(lb.transpose(2, 0, 1)[None])).float() / 255
    torch.set_num_threads(16)
    blks, mask, lines_map = OracleNet(ck)(x)
    return R.detector_tail(page, blks.numpy(), mask.numpy(), lines_map.numpy(), input_size=(size, size), dw=dw, dh=dh,
                           refine_mode=0, keep_undetected_mask=False)


@pytest.mark.parametrize("prec,size,shape", [("fp32", 512, (512, 512)), ("fp32s", 512, (512, 512)), ("fp32s", 1024, (1024, 1024)),
                                             ("fp32s", 512, (700, 495)), ("fp16", 512, (512, 512)),
                                             ("fp16", 1024, (1024, 1024)), ("fp16", 512, (700, 495))])
def test_detector_end_to_end_vs_oracle(prec, size, shape):
    p = pkg()
    ck = blob_ckpt()
    half = prec == "fp16"
    page = p.synth.text_like_page(shape, 3, n_blocks=8)
    det = p.detector.TextDetector(ck, input_size=size, device="cuda", precision=prec)
    got = det(page, refine_mode=0, keep_undetected_mask=False)
    ref = oracle_result(ck, page, size)
    rep = accept.compare(got, ref)
    print(f"\nacceptance engine={prec} size={size} page={shape}: {rep}")
    assert rep["lines"]["ref"] >= 5 and rep["blocks"]["ref"] >= 3            # the pages have something to compare
    if not half:
        # the reference's precision: single-level mask flips only, geometry identical
        assert rep["mask_u8_max_level_diff"] <= 1 and rep["mask_u8_equal_frac"] > 0.999
        assert rep["lines"]["identical"] == rep["lines"]["ref"] == rep["lines"]["ours"]
        assert rep["blocks"]["identical"] == rep["blocks"]["ref"] == rep["blocks"]["ours"]
        assert rep["refined_mask_equal_frac"] > 0.9999
    else:
        assert rep["mask_u8_max_level_diff"] <= 2
        assert rep["mask_iou_at_127"] > 0.995
        assert rep["lines"]["mean_iou"] > 0.9 and rep["blocks"]["mean_iou"] > 0.9
        assert rep["refined_mask_equal_frac"] > 0.995
