#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden outputs of the reference's OWN post-processing code (run with the
functional third-party stand-ins of oracle/ref_post_import.py; build container only) on the seeded
synthetic outputs of tests/test_post_host.py::fake_outputs.  tests/test_oracle_post_golden.py checks
the oracle restatement against them on any machine.

    python oracle/gen_golden_post.py
"""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import annot_ref as A              # noqa: E402
from oracle import ref_post_import as RP       # noqa: E402


def main():
    from test_gpu_post import random_blks
    from test_post_host import fake_outputs
    ref = RP.load_reference_post()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for seed in (0, 1, 2):
        page, mask_u8, prob, blks = fake_outputs(seed, 512)
        H, W = prob.shape
        pred = np.stack([prob, np.zeros_like(prob)])[None]
        boxes, scores = ref.DB.SegDetectorRepresenter(thresh=0.3)(None, pred.copy())
        lines = boxes[0][scores[0] > 0.6].astype(np.int32)
        blk_list = ref.TB.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8)
        records = json.dumps([b.to_dict() for b in blk_list], ensure_ascii=False, cls=A.NumpyEncoder)
        refined = [ref.TM.refine_mask(page, mask_u8, blk_list, refine_mode=m) for m in (0, 1)]
        m2 = mask_u8.copy()
        und = ref.TM.refine_undetected_mask(page, m2, refined[1].copy(), blk_list[: len(blk_list) // 2], refine_mode=1)
        rng = np.random.RandomState(100 + seed)
        yolo = random_blks(rng, 1, 2016, frac=0.1, size=512)
        nms = ref.YU.non_max_suppression(torch.from_numpy(yolo), 0.4, 0.35)[0].numpy()
        # the whole TextDetector.__call__ on a page that needs a letterbox (network outputs fixed)
        from test_reference_pin import letterboxed_case
        lpage, lblks, lmask, llines, (dw, dh) = letterboxed_case(seed)
        det = RP.reference_detector(ref, (torch.from_numpy(lblks.copy()), torch.from_numpy(lmask.copy()),
                                          torch.from_numpy(llines.copy())), input_size=(512, 512))
        dm, dr, db = det(lpage.copy(), refine_mode=seed % 2, keep_undetected_mask=bool(seed % 2))
        det_records = json.dumps([b.to_dict() for b in db], ensure_ascii=False, cls=A.NumpyEncoder)
        np.savez_compressed(os.path.join(out_dir, f"post_seed{seed}.npz"), boxes=boxes[0], scores=scores[0],
                            det_mask=dm, det_refined=np.packbits(dr > 0), det_dwdh=np.array([dw, dh]),
                            det_records=np.frombuffer(det_records.encode("utf8"), np.uint8),
                            records=np.frombuffer(records.encode("utf8"), np.uint8),
                            refined_inpaint=np.packbits(refined[0] > 0), refined_annot=np.packbits(refined[1] > 0),
                            undetected=np.packbits(und > 0), mask_after_undetected=m2, nms=nms)
        print("seed", seed, "boxes", len(boxes[0]), "blocks", len(blk_list), "nms", len(nms))


if __name__ == "__main__":
    main()
