#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- generates tests/golden/annot_seed*.json with the reference's OWN
`TextBlock.to_dict`, `xyxy2yolo` and `get_yololabel_strings` (imported from /root/reference with stub
modules; only possible in the build container).  The blocks come from the oracle's `group_output`
on the synthetic text-like outputs of tests/test_post_host.py and are rebuilt as reference
TextBlock objects, so the golden strings pin the record schema (attribute set and order), the
number formatting and the label arithmetic.  `NumpyEncoder` is restated (oracle/annot_ref.py):
the reference's does not import under NumPy 2.

    python oracle/gen_golden_annot.py
"""
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import annot_ref as A            # noqa: E402
from oracle import postproc_ref as R         # noqa: E402
from oracle import ref_import as RI          # noqa: E402


def reference_modules():
    RI._install_stubs()
    cv2 = sys.modules["cv2"]
    if not hasattr(cv2, "INTER_LINEAR"):
        def _const(name):                                     # constants used as default arguments
            if name.startswith("__"):
                raise AttributeError(name)
            return 0
        cv2.__getattr__ = _const
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "utils" or k.startswith("utils.")}
    sys.path.insert(0, RI.REFERENCE_ROOT)
    try:
        import utils.imgproc_utils as IU
        import utils.textblock as TB
    finally:
        sys.path.remove(RI.REFERENCE_ROOT)
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    return TB, IU


def main():
    from test_post_host import fake_outputs
    TB, IU = reference_modules()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for seed in (0, 1):
        page, mask_u8, prob, blks = fake_outputs(seed, 512)
        H, W = prob.shape
        boxes, scores = R.boxes_from_bitmap(prob, prob > 0.3, W, H)
        lines = boxes[scores > 0.6].astype(np.int32)
        oblks = R.group_output(copy.deepcopy(blks), lines.copy(), W, H, mask_u8)
        keys = ("lines", "language", "vertical", "font_size", "distance", "angle", "vec", "norm", "merged", "weight")
        rblks = [TB.TextBlock(b.xyxy, **{k: copy.deepcopy(getattr(b, k)) for k in keys}) for b in oblks]
        xy = IU.xyxy2yolo([b.xyxy for b in rblks], W, H)
        polys = []
        for b in rblks:
            polys += b.lines
        import io
        buf = io.StringIO()
        np.savetxt(buf, np.array(polys).reshape(-1, 8), fmt="%d")
        gold = {"seed": seed, "size": 512, "n_blocks": len(rblks),
                "yolo": IU.get_yololabel_strings([1] * len(xy), xy) if xy is not None else "",
                "lines": buf.getvalue(),
                "json": json.dumps([b.to_dict() for b in rblks], ensure_ascii=False, cls=A.NumpyEncoder),
                "record_keys": list(rblks[0].to_dict().keys())}
        with open(os.path.join(out_dir, f"annot_seed{seed}.json"), "w", encoding="utf8") as f:
            json.dump(gold, f, ensure_ascii=False, indent=1)
        print("seed", seed, "blocks", len(rblks), "json bytes", len(gold["json"]))


if __name__ == "__main__":
    main()
