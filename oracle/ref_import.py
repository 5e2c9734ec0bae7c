"""TEST INFRASTRUCTURE.  Imports the reference's OWN torch modules from
/root/reference (read-only, only present in the build container -- never on
the GPU box) with stub modules for the wheels that are not installed
(cv2, torchvision, torchsummary, wandb, pyclipper, shapely), and composes them
exactly as `TextDetBase.forward` does (`basemodel.py:240-244`).

Used to (1) validate `oracle/net_ref.py` and (2) generate the golden vectors in
`tests/golden/` (`oracle/gen_golden.py`).  Nothing is copied from the reference.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "basemodel.py"))


def _install_stubs() -> None:
    sys.dont_write_bytecode = True          # the reference tree is read-only
    for name in ("cv2", "torchvision", "torchvision.ops", "torchsummary", "wandb",
                 "pyclipper", "shapely", "shapely.geometry", "requests", "PIL", "PIL.Image"):
        if name in sys.modules:
            continue
        try:
            __import__(name)
            continue
        except Exception:
            pass
        m = types.ModuleType(name)
        m.__dict__.setdefault("__path__", [])
        sys.modules[name] = m
    sys.modules["torchsummary"].__dict__.setdefault("summary", lambda *a, **k: None)
    sys.modules["shapely.geometry"].__dict__.setdefault("Polygon", object)
    cv2 = sys.modules["cv2"]
    if not hasattr(cv2, "imshow"):
        cv2.imshow = lambda *a, **k: None
        cv2.setNumThreads = lambda *a, **k: None


_MODULES = None


def load_reference_modules():
    """Returns (Model, UnetHead, DBHead, load_yolov5_ckpt, fuse_conv_and_bn)."""
    global _MODULES
    if _MODULES is not None:
        return _MODULES
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    _install_stubs()
    # the reference uses top-level packages called `models` and `utils`
    for shadow in ("models", "utils"):
        if shadow in sys.modules and not getattr(sys.modules[shadow], "__file__", "").startswith(REFERENCE_ROOT):
            del sys.modules[shadow]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        from models.yolov5.yolo import Model, load_yolov5_ckpt  # type: ignore
        from basemodel import UnetHead, DBHead, TEXTDET_INFERENCE  # type: ignore
        from utils.yolov5_utils import fuse_conv_and_bn  # type: ignore
    finally:
        sys.path.remove(REFERENCE_ROOT)
    _MODULES = (Model, UnetHead, DBHead, load_yolov5_ckpt, fuse_conv_and_bn, TEXTDET_INFERENCE)
    return _MODULES


class ReferenceNet:
    """The reference network half, fp32 CPU, from a checkpoint dict in the
    reference's own format (same steps as `get_base_det_models`,
    `basemodel.py:211-220`, minus `torch.load`)."""

    def __init__(self, ckpt: dict, act: str = "leaky"):
        import copy
        import torch
        Model, UnetHead, DBHead, load_yolov5_ckpt, _fuse, self._mode = load_reference_modules()
        self.torch = torch
        blk = {"cfg": copy.deepcopy(ckpt["blk_det"]["cfg"]), "weights": ckpt["blk_det"]["weights"]}
        self.blk_det = load_yolov5_ckpt(blk, map_location="cpu").eval()
        self.text_seg = UnetHead(act=act)
        self.text_seg.load_state_dict(ckpt["text_seg"])
        self.text_seg.eval()
        self.text_det = DBHead(64, act=act)
        self.text_det.load_state_dict(ckpt["text_det"])
        self.text_det.eval()

    def __call__(self, x):
        """`TextDetBase.forward` (`basemodel.py:240-244`) but keeps the whole
        batch of blks (the reference returns `blks[0]` = the decoded tensor
        for ALL images; `[0]` there indexes the (pred, raw) tuple)."""
        with self.torch.no_grad():
            blks, feats = self.blk_det(x, detect=True)
            mask, feats2 = self.text_seg(*feats, forward_mode=self._mode)
            lines = self.text_det(*feats2, step_eval=False)
        return blks[0], mask, lines

    def features(self, x):
        with self.torch.no_grad():
            blks, feats = self.blk_det(x, detect=True)
        return blks[0], feats
