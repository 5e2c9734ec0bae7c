"""comic-text-detector_amd: MI355X (gfx950) native inference hot path of
dmMaze/comic-text-detector -- the fused CNN forward (YOLOv5s backbone + Detect,
UNet segmentation head, DBNet head) and its GPU post-processing -- behind the
reference's own backend seam `net(img_in) -> (blks, mask, lines_map)`
(reference inference.py:124-130,146).

Import with `importlib.import_module("comic-text-detector_amd")` (the hyphen is
the project's name) or through the `ctd_amd` alias module at the repo root.
"""
from . import arch, synth  # noqa: F401  (pure python, no GPU needed)

__all__ = ["arch", "synth", "graph", "backend", "HipTextDetBackend"]


def __getattr__(name):
    # lazy: these need torch + ctypes
    if name in ("graph", "backend", "_lib", "detector", "postproc", "dist", "textblock", "textmask", "tail", "annotations", "affinity"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    if name == "HipTextDetBackend":
        from .backend import HipTextDetBackend
        return HipTextDetBackend
    if name == "TextDetector":
        from .detector import TextDetector
        return TextDetector
    raise AttributeError(name)
