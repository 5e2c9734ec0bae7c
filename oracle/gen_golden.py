"""TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by running the
REFERENCE's own torch modules (imported from /root/reference through
oracle/ref_import.py) on seeded synthetic checkpoints and inputs.

    python -m oracle.gen_golden          (build container only; needs /root/reference)

The fixtures travel to the GPU box; /root/reference does not.  Inputs and
weights are regenerated from seeds at test time (torch's CPU generator is
deterministic for a fixed torch build); each fixture stores checksums of both
so generator drift is detected instead of silently invalidating the fixture.
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("comic-text-detector_amd.synth")

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

SMALL_CASES = {
    # name: (weight seed, input seed, (B, H, W))
    "net_small_a": (0, 1, (1, 128, 128)),
    "net_small_b": (0, 2, (2, 192, 128)),
    "net_small_c": (3, 4, (1, 64, 256)),
}
FULL_CASE = ("net_full_summary", 0, 0, 1024)     # weight seed, page seed, size
TILE = 16


def ckpt_checksum(ckpt: dict) -> float:
    s = 0.0
    for sd in (ckpt["blk_det"]["weights"], ckpt["text_seg"], ckpt["text_det"]):
        for k in sorted(sd):
            s += float(sd[k].double().sum())
    return s


def make_input(seed: int, shape) -> torch.Tensor:
    B, H, W = shape
    return torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(seed))


def page_to_input(page_bgr_u8: np.ndarray) -> torch.Tensor:
    """What `preprocess_img` hands the net for an already 1024x1024 page
    (reference inference.py:72-83): BGR->RGB then [::-1] on channels = BGR planes, /255."""
    x = page_bgr_u8.transpose(2, 0, 1).astype(np.float32) / 255
    return torch.from_numpy(np.ascontiguousarray(x))[None]


def tile_means(t: torch.Tensor) -> np.ndarray:
    return torch.nn.functional.avg_pool2d(t.double(), TILE, TILE).float().numpy()


def main() -> None:
    from oracle.ref_import import ReferenceNet
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    nets = {}
    for name, (wseed, iseed, shape) in SMALL_CASES.items():
        if wseed not in nets:
            ck = synth.make_checkpoint(wseed)
            nets[wseed] = (ck, ReferenceNet(ck))
        ck, ref = nets[wseed]
        x = make_input(iseed, shape)
        blks, mask, lines = ref(x)
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), blks=blks.numpy(), mask=mask.numpy(),
                            lines=lines.numpy(), wseed=wseed, iseed=iseed, shape=np.array(shape),
                            ckpt_sum=ckpt_checksum(ck), input_sum=float(x.double().sum()))
        print(name, tuple(blks.shape), tuple(mask.shape), tuple(lines.shape))

    name, wseed, pseed, size = FULL_CASE
    ck, ref = nets[wseed]
    page = synth.text_like_page((size, size), pseed)
    x = page_to_input(page)
    blks, mask, lines = ref(x)
    obj = blks[0, :, 4]
    top = torch.argsort(obj, descending=True, stable=True)[:256]
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"),
                        mask_tiles=tile_means(mask), lines_tiles=tile_means(lines),
                        top_rows=top.numpy(), top_blks=blks[0, top].numpy(),
                        mask_u8_hist=np.bincount((mask[0, 0] * 255).to(torch.uint8).flatten().numpy(), minlength=256),
                        bitmap_count=int((lines[0, 0] > 0.3).sum()),
                        wseed=wseed, pseed=pseed, size=size, ckpt_sum=ckpt_checksum(ck),
                        input_sum=float(x.double().sum()))
    print(name, "ok")


if __name__ == "__main__":
    main()
