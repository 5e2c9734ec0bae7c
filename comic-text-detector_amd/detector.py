"""`TextDetector`: mirror of the reference's L4 API (reference inference.py:116-178)
on top of the HIP backend.

    det = TextDetector(model_path_or_ckpt, input_size=1024, device='cuda')
    mask, mask_refined, blk_list = det(img_bgr_uint8, refine_mode, keep_undetected_mask)

Same constructor arguments, same call signature, same return triple as the
reference.  `detect_batch(pages)` is the batched form the reference lacks (it is
bs=1 only): one fused forward + one NMS + two labelling
launches for the whole batch, then the per-page grouping / refinement.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple, Union

import numpy as np
import torch

from . import backend as BK
from . import postproc as PP
from .textblock import TextBlock, group_output
from .textmask import REFINEMASK_ANNOTATION, REFINEMASK_INPAINT, refine_mask, refine_undetected_mask

__all__ = ["TextDetector", "TextBlock", "REFINEMASK_INPAINT", "REFINEMASK_ANNOTATION"]


class TextDetector:
    lang_list = ["eng", "ja", "unknown"]                      # inference.py:117
    langcls2idx = {"eng": 0, "ja": 1, "unknown": 2}

    def __init__(self, model_path: Union[str, dict], input_size=1024, device="cuda", half=True,
                 nms_thresh=0.35, conf_thresh=0.4, mask_thresh=0.3, act="leaky"):
        if isinstance(input_size, int):
            input_size = (input_size, input_size)
        self.input_size = input_size
        self.device = device
        self.half = half
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        self.net = BK.HipTextDetBackend(model_path, device=device, precision="fp16" if half else "fp32", act=act,
                                        bitmap_thresh=0.3)
        self.backend = "hip"
        self.seg_rep = PP.SegRepresenter(thresh=0.3)          # inference.py:139

    # -- preprocessing: host mirror of inference.py:72-83 for pages that already have the
    #    network size (the letterbox resize of arbitrary pages is SURVEY row f-1, not built yet)
    def _pack(self, pages: Sequence[np.ndarray]) -> torch.Tensor:
        H, W = self.input_size[1], self.input_size[0]
        for p in pages:
            if p.shape[:2] != (H, W) or p.dtype != np.uint8 or p.shape[2] != 3:
                raise NotImplementedError(
                    f"pages must be uint8 BGR of the network size {H}x{W}; letterbox resize of other sizes "
                    "(reference imgproc_utils.py:86-117) is not built yet")
        # BGR2RGB then [::-1] on channels (inference.py:74,77): the net consumes BGR planes
        return torch.from_numpy(np.stack(pages)).to(self.net.device)

    @torch.no_grad()
    def detect_batch(self, pages: Sequence[np.ndarray], refine_mode=REFINEMASK_INPAINT,
                     keep_undetected_mask=False) -> List[Tuple[np.ndarray, np.ndarray, List[TextBlock]]]:
        x = self._pack(pages)
        blks, mask, lines_map = self.net.forward_u8(x)                      # the seam (inference.py:146)
        return self.tail_batch(pages, blks, self.net.mask_u8, lines_map[:, 0], self.net.bitmap, refine_mode,
                               keep_undetected_mask)

    def tail_batch(self, pages: Sequence[np.ndarray], blks: torch.Tensor, mask_u8: torch.Tensor,
                   prob: torch.Tensor, bitmap: torch.Tensor, refine_mode=REFINEMASK_INPAINT,
                   keep_undetected_mask=False):
        """Everything after the network (inference.py:148-178) for a batch whose network outputs are
        already on the GPU: blks (B,rows,no) f32, mask_u8 (B,H,W) u8, prob = lines_map[:,0] (B,H,W) f32,
        bitmap (B,H,W) u8."""
        B = len(pages)
        im_h, im_w = pages[0].shape[:2]
        ratio = (im_w / self.input_size[0], im_h / self.input_size[1])      # dw = dh = 0 (:148)
        yolo = PP.postprocess_yolo(blks, self.conf_thresh, self.nms_thresh, [ratio] * B)      # :149
        mask_np = mask_u8.cpu().numpy()                                     # fused postprocess_mask (:156)
        boxes, scores = self.seg_rep(prob, bitmap)                          # :158
        out = []
        for b in range(B):
            keep = scores[b] > 0.6                                          # box_thresh (:159-161)
            lines = boxes[b][keep]
            if lines.size == 0:
                lines = []
            else:
                lines = lines.astype(np.float64)
                lines[..., 0] *= ratio[0]
                lines[..., 1] *= ratio[1]
                lines = lines.astype(np.int32)
            m = mask_np[b].copy()
            blk_list = group_output(yolo[b], lines, im_w, im_h, m)          # :173
            refined = refine_mask(pages[b], m, blk_list, refine_mode, self.net.device)        # :174
            if keep_undetected_mask:
                refined = refine_undetected_mask(pages[b], m, refined, blk_list, refine_mode, self.net.device)
            out.append((m, refined, blk_list))
        return out

    def __call__(self, img: np.ndarray, refine_mode=REFINEMASK_INPAINT, keep_undetected_mask=False):
        return self.detect_batch([img], refine_mode, keep_undetected_mask)[0]
