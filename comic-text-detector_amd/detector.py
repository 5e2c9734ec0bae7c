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
from .textmask import (REFINEMASK_ANNOTATION, REFINEMASK_INPAINT, refine_mask, refine_mask_batch,   # noqa: F401
                       refine_undetected_mask)

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

    # -- preprocessing: `preprocess_img` + `letterbox` (inference.py:72-83, imgproc_utils.py:86-117).
    #    The aspect-keeping bilinear resize and the bottom/right zero padding run on the GPU
    #    (ctd_resize_linear_u8); the /255 and the layout change are fused into the stem kernel.
    #    Channel order: BGR2RGB (:74) followed by [::-1] (:77) = the net consumes BGR planes,
    #    and the per-channel resize commutes with the swaps, so the BGR page is resized as is.
    def _prepare(self, pages: Sequence[np.ndarray]):
        Hn, Wn = self.input_size[1], self.input_size[0]
        canv, metas = [], []
        for p in pages:
            if p.dtype != np.uint8 or p.ndim != 3 or p.shape[2] != 3:
                raise ValueError("pages must be uint8 BGR (H,W,3) arrays")
            im_h, im_w = p.shape[:2]
            r = min(Hn / im_h, Wn / im_w)
            nw, nh = int(round(im_w * r)), int(round(im_h * r))
            dw, dh = int(Wn - nw), int(Hn - nh)
            src = torch.from_numpy(np.ascontiguousarray(p)).to(self.net.device)
            canv.append(BK.resize_linear_u8(src, (nh, nw), (Hn, Wn)))
            metas.append((im_h, im_w, dw, dh))
        return torch.stack(canv), metas

    @torch.no_grad()
    def detect_batch(self, pages: Sequence[np.ndarray], refine_mode=REFINEMASK_INPAINT,
                     keep_undetected_mask=False) -> List[Tuple[np.ndarray, np.ndarray, List[TextBlock]]]:
        x, metas = self._prepare(pages)
        blks, mask, lines_map = self.net.forward_u8(x)                      # the seam (inference.py:146)
        return self.tail_batch(pages, blks, self.net.mask_u8, lines_map[:, 0], self.net.bitmap, refine_mode,
                               keep_undetected_mask, metas)

    def tail_batch(self, pages: Sequence[np.ndarray], blks: torch.Tensor, mask_u8: torch.Tensor,
                   prob: torch.Tensor, bitmap: torch.Tensor, refine_mode=REFINEMASK_INPAINT,
                   keep_undetected_mask=False, metas=None):
        """Everything after the network (inference.py:148-178) for a batch whose network outputs are
        already on the GPU: blks (B,rows,no) f32, mask_u8 (B,H,W) u8, prob = lines_map[:,0] (B,H,W) f32,
        bitmap (B,H,W) u8.  metas[b] = (im_h, im_w, dw, dh) of the letterbox (default: no resize)."""
        B = len(pages)
        Hn, Wn = self.input_size[1], self.input_size[0]
        if metas is None:
            metas = [(p.shape[0], p.shape[1], 0, 0) for p in pages]
        ratios = [(im_w / (Wn - dw), im_h / (Hn - dh)) for im_h, im_w, dw, dh in metas]       # :148
        yolo = PP.postprocess_yolo(blks, self.conf_thresh, self.nms_thresh, ratios)             # :149
        boxes, scores = self.seg_rep(prob, bitmap)                                              # :158
        masks, masks_gpu, blk_lists = [], [], []
        for b in range(B):
            im_h, im_w, dw, dh = metas[b]
            keep = scores[b] > 0.6                                          # box_thresh (:159-161)
            lines = boxes[b][keep]
            if lines.size == 0:
                lines = []
            else:
                lines = lines.astype(np.float64)
                lines[..., 0] *= ratios[b][0]
                lines[..., 1] *= ratios[b][1]
                lines = lines.astype(np.int32)
            # fused postprocess_mask (:156), crop of the padding (:164), resize to the page (:165)
            m = mask_u8[b, : Hn - dh, : Wn - dw]
            if (im_h, im_w) != (Hn - dh, Wn - dw):
                m = BK.resize_linear_u8(m.contiguous(), (im_h, im_w))
            m = m.contiguous()
            masks_gpu.append(m)
            masks.append(BK.to_host(m, "tail.mask").copy())
            blk_lists.append(group_output(yolo[b], lines, im_w, im_h, masks[b]))     # :173
        # refine_mask (:174) for the whole batch: the windows of all pages share the launches
        dev = self.net.device
        gpu = [(torch.from_numpy(np.ascontiguousarray(pages[b])).to(dev), masks_gpu[b]) for b in range(B)]
        refined = refine_mask_batch(pages, masks, blk_lists, refine_mode, dev, gpu)
        out = []
        for b in range(B):
            r = refined[b]
            if keep_undetected_mask:
                r = refine_undetected_mask(pages[b], masks[b], r, blk_lists[b], refine_mode, dev)
            out.append((masks[b], r, blk_lists[b]))
        return out

    def __call__(self, img: np.ndarray, refine_mode=REFINEMASK_INPAINT, keep_undetected_mask=False):
        return self.detect_batch([img], refine_mode, keep_undetected_mask)[0]
