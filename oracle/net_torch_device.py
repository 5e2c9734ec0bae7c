"""TEST / BENCH INFRASTRUCTURE.  The oracle network (oracle/net_ref.py = the reference's torch modules restated with
`torch.nn.functional`) with its tensors on a DEVICE and in a chosen dtype: the "stock ROCm" baseline BASELINE.md 3.4
asks for -- the reference's own network as PyTorch-ROCm runs it (ATen + MIOpen; reference basemodel.py:222-244 with
`device='cuda'`, fp32, and `.half()` for fp16 as basemodel.py:218-219 intends).  Same operator sequence as the
reference: yolo Conv+BN folded (`Model.fuse`, yolo.py:185-192), the heads' BatchNorms left as separate eval-mode ops
(basemodel.py:226-227 never fuses them), NCHW tensors.

Only bench.py's `rocm_baseline` leg uses this.  It is a BASELINE, never the product path."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import net_ref as NR


class _DevRunner(NR._Runner):
    def __init__(self, sd: Dict[str, torch.Tensor], fuse_bn: bool, device, dtype):
        super().__init__(sd, fuse_bn)
        self.device, self.dtype = device, dtype
        self._cache: Dict[str, tuple] = {}

    def _to(self, t):
        return None if t is None else t.to(device=self.device, dtype=self.dtype)

    def _params(self, cs):
        hit = self._cache.get(cs.prefix)
        if hit is not None:
            return hit
        sd = self.sd
        if cs.bn_prefix is not None and self.fuse_bn:
            w, b = NR.fused_conv_params(sd, cs)
            out = (self._to(w), self._to(b), None)
        else:
            w = sd[cs.prefix + ".weight"].float()
            b = sd[cs.prefix + ".bias"].float() if cs.bias else None
            bn = None
            if cs.bn_prefix is not None:
                p = cs.bn_prefix
                bn = tuple(self._to(sd[p + k].float()) for k in (".running_mean", ".running_var", ".weight", ".bias"))
            out = (self._to(w), self._to(b), bn)
        self._cache[cs.prefix] = out
        return out

    def conv(self, x, cs):
        w, b, bn = self._params(cs)
        if cs.transposed:
            y = F.conv_transpose2d(x, w, b, cs.s, cs.p)
        else:
            y = F.conv2d(x, w, b, cs.s, cs.p)
        if bn is not None:
            y = F.batch_norm(y, bn[0], bn[1], bn[2], bn[3], False, 0.0, cs.bn_eps)
        return NR._act(y, cs.act)


class TorchDeviceNet(NR.OracleNet):
    """`OracleNet` on `device` in `dtype` (torch.float32 / torch.float16).  Call with (B,3,H,W) in [0,1] on the device."""

    def __init__(self, ckpt: dict, device="cuda", dtype=torch.float32, act: str = "leaky"):
        super().__init__(ckpt, act)
        self.device, self.dtype = torch.device(device), dtype
        self.y = _DevRunner(ckpt["blk_det"]["weights"], True, self.device, dtype)
        self.s = _DevRunner(ckpt["text_seg"], False, self.device, dtype)
        self.d = _DevRunner(ckpt["text_det"], False, self.device, dtype)
        self.anchors = self.anchors.to(self.device)

    def __call__(self, x: torch.Tensor):
        with torch.no_grad():
            blks, feats = self.yolo(x.to(device=self.device, dtype=self.dtype))
            mask, feats2 = self.seg(*feats)
            lines = self.det(*feats2)
        return blks, mask, lines
