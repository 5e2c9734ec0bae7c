"""TEST INFRASTRUCTURE.  CPU fp32 restatement of the reference's fused CNN
forward (`basemodel.py:240-244` TextDetBase.forward), written with plain
`torch.nn.functional` calls on the checkpoint's state dicts.  It exists
because /root/reference is not present on the GPU box: this file IS the oracle
there.  It is pinned against the reference's own modules (oracle/ref_import.py)
by tests/test_oracle_net.py (bit-exact on CPU) and against tests/golden/*.npz.

Each function cites the reference lines it restates.
"""
from __future__ import annotations

import importlib
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

arch = importlib.import_module("comic-text-detector_amd.arch")


def _act(x: torch.Tensor, act: str) -> torch.Tensor:
    if act == "silu":
        return F.silu(x)                      # common.py:37 (nn.SiLU)
    if act == "leaky":
        return F.leaky_relu(x, 0.1)           # common.py:39-40
    if act == "relu":
        return F.relu(x)
    if act == "sigmoid":
        return torch.sigmoid(x)
    return x


def fused_conv_params(sd: Dict[str, torch.Tensor], cs) -> Tuple[torch.Tensor, torch.Tensor]:
    """`fuse_conv_and_bn` (`utils/yolov5_utils.py:23-43`), same operation order."""
    w = sd[cs.prefix + ".weight"].float()
    g, b = sd[cs.bn_prefix + ".weight"].float(), sd[cs.bn_prefix + ".bias"].float()
    m, v = sd[cs.bn_prefix + ".running_mean"].float(), sd[cs.bn_prefix + ".running_var"].float()
    scale = g.div(torch.sqrt(cs.bn_eps + v))
    wf = (scale.view(-1, 1) * w.view(w.shape[0], -1)).view(w.shape)
    b_conv = sd[cs.prefix + ".bias"].float() if cs.bias else torch.zeros(w.shape[0])
    b_bn = b - g.mul(m).div(torch.sqrt(v + cs.bn_eps))
    return wf, scale * b_conv + b_bn


class _Runner:
    def __init__(self, sd: Dict[str, torch.Tensor], fuse_bn: bool):
        self.sd = sd
        self.fuse_bn = fuse_bn

    def conv(self, x: torch.Tensor, cs) -> torch.Tensor:
        """`Conv.forward` / `forward_fuse` (`common.py:45-49`) and the bare
        nn.Conv2d / nn.ConvTranspose2d (+BN +act) sequences of `basemodel.py`."""
        sd = self.sd
        if cs.bn_prefix is not None and self.fuse_bn:
            w, b = fused_conv_params(sd, cs)
            y = F.conv2d(x, w, b, cs.s, cs.p)
        else:
            w = sd[cs.prefix + ".weight"].float()
            b = sd[cs.prefix + ".bias"].float() if cs.bias else None
            if cs.transposed:
                y = F.conv_transpose2d(x, w, b, cs.s, cs.p)
            else:
                y = F.conv2d(x, w, b, cs.s, cs.p)
            if cs.bn_prefix is not None:
                p = cs.bn_prefix
                y = F.batch_norm(y, sd[p + ".running_mean"].float(), sd[p + ".running_var"].float(),
                                 sd[p + ".weight"].float(), sd[p + ".bias"].float(), False, 0.0, cs.bn_eps)
        return _act(y, cs.act)

    def bottleneck(self, x, bs):
        y = self.conv(self.conv(x, bs.cv1), bs.cv2)        # common.py:103-104
        return x + y if bs.add else y

    def c3(self, x, c3):
        a = self.conv(x, c3.cv1)                            # common.py:137-138
        for bs in c3.m:
            a = self.bottleneck(a, bs)
        return self.conv(torch.cat((a, self.conv(x, c3.cv2)), 1), c3.cv3)

    def sppf(self, x, sp):
        x = self.conv(x, sp.cv1)                            # common.py:190-196
        y1 = F.max_pool2d(x, sp.k, 1, sp.k // 2)
        y2 = F.max_pool2d(y1, sp.k, 1, sp.k // 2)
        y3 = F.max_pool2d(y2, sp.k, 1, sp.k // 2)
        return self.conv(torch.cat([x, y1, y2, y3], 1), sp.cv2)

    def up_block(self, x, ub):
        return self.conv(self.c3(x, ub.c3), ub.up)          # basemodel.py:21-32


def detect_decode(raw: Sequence[torch.Tensor], anchors: torch.Tensor, strides: Sequence[int],
                  na: int, no: int) -> torch.Tensor:
    """`Detect.forward` inference branch (`yolo.py:23-44`) + `_make_grid` (`:46-55`).
    raw[i]: (B, na*no, ny, nx) conv outputs.  anchors: (nl, na, 2) stride-normalised."""
    z = []
    for i, x in enumerate(raw):
        bs, _, ny, nx = x.shape
        x = x.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        yv, xv = torch.meshgrid([torch.arange(ny, device=x.device), torch.arange(nx, device=x.device)], indexing="ij")
        grid = torch.stack((xv, yv), 2).expand((1, na, ny, nx, 2)).float()
        anchor_grid = (anchors[i].clone() * strides[i]).view((1, na, 1, 1, 2)).expand((1, na, ny, nx, 2)).float()
        y = x.sigmoid()
        y[..., 0:2] = (y[..., 0:2] * 2 - 0.5 + grid) * strides[i]
        y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * anchor_grid
        z.append(y.view(bs, -1, no))
    return torch.cat(z, 1)


class OracleNet:
    """fp32 CPU oracle of `net(img_in) -> (blks, mask, lines_map)`."""

    def __init__(self, ckpt: dict, act: str = "leaky"):
        self.cfg = ckpt["blk_det"]["cfg"]
        self.layers, self.meta = arch.parse_yolo_cfg(self.cfg)
        self.strides = arch.detect_strides(self.layers)
        self.y = _Runner(ckpt["blk_det"]["weights"], fuse_bn=True)     # yolo.py:296 .fuse()
        self.s = _Runner(ckpt["text_seg"], fuse_bn=False)              # heads are never fused (basemodel.py:226-227)
        self.d = _Runner(ckpt["text_det"], fuse_bn=False)
        self.unet = arch.unet_spec(act)
        self.db = arch.db_spec(64, act)
        det = self.layers[-1]
        self.anchors = ckpt["blk_det"]["weights"][f"model.{det.i}.anchors"].float()

    # -- yolo: `Model._forward_once` (yolo.py:115-134) --------------------
    def yolo(self, x: torch.Tensor):
        outs: Dict[int, torch.Tensor] = {}
        feats: List[torch.Tensor] = []
        cur = x
        blks = None
        for L in self.layers:
            if L.f != -1:
                cur = outs[L.f] if isinstance(L.f, int) else [cur if j == -1 else outs[j] for j in L.f]
            if L.kind == "Conv":
                cur = self.y.conv(cur, L.spec)
            elif L.kind == "C3":
                cur = self.y.c3(cur, L.spec)
            elif L.kind == "SPPF":
                cur = self.y.sppf(cur, L.spec)
            elif L.kind == "Upsample":
                cur = F.interpolate(cur, scale_factor=2.0, mode="nearest")
            elif L.kind == "Concat":
                cur = torch.cat(cur, 1)
            elif L.kind == "Detect":
                raw = [self.y.conv(t, cs) for t, cs in zip(cur, L.spec)]
                blks = detect_decode(raw, self.anchors, self.strides, L.extra["na"], L.extra["no"])
                cur = blks
            outs[L.i] = cur
            if L.i in arch.OUT_INDICES:
                feats.append(cur)
        return blks, feats

    # -- `UnetHead.forward` (basemodel.py:62-78), TEXTDET_INFERENCE -------
    def seg(self, f160, f80, f40, f20, f3):
        u = self.unet
        d10 = self.s.c3(F.avg_pool2d(f3, 2, 2), u.down_conv1)          # basemodel.py:34-45
        u20 = self.s.up_block(d10, u.upconv0)
        u40 = self.s.up_block(torch.cat([f20, u20], 1), u.upconv2)
        u80 = self.s.up_block(torch.cat([f40, u40], 1), u.upconv3)
        u160 = self.s.up_block(torch.cat([f80, u80], 1), u.upconv4)
        u320 = self.s.up_block(torch.cat([f160, u160], 1), u.upconv5)
        mask = self.s.conv(u320, u.upconv6)
        return mask, (f80, f40, u40)

    # -- `DBHead.forward` (basemodel.py:106-125), eval, step_eval=False ---
    def det(self, f80, f40, u40):
        d = self.db
        u80 = self.d.up_block(torch.cat([f40, u40], 1), d.upconv3)
        x = self.d.up_block(torch.cat([f80, u80], 1), d.upconv4)
        x = self.d.conv(x, d.conv)
        outs = []
        for br in (d.binarize, d.thresh):
            t = self.d.conv(self.d.conv(self.d.conv(x, br.conv3), br.up1), br.up2)
            outs.append(t)
        return torch.cat(outs, 1)       # (shrink_maps, threshold_maps)

    def __call__(self, x: torch.Tensor):
        with torch.no_grad():
            blks, feats = self.yolo(x.float())
            mask, feats2 = self.seg(*feats)
            lines = self.det(*feats2)
        return blks, mask, lines

    @staticmethod
    def step_function(lines: torch.Tensor, k: float = 50.0) -> torch.Tensor:
        """`DBHead.step_function(shrink_maps, threshold_maps)` (reference basemodel.py:159-160), what
        `DBHead.forward(step_eval=True)` returns (:121-122): (B,2,H,W) -> (B,1,H,W)."""
        return torch.reciprocal(1 + torch.exp(-k * (lines[:, 0:1] - lines[:, 1:2])))

    def forward_with_taps(self, x: torch.Tensor):
        with torch.no_grad():
            blks, feats = self.yolo(x.float())
            mask, feats2 = self.seg(*feats)
            lines = self.det(*feats2)
        return blks, mask, lines, feats
