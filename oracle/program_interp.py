"""TEST INFRASTRUCTURE.  A CPU (torch fp32) interpreter of the lowered op
program (include/ctd_hip.h semantics).  It lets the CPU test-suite check the
HOST lowering (`comic-text-detector_amd/graph.py`: BN folding, concat/upsample
folding, sibling-conv merging, Detect row offsets, fused tail parameter packing)
against the oracle network without a GPU.  It is NOT a fallback: nothing in the
product imports it.
"""
from __future__ import annotations

import importlib
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

L = importlib.import_module("comic-text-detector_amd._lib")

_ACT = {0: lambda x: x, 1: F.silu, 2: lambda x: F.leaky_relu(x, 0.1), 3: F.relu, 4: torch.sigmoid}


def run_program(prog, x: torch.Tensor, bitmap_thresh: float = 0.3):
    """x: (B,3,H,W) f32 in [0,1].  Returns dict(blks, mask, lines, mask_u8, bitmap)."""
    blob = torch.from_numpy(prog.blob())
    B, _, H, W = x.shape
    T: Dict[int, torch.Tensor] = {}

    def tens(tid: int) -> torch.Tensor:
        if tid not in T:
            c, down, _ = prog.tensors[tid]
            T[tid] = torch.zeros(B, c, H >> down, W >> down)
        return T[tid]

    def view(o, which: str) -> torch.Tensor:
        tid = o[which]
        t = tens(tid)[:, o[which + "_coff"]: o[which + "_coff"] + o[which + "_c"]]
        if o[which + "_up"]:
            t = F.interpolate(t, scale_factor=2.0, mode="nearest")
        return t

    def srcs(o) -> torch.Tensor:
        a = view(o, "src0")
        if o["src1"] >= 0:
            a = torch.cat([a, view(o, "src1")], 1)
        return a

    def par(off: int, n: int) -> torch.Tensor:
        return blob[off: off + n]

    unit = (H // 64) * (W // 64)
    rows = sum(d["na"] * (64 // d["stride"]) ** 2 for d in prog.det_levels) * unit
    no = prog.meta["no"]
    out = dict(blks=torch.zeros(B, rows, no), mask=torch.zeros(B, 1, H, W), lines=torch.zeros(B, prog.meta.get("line_planes", 2), H, W),
               mask_u8=torch.zeros(B, H, W, dtype=torch.uint8), bitmap=torch.zeros(B, H, W, dtype=torch.uint8))

    for o in prog.ops:
        k = o["kind"]
        if k == L.OP_INPUT:
            tens(o["dst"])[:] = 0           # channels beyond the image's 3 are zero padding
            tens(o["dst"])[:, :3] = x
        elif k == L.OP_STEM:
            w = par(o["w_off"], o["cout"] * 3 * 36).view(o["cout"], 3, 6, 6)
            b = par(o["b_off"], o["cout"])
            tens(o["dst"])[:, o["dst_coff"]: o["dst_coff"] + o["cout"]] = _ACT[o["act"]](F.conv2d(x, w, b, 2, 2))
        elif k == L.OP_CONV:
            a = srcs(o)
            cin = a.shape[1]
            w = par(o["w_off"], o["cout"] * cin * o["k"] ** 2).view(o["cout"], cin, o["k"], o["k"])
            b = par(o["b_off"], o["cout"]) if o["b_off"] >= 0 else None
            y = _ACT[o["act"]](F.conv2d(a, w, b, o["stride"], o["pad"]))
            if o["res"] >= 0:
                y = y + tens(o["res"])[:, o["res_coff"]: o["res_coff"] + o["cout"]]
            tens(o["dst"])[:, o["dst_coff"]: o["dst_coff"] + o["cout"]] = y
        elif k == L.OP_CONVT:
            a = srcs(o)
            cin = a.shape[1]
            w = par(o["w_off"], o["cout"] * cin * o["k"] ** 2).view(cin, o["cout"], o["k"], o["k"])
            b = par(o["b_off"], o["cout"]) if o["b_off"] >= 0 else None
            y = _ACT[o["act"]](F.conv_transpose2d(a, w, b, o["stride"], o["pad"]))
            tens(o["dst"])[:, o["dst_coff"]: o["dst_coff"] + o["cout"]] = y
        elif k == L.OP_MAXPOOL:
            a = view(o, "src0")
            tens(o["dst"])[:, o["dst_coff"]: o["dst_coff"] + o["src0_c"]] = F.max_pool2d(a, o["k"], 1, o["k"] // 2)
        elif k == L.OP_AVGPOOL2:
            a = view(o, "src0")
            tens(o["dst"])[:, o["dst_coff"]: o["dst_coff"] + o["src0_c"]] = F.avg_pool2d(a, 2, 2)
        elif k == L.OP_DETECT:
            stride, row_unit, na, no_ = o["aux"][:4]
            raw = view(o, "src0")                       # (B, na*no, ny, nx)
            ny, nx = raw.shape[2:]
            r = raw.view(B, na, no_, ny, nx).permute(0, 1, 3, 4, 2).sigmoid()
            yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
            grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
            anc = torch.tensor(o["faux"][: 2 * na]).view(1, na, 1, 1, 2)
            y = r.clone()
            y[..., 0:2] = (r[..., 0:2] * 2 - 0.5 + grid) * stride
            y[..., 2:4] = (r[..., 2:4] * 2) ** 2 * anc
            r0 = row_unit * unit
            out["blks"][:, r0: r0 + na * ny * nx] = y.reshape(B, -1, no_)
        elif k == L.OP_EXPORT:
            a = view(o, "src0")[:, 0]
            which, plane = o["aux"][0], o["aux"][1]
            if which == L.OUT_MASK:
                out["mask"][:, 0] = a
                out["mask_u8"] = (a * 255).to(torch.uint8)
            else:
                out["lines"][:, plane] = a
                if plane == 0:
                    out["bitmap"] = (a > o["faux"][0]).to(torch.uint8)
        elif k == L.OP_SEG_FINAL:
            a = view(o, "src0")
            cin = a.shape[1]
            w = par(o["w_off"], cin * 16).view(cin, 1, 4, 4)
            m = torch.sigmoid(F.conv_transpose2d(a, w, None, 2, 1))
            out["mask"] = m
            out["mask_u8"] = (m[:, 0] * 255).to(torch.uint8)
        elif k == L.OP_DB_UP:
            a = view(o, "src0")
            q = o["aux"][1]
            pb = q * q * 4 + q + q * 4 + 1
            for br in range(o["aux"][2] or 2):
                p = par(o["w_off"] + br * pb, pb)
                w1 = p[: q * q * 4].view(q, q, 2, 2)
                b1 = p[q * q * 4: q * q * 4 + q]
                w2 = p[q * q * 4 + q: q * q * 4 + q + q * 4].view(q, 1, 2, 2)
                b2 = p[q * q * 4 + q + q * 4:]
                h = F.relu(F.conv_transpose2d(a[:, br * q: (br + 1) * q], w1, b1, 2, 0))
                y = torch.sigmoid(F.conv_transpose2d(h, w2, b2, 2, 0))
                out["lines"][:, br] = y[:, 0]
                if br == 0:
                    out["bitmap"] = (y[:, 0] > o["faux"][0]).to(torch.uint8)
        else:
            raise ValueError(f"unknown op kind {k}")
    return out
