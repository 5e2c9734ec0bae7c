"""Where does the fp16 engine's deviation from the fp32 network come from?  (CPU only.)

The oracle network (fp32) is run with the fp16 engine's two roundings -- weights (BN folded) rounded to fp16, every conv
output rounded to fp16 after its activation -- switched on for ONE layer group at a time, for all groups, and for all groups
but one.  Per run, against the exact fp32 maps: max |delta| of the shrink map and of the mask, the DB-bitmap pixels (0.3)
and mask pixels (level 127) that flip.  If a few groups carried the error, a mixed plan (those groups in fp32s) would shrink
the band; if every group contributes alike, it would not.        usage: python scripts/experiments/fp16_error_attribution.py"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("comic-text-detector_amd")
from oracle import net_ref as NR          # noqa: E402

GROUPS = {
    "backbone (model.0-9)": lambda run, pre: run == "y" and int(pre.split(".")[1]) <= 9,
    "neck + detect (model.10-24)": lambda run, pre: run == "y" and int(pre.split(".")[1]) >= 10,
    "seg down_conv1/upconv0/upconv2": lambda run, pre: run == "s" and pre.split(".")[0] in ("down_conv1", "upconv0", "upconv2"),
    "seg upconv3/upconv4": lambda run, pre: run == "s" and pre.split(".")[0] in ("upconv3", "upconv4"),
    "seg upconv5/upconv6 (last ConvT pair + sigmoid)": lambda run, pre: run == "s" and pre.split(".")[0] in ("upconv5", "upconv6"),
    "db upconv3/upconv4": lambda run, pre: run == "d" and pre.split(".")[0] in ("upconv3", "upconv4"),
    "db conv/binarize/thresh (sigmoid tails)": lambda run, pre: run == "d" and pre.split(".")[0] in ("conv", "binarize", "thresh"),
}


class HalfRunner(NR._Runner):
    """`_Runner` whose convs in the active groups use fp16-rounded (BN-folded) weights and round their outputs to fp16."""
    tag, active = "y", ()

    def conv(self, x, cs):
        if not any(GROUPS[g](self.tag, cs.prefix) for g in self.active):
            return super().conv(x, cs)
        sd = self.sd
        if cs.bn_prefix is not None:
            w, b = NR.fused_conv_params(sd, cs) if not cs.transposed else self._fold_t(cs)
        else:
            w = sd[cs.prefix + ".weight"].float()
            b = sd[cs.prefix + ".bias"].float() if cs.bias else None
        w = w.half().float()
        y = F.conv_transpose2d(x, w, b, cs.s, cs.p) if cs.transposed else F.conv2d(x, w, b, cs.s, cs.p)
        return NR._act(y, cs.act).half().float()

    def _fold_t(self, cs):               # ConvTranspose2d weight is (cin, cout, k, k): scale along dim 1
        sd = self.sd
        w = sd[cs.prefix + ".weight"].float()
        g, bb = sd[cs.bn_prefix + ".weight"].float(), sd[cs.bn_prefix + ".bias"].float()
        m, v = sd[cs.bn_prefix + ".running_mean"].float(), sd[cs.bn_prefix + ".running_var"].float()
        scale = g / torch.sqrt(v + cs.bn_eps)
        b_conv = sd[cs.prefix + ".bias"].float() if cs.bias else torch.zeros(w.shape[1])
        return w * scale.view(1, -1, 1, 1), scale * (b_conv - m) + bb


def net_with(ck, active):
    n = NR.OracleNet(ck)
    for tag in ("y", "s", "d"):
        r = getattr(n, tag)
        h = HalfRunner(r.sd, r.fuse_bn)
        h.tag, h.active = tag, tuple(active)
        setattr(n, tag, h)
    return n


if __name__ == "__main__":
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ck = pkg.synth.make_blob_checkpoint(0, sparse_det=True, line_density="fixture")
    pages = [pkg.synth.text_like_page((1024, 1024), s) for s in (0, 131)]
    xs = [torch.from_numpy(np.ascontiguousarray(p.transpose(2, 0, 1)[None])).float() / 255 for p in pages]
    exact = [NR.OracleNet(ck)(x) for x in xs]
    names = list(GROUPS)
    runs = [("all groups (= the fp16 engine's roundings)", names)] + [(f"only {g}", [g]) for g in names] + \
           [(f"all but {g}", [h for h in names if h != g]) for g in names]
    print(f"{'groups rounded to fp16':<62s} max|d prob|  max|d mask|  bitmap flips  mask@127 flips   (two 1024x1024 pages)")
    for title, act in runs:
        n = net_with(ck, act)
        dp = dm = 0.0
        fb = fm = 0
        for x, (eb, em, el) in zip(xs, exact):
            b, m, l = n(x)
            dp = max(dp, float((l[:, 0] - el[:, 0]).abs().max()))
            dm = max(dm, float((m - em).abs().max()))
            fb += int(((l[:, 0] > 0.3) != (el[:, 0] > 0.3)).sum())
            fm += int((((m * 255).to(torch.uint8) > 127) != ((em * 255).to(torch.uint8) > 127)).sum())
        print(f"{title:<62s} {dp:10.2e}  {dm:10.2e}  {fb:11d}  {fm:13d}", flush=True)
