"""Lowering of a reference-format checkpoint to the op program the HIP engine
executes (include/ctd_hip.h: ctd_tensor / ctd_op / parameter blob).

What the lowering does (host side, once per model load -- the counterpart of
`load_yolov5_ckpt(...).fuse()` + `get_base_det_models`, reference
`yolo.py:285-311`, `basemodel.py:211-220`):

  * folds every BatchNorm into the preceding conv / conv-transpose weights
    (backbone eps 1e-3 like `fuse_conv_and_bn`, `utils/yolov5_utils.py:23-43`;
    head BNs eps 1e-5 are folded too -- the reference leaves them unfused,
    `basemodel.py:226-238`, which changes fp rounding only);
  * removes `torch.cat` / `nn.Upsample` / `Concat` by giving consumers two
    (tensor, channel-slice, upsample) sources, and C3's internal cat by letting
    producers write into channel slices of one tensor;
  * merges the sibling 1x1 convs cv1/cv2 of every C3 (`common.py:131-132`) into
    one GEMM, and the two 3x3 convs of the DB binarize/thresh branches
    (`basemodel.py:96,135`) into one;
  * emits fused head/tail ops for the fp16 path (stem, seg-final, db-up) and a
    plain generic program for the fp32 path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import arch
from . import _lib as L


@dataclass
class View:
    """A logical NHWC activation: channel slice of a tensor, optionally seen
    through a nearest x2 upsample.  `down` = log2 of (input H / logical H)."""
    tid: int
    coff: int
    c: int
    down: int
    up: int = 0


Cat = List[View]


class Program:
    def __init__(self, prec: int):
        self.prec = prec
        self.tensors: List[Tuple[int, int, int]] = []   # (channels, log2_down, dtype)
        self.ops: List[dict] = []
        self._params: List[np.ndarray] = []
        self._noff = 0
        self.taps: Dict[str, int] = {}                   # name -> tensor id (debug / per-layer parity)
        self.det_levels: List[dict] = []
        self.meta: dict = {}

    # -- building blocks ----------------------------------------------------
    def tensor(self, channels: int, down: int, dtype: int = 0, name: Optional[str] = None) -> int:
        self.tensors.append((int(channels), int(down), int(dtype)))
        tid = len(self.tensors) - 1
        if name:
            self.taps[name] = tid
        return tid

    def param(self, arr: np.ndarray) -> int:
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        off = self._noff
        self._params.append(a)
        # keep every block 4-float aligned (16 B) for vector loads of packed params
        pad = (-a.size) % 4
        if pad:
            self._params.append(np.zeros(pad, np.float32))
        self._noff += a.size + pad
        return off

    def blob(self) -> np.ndarray:
        return np.concatenate(self._params) if self._params else np.zeros(1, np.float32)

    def op(self, kind: int, **kw) -> dict:
        o = dict(kind=kind, src0=-1, src0_coff=0, src0_c=0, src0_up=0, src1=-1, src1_coff=0, src1_c=0, src1_up=0,
                 res=-1, res_coff=0, dst=-1, dst_coff=0, cout=0, k=1, stride=1, pad=0, act=0, w_off=-1, b_off=-1,
                 aux=[0] * 8, faux=[0.0] * 8, name="")
        o.update(kw)
        self.ops.append(o)
        return o

    def _set_srcs(self, kw: dict, x: Cat) -> None:
        if len(x) > 2:
            raise NotImplementedError("concat of more than two producers is not supported by the engine")
        v = x[0]
        kw.update(src0=v.tid, src0_coff=v.coff, src0_c=v.c, src0_up=v.up)
        if len(x) == 2:
            u = x[1]
            if u.down != v.down:
                raise ValueError("concat of different resolutions")
            kw.update(src1=u.tid, src1_coff=u.coff, src1_c=u.c, src1_up=u.up)

    def conv(self, x: Cat, w: np.ndarray, b: Optional[np.ndarray], k: int, s: int, p: int, act: str,
             dst: Optional[View] = None, res: Optional[View] = None, name: str = "", dtype: int = 0,
             pad_channels_to: int = 1) -> View:
        cout = w.shape[0]
        down = x[0].down + int(math.log2(s))
        if dst is None:
            ch = -(-cout // pad_channels_to) * pad_channels_to
            dst = View(self.tensor(ch, down, dtype, name or None), 0, cout, down)
        kw = dict(dst=dst.tid, dst_coff=dst.coff, cout=cout, k=k, stride=s, pad=p, act=L.ACT[act],
                  w_off=self.param(w), b_off=self.param(b) if b is not None else -1, name=name)
        self._set_srcs(kw, x)
        if res is not None:
            kw.update(res=res.tid, res_coff=res.coff)
        self.op(L.OP_CONV, **kw)
        return View(dst.tid, dst.coff, cout, down)

    def convt(self, x: View, w: np.ndarray, b: Optional[np.ndarray], k: int, s: int, p: int, act: str,
              name: str = "") -> View:
        cout = w.shape[1]
        down = x.down - int(math.log2(s))
        dst = View(self.tensor(cout, down, 0, name or None), 0, cout, down)
        kw = dict(dst=dst.tid, dst_coff=0, cout=cout, k=k, stride=s, pad=p, act=L.ACT[act],
                  w_off=self.param(w), b_off=self.param(b) if b is not None else -1, name=name)
        self._set_srcs(kw, [x])
        self.op(L.OP_CONVT, **kw)
        return dst


# ---------------------------------------------------------------------------
# BN folding
# ---------------------------------------------------------------------------

def fold(sd, cs: arch.ConvSpec) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """Returns (weights in torch layout, bias or None), BatchNorm folded in
    float64 then rounded once to float32."""
    w = sd[cs.prefix + ".weight"].detach().cpu().numpy().astype(np.float64)
    b = sd[cs.prefix + ".bias"].detach().cpu().numpy().astype(np.float64) if cs.bias else None
    if cs.bn_prefix is not None:
        g = sd[cs.bn_prefix + ".weight"].detach().cpu().numpy().astype(np.float64)
        beta = sd[cs.bn_prefix + ".bias"].detach().cpu().numpy().astype(np.float64)
        mean = sd[cs.bn_prefix + ".running_mean"].detach().cpu().numpy().astype(np.float64)
        var = sd[cs.bn_prefix + ".running_var"].detach().cpu().numpy().astype(np.float64)
        scale = g / np.sqrt(var + cs.bn_eps)
        if cs.transposed:
            w = w * scale.reshape(1, -1, 1, 1)
        else:
            w = w * scale.reshape(-1, 1, 1, 1)
        b = (0.0 if b is None else b * scale) + (beta - mean * scale)
    return w.astype(np.float32), (None if b is None else np.asarray(b, np.float32))


class _Lower:
    def __init__(self, prog: Program, sd, fast: bool):
        self.p = prog
        self.sd = sd
        self.fast = fast

    def conv(self, x: Cat, cs: arch.ConvSpec, **kw) -> View:
        w, b = fold(self.sd, cs)
        return self.p.conv(x, w, b, cs.k, cs.s, cs.p, cs.act, name=kw.pop("name", cs.prefix), **kw)

    def c3(self, x: Cat, c3: arch.C3Spec, name: str) -> View:
        """`C3.forward` (reference common.py:137-138) without materialising the cat."""
        assert c3.cv1.act == c3.cv2.act
        w1, b1 = fold(self.sd, c3.cv1)
        w2, b2 = fold(self.sd, c3.cv2)
        c_ = w1.shape[0]
        y = self.p.conv(x, np.concatenate([w1, w2], 0), np.concatenate([b1, b2], 0), 1, 1, 0, c3.cv1.act,
                        name=name + ".cv1+cv2")
        a = View(y.tid, 0, c_, y.down)
        for j, bs in enumerate(c3.m):
            t = self.conv([a], bs.cv1)
            w, b = fold(self.sd, bs.cv2)
            self.p.conv([t], w, b, 3, 1, 1, bs.cv2.act, dst=a, res=a if bs.add else None, name=bs.cv2.prefix)
        return self.conv([View(y.tid, 0, 2 * c_, y.down)], c3.cv3)

    def sppf(self, x: Cat, sp: arch.SPPFSpec, name: str) -> View:
        """`SPPF.forward` (reference common.py:190-196): pools write the cat slots."""
        w, b = fold(self.sd, sp.cv1)
        c_ = w.shape[0]
        down = x[0].down
        P = self.p.tensor(4 * c_, down, 0, name + ".cat")
        self.p.conv(x, w, b, 1, 1, 0, sp.cv1.act, dst=View(P, 0, c_, down), name=sp.cv1.prefix)
        for j in range(3):
            self.p.op(L.OP_MAXPOOL, src0=P, src0_coff=j * c_, src0_c=c_, dst=P, dst_coff=(j + 1) * c_, k=sp.k,
                      name=f"{name}.m{j}")
        return self.conv([View(P, 0, 4 * c_, down)], sp.cv2)

    def up_block(self, x: Cat, ub: arch.UpBlockSpec, name: str) -> View:
        """`double_conv_up_c3` (reference basemodel.py:21-32)."""
        y = self.c3(x, ub.c3, name + ".conv.0")
        w, b = fold(self.sd, ub.up)
        return self.p.convt(y, w, b, ub.up.k, ub.up.s, ub.up.p, ub.up.act, name=ub.up.prefix)


def lower(ckpt: dict, precision: int, act: str = "leaky", bitmap_thresh: float = 0.3, db_thresh: bool = True) -> Program:
    """Checkpoint dict (reference format) -> Program."""
    fast = precision == L.PREC_F16
    prog = Program(precision)
    layers, meta = arch.parse_yolo_cfg(ckpt["blk_det"]["cfg"])
    strides = arch.detect_strides(layers)
    prog.meta = dict(meta, strides=strides)

    # ---- yolo: `Model._forward_once` (reference yolo.py:115-134) --------------
    ylo = _Lower(prog, ckpt["blk_det"]["weights"], fast)
    outs: Dict[int, Cat] = {}
    feats: List[View] = []
    cur: Optional[Cat] = None           # None = the network input
    for Lr in layers:
        src = cur if Lr.f == -1 else (outs[Lr.f] if isinstance(Lr.f, int) else None)
        if Lr.kind == "Conv":
            cs: arch.ConvSpec = Lr.spec
            if src is None:                      # first layer reads the image
                w, b = fold(ylo.sd, cs)
                if fast and (cs.k, cs.s, cs.p, cs.c1, cs.c2) == (6, 2, 2, 3, 32):
                    t = prog.tensor(cs.c2, 1, 0, cs.prefix)
                    prog.op(L.OP_STEM, dst=t, cout=cs.c2, k=6, stride=2, pad=2, act=L.ACT[cs.act],
                            w_off=prog.param(w), b_off=prog.param(b), name=cs.prefix)
                    v = View(t, 0, cs.c2, 1)
                else:
                    # the image is stored with a zero 4th channel (16-B pixels): the f32 MFMA kernel takes one tap
                    # per 16-B chunk; the matching weight channel is zero
                    t_in = prog.tensor(4, 0, 0, "input")
                    prog.op(L.OP_INPUT, dst=t_in, name="input")
                    w4 = np.concatenate([w, np.zeros_like(w[:, :1])], 1) if w.shape[1] == 3 else w
                    v = prog.conv([View(t_in, 0, w4.shape[1], 0)], w4, b, cs.k, cs.s, cs.p, cs.act, name=cs.prefix)
            else:
                v = ylo.conv(src, cs)
            cur = [v]
        elif Lr.kind == "C3":
            cur = [ylo.c3(src, Lr.spec, f"model.{Lr.i}")]
        elif Lr.kind == "SPPF":
            cur = [ylo.sppf(src, Lr.spec, f"model.{Lr.i}")]
        elif Lr.kind == "Upsample":
            if len(src) != 1 or src[0].up:
                raise NotImplementedError("Upsample of a concat / of an upsampled tensor")
            v = src[0]
            cur = [View(v.tid, v.coff, v.c, v.down - 1, up=1)]
        elif Lr.kind == "Concat":
            cat: Cat = []
            for j in Lr.f:
                cat += (cur if j == -1 else outs[j])
            cur = cat
        elif Lr.kind == "Detect":
            na, no = Lr.extra["na"], Lr.extra["no"]
            row_unit = 0
            det = ckpt["blk_det"]["weights"]
            anchors = det[f"model.{Lr.i}.anchors"].detach().cpu().numpy().astype(np.float32)   # (nl, na, 2), /stride
            for lvl, (j, cs) in enumerate(zip(Lr.f, Lr.spec)):
                w, b = fold(det, cs)
                raw = prog.conv(outs[j], w, b, 1, 1, 0, "none", name=cs.prefix, dtype=1 if fast else 0,
                                pad_channels_to=8)
                s = strides[lvl]
                anc_px = (anchors[lvl] * s).reshape(-1)
                prog.op(L.OP_DETECT, src0=raw.tid, src0_coff=0, src0_c=na * no,
                        aux=[s, row_unit, na, no, 0, 0, 0, 0],
                        faux=[float(a) for a in anc_px] + [0.0] * (8 - 2 * na), name=f"model.{Lr.i}.decode{lvl}")
                prog.det_levels.append(dict(stride=s, na=na, no=no, row_unit=row_unit))
                row_unit += na * (64 // s) ** 2
            cur = None
        outs[Lr.i] = cur
        if Lr.i in arch.OUT_INDICES:
            assert cur is not None and len(cur) == 1 and not cur[0].up
            feats.append(cur[0])
            prog.taps[f"feat{len(feats) - 1}"] = cur[0].tid
    f160, f80, f40, f20, f3 = feats

    # ---- `UnetHead.forward`, TEXTDET_INFERENCE (reference basemodel.py:62-78) ---
    u = arch.unet_spec(act)
    sl = _Lower(prog, ckpt["text_seg"], fast)
    pooled = prog.tensor(f3.c, f3.down + 1, 0, "seg.pool")
    prog.op(L.OP_AVGPOOL2, src0=f3.tid, src0_coff=f3.coff, src0_c=f3.c, dst=pooled, name="seg.down_conv1.down")
    d10 = sl.c3([View(pooled, 0, f3.c, f3.down + 1)], u.down_conv1, "seg.down_conv1.conv")
    u20 = sl.up_block([d10], u.upconv0, "seg.upconv0")
    u40 = sl.up_block([f20, u20], u.upconv2, "seg.upconv2")
    u80 = sl.up_block([f40, u40], u.upconv3, "seg.upconv3")
    u160 = sl.up_block([f80, u80], u.upconv4, "seg.upconv4")
    u320 = sl.up_block([f160, u160], u.upconv5, "seg.upconv5")
    prog.taps["u40"] = u40.tid
    prog.taps["u320"] = u320.tid
    w6, _ = fold(sl.sd, u.upconv6)
    if u.upconv6.c1 == 64:                     # fused 64 -> 1 ConvT + sigmoid + u8 mask (fp16 and fp32 engines)
        prog.op(L.OP_SEG_FINAL, src0=u320.tid, src0_coff=0, src0_c=u320.c, cout=1, k=4, stride=2, pad=1,
                act=L.ACT["sigmoid"], w_off=prog.param(w6), aux=[L.OUT_MASK] + [0] * 7, name="seg.upconv6")
    else:
        m = prog.convt(u320, w6, None, 4, 2, 1, "sigmoid", name="seg.upconv6")
        prog.op(L.OP_EXPORT, src0=m.tid, src0_coff=0, src0_c=1, aux=[L.OUT_MASK, 0] + [0] * 6, name="seg.export")

    # ---- `DBHead.forward`, eval, step_eval=False (reference basemodel.py:106-125) --
    d = arch.db_spec(64, act)
    dl = _Lower(prog, ckpt["text_det"], fast)
    du80 = dl.up_block([f40, u40], d.upconv3, "db.upconv3")
    dx = dl.up_block([f80, du80], d.upconv4, "db.upconv4")
    dx = dl.conv([dx], d.conv, name="db.conv.0")
    prog.taps["db.x"] = dx.tid
    q = d.binarize.conv3.c2
    # `db_thresh=False`: the threshold branch (`lines_map[:, 1]`) is not lowered at all -- `TextDetector` only
    # reads the shrink map (reference utils/db_utils.py:63 `pred[:, 0]`); `lines_map` then has ONE plane
    branches = (d.binarize, d.thresh) if db_thresh else (d.binarize,)
    nbr = len(branches)
    prog.meta["line_planes"] = nbr
    if q == 16:                                # fused DB tail (fp16 and fp32 engines)
        folded = [fold(dl.sd, br.conv3) for br in branches]
        y = prog.conv([dx], np.concatenate([f[0] for f in folded], 0), np.concatenate([f[1] for f in folded], 0), 3, 1, 1,
                      "relu", name="db.binarize.0+thresh.0" if db_thresh else "db.binarize.0")
        packed = []
        for br in branches:
            w1, b1 = fold(dl.sd, br.up1)        # (q, q, 2, 2), BN folded
            w2, b2 = fold(dl.sd, br.up2)        # (q, 1, 2, 2)
            packed += [w1.reshape(-1), b1.reshape(-1), w2.reshape(-1), b2.reshape(-1)]
        prog.op(L.OP_DB_UP, src0=y.tid, src0_coff=0, src0_c=nbr * q, w_off=prog.param(np.concatenate(packed)),
                aux=[L.OUT_LINES, q, nbr] + [0] * 5, faux=[bitmap_thresh] + [0.0] * 7, name="db.up")
    else:
        for plane, br in enumerate(branches):
            t = dl.conv([dx], br.conv3)
            w1, b1 = fold(dl.sd, br.up1)
            t = prog.convt(t, w1, b1, 2, 2, 0, "relu", name=br.up1.prefix)
            w2, b2 = fold(dl.sd, br.up2)
            t = prog.convt(t, w2, b2, 2, 2, 0, "sigmoid", name=br.up2.prefix)
            prog.op(L.OP_EXPORT, src0=t.tid, src0_coff=0, src0_c=1, aux=[L.OUT_LINES, plane, nbr] + [0] * 5,
                    faux=[bitmap_thresh] + [0.0] * 7, name=f"db.export{plane}")
    return prog


def to_ctypes(prog: Program):
    """Program -> (ctd_tensor[], ctd_op[], float blob) ready for ctd_engine_create."""
    T = (L.CtdTensor * len(prog.tensors))()
    for i, (c, d, dt) in enumerate(prog.tensors):
        T[i].channels, T[i].log2_down, T[i].dtype = c, d, dt
    O = (L.CtdOp * len(prog.ops))()
    for i, o in enumerate(prog.ops):
        for k, v in o.items():
            if k == "name":
                continue
            if k == "aux":
                for j in range(8):
                    O[i].aux[j] = int(v[j])
            elif k == "faux":
                for j in range(8):
                    O[i].faux[j] = float(v[j])
            else:
                setattr(O[i], k, int(v))
    return T, O, prog.blob()
