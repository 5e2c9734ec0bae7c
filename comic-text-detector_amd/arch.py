"""Architecture description of the fused 3-head detector, host side.

This is the *host-side mirror* of the reference's model construction:
  * YOLOv5 cfg dict -> layer list      (reference `models/yolov5/yolo.py:208-259` parse_model)
  * UNet segmentation head              (reference `basemodel.py:47-81` UnetHead)
  * DBNet head                          (reference `basemodel.py:83-160` DBHead)

Nothing here touches tensors; it only enumerates modules, their state-dict
prefixes (the reference's weight-file contract, `utils/export.py:23-28`,
`basemodel.py:211-217`) and their hyper-parameters, so that
  - `synth.py` can make a seeded random checkpoint in the reference's format,
  - `graph.py` can lower the network to the op program the HIP runtime executes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

# The release checkpoint stores the yolov5 cfg inside the file
# (`yolo.py:292`); there is no yaml in the reference tree.  This is the
# standard YOLOv5s v6.0 graph with nc=2 that satisfies every structural
# constraint the heads impose (tap channels 64/128/256/512/512,
# `basemodel.py:51-56`; SPPF present, `common.py:181`).
YOLOV5S_CFG: dict = {
    "nc": 2,
    "depth_multiple": 0.33,
    "width_multiple": 0.50,
    "anchors": [
        [10, 13, 16, 30, 33, 23],
        [30, 61, 62, 45, 59, 119],
        [116, 90, 156, 198, 373, 326],
    ],
    "backbone": [
        [-1, 1, "Conv", [64, 6, 2, 2]],
        [-1, 1, "Conv", [128, 3, 2]],
        [-1, 3, "C3", [128]],
        [-1, 1, "Conv", [256, 3, 2]],
        [-1, 6, "C3", [256]],
        [-1, 1, "Conv", [512, 3, 2]],
        [-1, 9, "C3", [512]],
        [-1, 1, "Conv", [1024, 3, 2]],
        [-1, 3, "C3", [1024]],
        [-1, 1, "SPPF", [1024, 5]],
    ],
    "head": [
        [-1, 1, "Conv", [512, 1, 1]],
        [-1, 1, "nn.Upsample", ["None", 2, "nearest"]],
        [[-1, 6], 1, "Concat", [1]],
        [-1, 3, "C3", [512, False]],
        [-1, 1, "Conv", [256, 1, 1]],
        [-1, 1, "nn.Upsample", ["None", 2, "nearest"]],
        [[-1, 4], 1, "Concat", [1]],
        [-1, 3, "C3", [256, False]],
        [-1, 1, "Conv", [256, 3, 2]],
        [[-1, 14], 1, "Concat", [1]],
        [-1, 3, "C3", [512, False]],
        [-1, 1, "Conv", [512, 3, 2]],
        [[-1, 10], 1, "Concat", [1]],
        [-1, 3, "C3", [1024, False]],
        [[17, 20, 23], 1, "Detect", ["nc", "anchors"]],
    ],
}

# Feature taps the heads consume (`basemodel.py:168`, `yolo.py:285`).
OUT_INDICES = (1, 3, 5, 7, 9)

BN_EPS_YOLO = 1e-3   # `utils/yolov5_utils.py:59` via `yolo.py:94`
BN_EPS_HEAD = 1e-5   # nn.BatchNorm2d default; heads are built outside Model()


def make_divisible(x: float, divisor: int) -> int:
    """`utils/yolov5_utils.py:64-68`."""
    return int(math.ceil(x / divisor) * divisor)


@dataclass
class ConvSpec:
    """One reference `Conv` (conv + BN + act, `common.py:30-49`) or bare conv."""
    prefix: str            # state-dict prefix of the nn.Conv2d ('.weight' follows)
    bn_prefix: Optional[str]  # state-dict prefix of the BatchNorm2d, None if no BN
    c1: int
    c2: int
    k: int = 1
    s: int = 1
    p: int = 0
    bias: bool = False     # the nn.Conv2d itself carries a bias
    act: str = "silu"      # 'silu' | 'leaky' | 'relu' | 'sigmoid' | 'none'
    bn_eps: float = BN_EPS_YOLO
    transposed: bool = False  # nn.ConvTranspose2d (weight is (c1, c2, k, k))


@dataclass
class BottleneckSpec:
    cv1: ConvSpec
    cv2: ConvSpec
    add: bool


@dataclass
class C3Spec:
    cv1: ConvSpec
    cv2: ConvSpec
    cv3: ConvSpec
    m: List[BottleneckSpec]


@dataclass
class SPPFSpec:
    cv1: ConvSpec
    cv2: ConvSpec
    k: int


@dataclass
class YoloLayer:
    """One row of the cfg after `parse_model`."""
    i: int
    f: Union[int, List[int]]
    kind: str                      # 'Conv' | 'C3' | 'SPPF' | 'Upsample' | 'Concat' | 'Detect'
    c2: int
    spec: object = None
    extra: dict = field(default_factory=dict)


def _conv(prefix: str, c1: int, c2: int, k: int = 1, s: int = 1, p: Optional[int] = None,
          act: str = "silu", eps: float = BN_EPS_YOLO) -> ConvSpec:
    if p is None:
        p = k // 2                               # autopad, `common.py:24-28`
    return ConvSpec(prefix + ".conv", prefix + ".bn", c1, c2, k, s, p, False, act, eps)


def _c3(prefix: str, c1: int, c2: int, n: int = 1, shortcut: bool = True,
        act: str = "silu", eps: float = BN_EPS_YOLO) -> C3Spec:
    """`common.py:126-138`; hidden width c_ = int(c2 * 0.5)."""
    c_ = int(c2 * 0.5)
    m = []
    for j in range(n):
        bp = f"{prefix}.m.{j}"
        # Bottleneck(c_, c_, shortcut, g, e=1.0): `common.py:94-104`
        m.append(BottleneckSpec(_conv(bp + ".cv1", c_, c_, 1, 1, None, act, eps),
                                _conv(bp + ".cv2", c_, c_, 3, 1, None, act, eps),
                                add=bool(shortcut)))
    return C3Spec(_conv(prefix + ".cv1", c1, c_, 1, 1, None, act, eps),
                  _conv(prefix + ".cv2", c1, c_, 1, 1, None, act, eps),
                  _conv(prefix + ".cv3", 2 * c_, c2, 1, 1, None, act, eps), m)


def parse_yolo_cfg(cfg: dict, ch: int = 3) -> Tuple[List[YoloLayer], dict]:
    """Host restatement of `parse_model` (`yolo.py:208-259`) for the module
    set reachable from a YOLOv5 v6 detection cfg: Conv, C3, SPPF,
    nn.Upsample, Concat, Detect.  Anything else raises (the reference would
    build it, but no released checkpoint of this detector uses it)."""
    anchors, nc = cfg["anchors"], cfg["nc"]
    gd, gw = cfg["depth_multiple"], cfg["width_multiple"]
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers: List[YoloLayer] = []
    chs: List[int] = [ch]
    c2 = ch
    for i, (f, n, m, args) in enumerate(cfg["backbone"] + cfg["head"]):
        args = list(args)
        for j, a in enumerate(args):
            if a == "nc":
                args[j] = nc
            elif a == "anchors":
                args[j] = anchors
            elif a == "None":
                args[j] = None
            elif a == "False":
                args[j] = False
            elif a == "True":
                args[j] = True
        n = max(round(n * gd), 1) if n > 1 else n
        name = m.split(".")[-1] if isinstance(m, str) else m.__name__
        prefix = f"model.{i}"
        fi = (lambda x: x if x < 0 else x + 1)
        if name in ("Conv", "C3", "SPPF"):
            c1 = chs[fi(f)] if f != -1 else chs[-1]
            c2 = args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            if name == "Conv":
                k = args[1] if len(args) > 1 else 1
                s = args[2] if len(args) > 2 else 1
                p = args[3] if len(args) > 3 else None
                spec = _conv(prefix, c1, c2, k, s, p)
            elif name == "C3":
                shortcut = args[1] if len(args) > 1 else True
                spec = _c3(prefix, c1, c2, n, shortcut)
            else:
                k = args[1] if len(args) > 1 else 5
                c_ = c1 // 2
                spec = SPPFSpec(_conv(prefix + ".cv1", c1, c_, 1, 1),
                                _conv(prefix + ".cv2", c_ * 4, c2, 1, 1), k)
            layers.append(YoloLayer(i, f, name, c2, spec))
        elif name == "Upsample":
            c2 = chs[-1] if f == -1 else chs[fi(f)]
            if args[1] != 2 or args[2] != "nearest":
                raise NotImplementedError("only nearest x2 Upsample is supported")
            layers.append(YoloLayer(i, f, "Upsample", c2))
        elif name == "Concat":
            c2 = sum(chs[-1] if x == -1 else chs[x + 1] for x in f)
            layers.append(YoloLayer(i, f, "Concat", c2))
        elif name == "Detect":
            in_ch = [chs[x + 1] for x in f]
            anc = args[1]
            if isinstance(anc, int):
                anc = [list(range(anc * 2))] * len(f)
            convs = [ConvSpec(f"{prefix}.m.{j}", None, c, no, 1, 1, 0, True, "none")
                     for j, c in enumerate(in_ch)]
            layers.append(YoloLayer(i, f, "Detect", no, convs,
                                    {"nc": args[0], "anchors": anc, "na": na, "no": nc + 5}))
        else:
            raise NotImplementedError(f"yolov5 module {m!r} is not supported by this backend")
        chs.append(c2)
    meta = {"nc": nc, "na": na, "no": nc + 5, "anchors": anchors}
    return layers, meta


# --------------------------------------------------------------------------
# heads
# --------------------------------------------------------------------------

@dataclass
class UpBlockSpec:
    """`double_conv_up_c3` (`basemodel.py:21-32`): C3 -> ConvT4x4s2 -> BN -> ReLU."""
    c3: C3Spec
    up: ConvSpec


def _up_block(prefix: str, in_ch: int, mid_ch: int, out_ch: int, act: str) -> UpBlockSpec:
    c3 = _c3(prefix + ".conv.0", in_ch + mid_ch, mid_ch, 1, True, act, BN_EPS_HEAD)
    up = ConvSpec(prefix + ".conv.1", prefix + ".conv.2", mid_ch, out_ch, 4, 2, 1, False,
                  "relu", BN_EPS_HEAD, transposed=True)
    return UpBlockSpec(c3, up)


@dataclass
class UnetSpec:
    down_conv1: C3Spec        # preceded by AvgPool2d(2), `basemodel.py:34-45`
    upconv0: UpBlockSpec
    upconv2: UpBlockSpec
    upconv3: UpBlockSpec
    upconv4: UpBlockSpec
    upconv5: UpBlockSpec
    upconv6: ConvSpec         # ConvT 64->1 + Sigmoid, `basemodel.py:58-61`


def unet_spec(act: str = "leaky") -> UnetSpec:
    """`UnetHead.__init__` (`basemodel.py:47-61`)."""
    return UnetSpec(
        down_conv1=_c3("down_conv1.conv", 512, 512, 1, True, act, BN_EPS_HEAD),
        upconv0=_up_block("upconv0", 0, 512, 256, act),
        upconv2=_up_block("upconv2", 256, 512, 256, act),
        upconv3=_up_block("upconv3", 0, 512, 256, act),
        upconv4=_up_block("upconv4", 128, 256, 128, act),
        upconv5=_up_block("upconv5", 64, 128, 64, act),
        upconv6=ConvSpec("upconv6.0", None, 64, 1, 4, 2, 1, False, "sigmoid",
                         BN_EPS_HEAD, transposed=True),
    )


@dataclass
class DBBranchSpec:
    conv3: ConvSpec     # 3x3 64->16 + BN + ReLU
    up1: ConvSpec       # ConvT 2x2 s2 16->16 + BN + ReLU
    up2: ConvSpec       # ConvT 2x2 s2 16->1 (+ sigmoid)


@dataclass
class DBSpec:
    upconv3: UpBlockSpec
    upconv4: UpBlockSpec
    conv: ConvSpec            # 1x1 128->64 (bias) + BN + ReLU
    binarize: DBBranchSpec
    thresh: DBBranchSpec


def db_spec(in_channels: int = 64, act: str = "leaky") -> DBSpec:
    """`DBHead.__init__` / `_init_thresh` (`basemodel.py:83-143`)."""
    q = in_channels // 4

    def branch(name: str, first_bias: bool) -> DBBranchSpec:
        return DBBranchSpec(
            ConvSpec(f"{name}.0", f"{name}.1", in_channels, q, 3, 1, 1, first_bias, "relu", BN_EPS_HEAD),
            ConvSpec(f"{name}.3", f"{name}.4", q, q, 2, 2, 0, True, "relu", BN_EPS_HEAD, transposed=True),
            ConvSpec(f"{name}.6", None, q, 1, 2, 2, 0, True, "sigmoid", BN_EPS_HEAD, transposed=True),
        )

    return DBSpec(
        upconv3=_up_block("upconv3", 0, 512, 256, act),
        upconv4=_up_block("upconv4", 128, 256, 128, act),
        conv=ConvSpec("conv.0", "conv.1", 128, in_channels, 1, 1, 0, True, "relu", BN_EPS_HEAD),
        binarize=branch("binarize", True),      # nn.Conv2d(.., 3, padding=1) default bias
        thresh=branch("thresh", False),         # `_init_thresh(bias=False)` on the 3x3 only
    )


# --------------------------------------------------------------------------
# enumeration helpers
# --------------------------------------------------------------------------

def iter_convs(obj):
    """Yield every ConvSpec reachable from a spec object, in definition order."""
    if isinstance(obj, ConvSpec):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from iter_convs(o)
    elif isinstance(obj, YoloLayer):
        if obj.spec is not None:
            yield from iter_convs(obj.spec)
    elif hasattr(obj, "__dataclass_fields__"):
        for name in obj.__dataclass_fields__:
            yield from iter_convs(getattr(obj, name))


def detect_strides(layers: Sequence[YoloLayer]) -> List[int]:
    """Stride of each Detect input level.  The reference measures it with a
    256x256 dry run (`yolo.py:85-88`); for the supported module set it is the
    product of conv strides / upsample factors along the path."""
    scale: Dict[int, float] = {}
    cur = 1.0
    out: List[int] = []
    for L in layers:
        src = L.f if isinstance(L.f, int) else L.f[0]
        base = cur if src == -1 else scale[src]
        if L.kind == "Conv":
            base = base * L.spec.s
        elif L.kind == "Upsample":
            base = base / 2
        elif L.kind == "Detect":
            out = [int(scale[x]) for x in L.f]
        scale[L.i] = base
        cur = base
    return out
