"""Host logic: the lowering of a checkpoint to the engine's op program
(graph.py) is checked on CPU by interpreting the program with torch
(oracle/program_interp.py) and comparing with the oracle network."""
import numpy as np
import pytest
import torch

from conftest import checkpoint, pkg
from oracle.net_ref import OracleNet
from oracle.program_interp import run_program


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_program_semantics_match_oracle(prec):
    p = pkg()
    L = p._lib
    ck = checkpoint(0)
    prog = p.graph.lower(ck, L.PREC_F16 if prec == "fp16" else L.PREC_F32)
    x = torch.rand(2, 3, 128, 192, generator=torch.Generator().manual_seed(1))
    ob, om, ol = OracleNet(ck)(x)
    out = run_program(prog, x)
    np.testing.assert_allclose(out["mask"].numpy(), om.numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(out["lines"].numpy(), ol.numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(out["blks"].numpy(), ob.numpy(), rtol=1e-4, atol=1e-3)
    # fused side outputs (reference inference.py:96-99, db_utils.py:71-72)
    assert ((om[:, 0] * 255).to(torch.uint8) != out["mask_u8"]).float().mean() < 1e-3
    assert ((ol[:, 0] > 0.3).to(torch.uint8) != out["bitmap"]).float().mean() < 1e-3


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_detector_outputs_lowering_drops_only_the_threshold_branch(prec):
    """`lower(db_thresh=False)` (what `TextDetector` builds): `lines_map` has the shrink map only and every other
    output is what the full program gives -- `SegDetectorRepresenter` reads `pred[:, 0]` (reference db_utils.py:63)."""
    p = pkg()
    L = p._lib
    ck = checkpoint(0)
    pr = L.PREC_F16 if prec == "fp16" else L.PREC_F32
    full, det = p.graph.lower(ck, pr), p.graph.lower(ck, pr, db_thresh=False)
    assert full.meta["line_planes"] == 2 and det.meta["line_planes"] == 1
    assert len(det.ops) == len(full.ops)                             # same ops, the fused DB ones are narrower
    assert not any("thresh" in o["name"] for o in det.ops)
    x = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(2))
    a, b = run_program(full, x), run_program(det, x)
    assert b["lines"].shape[1] == 1
    assert torch.equal(a["lines"][:, 0], b["lines"][:, 0])
    for k in ("blks", "mask", "mask_u8", "bitmap"):
        assert torch.equal(a[k], b[k]), k


def test_program_structure():
    p = pkg()
    L = p._lib
    prog = p.graph.lower(checkpoint(0), L.PREC_F16)
    kinds = [o["kind"] for o in prog.ops]
    assert kinds[0] == L.OP_STEM and L.OP_SEG_FINAL in kinds and L.OP_DB_UP in kinds
    assert kinds.count(L.OP_DETECT) == 3 and kinds.count(L.OP_CONVT) == 7
    # every conv the MFMA kernel should take has channel counts that are multiples of 32
    for o in prog.ops:
        if o["kind"] == L.OP_CONV:
            assert o["src0_c"] % 32 == 0 and (o["src1"] < 0 or o["src1_c"] % 32 == 0), o["name"]
    T, O, blob = p.graph.to_ctypes(prog)
    assert 23.3e6 < blob.size < 23.5e6
    assert prog.det_levels[0]["row_unit"] == 0 and prog.det_levels[1]["row_unit"] == 3 * 64
    assert sum(d["na"] * (64 // d["stride"]) ** 2 for d in prog.det_levels) * 256 == 64512   # 1024x1024


def test_parse_cfg_matches_survey_shapes():
    a = pkg().arch
    layers, meta = a.parse_yolo_cfg(a.YOLOV5S_CFG)
    assert [L.c2 for L in layers[:10]] == [32, 64, 64, 128, 128, 256, 256, 512, 512, 512]
    assert a.detect_strides(layers) == [8, 16, 32]
    assert len(layers[4].spec.m) == 2 and len(layers[6].spec.m) == 3     # depth 0.33: 6->2, 9->3
    n_convs = len(list(a.iter_convs(layers))) + len(list(a.iter_convs(a.unet_spec()))) + \
        len(list(a.iter_convs(a.db_spec())))
    assert n_convs == 115


def test_unsupported_module_raises():
    import copy
    a = pkg().arch
    cfg = copy.deepcopy(a.YOLOV5S_CFG)
    cfg["backbone"][2] = [-1, 3, "BottleneckCSP", [128]]
    with pytest.raises(NotImplementedError):
        a.parse_yolo_cfg(cfg)


def test_program_contains_the_patterns_the_engine_fuses():
    """engine.hip replaces three op patterns of the fp16 program by multi-layer kernels at plan time (stem + 3x3/s2 conv
    on a private stem output; a C3 block with 32 hidden channels and one bottleneck whose intermediates are private; three
    chained stride-1 max pools over the slots of one tensor).  A lowering change that breaks a pattern would silently
    fall back to one launch per layer -- this pins that the yolov5s program still contains exactly one of each."""
    import importlib
    p = importlib.import_module("comic-text-detector_amd")
    G = importlib.import_module("comic-text-detector_amd.graph")
    L = importlib.import_module("comic-text-detector_amd._lib")
    prog = G.lower(p.synth.make_checkpoint(0), L.PREC_F16)
    ops, T = prog.ops, prog.tensors
    first, last = {}, {}
    for i, o in enumerate(ops):
        reads = [o["src0"], o["src1"], o["res"]] if o["kind"] in (L.OP_CONV, L.OP_CONVT) else [o["src0"]]
        writes = [o["dst"]] if o["kind"] in (L.OP_INPUT, L.OP_STEM, L.OP_CONV, L.OP_CONVT, L.OP_MAXPOOL, L.OP_AVGPOOL2) else []
        for t in reads + writes:
            if t >= 0:
                last[t] = i
        for t in writes:
            first.setdefault(t, i)
    private = lambda t, a, b: first[t] == a and last[t] == b       # noqa: E731
    stem2 = [i for i in range(len(ops) - 1)
             if ops[i]["kind"] == L.OP_STEM and ops[i]["cout"] == 32 and ops[i + 1]["kind"] == L.OP_CONV
             and (ops[i + 1]["k"], ops[i + 1]["stride"], ops[i + 1]["pad"], ops[i + 1]["cout"]) == (3, 2, 1, 64)
             and ops[i + 1]["src0"] == ops[i]["dst"] and ops[i + 1]["src1"] < 0 and private(ops[i]["dst"], i, i + 1)]
    c3 = []
    for i in range(len(ops) - 3):
        a, b, c, d = ops[i:i + 4]
        if not all(x["kind"] == L.OP_CONV for x in (a, b, c, d)):
            continue
        Y, Tt = a["dst"], b["dst"]
        if (a["k"] == 1 and a["cout"] == 64 and a["res"] < 0 and T[Y][0] == 64 and b["k"] == 1 and b["src0"] == Y
                and b["src0_c"] == 32 and b["cout"] == 32 and T[Tt][0] == 32 and c["k"] == 3 and c["stride"] == 1
                and c["src0"] == Tt and c["dst"] == Y and c["res"] == Y and c["cout"] == 32 and d["k"] == 1
                and d["src0"] == Y and d["src0_c"] == 64 and d["cout"] == 64 and len({x["act"] for x in (a, b, c, d)}) == 1
                and private(Y, i, i + 3) and private(Tt, i + 1, i + 2)):
            c3.append(a["name"])
    pools = [i for i in range(len(ops) - 2)
             if all(ops[i + j]["kind"] == L.OP_MAXPOOL and ops[i + j]["src0"] == ops[i]["src0"] == ops[i + j]["dst"]
                    and ops[i + j]["k"] == ops[i]["k"] and ops[i + j]["dst_coff"] == ops[i + j]["src0_coff"] + ops[i]["src0_c"]
                    for j in range(3))
             and ops[i + 1]["src0_coff"] == ops[i]["dst_coff"] and ops[i + 2]["src0_coff"] == ops[i + 1]["dst_coff"]]
    assert stem2 == [0]
    assert c3 == ["model.2.cv1+cv2"]
    assert len(pools) == 1 and ops[pools[0]]["name"].endswith("model.9.m0")
