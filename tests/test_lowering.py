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
