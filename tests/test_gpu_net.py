"""-m gpu: parity of the HIP forward (through the C ABI) with the oracle and
with the golden vectors produced by the reference's own modules.

Tolerances (fp is not bit-exact across different summation orders):
  fp32 engine: sigmoid outputs within 2e-5 abs of the reference, Detect rows
               within 1e-4 relative; u8 mask may differ by one level on <0.1 % px.
  fp16 engine: fp16 operands/activations with fp32 accumulation vs an fp32
               reference: sigmoid outputs within 2e-2 abs, mean abs < 2e-3;
               thresholded (binary) maps IoU >= 0.99 on the synthetic-weight net
               whose outputs sit near 0.5 (worst case for a threshold).
"""
import numpy as np
import pytest
import torch

from conftest import checkpoint, load_golden, pkg
from oracle import gen_golden
from oracle.net_ref import OracleNet

pytestmark = pytest.mark.gpu

_BACKENDS = {}


def backend(prec: str, seed: int = 0):
    key = (prec, seed)
    if key not in _BACKENDS:
        _BACKENDS[key] = pkg().backend.HipTextDetBackend(checkpoint(seed), device="cuda", precision=prec)
    return _BACKENDS[key]


def iou(a, b):
    a, b = a.astype(bool), b.astype(bool)
    u = (a | b).sum()
    return 1.0 if u == 0 else (a & b).sum() / u


@pytest.mark.parametrize("prec", ["fp32", "fp32s"])
@pytest.mark.parametrize("name", sorted(gen_golden.SMALL_CASES))
def test_fp32_engine_matches_reference_golden(name, prec):
    """Both fp32-level engines against the reference's own modules: "fp32" (f32-operand MFMA) and "fp32s" (the same
    fp32 tensors, products from split fp16 operands on the fp16 MFMA) -- same tolerances."""
    g = load_golden(name)
    be = backend(prec, int(g["wseed"]))
    x = gen_golden.make_input(int(g["iseed"]), tuple(int(v) for v in g["shape"]))
    blks, mask, lines = be(x.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(mask.cpu().numpy(), g["mask"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(lines.cpu().numpy(), g["lines"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(blks.cpu().numpy(), g["blks"], rtol=1e-4, atol=2e-3)
    ref_u8 = (g["mask"][:, 0] * 255).astype(np.uint8)
    got_u8 = be.mask_u8.cpu().numpy()
    diff = np.abs(ref_u8.astype(int) - got_u8.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    assert ((g["lines"][:, 0] > 0.3) != be.bitmap.cpu().numpy().astype(bool)).mean() < 1e-3


@pytest.mark.parametrize("name", sorted(gen_golden.SMALL_CASES))
def test_fp16_engine_close_to_reference_golden(name):
    g = load_golden(name)
    be = backend("fp16", int(g["wseed"]))
    x = gen_golden.make_input(int(g["iseed"]), tuple(int(v) for v in g["shape"]))
    blks, mask, lines = be(x.cuda())
    torch.cuda.synchronize()
    for got, ref in ((mask, g["mask"]), (lines, g["lines"])):
        d = np.abs(got.cpu().numpy() - ref)
        assert d.max() < 2e-2 and d.mean() < 2e-3, (d.max(), d.mean())
    # objectness / class scores are probabilities: same absolute tolerance
    b = blks.cpu().numpy()
    assert np.abs(b[..., 4:] - g["blks"][..., 4:]).max() < 2e-2
    # boxes of confident rows: relative 1 % (fp16 logits -> exp)
    conf = g["blks"][..., 4] > 0.25
    if conf.any():
        rel = np.abs(b[conf][:, :4] - g["blks"][conf][:, :4]) / (np.abs(g["blks"][conf][:, :4]) + 8.0)
        assert rel.max() < 2e-2, rel.max()
    assert iou(mask.cpu().numpy() > 0.5, g["mask"] > 0.5) > 0.98
    assert iou(be.bitmap.cpu().numpy(), g["lines"][:, 0] > 0.3) > 0.98


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "fp16"])
def test_full_size_page_vs_reference_summary(prec):
    """1024x1024 text-like page: tile means / top Detect rows of the reference."""
    g = load_golden("net_full_summary")
    p = pkg()
    page = p.synth.text_like_page((int(g["size"]),) * 2, int(g["pseed"]))
    x = gen_golden.page_to_input(page)
    assert float(x.double().sum()) == pytest.approx(float(g["input_sum"]), abs=1e-6)
    be = backend(prec, int(g["wseed"]))
    blks, mask, lines = be(x.cuda())
    torch.cuda.synchronize()
    exact = prec != "fp16"
    tol = 1e-5 if exact else 3e-3
    np.testing.assert_allclose(gen_golden.tile_means(mask.cpu()), g["mask_tiles"], rtol=0, atol=tol)
    np.testing.assert_allclose(gen_golden.tile_means(lines.cpu()), g["lines_tiles"], rtol=0, atol=tol)
    top = blks[0].cpu().numpy()[g["top_rows"]]
    if exact:
        np.testing.assert_allclose(top, g["top_blks"], rtol=2e-4, atol=5e-3)
    else:
        assert np.abs(top[:, 4:] - g["top_blks"][:, 4:]).max() < 3e-2
    # u8 histogram of the mask: the fused quantiser saw (almost) the same values
    hist = np.bincount(be.mask_u8[0].cpu().numpy().ravel(), minlength=256)
    moved = np.abs(hist - g["mask_u8_hist"]).sum() / hist.sum()
    assert moved < (2e-3 if exact else 0.2)
    # the same page through the u8 entry point (fused /255) gives the same result
    pages = torch.from_numpy(page)[None].cuda()
    blks2, mask2, lines2 = be.forward_u8(pages)
    torch.cuda.synchronize()
    assert torch.allclose(mask2, mask, atol=1e-6 if exact else 2e-3)


@pytest.mark.parametrize("prec", ["fp32", "fp32s", "fp16"])
def test_determinism_and_batch_independence(prec):
    be = backend(prec)
    x = gen_golden.make_input(11, (3, 128, 192)).cuda()
    a = [t.clone() for t in be(x)]
    b = [t.clone() for t in be(x)]
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v), "two runs on the same input differ"
    c = be(x[1:2].contiguous())
    torch.cuda.synchronize()
    for u, v in zip(a, c):
        assert torch.equal(u[1:2], v), "page result depends on its batch neighbours"


def test_sub_batches_write_into_one_output_set():
    """Batches whose largest activation would pass the 32-bit offset range run as consecutive sub-batches
    (B > 51 at 1024x1024); here the budget is shrunk so that B=5 at 128x192 splits into 2+2+1."""
    be = backend("fp16")
    x = gen_golden.make_input(12, (5, 128, 192)).cuda()
    want = [t.clone() for t in be(x)]
    want_side = (be.mask_u8.clone(), be.bitmap.clone())
    be._offset_budget = 2 * 128 * 192 * 40
    got = be(x)
    torch.cuda.synchronize()
    for u, v in zip(want, got):
        assert u.shape == v.shape and torch.equal(u, v)
    assert torch.equal(be.mask_u8, want_side[0]) and torch.equal(be.bitmap, want_side[1])


def test_fp16_vs_oracle_mid_size_statistics():
    """512x512, B=2: error statistics of the MFMA path against the fp32 oracle."""
    ck = checkpoint(0)
    x = gen_golden.make_input(21, (2, 512, 512))
    ob, om, ol = OracleNet(ck)(x)
    be = backend("fp16")
    blks, mask, lines = be(x.cuda())
    torch.cuda.synchronize()
    dm = (mask.cpu() - om).abs()
    dl = (lines.cpu() - ol).abs()
    print(f"fp16 vs fp32-oracle @512: mask max {dm.max():.3e} mean {dm.mean():.3e}; "
          f"lines max {dl.max():.3e} mean {dl.mean():.3e}")
    assert dm.max() < 3e-2 and dm.mean() < 2e-3 and dl.max() < 3e-2 and dl.mean() < 2e-3


def test_bad_shapes_raise():
    be = backend("fp16")
    with pytest.raises(ValueError):
        be(torch.zeros(1, 3, 100, 128, device="cuda"))
    with pytest.raises(ValueError):
        be(torch.zeros(1, 4, 128, 128, device="cuda"))


def test_hipgraph_replay_matches_eager():
    """The captured forward (one graph launch) reproduces the eager launches bit for bit."""
    be = backend("fp16")
    pages = torch.randint(0, 256, (2, 256, 320, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(5))
    eager = [t.clone() for t in be.forward_u8(pages.cuda())]
    m8 = be.mask_u8.clone()
    static_in, replay = be.capture(2, 256, 320, "u8")
    static_in.copy_(pages.cuda())
    out = replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, out):
        assert torch.equal(a, b)
    assert torch.equal(m8, be.mask_u8)
    static_in.copy_(torch.flip(pages, dims=[0]).cuda())           # new data, same graph
    out2 = [t.clone() for t in replay()]
    torch.cuda.synchronize()
    assert torch.equal(out2[1][0], eager[1][1]) and torch.equal(out2[1][1], eager[1][0])


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-3), ("fp32s", 2e-3), ("fp16", 2e-1)])
def test_db_step_function_matches_oracle(prec, tol):
    """N8: `DBHead.forward(step_eval=True)` = step_function(shrink, thresh), k = 50 (reference
    basemodel.py:121-122,159-160) through `ctd_db_step`.  k amplifies the map error 12.5x at the steepest point."""
    p = pkg()
    ck = checkpoint(0)
    be = p.backend.HipTextDetBackend(ck, device="cuda", precision=prec, step_eval=True)
    x = torch.rand(2, 3, 128, 192, generator=torch.Generator().manual_seed(5))
    blks, mask, step = be(x.cuda())
    torch.cuda.synchronize()
    _, _, ol = OracleNet(ck)(x)
    ref = OracleNet.step_function(ol, 50.0)
    assert step.shape == ref.shape == (2, 1, 128, 192)
    assert float((step.cpu() - ref).abs().max()) < tol
    assert float((step.cpu() - ref).abs().mean()) < tol / 20
    # the exact relation on the engine's OWN planes (the kernel itself, no network error)
    be2 = backend(prec)
    _, _, lines = be2(x.cuda())
    own = torch.reciprocal(1 + torch.exp(-50.0 * (lines[:, 0:1] - lines[:, 1:2])))
    torch.testing.assert_close(step, own, rtol=1e-5, atol=1e-6)
    assert torch.equal(be.bitmap.bool(), step[:, 0] > 0.3)


def test_checkpoint_file_and_onnx_contract_entry(tmp_path):
    """f-3: the reference's weight FILE (`torch.save` of the dict, basemodel.py:212) loaded by path, and the
    `TextDetBaseDNN`-style entry (uint8 HWC in, numpy out, tensors named images -> blk / seg / det,
    utils/export.py:43-44, basemodel.py:246-256)."""
    p = pkg()
    ck = checkpoint(0)
    path = str(tmp_path / "comictextdetector.pt")
    torch.save(ck, path)
    det = p.detector.TextDetector(path, input_size=256, device="cuda", half=False)
    page = p.synth.text_like_page((256, 256), 5, n_blocks=4)
    m0, r0, b0 = det(page)
    m1, r1, b1 = p.detector.TextDetector(ck, input_size=256, device="cuda", half=False)(page)
    np.testing.assert_array_equal(m0, m1)
    np.testing.assert_array_equal(r0, r1)
    assert len(b0) == len(b1)
    dnn = p.backend.HipTextDetDNN(256, path)
    assert (dnn.input_name, tuple(dnn.uoln)) == ("images", ("blk", "seg", "det"))
    im_in = page[:, :, ::-1].copy()                       # preprocess_img(..., to_tensor=False): RGB uint8 HWC
    blks, mask, lines = dnn(im_in)
    x = torch.from_numpy(im_in.transpose(2, 0, 1)[None].astype(np.float32) * np.float32(1 / 255.0))
    ob, om, ol = OracleNet(ck)(x)
    assert isinstance(blks, np.ndarray) and mask.shape == (1, 1, 256, 256) and lines.shape == (1, 2, 256, 256)
    np.testing.assert_allclose(mask, om.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(lines, ol.numpy(), rtol=0, atol=2e-5)
    named = dnn.forward_named(x.numpy())
    assert sorted(named) == ["blk", "det", "seg"]
    np.testing.assert_array_equal(named["seg"], mask)


def test_captured_forward_goes_stale_when_the_arena_grows():
    """A hipGraph captured from the forward holds the arena's addresses.  Graphs of several shapes coexist
    when the largest shape ran first; a later larger shape reallocates the arena and every older graph must
    refuse to replay (it used to read freed memory silently)."""
    p = pkg()
    be = p.backend.HipTextDetBackend(checkpoint(0), device="cuda", precision="fp16")
    x_big = torch.randint(0, 256, (2, 256, 256, 3), dtype=torch.uint8, device="cuda")
    x_small = torch.randint(0, 256, (1, 128, 192, 3), dtype=torch.uint8, device="cuda")
    in_big, replay_big = be.capture(2, 256, 256, "u8")            # largest first
    in_small, replay_small = be.capture(1, 128, 192, "u8")
    gen = be.arena_generation()
    for static_in, replay, x in ((in_big, replay_big, x_big), (in_small, replay_small, x_small), (in_big, replay_big, x_big)):
        static_in.copy_(x)
        blks, mask, lines = replay()
        torch.cuda.synchronize()
        eb, em, el = be.forward_u8(x)
        torch.cuda.synchronize()
        assert torch.equal(mask, em) and torch.equal(lines, el) and torch.equal(blks, eb)
        static_in.copy_(x)                      # the eager call re-planned; the graph must still be valid
        blks, mask, lines = replay()
        torch.cuda.synchronize()
        assert torch.equal(mask, em)
    assert be.arena_generation() == gen
    be.forward_u8(torch.zeros((4, 512, 512, 3), dtype=torch.uint8, device="cuda"))   # needs a larger arena
    torch.cuda.synchronize()
    assert be.arena_generation() != gen
    with pytest.raises(p._lib.CtdError):
        replay_big()


def test_split_engine_tracks_the_f32_engine_at_full_size():
    """fp32s vs fp32 on a 1024x1024 page, B=2: the two engines run the SAME program on the same fp32 tensors and
    differ only in how a product is formed (three fp16 MFMAs on split operands vs one f32 MFMA), so their maps must
    agree far inside the fp32 engine's own tolerance against the reference (2e-5), and their u8 / bitmap side
    outputs on all but a handful of pixels."""
    p = pkg()
    pages = torch.from_numpy(np.stack([p.synth.text_like_page((1024, 1024), s) for s in (1, 2)])).cuda()
    a = backend("fp32")
    ba, ma, la = [t.clone() for t in a.forward_u8(pages)]
    mu_a, bm_a = a.mask_u8.clone(), a.bitmap.clone()
    b = backend("fp32s")
    bb, mb, lb = b.forward_u8(pages)
    torch.cuda.synchronize()
    dm, dl = float((ma - mb).abs().max()), float((la - lb).abs().max())
    du = float((mu_a != b.mask_u8).float().mean())
    dbm = float((bm_a != b.bitmap).float().mean())
    print(f"fp32s vs fp32 @1024 B=2: max|dmask| {dm:.2e}, max|dlines| {dl:.2e}, u8 mask differs on {du:.2e}, bitmap on {dbm:.2e}")
    assert torch.isfinite(mb).all() and torch.isfinite(lb).all() and torch.isfinite(bb).all()
    assert dm < 1e-5 and dl < 1e-5
    assert du < 1e-3 and dbm < 1e-4
    torch.testing.assert_close(bb, ba, rtol=1e-4, atol=2e-3)


def test_split_engine_layouts_and_kernels_agree():
    """The fp32s engine with its round-3 machinery switched off piece by piece (`ctd_tuning_set`): fp32 tensors instead of
    split-plane ones, the 128-pixel kernel instead of the haloed-patch kernel, INPUT + generic kernel instead of the first
    layer from the page.  Every variant forms the same products from the same (hi, lo) operands -- only the summation order
    of a K loop and the 22-bit residual inputs differ -- so the maps agree to a few 1e-7, far inside the tolerance against
    the oracle, and the first-layer switch alone changes nothing at all."""
    p = pkg()
    L = p._lib.lib()
    pages = torch.from_numpy(np.stack([p.synth.text_like_page((512, 512), s) for s in (3, 4)])).cuda()

    def run(**knobs):
        for k, v in knobs.items():
            assert L.ctd_tuning_set(k.encode(), v) == p._lib.OK
        try:
            be = p.backend.HipTextDetBackend(checkpoint(0), device="cuda", precision="fp32s")
            blks, mask, lines = [t.clone() for t in be.forward_u8(pages)]
            torch.cuda.synchronize()
            return blks, mask, lines, be.mask_u8.clone(), be.bitmap.clone()
        finally:
            for k in knobs:
                assert L.ctd_tuning_set(k.encode(), 1) == p._lib.OK

    ref = run()
    assert all(torch.isfinite(t).all() for t in ref[:3])
    same = run(split_stem=0)
    for a, b in zip(ref, same):
        assert torch.equal(a, b)                      # the first layer from the page is bit-identical to INPUT + generic kernel
    for knobs in (dict(split_halo=0), dict(split_planes=0), dict(split_planes=0, split_halo=0, split_stem=0)):
        got = run(**knobs)
        dm, dl = float((ref[1] - got[1]).abs().max()), float((ref[2] - got[2]).abs().max())
        du = float((ref[3] != got[3]).float().mean())
        print(f"fp32s {knobs}: max|dmask| {dm:.2e}, max|dlines| {dl:.2e}, u8 mask differs on {du:.2e}")
        assert dm < 3e-6 and dl < 3e-6 and du < 2e-4
        torch.testing.assert_close(got[0], ref[0], rtol=1e-4, atol=1e-3)
