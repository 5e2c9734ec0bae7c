"""-m gpu: edge cases of the hot path -- empty / saturated maps, smallest and largest page
sizes, non-square inputs, batch remainders (reference behaviour: empty results are legal,
inference.py:166-167; H, W multiples of 64)."""
import numpy as np
import pytest
import torch

from conftest import checkpoint, pkg
from oracle import gen_golden
from oracle import postproc_ref as R
from oracle.net_ref import OracleNet
from test_post_host import blocks_equal

pytestmark = pytest.mark.gpu


def _det(size):
    from test_gpu_e2e import detector
    return detector(size)


def test_no_detections_gives_empty_results():
    size = 256
    det = _det(size)
    page = np.full((size, size, 3), 255, np.uint8)
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")       # noqa: E731
    m, refined, blks = det.tail_batch([page], z(1, 1008, 7), z(1, size, size, dt=torch.uint8), z(1, size, size),
                                      z(1, size, size, dt=torch.uint8), keep_undetected_mask=True)[0]
    assert blks == [] and refined.sum() == 0 and m.sum() == 0
    ref = R.detector_tail(page, np.zeros((1, 1008, 7), np.float32), np.zeros((1, 1, size, size), np.float32),
                          np.zeros((1, 2, size, size), np.float32), input_size=(size, size), keep_undetected_mask=True)
    assert ref[2] == [] and ref[1].sum() == 0


def test_saturated_maps_single_component():
    """Everything is text: one component touching all borders, no holes; one block covering the page."""
    size = 256
    det = _det(size)
    page = np.random.RandomState(0).randint(0, 256, (size, size, 3)).astype(np.uint8)
    blks = np.zeros((1, 1008, 7), np.float32)
    blks[0, 0] = [size / 2, size / 2, size - 20, size - 20, 0.95, 0.1, 0.9]
    mask_u8 = np.full((size, size), 230, np.uint8)
    prob = np.full((size, size), 0.9, np.float32)
    got = det.tail_batch([page], torch.from_numpy(blks).cuda(), torch.from_numpy(mask_u8)[None].cuda(),
                         torch.from_numpy(prob)[None].cuda(), torch.ones(1, size, size, dtype=torch.uint8, device="cuda"),
                         keep_undetected_mask=True)[0]
    ref = R.detector_tail(page, blks, ((mask_u8.astype(np.float32) + 0.5) / 255)[None, None],
                          np.stack([prob, np.zeros_like(prob)])[None], input_size=(size, size), keep_undetected_mask=True)
    np.testing.assert_array_equal(got[0], ref[0])
    blocks_equal(got[2], ref[2])
    np.testing.assert_array_equal(got[1], ref[1])


@pytest.mark.parametrize("shape", [(1, 64, 64), (1, 640, 1024), (3, 128, 64), (1, 1536, 1536)])
def test_network_sizes_against_oracle(shape):
    """Smallest legal input, non-square, odd batch, and the largest bucket of BASELINE config 5."""
    ck = checkpoint(0)
    x = gen_golden.make_input(31, shape)
    ob, om, ol = OracleNet(ck)(x)
    p = pkg()
    for prec, tol in (("fp32", 3e-5), ("fp32s", 3e-5), ("fp16", 3e-2)):   # all three engines, the fp32-level ones at the same bar
        be = p.backend.HipTextDetBackend(ck, device="cuda", precision=prec)
        blks, mask, lines = be(x.cuda())
        torch.cuda.synchronize()
        assert blks.shape == ob.shape
        assert float((mask.cpu() - om).abs().max()) < tol
        assert float((lines.cpu() - ol).abs().max()) < tol
        assert float((blks.cpu()[..., 4:] - ob[..., 4:]).abs().max()) < tol
        del be


@pytest.mark.parametrize("shape", [(1, 64, 64), (3, 128, 64), (2, 320, 448)])
def test_halo_kernel_forced_onto_small_maps_matches_oracle(shape):
    """The halo-tile conv kernel normally takes only maps with >= 1024 patches; forced onto tiny
    ones (`ctd_tuning_set("halo_min_patches", 1)`) every 16x16 patch is partial: 2x2 ... 14x10 pixel maps, patches
    hanging over the right / bottom edge, ConvTranspose phases on 2x2 inputs."""
    ck = checkpoint(0)
    x = gen_golden.make_input(33, shape)
    ob, om, ol = OracleNet(ck)(x)
    p = pkg()
    be = p.backend.HipTextDetBackend(ck, device="cuda", precision="fp16")
    ref = [t.clone() for t in be(x.cuda())]
    L = p._lib
    L.check(L.lib().ctd_tuning_set(b"halo_min_patches", 1), "ctd_tuning_set")
    try:
        got = [t.clone() for t in be(x.cuda())]
        torch.cuda.synchronize()
    finally:
        L.check(L.lib().ctd_tuning_set(b"halo_min_patches", 1024), "ctd_tuning_set")
    assert not all(torch.equal(g, r) for g, r in zip(got, ref)), "the forced dispatch did not change any kernel"
    # the fp16 golden tolerances of tests/test_gpu_net.py (2e-2 max, 2e-3 mean), not a looser bar for the forced dispatch
    for g, o in ((got[1].cpu(), om), (got[2].cpu(), ol)):
        d = (g - o).abs()
        assert float(d.max()) < 2e-2 and float(d.mean()) < 2e-3, (float(d.max()), float(d.mean()))
    assert float((got[0].cpu()[..., 4:] - ob[..., 4:]).abs().max()) < 2e-2
    # same arithmetic up to the summation order of the K walk
    assert float((got[1] - ref[1]).abs().max()) < 5e-3 and float((got[2] - ref[2]).abs().max()) < 5e-3


def test_replanning_between_sizes_keeps_results():
    """A mixed-size stream re-plans the arena; going back to an earlier size reproduces its result."""
    be = pkg().backend.HipTextDetBackend(checkpoint(0), device="cuda", precision="fp16")
    xa = gen_golden.make_input(41, (2, 256, 256)).cuda()
    xb = gen_golden.make_input(42, (1, 512, 384)).cuda()
    a1 = [t.clone() for t in be(xa)]
    b1 = [t.clone() for t in be(xb)]
    a2 = be(xa)
    torch.cuda.synchronize()
    for u, v in zip(a1, a2):
        assert torch.equal(u, v)
    assert b1[1].shape == (1, 1, 512, 384)


def test_large_batch_is_split_transparently():
    """B above the 2 GiB-per-tensor limit of the MFMA kernel's 32-bit offsets runs as sub-batches."""
    be = pkg().backend.HipTextDetBackend(checkpoint(0), device="cuda", precision="fp16")
    H = W = 256
    max_b = (2 ** 31 - 1) // (H * W * 40)
    B = max_b + 3
    pages = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
    blks, mask, lines = be.forward_u8(pages)
    assert mask.shape[0] == B and be.mask_u8.shape[0] == B
    one = be.forward_u8(pages[B - 1: B].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(one[1][0], mask[B - 1])


def _tune(key, value):
    L = pkg()._lib
    L.check(L.lib().ctd_tuning_set(key, value), "ctd_tuning_set")


@pytest.mark.parametrize("shape,u8", [((1, 64, 64), False), ((3, 128, 64), True), ((2, 320, 448), False),
                                      ((2, 1024, 1024), True), ((1, 1536, 1536), True)])
def test_fused_blocks_equal_the_layer_per_launch_program_bit_for_bit(shape, u8):
    """The multi-layer kernels of the fp16 engine (stem + layer 1, the 32-channel C3 block, SPPF's three pools, bit 8:
    bottleneck + cv3 of the 64 / 128-channel C3 blocks of backbone, neck and heads -- kernels_c3b.hip; bit 16: a 128-channel
    ConvTranspose and the 1x1 conv that is its only consumer, bit 32: the last 64-channel ConvTranspose and the tap
    products of the 64 -> 1 one behind it -- kernels_halo3.hip;
    `ctd_tuning_set("fuse", mask)`) do the arithmetic of the launches they replace in the same order: every output
    of the network must be IDENTICAL with and without them -- interior and border patches, float and uint8 input,
    maps smaller than one patch (the C3 kernels are forced onto them with c3_min_patches = c3b_min_patches = 1), and
    for bit 8 both K walks of the 3x3 it absorbs (the halo kernel's, forced onto every map by halo_min_patches = 1,
    and the implicit GEMM's).  (This test is what found that the compiler rounded SiLU outputs once or twice depending
    on the kernel: ctd_common.h ctd_act_fast.)"""
    be = pkg().backend.HipTextDetBackend(checkpoint(0), device="cuda", precision="fp16")
    if u8:
        x = torch.randint(0, 256, (shape[0], shape[1], shape[2], 3), dtype=torch.uint8,
                          generator=torch.Generator().manual_seed(5)).cuda()
        run = lambda: [t.clone() for t in be.forward_u8(x)] + [be.mask_u8.clone(), be.bitmap.clone()]   # noqa: E731
    else:
        x = gen_golden.make_input(77, shape).cuda()
        run = lambda: [t.clone() for t in be(x)] + [be.mask_u8.clone(), be.bitmap.clone()]              # noqa: E731
    try:
        _tune(b"fuse", 0)
        ref = run()
        _tune(b"c3_min_patches", 1)
        _tune(b"c3b_min_patches", 1)
        outs = {}
        for mask in (1, 2, 4, 6, 7, 8, 15, 16, 31, 32, 63):
            _tune(b"fuse", mask)
            outs[mask] = run()
        _tune(b"halo_min_patches", 1)                   # the 3x3s (and ConvT phases) on the halo kernel everywhere
        _tune(b"fuse", 0)
        ref_h = run()
        _tune(b"fuse", 8)
        outs["8 + halo"] = run()
        _tune(b"c3b_max_ch", 64)
        outs["8 + halo, 64 only"] = run()
        # bit 16 (a 128-channel ConvTranspose + its single 1x1 consumer in one launch of the big-tile kernel) needs that
        # kernel: lift its grid threshold so that also the small shapes go through it (maps that are multiples of 16)
        _tune(b"c3b_max_ch", 128)
        _tune(b"halo3_min_blocks", 1)
        _tune(b"fuse", 0)
        ref_3 = run()
        for mask in (16, 31, 32, 63):
            _tune(b"fuse", mask)
            outs[f"{mask} + halo3"] = run()
        torch.cuda.synchronize()
    finally:
        _tune(b"fuse", 63)
        _tune(b"halo3_min_blocks", 1024)
        _tune(b"c3_min_patches", 1024)
        _tune(b"c3b_min_patches", 1024)
        _tune(b"c3b_max_ch", 128)
        _tune(b"halo_min_patches", 1024)
    for mask, got in outs.items():
        base = ref if not isinstance(mask, str) else (ref_3 if "halo3" in mask else ref_h)
        for i, (g, r) in enumerate(zip(got, base)):
            assert torch.equal(g, r), f"fuse mask {mask}: output {i} differs from the unfused program " \
                                      f"(max |d| {float((g.float() - r.float()).abs().max()):.3g})"


@pytest.mark.parametrize("shape", [(2, 512, 512), (1, 1024, 768), (3, 256, 512)])
def test_big_tile_convt_kernels_reproduce_the_256x128_kernel_bit_for_bit(shape):
    """The ConvTranspose layers on kernels_halo3.hip (256 x 128 tiles, four waves, two blocks per CU) and on
    kernels_halo2.hip (selftest build only) walk K and issue their MFMAs per accumulator in the order of kernels_halo.hip:
    with the grid threshold lifted (`halo3_min_blocks` = 1: also maps whose ConvT inputs are 16x16 ... 64x48, i.e. every
    tile at an image border) every output of the network equals the output without it, bit for bit."""
    ck = checkpoint(0)
    x = gen_golden.make_input(51, shape).cuda()
    be = pkg().backend.HipTextDetBackend(ck, device="cuda", precision="fp16")
    _tune(b"halo3", 0)
    _tune(b"halo_min_patches", 1)                       # the reference side: the 256 x 128 halo kernel on every ConvT layer
    try:
        ref = [t.clone() for t in be(x)]
        ref_side = (be.mask_u8.clone(), be.bitmap.clone())
        _tune(b"halo3", 1)
        _tune(b"halo3_min_blocks", 1)
        got = [t.clone() for t in be(x)]
        got_side = (be.mask_u8.clone(), be.bitmap.clone())
        torch.cuda.synchronize()
    finally:
        _tune(b"halo3", 1)
        _tune(b"halo3_min_blocks", 1024)
        _tune(b"halo_min_patches", 1024)
    for g, r in zip(got + list(got_side), ref + list(ref_side)):
        assert torch.equal(g, r)
