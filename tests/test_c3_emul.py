"""CPU check of the fused C3 kernel's data flow (kernels_c3.hip) through its lane-level emulation
(tests/c3_emul.py): swizzled LDS-DMA placement, fragment addressing, zero padding of t, in-place shortcut, region
aliasing.  The GPU side (bit-identity with the four launches) is tests/test_gpu_edge.py + ctd_selftest."""
import numpy as np
import pytest

import c3_emul as E


@pytest.mark.parametrize("H,W,cin,kind", [(9, 17, 64, "silu"), (8, 16, 96, "leaky")])
def test_fused_c3_block_emulation_matches_numpy(H, W, cin, kind):
    rs = np.random.RandomState(H * 100 + W)
    r16 = lambda *s, sc=1.0: (rs.standard_normal(s) * sc).astype(np.float16)     # noqa: E731
    x = r16(1, H, W, cin)
    W12, Wm1, Wm2, Wc3 = r16(64, cin, sc=0.15), r16(32, 32, sc=0.2), r16(32, 288, sc=0.08), r16(64, 64, sc=0.15)
    b12, bm1, bm2, bc3 = (rs.standard_normal(n).astype(np.float32) * 0.3 for n in (64, 32, 32, 64))
    ref = E.reference(x[0], W12, Wm1, Wm2, Wc3, b12, bm1, bm2, bc3, kind)
    packed = (E.pack_tiled(W12, 64), E.pack_tiled(Wm1, 32), E.pack_tiled(Wm2, 32), E.pack_tiled(Wc3, 64))
    got = np.full((H, W, 64), np.nan)
    for tpy in range((H + E.TH - 1) // E.TH):
        for tpx in range((W + E.TW - 1) // E.TW):
            out = E.Block(x, *packed, b12, bm1, bm2, bc3, kind, 0, tpy, tpx).run()
            for (oy, ox, cch), v in out.items():
                assert np.isnan(got[oy, ox, cch * 8]), "pixel written twice"
                got[oy, ox, cch * 8:cch * 8 + 8] = v
    assert not np.isnan(got).any(), "output pixels missing / uninitialised LDS reached the output"
    err = np.abs(got - ref)
    assert float(err.max()) <= 2e-2 * (1 + float(np.abs(ref).max())), float(err.max())
    assert float((err > 4e-3 * (1 + np.abs(ref))).mean()) < 1e-3
