"""CPU check of the fused stem + layer-1 kernel's data flow (kernels_stem2.hip) through its lane-level emulation
(tests/stem2_emul.py): both input-staging paths, the stem's MFMA fragment addressing, zeroed out-of-map stem pixels,
the split tap tiles, the aliased LDS regions.  The GPU side (bit-identity with the two launches) is
tests/test_gpu_edge.py + ctd_selftest."""
import numpy as np
import pytest

import c3_emul as C
import stem2_emul as E


@pytest.mark.parametrize("H,W,u8in,kind", [(64, 128, True, "silu"), (32, 64, False, "leaky"), (160, 192, True, "relu")])
def test_fused_stem_layer1_emulation_matches_numpy(H, W, u8in, kind):
    rs = np.random.RandomState(H + W)
    img = rs.randint(0, 256, (1, H, W, 3)).astype(np.uint8) if u8in else rs.rand(1, 3, H, W).astype(np.float32)
    W0 = (rs.standard_normal((32, 3, 6, 6)) * 0.15).astype(np.float32)
    lg1 = (rs.standard_normal((64, 288)) * 0.08).astype(np.float16)
    b0, b1 = rs.standard_normal(32).astype(np.float32) * 0.3, rs.standard_normal(64).astype(np.float32) * 0.3
    ref = E.reference(img[0], u8in, W0, b0, lg1, b1, kind)
    wfrag, w1 = E.stem_pack_weights(W0), C.pack_tiled(lg1, 64)
    Ho, Wo = H // 4, W // 4
    got = np.full((Ho, Wo, 64), np.nan)
    ny, nx = (Ho + E.TH - 1) // E.TH, (Wo + E.TW - 1) // E.TW
    tiles = [(ty, tx) for ty in range(ny) for tx in range(nx)]
    if len(tiles) > 6:                      # the big case: corners, edges and the interior (dword path) tiles only
        tiles = [(0, 0), (ny - 1, nx - 1), (0, 1), (1, 0), (1, 1), (ny - 2, 1)]
    for tpy, tpx in tiles:
        out = E.run_block(img, u8in, wfrag, b0, w1, b1, kind, 0, tpy, tpx)
        for (oy, ox, cch), v in out.items():
            got[oy, ox, cch * 8:cch * 8 + 8] = v
    done = ~np.isnan(got[..., 0])
    assert done.sum() == sum(min(E.TH, Ho - ty * E.TH) * min(E.TW, Wo - tx * E.TW) for ty, tx in tiles)
    assert not np.isnan(got[done]).any(), "uninitialised LDS reached the output"
    err = np.abs(got[done] - ref[done])
    assert float(err.max()) <= 2e-2 * (1 + float(np.abs(ref).max())), float(err.max())
    assert float((err > 4e-3 * (1 + np.abs(ref[done]))).mean()) < 2e-3


def test_layer1_fragment_reads_meet_sixteen_bank_slots():
    """Round 6: the stem patch is stored de-interleaved by column parity and the second patch row's lanes take rotated
    columns, so that every ds_read_b128 lane group of a layer-1 pixel-fragment read ({0-3, 12-15, 20-27} and {4-11, 16-19,
    28-31} of each half-wave, MI355X_MICROARCH.md) touches 16 different 16-B slots of the 256-B LDS line (the stride-2 walk
    over the interleaved patch reached 8: a two-way conflict on every read)."""
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (1, 64, 128, 3)).astype(np.uint8)
    W0 = (rs.standard_normal((32, 3, 6, 6)) * 0.15).astype(np.float32)
    lg1 = (rs.standard_normal((64, 288)) * 0.08).astype(np.float16)
    E.run_block(img, True, E.stem_pack_weights(W0), np.zeros(32, np.float32), C.pack_tiled(lg1, 64), np.zeros(64, np.float32),
                "silu", 0, 0, 1)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    assert len(E.READS) == 4 * 9 * 2
    for row, kc in E.READS:
        slot = (row % 4) * 4 + (kc ^ C.swz(row))           # 16-B slot of the 256-B line: row in the line, swizzled chunk
        for g in groups:
            assert len(set(int(slot[l]) for l in g)) == 16
