"""CPU: the index arithmetic of the big-tile ConvTranspose kernel that ships (csrc/kernels_halo3.hip: 256 pixels x 128 columns,
four waves) replayed lane by lane in numpy -- six patch passes -> LDS with the swizzle on the source chunk, the weight tile's
two passes (the second one = the px = 1 phase at 64 channels, the lower 64 rows of the packed tile otherwise, the second packed
tile at 256 channels), fragment read addresses `ra` / `wa` as the kernel computes them under the MFMAs (row base, phase shift,
lane rotation `xrot`, tap offsets, swizzle), the 32x32x16 MFMA operand / accumulator layout, the epilogue's lane -> staged pixel
/ column map and the store-out's (pixel, phase, channel) placement -- for N = 64 (both px phases of a py per block), N = 128
(one phase per block) and N = 256 (half a phase's channels per block), two sources, image borders included.  Values are
float64: what is checked is that every product lands where a direct ConvTranspose2d 4x4/s2/p1 puts it.  The GPU side (bit
identity with kernels_halo.hip, repeatability) is ctd_selftest + tests/test_gpu_edge.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

TWP = THP = 16
BKH = 32
NTHR = 256
A_ROWS = 384
BN3 = 128


def swz(row):
    return (row >> 2) & 3


def pack_tiles(wt, bnp):
    """ConvTranspose2d weight (Cin, N, 4, 4) -> the implicit-GEMM packing [phase][N / bnp][K / 32][bnp][32] with
    K index = tap * Cin + c, tap = ty * 2 + tx of the phase's 2x2 taps (csrc/selftest.hip, graph.py)."""
    cin, n = wt.shape[:2]
    K = 4 * cin
    lg = np.zeros((4, n, K))
    for ph in range(4):
        py, px = ph >> 1, ph & 1
        for ty in range(2):
            for tx in range(2):
                dy, dx = (0 if py else -1) + ty, (0 if px else -1) + tx
                ky, kx = py + 1 - 2 * dy, px + 1 - 2 * dx
                lg[ph, :, (ty * 2 + tx) * cin: (ty * 2 + tx + 1) * cin] = wt[:, :, ky, kx].T
    out = np.zeros((4, n // bnp, K // BKH, bnp, BKH))
    for ph in range(4):
        for tn in range(n // bnp):
            for ks in range(K // BKH):
                out[ph, tn, ks] = lg[ph, tn * bnp: (tn + 1) * bnp, ks * BKH: (ks + 1) * BKH]
    return out


def run_block(xs, wpk, bias, N, Hin, Win, b_y0, b_x0, pg, tile_n, out):
    """One block of conv_halo3_kernel<NPH, NT>: xs = list of (H, W, C) sources (concatenated channels), wpk = pack_tiles(...).
    Writes its 256 pixels x 128 columns into out (2H, 2W, N)."""
    NPH = 2 if N == 64 else 1
    CP = BN3 // NPH
    HW, HH = (18 if NPH == 2 else 17), 17
    py_b = pg if NPH == 2 else (pg >> 1)
    dy0 = 0 if py_b else -1
    dx0 = -1 if NPH == 2 else (0 if (pg & 1) else -1)
    x = np.concatenate(xs, axis=2)
    nchunk = x.shape[2] // BKH
    lanes = np.arange(64)
    l31, khalf, hi = lanes & 31, lanes >> 5, lanes >> 5
    xrot = np.where(l31 < 16, l31, (l31 - (HW - 16)) & 15)
    acc = np.zeros((4, 2, 4, 64, 16))                              # [wave][i][j][lane][register]
    for c in range(nchunk):
        # ---- patch: piece q = i * 256 + t of pass i: LDS row q / 4, chunk position q % 4 <- source chunk (q % 4) ^ swz(row)
        lds_a = np.zeros((A_ROWS, 4, 8))
        for q in range(6 * NTHR):
            r, pos = q >> 2, q & 3
            hy, hx = divmod(r, HW)
            iy, ix = b_y0 + hy + dy0, b_x0 + hx + dx0
            ok = r < HH * HW and 0 <= iy < Hin and 0 <= ix < Win
            src = pos ^ swz(r)
            lds_a[r, pos] = x[iy, ix, c * BKH + src * 8: c * BKH + src * 8 + 8] if ok else 0.0
        for tap in range(4):
            ks = tap * nchunk + c
            # ---- weight tile: thread t, pass j: LDS row j * 64 + t / 4; pass 1 = phase px = 1 (N = 64) or 64 rows further down
            lds_w = np.zeros((BN3, 4, 8))
            bnp = 64 if NPH == 2 else 128
            for j in range(2):
                for t in range(NTHR):
                    r, pos = t >> 2, t & 3
                    ps, n = divmod(r, CP)
                    phase = (pg * 2 + ps) if NPH == 2 else pg
                    src = pos ^ swz(r)
                    if NPH == 2:
                        row = wpk[phase + j, 0, ks, n]             # + w_phase_stride for pass 1
                    else:
                        row = wpk[phase, tile_n, ks, n + 64 * j]   # 64 rows further down the packed 128-row tile
                    lds_w[j * 64 + r, pos] = row[src * 8: src * 8 + 8]
            for wave in range(4):
                wn, wm = wave & 1, wave >> 1
                sx = wn if NPH == 2 else 0
                row_base = (2 * (wm * 4) + (l31 >> 4)) * HW + xrot + sx
                tapoff = (tap >> 1) * HW + (tap & 1)
                for kk in range(2):
                    fw, fx = [], []
                    for i in range(2):
                        wr = wn * 64 + l31 + i * 32
                        ch = (kk * 2 + khalf) ^ swz(l31)
                        fw.append(lds_w[wr, ch])                                   # (64 lanes, 8)
                    for j in range(4):
                        row = row_base + j * 2 * HW + tapoff
                        ch = (kk * 2 + khalf) ^ swz(row)
                        fx.append(lds_a[row, ch])
                    for i in range(2):
                        for j in range(4):
                            # D[n][m] += sum_k A[n][k] B[k][m]: lane l holds A row l % 32 / B column l % 32 for k = (l / 32) * 8 ..
                            A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
                            for l in range(64):
                                A[l & 31, (l >> 5) * 8: (l >> 5) * 8 + 8] = fw[i][l]
                                Bm[(l >> 5) * 8: (l >> 5) * 8 + 8, l & 31] = fx[j][l]
                            D = A @ Bm
                            for l in range(64):
                                for r_ in range(16):
                                    acc[wave, i, j, l, r_] += D[8 * (r_ // 4) + 4 * (l >> 5) + (r_ % 4), l & 31]
    # ---- epilogue: lane -> staged (pixel, column), then the store-out's (pixel, phase, channel)
    stage = np.zeros((256, BN3))
    for wave in range(4):
        wn, wm = wave & 1, wave >> 1
        for j in range(4):
            pl = (wm * 4 + j) * 32 + (l31 & 16) + xrot
            for i in range(2):
                for g in range(4):
                    for e in range(4):
                        nl = (wn * 2 + i) * 32 + 4 * hi + 8 * g + e
                        stage[pl, nl] = acc[wave, i, j, lanes, 4 * g + e] + bias[tile_n * BN3 + nl % CP]
    for t in range(NTHR):
        cch, pcol = t % 16, t // 16
        col = cch * 8
        ps_o = col // CP
        n = tile_n * BN3 + col - ps_o * CP
        px = ps_o if NPH == 2 else (pg & 1)
        for it in range(16):
            pl = it * 16 + pcol
            oy, ox = (b_y0 + it) * 2 + py_b, (b_x0 + pcol) * 2 + px
            out[oy, ox, n: n + 8] = stage[pl, col: col + 8]


@pytest.mark.parametrize("N,c0,c1,H,W", [(64, 32, 32, 16, 32), (128, 64, 0, 32, 16), (256, 32, 0, 16, 16)])
def test_halo3_index_arithmetic_against_conv_transpose(N, c0, c1, H, W):
    rng = np.random.RandomState(N + c0)
    cin = c0 + c1
    xs = [rng.randn(H, W, c0)] + ([rng.randn(H, W, c1)] if c1 else [])
    wt = rng.randn(cin, N, 4, 4) / np.sqrt(4 * cin)
    bias = rng.randn(N)
    bnp = 64 if N == 64 else 128
    wpk = pack_tiles(wt, bnp)
    out = np.full((2 * H, 2 * W, N), np.nan)
    NPH, NT = (2, 1) if N == 64 else ((1, 1) if N == 128 else (1, 2))
    for ty in range(H // THP):
        for tx in range(W // TWP):
            for pg in range(4 // NPH):
                for tile_n in range(NT):
                    run_block(xs, wpk, bias, N, H, W, ty * THP, tx * TWP, pg, tile_n, out)
    assert not np.isnan(out).any(), "some output element was never written"
    xt = torch.from_numpy(np.concatenate(xs, axis=2).transpose(2, 0, 1)[None])
    ref = F.conv_transpose2d(xt, torch.from_numpy(wt), torch.from_numpy(bias), stride=2, padding=1)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=1e-9)
