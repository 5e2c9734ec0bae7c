"""CPU: the index arithmetic of the haloed-patch split kernel (csrc/kernels_split_halo.hip) replayed lane by lane in numpy --
patch rows -> LDS-DMA with the swizzle on the source chunk -> fragment reads at shifted rows with the lane rotation `xrot` ->
32x32x16 MFMA operand / accumulator layout -> the epilogue's lane -> pixel map; for a 3x3 convolution, one ConvTranspose
phase, and the PAIR mode (both px phases of a 64-channel ConvTranspose per block).  Values are float64 (the hi / lo split is
tests/test_split_emul.py's subject): what is checked is that every product lands where a direct convolution puts it.  The
GPU side is `ST_SPLIT=1 ctd_selftest` (tests/test_gpu_selftest.py)."""
import numpy as np

TWP = 16
BKH = 32


def swz(row):
    return (row >> 2) & 3


def run_block(x, wpk, Hin, Win, y0, x0, dy0, dx0, KH, KW, BN, pair, THP=16, WGN=2, WGM=4):
    """One block: x (Hin, Win, C) input, wpk packed weight rows as the kernel addresses them: wpk[row, ks, 32] where `row` is
    the LDS weight row (0..BN-1) already resolved to its source row (PAIR: rows 64.. come from the next phase), ks = tap * nchunk
    + chunk.  THP = 16: 256-pixel patch, 512 threads; THP = 8: 128-pixel patch, 256 threads.  Returns out[tile pixels, BN staged columns]."""
    NTHR, AROWS_PAD, NW = 32 * THP, (336 if THP == 16 else 192), THP // 2
    assert WGN * WGM == NW
    C = x.shape[2]
    nchunk = C // BKH
    HW = TWP + (3 if pair else KW) - 1
    HH = THP + KH - 1
    taps = KH * KW
    TN, TM = BN // (32 * WGN), 16 * THP // (32 * WGM)
    acc = np.zeros((NW, TN, TM, 64, 16))                          # [wave][i][j][lane][register]
    lanes = np.arange(64)
    l31, khalf = lanes & 31, lanes >> 5
    xrot = np.where(l31 < 16, l31, (l31 - (HW - 16)) & 15)
    for c in range(nchunk):
        # ---- patch of this chunk: thread t of pass i stages chunk q = i * 512 + t: row q / 4 at LDS position q % 4, fetched
        # from source chunk (q % 4) ^ swz(row)
        lds = np.zeros((AROWS_PAD, 4, 8))
        for q in range(3 * NTHR):
            r, pos = q >> 2, q & 3
            if r >= AROWS_PAD:
                continue
            hy, hx = divmod(r, HW)
            iy, ix = y0 + hy + dy0, x0 + hx + dx0
            ok = r < HH * HW and 0 <= iy < Hin and 0 <= ix < Win
            src = pos ^ swz(r)
            lds[r, pos] = x[iy, ix, c * BKH + src * 8: c * BKH + src * 8 + 8] if ok else 0.0
        for tap in range(taps):
            ty, tx = divmod(tap, KW)
            ks = tap * nchunk + c
            # ---- weight tile of this step in LDS: thread t stages chunk t: row t / 4, position t % 4 <- source chunk ^ swz(row)
            wl = np.zeros((BN, 4, 8))
            for t in range(BN * 4):
                wr, pos = t >> 2, t & 3
                src = pos ^ swz(wr)
                wl[wr, pos] = wpk[wr, ks, src * 8: src * 8 + 8]
            for wave in range(NW):
                wn, wm = wave % WGN, wave // WGN
                tapoff = ty * HW + tx + (wn if pair else 0)
                for kk in range(2):
                    for i in range(TN):
                        wrow = (wn * TN + i) * 32 + l31                                       # A operand: lane = weight row
                        A = wl[wrow, (kk * 2 + khalf) ^ swz(l31)]                            # [64 lanes, 8 k]
                        for j in range(TM):
                            f = wm * TM + j
                            row = (2 * f + (l31 >> 4)) * HW + xrot + tapoff                   # B operand: lane = pixel column
                            Bv = lds[row, (kk * 2 + khalf) ^ swz(row)]                        # [64 lanes, 8 k]
                            # v_mfma_f32_32x32x16: D[n][m] += sum_k A[n][k] B[m][k]; lane (m = l31, hi = lane / 32) holds rows
                            # n = 4 hi + 8 g + e in register 4 g + e; K half `khalf` comes from lanes with that khalf
                            An = np.zeros((32, 16))
                            Bm = np.zeros((32, 16))
                            for ln in range(64):
                                An[l31[ln], khalf[ln] * 8: khalf[ln] * 8 + 8] = A[ln]
                                Bm[l31[ln], khalf[ln] * 8: khalf[ln] * 8 + 8] = Bv[ln]
                            D = An @ Bm.T                                                     # [n, m]
                            for ln in range(64):
                                hi = ln >> 5
                                for g in range(4):
                                    for e in range(4):
                                        acc[wave, i, j, ln, 4 * g + e] += D[4 * hi + 8 * g + e, l31[ln]]
    # ---- epilogue: register (lane, g, e) of fragment (i, j) -> staged tile [pixel][column]
    out = np.zeros((16 * THP, BN))
    for wave in range(NW):
        wn, wm = wave % WGN, wave // WGN
        for i in range(TN):
            for j in range(TM):
                for ln in range(64):
                    hi = ln >> 5
                    p = (wm * TM + j) * 32 + (l31[ln] & 16) + xrot[ln]
                    for g in range(4):
                        nl = (wn * TN + i) * 32 + 4 * hi + 8 * g
                        out[p, nl: nl + 4] = acc[wave, i, j, ln, 4 * g: 4 * g + 4]
    return out


def pack(w, N, K):
    """logical weights [N][K] (K = tap * C + c) -> rows addressed as [row][K / 32][32]"""
    return w.reshape(N, K // 32, 32)


def test_3x3_patch_with_partial_tiles_matches_a_direct_convolution():
    rng = np.random.RandomState(0)
    H = W = 21
    C, N = 64, 128
    x = rng.standard_normal((H, W, C))
    w = rng.standard_normal((N, 9 * C))                                            # K = tap * C + c, tap = ty * 3 + tx
    ref = np.zeros((H, W, N))
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    for ty in range(3):
        for tx in range(3):
            ref += xp[ty: ty + H, tx: tx + W] @ w[:, (ty * 3 + tx) * C: (ty * 3 + tx + 1) * C].T
    for y0, x0 in ((0, 0), (16, 16), (0, 16)):
        out = run_block(x, pack(w, N, 9 * C), H, W, y0, x0, -1, -1, 3, 3, 128, False)
        for p in range(256):
            oy, ox = y0 + (p >> 4), x0 + (p & 15)
            if oy < H and ox < W:
                np.testing.assert_allclose(out[p], ref[oy, ox], rtol=0, atol=1e-9)


def convt_phase_ref(x, wph, py, px):
    """phase (py, px) of ConvTranspose 4x4/s2/p1 as a 2x2-tap convolution: out[y, x] = sum_{ty,tx} x[y + dy0 + ty, x + dx0 + tx] W[ty*2+tx]"""
    H, W, C = x.shape
    dy0, dx0 = (0 if py else -1), (0 if px else -1)
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    out = np.zeros((H, W, wph.shape[0]))
    for ty in range(2):
        for tx in range(2):
            sl = xp[1 + dy0 + ty: 1 + dy0 + ty + H, 1 + dx0 + tx: 1 + dx0 + tx + W]
            out += sl @ wph[:, (ty * 2 + tx) * C: (ty * 2 + tx + 1) * C].T
    return out


def test_convt_phase_and_the_phase_pair_match_direct_phase_convolutions():
    rng = np.random.RandomState(1)
    H = W = 19
    C = 64
    x = rng.standard_normal((H, W, C))
    # -- one phase per block, 128 output channels
    w = rng.standard_normal((4, 128, 4 * C))
    for phase in (0, 3):
        py, px = phase >> 1, phase & 1
        ref = convt_phase_ref(x, w[phase], py, px)
        out = run_block(x, pack(w[phase], 128, 4 * C), H, W, 16, 0, 0 if py else -1, 0 if px else -1, 2, 2, 128, False)
        for p in range(256):
            oy, ox = 16 + (p >> 4), p & 15
            if oy < H and ox < W:
                np.testing.assert_allclose(out[p], ref[oy, ox], rtol=0, atol=1e-9)
    # -- PAIR: 64 output channels, the block computes (py, px = 0) in columns 0-63 and (py, px = 1) in columns 64-127 from ONE
    #    patch spanning dx = -1 .. +1; LDS weight rows 64.. come from the next phase's rows
    w64 = rng.standard_normal((4, 64, 4 * C))
    for py in (0, 1):
        wpk = np.concatenate([pack(w64[2 * py], 64, 4 * C), pack(w64[2 * py + 1], 64, 4 * C)], 0)      # rows 0-63 | 64-127
        out = run_block(x, wpk, H, W, 0, 16, 0 if py else -1, -1, 2, 2, 128, True)
        for px in (0, 1):
            ref = convt_phase_ref(x, w64[2 * py + px], py, px)
            for p in range(256):
                oy, ox = p >> 4, 16 + (p & 15)
                if oy < H and ox < W:
                    np.testing.assert_allclose(out[p, 64 * px: 64 * px + 64], ref[oy, ox], rtol=0, atol=1e-9)


def test_3x3_on_16x8_patches_64_channels():
    """the 64-channel configuration: 16x8 patches, four waves of 64 channels x 32 pixels each"""
    rng = np.random.RandomState(2)
    H, W = 13, 21
    C, N = 32, 64
    x = rng.standard_normal((H, W, C))
    w = rng.standard_normal((N, 9 * C))
    ref = np.zeros((H, W, N))
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    for ty in range(3):
        for tx in range(3):
            ref += xp[ty: ty + H, tx: tx + W] @ w[:, (ty * 3 + tx) * C: (ty * 3 + tx + 1) * C].T
    for y0, x0 in ((0, 0), (8, 16)):
        out = run_block(x, pack(w, N, 9 * C), H, W, y0, x0, -1, -1, 3, 3, 64, False, THP=8, WGN=1, WGM=4)
        for p in range(128):
            oy, ox = y0 + (p >> 4), x0 + (p & 15)
            if oy < H and ox < W:
                np.testing.assert_allclose(out[p], ref[oy, ox], rtol=0, atol=1e-9)
