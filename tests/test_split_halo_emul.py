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


# ---- the first layer straight from the page (csrc/kernels_split_stem.hip) -------------------------------------------------------
def test_first_layer_tap_pairs_and_tile_map_match_a_direct_convolution():
    """6x6 / stride 2 / pad 2 over a 3-channel page: a block owns 16x16 outputs and keeps the 36x36 input patch as 8-B pixels
    (4 channels, the 4th zero); a lane's K = 16 operand at step kk is taps (4 kk + 2 khalf, + 1) -- two horizontally adjacent
    pixels = 16 contiguous bytes -- with taps 36..39 padding K to 160 (zero weights, any finite operand).  K index = tap * 4 + c."""
    rng = np.random.RandomState(3)
    H, W = 44, 52                                  # outputs 22 x 26: partial tiles
    x = rng.standard_normal((H, W, 3))
    w = rng.standard_normal((32, 36, 3))           # [n][tap = ty * 6 + tx][c]
    Ho, Wo = H // 2, W // 2
    xp = np.pad(x, ((2, 2), (2, 2), (0, 0)))
    ref = np.zeros((Ho, Wo, 32))
    for ty in range(6):
        for tx in range(6):
            ref += xp[ty: ty + 2 * Ho: 2, tx: tx + 2 * Wo: 2] @ w[:, ty * 6 + tx].T
    wk = np.zeros((32, 160))                       # packed K axis: tap * 4 + c, taps 36..39 and channel 3 zero
    for tap in range(36):
        wk[:, tap * 4: tap * 4 + 3] = w[:, tap]
    ST, SP = 16, 36
    lanes = np.arange(64)
    l31, khalf = lanes & 31, lanes >> 5
    for y0, x0 in ((0, 0), (16, 16)):
        patch = np.zeros((SP * SP, 4))
        for q in range(SP * SP):
            hy, hx = divmod(q, SP)
            iy, ix = 2 * y0 - 2 + hy, 2 * x0 - 2 + hx
            if 0 <= iy < H and 0 <= ix < W:
                patch[q, :3] = x[iy, ix]
        flat = patch.reshape(-1)                   # halves of one plane: pixel q at [4 q, 4 q + 4)
        out = np.zeros((256, 32))
        for wave in range(4):
            for j in range(2):
                py, px = 4 * wave + 2 * j + (l31 >> 4), l31 & 15
                base = (2 * py) * SP + 2 * px
                D = np.zeros((32, 32))                                            # [n][pixel column]
                for kk in range(10):
                    tp = np.minimum(kk * 4 + khalf * 2, 34)
                    ty, tx = tp // 6, tp % 6
                    o = (ty * SP + tx) * 4
                    Bm = np.zeros((32, 16))
                    An = np.zeros((32, 16))
                    for ln in range(64):
                        s = base[ln] * 4 + o[ln]
                        Bm[l31[ln], khalf[ln] * 8: khalf[ln] * 8 + 8] = flat[s: s + 8]
                        k0 = kk * 16 + khalf[ln] * 8
                        An[l31[ln], khalf[ln] * 8: khalf[ln] * 8 + 8] = wk[l31[ln], k0: k0 + 8]
                    D += An @ Bm.T
                for ln in range(32):
                    p = (2 * wave + j) * 32 + ln                                  # tile pixel of MFMA column ln
                    out[p] = D[:, ln]
        for p in range(256):
            oy, ox = y0 + (p >> 4), x0 + (p & 15)
            if oy < Ho and ox < Wo:
                np.testing.assert_allclose(out[p], ref[oy, ox], rtol=0, atol=1e-9)
