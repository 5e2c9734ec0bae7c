"""Lane-level CPU emulation of kernels_stem2.hip (stem + layer 1 in one kernel) -- test infrastructure, no GPU.
Mirrors the kernel statement by statement (LDS map, both input-staging paths, stem fragments, zeroed out-of-map stem
pixels, the two tap groups of layer 1, the aliased output tile); keep the two in sync.  See tests/c3_emul.py."""
from __future__ import annotations

import numpy as np

from c3_emul import act, swz

TW, TH = 16, 8
SW, SH = 2 * TW + 1, 2 * TH + 1
SPX = SW * SH
SFRAG = (SPX + 31) // 32
SROWS = SFRAG * 32
IW, IH = 2 * SW + 4, 2 * SH + 4
PITCH = 72 * 3
INH = IH * PITCH + 16
WTILE = 64 * 32
W0, W5 = 0, 5 * WTILE
IN = W5
U = W5 + max(INH, 4 * WTILE)
S = (U + 7) // 8 * 8
LDS = S + SROWS * 32
OP = 72


def stem_pack_weights(W):
    """kernels_fused.hip stem_pack_weights: [variant][9][64 lanes][8]."""
    out = np.zeros((2, 9, 64, 8), np.float16)
    for var in range(2):
        for s in range(9):
            for lane in range(64):
                for e in range(8):
                    n, k = lane & 31, 16 * s + 8 * (lane >> 5) + e
                    ky, j = divmod(k, 24)
                    if j >= 18:
                        continue
                    kx, c = divmod(j, 3)
                    out[var, s, lane, e] = np.float16(W[n, c, ky, kx] * (128.0 / 255.0 if var else 1.0))
    return out


def srow(py, px):
    """LDS row of stem pixel (py, px) of the 33 x 17 patch: a stem row's even columns first, then the odd ones (round 6)"""
    return py * SW + np.where(px & 1, 17 + (px >> 1), px >> 1)


READS = []      # (row per lane, chunk per lane) of every layer-1 pixel-fragment read of the last run_block: bank-slot audit


def mfma(fw, fx, acc):
    A = np.zeros((32, 16))
    Bm = np.zeros((16, 32))
    for lane in range(64):
        n, kg = lane & 31, (lane >> 5) * 8
        A[n, kg:kg + 8] = fw[lane].astype(np.float64)
        Bm[kg:kg + 8, n] = fx[lane].astype(np.float64)
    D = A @ Bm
    for lane in range(64):
        m, hi = lane & 31, lane >> 5
        for r in range(16):
            acc[lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hi, m]


def run_block(img, u8in, wfrag, b0, w1, b1, kind1, b, tpy, tpx):
    """img: (B,H,W,3) uint8 or (B,3,H,W) float32.  Returns {(oy, ox, chunk): 8 halves}."""
    if u8in:
        _, H, W, _ = img.shape
    else:
        _, _, H, W = img.shape
    Hs, Ws, Ho, Wo = H // 2, W // 2, H // 4, W // 4
    lds = np.full(LDS + 64, np.nan, np.float16)
    oy0, ox0 = tpy * TH, tpx * TW
    sy0, sx0 = 2 * oy0 - 1, 2 * ox0 - 1
    iy0, ix0 = 2 * sy0 - 2, 2 * sx0 - 2
    bias = np.concatenate([b0, b1]).astype(np.float32)

    def dma_w(pas, dst_base):
        for t in range(256):
            row, pos = (pas * 256 + t) >> 2, t & 3
            so = row * 32 + ((pos ^ swz(row)) * 8)
            d = dst_base + (t >> 6) * 64 * 8 + (t & 63) * 8
            lds[d:d + 8] = w1[so:so + 8]

    for i in range(5):
        dma_w(i, W0 + i * WTILE)
    wf = wfrag[1 if u8in else 0]            # (9, 64, 8)
    interior = iy0 >= 0 and iy0 + IH <= H and ix0 >= 0 and ix0 + 72 <= W
    if u8in and interior:
        flat = img[b].reshape(-1)
        base = (iy0 * W + ix0) * 3
        DPR = 53
        for i in range(IH * DPR):
            r, d = divmod(i, DPR)
            o = base + r * W * 3 + d * 4
            by = flat[o:o + 4]
            p = IN + r * PITCH + d * 4
            lds[p:p + 2] = by[:2].astype(np.float16)
            if d < DPR - 1:
                lds[p + 2:p + 4] = by[2:4].astype(np.float16)
    else:
        NEL = 3 * IH * IW
        for i in range(NEL):
            if u8in:
                r, j = divmod(i, IW * 3)
                q, c = divmod(j, 3)
            else:
                c, j = divmod(i, IH * IW)
                r, q = divmod(j, IW)
            iy, ix = iy0 + r, ix0 + q
            ok = 0 <= iy < H and 0 <= ix < W
            v = 0.0
            if ok:
                v = float(img[b, iy, ix, c]) if u8in else float(img[b, c, iy, ix])
            lds[IN + r * PITCH + q * 3 + c] = np.float16(v)
    for i in range(IH * 6):
        lds[IN + (i // 6) * PITCH + IW * 3 + i % 6] = 0
    lds[IN + IH * PITCH: IN + IH * PITCH + 16] = 0
    # ---- stem
    oscale = 1.0 / 128.0 if u8in else 1.0
    lanes = np.arange(64)
    l31, khalf = lanes & 31, lanes >> 5
    for w in range(4):
        for f in range(w, SFRAG, 4):
            p = 32 * f + l31
            pc = np.minimum(p, SPX - 1)
            py, px = pc // SW, pc % SW
            prow = IN + (2 * py) * PITCH + px * 6
            acc = np.zeros((64, 16))
            for s in range(9):
                offA = ((2 * s) // 3) * PITCH + ((2 * s) % 3) * 8
                offB = ((2 * s + 1) // 3) * PITCH + ((2 * s + 1) % 3) * 8
                fx = np.stack([lds[prow[l] + (offB if khalf[l] else offA): prow[l] + (offB if khalf[l] else offA) + 8]
                               for l in range(64)])
                mfma(wf[s], fx, acc)
            sy, sx = sy0 + py, sx0 + px
            keep = (p < SPX) & (sy >= 0) & (sy < Hs) & (sx >= 0) & (sx < Ws)
            rows = np.where(p < SPX, srow(py, px), p)
            for lane in range(64):
                kh, pp = lane >> 5, int(rows[lane])
                for g in range(4):
                    v = acc[lane, 4 * g:4 * g + 4].astype(np.float32) * np.float32(oscale) + bias[8 * g + 4 * kh: 8 * g + 4 * kh + 4]
                    o = act(v, "silu").astype(np.float16) if keep[lane] else np.zeros(4, np.float16)
                    a = S + pp * 32 + ((g ^ swz(pp)) * 8) + 4 * kh
                    lds[a:a + 4] = o
    lds[IN:IN + INH] = np.nan                    # the input patch is dead
    for i in range(4):
        dma_w(5 + i, W5 + i * WTILE)

    def ld(base, row, kc):
        o = base + row * 32 + ((kc ^ swz(row)) * 8)
        return lds[o:o + 8]

    accs = {}
    del READS[:]
    for w in range(4):
        prow1, pcol1 = 2 * w + (l31 >> 4), np.where(l31 < 16, l31, (l31 - 2) & 15)
        a0, a1 = np.zeros((64, 16)), np.zeros((64, 16))
        for tap in range(9):
            ty, tx = divmod(tap, 3)
            row = (2 * prow1 + ty) * SW + np.where(tx == 1, 17 + pcol1, pcol1 + (tx >> 1))
            assert np.array_equal(row, srow(2 * prow1 + ty, 2 * pcol1 + tx))
            Wb = W0 + tap * WTILE
            for kk in range(2):
                fw0 = np.stack([ld(Wb, int(l31[l]), kk * 2 + int(khalf[l])) for l in range(64)])
                fw1 = np.stack([ld(Wb, 32 + int(l31[l]), kk * 2 + int(khalf[l])) for l in range(64)])
                fx = np.stack([ld(S, int(row[l]), kk * 2 + int(khalf[l])) for l in range(64)])
                READS.append((row.copy(), kk * 2 + khalf))
                mfma(fw0, fx, a0)
                mfma(fw1, fx, a1)
        accs[w] = (a0, a1)
    lds[S:S + SROWS * 32] = np.nan               # the output tile overwrites the stem patch
    for w in range(4):
        prow1, pcol1 = 2 * w + (l31 >> 4), np.where(l31 < 16, l31, (l31 - 2) & 15)
        pl = prow1 * TW + pcol1
        for lane in range(64):
            kh = lane >> 5
            for i in range(2):
                for g in range(4):
                    v = accs[w][i][lane, 4 * g:4 * g + 4].astype(np.float32) + bias[32 + 32 * i + 8 * g + 4 * kh: 32 + 32 * i + 8 * g + 4 * kh + 4]
                    a = S + int(pl[lane]) * OP + 32 * i + 8 * g + 4 * kh
                    lds[a:a + 4] = act(v, kind1).astype(np.float16)
    out = {}
    for it in range(TW * TH // 32):
        for t in range(256):
            p, cch = it * 32 + (t >> 3), t & 7
            oy, ox = oy0 + (p >> 4), ox0 + (p & 15)
            if oy < Ho and ox < Wo:
                out[(oy, ox, cch)] = lds[S + p * OP + cch * 8: S + p * OP + cch * 8 + 8].copy()
    return out


def reference(img, u8in, W0f, b0, lg1, b1, kind1):
    """numpy: stem 6x6/s2/p2 (weights rounded to fp16 as packed) -> fp16 -> 3x3/s2/p1 (K = tap*32 + c) -> fp16.
    img: (H,W,3) u8 or (3,H,W) f32."""
    f16 = lambda a: a.astype(np.float16).astype(np.float64)        # noqa: E731
    if u8in:
        x = img.astype(np.float64)
        Wq = f16(W0f * np.float32(128.0 / 255.0))
        osc = 1.0 / 128.0
    else:
        x = f16(img.transpose(1, 2, 0))
        Wq = f16(W0f)
        osc = 1.0
    H, W, _ = x.shape
    xp = np.zeros((H + 4, W + 4, 3))
    xp[2:-2, 2:-2] = x
    Hs, Ws = H // 2, W // 2
    s = np.zeros((Hs, Ws, 32))
    for ky in range(6):
        for kx in range(6):
            s += xp[ky:ky + 2 * Hs:2, kx:kx + 2 * Ws:2] @ Wq[:, :, ky, kx].T
    s = f16(act((s * osc + b0).astype(np.float32), "silu"))
    sp = np.zeros((Hs + 2, Ws + 2, 32))
    sp[1:-1, 1:-1] = s
    Ho, Wo = H // 4, W // 4
    o = np.zeros((Ho, Wo, 64))
    for tap in range(9):
        ty, tx = divmod(tap, 3)
        o += sp[ty:ty + 2 * Ho:2, tx:tx + 2 * Wo:2] @ lg1[:, tap * 32:(tap + 1) * 32].T.astype(np.float64)
    return f16(act((o + b1).astype(np.float32), kind1))
