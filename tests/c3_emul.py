"""Lane-level CPU emulation of kernels_c3.hip's index arithmetic (test infrastructure, no GPU).

The fused C3 kernel is LDS choreography: swizzled LDS-DMA placement, MFMA fragment addressing, in-place shortcut,
aliasing of dead regions.  This module replays ONE block of it on the CPU -- same LDS map, same per-lane addresses,
the MFMA as the (A rows = lane & 31, k group = lane >> 5) x (B cols = lane & 31) outer-product step that
`mfma_probe_kernel` pins on the GPU -- so `tests/test_c3_emul.py` can check the data flow against a plain numpy
C3 block without hardware.  It mirrors the kernel statement by statement; keep the two in sync.
"""
from __future__ import annotations

import numpy as np

TW, TH, HW, HH = 16, 8, 18, 10
HROWS, NROW, PX = 180, 192, 128
XR, XBUF = 0, NROW * 32
WR, WBUF = 2 * XBUF, 64 * 32
Y1 = WR + 2 * WBUF
T = Y1 + NROW * 32
Y2 = T + NROW * 32
LDS = Y2 + PX * 32
OP = 72
WM1 = 9 * 32 * 32


def swz(row):
    return (row >> 2) & 3


def act(v, kind):
    if kind == "silu":
        return v / (1.0 + np.exp(-v))
    if kind == "leaky":
        return np.where(v > 0, v, 0.1 * v)
    return np.maximum(v, 0)


def pack_tiled(lg, bn):
    """igemm_pack_weights(tiled, bk = 32): [N/bn][K/32][bn][32]."""
    N, K = lg.shape
    out = np.zeros(N * K, np.float16)
    nk = K // 32
    for n in range(N):
        for k in range(K):
            out[(((n // bn) * nk + k // 32) * bn + n % bn) * 32 + k % 32] = lg[n, k]
    return out


class Block:
    def __init__(self, x, w12, wm1, wm2, wc3, b12, bm1, bm2, bc3, kind, b, tpy, tpx):
        """x: (B,H,W,Cin) fp16; w*: packed fp16 flat arrays; biases float32."""
        self.x, self.kind = x, kind
        self.B, self.H, self.W, self.cin = x.shape
        self.w12, self.wm1, self.wm2, self.wc3 = w12, wm1, wm2, wc3
        self.bias = np.concatenate([b12, bm1, bm2, bc3]).astype(np.float32)
        self.lds = np.full(LDS, np.nan, np.float16)      # NaN = never written
        self.b, self.y0, self.x0 = b, tpy * TH, tpx * TW

    # ---- helpers mirroring the kernel's lambdas ----
    def inside(self, r):
        hy, hx = divmod(r, HW)
        iy, ix = self.y0 - 1 + hy, self.x0 - 1 + hx
        return r < HROWS and 0 <= iy < self.H and 0 <= ix < self.W

    def dma(self, src16, dst):
        """one lane's 16-B LDS-DMA: 8 halves to LDS half index dst"""
        self.lds[dst:dst + 8] = src16

    def ld(self, base, row, kc):
        o = base + row * 32 + ((kc ^ swz(row)) * 8)
        v = self.lds[o:o + 8]
        return v

    def mfma(self, fw, fx, acc):
        """fw, fx: (64, 8) fp16 per lane; acc: (64, 16) f32 per lane."""
        A = np.zeros((32, 16), np.float64)
        Bm = np.zeros((16, 32), np.float64)
        for lane in range(64):
            n, kg = lane & 31, (lane >> 5) * 8
            A[n, kg:kg + 8] = fw[lane].astype(np.float64)
            Bm[kg:kg + 8, n] = fx[lane].astype(np.float64)
        D = A @ Bm
        for lane in range(64):
            m, hi = lane & 31, lane >> 5
            for r in range(16):
                acc[lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hi, m]

    def store_frag(self, base, rows, acc, boff, keep):
        for lane in range(64):
            kh = lane >> 5
            row = rows[lane]
            for g in range(4):
                v = acc[lane, 4 * g:4 * g + 4] + self.bias[boff + 8 * g + 4 * kh: boff + 8 * g + 4 * kh + 4]
                o = act(v.astype(np.float32), self.kind).astype(np.float16) if keep[lane] else np.zeros(4, np.float16)
                a = base + row * 32 + ((g ^ swz(row)) * 8) + 4 * kh
                self.lds[a:a + 4] = o

    def run(self):
        x, b, y0, x0 = self.x, self.b, self.y0, self.x0
        nch = self.cin // 32

        def dma_x(chunk, buf):
            for i in range(3):
                for t in range(256):
                    q = i * 256 + t
                    r, pos = q >> 2, q & 3
                    hy, hx = divmod(r, HW)
                    iy, ix = y0 - 1 + hy, x0 - 1 + hx
                    gs = (pos ^ swz(r)) * 8
                    src = x[b, iy, ix, chunk * 32 + gs: chunk * 32 + gs + 8] if self.inside(r) else np.zeros(8, np.float16)
                    w = t >> 6
                    self.dma(src, XR + buf * XBUF + (i * 256 + w * 64) * 8 + (t & 63) * 8)

        def dma_w12(chunk, buf):
            for t in range(256):
                wrow, pos = t >> 2, t & 3
                woff = wrow * 32 + ((pos ^ swz(wrow)) * 8)
                src = self.w12[chunk * WBUF + woff: chunk * WBUF + woff + 8]
                self.dma(src, WR + buf * WBUF + (t >> 6) * 64 * 8 + (t & 63) * 8)

        lanes = np.arange(64)
        l31, khalf = lanes & 31, lanes >> 5
        accs = {}
        for w in range(4):
            accs[w] = dict(A=np.zeros((64, 16)), B=np.zeros((64, 16)), C=np.zeros((64, 16)))

        def geom(w):
            prow = 2 * w + (l31 >> 4)
            pcol = np.where(l31 < 16, l31, (l31 - (HW - 16)) & 15)
            return prow, pcol, prow * TW + pcol, (prow + 1) * HW + pcol + 1, 32 * w + l31, 32 * (4 + w) + l31

        def frag(base, rows, kk):
            return np.stack([self.ld(base, int(rows[l]), kk * 2 + int(khalf[l])) for l in range(64)])

        # S1
        dma_x(0, 0)
        dma_w12(0, 0)
        if nch > 1:
            dma_x(1, 1)
            dma_w12(1, 1)
        for c in range(nch):
            if c >= 1 and c + 1 < nch:
                dma_x(c + 1, (c + 1) & 1)
                dma_w12(c + 1, (c + 1) & 1)
            Xb, Wb = XR + (c & 1) * XBUF, WR + (c & 1) * WBUF
            for w in range(4):
                prow, pcol, pl, rowIn, rowA, rowC = geom(w)
                for kk in range(2):
                    fw1, fw2 = frag(Wb, l31, kk), frag(Wb, 32 + l31, kk)
                    self.mfma(fw1, frag(Xb, rowA, kk), accs[w]["A"])
                    self.mfma(fw2, frag(Xb, rowIn, kk), accs[w]["B"])
                    if w < 2:
                        self.mfma(fw1, frag(Xb, rowC, kk), accs[w]["C"])
        # weights of S2-S4 overwrite the x buffers / w12 ring
        self.lds[XR:XR + 2 * XBUF] = np.nan
        self.lds[WR:WR + 2 * WBUF] = np.nan
        for i in range(5):
            for t in range(256):
                q = i * 256 + t
                row, pos = q >> 2, q & 3
                so = ((pos ^ swz(row)) * 8)
                src = self.wm2[row * 32 + so: row * 32 + so + 8] if row < 288 else self.wm1[(row - 288) * 32 + so:(row - 288) * 32 + so + 8]
                self.dma(src, XR + (i * 256 + (t >> 6) * 64) * 8 + (t & 63) * 8)
        for i in range(2):
            for t in range(256):
                q = i * 256 + t
                row, pos = q >> 2, q & 3
                so = row * 32 + ((pos ^ swz(row)) * 8)
                self.dma(self.wc3[so:so + 8], WR + (i * 256 + (t >> 6) * 64) * 8 + (t & 63) * 8)
        ones = np.ones(64, bool)
        for w in range(4):
            prow, pcol, pl, rowIn, rowA, rowC = geom(w)
            self.store_frag(Y1, rowA, accs[w]["A"], 0, ones)
            self.store_frag(Y2, pl, accs[w]["B"], 32, ones)
            if w < 2:
                self.store_frag(Y1, rowC, accs[w]["C"], 0, ones)
        # S2
        newT = {}
        for w in range(4):
            prow, pcol, pl, rowIn, rowA, rowC = geom(w)
            a = np.zeros((64, 16))
            c = np.zeros((64, 16))
            for kk in range(2):
                fw = frag(XR + WM1, l31, kk)
                self.mfma(fw, frag(Y1, rowA, kk), a)
                if w < 2:
                    self.mfma(fw, frag(Y1, rowC, kk), c)
            newT[w] = (a, c)
        for w in range(4):
            prow, pcol, pl, rowIn, rowA, rowC = geom(w)
            self.store_frag(T, rowA, newT[w][0], 64, np.array([self.inside(int(r)) for r in rowA]))
            if w < 2:
                self.store_frag(T, rowC, newT[w][1], 64, np.array([self.inside(int(r)) for r in rowC]))
        # S3
        accB = {}
        for w in range(4):
            prow, pcol, pl, rowIn, rowA, rowC = geom(w)
            a = np.zeros((64, 16))
            for tap in range(9):
                ty, tx = divmod(tap, 3)
                row = (prow + ty) * HW + pcol + tx
                for kk in range(2):
                    self.mfma(frag(XR + tap * 1024, l31, kk), frag(T, row, kk), a)
            accB[w] = a
        for w in range(4):
            prow, pcol, pl, rowIn, rowA, rowC = geom(w)
            for lane in range(64):
                kh, ri = lane >> 5, int(rowIn[lane])
                for g in range(4):
                    p = Y1 + ri * 32 + ((g ^ swz(ri)) * 8) + 4 * kh
                    y = self.lds[p:p + 4].astype(np.float32)
                    v = accB[w][lane, 4 * g:4 * g + 4] + self.bias[96 + 8 * g + 4 * kh: 96 + 8 * g + 4 * kh + 4]
                    u = act(v.astype(np.float32), self.kind).astype(np.float16).astype(np.float32)
                    self.lds[p:p + 4] = (u + y).astype(np.float16)
        # S4
        acc4 = {}
        for w in range(4):
            prow, pcol, pl, rowIn, rowA, rowC = geom(w)
            a0, a1 = np.zeros((64, 16)), np.zeros((64, 16))
            for kchunk in range(2):
                Wb = WR + kchunk * WBUF
                for kk in range(2):
                    fx = frag(Y1, rowIn, kk) if kchunk == 0 else frag(Y2, pl, kk)
                    self.mfma(frag(Wb, l31, kk), fx, a0)
                    self.mfma(frag(Wb, 32 + l31, kk), fx, a1)
            acc4[w] = (a0, a1)
        self.lds[XR:XR + 2 * XBUF] = np.nan          # the output tile overwrites the tap tiles
        for w in range(4):
            prow, pcol, pl, rowIn, rowA, rowC = geom(w)
            for lane in range(64):
                kh = lane >> 5
                for i in range(2):
                    for g in range(4):
                        v = acc4[w][i][lane, 4 * g:4 * g + 4] + self.bias[128 + 32 * i + 8 * g + 4 * kh: 128 + 32 * i + 8 * g + 4 * kh + 4]
                        a = XR + int(pl[lane]) * OP + 32 * i + 8 * g + 4 * kh
                        self.lds[a:a + 4] = act(v.astype(np.float32), self.kind).astype(np.float16)
        out = {}
        for it in range(PX // 32):
            for t in range(256):
                p, cch = it * 32 + (t >> 3), t & 7
                oy, ox = y0 + (p >> 4), x0 + (p & 15)
                if oy < self.H and ox < self.W:
                    out[(oy, ox, cch)] = self.lds[XR + p * OP + cch * 8: XR + p * OP + cch * 8 + 8].copy()
        return out


def reference(x, W12, Wm1, Wm2, Wc3, b12, bm1, bm2, bc3, kind):
    """Plain numpy C3 block with the fp16 rounding points of the unfused engine.  x: (H,W,Cin) fp16;
    W12 (64,Cin), Wm1 (32,32), Wm2 (32, 9*32) [K = tap*32 + c], Wc3 (64,64): fp16-representable floats."""
    f = lambda a: a.astype(np.float16).astype(np.float64)          # noqa: E731
    H, W, _ = x.shape
    xx = x.astype(np.float64)
    y = f(act(xx @ W12.T.astype(np.float64) + b12, kind))
    y1, y2 = y[..., :32], y[..., 32:]
    t = f(act(y1 @ Wm1.T.astype(np.float64) + bm1, kind))
    tp = np.zeros((H + 2, W + 2, 32))
    tp[1:-1, 1:-1] = t
    u = np.zeros((H, W, 32))
    for tap in range(9):
        ty, tx = divmod(tap, 3)
        u += tp[ty:ty + H, tx:tx + W] @ Wm2[:, tap * 32:(tap + 1) * 32].T.astype(np.float64)
    bsum = f(f(act(u + bm2, kind)) + y1)
    return f(act(np.concatenate([bsum, y2], -1) @ Wc3.T.astype(np.float64) + bc3, kind))
