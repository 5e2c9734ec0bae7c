"""CPU check of seg_final_mfma_kernel's lane mapping (kernels_fused.hip): the A fragments cut from the packed
weights [C/8][16 taps][8], the accumulator rows -> taps map of the P store, and the col2im gather, replayed with the
MFMA as the outer-product step `mfma_probe_kernel` pins on the GPU, against a direct ConvTranspose2d(64, 1, 4, 2, 1)."""
import numpy as np

C, TP = 64, 18


def test_seg_final_mfma_lane_mapping():
    rs = np.random.RandomState(1)
    H = W = 16                                        # one 16x16 tile, zero halo
    x = rs.randn(H, W, C)
    Wt = rs.randn(C, 4, 4) * 0.2                      # (cin, ky, kx)
    wp = np.zeros((C // 8, 16, 8))
    for c in range(C):
        for kk in range(16):
            wp[c // 8, kk, c % 8] = Wt[c, kk // 4, kk % 4]
    wp = wp.reshape(-1)
    Ps = np.full((11 * 32, 16), np.nan)
    for f in range(11):
        acc = np.zeros((64, 16))
        for ks in range(C // 16):
            A = np.zeros((32, 16))
            Bm = np.zeros((16, 32))
            for lane in range(64):
                l31, hi = lane & 31, lane >> 5
                if l31 < 16:
                    o = ((2 * ks + hi) * 16 + l31) * 8
                    A[l31, 8 * hi:8 * hi + 8] = wp[o:o + 8]
                p = 32 * f + l31
                ty, tx = divmod(p, TP)
                yy, xx = ty - 1, tx - 1
                ok = p < TP * TP and 0 <= yy < H and 0 <= xx < W
                if ok:
                    Bm[8 * hi:8 * hi + 8, l31] = x[yy, xx, 16 * ks + 8 * hi:16 * ks + 8 * hi + 8]
            D = A @ Bm
            for lane in range(64):
                m, hi = lane & 31, lane >> 5
                for r in range(16):
                    acc[lane, r] += D[(r & 3) + 8 * (r >> 2) + 4 * hi, m]
        for lane in range(64):
            l31, hi = lane & 31, lane >> 5
            p = 32 * f + l31
            Ps[p, 4 * hi:4 * hi + 4] = acc[lane, 0:4]
            Ps[p, 8 + 4 * hi:8 + 4 * hi + 4] = acc[lane, 4:8]
    got = np.zeros((2 * H, 2 * W))
    for ly in range(16):
        for lx in range(16):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    Pn = Ps[(ly + 1 + dy) * TP + (lx + 1 + dx)]
                    for py in range(2):
                        ky = py + 1 - 2 * dy
                        if not 0 <= ky <= 3:
                            continue
                        for px in range(2):
                            kx = px + 1 - 2 * dx
                            if not 0 <= kx <= 3:
                                continue
                            got[2 * ly + py, 2 * lx + px] += Pn[ky * 4 + kx]
    # direct transposed convolution: out[2y - 1 + ky, 2x - 1 + kx] += x[y, x, c] * Wt[c, ky, kx]
    want = np.zeros((2 * H + 2, 2 * W + 2))
    for y in range(H):
        for xx in range(W):
            want[2 * y:2 * y + 4, 2 * xx:2 * xx + 4] += np.tensordot(x[y, xx], Wt, axes=(0, 0))
    want = want[1:-1, 1:-1]
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
