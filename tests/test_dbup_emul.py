"""CPU check of db_up_mfma_kernel's lane mapping (kernels_fused.hip): MFMA fragment rows n = pp * 16 + o, the
accumulator-register -> (position, hidden channel) map, the half-sum swap with lane ^ 32 and the output rows /
columns each lane stores, replayed with the MFMA as the outer-product step `mfma_probe_kernel` pins on the GPU."""
import numpy as np

Q = 16


def emulate(x, W1p, b1, W2p, b2):
    """x: (32, 16) one pixel group.  Returns out[m][4 rows][4 cols] as the kernel stores them (row = 2 py + qy)."""
    out = np.full((32, 4, 4), np.nan)
    part = np.zeros((64, 4, 4))
    for f in range(2):
        A = np.zeros((32, 16))
        Bm = np.zeros((16, 32))
        for lane in range(64):
            l31, hi = lane & 31, lane >> 5
            n = f * 32 + l31
            pp, o = n >> 4, n & 15
            A[l31, 8 * hi:8 * hi + 8] = [W1p[pp][8 * hi + e][o] for e in range(8)]
            Bm[8 * hi:8 * hi + 8, l31] = x[l31, 8 * hi:8 * hi + 8]
        D = A @ Bm
        for lane in range(64):
            m, hi = lane & 31, lane >> 5
            acc = [D[(r & 3) + 8 * (r >> 2) + 4 * hi, m] for r in range(16)]
            for ph in range(2):
                s4 = np.zeros(4)
                for j in range(8):
                    o = (j & 3) + 8 * ((j >> 2) & 1) + 4 * hi
                    hv = max(acc[ph * 8 + j] + b1[o], 0.0)
                    for qq in range(4):
                        s4[qq] += hv * W2p[qq][o]
                part[lane, 2 * f + ph] = s4
    for lane in range(64):
        m, hi = lane & 31, lane >> 5
        r8 = np.zeros((2, 4))
        for j in range(2):
            for qq in range(4):
                send_partner = part[lane ^ 32, j, qq] if (lane ^ 32) >> 5 else part[lane ^ 32, 2 + j, qq]
                mine = part[lane, 2 + j, qq] if hi else part[lane, j, qq]
                r8[j, qq] = 1 / (1 + np.exp(-(mine + send_partner + b2)))
        for qy in range(2):
            row = 2 * hi + qy
            assert np.isnan(out[m, row]).all()
            out[m, row] = [r8[0, qy * 2], r8[0, qy * 2 + 1], r8[1, qy * 2], r8[1, qy * 2 + 1]]
    return out


def test_db_up_mfma_lane_mapping():
    rs = np.random.RandomState(0)
    x = rs.rand(32, Q)
    W1p, b1, W2p, b2 = rs.randn(4, Q, Q) * 0.3, rs.randn(Q) * 0.2, rs.randn(4, Q), 0.1
    got = emulate(x, W1p, b1, W2p, b2)
    want = np.zeros((32, 4, 4))
    for m in range(32):
        for pp in range(4):
            h = np.maximum(x[m] @ W1p[pp] + b1, 0)          # W1p[pp][c][o]
            py, px = pp >> 1, pp & 1
            for qq in range(4):
                qy, qx = qq >> 1, qq & 1
                want[m, 2 * py + qy, 2 * px + qx] = 1 / (1 + np.exp(-(h @ W2p[qq] + b2)))
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
