"""CPU: the arithmetic of the split-operand engine (`precision="fp32s"`, csrc/kernels_split.hip) restated in numpy --
x = hi + lo in fp16, w scaled per output channel by a power of two and split the same way, three products.  What the GPU
selftest measures against a float64 reference is asserted here as bounds: the representation error of the three-product form
is far below the rounding noise of an fp32 accumulation of the same length, the scale is exact, and the dropped lo x lo term
is the 2^-22 it is said to be."""
import numpy as np


def split(a):
    a = a.astype(np.float32)
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def row_scale(w):
    """the packer's scale (split_pack_weights): the largest |w| of a row lands in [512, 1024)"""
    mx = np.abs(w).max(1, keepdims=True)
    _, e = np.frexp(mx)
    return np.ldexp(np.float32(1), 10 - e).astype(np.float32)


def test_three_product_form_is_more_accurate_than_an_fp32_chain():
    rng = np.random.RandomState(0)
    K, M, N = 1152, 2048, 48                       # a 3x3 conv over 128 channels
    x = rng.standard_normal((M, K)).astype(np.float32) * 2
    x = (x / (1 + np.exp(-x))).astype(np.float32)   # SiLU-like activations: many small values, a few large
    w = (rng.standard_normal((N, K)) * rng.uniform(0.002, 0.3, (N, 1))).astype(np.float32)
    truth = x.astype(np.float64) @ w.astype(np.float64).T
    s = row_scale(w)
    ws = w * s
    assert np.array_equal(ws / s, w)                                        # a power of two: exact both ways
    assert (np.abs(ws).max(1) >= 512).all() and (np.abs(ws).max(1) < 1024).all()
    xh, xl = split(x)
    wh, wl = split(ws)
    f = lambda a: a.astype(np.float64)                                      # noqa: E731
    assert np.array_equal(f(xh) + f(xl), f((xh.astype(np.float32) + xl.astype(np.float32))))
    three = (f(xh) @ f(wh).T + f(xl) @ f(wh).T + f(xh) @ f(wl).T) / s.T     # exact products, exact sums: representation only
    rms = lambda d: float(np.sqrt((d ** 2).mean()))                         # noqa: E731
    e_split = rms(three - truth)
    e_fp32 = rms((x @ w.T).astype(np.float64) - truth)
    out = rms(truth)
    assert e_split < 0.5 * e_fp32, (e_split, e_fp32)                        # measured: ~0.2x
    assert e_split < 2e-7 * out
    # the dropped term is what is missing, and it is tiny
    dropped = (f(xl) @ f(wl).T) / s.T
    full = (f(xh) + f(xl)) @ (f(wh) + f(wl)).T / s.T
    np.testing.assert_allclose(three + dropped, full, rtol=0, atol=1e-9 * out)
    assert rms(dropped) < 2e-7 * out
    # without the per-channel scale the low halves of small weights are fp16 subnormals: several times worse
    wh0, wl0 = split(w)
    three0 = f(xh) @ f(wh0).T + f(xl) @ f(wh0).T + f(xh) @ f(wl0).T
    small = np.argsort(np.abs(w).max(1))[:8]                                # the channels with the smallest weights
    rel = lambda d, cols: float(np.sqrt((d[:, cols] ** 2).mean() / (truth[:, cols] ** 2).mean()))      # noqa: E731
    assert rel(three0 - truth, small) > 4 * rel(three - truth, small)
    assert rms(three0 - truth) > 1.3 * e_split


def test_values_beyond_fp16_range_do_not_pass_silently():
    x = np.array([7.0e4, 1.0], np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        hi, lo = split(x)
        y = hi.astype(np.float32) * 2 + lo.astype(np.float32) * 2
    assert np.isinf(hi[0]) and not np.isfinite(y[0]) and np.isfinite(y[1])      # inf - inf = NaN in the output, loudly


# ---- split-plane tensors (kernels_split.hip, `x_sp` / `d_sp`): layout, epilogue store, LDS-DMA load, fragment read ----------
def to_planes(x):
    """[..., C] float32 -> the bytes of the split-plane tensor viewed as float16 [..., C/32, 64]: 32 hi halves then 32 lo halves
    per 32-channel group, in the 128 B the group's 32 floats would occupy."""
    hi, lo = split(x)
    g = x.shape[:-1] + (x.shape[-1] // 32, 32)
    return np.concatenate([hi.reshape(g), lo.reshape(g)], -1)


def test_split_plane_layout_keeps_offsets_and_holds_hi_plus_lo():
    rng = np.random.RandomState(1)
    x = (rng.standard_normal((5, 96)) * 3).astype(np.float32)
    p = to_planes(x)
    assert p.nbytes == x.nbytes and p.shape == (5, 3, 64)
    # a channel slice at a multiple of 32 starts at the byte offset it has in the fp32 tensor
    assert np.array_equal(to_planes(x[:, 32:64])[:, 0], p[:, 1])
    back = (p[..., :32].astype(np.float32) + p[..., 32:].astype(np.float32)).reshape(5, 96)
    np.testing.assert_allclose(back, x, rtol=2.0 ** -21, atol=2.0 ** -24)       # 22 bits, fp16's subnormal floor below
    hi, lo = split(x)
    assert np.array_equal(back, hi.astype(np.float32) + lo.astype(np.float32))  # exactly what the matrix cores are fed


def test_split_plane_epilogue_store_then_dma_load_feeds_the_right_fragments():
    """One 128-pixel x 64-channel tile through the kernel's index arithmetic: the epilogue's lane -> (pixel, channel) map fills the
    swizzled LDS tile, the store loop writes the split-plane tensor, the LDS-DMA of the consumer (lane -> row / chunk, swizzle applied on the SOURCE chunk) fills the
    two LDS planes, and the fragment reads of the MFMA loop (chunk (2 kk + khalf) ^ swz(row)) see X[m][k]."""
    rng = np.random.RandomState(2)
    SBM, C = 128, 64
    vals = (rng.standard_normal((SBM, C)) * 4).astype(np.float32)
    # -- producer epilogue, phase A: accumulator registers (lane, g, e) of fragment (i = N fragment, j = M fragment) hold
    # channels 32 i + 4 hi + 8 g + e of pixel 32 j + lane % 32 and go to the LDS tile as 16-B chunks, XOR-swizzled by the pixel
    BN = C
    NCH = BN // 4
    stg = np.full((SBM, BN), np.nan, np.float32)
    for j in range(SBM // 32):
        for i in range(C // 32):
            for lane in range(64):
                l31, hi = lane & 31, lane >> 5
                pp = j * 32 + l31
                for g in range(4):
                    nl = i * 32 + 4 * hi + 8 * g
                    c = ((nl >> 2) ^ (pp & (NCH - 1))) << 2
                    stg[pp, c:c + 4] = vals[pp, nl:nl + 4]
    assert not np.isnan(stg).any()
    # -- phase B: thread t, unit u = t + 256 it: pixel u / UR, channels 8 (u % UR) .. + 8; split-plane store
    mem = np.zeros((SBM, C // 32, 64), np.float16)                  # the tensor in HBM, pitch = C floats
    UR = BN // 8
    for u in range(SBM * UR):
        pp, cu = u // UR, u % UR
        sw = pp & (NCH - 1)
        v = np.concatenate([stg[pp, ((2 * cu) ^ sw) * 4:((2 * cu) ^ sw) * 4 + 4], stg[pp, ((2 * cu + 1) ^ sw) * 4:((2 * cu + 1) ^ sw) * 4 + 4]])
        assert np.array_equal(v, vals[pp, 8 * cu: 8 * cu + 8])
        oh = v.astype(np.float16)
        ol = (v - oh.astype(np.float32)).astype(np.float16)
        n = 8 * cu
        grp, w = n >> 5, n & 31                                      # gp = row + (n & ~31) floats + (n & 31) * 2 B; lo plane at + 64 B
        mem[pp, grp, w:w + 8] = oh
        mem[pp, grp, 32 + w:32 + w + 8] = ol
    assert np.array_equal(mem, to_planes(vals))
    # -- consumer: K step `ks` (channels 32 ks ..), thread t stages rows t/4 + 64 i, LDS chunk position t%4
    swz = lambda r: (r >> 2) & 3                                     # noqa: E731
    for ks in range(C // 32):
        lds = np.full((2, SBM, 32), np.nan, np.float16)              # [hi / lo plane][row][32 halves]
        for t in range(256):
            seg = t & 3
            srcc = seg ^ ((t >> 4) & 3)
            for i in range(SBM // 64):
                r = (t >> 2) + 64 * i
                assert swz(r) == (t >> 4) & 3
                wave, lane = t >> 6, t & 63
                flat = (i * 256 + wave * 64) * 8 + lane * 8          # dma(): plane + (i * 256 + wave * 64) * 8 halves, lane-linear 16 B
                assert flat == r * 32 + seg * 8
                lds[0].reshape(-1)[flat:flat + 8] = mem[r, ks, srcc * 8: srcc * 8 + 8]
                lds[1].reshape(-1)[flat:flat + 8] = mem[r, ks, 32 + srcc * 8: 32 + srcc * 8 + 8]
        assert not np.isnan(lds.astype(np.float32)).any()
        xh, xl = split(vals[:, 32 * ks: 32 * ks + 32])
        for j in range(SBM // 32):
            for lane in range(64):
                l31, khalf = lane & 31, lane >> 5
                row = j * 32 + l31
                for kk in range(2):
                    co = ((kk * 2 + khalf) ^ swz(l31)) * 8
                    k0 = (kk * 2 + khalf) * 8                        # the 8 K values this lane owns in this MFMA
                    assert np.array_equal(lds[0, row, co:co + 8], xh[row, k0:k0 + 8])
                    assert np.array_equal(lds[1, row, co:co + 8], xl[row, k0:k0 + 8])
