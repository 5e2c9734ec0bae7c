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
