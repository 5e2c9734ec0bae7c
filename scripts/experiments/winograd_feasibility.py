#!/usr/bin/env python3
"""Numerical go / no-go for Winograd on the MFMA layers, CPU only.

The ConvTranspose 4x4/s2 layers (2.9 of the forward's 6.0 TFLOP) are four stride-1 2x2-tap convolutions, the 3x3
stride-1 layers another 1.2 TFLOP.  Winograd's F(2x2, 2x2) / F(2x2, 3x3) need 9/16 and 16/36 of the multiplications.
What would they cost in accuracy on THIS engine's number formats?  Emulated here with the engine's roundings:
activations fp16, weights fp16 (transformed in float64, rounded once), transformed activations rounded to fp16 (they
would sit in LDS as MFMA operands), products exact, accumulation and the inverse transform in fp32, output rounded
to fp16.  Compared against the float64 result: the direct kernel's error (what the engine has today) next to Winograd's.
"""
import numpy as np

rs = np.random.RandomState(0)
f16 = lambda a: a.astype(np.float16)          # noqa: E731
r32 = lambda a: a.astype(np.float32)          # noqa: E731


def direct_conv(x, w, taps):
    """x (H+th-1, W+tw-1, C) padded input, w (th, tw, C, N): valid correlation, float64 or fp32-accumulated fp16."""
    th, tw = taps
    H, W = x.shape[0] - th + 1, x.shape[1] - tw + 1
    out = np.zeros((H, W, w.shape[3]), x.dtype)
    for i in range(th):
        for j in range(tw):
            out += x[i:i + H, j:j + W] @ w[i, j]
    return out


def winograd(x16, w64, r):
    """F(2x2, rxr), r = 2 or 3.  x16: padded fp16 input (H + r - 1, W + r - 1, C), H, W even.  Returns fp32 (H, W, N)."""
    if r == 3:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    else:
        BT = np.array([[1, -1, 0], [0, 1, 0], [0, 1, -1]], np.float64)
        G = np.array([[1, 0], [1, 1], [0, 1]], np.float64)
        AT = np.array([[1, 1, 0], [0, 1, -1]], np.float64)
    t = BT.shape[0]
    U = f16(np.einsum("ai,ijcn,bj->abcn", G, w64, G))                     # transformed weights, rounded once
    H, W = x16.shape[0] - r + 1, x16.shape[1] - r + 1
    out = np.zeros((H, W, w64.shape[3]), np.float32)
    for y in range(0, H, 2):
        d = r32(x16[y:y + t])                                              # (t, W + r - 1, C)
        for x in range(0, W, 2):
            tile = d[:, x:x + t]                                           # (t, t, C)
            V = f16(np.einsum("ai,ijc,bj->abc", r32(BT), tile, r32(BT)))   # fp32 transform, stored as fp16
            M = np.einsum("abc,abcn->abn", r32(V), r32(U))                 # t*t GEMM points, fp32 accumulation
            out[y:y + 2, x:x + 2] = np.einsum("ia,abn,jb->ijn", r32(AT), M, r32(AT))
    return out


def report(name, x, w, r):
    ref = direct_conv(x.astype(np.float64), w.astype(np.float64), (r, r))
    x16, w16 = f16(x), f16(w)
    d = r32(f16(direct_conv(r32(x16), r32(w16), (r, r))))
    wg = r32(f16(winograd(x16, w.astype(np.float64), r)))
    scale = np.abs(ref).mean()
    e = lambda a: (np.abs(a - ref).max() / scale, np.sqrt(((a - ref) ** 2).mean()) / scale)   # noqa: E731
    ed, ew = e(d), e(wg)
    print(f"{name:34s} |out| mean {scale:.3f}   direct fp16: max {ed[0]:.2e} rms {ed[1]:.2e}   winograd fp16: max {ew[0]:.2e} "
          f"rms {ew[1]:.2e}   ratio rms {ew[1] / ed[1]:.2f}")


if __name__ == "__main__":
    # activations: post-activation statistics of the engine's maps (ReLU-like, O(1)); weights ~ 1 / sqrt(K) as in the
    # folded synthetic checkpoints
    for C, N, S in ((128, 64, 24), (256, 128, 16), (512, 256, 12)):
        x = np.maximum(rs.randn(S + 1, S + 1, C), 0) * 1.2
        w = rs.randn(2, 2, C, N) / np.sqrt(4 * C)
        report(f"ConvT phase 2x2 taps {C}->{N}", x, w, 2)
    for C, N, S in ((32, 32, 24), (128, 128, 16), (256, 256, 12)):
        x = np.maximum(rs.randn(S + 2, S + 2, C), 0) * 1.2
        w = rs.randn(3, 3, C, N) / np.sqrt(9 * C)
        report(f"3x3 {C}->{N}", x, w, 3)
