"""Accuracy of the split-operand product (kernels_split.hip) on the CPU: x and w as fp16 hi + lo, three products, against an
fp32 matmul and float64 -- with and without the per-output-channel power-of-two weight scale.  K = 1152 (3x3, 128 ch)."""
import numpy as np
rng=np.random.default_rng(0)
K=1152; M=4096; N=64
x=rng.standard_normal((M,K)).astype(np.float32)*2
x=(x/(1+np.exp(-x))).astype(np.float32)   # silu-like
w=(rng.standard_normal((N,K))*0.03).astype(np.float32)
truth=x.astype(np.float64)@w.astype(np.float64).T
f32=(x@w.T)
def split(a,scale):
    a=a.astype(np.float32)*np.float32(scale)
    hi=a.astype(np.float16)
    lo=(a-hi.astype(np.float32)).astype(np.float16)
    return hi,lo
for xs in (1,16,256):
  for wsmode in ('none','row'):
    if wsmode=='none': ws=np.ones((N,1),np.float32)
    else: ws=(2.0**np.floor(np.log2(1024/np.abs(w).max(1,keepdims=True)))).astype(np.float32)
    xh,xl=split(x,xs); wh,wl=split(w*ws,1)
    f=lambda a:a.astype(np.float64)
    # products exact in f32; accumulate in f32 (emulate with float32 matmul of exactly-representable?) -> use f64 accumulate + and f32 accumulate separately
    acc64=f(xh)@f(wh).T+f(xl)@f(wh).T+f(xh)@f(wl).T
    acc32=(xh.astype(np.float32)@wh.astype(np.float32).T+xl.astype(np.float32)@wh.astype(np.float32).T+xh.astype(np.float32)@wl.astype(np.float32).T)
    r64=acc64/xs/ws.T; r32=acc32.astype(np.float64)/xs/ws.T
    print(xs,wsmode,'split repr err rms %.3e  with f32 acc rms %.3e | fp32 matmul rms %.3e | out rms %.3f'%(np.sqrt(((r64-truth)**2).mean()),np.sqrt(((r32-truth)**2).mean()),np.sqrt(((f32-truth)**2).mean()),np.sqrt((truth**2).mean())))
