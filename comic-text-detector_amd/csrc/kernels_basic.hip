// Direct (VALU) kernels: the exact-fp32 reference mode of the engine and the
// universal fallback for shapes the MFMA kernels do not cover.  NHWC layout,
// channel index fastest across lanes so weight reads and stores coalesce.
#include "kernels.h"

namespace {

template <typename T>
__device__ __forceinline__ const T* src_pixel(const SrcView& s, int b, int iy, int ix) {
  int sy = s.up ? (iy >> 1) : iy;
  int sx = s.up ? (ix >> 1) : ix;
  return (const T*)s.ptr + ((size_t)((size_t)b * s.H + sy) * s.W + sx) * s.pitch;
}

// out[m][n] = act(bias[n] + sum_{ty,tx,c} in[b, oy*s+dy0+ty, ox*s+dx0+tx, c] * w[(ty*KW+tx)*Ct + c][n]) (+res)
// (reference: nn.Conv2d as used by common.py:30-49 Conv, basemodel.py:91,96,135)
template <typename T, bool PRECISE>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  const long long total = (long long)a.M * a.N;
  const int Ct = a.s0.c + a.s1.c;
  const float* __restrict__ w = (const float*)a.w;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % a.N);
    const int m = (int)(idx / a.N);
    const int ox = m % a.Mw;
    const int oy = (m / a.Mw) % a.Mh;
    const int b = m / (a.Mw * a.Mh);
    float acc = 0.f;
    for (int ty = 0; ty < a.KH; ++ty) {
      const int iy = oy * a.stride + a.dy0 + ty;
      if (iy < 0 || iy >= a.Hin) continue;
      for (int tx = 0; tx < a.KW; ++tx) {
        const int ix = ox * a.stride + a.dx0 + tx;
        if (ix < 0 || ix >= a.Win) continue;
        const float* wp = w + (size_t)((ty * a.KW + tx) * Ct) * a.N + n;
        const T* p0 = src_pixel<T>(a.s0, b, iy, ix);
        for (int c = 0; c < a.s0.c; ++c) acc = fmaf(to_f(p0[c]), wp[(size_t)c * a.N], acc);
        if (a.s1.c) {
          const T* p1 = src_pixel<T>(a.s1, b, iy, ix);
          const float* wq = wp + (size_t)a.s0.c * a.N;
          for (int c = 0; c < a.s1.c; ++c) acc = fmaf(to_f(p1[c]), wq[(size_t)c * a.N], acc);
        }
      }
    }
    if (a.bias) acc += a.bias[n];
    acc = PRECISE ? ctd_act_precise(acc, a.act) : ctd_act(acc, a.act);
    const size_t opix = ((size_t)b * a.oH + (oy * a.osy + a.ooy)) * a.oW + (ox * a.osx + a.oox);
    if (a.res) acc += to_f(((const T*)a.res)[opix * a.pitchR + n]);
    ((T*)a.dst)[opix * a.pitchD + n] = from_f<T>(acc);
  }
}

// Generic ConvTranspose2d (reference basemodel.py:26,58,99,102): a.KH = k, a.stride = s, a.dy0 = pad.
// out[b,oy,ox,n] = bias[n] + sum_{ky,kx,c : oy+p-ky = s*iy} in[b,iy,ix,c] * w[(ky*k+kx)*C + c][n]
template <typename T, bool PRECISE>
__global__ __launch_bounds__(256) void convt_direct_kernel(ConvArgs a) {
  const long long total = (long long)a.B * a.oH * a.oW * a.N;
  const int C = a.s0.c;
  const int k = a.KH, s = a.stride, p = a.dy0;
  const float* __restrict__ w = (const float*)a.w;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx % a.N);
    const long long pix = idx / a.N;
    const int ox = (int)(pix % a.oW);
    const int oy = (int)((pix / a.oW) % a.oH);
    const int b = (int)(pix / ((long long)a.oW * a.oH));
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) {
      const int ty = oy + p - ky;
      if (ty < 0 || (ty % s) != 0) continue;
      const int iy = ty / s;
      if (iy >= a.Hin) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int tx = ox + p - kx;
        if (tx < 0 || (tx % s) != 0) continue;
        const int ix = tx / s;
        if (ix >= a.Win) continue;
        const T* p0 = src_pixel<T>(a.s0, b, iy, ix);
        const float* wp = w + (size_t)((ky * k + kx) * C) * a.N + n;
        for (int c = 0; c < C; ++c) acc = fmaf(to_f(p0[c]), wp[(size_t)c * a.N], acc);
      }
    }
    if (a.bias) acc += a.bias[n];
    acc = PRECISE ? ctd_act_precise(acc, a.act) : ctd_act(acc, a.act);
    ((T*)a.dst)[(size_t)pix * a.pitchD + n] = from_f<T>(acc);
  }
}

// reference inference.py:77-82: the net sees (B,3,H,W) f32 in [0,1]
template <typename T>
__global__ void input_nchw_kernel(const float* __restrict__ in, T* __restrict__ dst, int pitch, int B, int H, int W) {
  const long long hw = (long long)H * W;
  const long long total = (long long)B * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / hw, p = i % hw;
    const float* src = in + b * 3 * hw + p;
    T* d = dst + i * pitch;
    d[0] = from_f<T>(src[0]);
    d[1] = from_f<T>(src[hw]);
    d[2] = from_f<T>(src[2 * hw]);
    for (int c = 3; c < pitch; ++c) d[c] = from_f<T>(0.f);   // padding channels (the f32 MFMA stem reads 4)
  }
}

// u8 page in the channel order the net consumes; x/255 as float32 like
// `astype(np.float32) / 255` (reference inference.py:78)
template <typename T>
__global__ void input_u8_kernel(const uint8_t* __restrict__ in, T* __restrict__ dst, int pitch, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const uint8_t* s = in + i * 3;
    T* d = dst + i * pitch;
    d[0] = from_f<T>((float)s[0] / 255.0f);
    d[1] = from_f<T>((float)s[1] / 255.0f);
    d[2] = from_f<T>((float)s[2] / 255.0f);
    for (int c = 3; c < pitch; ++c) d[c] = from_f<T>(0.f);
  }
}

// nn.MaxPool2d(k, stride 1, pad k/2) (reference common.py:188); padding is -inf
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ src, int pitchS, T* __restrict__ dst, int pitchD, int C,
                               int B, int H, int W, int k) {
  const long long total = (long long)B * H * W * C;
  const int r = k / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long pix = i / C;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    float m = -INFINITY;
    for (int dy = -r; dy <= r; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      for (int dx = -r; dx <= r; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= W) continue;
        m = fmaxf(m, to_f(src[((b * H + yy) * W + xx) * pitchS + c]));
      }
    }
    dst[pix * pitchD + c] = from_f<T>(m);
  }
}

// fp16 fast path of the same pool: one lane = 8 channels (16-B loads / stores)
__global__ void maxpool_h8_kernel(const half_t* __restrict__ src, int pitchS, half_t* __restrict__ dst, int pitchD,
                                  int C8, int B, int H, int W, int k) {
  const long long total = (long long)B * H * W * C8;
  const int r = k / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8) * 8;
    const long long pix = i / C8;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    half8_t m;
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = (half_t)-65504.f;
    for (int dy = -r; dy <= r; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      for (int dx = -r; dx <= r; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= W) continue;
        const half8_t v = *(const half8_t*)(src + ((b * H + yy) * W + xx) * pitchS + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
    *(half8_t*)(dst + pix * pitchD + c) = m;
  }
}

// SPPF's three chained MaxPool2d(5, 1, 2) (reference models/yolov5/common.py:190-196: y1 = m(x), y2 = m(y1),
// y3 = m(y2), written into the channel slots 1..3 of the cat tensor) in ONE launch.  The maps are tiny
// (H/32 x W/32) and the three dependent launches were latency, not bandwidth: 3 x 45 us for 17 MB each.
// A block owns one channel group (V = 8 or 4 fp16 channels) of one page, keeps the whole map in LDS and applies
// the pool as a row pass + a column pass, three times; every level is stored as it appears.  max is exact, so the
// result equals the chained pools bit for bit (out-of-image taps are skipped = -inf padding, as nn.MaxPool2d).
constexpr int SPPF_LDS_BYTES = 65536;
template <typename T, typename V, int NV>
__global__ __launch_bounds__(256) void sppf_pool3_kernel(T* __restrict__ cat, int pitch, int slot, int CG, int H, int W, int r) {
  __shared__ __attribute__((aligned(16))) unsigned char sm_raw[SPPF_LDS_BYTES];
  V* A = (V*)sm_raw;
  V* Bf = A + H * W;
  const int HW = H * W;
  const int b = blockIdx.x / CG, cg = blockIdx.x - b * CG;
  T* base = cat + (size_t)b * HW * pitch + cg * NV;
  for (int p = threadIdx.x; p < HW; p += 256) A[p] = *(const V*)(base + (size_t)p * pitch);
  __syncthreads();
  auto vmax = [](V a, V b2) {
    V o;
#pragma unroll
    for (int e = 0; e < NV; ++e) o[e] = b2[e] > a[e] ? b2[e] : a[e];
    return o;
  };
  for (int lvl = 1; lvl <= 3; ++lvl) {
    for (int p = threadIdx.x; p < HW; p += 256) {        // row pass: A -> Bf
      const int y = p / W, x = p - y * W;
      V m = A[p];
      for (int dx = -r; dx <= r; ++dx)
        if (dx && (unsigned)(x + dx) < (unsigned)W) m = vmax(m, A[p + dx]);
      Bf[p] = m;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {        // column pass: Bf -> A (+ this level's slot)
      const int y = p / W;
      V m = Bf[p];
      for (int dy = -r; dy <= r; ++dy)
        if (dy && (unsigned)(y + dy) < (unsigned)H) m = vmax(m, Bf[p + dy * W]);
      A[p] = m;
      *(V*)(base + (size_t)lvl * slot + (size_t)p * pitch) = m;
    }
    __syncthreads();
  }
}

// nn.AvgPool2d(2, stride=2) (reference basemodel.py:38)
template <typename T>
__global__ void avgpool2_kernel(const T* __restrict__ src, int pitchS, T* __restrict__ dst, int pitchD, int C,
                                int B, int Ho, int Wo) {
  const long long total = (long long)B * Ho * Wo * C;
  const int Wi = Wo * 2, Hi = Ho * 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long pix = i / C;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho);
    const long long b = pix / ((long long)Wo * Ho);
    const T* p = src + ((b * Hi + 2 * y) * Wi + 2 * x) * pitchS + c;
    const float s = (to_f(p[0]) + to_f(p[pitchS])) + (to_f(p[(size_t)Wi * pitchS]) + to_f(p[(size_t)(Wi + 1) * pitchS]));
    dst[pix * pitchD + c] = from_f<T>(s * 0.25f);
  }
}

// reference yolo.py:26-42: view(bs,na,no,ny,nx).permute(0,1,3,4,2); sigmoid;
// xy = (2s - 0.5 + grid) * stride; wh = (2s)^2 * anchor_grid; rows = (a, y, x)
template <typename T>
__global__ void detect_decode_kernel(const T* __restrict__ raw, int pitch, float* __restrict__ blks, int rows_total,
                                     int row_off, int B, int ny, int nx, int na, int no, float stride,
                                     const float* __restrict__ anchors) {
  const long long total = (long long)B * na * ny * nx;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx), y = (int)((i / nx) % ny);
    const int an = (int)((i / ((long long)nx * ny)) % na);
    const long long b = i / ((long long)nx * ny * na);
    const T* p = raw + ((b * ny + y) * nx + x) * pitch + an * no;
    float* o = blks + (b * rows_total + row_off + ((long long)an * ny + y) * nx + x) * no;
    for (int j = 0; j < no; ++j) {
      const float s = 1.0f / (1.0f + expf(-to_f(p[j])));
      float v = s;
      if (j == 0) v = (s * 2.f - 0.5f + (float)x) * stride;
      else if (j == 1) v = (s * 2.f - 0.5f + (float)y) * stride;
      else if (j == 2) { const float t = s * 2.f; v = t * t * anchors[an * 2 + 0]; }
      else if (j == 3) { const float t = s * 2.f; v = t * t * anchors[an * 2 + 1]; }
      o[j] = v;
    }
  }
}

template <typename T>
__global__ void export_plane_kernel(const T* __restrict__ src, int pitch, float* __restrict__ out, int nplanes,
                                    int plane, uint8_t* __restrict__ u8, int u8_mode, float thresh, int B, int H,
                                    int W) {
  const long long hw = (long long)H * W;
  const long long total = (long long)B * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / hw, p = i % hw;
    const float v = to_f(src[i * pitch]);
    if (out) out[(b * nplanes + plane) * hw + p] = v;
    if (u8_mode == 1) u8[i] = (uint8_t)(v * 255.0f);       // reference inference.py:96-99 (truncation)
    else if (u8_mode == 2) u8[i] = v > thresh ? 1 : 0;      // reference db_utils.py:71-72
  }
}

inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  if (g > 256LL * 32) g = 256LL * 32;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

void launch_conv_direct(const ConvArgs& a, bool f16, hipStream_t st) {
  const int g = grid_for((long long)a.M * a.N);
  if (f16) hipLaunchKernelGGL((conv_direct_kernel<half_t, false>), dim3(g), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_direct_kernel<float, true>), dim3(g), dim3(256), 0, st, a);
}

void launch_convt_direct(const ConvArgs& a, bool f16, hipStream_t st) {
  const int g = grid_for((long long)a.B * a.oH * a.oW * a.N);
  if (f16) hipLaunchKernelGGL((convt_direct_kernel<half_t, false>), dim3(g), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((convt_direct_kernel<float, true>), dim3(g), dim3(256), 0, st, a);
}

void launch_input_nchw(const float* in, void* dst, int pitch, int B, int H, int W, bool f16, hipStream_t st) {
  const int g = grid_for((long long)B * H * W);
  if (f16) hipLaunchKernelGGL((input_nchw_kernel<half_t>), dim3(g), dim3(256), 0, st, in, (half_t*)dst, pitch, B, H, W);
  else hipLaunchKernelGGL((input_nchw_kernel<float>), dim3(g), dim3(256), 0, st, in, (float*)dst, pitch, B, H, W);
}

void launch_input_u8(const uint8_t* in, void* dst, int pitch, int B, int H, int W, bool f16, hipStream_t st) {
  const long long t = (long long)B * H * W;
  const int g = grid_for(t);
  if (f16) hipLaunchKernelGGL((input_u8_kernel<half_t>), dim3(g), dim3(256), 0, st, in, (half_t*)dst, pitch, t);
  else hipLaunchKernelGGL((input_u8_kernel<float>), dim3(g), dim3(256), 0, st, in, (float*)dst, pitch, t);
}

void launch_maxpool(const void* src, int pitchS, void* dst, int pitchD, int C, int B, int H, int W, int k,
                    bool f16, hipStream_t st) {
  const int g = grid_for((long long)B * H * W * C);
  if (f16 && C % 8 == 0 && pitchS % 8 == 0 && pitchD % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0)
    hipLaunchKernelGGL(maxpool_h8_kernel, dim3(grid_for((long long)B * H * W * (C / 8))), dim3(256), 0, st,
                       (const half_t*)src, pitchS, (half_t*)dst, pitchD, C / 8, B, H, W, k);
  else if (f16)
    hipLaunchKernelGGL((maxpool_kernel<half_t>), dim3(g), dim3(256), 0, st, (const half_t*)src, pitchS,
                       (half_t*)dst, pitchD, C, B, H, W, k);
  else
    hipLaunchKernelGGL((maxpool_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)src, pitchS, (float*)dst,
                       pitchD, C, B, H, W, k);
}

// cat: slot 0 of the SPPF cat tensor (already offset), slots `slot` channels apart; C channels per slot
bool sppf_pool3_supported(int pitch, int slot, int C, int H, int W, int k, const void* cat, int esize) {
  if (k != 5 && k != 3 && k != 7) return false;
  const int v = 16 / esize;                                // elements of a 16-B entry
  if (C % v || pitch % v || slot % v || ((uintptr_t)cat & 15)) return false;
  return (long long)H * W * 2 * 8 <= SPPF_LDS_BYTES;      // two planes of 8-B entries at least
}
void launch_sppf_pool3(void* cat, int pitch, int slot, int C, int B, int H, int W, int k, hipStream_t st, int esize) {
  typedef float float2_t __attribute__((ext_vector_type(2)));
  const bool wide = (long long)H * W * 2 * 16 <= SPPF_LDS_BYTES;
  if (esize == 4) {
    if (wide)
      hipLaunchKernelGGL((sppf_pool3_kernel<float, float4_t, 4>), dim3(B * (C / 4)), dim3(256), 0, st, (float*)cat, pitch, slot, C / 4, H, W, k / 2);
    else
      hipLaunchKernelGGL((sppf_pool3_kernel<float, float2_t, 2>), dim3(B * (C / 2)), dim3(256), 0, st, (float*)cat, pitch, slot, C / 2, H, W, k / 2);
  } else if (wide) {
    hipLaunchKernelGGL((sppf_pool3_kernel<half_t, half8_t, 8>), dim3(B * (C / 8)), dim3(256), 0, st, (half_t*)cat, pitch, slot, C / 8, H, W, k / 2);
  } else {
    hipLaunchKernelGGL((sppf_pool3_kernel<half_t, half4_t, 4>), dim3(B * (C / 4)), dim3(256), 0, st, (half_t*)cat, pitch, slot, C / 4, H, W, k / 2);
  }
}

void launch_avgpool2(const void* src, int pitchS, void* dst, int pitchD, int C, int B, int Ho, int Wo, bool f16,
                     hipStream_t st) {
  const int g = grid_for((long long)B * Ho * Wo * C);
  if (f16)
    hipLaunchKernelGGL((avgpool2_kernel<half_t>), dim3(g), dim3(256), 0, st, (const half_t*)src, pitchS,
                       (half_t*)dst, pitchD, C, B, Ho, Wo);
  else
    hipLaunchKernelGGL((avgpool2_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)src, pitchS, (float*)dst,
                       pitchD, C, B, Ho, Wo);
}

void launch_detect_decode(const void* raw, int pitch, bool raw_f16, float* blks, int rows_total, int row_off, int B,
                          int ny, int nx, int na, int no, float stride, const float* anchors_px, hipStream_t st) {
  const int g = grid_for((long long)B * na * ny * nx);
  if (raw_f16)
    hipLaunchKernelGGL((detect_decode_kernel<half_t>), dim3(g), dim3(256), 0, st, (const half_t*)raw, pitch, blks,
                       rows_total, row_off, B, ny, nx, na, no, stride, anchors_px);
  else
    hipLaunchKernelGGL((detect_decode_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)raw, pitch, blks,
                       rows_total, row_off, B, ny, nx, na, no, stride, anchors_px);
}

// DBHead.step_function (reference basemodel.py:159-160): 1 / (1 + exp(-k (shrink - thresh))) of the two planes of
// `lines_map`, the map `DBHead.forward(step_eval=True)` returns (basemodel.py:121-122), plus its bitmap (> thresh)
__global__ void db_step_kernel(const float* __restrict__ lines, float k, float* __restrict__ out,
                               uint8_t* __restrict__ bitmap, float thresh, int B, int hw) {
  const long long total = (long long)B * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / hw, p = i % hw;
    const float x = lines[(b * 2) * hw + p], y = lines[(b * 2 + 1) * hw + p];
    const float v = 1.0f / (1.0f + expf(-k * (x - y)));
    out[i] = v;
    if (bitmap) bitmap[i] = v > thresh ? 1 : 0;
  }
}

void launch_db_step(const float* lines, float k, float* out, uint8_t* bitmap, float thresh, int B, int H, int W,
                    hipStream_t st) {
  hipLaunchKernelGGL(db_step_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, st, lines, k, out, bitmap, thresh,
                     B, H * W);
}

void launch_export_plane(const void* src, int pitch, bool f16, float* out, int nplanes, int plane, uint8_t* u8,
                         int u8_mode, float thresh, int B, int H, int W, hipStream_t st) {
  const int g = grid_for((long long)B * H * W);
  if (!u8) u8_mode = 0;
  if (f16)
    hipLaunchKernelGGL((export_plane_kernel<half_t>), dim3(g), dim3(256), 0, st, (const half_t*)src, pitch, out,
                       nplanes, plane, u8, u8_mode, thresh, B, H, W);
  else
    hipLaunchKernelGGL((export_plane_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)src, pitch, out,
                       nplanes, plane, u8, u8_mode, thresh, B, H, W);
}
