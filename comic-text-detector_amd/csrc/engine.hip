// Engine: executes the lowered op program (built by the python host, graph.py)
// on one MI355X.  Responsibilities:
//   * validate the program, choose a kernel per op (MFMA implicit GEMM / fused /
//     direct), repack the f32 parameter blob into the layouts those kernels read;
//   * plan the activation arena for a given (B,H,W) with liveness-based reuse;
//   * launch the ops in order on the caller's stream (no host syncs, so the whole
//     forward can be captured in a hipGraph by the caller);
//   * per-op hipEvent profiling for bench.py's roofline numbers.
// It mirrors `TextDetBase.forward` (reference basemodel.py:240-244) only through
// the program it is given; it has no knowledge of the network itself.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(CTD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                 \
  } while (0)

int g_fuse_epoch = 0;   // bumped when a fusion threshold changes: cached plans are re-made
int g_no_reuse = 0;     // engines created from now on: no arena reuse (debug: `read_tensor` of any activation); "no_reuse"
int g_fwd_prio = 1;      // the network's big kernels raise their waves' issue priority (s_setprio 3): a co-running tail stretched the
                         // VALU-bound stem kernel from 0.42 to 2.1 ms (rocprofv3 timeline); "fwd_prio"
int g_split_stem = 1;    // fp32s engine: the first conv reads the network input itself (kernels_split_stem.hip); "split_stem"
int g_split_planes = 1;  // fp32s engine: keep conv-to-conv tensors split in HBM (0: fp32 everywhere, split in the K loop); "split_planes"
int g_f32_mfma = 1;     // engines created from now on: fp32 convs on the f32 MFMA kernel (0: exact-order direct kernels); "f32_mfma"

enum Impl { IMPL_POINT = 0, IMPL_IGEMM = 1, IMPL_IGEMM_T = 2, IMPL_DIRECT = 3, IMPL_FUSED = 4 };

struct OpState {
  ctd_op op;
  int impl = IMPL_POINT;
  void* w_dev = nullptr;      // packed weights
  float* b_dev = nullptr;     // bias (padded)
  float* aux_dev = nullptr;   // anchors etc.
  int npad = 0;
  int bk = 0;
  bool split = false;         // fp32s engine: split-operand kernel (kernels_split.hip); w_dev = hi plane | lo plane
  int kpad = 0;               // ... and its K (the stem's taps x 4 padded to a multiple of 32)
  float* oscale_dev = nullptr;
  // filled by plan()
  ConvArgs args{};
  double flops = 0, bytes = 0;
  bool c3_head = false;   // this op launches the fused C3 kernel for ops [i, i+3] ...
  bool skip = false;      // ... and the other three launch nothing (also: pools 2, 3 of a fused SPPF)
  bool sppf_head = false; // first of SPPF's three chained max pools: one launch does all three
  bool stem2_head = false; // STEM op that also computes the following 3x3/s2 conv (which is `skip`)
  bool stemsp_head = false; // fp32s: first conv reads the network input itself (kernels_split_stem.hip); its INPUT op is `skip`
  bool segp_head = false; // 64-channel ConvTranspose feeding only SEG_FINAL: stores the 16 tap products per pixel instead (halo3 SEGP)
  bool segp_ran = false;  // (on the SEG_FINAL op, set per launch) its source holds P (B,H,W,16) f32, not the 64-channel map
  bool post_head = false; // 128-channel ConvTranspose whose single consumer, the next op (a 1x1 conv), runs on its output tile
  ConvArgs post_args{};   // ... its arguments with post_* filled (kernels_halo3.hip)
  bool c3b_head = false;  // m.cv1 of a 64 / 128-channel bottleneck: launches kernels_c3b.hip for [m.cv1, m.cv2 (+ cv3)]
  int c3b_conv3 = -1;     // ... index of that bottleneck's 3x3 op (its dispatch decides the K walk, at launch time)
  Stem2Args st2{};
  C3Args c3{};
  C3bArgs c3b{};
};

// A C3 block in the op list: [cv1+cv2 -> Y] ([m.j.cv1: Y[0:c] -> Tj] [m.j.cv2: Tj -> Y[0:c] (+ Y[0:c])]) x n [cv3: Y -> out]
struct C3Chain {
  int a = -1, d = -1, n = 0, c = 0;   // op indices of cv1+cv2 and cv3, bottlenecks, hidden channels
};

struct TensorState {
  ctd_tensor t;
  int esize = 2;
  size_t offset = 0;   // arena offset (bytes)
  size_t bytes = 0;
  int H = 0, W = 0;
  int first_def = -1, last_use = -1;
  bool sp = false;     // fp32s engine: stored split-plane (kernels_split.hip) -- every writer and reader is the split kernel
};

}  // namespace

struct ctd_engine {
  int device = 0;
  int prec = CTD_PREC_F16;
  std::vector<TensorState> tensors;
  std::vector<OpState> ops;
  std::vector<void*> owned;  // device allocations to free
  char* arena = nullptr;
  size_t arena_bytes = 0;
  int arena_gen = 0;           // bumped whenever the arena is reallocated: captured graphs hold the old addresses
  int pB = 0, pH = 0, pW = 0;  // current plan
  int p_fuse = -1;             // value of g_fuse the current plan was made under
  bool no_reuse = false;
  bool w_tiled = true;   // tile-major MFMA weight packing (CTD_W_TILED=0 disables)
  bool f32_mfma = true;  // fp32 engine: f32-operand MFMA kernel (CTD_F32_MFMA=0: exact-order direct kernels only)
  int det_rows_per_unit = 0;
  int det_no = 0;
  void* zeros = nullptr;  // CTD_ZEROS_BYTES of zeros (padding source of the LDS-DMA loads)
};

namespace {

template <typename T>
int upload(ctd_engine* e, const std::vector<T>& host, void** out) {
  void* d = nullptr;
  HIP_TRY(hipMalloc(&d, std::max<size_t>(host.size() * sizeof(T), 16)));
  HIP_TRY(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  e->owned.push_back(d);
  *out = d;
  return CTD_OK;
}

bool tensor_ok(const ctd_engine* e, int id) { return id >= 0 && id < (int)e->tensors.size(); }

// ---- weight packing --------------------------------------------------------
int pack_op(ctd_engine* e, OpState& s, const float* P, int64_t nP) {
  const ctd_op& o = s.op;
  const bool f16 = e->prec == CTD_PREC_F16;
  auto need = [&](int64_t off, int64_t n) { return off >= 0 && off + n <= nP; };
  const int cin = o.src0_c + (o.src1 >= 0 ? o.src1_c : 0);
  const int N = o.cout, k = o.k;

  auto pack_bias = [&](int npad) -> int {
    std::vector<float> b(npad, 0.f);
    if (o.b_off >= 0) {
      if (!need(o.b_off, N)) return fail(CTD_ERR_INVALID, "bias out of range");
      std::memcpy(b.data(), P + o.b_off, sizeof(float) * N);
    }
    return upload(e, b, (void**)&s.b_dev);
  };

  switch (o.kind) {
    case CTD_OP_CONV: {
      if (!need(o.w_off, (int64_t)N * cin * k * k)) return fail(CTD_ERR_INVALID, "conv weights out of range");
      const float* W = P + o.w_off;  // (N, cin, k, k)
      // can the MFMA kernel take it?  (channel counts multiple of 32, aligned pitches)
      bool ig = f16 && (o.src0_c % 32 == 0) && (o.src1 < 0 || o.src1_c % 32 == 0);
      if (ig) {
        const ctd_tensor& t0 = e->tensors[o.src0].t;
        const ctd_tensor& td = e->tensors[o.dst].t;
        if (t0.channels % 8 || o.src0_coff % 8) ig = false;
        if (o.src1 >= 0 && (e->tensors[o.src1].t.channels % 8 || o.src1_coff % 8)) ig = false;
        if (td.channels % 4 || o.dst_coff % 4) ig = false;
        if (o.res >= 0 && (e->tensors[o.res].t.channels % 4 || o.res_coff % 4)) ig = false;
        if (t0.dtype != 0 || (o.src1 >= 0 && e->tensors[o.src1].t.dtype != 0)) ig = false;
        if (o.res >= 0 && e->tensors[o.res].t.dtype != 0) ig = false;
      }
      if (ig) {
        s.impl = IMPL_IGEMM;
        const int bn = igemm_ntile(N);
        s.npad = (N + bn - 1) / bn * bn;
        const int K = k * k * cin;
        s.bk = igemm_pick_bk(o.src0_c, o.src1 >= 0 ? o.src1_c : 0, K, N, e->tensors[o.dst].t.log2_down);
        std::vector<float> lg((size_t)N * K);
        for (int n = 0; n < N; ++n)
          for (int c = 0; c < cin; ++c)
            for (int ky = 0; ky < k; ++ky)
              for (int kx = 0; kx < k; ++kx)
                lg[(size_t)n * K + (size_t)(ky * k + kx) * cin + c] = W[(((size_t)n * cin + c) * k + ky) * k + kx];
        std::vector<half_t> wp;
        igemm_pack_weights(lg.data(), 1, N, K, bn, s.bk, e->w_tiled, wp);
        if (int rc = upload(e, wp, &s.w_dev)) return rc;
        return pack_bias(s.npad);
      }
      if (e->tensors[o.dst].t.dtype != 0 && f16)
        return fail(CTD_ERR_UNSUPPORTED, "f32 destination needs the MFMA path");
      // fp32 engine: f32-operand MFMA (kernels_f32.hip) when a K step of 16 channels never crosses a tap
      auto al4 = [&](int id, int coff) { return e->tensors[id].t.channels % 4 == 0 && coff % 4 == 0; };
      const bool ch_ok = (o.src0_c % 16 == 0 && (o.src1 < 0 || o.src1_c % 16 == 0)) ||
                         (o.src0_c == 4 && o.src1 < 0 && (k * k) % 4 == 0);     // 4-channel source: one tap per 16-B chunk
      if (!f16 && e->f32_mfma && ch_ok && al4(o.src0, o.src0_coff) &&
          (o.src1 < 0 || al4(o.src1, o.src1_coff)) && al4(o.dst, o.dst_coff) && (o.res < 0 || al4(o.res, o.res_coff))) {
        s.impl = IMPL_IGEMM;
        const int bn = f32_mfma_ntile(N);
        s.npad = (N + bn - 1) / bn * bn;
        const int K = k * k * cin;
        // fp32s engine: split operands on the fp16 MFMA wherever a 32-channel K step never crosses a tap / source
        const bool stem4 = o.src0_c == 4 && o.src1 < 0;          // zero-padded image: K = taps x 4
        s.split = e->prec == CTD_PREC_F32S && (stem4 || (o.src0_c % 32 == 0 && (o.src1 < 0 || o.src1_c % 32 == 0)));
        // the split kernel walks K in steps of 32: the stem's K = 144 is padded with zero taps
        const int Kp = s.split ? (K + 31) / 32 * 32 : K;
        s.kpad = Kp;
        std::vector<float> wp((size_t)(s.split ? N : s.npad) * Kp, 0.f);
        for (int n = 0; n < N; ++n)
          for (int c = 0; c < cin; ++c)
            for (int ky = 0; ky < k; ++ky)
              for (int kx = 0; kx < k; ++kx)
                wp[(size_t)n * Kp + (size_t)(ky * k + kx) * cin + c] = W[(((size_t)n * cin + c) * k + ky) * k + kx];
        if (s.split) {
          std::vector<half_t> ws;
          std::vector<float> osc;
          split_pack_weights(wp.data(), 1, N, Kp, s.npad, ws, osc);
          if (int rc = upload(e, ws, &s.w_dev)) return rc;
          if (int rc = upload(e, osc, (void**)&s.oscale_dev)) return rc;
          return pack_bias(s.npad);
        }
        if (int rc = upload(e, wp, &s.w_dev)) return rc;
        return pack_bias(s.npad);
      }
      s.impl = IMPL_DIRECT;
      s.npad = N;
      std::vector<float> wp((size_t)k * k * cin * N);
      for (int n = 0; n < N; ++n)
        for (int c = 0; c < cin; ++c)
          for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
              wp[((size_t)(ky * k + kx) * cin + c) * N + n] = W[(((size_t)n * cin + c) * k + ky) * k + kx];
      if (int rc = upload(e, wp, &s.w_dev)) return rc;
      return pack_bias(N);
    }
    case CTD_OP_CONVT: {
      if (o.src1 >= 0) return fail(CTD_ERR_UNSUPPORTED, "convT takes one source");
      if (!need(o.w_off, (int64_t)N * cin * k * k)) return fail(CTD_ERR_INVALID, "convT weights out of range");
      const float* W = P + o.w_off;  // (cin, N, k, k)
      const ctd_tensor& t0 = e->tensors[o.src0].t;
      const ctd_tensor& td = e->tensors[o.dst].t;
      const bool ig = f16 && k == 4 && o.stride == 2 && o.pad == 1 && cin % 32 == 0 && t0.channels % 8 == 0 &&
                      o.src0_coff % 8 == 0 && td.channels % 4 == 0 && o.dst_coff % 4 == 0 && t0.dtype == 0 &&
                      td.dtype == 0 && N >= 32;
      if (ig) {
        s.impl = IMPL_IGEMM_T;
        const int bn = igemm_ntile(N);
        s.npad = (N + bn - 1) / bn * bn;
        const int K = 4 * cin;
        s.bk = igemm_pick_bk(cin, 0, K, N, e->tensors[o.src0].t.log2_down);
        std::vector<float> lg((size_t)4 * N * K);
        for (int ph = 0; ph < 4; ++ph) {
          const int py = ph >> 1, px = ph & 1;
          for (int ty = 0; ty < 2; ++ty)
            for (int tx = 0; tx < 2; ++tx) {
              const int dy = (py ? 0 : -1) + ty, dx = (px ? 0 : -1) + tx;
              const int ky = py + 1 - 2 * dy, kx = px + 1 - 2 * dx;
              for (int n = 0; n < N; ++n)
                for (int c = 0; c < cin; ++c)
                  lg[((size_t)ph * N + n) * K + (size_t)(ty * 2 + tx) * cin + c] =
                      W[(((size_t)c * N + n) * 4 + ky) * 4 + kx];
            }
        }
        std::vector<half_t> wp;
        igemm_pack_weights(lg.data(), 4, N, K, bn, s.bk, e->w_tiled, wp);
        if (int rc = upload(e, wp, &s.w_dev)) return rc;
        return pack_bias(s.npad);
      }
      if (!f16 && e->f32_mfma && k == 4 && o.stride == 2 && o.pad == 1 && cin % 16 == 0 &&
          t0.channels % 4 == 0 && o.src0_coff % 4 == 0 && (N < 4 || (td.channels % 4 == 0 && o.dst_coff % 4 == 0))) {
        s.impl = IMPL_IGEMM_T;
        const int bn = f32_mfma_ntile(N);
        s.npad = (N + bn - 1) / bn * bn;
        const int K = 4 * cin;
        s.split = e->prec == CTD_PREC_F32S && cin % 32 == 0;
        const int rows = s.split ? N : s.npad;          // the split packer pads by itself
        std::vector<float> wp((size_t)4 * rows * K, 0.f);
        for (int ph = 0; ph < 4; ++ph) {
          const int py = ph >> 1, px = ph & 1;
          for (int ty = 0; ty < 2; ++ty)
            for (int tx = 0; tx < 2; ++tx) {
              const int dy = (py ? 0 : -1) + ty, dx = (px ? 0 : -1) + tx;
              const int ky = py + 1 - 2 * dy, kx = px + 1 - 2 * dx;
              for (int n = 0; n < N; ++n)
                for (int c = 0; c < cin; ++c)
                  wp[((size_t)ph * rows + n) * K + (size_t)(ty * 2 + tx) * cin + c] =
                      W[(((size_t)c * N + n) * 4 + ky) * 4 + kx];
            }
        }
        if (s.split) {
          std::vector<half_t> ws;
          std::vector<float> osc;
          split_pack_weights(wp.data(), 4, N, K, s.npad, ws, osc);
          if (int rc = upload(e, ws, &s.w_dev)) return rc;
          if (int rc = upload(e, osc, (void**)&s.oscale_dev)) return rc;
          return pack_bias(s.npad);
        }
        if (int rc = upload(e, wp, &s.w_dev)) return rc;
        return pack_bias(s.npad);
      }
      s.impl = IMPL_DIRECT;
      s.npad = N;
      std::vector<float> wp((size_t)k * k * cin * N);
      for (int c = 0; c < cin; ++c)
        for (int n = 0; n < N; ++n)
          for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx)
              wp[((size_t)(ky * k + kx) * cin + c) * N + n] = W[(((size_t)c * N + n) * k + ky) * k + kx];
      if (int rc = upload(e, wp, &s.w_dev)) return rc;
      return pack_bias(N);
    }
    case CTD_OP_STEM: {
      if (!f16) return fail(CTD_ERR_UNSUPPORTED, "STEM op is fp16-path only; emit INPUT + CONV for fp32");
      if (N != 32 || k != 6 || o.stride != 2 || o.pad != 2)
        return fail(CTD_ERR_UNSUPPORTED, "fused stem is 6x6/s2/p2, 3->32 only");
      if (!need(o.w_off, (int64_t)N * 3 * 36)) return fail(CTD_ERR_INVALID, "stem weights out of range");
      const float* W = P + o.w_off;
      std::vector<half_t> wp;
      stem_pack_weights(W, wp);
      s.impl = IMPL_FUSED;
      if (int rc = upload(e, wp, &s.w_dev)) return rc;
      return pack_bias(N);
    }
    case CTD_OP_SEG_FINAL: {
      if (cin != 64 || N != 1) return fail(CTD_ERR_UNSUPPORTED, "fused seg-final is 64->1 only");
      if (!need(o.w_off, (int64_t)cin * 16)) return fail(CTD_ERR_INVALID, "seg-final weights out of range");
      const float* W = P + o.w_off;  // (cin, 1, 4, 4)
      if (!f16) {                    // exact-fp32 engine: the checkpoint layout as it is
        std::vector<float> wf(W, W + (size_t)cin * 16);
        s.impl = IMPL_FUSED;
        if (int rc = upload(e, wf, &s.w_dev)) return rc;
        return pack_bias(1);
      }
      // fp16, [cin/8][16 taps][8 channels]: one 8-channel group of all taps = 64 dwords of scalar loads
      std::vector<half_t> wp((size_t)16 * cin);
      for (int c = 0; c < cin; ++c)
        for (int kk = 0; kk < 16; ++kk) wp[((size_t)(c / 8) * 16 + kk) * 8 + (c % 8)] = (half_t)W[(size_t)c * 16 + kk];
      s.impl = IMPL_FUSED;
      if (int rc = upload(e, wp, &s.w_dev)) return rc;
      return pack_bias(1);
    }
    case CTD_OP_DB_UP: {
      const int q = o.aux[1];
      if (q != 16) return fail(CTD_ERR_UNSUPPORTED, "fused db tail is q=16 only");
      const int nbr = o.aux[2] > 0 ? o.aux[2] : 2;                 // branches lowered (1 = shrink map only)
      if (nbr > 2) return fail(CTD_ERR_INVALID, "db-up: at most two branches");
      const int PB = q * q * 4 + q + q * 4 + 1;
      if (!need(o.w_off, nbr * PB)) return fail(CTD_ERR_INVALID, "db-up params out of range");
      // blob per branch: W1 (c,o,py,px), b1 (o), W2 (o,0,qy,qx), b2  ->  device layout of
      // kernels_fused.hip DbUpLayout: W1p[pp][c][o], b1, W2p[qq][o], b2, padded to x4 floats
      const int SIZE = (4 * q * q + q + 4 * q + 1 + 3) / 4 * 4;
      std::vector<float> wp((size_t)2 * SIZE, 0.f);
      for (int br = 0; br < nbr; ++br) {
        const float* src = P + o.w_off + (size_t)br * PB;
        float* dst = wp.data() + (size_t)br * SIZE;
        for (int c = 0; c < q; ++c)
          for (int oo = 0; oo < q; ++oo)
            for (int pp = 0; pp < 4; ++pp) dst[(pp * q + c) * q + oo] = src[(c * q + oo) * 4 + pp];
        for (int oo = 0; oo < q; ++oo) dst[4 * q * q + oo] = src[4 * q * q + oo];
        for (int oo = 0; oo < q; ++oo)
          for (int qq = 0; qq < 4; ++qq) dst[4 * q * q + q + qq * q + oo] = src[4 * q * q + q + oo * 4 + qq];
        dst[4 * q * q + q + 4 * q] = src[4 * q * q + q + 4 * q];
      }
      s.impl = IMPL_FUSED;
      return upload(e, wp, &s.w_dev);
    }
    case CTD_OP_DETECT: {
      const int na = o.aux[2];
      if (na < 1 || na > 4) return fail(CTD_ERR_INVALID, "detect: na must be 1..4");
      std::vector<float> a(o.faux, o.faux + 2 * na);
      return upload(e, a, (void**)&s.aux_dev);
    }
    default:
      return CTD_OK;
  }
}

int validate(ctd_engine* e) {
  const int nT = (int)e->tensors.size();
  for (auto& t : e->tensors) {
    if (t.t.channels < 1 || t.t.log2_down < 0 || t.t.log2_down > 6)
      return fail(CTD_ERR_INVALID, "tensor: bad channels/log2_down");
  }
  for (size_t i = 0; i < e->ops.size(); ++i) {
    const ctd_op& o = e->ops[i].op;
    auto bad = [&](const char* m) { return fail(CTD_ERR_INVALID, "op " + std::to_string(i) + ": " + m); };
    auto slice_ok = [&](int id, int coff, int c) {
      return tensor_ok(e, id) && coff >= 0 && c >= 1 && coff + c <= e->tensors[id].t.channels;
    };
    switch (o.kind) {
      case CTD_OP_INPUT:
        if (!slice_ok(o.dst, 0, 3)) return bad("input dst");
        break;
      case CTD_OP_STEM:
        if (!slice_ok(o.dst, o.dst_coff, o.cout)) return bad("stem dst");
        break;
      case CTD_OP_CONV:
      case CTD_OP_CONVT:
        if (!slice_ok(o.src0, o.src0_coff, o.src0_c)) return bad("src0");
        if (o.src1 >= 0 && !slice_ok(o.src1, o.src1_coff, o.src1_c)) return bad("src1");
        if (o.res >= 0 && !slice_ok(o.res, o.res_coff, o.cout)) return bad("res");
        if (!slice_ok(o.dst, o.dst_coff, o.cout)) return bad("dst");
        if (o.k < 1 || o.k > 7 || o.stride < 1 || o.stride > 2) return bad("k/stride");
        break;
      case CTD_OP_MAXPOOL:
      case CTD_OP_AVGPOOL2:
        if (!slice_ok(o.src0, o.src0_coff, o.src0_c) || !slice_ok(o.dst, o.dst_coff, o.src0_c)) return bad("pool");
        break;
      case CTD_OP_DETECT:
        if (!slice_ok(o.src0, o.src0_coff, o.aux[2] * o.aux[3])) return bad("detect src");
        break;
      case CTD_OP_EXPORT:
        if (!slice_ok(o.src0, o.src0_coff, 1)) return bad("export src");
        break;
      case CTD_OP_SEG_FINAL:
      case CTD_OP_DB_UP:
        if (!slice_ok(o.src0, o.src0_coff, o.src0_c)) return bad("fused src");
        break;
      default:
        return bad("unknown op kind");
    }
  }
  (void)nT;
  return CTD_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- arena planning -----------------------------------------------------------
int plan(ctd_engine* e, int B, int H, int W, hipStream_t cap_stream = nullptr) {
  if (B < 1 || H < 64 || W < 64 || H % 64 || W % 64)
    return fail(CTD_ERR_INVALID, "H and W must be positive multiples of 64 and B >= 1");
  const int nT = (int)e->tensors.size(), nO = (int)e->ops.size();
  for (auto& t : e->tensors) {
    t.esize = (e->prec != CTD_PREC_F16 || t.t.dtype == 1) ? 4 : 2;
    t.H = H >> t.t.log2_down;
    t.W = W >> t.t.log2_down;
    t.bytes = align_up((size_t)B * t.H * t.W * t.t.channels * t.esize, 256);
    t.first_def = -1;
    t.last_use = -1;
    // the MFMA kernel addresses a tensor with signed 32-bit byte offsets
    if (e->prec == CTD_PREC_F16 && t.bytes >= (1ull << 31))
      return fail(CTD_ERR_UNSUPPORTED, "an activation tensor would exceed 2 GiB; split the batch (B*H*W*64 B < 2^31)");
  }
  auto use = [&](int id, int i) {
    if (id >= 0) e->tensors[id].last_use = std::max(e->tensors[id].last_use, i);
  };
  auto def = [&](int id, int i) {
    if (id >= 0) {
      if (e->tensors[id].first_def < 0) e->tensors[id].first_def = i;
      e->tensors[id].last_use = std::max(e->tensors[id].last_use, i);
    }
  };
  for (int i = 0; i < nO; ++i) {
    const ctd_op& o = e->ops[i].op;
    switch (o.kind) {
      case CTD_OP_INPUT:
      case CTD_OP_STEM: def(o.dst, i); break;
      case CTD_OP_CONV:
      case CTD_OP_CONVT:
        use(o.src0, i); if (o.src1 >= 0) use(o.src1, i); if (o.res >= 0) use(o.res, i); def(o.dst, i); break;
      case CTD_OP_MAXPOOL:
      case CTD_OP_AVGPOOL2: use(o.src0, i); def(o.dst, i); break;
      default: use(o.src0, i); break;
    }
  }
  for (int t = 0; t < nT; ++t)
    if (e->tensors[t].last_use >= 0 && e->tensors[t].first_def < 0)
      return fail(CTD_ERR_INVALID, "tensor " + std::to_string(t) + " is read but never written");

  // ---- C3 blocks with 64 / 128 hidden channels (kernels_c3b.hip): found here, BEFORE the arena is laid out, because the
  // fused launches change what is alive when.  Bottleneck j < n - 1 becomes one launch at m.j.cv1's position that reads the
  // block's running tensor and writes Tj (NOT Y[0:c] in place: neighbouring patches still read its halo), so Tj lives until
  // the next bottleneck's launch has read it; the last bottleneck + cv3 become one launch at m.(n-1).cv1's position, so
  // cv3's output exists from there on.  Both changes only lengthen lifetimes: a chain the kernel then declines (grid too
  // small) runs layer per launch on the same arena.
  std::vector<C3Chain> chains;
  // (only under the pattern's fuse bit: with it off the lifetimes -- and the 32-channel C3 matcher, which wants a cv1 | cv2
  // output that is still first defined by its own op -- see the unfused program; a change of `fuse` re-plans)
  for (int i = 0; e->prec == CTD_PREC_F16 && (g_fuse & 8) && i + 3 < nO; ++i) {
    const OpState& A = e->ops[i];
    const ctd_op& a = A.op;
    auto mfma16 = [&](const OpState& s) { return s.op.kind == CTD_OP_CONV && s.impl == IMPL_IGEMM && s.bk == 32 && e->w_tiled; };
    if (!mfma16(A) || a.k != 1 || a.stride != 1 || a.res >= 0 || a.dst_coff != 0) continue;
    const int Y = a.dst, c = a.cout / 2;
    if (a.cout % 2 || !(c == 64 || c == 128) || e->tensors[Y].t.channels != 2 * c || e->tensors[Y].first_def != i) continue;
    int j = i + 1, n = 0;
    bool ok = true;
    while (ok && j + 1 < nO) {
      const OpState &Bo = e->ops[j], &C = e->ops[j + 1];
      const ctd_op &b = Bo.op, &cc = C.op;
      if (!mfma16(Bo) || !mfma16(C)) break;
      const int T = b.dst;
      if (b.k != 1 || b.stride != 1 || b.src0 != Y || b.src0_coff != 0 || b.src0_c != c || b.src0_up || b.src1 >= 0 ||
          b.cout != c || b.res >= 0 || b.dst_coff != 0 || T == Y || e->tensors[T].t.channels != c || b.act != a.act)
        break;
      if (cc.k != 3 || cc.stride != 1 || cc.pad != 1 || cc.src0 != T || cc.src0_coff != 0 || cc.src0_c != c || cc.src0_up ||
          cc.src1 >= 0 || cc.cout != c || cc.dst != Y || cc.dst_coff != 0 || cc.act != a.act ||
          !(cc.res < 0 || (cc.res == Y && cc.res_coff == 0)))
        break;
      if (e->tensors[T].first_def != j || e->tensors[T].last_use != j + 1 || e->tensors[T].esize != 2) break;
      ++n;
      j += 2;
    }
    if (!n || j >= nO) continue;
    const OpState& D = e->ops[j];
    const ctd_op& d = D.op;
    if (!mfma16(D) || d.k != 1 || d.stride != 1 || d.src0 != Y || d.src0_coff != 0 || d.src0_c != 2 * c || d.src0_up ||
        d.src1 >= 0 || d.cout != 2 * c || d.res >= 0 || d.dst == Y || d.act != a.act)
      continue;
    if (e->tensors[Y].last_use != j || e->tensors[Y].esize != 2 || e->tensors[d.dst].esize != 2) continue;
    bool t_is_dst = false;
    for (int q = 0; q < n; ++q) t_is_dst = t_is_dst || e->ops[i + 1 + 2 * q].op.dst == d.dst;
    if (t_is_dst) continue;
    C3Chain ch;
    ch.a = i; ch.d = j; ch.n = n; ch.c = c;
    chains.push_back(ch);
    for (int q = 0; q + 1 < n; ++q) {   // Tq is read by the launch of bottleneck q + 1
      TensorState& tq = e->tensors[e->ops[i + 1 + 2 * q].op.dst];
      tq.last_use = std::max(tq.last_use, i + 1 + 2 * (q + 1));
    }
    TensorState& td = e->tensors[d.dst];   // written by the launch at the last bottleneck's m.cv1
    td.first_def = std::min(td.first_def, i + 1 + 2 * (n - 1));
    i = j;
  }
  // (after the C3 chains: their match wants the 1x1's output still defined by the 1x1)
  // ---- a 128-channel ConvTranspose followed by its ONLY consumer, a 1x1 conv over it (or over [x ; it], x <= 64 channels):
  // one launch of kernels_halo3.hip writes the 1x1's output, at the ConvTranspose's position -- so that tensor is alive
  // from there on.  (UNet: upconv4.conv.1 -> upconv5.conv.0.cv1+cv2 over [f160 ; u160]; DB head: upconv4.conv.1 -> conv.0.)
  std::vector<int> posts;
  for (int i = 0; e->prec == CTD_PREC_F16 && (g_fuse & 16) && i + 1 < nO; ++i) {
    const OpState &T = e->ops[i], &P = e->ops[i + 1];
    const ctd_op &t = T.op, &p = P.op;
    if (t.kind != CTD_OP_CONVT || T.impl != IMPL_IGEMM_T || t.cout != 128 || t.dst_coff != 0 || T.bk != 32 || !e->w_tiled) continue;
    const int U = t.dst;
    if (e->tensors[U].t.channels != 128 || e->tensors[U].first_def != i || e->tensors[U].last_use != i + 1 || e->tensors[U].esize != 2)
      continue;
    if (p.kind != CTD_OP_CONV || P.impl != IMPL_IGEMM || P.bk != 32 || p.k != 1 || p.stride != 1 || p.res >= 0) continue;
    if (!(p.cout == 64 || p.cout == 128) || P.npad != p.cout || e->tensors[p.dst].esize != 2 || p.dst == U) continue;
    const bool alone = p.src0 == U && p.src0_coff == 0 && p.src0_c == 128 && !p.src0_up && p.src1 < 0;
    const bool second = p.src1 == U && p.src1_coff == 0 && p.src1_c == 128 && !p.src1_up && p.src0 != U && !p.src0_up &&
                        p.src0_c % 32 == 0 && p.src0_c <= 64 && e->tensors[p.src0].esize == 2 &&
                        e->tensors[p.src0].first_def >= 0 && e->tensors[p.src0].first_def < i;
    if (!alone && !second) continue;
    posts.push_back(i);
    TensorState& td = e->tensors[p.dst];
    td.first_def = std::min(td.first_def, i);
  }

  // first-fit allocation over a free list, in op order
  struct Blk { size_t off, size; };
  std::vector<Blk> freel;
  size_t top = 0;
  auto alloc = [&](size_t sz) {
    if (!e->no_reuse) {
      for (size_t i = 0; i < freel.size(); ++i)
        if (freel[i].size >= sz) {
          const size_t off = freel[i].off;
          freel[i].off += sz;
          freel[i].size -= sz;
          if (!freel[i].size) freel.erase(freel.begin() + i);
          return off;
        }
    }
    const size_t off = top;
    top += sz;
    return off;
  };
  auto release = [&](size_t off, size_t sz) {
    freel.push_back({off, sz});
    std::sort(freel.begin(), freel.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
    for (size_t i = 0; i + 1 < freel.size();) {
      if (freel[i].off + freel[i].size == freel[i + 1].off) {
        freel[i].size += freel[i + 1].size;
        freel.erase(freel.begin() + i + 1);
      } else ++i;
    }
  };
  for (int i = 0; i < nO; ++i) {
    for (int t = 0; t < nT; ++t)
      if (e->tensors[t].first_def == i) e->tensors[t].offset = alloc(e->tensors[t].bytes);
    for (int t = 0; t < nT; ++t)
      if (e->tensors[t].first_def >= 0 && e->tensors[t].last_use == i) release(e->tensors[t].offset, e->tensors[t].bytes);
  }
  if (top > e->arena_bytes) {
    // growing the arena frees the old one: illegal inside a stream capture, and it invalidates every graph
    // captured before (their kernels hold the old addresses; `ctd_engine_arena_generation` lets callers check)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (cap_stream && hipStreamIsCapturing(cap_stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
      return fail(CTD_ERR_INVALID, "the activation arena must grow for this (B,H,W), which cannot happen inside a stream "
                                   "capture: run one eager forward of the largest shape first");
    HIP_TRY(hipDeviceSynchronize());
    ++e->arena_gen;
    if (e->arena) HIP_TRY(hipFree(e->arena));
    e->arena = nullptr;
    e->arena_bytes = 0;
    hipError_t me = hipMalloc((void**)&e->arena, top);
    if (me != hipSuccess) return fail(CTD_ERR_NOMEM, "arena hipMalloc of " + std::to_string(top) + " bytes failed");
    e->arena_bytes = top;
  }

  // fp32s: which tensors live split-plane.  A tensor qualifies when every op that writes it and every op that reads it
  // is a launch of the split kernel on 32-channel-aligned slices; the two sources of one launch must agree.
  for (auto& t : e->tensors) t.sp = false;
  if (e->prec == CTD_PREC_F32S && g_split_planes) {
    std::vector<char> sp(nT, 0);
    for (int t = 0; t < nT; ++t) sp[t] = e->tensors[t].esize == 4 && e->tensors[t].t.dtype != 1 && e->tensors[t].t.channels % 32 == 0;
    auto split_conv = [&](const OpState& s) {
      return s.split && ((s.op.kind == CTD_OP_CONV && s.impl == IMPL_IGEMM) || (s.op.kind == CTD_OP_CONVT && s.impl == IMPL_IGEMM_T));
    };
    for (int i = 0; i < nO; ++i) {
      const OpState& s = e->ops[i];
      const ctd_op& o = s.op;
      const bool sc = split_conv(s);
      switch (o.kind) {
        case CTD_OP_CONV:
        case CTD_OP_CONVT:
          if (!(sc && o.src0_coff % 32 == 0 && o.src0_c % 32 == 0)) sp[o.src0] = 0;
          if (o.src1 >= 0 && !(sc && o.src1_coff % 32 == 0 && o.src1_c % 32 == 0)) sp[o.src1] = 0;
          if (o.res >= 0 && !(sc && o.res_coff % 32 == 0 && o.cout % 32 == 0)) sp[o.res] = 0;
          if (!(sc && o.dst_coff % 32 == 0 && o.cout % 32 == 0)) sp[o.dst] = 0;
          break;
        case CTD_OP_INPUT:
        case CTD_OP_STEM: sp[o.dst] = 0; break;
        case CTD_OP_MAXPOOL:
        case CTD_OP_AVGPOOL2: sp[o.src0] = 0; sp[o.dst] = 0; break;
        default: if (o.src0 >= 0) sp[o.src0] = 0; break;
      }
    }
    for (bool changed = true; changed;) {
      changed = false;
      for (int i = 0; i < nO; ++i) {
        const ctd_op& o = e->ops[i].op;
        if ((o.kind == CTD_OP_CONV || o.kind == CTD_OP_CONVT) && o.src1 >= 0 && sp[o.src0] != sp[o.src1]) {
          sp[o.src0] = sp[o.src1] = 0;
          changed = true;
        }
      }
    }
    for (int t = 0; t < nT; ++t) e->tensors[t].sp = sp[t] != 0;
  }

  // per-op launch arguments
  const bool f16 = e->prec == CTD_PREC_F16;
  for (int i = 0; i < nO; ++i) {
    OpState& s = e->ops[i];
    const ctd_op& o = s.op;
    ConvArgs a{};
    auto view = [&](int id, int coff, int c, int up) {
      SrcView v{};
      const TensorState& t = e->tensors[id];
      v.ptr = e->arena + t.offset + (size_t)coff * t.esize;
      v.pitch = t.t.channels;
      v.c = c;
      v.up = up;
      v.H = t.H;
      v.W = t.W;
      return v;
    };
    s.flops = 0;
    s.bytes = 0;
    if (o.kind == CTD_OP_CONV || o.kind == CTD_OP_CONVT) {
      const TensorState& td = e->tensors[o.dst];
      a.s0 = view(o.src0, o.src0_coff, o.src0_c, o.src0_up);
      if (o.src1 >= 0) a.s1 = view(o.src1, o.src1_coff, o.src1_c, o.src1_up);
      a.B = B;
      a.Hin = a.s0.H << (o.src0_up ? 1 : 0);
      a.Win = a.s0.W << (o.src0_up ? 1 : 0);
      if (o.src1 >= 0) {
        const int h1 = a.s1.H << (o.src1_up ? 1 : 0), w1 = a.s1.W << (o.src1_up ? 1 : 0);
        if (h1 != a.Hin || w1 != a.Win) return fail(CTD_ERR_INVALID, "op " + std::to_string(i) + ": source sizes differ");
      }
      a.dst = e->arena + td.offset + (size_t)o.dst_coff * td.esize;
      a.pitchD = td.t.channels;
      a.oH = td.H;
      a.oW = td.W;
      if (o.res >= 0) {
        const TensorState& tr = e->tensors[o.res];
        a.res = e->arena + tr.offset + (size_t)o.res_coff * tr.esize;
        a.pitchR = tr.t.channels;
        if (tr.H != td.H || tr.W != td.W) return fail(CTD_ERR_INVALID, "residual size mismatch");
      }
      a.act = o.act;
      a.N = o.cout;
      a.Npad = s.npad;
      a.w = s.w_dev;
      a.bias = s.b_dev;
      a.bk = s.bk;
      a.zeros = e->zeros;
      a.w_tiled = e->w_tiled;
      a.k_rot = 0;
      a.prio = g_fwd_prio;
      const int cin = o.src0_c + (o.src1 >= 0 ? o.src1_c : 0);
      a.nphase = 1;
      a.osy = a.osx = 1;
      if (o.kind == CTD_OP_CONV) {
        const int Ho = (a.Hin + 2 * o.pad - o.k) / o.stride + 1, Wo = (a.Win + 2 * o.pad - o.k) / o.stride + 1;
        if (Ho != td.H || Wo != td.W)
          return fail(CTD_ERR_INVALID, "op " + std::to_string(i) + ": conv output size does not match dst tensor");
        a.Mh = Ho; a.Mw = Wo;
        a.KH = a.KW = o.k;
        a.stride = o.stride;
        a.dy0 = a.dx0 = -o.pad;
        a.K = o.k * o.k * cin;
        a.M = B * Ho * Wo;
        s.flops = 2.0 * a.M * a.N * a.K;
      } else {
        const int Ho = (a.Hin - 1) * o.stride - 2 * o.pad + o.k, Wo = (a.Win - 1) * o.stride - 2 * o.pad + o.k;
        if (Ho != td.H || Wo != td.W)
          return fail(CTD_ERR_INVALID, "op " + std::to_string(i) + ": convT output size does not match dst tensor");
        if (s.impl == IMPL_IGEMM_T) {
          a.Mh = a.Hin; a.Mw = a.Win;
          a.KH = a.KW = 2;
          a.stride = 1;
          a.K = 4 * cin;
          a.M = B * a.Hin * a.Win;
          a.nphase = 4;
          a.osy = a.osx = 2;
          a.w_phase_stride = (long long)s.npad * a.K;
        } else {
          a.KH = a.KW = o.k;
          a.stride = o.stride;
          a.dy0 = a.dx0 = o.pad;
          a.M = B * Ho * Wo;
          a.K = o.k * o.k * cin;
        }
        // every input pixel meets k*k taps
        s.flops = 2.0 * (double)B * a.Hin * a.Win * o.k * o.k * cin * a.N;
      }
      if (s.split) {
        if (o.kind == CTD_OP_CONV) a.K = s.kpad;
        a.w2 = (const half_t*)s.w_dev + (size_t)a.nphase * s.npad * a.K;
        a.oscale = s.oscale_dev;
        a.x_sp = e->tensors[o.src0].sp;
        a.d_sp = td.sp;
        a.r_sp = o.res >= 0 && e->tensors[o.res].sp;
      }
      if ((s.impl == IMPL_IGEMM || s.impl == IMPL_IGEMM_T) &&
          !(f16 ? igemm_supported(a) : s.split ? conv_split_supported(a) : conv_f32_mfma_supported(a)))
        return fail(CTD_ERR_UNSUPPORTED, "op " + std::to_string(i) + ": shape rejected by the MFMA kernel");
      const double es = f16 ? 2 : 4;
      s.bytes = ((double)B * a.s0.H * a.s0.W * a.s0.c + (o.src1 >= 0 ? (double)B * a.s1.H * a.s1.W * a.s1.c : 0)) * es +
                (double)B * td.H * td.W * a.N * td.esize + (o.res >= 0 ? (double)B * td.H * td.W * a.N * es : 0) +
                (double)a.N * o.k * o.k * cin * es;
    }
    s.args = a;
    s.c3_head = s.skip = s.sppf_head = s.stem2_head = s.stemsp_head = s.c3b_head = s.post_head = s.segp_head = s.segp_ran = false;
    s.c3b_conv3 = -1;
  }
  // ---- fp32s: INPUT (page -> fp32 NHWC, zero 4th channel) + the 6x6/s2 first conv -> one launch that reads the page itself
  for (int i = 0; e->prec == CTD_PREC_F32S && g_split_stem && i + 1 < nO; ++i) {
    OpState &S0 = e->ops[i], &S1 = e->ops[i + 1];
    const ctd_op &o0 = S0.op, &o1 = S1.op;
    if (o0.kind != CTD_OP_INPUT || o1.kind != CTD_OP_CONV || !S1.split || S1.impl != IMPL_IGEMM) continue;
    const int T0 = o0.dst;
    if (o1.src0 != T0 || o1.src0_coff != 0 || o1.src1 >= 0 || o1.res >= 0) continue;
    if (e->tensors[T0].first_def != i || e->tensors[T0].last_use != i + 1) continue;
    if (!stem_split_supported(S1.args)) continue;
    S0.skip = true;
    S1.stemsp_head = true;
    // algorithmic bytes: the page (3 B per pixel as uint8; the float format is 12) in, the 32-channel map out, the weights
    S1.bytes = (double)B * H * W * 3 + (double)B * S1.args.oH * S1.args.oW * S1.args.N * 4 + (double)S1.args.N * 36 * 3 * 4;
  }
  // ---- stem + layer 1: the stem's output has one consumer, a 3x3/s2 conv 32 -> 64 -> one launch, never stored
  for (int i = 0; f16 && (g_fuse & 4) && i + 1 < nO; ++i) {
    OpState &S0 = e->ops[i], &S1 = e->ops[i + 1];
    const ctd_op &o0 = S0.op, &o1 = S1.op;
    if (o0.kind != CTD_OP_STEM || o1.kind != CTD_OP_CONV || S1.impl != IMPL_IGEMM || S1.bk != 32 || !e->w_tiled) continue;
    const int T0 = o0.dst;
    if (o0.cout != 32 || o0.dst_coff != 0 || e->tensors[T0].t.channels != 32) continue;
    if (o1.k != 3 || o1.stride != 2 || o1.pad != 1 || o1.src0 != T0 || o1.src0_coff != 0 || o1.src0_c != 32 ||
        o1.src0_up || o1.src1 >= 0 || o1.res >= 0 || o1.cout != 64 || o1.dst == T0)
      continue;
    if (e->tensors[T0].first_def != i || e->tensors[T0].last_use != i + 1 || e->tensors[o1.dst].esize != 2) continue;
    Stem2Args f{};
    f.prio = g_fwd_prio;
    f.B = B; f.H = H; f.W = W;
    f.wfrag = (const half_t*)S0.w_dev; f.bias0 = S0.b_dev; f.act0 = o0.act;
    f.w1 = (const half_t*)S1.w_dev; f.bias1 = S1.b_dev; f.act1 = o1.act;
    f.dst = (half_t*)S1.args.dst; f.pitchD = S1.args.pitchD;
    if (S1.args.oH != H / 4 || S1.args.oW != W / 4 || !stem_conv2_supported(f)) continue;
    S0.stem2_head = true;
    S0.st2 = f;
    S1.skip = true;
    S0.flops += S1.flops; S0.bytes += S1.bytes;
    S1.flops = 0; S1.bytes = 0;
  }
  // ---- SPPF: three chained stride-1 max pools over the slots of one cat tensor -> one launch
  for (int i = 0; (g_fuse & 2) && i + 2 < nO; ++i) {   // every engine: max is exact in any precision
    const ctd_op &p0 = e->ops[i].op, &p1 = e->ops[i + 1].op, &p2 = e->ops[i + 2].op;
    if (p0.kind != CTD_OP_MAXPOOL || p1.kind != CTD_OP_MAXPOOL || p2.kind != CTD_OP_MAXPOOL) continue;
    const int c = p0.src0_c, P = p0.src0;
    if (p0.dst != P || p1.src0 != P || p1.dst != P || p2.src0 != P || p2.dst != P) continue;
    if (p1.src0_c != c || p2.src0_c != c || p1.k != p0.k || p2.k != p0.k) continue;
    if (p0.dst_coff != p0.src0_coff + c || p1.src0_coff != p0.dst_coff || p1.dst_coff != p1.src0_coff + c ||
        p2.src0_coff != p1.dst_coff || p2.dst_coff != p2.src0_coff + c)
      continue;
    const TensorState& tp = e->tensors[P];
    if (!sppf_pool3_supported(tp.t.channels, c, c, tp.H, tp.W, p0.k, e->arena + tp.offset + (size_t)p0.src0_coff * tp.esize, tp.esize))
      continue;
    e->ops[i].sppf_head = true;
    e->ops[i + 1].skip = e->ops[i + 2].skip = true;
    i += 2;
  }
  // ---- C3 blocks with 32 hidden channels and one bottleneck: ops [cv1+cv2, m.cv1, m.cv2 (+shortcut), cv3] whose
  // intermediates nobody else reads become one launch of kernels_c3.hip (same packed weights, same arithmetic)
  for (int i = 0; f16 && (g_fuse & 1) && i + 3 < nO; ++i) {
    OpState &A = e->ops[i], &Bo = e->ops[i + 1], &C = e->ops[i + 2], &D = e->ops[i + 3];
    const ctd_op &a = A.op, &b = Bo.op, &c = C.op, &d = D.op;
    auto mfma16 = [&](const OpState& s) { return s.op.kind == CTD_OP_CONV && s.impl == IMPL_IGEMM && s.bk == 32 && e->w_tiled; };
    if (!mfma16(A) || !mfma16(Bo) || !mfma16(C) || !mfma16(D)) continue;
    const int Y = a.dst, T = b.dst;
    if (a.k != 1 || a.stride != 1 || a.cout != 64 || a.res >= 0 || a.dst_coff != 0 || e->tensors[Y].t.channels != 64) continue;
    if (b.k != 1 || b.stride != 1 || b.src0 != Y || b.src0_coff != 0 || b.src0_c != 32 || b.src1 >= 0 || b.cout != 32 ||
        b.res >= 0 || b.dst_coff != 0 || e->tensors[T].t.channels != 32 || T == Y)
      continue;
    if (c.k != 3 || c.stride != 1 || c.pad != 1 || c.src0 != T || c.src0_coff != 0 || c.src0_c != 32 || c.src1 >= 0 ||
        c.cout != 32 || c.dst != Y || c.dst_coff != 0 || c.res != Y || c.res_coff != 0)
      continue;
    if (d.k != 1 || d.stride != 1 || d.src0 != Y || d.src0_coff != 0 || d.src0_c != 64 || d.src1 >= 0 || d.cout != 64 ||
        d.res >= 0 || d.dst == Y || d.dst == T)
      continue;
    if (a.act != b.act || a.act != c.act || a.act != d.act) continue;
    if (b.src0_up || c.src0_up || d.src0_up) continue;
    // the intermediates must be private to the block
    if (e->tensors[Y].first_def != i || e->tensors[Y].last_use != i + 3 || e->tensors[T].first_def != i + 1 ||
        e->tensors[T].last_use != i + 2)
      continue;
    if (e->tensors[d.dst].esize != 2 || e->tensors[Y].esize != 2 || e->tensors[T].esize != 2) continue;
    C3Args f{};
    f.prio = g_fwd_prio;
    f.s0 = A.args.s0;
    f.s1 = A.args.s1;
    f.B = B; f.H = A.args.Hin; f.W = A.args.Win;
    f.w12 = (const half_t*)A.w_dev; f.wm1 = (const half_t*)Bo.w_dev; f.wm2 = (const half_t*)C.w_dev; f.wc3 = (const half_t*)D.w_dev;
    f.b12 = A.b_dev; f.bm1 = Bo.b_dev; f.bm2 = C.b_dev; f.bc3 = D.b_dev;
    f.dst = D.args.dst; f.pitchD = D.args.pitchD;
    f.act = a.act;
    f.zeros = e->zeros;
    if (D.args.oH != f.H || D.args.oW != f.W || !c3_fused_supported(f)) continue;
    A.c3_head = true;
    A.c3 = f;
    Bo.skip = C.skip = D.skip = true;
    // the block's work is booked on its first op (the algorithmic bytes of the four layers stay the yardstick)
    A.flops += Bo.flops + C.flops + D.flops;
    A.bytes += Bo.bytes + C.bytes + D.bytes;
    Bo.flops = C.flops = D.flops = 0;
    Bo.bytes = C.bytes = D.bytes = 0;
    i += 3;
  }
  // ---- ConvTranspose + its 1x1 consumer (found above)
  for (int i : posts) {
    if (!(g_fuse & 16)) break;
    OpState &T = e->ops[i], &P = e->ops[i + 1];
    ConvArgs a = T.args;
    a.post_w = P.w_dev;
    a.post_bias = P.b_dev;
    a.post_dst = P.args.dst;
    a.post_pitch = P.args.pitchD;
    a.post_n = P.op.cout;
    a.post_act = P.op.act;
    a.post_x = SrcView{};
    if (P.op.src1 >= 0) a.post_x = P.args.s0;
    if (P.args.oH != T.args.oH || P.args.oW != T.args.oW || !conv_halo3_post_supported(a)) continue;
    T.post_head = true;
    T.post_args = a;
    P.skip = true;
    T.flops += P.flops; T.bytes += P.bytes;
    P.flops = P.bytes = 0;
  }
  // ---- the UNet's last pair: ConvTranspose 128 -> 64 (+BN+ReLU) whose only reader is the 64 -> 1 ConvTranspose + sigmoid
  // (SEG_FINAL): the first one's launch also forms the second one's 16 tap products per pixel and stores THOSE (64 B per
  // pixel, in the 64-channel tensor's own arena slot, which is 128 B per pixel) -- kernels_halo3.hip SEGP
  for (int i = 0; f16 && (g_fuse & 32) && i + 1 < nO; ++i) {
    OpState &T = e->ops[i], &F = e->ops[i + 1];
    const ctd_op &t = T.op, &fo = F.op;
    if (t.kind != CTD_OP_CONVT || T.impl != IMPL_IGEMM_T || t.cout != 64 || t.dst_coff != 0 || T.bk != 32 || !e->w_tiled) continue;
    if (fo.kind != CTD_OP_SEG_FINAL || fo.src0 != t.dst || fo.src0_coff != 0 || fo.src0_c != 64 || !g_seg_final_mfma) continue;
    const TensorState& tu = e->tensors[t.dst];
    if (tu.t.channels != 64 || tu.esize != 2 || tu.first_def != i || tu.last_use != i + 1) continue;
    ConvArgs a = T.args;
    a.post_w = F.w_dev;
    a.post_dst = T.args.dst;             // P takes the place of the map it replaces
    a.post_n = -16;
    if (!conv_halo3_segp_supported(a)) continue;
    T.segp_head = true;
    T.post_args = a;
  }
  // ---- the wider C3 blocks found above: one launch per bottleneck, the last one with cv3 (kernels_c3b.hip)
  for (const C3Chain& ch : chains) {
    if (!(g_fuse & 8)) break;
    OpState &A = e->ops[ch.a], &D = e->ops[ch.d];
    const ctd_op& d = D.op;
    std::vector<C3bArgs> fs(ch.n);
    bool ok = true;
    for (int q = 0; q < ch.n && ok; ++q) {
      OpState &Bo = e->ops[ch.a + 1 + 2 * q], &C = e->ops[ch.a + 2 + 2 * q];
      const bool last = q + 1 == ch.n;
      C3bArgs f{};
      f.prio = g_fwd_prio;
      f.B = B; f.H = A.args.oH; f.W = A.args.oW;
      f.ch = ch.c;
      f.y1 = q == 0 ? Bo.args.s0 : SrcView{e->ops[ch.a + 1 + 2 * (q - 1)].args.dst, ch.c, ch.c, 0, f.H, f.W};
      f.wm1 = (const half_t*)Bo.w_dev; f.bm1 = Bo.b_dev;
      f.wm2 = (const half_t*)C.w_dev; f.bm2 = C.b_dev;
      f.add = C.op.res >= 0;
      f.act = Bo.op.act;
      f.zeros = e->zeros;
      if (last) {
        f.cv3 = 1;
        f.y2 = D.args.s0;
        f.y2.ptr = (const char*)D.args.s0.ptr + (size_t)ch.c * 2;
        f.y2.c = ch.c;
        f.wc3 = (const half_t*)D.w_dev; f.bc3 = D.b_dev;
        f.dst = D.args.dst; f.pitchD = D.args.pitchD;
        ok = D.args.oH == f.H && D.args.oW == f.W && D.npad == 2 * ch.c;
      } else {
        f.cv3 = 0;
        f.dst = Bo.args.dst; f.pitchD = Bo.args.pitchD;
      }
      ok = ok && Bo.npad == ch.c && C.npad == ch.c && Bo.args.oH == f.H && Bo.args.oW == f.W && c3b_supported(f);
      fs[q] = f;
    }
    if (!ok) continue;
    for (int q = 0; q < ch.n; ++q) {
      OpState &Bo = e->ops[ch.a + 1 + 2 * q], &C = e->ops[ch.a + 2 + 2 * q];
      Bo.c3b_head = true;
      Bo.c3b = fs[q];
      Bo.c3b_conv3 = ch.a + 2 + 2 * q;
      C.skip = true;
      // the launch's work is booked on its first op (the algorithmic bytes of the layers stay the yardstick)
      Bo.flops += C.flops; Bo.bytes += C.bytes;
      C.flops = C.bytes = 0;
      if (q + 1 == ch.n) {
        D.skip = true;
        Bo.flops += D.flops; Bo.bytes += D.bytes;
        D.flops = D.bytes = 0;
      }
    }
    (void)d;
  }
  e->p_fuse = g_fuse + 64 * g_fuse_epoch;
  e->pB = B; e->pH = H; e->pW = W;
  return CTD_OK;
}

struct Outs {
  const void* input; int in_fmt;
  float* blks; float* mask; float* lines; uint8_t* mask_u8; uint8_t* bitmap;
  int blk_rows;
};

int launch_op(ctd_engine* e, int i, const Outs& x, hipStream_t st) {
  OpState& s = e->ops[i];
  const ctd_op& o = s.op;
  const bool f16 = e->prec == CTD_PREC_F16;
  const int B = e->pB, H = e->pH, W = e->pW;
  auto tptr = [&](int id, int coff) {
    const TensorState& t = e->tensors[id];
    return (void*)(e->arena + t.offset + (size_t)coff * t.esize);
  };
  switch (o.kind) {
    case CTD_OP_INPUT: {
      if (s.skip) break;
      void* d = tptr(o.dst, 0);
      const int pitch = e->tensors[o.dst].t.channels;
      if (x.in_fmt == CTD_IN_NCHW_F32) launch_input_nchw((const float*)x.input, d, pitch, B, H, W, f16, st);
      else launch_input_u8((const uint8_t*)x.input, d, pitch, B, H, W, f16, st);
      break;
    }
    case CTD_OP_STEM: {
      if (s.stem2_head) {
        Stem2Args f = s.st2;
        f.in = x.input;
        f.in_fmt = x.in_fmt;
        launch_stem_conv2(f, st);
        break;
      }
      const TensorState& td = e->tensors[o.dst];
      launch_stem(x.input, x.in_fmt, (half_t*)tptr(o.dst, o.dst_coff), td.t.channels, B, H, W, o.cout,
                  (const half_t*)s.w_dev, s.b_dev, o.act, st);
      break;
    }
    case CTD_OP_CONV:
      if (s.skip) break;
      if (s.c3_head) { launch_c3_fused(s.c3, st); break; }
      if (s.c3b_head) {   // the 3x3's K walk is the one its own dispatch would take NOW (a tuning key may have moved it)
        C3bArgs f = s.c3b;
        f.tap_major = conv_halo_supported(e->ops[s.c3b_conv3].args, false) ? 0 : 1;
        launch_c3b(f, st);
        break;
      }
      if (s.stemsp_head) { launch_stem_split(s.args, x.input, x.in_fmt, st); break; }
      if (s.impl == IMPL_IGEMM && s.split) launch_conv_split(s.args, st);
      else if (s.impl == IMPL_IGEMM && !f16) launch_conv_f32_mfma(s.args, st);
      else if (s.impl == IMPL_IGEMM) launch_conv_igemm(s.args, e->tensors[o.dst].esize == 4, st);
      else launch_conv_direct(s.args, f16, st);
      break;
    case CTD_OP_CONVT:
      if (s.segp_head) {   // decided per launch (a tuning key may have taken the big-tile kernel away): SEG_FINAL is told
        const bool fused = conv_halo3_segp_supported(s.post_args) && (x.mask || x.mask_u8);
        e->ops[i + 1].segp_ran = fused;
        launch_conv_igemm(fused ? s.post_args : s.args, false, st);
        break;
      }
      if (s.post_head) {
        if (conv_halo3_post_supported(s.post_args)) { launch_conv_igemm(s.post_args, false, st); break; }
        // a tuning key has taken the big-tile kernel away since the plan was made: the two layers, one launch each
        launch_conv_igemm(s.args, false, st);
        launch_conv_igemm(e->ops[i + 1].args, false, st);
        break;
      }
      if (s.impl == IMPL_IGEMM_T && s.split) launch_conv_split(s.args, st);
      else if (s.impl == IMPL_IGEMM_T && !f16) launch_conv_f32_mfma(s.args, st);
      else if (s.impl == IMPL_IGEMM_T) launch_conv_igemm(s.args, false, st);
      else launch_convt_direct(s.args, f16, st);
      break;
    case CTD_OP_MAXPOOL: {
      if (s.skip) break;
      if (s.sppf_head) {
        const TensorState& tp = e->tensors[o.src0];
        launch_sppf_pool3(tptr(o.src0, o.src0_coff), tp.t.channels, o.src0_c, o.src0_c, B, tp.H, tp.W, o.k, st, tp.esize);
        break;
      }
      const TensorState& ts = e->tensors[o.src0];
      const TensorState& td = e->tensors[o.dst];
      launch_maxpool(tptr(o.src0, o.src0_coff), ts.t.channels, tptr(o.dst, o.dst_coff), td.t.channels, o.src0_c, B,
                     ts.H, ts.W, o.k, f16, st);
      break;
    }
    case CTD_OP_AVGPOOL2: {
      const TensorState& ts = e->tensors[o.src0];
      const TensorState& td = e->tensors[o.dst];
      launch_avgpool2(tptr(o.src0, o.src0_coff), ts.t.channels, tptr(o.dst, o.dst_coff), td.t.channels, o.src0_c, B,
                      td.H, td.W, f16, st);
      break;
    }
    case CTD_OP_DETECT: {
      const TensorState& ts = e->tensors[o.src0];
      if (!x.blks) break;
      // aux[1] is the row offset for a 64x64 unit input scaled by (H/64)*(W/64)
      const int unit = (H / 64) * (W / 64);
      launch_detect_decode(tptr(o.src0, o.src0_coff), ts.t.channels, ts.esize == 2, x.blks, x.blk_rows,
                           o.aux[1] * unit, B, ts.H, ts.W, o.aux[2], o.aux[3], (float)o.aux[0], s.aux_dev, st);
      break;
    }
    case CTD_OP_EXPORT: {
      const TensorState& ts = e->tensors[o.src0];
      float* out = o.aux[0] == CTD_OUT_MASK ? x.mask : x.lines;
      const int nplanes = o.aux[0] == CTD_OUT_MASK ? 1 : (o.aux[2] > 0 ? o.aux[2] : 2);   // aux[2]: planes of lines_map
      uint8_t* u8 = nullptr;
      int mode = 0;
      if (o.aux[0] == CTD_OUT_MASK) { u8 = x.mask_u8; mode = 1; }
      else if (o.aux[1] == 0) { u8 = x.bitmap; mode = 2; }
      if (!out && !u8) break;                      // the f32 plane may be skipped while its u8 side output is wanted
      launch_export_plane(tptr(o.src0, o.src0_coff), ts.t.channels, ts.esize == 2, out, nplanes, o.aux[1], u8, mode,
                          o.faux[0], B, ts.H, ts.W, st);
      break;
    }
    case CTD_OP_SEG_FINAL: {
      const TensorState& ts = e->tensors[o.src0];
      if (!x.mask && !x.mask_u8) break;
      if (s.segp_ran)      // the producer stored the tap products: col2im + sigmoid + u8 from them
        launch_seg_final_gather((const float*)tptr(o.src0, o.src0_coff), B, ts.H, ts.W, 0.f, x.mask, x.mask_u8, st);
      else if (ts.esize == 4)
        launch_seg_final_f32((const float*)tptr(o.src0, o.src0_coff), ts.t.channels, o.src0_c, B, ts.H, ts.W,
                             (const float*)s.w_dev, x.mask, x.mask_u8, st);
      else
        launch_seg_final((const half_t*)tptr(o.src0, o.src0_coff), ts.t.channels, o.src0_c, B, ts.H, ts.W,
                         (const float*)s.w_dev, 0.f, x.mask, x.mask_u8, st);
      break;
    }
    case CTD_OP_DB_UP: {
      const TensorState& ts = e->tensors[o.src0];
      if (!x.lines) break;
      launch_db_up(tptr(o.src0, o.src0_coff), ts.esize == 4, ts.t.channels, o.aux[1], o.aux[2] > 0 ? o.aux[2] : 2, B, ts.H, ts.W,
                   (const float*)s.w_dev, x.lines, x.bitmap, o.faux[0], st);
      break;
    }
  }
  return CTD_OK;
}

int prepare(ctd_engine* e, int B, int H, int W, hipStream_t st = nullptr) {
  HIP_TRY(hipSetDevice(e->device));
  if (B != e->pB || H != e->pH || W != e->pW || e->p_fuse != g_fuse + 64 * g_fuse_epoch) return plan(e, B, H, W, st);
  return CTD_OK;
}

}  // namespace

int ctd_fail_msg(int code, const std::string& msg) { return fail(code, msg); }
extern int g_tail_priority;   // tail.hip
extern int g_tail_cus, g_tail_cu_first;
extern long long g_tail_dma_min;
#ifdef CTD_MEASURE_KNOBS
extern int g_tail_skip_pages;
extern int g_tail_ablate;
#endif
extern int g_tail_chain;
extern int g_tail_fused_rounds;
extern long long g_tail_fused_max_pix;
extern int g_tail_lds, g_tail_lds_rcap, g_tw_lds_runs_x10, g_tw_lds_threads;
extern long long g_tail_lds_cls0, g_tail_lds_cls1;
extern long long g_tail_lds_max_bytes;

extern "C" {

const char* ctd_last_error(void) { return g_err.c_str(); }
int32_t ctd_abi_version(void) { return CTD_ABI_VERSION; }

int ctd_device_info(int32_t device, char* name, int32_t* cu_count, int64_t* hbm_bytes) {
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  if (name) { std::strncpy(name, p.name, 255); name[255] = 0; }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  int arch = 0;
  const char* g = std::strstr(p.gcnArchName, "gfx");
  if (g) arch = std::atoi(g + 3);
  return arch;
}

int ctd_engine_create(ctd_engine** out, const ctd_tensor* tensors, int32_t n_tensors, const ctd_op* ops,
                      int32_t n_ops, const float* params, int64_t n_params, int32_t precision, int32_t device) {
  if (!out || !tensors || !ops || !params || n_tensors < 1 || n_ops < 1)
    return fail(CTD_ERR_INVALID, "null/empty program");
  if (precision != CTD_PREC_F32 && precision != CTD_PREC_F16 && precision != CTD_PREC_F32S)
    return fail(CTD_ERR_INVALID, "bad precision");
  HIP_TRY(hipSetDevice(device));
  ctd_engine* e = new ctd_engine();
  e->device = device;
  e->prec = precision;
  e->no_reuse = g_no_reuse != 0;
  e->f32_mfma = g_f32_mfma != 0;

  e->tensors.resize(n_tensors);
  for (int i = 0; i < n_tensors; ++i) e->tensors[i].t = tensors[i];
  e->ops.resize(n_ops);
  for (int i = 0; i < n_ops; ++i) e->ops[i].op = ops[i];
  {
    std::vector<uint8_t> z(CTD_ZEROS_BYTES, 0);
    if (int zrc = upload(e, z, &e->zeros)) { ctd_engine_destroy(e); return zrc; }
  }
  int rc = validate(e);
  for (int i = 0; rc == CTD_OK && i < n_ops; ++i) rc = pack_op(e, e->ops[i], params, n_params);
  if (rc == CTD_OK) {
    int rows = 0, no = 0;
    for (auto& s : e->ops)
      if (s.op.kind == CTD_OP_DETECT) {
        const int st = s.op.aux[0];
        if (st < 1 || 64 % st) { rc = fail(CTD_ERR_INVALID, "detect stride must divide 64"); break; }
        rows += s.op.aux[2] * (64 / st) * (64 / st);
        no = s.op.aux[3];
      }
    e->det_rows_per_unit = rows;
    e->det_no = no;
  }
  if (rc != CTD_OK) {
    ctd_engine_destroy(e);
    return rc;
  }
  *out = e;
  return CTD_OK;
}

void ctd_engine_destroy(ctd_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  for (void* p : e->owned) (void)hipFree(p);
  if (e->arena) (void)hipFree(e->arena);
  delete e;
}

int ctd_engine_blks_shape(const ctd_engine* e, int32_t H, int32_t W, int32_t* rows, int32_t* no) {
  if (!e || H % 64 || W % 64) return fail(CTD_ERR_INVALID, "H, W must be multiples of 64");
  if (rows) *rows = e->det_rows_per_unit * (H / 64) * (W / 64);
  if (no) *no = e->det_no;
  return CTD_OK;
}

int ctd_engine_forward(ctd_engine* e, const void* input_dev, int32_t input_fmt, int32_t B, int32_t H, int32_t W,
                       float* blks_dev, float* mask_dev, float* lines_dev, uint8_t* mask_u8_dev, uint8_t* bitmap_dev,
                       void* stream) {
  if (!e || !input_dev) return fail(CTD_ERR_INVALID, "null engine/input");
  if (input_fmt != CTD_IN_NCHW_F32 && input_fmt != CTD_IN_NHWC_U8) return fail(CTD_ERR_INVALID, "bad input format");
  if (int rc = prepare(e, B, H, W, (hipStream_t)stream)) return rc;
  Outs x{input_dev, input_fmt, blks_dev, mask_dev, lines_dev, mask_u8_dev, bitmap_dev,
         e->det_rows_per_unit * (H / 64) * (W / 64)};
  for (int i = 0; i < (int)e->ops.size(); ++i)
    if (int rc = launch_op(e, i, x, (hipStream_t)stream)) return rc;
  HIP_TRY(hipGetLastError());
  return CTD_OK;
}

int32_t ctd_engine_n_ops(const ctd_engine* e) { return e ? (int32_t)e->ops.size() : 0; }

int ctd_engine_op_work(const ctd_engine* e, double* flops, double* bytes, int32_t* kernel_class) {
  if (!e) return fail(CTD_ERR_INVALID, "null engine");
  for (size_t i = 0; i < e->ops.size(); ++i) {
    if (flops) flops[i] = e->ops[i].flops;
    if (bytes) bytes[i] = e->ops[i].bytes;
    if (kernel_class) kernel_class[i] = e->ops[i].impl;
  }
  return CTD_OK;
}

int ctd_engine_op_kernel(const ctd_engine* e, int32_t i, char* name, int32_t cap) {
  if (!e || !name || cap < 2) return fail(CTD_ERR_INVALID, "null argument");
  if (i < 0 || i >= (int)e->ops.size()) return fail(CTD_ERR_INVALID, "op index out of range");
  if (!e->pB) return fail(CTD_ERR_INVALID, "no plan yet: run a forward of the shape first");
  const OpState& s = e->ops[i];
  const ctd_op& o = s.op;
  const bool f16 = e->prec == CTD_PREC_F16;
  const char* k = "?";
  auto conv_kernel = [&]() -> const char* {   // mirrors launch_op / launch_conv_igemm / launch_conv_split
    if (s.skip) return "(fused)";
    if (s.c3_head) return "c3_fused_kernel";
    if (s.c3b_head) return "c3b_kernel";
    if (s.post_head && conv_halo3_post_supported(s.post_args)) return "conv_halo3_kernel+1x1";
    if (s.segp_head && conv_halo3_segp_supported(s.post_args)) return "conv_halo3_kernel+taps";
    if (s.stemsp_head) return "stem_split_kernel";
    const bool mfma = s.impl == IMPL_IGEMM || s.impl == IMPL_IGEMM_T;
    if (mfma && s.split) return conv_split_halo_supported(s.args) ? "conv_split_halo_kernel" : "conv_split_kernel";
    if (mfma && !f16) return "conv_f32_mfma_kernel";
    if (mfma) {
      ConvArgs a = s.args;
      if (!a.bk) a.bk = igemm_pick_bk(a.s0.c, a.s1.c, a.K, a.N, 0);
      const bool d32 = o.kind == CTD_OP_CONV && e->tensors[o.dst].esize == 4;
      if (conv_halo3_supported(a, d32)) return "conv_halo3_kernel";
      if (conv_halo2_supported(a, d32)) return "conv_halo2_kernel";
      if (conv_halo_supported(a, d32)) return "conv_halo_kernel";
      return "conv_igemm_kernel";
    }
    return o.kind == CTD_OP_CONV ? "conv_direct_kernel" : "convt_direct_kernel";
  };
  switch (o.kind) {
    case CTD_OP_INPUT: k = s.skip ? "(fused)" : "input_kernel"; break;
    case CTD_OP_STEM: k = s.stem2_head ? "stem_conv2_kernel" : "stem_mfma_kernel"; break;
    case CTD_OP_CONV:
    case CTD_OP_CONVT: k = conv_kernel(); break;
    case CTD_OP_MAXPOOL: k = s.skip ? "(fused)" : s.sppf_head ? "sppf_pool3_kernel" : "maxpool_kernel"; break;
    case CTD_OP_AVGPOOL2: k = "avgpool2_kernel"; break;
    case CTD_OP_DETECT: k = "detect_decode_kernel"; break;
    case CTD_OP_EXPORT: k = "export_plane_kernel"; break;
    case CTD_OP_SEG_FINAL: k = (i > 0 && e->ops[i - 1].segp_head && conv_halo3_segp_supported(e->ops[i - 1].post_args)) ? "seg_final_gather_kernel" : e->tensors[o.src0].esize == 4 ? "seg_final_f32_kernel" : (g_seg_final_mfma ? "seg_final_mfma_kernel" : "seg_final_kernel"); break;
    case CTD_OP_DB_UP: k = (e->tensors[o.src0].esize == 2 && g_db_up_mfma) ? "db_up_mfma_kernel" : "db_up_kernel"; break;
  }
  std::snprintf(name, (size_t)cap, "%s", k);
  return CTD_OK;
}

int ctd_engine_profile(ctd_engine* e, const void* input_dev, int32_t input_fmt, int32_t B, int32_t H, int32_t W,
                       float* blks_dev, float* mask_dev, float* lines_dev, uint8_t* mask_u8_dev, uint8_t* bitmap_dev,
                       void* stream, float* op_ms) {
  if (!e || !input_dev || !op_ms) return fail(CTD_ERR_INVALID, "null argument");
  if (int rc = prepare(e, B, H, W)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)e->ops.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& v : ev) HIP_TRY(hipEventCreate(&v));
  Outs x{input_dev, input_fmt, blks_dev, mask_dev, lines_dev, mask_u8_dev, bitmap_dev,
         e->det_rows_per_unit * (H / 64) * (W / 64)};
  HIP_TRY(hipEventRecord(ev[0], st));
  for (int i = 0; i < n; ++i) {
    if (int rc = launch_op(e, i, x, st)) return rc;
    HIP_TRY(hipEventRecord(ev[i + 1], st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) HIP_TRY(hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]));
  for (auto& v : ev) (void)hipEventDestroy(v);
  return CTD_OK;
}

int ctd_engine_read_tensor(ctd_engine* e, int32_t tensor_id, float* host_out, int64_t n_floats) {
  if (!e || !tensor_ok(e, tensor_id) || !host_out) return fail(CTD_ERR_INVALID, "bad tensor id");
  if (!e->arena) return fail(CTD_ERR_INVALID, "no forward has run");
  const TensorState& t = e->tensors[tensor_id];
  const int64_t n = (int64_t)e->pB * t.H * t.W * t.t.channels;
  if (n_floats != n) return fail(CTD_ERR_INVALID, "size mismatch: tensor has " + std::to_string(n) + " elements");
  HIP_TRY(hipDeviceSynchronize());
  if (t.esize == 4) {
    HIP_TRY(hipMemcpy(host_out, e->arena + t.offset, n * 4, hipMemcpyDeviceToHost));
    if (t.sp) {   // split-plane: 32 hi halves + 32 lo halves per 32-channel group -> hi + lo
      for (int64_t g = 0; g < n / 32; ++g) {
        half_t h[64];
        std::memcpy(h, host_out + 32 * g, 128);
        for (int c = 0; c < 32; ++c) host_out[32 * g + c] = (float)h[c] + (float)h[32 + c];
      }
    }
  } else {
    std::vector<half_t> tmp(n);
    HIP_TRY(hipMemcpy(tmp.data(), e->arena + t.offset, n * 2, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) host_out[i] = (float)tmp[i];
  }
  return CTD_OK;
}

int64_t ctd_engine_workspace_bytes(const ctd_engine* e) { return e ? (int64_t)e->arena_bytes : 0; }
int32_t ctd_engine_arena_generation(const ctd_engine* e) { return e ? e->arena_gen : -1; }

int ctd_tuning_set(const char* key, int64_t value) {
  if (key && std::string(key) == "fuse") { g_fuse = (int)value; return CTD_OK; }
  if (key && std::string(key) == "tail_max_blocks") { g_tail_max_blocks = (int)std::max<int64_t>(1, value); return CTD_OK; }
  if (key && std::string(key) == "tail_chain") { g_tail_chain = (int)value; return CTD_OK; }
  if (key && std::string(key) == "tail_fused_rounds") { g_tail_fused_rounds = (int)value; return CTD_OK; }
  if (key && std::string(key) == "tail_fused_max_pix") { g_tail_fused_max_pix = value; return CTD_OK; }
  if (key && std::string(key) == "tail_lds") { g_tail_lds = (int)value; return CTD_OK; }
  if (key && std::string(key) == "tail_lds_rcap") { g_tail_lds_rcap = (int)value; return CTD_OK; }
  if (key && std::string(key) == "tail_lds_max_bytes") { g_tail_lds_max_bytes = value; return CTD_OK; }
  if (key && std::string(key) == "tail_lds_runs_x10") { g_tw_lds_runs_x10 = (int)std::max<int64_t>(1, value); return CTD_OK; }
  if (key && std::string(key) == "tail_lds_threads") { g_tw_lds_threads = (int)value; return CTD_OK; }
  if (key && std::string(key) == "tail_lds_cls0") { g_tail_lds_cls0 = value; return CTD_OK; }
  if (key && std::string(key) == "tail_lds_cls1") { g_tail_lds_cls1 = value; return CTD_OK; }
  if (key && std::string(key) == "tail_dma_min") { g_tail_dma_min = value; return CTD_OK; }
#ifdef CTD_MEASURE_KNOBS   // not in the shipped library: these return incomplete results
  if (key && std::string(key) == "tail_skip_page_download") { g_tail_skip_pages = value != 0; return CTD_OK; }
  if (key && std::string(key) == "tail_ablate") { g_tail_ablate = (int)value; return CTD_OK; }
#endif
  if (key && std::string(key) == "tail_priority") { g_tail_priority = (int)value; return CTD_OK; }
  if (key && std::string(key) == "tail_cus") { g_tail_cus = (int)value; return CTD_OK; }
  if (key && std::string(key) == "tail_cu_first") { g_tail_cu_first = (int)value; return CTD_OK; }
  if (key && std::string(key) == "no_reuse") { g_no_reuse = (int)value; return CTD_OK; }
  if (key && std::string(key) == "f32_mfma") { g_f32_mfma = (int)value; return CTD_OK; }
  if (key && std::string(key) == "split_halo") { g_split_halo = (int)value; return CTD_OK; }
#ifdef CTD_AB_VARIANTS
  if (key && std::string(key) == "split_halo_small") { g_split_halo_small = (int)value; return CTD_OK; }
#endif
  if (key && std::string(key) == "split_halo_min_patches") { g_split_halo_min_patches = value; return CTD_OK; }
  if (key && std::string(key) == "fwd_prio") { g_fwd_prio = (int)value; ++g_fuse_epoch; return CTD_OK; }
  if (key && std::string(key) == "split_stem") { g_split_stem = (int)value; ++g_fuse_epoch; return CTD_OK; }
  if (key && std::string(key) == "split_planes") { g_split_planes = (int)value; ++g_fuse_epoch; return CTD_OK; }
#ifdef CTD_AB_VARIANTS
  if (key && std::string(key) == "split_wdma") { g_split_wdma = (int)value; return CTD_OK; }
  if (key && std::string(key) == "split_bm256") { g_split_bm256 = (int)value; return CTD_OK; }
#endif
  if (key && std::string(key) == "db_up_mfma") { g_db_up_mfma = (int)value; return CTD_OK; }
  if (key && std::string(key) == "seg_final_mfma") { g_seg_final_mfma = (int)value; return CTD_OK; }
  if (key && std::string(key) == "c3_min_patches") { g_c3_min_patches = value; g_fuse_epoch++; return CTD_OK; }
  if (key && std::string(key) == "c3b_min_patches") { g_c3b_min_patches = value; g_fuse_epoch++; return CTD_OK; }
  if (key && std::string(key) == "c3b_max_ch") { g_c3b_max_ch = (int)value; g_fuse_epoch++; return CTD_OK; }
  if (key && std::string(key) == "c3b_cfg64") { g_c3b_cfg64 = (int)value; g_fuse_epoch++; return CTD_OK; }
  if (key && std::string(key) == "c3b_cfg128") { g_c3b_cfg128 = (int)value; g_fuse_epoch++; return CTD_OK; }
  if (conv_tuning_set(key, (long long)value) != 0) return fail(CTD_ERR_INVALID, "unknown tuning key");
  return CTD_OK;
}

// ---- post-processing entry points (kernels_post.hip) ------------------------------
size_t ctd_nms_workspace_bytes(int32_t B, int32_t rows) { return nms_workspace_bytes(B, rows); }

int ctd_nms(const float* blks_dev, int32_t B, int32_t rows, int32_t no, float conf_thres, float iou_thres,
            int32_t max_det, int32_t max_nms, float max_wh, float* dets_dev, int32_t* counts_dev, void* ws_dev,
            size_t ws_bytes, void* stream) {
  if (!blks_dev || !dets_dev || !counts_dev || !ws_dev) return fail(CTD_ERR_INVALID, "null pointer");
  if (B < 1 || rows < 1 || no < 6 || max_det < 1) return fail(CTD_ERR_INVALID, "bad sizes");
  // reference utils/yolov5_utils.py:139-140 asserts
  if (!(conf_thres >= 0.f && conf_thres <= 1.f)) return fail(CTD_ERR_INVALID, "Invalid Confidence threshold");
  if (!(iou_thres >= 0.f && iou_thres <= 1.f)) return fail(CTD_ERR_INVALID, "Invalid IoU");
  if (ws_bytes < nms_workspace_bytes(B, rows)) return fail(CTD_ERR_INVALID, "workspace too small");
  launch_nms(blks_dev, B, rows, no, conf_thres, iou_thres, max_det, max_nms, max_wh, dets_dev, counts_dev, ws_dev,
             (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return CTD_OK;
}

int ctd_db_step(const float* lines_dev, int32_t B, int32_t H, int32_t W, float k, float* out_dev, uint8_t* bitmap_dev,
                float thresh, void* stream) {
  if (!lines_dev || !out_dev || B < 1 || H < 1 || W < 1) return fail(CTD_ERR_INVALID, "ctd_db_step: bad arguments");
  launch_db_step(lines_dev, k, out_dev, bitmap_dev, thresh, B, H, W, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return CTD_OK;
}

size_t ctd_ccl_workspace_bytes(int32_t B, int32_t H, int32_t W) { return ccl_workspace_bytes(B, H, W); }

int ctd_ccl(const uint8_t* img_dev, int32_t B, int32_t H, int32_t W, int32_t thresh, int32_t connectivity,
            int32_t* labels_dev, int32_t* n_dev, int32_t* stats_dev, int32_t max_labels, void* ws_dev, size_t ws_bytes,
            void* stream) {
  if (!img_dev || !labels_dev || !n_dev || !ws_dev) return fail(CTD_ERR_INVALID, "null pointer");
  if (connectivity != 4 && connectivity != 8) return fail(CTD_ERR_INVALID, "connectivity must be 4 or 8");
  if (B < 1 || H < 1 || W < 1 || (long long)H * W >= (1LL << 30)) return fail(CTD_ERR_INVALID, "bad sizes");
  if (ws_bytes < ccl_workspace_bytes(B, H, W)) return fail(CTD_ERR_INVALID, "workspace too small");
  launch_ccl(img_dev, B, H, W, thresh, connectivity, labels_dev, n_dev, stats_dev, max_labels, ws_dev,
             (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return CTD_OK;
}

int ctd_ccl_dual(const uint8_t* img_dev, int32_t B, int32_t H, int32_t W, int32_t thresh, int32_t* labels_dev,
                 int32_t* n_f_dev, int32_t* n_b_dev, int32_t* stats_f_dev, int32_t* stats_b_dev, int32_t* first_f_dev,
                 int32_t* first_b_dev, int32_t max_labels, void* ws_dev, size_t ws_bytes, void* stream) {
  if (!img_dev || !labels_dev || !n_f_dev || !n_b_dev || !stats_f_dev || !stats_b_dev || !first_f_dev || !first_b_dev || !ws_dev)
    return fail(CTD_ERR_INVALID, "null pointer");
  if (B < 1 || H < 1 || W < 1 || (long long)H * W >= (1LL << 30) || max_labels < 1) return fail(CTD_ERR_INVALID, "bad sizes");
  if (ws_bytes < ccl_workspace_bytes(B, H, W)) return fail(CTD_ERR_INVALID, "workspace too small");
  launch_ccl_dual(img_dev, B, H, W, thresh, labels_dev, n_f_dev, n_b_dev, stats_f_dev, stats_b_dev, first_f_dev, first_b_dev,
                  max_labels, ws_dev, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return CTD_OK;
}

int ctd_resize_linear_u8(const uint8_t* src_dev, int32_t sH, int32_t sW, int32_t C, uint8_t* dst_dev, int32_t dH,
                         int32_t dW, int32_t canvasH, int32_t canvasW, void* stream) {
  if (!src_dev || !dst_dev) return fail(CTD_ERR_INVALID, "null pointer");
  if (C != 1 && C != 3) return fail(CTD_ERR_UNSUPPORTED, "resize: 1 or 3 channels");
  if (sH < 1 || sW < 1 || dH < 1 || dW < 1 || canvasH < dH || canvasW < dW) return fail(CTD_ERR_INVALID, "bad sizes");
  launch_resize_linear_u8(src_dev, sH, sW, C, dst_dev, dH, dW, canvasH, canvasW, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return CTD_OK;
}

}  // extern "C"
