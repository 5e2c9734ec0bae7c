// The detector tail: everything `TextDetector.__call__` does after `self.net(img_in)` (reference
// inference.py:148-178) for a whole batch of pages, driven natively:
//
//   postprocess_yolo   (inference.py:101-114)   launch_nms                      + host unpack
//   postprocess_mask   (:85-99)                 fused in the network epilogue (mask_u8)
//   SegDetectorRepresenter (utils/db_utils.py)  launch_ccl_dual + launch_dbc      + host hull / calipers / unclip
//   crop + resize of the mask (:164-165)        copy2d / resize_linear_u8
//   group_output       (utils/textblock.py)     host (csrc/host_group.cpp)
//   refine_mask        (utils/textmask.py)      tw_* kernels + launch_ccl x2     + host colour / threshold picks
//   refine_undetected_mask (:135-156)           mask_clear_where + launch_ccl    + host block test + a 2nd refine pass
//
// One `ctd_tail` object owns a HIP stream, grow-only device and pinned-host buffers and the results of
// its last run.  A run enqueues every kernel of a stage for ALL pages of the batch, then waits once
// for the few KB the host decisions of the next stage need (5 stream synchronisations per batch, none
// per page, block or window).  Objects are independent: several of them can run on different host threads
// (the C ABI is called without the Python interpreter lock), so the tail of batch k overlaps the
// network forward of batch k+1 (comic-text-detector_amd/detector.py `detect_stream`).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "host_refine.h"
#include "kernels.h"
#include "tail.h"

int ctd_fail_msg(int code, const std::string& msg);   // engine.hip: sets the thread-local error text
// Stream priority of tails created from now on ("tail_priority"): 1 = the device's default priority (the default since
// round 4), 0 = highest, 2 = lowest.  Rounds 2-3 created the tails' streams at the HIGHEST priority; the runtime has fewer
// hardware queues for that class than a pipeline with a 4th worker, a loader stream or a second pool needs, and a stream
// beyond them shares a queue: 4 workers 1994 pages/s at priority 0 against 2554 at priority 1 (3 workers: 2545 either way),
// pages from host memory 2025-2380 against 2500, the dense-block pages 1828 against 1926-1993.
int g_tail_priority = 1;
// > 0: the tails' streams may only use this many CUs, mask bits [g_tail_cu_first, + g_tail_cus) (hipExtStreamCreateWithCUMask; the
// driver deals mask bits round-robin over the XCDs, so a contiguous run is the same share of every XCD); the caller gives the
// network's stream the complementary mask ("tail_cus", "tail_cu_first"; bench.py --cu-split)
int g_tail_cus = 0, g_tail_cu_first = 0;

#define T_TRY(expr)                                                                                  \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return ctd_fail_msg(CTD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int get(size_t bytes, void** out) {
    if (bytes > cap) {
      if (p) (void)hipFree(p);
      p = nullptr, cap = 0;
      const size_t want = bytes + bytes / 4 + 4096;
      hipError_t e = hipMalloc(&p, want);
      if (e != hipSuccess) return ctd_fail_msg(CTD_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
      cap = want;
    }
    *out = p;
    return CTD_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr, cap = 0;
  }
};

struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  int get(size_t bytes, void** out) {
    if (bytes > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr, cap = 0;
      const size_t want = bytes + bytes / 4 + 4096;
      hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
      if (e != hipSuccess) return ctd_fail_msg(CTD_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
      // the copy kernel (launch_multi_copy) addresses these buffers from the device by their host pointer
      void* dv = nullptr;
      if (hipHostGetDevicePointer(&dv, p, 0) != hipSuccess || dv != p) {
        (void)hipHostFree(p);
        p = nullptr;
        return ctd_fail_msg(CTD_ERR_HIP, "page-locked host memory is not device accessible at its host address");
      }
      cap = want;
    }
    *out = p;
    return CTD_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr, cap = 0;
  }
};

#define GET(buf, bytes, type, var) \
  type* var;                       \
  if (int rc_ = (buf).get((bytes), (void**)&var)) return rc_

struct PageOut {
  std::vector<ctd_blk> blks;
  std::vector<int32_t> lines;   // (n,8)
  std::vector<double> dist;     // (m,3)
  std::vector<int16_t> db_boxes;   // every contour's box (n,4,2) as `seg_rep` returns them
  std::vector<float> db_scores;
  std::vector<int32_t> yolo;    // (n,4) blines, then cls in yolo_cls, confs
  std::vector<int32_t> yolo_cls;
  std::vector<float> yolo_conf;
};

struct WinReq {
  int page, x1, y1, x2, y2;
};

constexpr int kMaxDet = 300;
constexpr int kCompCap = 1 << 16;   // components per polarity and page the compact DB path holds
constexpr int kRowCap = 1 << 18;    // row-table entries per page

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The per-page / per-window host work of a batch is independent: a few short-lived threads per call
// (a GPU host has cores to spare; the Python caller holds none of them while inside this library).
template <class F>
void parallel_for(int n, int max_threads, F f) {
  const int nt = std::min(max_threads, n);
  if (nt <= 1) {
    for (int i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<int> next{0};
  auto work = [&]() {
    for (int i; (i = next.fetch_add(1)) < n;) f(i);
  };
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  for (int k = 1; k < nt; ++k) th.emplace_back(work);
  work();
  for (auto& x : th) x.join();
}

}  // namespace
// Three work items' labellings at a time stretch the network's first kernel 3.5x, one at a time 1.4x (selftest ST_CORUN).
// "tail_chain": 1 = the stage-1 kernels (NMS, labelling, contour tables) of concurrent work items of this process run one
// after the other on the GPU (an event chain across their streams) instead of next to each other; 2 = the refine stage's
// big enqueue (render, labelling, accept rounds, dilate, labelling, holes) too; 0 = side by side.
int g_tail_chain = 1;
// "tail_fused_rounds": 1 = a window's merge rounds and its hole-filling passes as one launch each (a block per window);
// 0 = one count + one apply launch per round and four hole-filling launches over all windows (rounds 2-3)
int g_tail_fused_rounds = 1;
// ... for canvas-path window sets whose largest window has fewer pixels than this ("tail_fused_max_pix")
long long g_tail_fused_max_pix = 100000;
// "tail_lds": 1 = the merge stage of a window as ONE block on bit planes in LDS (kernels_twlds.hip), 0 = every window through
// the canvas path; "tail_lds_max_bytes": windows needing more LDS than this take the canvas path; "tail_lds_rcap" > 0: the
// run-table capacity of every launch (tests force overflows with a tiny one)
int g_tail_lds = 1, g_tail_lds_rcap = 0;
long long g_tail_lds_max_bytes = 150 << 10;
// the window-local kernel runs in up to three launches by LDS footprint ("tail_lds_cls0" / "tail_lds_cls1": the first two limits)
long long g_tail_lds_cls0 = 40 << 10, g_tail_lds_cls1 = 80 << 10;
namespace {
// The chain is PER DEVICE: events belong to the device that was current when they were created, a stream can only record
// its own device's events (hipErrorInvalidHandle otherwise), and tails on different GPUs have nothing to serialise.
struct ChainState {
  std::mutex mu;
  hipEvent_t last = nullptr;
  hipEvent_t ring[16] = {};
  unsigned next = 0;
};
constexpr int kChainDevices = 64;
inline ChainState* chain_state(int device) {
  static ChainState states[kChainDevices];
  return device >= 0 && device < kChainDevices ? &states[device] : nullptr;
}
struct GpuChain {
  hipStream_t st;
  ChainState* cs;                            // null: chaining off (or a device index beyond the table)
  std::unique_lock<std::mutex> lk;
  // `device` must be the current device of the calling thread (the entry points call hipSetDevice(t->device) first)
  GpuChain(hipStream_t s, int device, bool enabled) : st(s), cs(enabled ? chain_state(device) : nullptr) {
    if (cs) lk = std::unique_lock<std::mutex>(cs->mu, std::defer_lock);
  }
  hipError_t begin() {                       // the lock is held while this section is being enqueued (~0.1 ms of host time)
    if (!cs) return hipSuccess;
    lk.lock();
    return cs->last ? hipStreamWaitEvent(st, cs->last, 0) : hipSuccess;
  }
  hipError_t end() {
    if (!cs || !lk.owns_lock()) return hipSuccess;
    hipEvent_t& ev = cs->ring[cs->next++ % 16];
    hipError_t e = ev ? hipSuccess : hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, st);
    if (e == hipSuccess) cs->last = ev;
    lk.unlock();
    return e;
  }
};
}  // namespace
#ifdef CTD_MEASURE_KNOBS               // `make MEASURE=1` only: knobs that return INCOMPLETE results, for ablation timings
int g_tail_ablate = 0;                 // bit 1 = no refine stage ("tail_ablate")
int g_tail_skip_pages = 0;             // 1 = the page-size results are not downloaded ("tail_skip_page_download")
#endif
long long g_tail_dma_min = 256 << 10;   // device -> host copies of at least this many bytes use the copy engines ("tail_dma_min"; huge = never)
namespace {

inline void* device_view(const void* host);

// A stage's copies and fills as segments of one kernel launch (launch_multi_copy, kernels_tail.hip).
struct Batch {
  MSegs m;
  hipStream_t st;
  explicit Batch(hipStream_t s) : st(s) {}
  void seg(void* dst, const void* src, size_t row_bytes, int rows, long long dp, long long sp, int fill) {
    if (!row_bytes || rows < 1) return;
    if (m.n == kMSegMax) launch_multi_copy(m, st);
    MSeg& g = m.s[m.n++];
    g.dst = dst, g.src = src, g.row_bytes = row_bytes, g.rows = rows, g.dpitch = dp, g.spitch = sp, g.fill = fill, g.vec = 1, g.pad_ = 0;
  }
  void copy(void* dst, const void* src, size_t bytes) { seg(dst, src, bytes, 1, 0, 0, 0); }
  void copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int rows) {
    if (rows == 1 || (dpitch == width && spitch == width)) seg(dst, src, width * (size_t)rows, 1, 0, 0, 0);
    else seg(dst, src, width, rows, (long long)dpitch, (long long)spitch, 0);
  }
  void fill(void* dst, int byte, size_t bytes) { seg(dst, nullptr, bytes, 1, 0, 0, byte); }
  void flush() { launch_multi_copy(m, st); }
  // Device -> page-locked host.  Large ones go to the copy engines (hipMemcpyAsync): a KERNEL that stores to host memory
  // stalls every other kernel's memory traffic for as long as it runs (selftest ST_CORUN: the network's first kernel takes
  // 4.9 ms instead of 0.43 ms next to a kernel streaming 64 MB to pinned memory -- and the tail's mask downloads are
  // 64 MB per 32 pages).  Small ones stay segments of the stage's launch.  Earlier segments of this batch are launched
  // first: the stream order is the call order.
  hipError_t d2h(void* host, const void* dev, size_t bytes) {
    if (bytes < (size_t)g_tail_dma_min) { copy(host, dev, bytes); return hipSuccess; }
    flush();
    return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st);
  }
  // Device -> a CALLER's host array, which `ctd_hip.h` only promises to be host memory: a copy-kernel segment may store to
  // it only through its device-visible address, i.e. when it is page-locked; a pageable array (a C caller's malloc, an
  // unpinned numpy array) always goes through hipMemcpyAsync, whatever its size (a kernel store into pageable memory is a
  // memory fault without XNACK).
  hipError_t d2h_user(void* host, const void* dev, size_t bytes) {
    if (bytes && bytes < (size_t)g_tail_dma_min) {
      void* dv = device_view(host);
      // both ends page-locked and mapped contiguously (several arrays handed over as one back-to-back range)
      void* de = dv ? device_view((const char*)host + bytes - 1) : nullptr;
      if (dv && de == (char*)dv + bytes - 1) { copy(dv, dev, bytes); return hipSuccess; }
    }
    flush();
    return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st);
  }
  hipError_t h2d(void* dev, const void* host, size_t bytes) {      // page-locked host -> device, same rule
    if (bytes < (size_t)g_tail_dma_min) { copy(dev, host, bytes); return hipSuccess; }
    flush();
    return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st);
  }
  hipError_t d2h2d(void* host, size_t dpitch, const void* dev, size_t spitch, size_t width, int rows) {
    if (rows == 1 || (dpitch == width && spitch == width)) return d2h(host, dev, width * (size_t)rows);
    if (width * (size_t)rows < (size_t)g_tail_dma_min) { copy2d(host, dpitch, dev, spitch, width, rows); return hipSuccess; }
    flush();
    return hipMemcpy2DAsync(host, dpitch, dev, spitch, width, (size_t)rows, hipMemcpyDeviceToHost, st);
  }
};

// Device-visible address of a caller's host array when it is page-locked (hipHostMalloc / hipHostRegister), else null:
// the result arrays of ctd_tail_run are written by the copy kernel directly when they are, by hipMemcpyAsync otherwise.
inline void* device_view(const void* host) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, host) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (a.type == hipMemoryTypeHost && a.devicePointer) return a.devicePointer;
  return nullptr;
}

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct ctd_tail {
  int device = 0;
  hipStream_t st = nullptr;
  // device
  DevBuf d_dets, d_nms_ws, d_lab_f, d_ccl_ws, d_ccl_small, d_dbc_i, d_dbc_d, d_rows, d_pmask, d_refined;
  DevBuf d_wins, d_rules, d_bands, d_hist, d_sums, d_canvas, d_clab, d_cstats, d_cnt, d_merged, d_mlab, d_mstats, d_cnt2, d_small, d_crop;
  DevBuf d_wins_c, d_bands_c, d_lds;
  // pinned host
  PinBuf h_dets, h_hdr, h_tab, h_pmask, h_refined, h_hist, h_sums, h_wins, h_rules, h_bands, h_small, h_lab;
  PinBuf h_wins_c, h_bands_c, h_lds;
  // per-run state
  int B = 0;
  std::vector<ctd_tail_page> pages;
  std::vector<size_t> poff;           // byte offset of page b in the page-mask / refined buffers
  std::vector<uint8_t*> hmask;        // host copy of page b's mask during a run (the caller's array or h_pmask)
  size_t ptotal = 0;
  std::vector<PageOut> out;
  // host wall clock of the last run, ms: [0] enqueue of stage 1, [1] wait for stage 1, [2] table download +
  // contour geometry, [3] yolo unpack + group_output, [4] refine: histograms, [5] refine: xor sums, [6] refine:
  // enqueue of the merge stage, [7] undetected pass, [8] final wait + copies, [9] total
  double ms_stage[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double ms_db_wait = 0;              // inside [2]: waiting for the table download
  double ms_sub[5] = {0, 0, 0, 0, 0}; // inside [0]: NMS + buffers, labelling + contour tables, page-mask copies; inside [8]: the wait
  int host_threads = 8;               // threads of the per-page / per-window host loops
  // refine windows of the last run by path: window-local kernel, canvas path (too big, or after an overflow), overflows
  int n_lds = 0, n_canvas = 0, n_ovf = 0;
  long long lds_per_block = 64 << 10;  // hipDeviceAttributeMaxSharedMemoryPerBlock (160 KB on gfx950)
  double ms_lds_wait = 0;             // waiting for the window-local merge kernel's overflow flags
};

namespace {

// ---------------------------------------------------------------------------------------------------
// packing of rectangles into a canvas of fixed width: shelves, one empty column / row between bands
// ---------------------------------------------------------------------------------------------------
struct Packed {
  std::vector<int> x, y;
  int W = 0, H = 0;
};
void shelf_pack(const std::vector<int>& w, const std::vector<int>& h, int min_width, Packed& out) {
  const int n = (int)w.size();
  out.x.assign(n, 0);
  out.y.assign(n, 0);
  int cw = min_width;
  for (int i = 0; i < n; ++i) cw = std::max(cw, w[i]);
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h[a] > h[b]; });
  int x = 0, y = 0, shelf = 0;
  for (int i : order) {
    if (x > 0 && x + w[i] > cw) {
      y += shelf + 1;
      x = 0;
      shelf = 0;
    }
    out.x[i] = x, out.y[i] = y;
    x += w[i] + 1;
    shelf = std::max(shelf, h[i]);
  }
  out.W = cw;
  out.H = y + shelf;
}

// expand_textwindow(expand_r=16) of a block box (reference utils/imgproc_utils.py:151-161, utils/textmask.py:162)
bool block_window(const int32_t* xyxy, int im_w, int im_h, WinReq& wq) {
  int x1 = xyxy[0], y1 = xyxy[1], x2 = xyxy[2], y2 = xyxy[3];
  const int w = x2 - x1, h = y2 - y1;
  const int pad = (int)std::nearbyint(((double)std::max(h, w) * 0.25 + (double)std::min(h, w) * 0.75) / 16);
  x1 = std::max(0, x1 - pad), y1 = std::max(0, y1 - pad);
  x2 = std::min(im_w - 1, x2 + pad), y2 = std::min(im_h - 1, y2 + pad);
  if (x2 <= x1 || y2 <= y1) return false;
  wq.x1 = x1, wq.y1 = y1, wq.x2 = x2, wq.y2 = y2;
  return true;
}

// ---------------------------------------------------------------------------------------------------
// refine_mask for a list of windows over the pages of the batch (reference utils/textmask.py:159-169):
// ORs the merged window masks into the refined page buffers on the device.
// ---------------------------------------------------------------------------------------------------
// The canvas path of the merge stage for a SUBSET of the windows: candidates rendered into one packed canvas, one
// page-scale labelling launch, merge rounds, dilation, a second labelling of the complement, hole filling, commit
// (kernels_tail.hip + kernels_post.hip).  Since round 6 this is where windows go whose bit planes do not fit the
// window-local kernel's LDS (kernels_twlds.hip) or whose run table overflowed there.  `hw` / `bands`: the full tables.
int refine_canvas(ctd_tail* t, const TWin* hw, const std::vector<TBand>& bands, const std::vector<int>& subset, int refine_mode) {
  const int n = (int)subset.size();
  if (n == 0) return CTD_OK;
  hipStream_t st = t->st;
  std::vector<int> ww(n), wh(n);
  int max_pix = 1;
  for (int k = 0; k < n; ++k) {
    ww[k] = hw[subset[k]].w, wh[k] = hw[subset[k]].h;
    max_pix = std::max(max_pix, ww[k] * wh[k]);
  }
  Packed pm;
  shelf_pack(ww, wh, 2048, pm);
  GET(t->h_wins_c, sizeof(TWin) * n, TWin, cw);
  std::vector<TBand> cb;
  std::vector<int> bw, bh;
  int rounds = 0;
  for (int k = 0; k < n; ++k) {
    cw[k] = hw[subset[k]];
    cw[k].mx = pm.x[k], cw[k].my = pm.y[k];
    const int b0 = cw[k].band0;
    cw[k].band0 = (int)cb.size();
    rounds = std::max(rounds, cw[k].nband);
    for (int r = 0; r < cw[k].nband; ++r) {
      TBand b = bands[b0 + r];
      b.win = k;
      cb.push_back(b);
      bw.push_back(ww[k]);
      bh.push_back(wh[k]);
    }
  }
  const int nbands = (int)cb.size();
  Packed pc;
  shelf_pack(bw, bh, 2048, pc);
  long long bound1 = 1, bound2 = 1;                      // 8-connected components: at most one per 2x2 cell
  for (int j = 0; j < nbands; ++j) {
    cb[j].cx = pc.x[j], cb[j].cy = pc.y[j];
    bound1 += (long long)((bw[j] + 1) / 2) * ((bh[j] + 1) / 2);
  }
  for (int k = 0; k < n; ++k) bound2 += (long long)((ww[k] + 1) / 2) * ((wh[k] + 1) / 2);
  if ((long long)pc.W * pc.H >= (1LL << 30) || bound1 >= (1LL << 28) || (long long)pm.W * pm.H >= (1LL << 30) || bound2 >= (1LL << 28))
    return ctd_fail_msg(CTD_ERR_UNSUPPORTED, "refine: too many window pixels in one batch");
  const int cap1 = (int)bound1, cap2 = (int)bound2;
  const size_t mpx = (size_t)pm.W * pm.H, cpx = (size_t)pc.W * pc.H;
  GET(t->d_wins_c, sizeof(TWin) * n, TWin, dw);
  GET(t->h_bands_c, sizeof(TBand) * std::max(nbands, 1), TBand, hb);
  std::memcpy(hb, cb.data(), sizeof(TBand) * nbands);
  GET(t->d_bands_c, sizeof(TBand) * std::max(nbands, 1), TBand, db);
  GET(t->d_merged, mpx * 3, uint8_t, merged_a);
  uint8_t* merged_b = merged_a + mpx;
  uint8_t* comp = merged_a + 2 * mpx;
  GET(t->d_small, (size_t)n * 16 + 64, int, small);       // [n2 (1) | pad | count255 (n) | top2 (n,3)]
  int* n_dev = small;
  unsigned* count255 = (unsigned*)(small + 16);
  int* top2 = small + 16 + n;
  GET(t->d_cnt2, ((size_t)cap2 + 1) * 8, unsigned, counters2);
  GET(t->d_canvas, cpx, uint8_t, canvas);
  GET(t->d_clab, cpx * 4, int, clab);
  GET(t->d_cstats, (size_t)cap1 * 5 * 4, int, cstats);
  GET(t->d_ccl_ws, ccl_workspace_bytes(1, std::max(pc.H, pm.H), std::max(pc.W, pm.W)), uint8_t, ws);
  GET(t->d_cnt, ((size_t)cap1 + 1) * 8, unsigned, counters);
  Batch bt(st);
  T_TRY(bt.h2d(dw, cw, sizeof(TWin) * n));
  T_TRY(bt.h2d(db, hb, sizeof(TBand) * nbands));
  bt.fill(merged_a, 0, mpx * 3);
  bt.fill(count255, 0, (size_t)n * 4);
  bt.fill(top2, 0xFF, (size_t)n * 12);
  bt.fill(canvas, 0, cpx);
  bt.flush();
  // One block per window for all merge rounds / hole passes (`tw_accept_all`, `tw_holes_all`) suits windows of a few 10 K
  // pixels; since round 6 those are merged in LDS and what arrives here is mostly LARGE (page-sized windows of
  // refine_undetected_mask's left-over components: one block walked 1 M pixels six times, 5-7 ms per launch next to a
  // 10-ms forward) -- those take the per-round grid kernels, many blocks per window.  Same results either way
  // (test_fused_merge_rounds_equal_the_per_round_launches).
  const bool fused = g_tail_fused_rounds && max_pix < g_tail_fused_max_pix;
  GpuChain chain(st, t->device, g_tail_chain >= 2);
  T_TRY(chain.begin());
  launch_tw_render(dw, db, nbands, max_pix, canvas, pc.W, st);
  launch_ccl(canvas, 1, pc.H, pc.W, 0, 8, clab, n_dev, cstats, cap1, ws, st, 0, nullptr, 1);   // the window kernels test `label > 0`
  launch_label_counters_zero(counters, n_dev, cap1, st);
  if (fused) {
    launch_tw_accept_all(dw, db, n, clab, pc.W, cstats, cap1, 3, merged_a, pm.W, counters, st);
  } else {
    for (int r = 0; r < rounds; ++r)
      launch_tw_accept(dw, db, nbands, max_pix, r, clab, pc.W, cstats, cap1, 3, merged_a, pm.W, counters, st);
  }
  launch_tw_dilate(dw, n, max_pix, merged_a, merged_b, comp, pm.W, count255, refine_mode == 0 ? 1 : 0, st);
  // ---- hole filling on the complement, then the OR into the pages
  GET(t->d_mlab, mpx * 4, int, mlab);
  GET(t->d_mstats, (size_t)cap2 * 6 * 4, int, mstats);
  int* mfirst = mstats + (size_t)cap2 * 5;
  launch_ccl(comp, 1, pm.H, pm.W, 0, 8, mlab, n_dev, mstats, cap2, ws, st, 0, mfirst, 1);
  launch_label_counters_zero(counters2, n_dev, cap2, st);
  if (fused) launch_tw_holes_all(dw, n, mlab, mstats, mfirst, cap2, count255, merged_b, pm.W, counters2, st);
  else launch_tw_holes(dw, n, max_pix, mlab, mstats, mfirst, cap2, count255, top2, merged_b, pm.W, counters2, st);
  launch_tw_commit(dw, n, max_pix, merged_b, pm.W, st);
  T_TRY(chain.end());
  T_TRY(hipGetLastError());
  return CTD_OK;
}

// ---------------------------------------------------------------------------------------------------
// refine_mask for a list of windows over the pages of the batch (reference utils/textmask.py:159-169):
// ORs the merged window masks into the refined page buffers on the device.
//   histograms -> host rules -> xor distances -> host polarity / merge order          (2 waits, as before)
//   merge stage: ONE block per window from render to commit, on bit planes in LDS (kernels_twlds.hip), in up to three
//   launches by LDS footprint; windows too big for that -> refine_canvas; after the wait for the overflow flags, windows
//   whose run table overflowed -> refine_canvas too
// ---------------------------------------------------------------------------------------------------
int refine_windows(ctd_tail* t, const std::vector<WinReq>& reqs, int refine_mode) {
  const int n = (int)reqs.size();
  if (n == 0) return CTD_OK;
  hipStream_t st = t->st;
  GET(t->d_pmask, 0, uint8_t, pmask);
  GET(t->d_refined, 0, uint8_t, refined);
  std::vector<int> ww(n), wh(n);
  int max_pix = 1;
  for (int i = 0; i < n; ++i) {
    ww[i] = reqs[i].x2 - reqs[i].x1, wh[i] = reqs[i].y2 - reqs[i].y1;
    max_pix = std::max(max_pix, ww[i] * wh[i]);
  }
  GET(t->h_wins, sizeof(TWin) * n, TWin, hw);
  for (int i = 0; i < n; ++i) {
    const ctd_tail_page& pg = t->pages[reqs[i].page];
    TWin& w = hw[i];
    w.img = pg.img_dev;
    w.mask = pmask + t->poff[reqs[i].page];
    w.out = refined + t->poff[reqs[i].page];
    w.img_w = pg.im_w, w.mask_w = pg.im_w, w.out_w = pg.im_w;
    w.x1 = reqs[i].x1, w.y1 = reqs[i].y1, w.w = ww[i], w.h = wh[i];
    w.mx = 0, w.my = 0;
    w.band0 = 0, w.nband = 0;
  }
  GET(t->d_wins, sizeof(TWin) * n, TWin, dw);
  // ---- one launch uploads the window table and zeroes the histograms and the xor sums
  GET(t->d_hist, (size_t)n * 1024 * 4, unsigned, dhist);
  GET(t->h_hist, (size_t)n * 1024 * 4, uint32_t, hhist);
  GET(t->d_sums, (size_t)n * 6 * 8, unsigned long long, dsums);
  GET(t->h_sums, (size_t)n * 6 * 8, uint64_t, hsums);
  Batch bt(st);
  T_TRY(bt.h2d(dw, hw, sizeof(TWin) * n));
  bt.fill(dhist, 0, (size_t)n * 1024 * 4);
  bt.fill(dsums, 0, (size_t)n * 6 * 8);
  bt.flush();
  // ---- histograms -> rules
  launch_tw_hist(dw, n, max_pix, dhist, st);
  T_TRY(bt.d2h(hhist, dhist, (size_t)n * 1024 * 4));
  bt.flush();
  const double tr0 = now_ms();
  T_TRY(hipStreamSynchronize(st));
  const double tr1 = now_ms();
  GET(t->h_rules, sizeof(RRule) * 6 * n, RRule, hrules);
  parallel_for(n, t->host_threads, [&](int i) { refine_rules(hhist + (size_t)i * 1024, hrules + (size_t)i * 6); });
  static_assert(sizeof(RRule) == sizeof(TRule), "rule layouts must agree");
  GET(t->d_rules, sizeof(TRule) * 6 * n, TRule, drules);
  T_TRY(bt.h2d(drules, hrules, sizeof(TRule) * 6 * n));
  bt.flush();
  // ---- xor distances -> polarity and merge order
  launch_tw_xor(dw, drules, n, max_pix, dsums, st);
  T_TRY(bt.d2h(hsums, dsums, (size_t)n * 6 * 8));
  bt.flush();
  const double tr2 = now_ms();
  T_TRY(hipStreamSynchronize(st));
  const double tr3 = now_ms();
  std::vector<TBand> bands;
  for (int i = 0; i < n; ++i) {
    RCand c[4];
    const int nc = refine_candidates(hrules + (size_t)i * 6, hsums + (size_t)i * 6, (long long)ww[i] * wh[i], c);
    hw[i].band0 = (int)bands.size(), hw[i].nband = nc;     // the window's bands are contiguous, in merge order
    for (int r = 0; r < nc; ++r) {
      const RRule& rl = hrules[(size_t)i * 6 + c[r].rule];
      TBand b;
      b.win = i, b.cx = b.cy = 0, b.kind = rl.kind, b.lo = rl.lo, b.hi = rl.hi, b.invert = c[r].invert, b.round = r;
      bands.push_back(b);
    }
  }
  const int nbands = (int)bands.size();
  // ---- who goes where: by the LDS a window's planes + run table need
  const long long lds_max = std::min(g_tail_lds_max_bytes, t->lds_per_block - 2048);
  std::vector<int> lds_win, canvas_win;
  std::vector<int> words(n);
  for (int i = 0; i < n; ++i) {
    words[i] = ((ww[i] + 31) >> 5) * wh[i];
    const int rc = g_tail_lds_rcap > 0 ? g_tail_lds_rcap : tw_lds_rcap(words[i]);
    if (g_tail_lds && (long long)tw_lds_bytes(words[i], std::max(rc, (words[i] + 1) / 2)) <= lds_max) lds_win.push_back(i);
    else canvas_win.push_back(i);
  }
  int nl = (int)lds_win.size();
  int* hovf = nullptr;
  if (nl) {
    std::stable_sort(lds_win.begin(), lds_win.end(), [&](int a, int b) { return words[a] < words[b]; });
    GET(t->h_bands, sizeof(TBand) * std::max(nbands, 1), TBand, hb);
    std::memcpy(hb, bands.data(), sizeof(TBand) * nbands);
    GET(t->d_bands, sizeof(TBand) * std::max(nbands, 1), TBand, db);
    GET(t->h_lds, (size_t)n * 8, int, hl);                  // [order (n) | overflow flags (n)]
    GET(t->d_lds, (size_t)n * 8, int, dl);
    std::memcpy(hl, lds_win.data(), sizeof(int) * nl);
    hovf = hl + n;
    T_TRY(bt.h2d(db, hb, sizeof(TBand) * nbands));
    T_TRY(bt.h2d(dw, hw, sizeof(TWin) * n));                // again: with every window's band range (the stream is idle here)
    T_TRY(bt.h2d(dl, hl, sizeof(int) * nl));
    bt.fill(dl + n, 0, sizeof(int) * n);
    bt.flush();
    // up to three launches: a block's dynamic LDS is its launch's largest window's, and small blocks share a CU
    const long long cls[3] = {g_tail_lds_cls0, g_tail_lds_cls1, lds_max};
    std::vector<int> refused;                               // a launch that did not get its LDS: those windows take the canvases
    int k0 = 0;
    for (int c = 0; c < 3 && k0 < nl; ++c) {
      int k1 = k0;
      auto need = [&](int k) {
        const int wd = words[lds_win[k]];
        const int rc = g_tail_lds_rcap > 0 ? g_tail_lds_rcap : tw_lds_rcap(wd);
        return (long long)tw_lds_bytes(wd, std::max(rc, (wd + 1) / 2));
      };
      while (k1 < nl && (c == 2 || need(k1) <= cls[c])) ++k1;
      if (k1 > k0) {
        const int mw = words[lds_win[k1 - 1]];
        if (!launch_tw_lds(dw, db, dl + k0, k1 - k0, mw, g_tail_lds_rcap > 0 ? g_tail_lds_rcap : tw_lds_rcap(mw),
                           refine_mode == 0 ? 1 : 0, dl + n, st))
          for (int k = k0; k < k1; ++k) refused.push_back(lds_win[k]);
      }
      k0 = k1;
    }
    T_TRY(bt.d2h(hovf, dl + n, sizeof(int) * n));
    bt.flush();
    if (!refused.empty()) {
      std::sort(refused.begin(), refused.end());
      canvas_win.insert(canvas_win.end(), refused.begin(), refused.end());
      std::sort(canvas_win.begin(), canvas_win.end());
      std::vector<int> kept;
      for (int i : lds_win)
        if (!std::binary_search(refused.begin(), refused.end(), i)) kept.push_back(i);
      lds_win.swap(kept);
      nl = (int)lds_win.size();
    }
  }
  if (int rc = refine_canvas(t, hw, bands, canvas_win, refine_mode)) return rc;
  const double tr4 = now_ms();
  double tr5 = tr4;
  if (nl) {
    T_TRY(hipStreamSynchronize(st));                        // the overflow flags (the merge stage has run when they arrive)
    tr5 = now_ms();
    std::vector<int> again;
    for (int k = 0; k < nl; ++k)
      if (hovf[lds_win[k]]) again.push_back(lds_win[k]);
    std::sort(again.begin(), again.end());
    t->n_lds += nl - (int)again.size(), t->n_canvas += (int)canvas_win.size() + (int)again.size(), t->n_ovf += (int)again.size();
    if (int rc = refine_canvas(t, hw, bands, again, refine_mode)) return rc;
  } else {
    t->n_canvas += (int)canvas_win.size();
  }
  T_TRY(hipGetLastError());
  t->ms_stage[4] += tr1 - tr0;
  t->ms_stage[5] += tr3 - tr2;
  t->ms_stage[6] += (now_ms() - tr5) + (tr4 - tr3) + (tr2 - tr1);
  t->ms_lds_wait += tr5 - tr4;
  return CTD_OK;
}

// Page-mask / refined buffers: page b at byte offset poff[b], im_h * im_w bytes, padded to 256.
int layout_pages(ctd_tail* t, int B, const ctd_tail_page* pages) {
  t->B = B;
  t->pages.assign(pages, pages + B);
  t->poff.resize(B);
  size_t off = 0;
  for (int b = 0; b < B; ++b) {
    if (pages[b].im_h < 1 || pages[b].im_w < 1) return ctd_fail_msg(CTD_ERR_INVALID, "bad page size");
    t->poff[b] = off;
    // (a page whose size is a multiple of 256 bytes needs no padding: pages of one such size then lie back to back, and
    // refine_undetected_mask labels them in ONE launch)
    const size_t px = (size_t)pages[b].im_h * pages[b].im_w;
    off += px % 256 == 0 ? px : align_up(px + 4, 256);
  }
  t->ptotal = off;
  t->out.assign(B, PageOut());
  return CTD_OK;
}

// refine_undetected_mask (reference utils/textmask.py:135-156) for every page of the batch: the masks are
// edited in place on the device, components of what is left become extra blocks when no detected
// block covers half of their box, and those blocks get their own refine pass.
int undetected_pass(ctd_tail* t, const std::vector<std::vector<int32_t>>& blk_xyxy, int refine_mode) {
  hipStream_t st = t->st;
  const int B = t->B;
  GET(t->d_pmask, 0, uint8_t, pmask);
  GET(t->d_refined, 0, uint8_t, refined);
  launch_mask_clear_where(pmask, refined, (long long)t->ptotal, 30, st);
  // labelling page by page (pages may differ in size); stats rows are downloaded per page
  const int cap = kCompCap;
  size_t max_px = 0;
  for (int b = 0; b < B; ++b) max_px = std::max(max_px, (size_t)t->pages[b].im_h * t->pages[b].im_w);
  // pages of one size that lie back to back (layout_pages) are labelled in one launch; else page by page
  bool same = true;
  for (int b = 1; b < B; ++b)
    same = same && t->pages[b].im_h == t->pages[0].im_h && t->pages[b].im_w == t->pages[0].im_w &&
           t->poff[b] == t->poff[b - 1] + max_px;
  const int nb = same ? B : 1;
  GET(t->d_lab_f, (size_t)nb * max_px * 4, int, lab);
  GET(t->d_ccl_ws, same ? ccl_workspace_bytes(B, t->pages[0].im_h, t->pages[0].im_w) : ccl_workspace_bytes(1, 1, (int)max_px), uint8_t, ws);
  GET(t->d_ccl_small, (size_t)B * 4 + (size_t)B * cap * 5 * 4, int, nst);
  int* n_dev = nst;
  int* st_dev = nst + B;
  GET(t->h_small, (size_t)B * 4, int, n_host);
  if (same) {
    launch_ccl(pmask + t->poff[0], B, t->pages[0].im_h, t->pages[0].im_w, 30, 4, lab, n_dev, st_dev, cap, ws, st, 0, nullptr, 1);
  } else {
    for (int b = 0; b < B; ++b)
      launch_ccl(pmask + t->poff[b], 1, t->pages[b].im_h, t->pages[b].im_w, 30, 4, lab, n_dev + b,
                 st_dev + (size_t)b * cap * 5, cap, ws, st, 0, nullptr, 1);   // only the statistics are used
  }
  T_TRY(hipMemcpyAsync(n_host, n_dev, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  T_TRY(hipStreamSynchronize(st));
  int nmax = 0;
  for (int b = 0; b < B; ++b) nmax = std::max(nmax, std::min(n_host[b], cap));
  std::vector<WinReq> reqs;
  if (nmax > 0) {
    GET(t->h_tab, (size_t)B * nmax * 5 * 4, int, sth);
    T_TRY(hipMemcpy2DAsync(sth, (size_t)nmax * 20, st_dev, (size_t)cap * 20, (size_t)nmax * 20, B, hipMemcpyDeviceToHost, st));
    T_TRY(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) {
      const int n = std::min(n_host[b], cap);
      const int im_w = t->pages[b].im_w, im_h = t->pages[b].im_h;
      const int* s = sth + (size_t)b * nmax * 5;
      long long fg = 0;
      for (int l = 0; l < n; ++l) fg += s[5 * l + 4];
      // the reference's stats include the background row 0; `valid_labels[1:]` drops the FIRST row with
      // area > 50, which is the background whenever that has more than 50 pixels (:139-142)
      bool first_dropped = false;
      if ((long long)im_w * im_h - fg > 50) first_dropped = true;
      const std::vector<int32_t>& bx = blk_xyxy[b];
      for (int l = 0; l < n; ++l) {
        if (s[5 * l + 4] <= 50) continue;
        if (!first_dropped) {
          first_dropped = true;
          continue;
        }
        const int x = s[5 * l], y = s[5 * l + 1], w = s[5 * l + 2], h = s[5 * l + 3];
        long long best = -1;
        for (size_t k = 0; k + 3 < bx.size(); k += 4) {
          const int ix1 = std::max(bx[k], x), iy1 = std::max(bx[k + 1], y);
          const int ix2 = std::min(bx[k + 2], x + w), iy2 = std::min(bx[k + 3], y + h);
          const long long sc = (iy2 < iy1 || ix2 < ix1) ? -1 : (long long)(iy2 - iy1) * (ix2 - ix1);
          best = std::max(best, sc);
        }
        if ((double)best / (double)w / (double)h < 0.5) {
          const int32_t q[4] = {x, y, x + w, y + h};
          WinReq wq;
          wq.page = b;
          if (block_window(q, im_w, im_h, wq)) reqs.push_back(wq);
        }
      }
    }
  }
  return refine_windows(t, reqs, refine_mode);
}

// device -> host of the page-size outputs, straight into the caller's arrays (page-locked arrays make
// these DMA transfers; comic-text-detector_amd/tail.py allocates them pinned)
int download_pages(ctd_tail* t, bool mask_too, uint8_t* const* mask_out, uint8_t* const* refined_out) {
#ifdef CTD_MEASURE_KNOBS
  if (g_tail_skip_pages) return CTD_OK;    // ("tail_skip_page_download"): what do the page downloads cost the step?
#endif
  hipStream_t st = t->st;
  GET(t->d_pmask, 0, uint8_t, pmask);
  GET(t->d_refined, 0, uint8_t, refined);
  Batch bt(st);
  auto one = [&](uint8_t* host, const uint8_t* dev, size_t nb) -> hipError_t {
    if (nb < (size_t)g_tail_dma_min)
      if (void* dv = device_view(host)) {                // small page-locked array: a segment of the copy kernel
        bt.copy(dv, dev, nb);
        return hipSuccess;
      }
    return hipMemcpyAsync(host, dev, nb, hipMemcpyDeviceToHost, st);   // page-locked arrays make these DMA transfers
  };
  // the caller's arrays usually lie back to back (one pinned allocation per batch): equal-size pages then go as ONE
  // strided copy (device pitch = the padded page slot) instead of one copy per page
  const size_t nb0 = (size_t)t->pages[0].im_h * t->pages[0].im_w;
  const size_t slot = t->B > 1 ? t->poff[1] - t->poff[0] : nb0;
  auto dense = [&](uint8_t* const* out) {
    if (!out || !out[0] || t->B < 2) return false;
    for (int b = 0; b < t->B; ++b) {
      if ((size_t)t->pages[b].im_h * t->pages[b].im_w != nb0 || t->poff[b] != (size_t)b * slot) return false;
      if (out[b] != out[0] + (size_t)b * nb0) return false;
    }
    return true;
  };
  const bool dm = mask_too && dense(mask_out), dr = dense(refined_out);
  if (dm) T_TRY(hipMemcpy2DAsync(mask_out[0], nb0, pmask, slot, nb0, (size_t)t->B, hipMemcpyDeviceToHost, st));
  if (dr) T_TRY(hipMemcpy2DAsync(refined_out[0], nb0, refined, slot, nb0, (size_t)t->B, hipMemcpyDeviceToHost, st));
  for (int b = 0; b < t->B; ++b) {
    const size_t nb = (size_t)t->pages[b].im_h * t->pages[b].im_w;
    if (mask_too && !dm && mask_out && mask_out[b]) T_TRY(one(mask_out[b], pmask + t->poff[b], nb));
    if (!dr && refined_out && refined_out[b]) T_TRY(one(refined_out[b], refined + t->poff[b], nb));
  }
  bt.flush();
  T_TRY(hipGetLastError());
  const double w0 = now_ms();
  T_TRY(hipStreamSynchronize(st));
  t->ms_sub[3] = now_ms() - w0;
  return CTD_OK;
}

// ---------------------------------------------------------------------------------------------------
// DB text-line stage (reference utils/db_utils.py:32-211) for a batch: `db_enqueue` launches the two
// labelling passes and the contour-table kernels and starts the download of the per-page counts;
// after the caller's stream synchronisation `db_collect` fetches the tables at their actual sizes and
// runs the host geometry, filling PageOut::db_boxes / db_scores of every page.
// ---------------------------------------------------------------------------------------------------
struct DbStage {
  int B = 0, Hn = 0, Wn = 0;
  const float* prob = nullptr;
  long long prob_stride = 0;
  int *lab = nullptr, *n_f = nullptr, *n_b = nullptr, *st_f = nullptr, *st_b = nullptr;
  int *first_f = nullptr, *first_b = nullptr, *par_f = nullptr, *par_b = nullptr, *off_f = nullptr, *off_b = nullptr;
  int *hdr = nullptr, *ring_cnt = nullptr, *row_lo = nullptr, *row_hi = nullptr;
  double *sum_f = nullptr, *sum_b = nullptr, *ring_sum = nullptr;
  int* hhdr = nullptr;
};

// `pre`: fills this stage needs before its kernels (the caller may have added its own; flushed here);
// `post`: receives the download of the per-page table header (the caller flushes it with its own copies).
int db_enqueue(ctd_tail* t, DbStage& d, int B, int Hn, int Wn, const float* prob_dev, long long prob_stride,
               const uint8_t* bitmap_dev, Batch& pre, Batch& post) {
  hipStream_t st = t->st;
  const size_t hw = (size_t)Hn * Wn;
  const int cap = kCompCap, rcap = kRowCap;
  d.B = B, d.Hn = Hn, d.Wn = Wn, d.prob = prob_dev, d.prob_stride = prob_stride;
  GET(t->d_lab_f, (size_t)B * hw * 4, int, lab);
  GET(t->d_ccl_ws, ccl_workspace_bytes(B, Hn, Wn), uint8_t, ccl_ws);
  // int tables: n_f, n_b (B each) | st_f, st_b (B,cap,5) | first, par, off x2 (B,cap) | hdr (B,4) | ring_cnt (B,cap)
  const size_t bc = (size_t)B * cap;
  GET(t->d_dbc_i, (2 * (size_t)B + 10 * bc + 6 * bc + 4 * (size_t)B + bc) * 4, int, ti);
  d.lab = lab;
  d.n_f = ti;
  d.n_b = d.n_f + B;
  d.st_f = d.n_b + B;
  d.st_b = d.st_f + 5 * bc;
  d.first_f = d.st_b + 5 * bc;
  d.first_b = d.first_f + bc;
  d.par_f = d.first_b + bc;
  d.par_b = d.par_f + bc;
  d.off_f = d.par_b + bc;
  d.off_b = d.off_f + bc;
  d.hdr = d.off_b + bc;
  d.ring_cnt = d.hdr + 4 * (size_t)B;
  GET(t->d_dbc_d, 3 * bc * 8, double, td);
  d.sum_f = td, d.sum_b = td + bc, d.ring_sum = td + 2 * bc;
  GET(t->d_rows, 2 * (size_t)B * rcap * 4, int, rows);
  d.row_lo = rows, d.row_hi = rows + (size_t)B * rcap;
  // foreground 8-connected + background 4-connected components in one pass: signed label image
  pre.fill(td, 0, 3 * bc * 8);
  pre.fill(d.ring_cnt, 0, bc * 4);
  pre.flush();
  launch_ccl_dual(bitmap_dev, B, Hn, Wn, 0, lab, d.n_f, d.n_b, d.st_f, d.st_b, d.first_f, d.first_b, cap, ccl_ws, st);
  DbcTables dt;
  dt.B = B, dt.H = Hn, dt.W = Wn, dt.cap = cap, dt.rcap = rcap;
  dt.prob = prob_dev, dt.prob_stride = prob_stride;
  dt.lab = lab, dt.n_f = d.n_f, dt.n_b = d.n_b, dt.st_f = d.st_f, dt.st_b = d.st_b;
  dt.first_f = d.first_f, dt.first_b = d.first_b, dt.par_f = d.par_f, dt.par_b = d.par_b, dt.off_f = d.off_f;
  dt.off_b = d.off_b, dt.hdr = d.hdr, dt.row_lo = d.row_lo, dt.row_hi = d.row_hi, dt.sum_f = d.sum_f, dt.sum_b = d.sum_b;
  dt.ring_sum = d.ring_sum, dt.ring_cnt = d.ring_cnt;
  launch_dbc(dt, st);
  GET(t->h_hdr, (size_t)B * 4 * 4, int, hhdr);
  d.hhdr = hhdr;
  T_TRY(post.d2h(hhdr, d.hdr, (size_t)B * 16));
  T_TRY(hipGetLastError());
  return CTD_OK;
}

int db_collect(ctd_tail* t, const DbStage& d, const ctd_tail_params* prm) {
  hipStream_t st = t->st;
  const int B = d.B, Hn = d.Hn, Wn = d.Wn, cap = kCompCap, rcap = kRowCap;
  const size_t hw = (size_t)Hn * Wn;
  const int* hhdr = d.hhdr;
  int nfm = 0, nbm = 0, rwm = 0;
  for (int b = 0; b < B; ++b)
    nfm = std::max(nfm, hhdr[4 * b]), nbm = std::max(nbm, hhdr[4 * b + 1]), rwm = std::max(rwm, std::min(hhdr[4 * b + 2], rcap));
  // pinned layout: per table a (B, max) block
  const size_t nf = (size_t)std::max(nfm, 1), nb = (size_t)std::max(nbm, 1), nr = (size_t)std::max(rwm, 1);
  const size_t tab_bytes = (size_t)B * (nf * (5 + 3) * 4 + nf * 8 + nb * (5 + 3 + 1) * 4 + nb * 16 + nr * 8) + 14 * 8;
  GET(t->h_tab, tab_bytes + 256, uint8_t, tab);
  uint8_t* cur = tab;
  auto take = [&](size_t bytes) {
    uint8_t* p = cur;
    cur += align_up(bytes, 8);
    return p;
  };
  int* h_st_f = (int*)take((size_t)B * nf * 20);
  int* h_first_f = (int*)take((size_t)B * nf * 4);
  int* h_par_f = (int*)take((size_t)B * nf * 4);
  int* h_off_f = (int*)take((size_t)B * nf * 4);
  double* h_sum_f = (double*)take((size_t)B * nf * 8);
  int* h_st_b = (int*)take((size_t)B * nb * 20);
  int* h_first_b = (int*)take((size_t)B * nb * 4);
  int* h_par_b = (int*)take((size_t)B * nb * 4);
  int* h_off_b = (int*)take((size_t)B * nb * 4);
  int* h_ring_cnt = (int*)take((size_t)B * nb * 4);
  double* h_sum_b = (double*)take((size_t)B * nb * 8);
  double* h_ring_sum = (double*)take((size_t)B * nb * 8);
  int* h_row_lo = (int*)take((size_t)B * nr * 4);
  int* h_row_hi = (int*)take((size_t)B * nr * 4);
  Batch bt(st);
  auto d2h = [&](void* dst, const void* src, size_t elem, size_t n_used, size_t n_cap) -> hipError_t {
    return bt.d2h2d(dst, n_used * elem, src, n_cap * elem, n_used * elem, B);
  };
  if (nfm > 0) {
    T_TRY(d2h(h_st_f, d.st_f, 20, nf, cap));
    T_TRY(d2h(h_first_f, d.first_f, 4, nf, cap));
    T_TRY(d2h(h_par_f, d.par_f, 4, nf, cap));
    T_TRY(d2h(h_off_f, d.off_f, 4, nf, cap));
    T_TRY(d2h(h_sum_f, d.sum_f, 8, nf, cap));
  }
  if (nbm > 0) {
    T_TRY(d2h(h_st_b, d.st_b, 20, nb, cap));
    T_TRY(d2h(h_first_b, d.first_b, 4, nb, cap));
    T_TRY(d2h(h_par_b, d.par_b, 4, nb, cap));
    T_TRY(d2h(h_off_b, d.off_b, 4, nb, cap));
    T_TRY(d2h(h_ring_cnt, d.ring_cnt, 4, nb, cap));
    T_TRY(d2h(h_sum_b, d.sum_b, 8, nb, cap));
    T_TRY(d2h(h_ring_sum, d.ring_sum, 8, nb, cap));
  }
  if (rwm > 0) {
    T_TRY(d2h(h_row_lo, d.row_lo, 4, nr, rcap));
    T_TRY(d2h(h_row_hi, d.row_hi, 4, nr, rcap));
  }
  bt.flush();
  T_TRY(hipGetLastError());
  const double td0 = now_ms();
  T_TRY(hipStreamSynchronize(st));
  t->ms_db_wait = now_ms() - td0;
  const int maxc = std::max(prm->max_candidates, 0);
  std::atomic<int> err{CTD_OK};
  parallel_for(B, t->host_threads, [&](int b) {
    if (hhdr[4 * b + 3]) return;                        // overflowed page: handled below, one at a time
    PageOut& po = t->out[b];
    po.db_boxes.assign((size_t)std::max(maxc, 1) * 8, 0);
    po.db_scores.assign((size_t)std::max(maxc, 1), 0.f);
    int nbox = 0;
    const int rc = ctd_db_boxes_compact(Wn, Hn, hhdr[4 * b], h_st_f + (size_t)b * nf * 5, h_first_f + (size_t)b * nf,
                                        h_par_f + (size_t)b * nf, h_off_f + (size_t)b * nf, h_sum_f + (size_t)b * nf,
                                        hhdr[4 * b + 1], h_st_b + (size_t)b * nb * 5, h_first_b + (size_t)b * nb,
                                        h_par_b + (size_t)b * nb, h_off_b + (size_t)b * nb, h_sum_b + (size_t)b * nb,
                                        h_ring_sum + (size_t)b * nb, h_ring_cnt + (size_t)b * nb, h_row_lo + (size_t)b * nr,
                                        h_row_hi + (size_t)b * nr, maxc, prm->unclip_ratio, po.db_boxes.data(),
                                        po.db_scores.data(), &nbox);
    if (rc) err = rc;
    po.db_boxes.resize((size_t)nbox * 8);
    po.db_scores.resize(nbox);
  });
  if (err) return ctd_fail_msg(err, "ctd_db_boxes_compact failed");
  std::vector<int16_t> boxes((size_t)std::max(maxc, 1) * 8);
  std::vector<float> scores((size_t)std::max(maxc, 1));
  for (int b = 0; b < B; ++b) {
    if (!hhdr[4 * b + 3]) continue;
    PageOut& po = t->out[b];
    int nbox = 0;
    {
      // more components than the compact tables hold (a noise bitmap): label images to the host
      std::vector<int32_t> lab_host_f(hw), lab_host_b(hw);
      std::vector<float> prob_h(hw);
      int nfb[2];
      T_TRY(hipMemcpyAsync(nfb, d.n_f + b, 4, hipMemcpyDeviceToHost, st));
      T_TRY(hipMemcpyAsync(nfb + 1, d.n_b + b, 4, hipMemcpyDeviceToHost, st));
      T_TRY(hipMemcpyAsync(lab_host_f.data(), d.lab + (size_t)b * hw, hw * 4, hipMemcpyDeviceToHost, st));
      T_TRY(hipMemcpyAsync(prob_h.data(), d.prob + (size_t)b * d.prob_stride, hw * 4, hipMemcpyDeviceToHost, st));
      T_TRY(hipStreamSynchronize(st));
      for (size_t i = 0; i < hw; ++i) {            // split the signed image into the two label images ctd_db_boxes takes
        const int32_t l = lab_host_f[i];
        lab_host_f[i] = l > 0 ? l : 0;
        lab_host_b[i] = l < 0 ? -l : 0;
      }
      // stats beyond `cap` rows were dropped on the device: recompute them from the labels
      std::vector<int32_t> sf((size_t)nfb[0] * 5), sb((size_t)nfb[1] * 5);
      auto stats_of = [&](const std::vector<int32_t>& lab, std::vector<int32_t>& s, int n) {
        for (int l = 0; l < n; ++l) {
          int32_t* q = s.data() + 5 * (size_t)l;
          q[0] = Wn, q[1] = Hn, q[2] = -1, q[3] = -1, q[4] = 0;
        }
        for (int y = 0; y < Hn; ++y)
          for (int x = 0; x < Wn; ++x) {
            const int l = lab[(size_t)y * Wn + x];
            if (l <= 0 || l > n) continue;
            int32_t* q = s.data() + 5 * (size_t)(l - 1);
            q[0] = std::min(q[0], x), q[1] = std::min(q[1], y), q[2] = std::max(q[2], x), q[3] = std::max(q[3], y), ++q[4];
          }
        for (int l = 0; l < n; ++l) {
          int32_t* q = s.data() + 5 * (size_t)l;
          q[2] -= q[0] - 1, q[3] -= q[1] - 1;
        }
      };
      stats_of(lab_host_f, sf, nfb[0]);
      stats_of(lab_host_b, sb, nfb[1]);
      if (int rc = ctd_db_boxes(prob_h.data(), lab_host_f.data(), sf.data(), nfb[0], lab_host_b.data(), sb.data(), nfb[1], Wn, Hn,
                                maxc, prm->unclip_ratio, boxes.data(), scores.data(), &nbox))
        return ctd_fail_msg(rc, "ctd_db_boxes failed");
    }
    po.db_boxes.assign(boxes.begin(), boxes.begin() + (size_t)nbox * 8);
    po.db_scores.assign(scores.begin(), scores.begin() + nbox);
  }
  return CTD_OK;
}

}  // namespace

extern "C" {

int ctd_tail_create(ctd_tail** out, int32_t device) {
  if (!out) return ctd_fail_msg(CTD_ERR_INVALID, "null out");
  T_TRY(hipSetDevice(device));
  ctd_tail* t = new ctd_tail();
  t->device = device;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically lowest = highest priority
  // ctd_tuning_set("tail_priority"): 1 = the device's default priority (default), 0 = highest, 2 = lowest (measured: -13 %)
  const int prio = g_tail_priority == 2 ? lo : (g_tail_priority == 1 ? (lo + hi) / 2 : hi);
  if (g_tail_cus > 0) {
    hipDeviceProp_t p;
    T_TRY(hipGetDeviceProperties(&p, device));
    const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> mask((size_t)words, 0u);
    for (int c = g_tail_cu_first; c < g_tail_cu_first + g_tail_cus && c < ncu; ++c) mask[c >> 5] |= 1u << (c & 31);
    if (hipExtStreamCreateWithCUMask(&t->st, (uint32_t)words, mask.data()) != hipSuccess) {
      delete t;
      return ctd_fail_msg(CTD_ERR_HIP, "hipExtStreamCreateWithCUMask failed");
    }
  } else if (hipStreamCreateWithPriority(&t->st, hipStreamNonBlocking, prio) != hipSuccess) {
    delete t;
    return ctd_fail_msg(CTD_ERR_HIP, "hipStreamCreateWithPriority failed");
  }
  // what one block may hold of the CU's LDS on this device: the window-local merge kernel's launches stay below it
  int smem = 0;
  if (hipDeviceGetAttribute(&smem, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && smem > 0) t->lds_per_block = smem;
  *out = t;
  return CTD_OK;
}

void ctd_tail_destroy(ctd_tail* t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  if (t->st) {
    (void)hipStreamSynchronize(t->st);
    (void)hipStreamDestroy(t->st);
  }
  DevBuf* dv[] = {&t->d_dets, &t->d_nms_ws, &t->d_lab_f, &t->d_ccl_ws, &t->d_ccl_small, &t->d_dbc_i, &t->d_dbc_d,
                  &t->d_rows, &t->d_pmask, &t->d_refined, &t->d_wins, &t->d_rules, &t->d_bands, &t->d_hist, &t->d_sums,
                  &t->d_canvas, &t->d_clab, &t->d_cstats, &t->d_cnt, &t->d_merged, &t->d_mlab, &t->d_mstats, &t->d_cnt2,
                  &t->d_small, &t->d_crop, &t->d_wins_c, &t->d_bands_c, &t->d_lds};
  for (DevBuf* d : dv) d->release();
  PinBuf* pv[] = {&t->h_dets, &t->h_hdr, &t->h_tab, &t->h_pmask, &t->h_refined, &t->h_hist, &t->h_sums, &t->h_wins,
                  &t->h_rules, &t->h_bands, &t->h_small, &t->h_lab, &t->h_wins_c, &t->h_bands_c, &t->h_lds};
  for (PinBuf* p : pv) p->release();
  delete t;
}

void* ctd_tail_stream(ctd_tail* t) { return t ? (void*)t->st : nullptr; }

// Every failure exit of the three entry points that enqueue work drains the tail's stream first: kernels and copies
// already enqueued still use the caller's tensors and page-locked result arrays, which the caller frees (and its
// caching allocators hand out again) as soon as the error is raised.
static int drained(ctd_tail* t, int rc) {
  if (rc != CTD_OK && t && t->st) (void)hipStreamSynchronize(t->st);
  return rc;
}

static int tail_run_impl(ctd_tail* t, int32_t B, int32_t Hn, int32_t Wn, const float* blks_dev, int32_t rows, int32_t no,
                         const uint8_t* mask_u8_dev, const float* prob_dev, int64_t prob_stride, const uint8_t* bitmap_dev,
                         const ctd_tail_page* pages, const ctd_tail_params* prm, uint8_t* const* mask_out,
                         uint8_t* const* refined_out, void* ready_event) {
  if (!t || !blks_dev || !mask_u8_dev || !prob_dev || !bitmap_dev || !pages || !prm || B < 1 || Hn < 1 || Wn < 1 ||
      rows < 1 || no < 6)
    return ctd_fail_msg(CTD_ERR_INVALID, "ctd_tail_run: bad arguments");
#ifdef CTD_MEASURE_KNOBS
  ctd_tail_params prm_ablate;
  if (g_tail_ablate & 1) {                   // ("tail_ablate" bit 1): the tail without its refine stage
    prm_ablate = *prm;
    prm_ablate.refine = 0;
    prm = &prm_ablate;
  }
#endif
  T_TRY(hipSetDevice(t->device));
  hipStream_t st = t->st;
  const double t0 = now_ms();
  for (double& v : t->ms_stage) v = 0;
  t->n_lds = t->n_canvas = t->n_ovf = 0, t->ms_lds_wait = 0;
  if (ready_event) T_TRY(hipStreamWaitEvent(st, (hipEvent_t)ready_event, 0));
  if (int rc = layout_pages(t, B, pages)) return rc;
  for (int b = 0; b < B; ++b)
    if (pages[b].dw < 0 || pages[b].dh < 0 || pages[b].dw >= Wn || pages[b].dh >= Hn || (prm->refine && !pages[b].img_dev))
      return ctd_fail_msg(CTD_ERR_INVALID, "ctd_tail_run: bad page record");
  const size_t hw = (size_t)Hn * Wn;

  // ================= stage 1: NMS, two labelling passes, contour tables, page masks =================
  GpuChain chain(st, t->device, g_tail_chain >= 1);
  T_TRY(chain.begin());
  GET(t->d_dets, (size_t)B * kMaxDet * 6 * 4 + (size_t)B * 4, float, dets);
  int* counts = (int*)(dets + (size_t)B * kMaxDet * 6);
  GET(t->d_nms_ws, nms_workspace_bytes(B, rows), uint8_t, nms_ws);
  launch_nms(blks_dev, B, rows, no, prm->conf_thresh, prm->nms_thresh, kMaxDet, 30000, 4096.f, dets, counts, nms_ws, st);
  GET(t->h_dets, (size_t)B * kMaxDet * 6 * 4 + (size_t)B * 4, float, hdets);
  // page masks: crop of the letterbox padding, resize to the page (inference.py:164-165)
  GET(t->d_pmask, t->ptotal, uint8_t, pmask);
  GET(t->d_refined, t->ptotal, uint8_t, refined);
  Batch pre(st), post(st);                      // this stage's fills / its copies: one launch each
  pre.fill(refined, 0, t->ptotal);
  const double t0a = now_ms();
  DbStage db;
  if (int rc = db_enqueue(t, db, B, Hn, Wn, prob_dev, prob_stride, bitmap_dev, pre, post)) return rc;
  T_TRY(post.d2h(hdets, dets, (size_t)B * kMaxDet * 6 * 4 + (size_t)B * 4));
  size_t crop_px = 1;                       // pages whose letterbox padded the right side are cropped into a dense
  for (int b = 0; b < B; ++b)               // temporary first (one buffer: the stream runs the pages in order)
    if (pages[b].dw > 0) crop_px = std::max(crop_px, (size_t)(Hn - pages[b].dh) * (Wn - pages[b].dw));
  GET(t->d_crop, crop_px, uint8_t, tmp);
  bool plain = true;                        // every page is the network input itself: one strided copy for the batch
  for (int b = 0; b < B; ++b)
    plain = plain && pages[b].im_h == Hn && pages[b].im_w == Wn && pages[b].dw == 0 && pages[b].dh == 0;
  // Host copy of the page masks (`group_output` reads it, and it IS the `mask` the caller gets unless
  // refine_undetected_mask edits the masks later): straight into the caller's arrays when there are any -- a second
  // 1 MB-per-page memcpy at the end of the call was 3 ms per 32 pages -- else into this object's pinned buffer.
  const bool mask_final = !(prm->refine && prm->keep_undetected_mask);
  bool direct = mask_final && mask_out != nullptr;
  for (int b = 0; b < B && direct; ++b) direct = mask_out[b] != nullptr;
  GET(t->h_pmask, direct ? 256 : t->ptotal, uint8_t, hpmask);
  t->hmask.assign(B, nullptr);
  for (int b = 0; b < B; ++b) t->hmask[b] = direct ? mask_out[b] : hpmask + t->poff[b];
  bool dense = direct;                      // the caller's arrays lie back to back (one pinned allocation per batch)
  for (int b = 1; b < B && dense; ++b) dense = mask_out[b] == mask_out[b - 1] + (size_t)pages[b - 1].im_h * pages[b - 1].im_w;
  const double t0b = now_ms();
  if (plain) {   // device and host copy of the page masks both read the network's u8 mask: same launch, no ordering needed
    const size_t stride = B > 1 ? t->poff[1] - t->poff[0] : hw;
    post.copy2d(pmask, stride, mask_u8_dev, hw, hw, B);
    if (dense) T_TRY(post.d2h_user(mask_out[0], mask_u8_dev, hw * (size_t)B));
    else if (direct)
      for (int b = 0; b < B; ++b) T_TRY(post.d2h_user(mask_out[b], mask_u8_dev + (size_t)b * hw, hw));
    else T_TRY(post.d2h2d(hpmask, stride, mask_u8_dev, hw, hw, B));
  }
  for (int b = 0; b < B && !plain; ++b) {
    const ctd_tail_page& pg = pages[b];
    const int ch = Hn - pg.dh, cw = Wn - pg.dw;
    const uint8_t* src = mask_u8_dev + (size_t)b * hw;
    uint8_t* dst = pmask + t->poff[b];
    if (pg.im_h == ch && pg.im_w == cw) {
      launch_copy2d_u8(src, Wn, dst, cw, ch, cw, st);
    } else if (cw != Wn) {                  // the resize kernel reads a dense source
      launch_copy2d_u8(src, Wn, tmp, cw, ch, cw, st);
      launch_resize_linear_u8(tmp, ch, cw, 1, dst, pg.im_h, pg.im_w, pg.im_h, pg.im_w, st);
    } else {
      launch_resize_linear_u8(src, ch, cw, 1, dst, pg.im_h, pg.im_w, pg.im_h, pg.im_w, st);
    }
  }
  if (!plain) {
    if (!direct) T_TRY(post.d2h(hpmask, pmask, t->ptotal));
    else
      for (int b = 0; b < B; ++b) T_TRY(post.d2h_user(mask_out[b], pmask + t->poff[b], (size_t)pages[b].im_h * pages[b].im_w));
  }
  post.flush();
  T_TRY(hipGetLastError());
  T_TRY(chain.end());
  const double t1 = now_ms();
  t->ms_sub[0] = t0a - t0, t->ms_sub[1] = t0b - t0a, t->ms_sub[2] = t1 - t0b;
  T_TRY(hipStreamSynchronize(st));                                     // sync 1: counts are known
  const double t2 = now_ms();

  // ================= stage 2: contour geometry on the host, grouping =================
  if (int rc = db_collect(t, db, prm)) return rc;                      // sync 2 inside: the sized tables
  const double t3 = now_ms();
  std::vector<WinReq> reqs;
  std::vector<std::vector<int32_t>> blk_xyxy(B);
  // the pages of a work item are independent here too: one thread per page like the contour geometry (a work item's latency, not
  // its CPU time: 1.9 ms per 32 headline pages in a row, 5.9 ms on the dense ones)
  std::atomic<int> gerr{CTD_OK};
  parallel_for(B, t->host_threads, [&](int b) {
    const ctd_tail_page& pg = pages[b];
    PageOut& po = t->out[b];
    const double rx = (double)pg.im_w / (double)(Wn - pg.dw), ry = (double)pg.im_h / (double)(Hn - pg.dh);   // :148
    // ---- postprocess_yolo (:101-114): float32 scale, truncation to int32
    const int nd = std::min(std::max(((const int*)(hdets + (size_t)B * kMaxDet * 6))[b], 0), kMaxDet);
    po.yolo.resize((size_t)nd * 4);
    po.yolo_cls.resize(nd);
    po.yolo_conf.resize(nd);
    const float frx = (float)rx, fry = (float)ry;
    for (int i = 0; i < nd; ++i) {
      const float* d = hdets + ((size_t)b * kMaxDet + i) * 6;
      po.yolo[4 * i] = (int32_t)(d[0] * frx), po.yolo[4 * i + 1] = (int32_t)(d[1] * fry);
      po.yolo[4 * i + 2] = (int32_t)(d[2] * frx), po.yolo[4 * i + 3] = (int32_t)(d[3] * fry);
      po.yolo_cls[i] = (int32_t)d[5];
      po.yolo_conf[i] = d[4];
    }
    // ---- lines = boxes with score > box_thresh, mapped to the page (:159-172)
    std::vector<int32_t> lines;
    const int nbox = (int)po.db_scores.size();
    for (int i = 0; i < nbox; ++i) {
      if (!(po.db_scores[i] > prm->box_thresh)) continue;
      for (int k = 0; k < 4; ++k) {
        lines.push_back((int32_t)((double)po.db_boxes[(size_t)i * 8 + 2 * k] * rx));
        lines.push_back((int32_t)((double)po.db_boxes[(size_t)i * 8 + 2 * k + 1] * ry));
      }
    }
    // ---- group_output (:173)
    const int nl = (int)lines.size() / 8;
    const int bcap = nd + nl, dcap = std::max(1, bcap) * std::max(1, nl);
    po.blks.resize(std::max(bcap, 1));
    po.lines.resize((size_t)std::max(bcap, 1) * 8);
    po.dist.resize((size_t)std::max(dcap, 1) * 3);
    int nb_out = 0, nl_out = 0, nd_out = 0;
    if (int rc = ctd_group_output(po.yolo.data(), po.yolo_cls.data(), nd, lines.data(), nl, pg.im_w, pg.im_h,
                                  t->hmask[b], pg.im_w, po.blks.data(), bcap, po.lines.data(), bcap, po.dist.data(),
                                  dcap, &nb_out, &nl_out, &nd_out)) {
      gerr.store(rc);
      nb_out = nl_out = nd_out = 0;
    }
    po.blks.resize(nb_out);
    po.lines.resize((size_t)nl_out * 8);
    po.dist.resize((size_t)nd_out * 3);
  });
  if (int rc = gerr.load()) return ctd_fail_msg(rc, "ctd_group_output failed");
  for (int b = 0; b < B; ++b) {               // the refine windows in page order, block order (the reference's loop order)
    const ctd_tail_page& pg = pages[b];
    for (const ctd_blk& k : t->out[b].blks) {
      WinReq wq;
      wq.page = b;
      if (block_window(k.xyxy, pg.im_w, pg.im_h, wq)) reqs.push_back(wq);
      blk_xyxy[b].insert(blk_xyxy[b].end(), k.xyxy, k.xyxy + 4);
    }
  }

  // ================= stage 3: mask refinement =================
  const double t4 = now_ms();
  double t5 = t4;
  if (prm->refine) {
    if (int rc = refine_windows(t, reqs, prm->refine_mode)) return rc;
    t5 = now_ms();
    if (prm->keep_undetected_mask) {
      const double k4 = t->ms_stage[4], k5 = t->ms_stage[5], k6 = t->ms_stage[6];
      if (int rc = undetected_pass(t, blk_xyxy, prm->refine_mode)) return rc;
      t->ms_stage[4] = k4, t->ms_stage[5] = k5, t->ms_stage[6] = k6;
    }
  }
  const double t6 = now_ms();
  int rc = CTD_OK;
  if (prm->refine && prm->keep_undetected_mask) {
    rc = download_pages(t, true, mask_out, refined_out);
  } else {   // the mask was not edited: the early download is the result (already in the caller's arrays when `direct`)
    rc = download_pages(t, false, nullptr, prm->refine ? refined_out : nullptr);
    if (!rc && mask_out && !direct)
      for (int b = 0; b < B; ++b)
        if (mask_out[b]) std::memcpy(mask_out[b], t->hmask[b], (size_t)pages[b].im_h * pages[b].im_w);
  }
  const double t7 = now_ms();
  t->ms_stage[0] = t1 - t0, t->ms_stage[1] = t2 - t1, t->ms_stage[2] = t3 - t2, t->ms_stage[3] = t4 - t3;
  t->ms_stage[7] = t6 - t5, t->ms_stage[8] = t7 - t6, t->ms_stage[9] = t7 - t0;
  return rc;
}

int ctd_tail_run(ctd_tail* t, int32_t B, int32_t Hn, int32_t Wn, const float* blks_dev, int32_t rows, int32_t no,
                 const uint8_t* mask_u8_dev, const float* prob_dev, int64_t prob_stride, const uint8_t* bitmap_dev,
                 const ctd_tail_page* pages, const ctd_tail_params* prm, uint8_t* const* mask_out,
                 uint8_t* const* refined_out, void* ready_event) {
  return drained(t, tail_run_impl(t, B, Hn, Wn, blks_dev, rows, no, mask_u8_dev, prob_dev, prob_stride, bitmap_dev, pages, prm,
                                  mask_out, refined_out, ready_event));
}

int ctd_tail_timings(const ctd_tail* t, double* ms16) {   // 16 entries
  if (!t || !ms16) return ctd_fail_msg(CTD_ERR_INVALID, "null argument");
  std::memcpy(ms16, t->ms_stage, sizeof(t->ms_stage));
  ms16[10] = t->ms_db_wait;
  for (int i = 0; i < 5; ++i) ms16[11 + i] = t->ms_sub[i];
  ms16[15] = t->ms_lds_wait;
  return CTD_OK;
}

int ctd_tail_refine_paths(const ctd_tail* t, int32_t* counts3) {
  if (!t || !counts3) return ctd_fail_msg(CTD_ERR_INVALID, "null argument");
  counts3[0] = t->n_lds, counts3[1] = t->n_canvas, counts3[2] = t->n_ovf;
  return CTD_OK;
}

static int tail_db_boxes_impl(ctd_tail* t, int32_t B, int32_t Hn, int32_t Wn, const float* prob_dev, int64_t prob_stride,
                             const uint8_t* bitmap_dev, int32_t max_candidates, double unclip_ratio) {
  if (!t || !prob_dev || !bitmap_dev || B < 1 || Hn < 1 || Wn < 1) return ctd_fail_msg(CTD_ERR_INVALID, "bad arguments");
  T_TRY(hipSetDevice(t->device));
  t->B = B;
  t->out.assign(B, PageOut());
  DbStage db;
  Batch pre(t->st), post(t->st);
  if (int rc = db_enqueue(t, db, B, Hn, Wn, prob_dev, prob_stride, bitmap_dev, pre, post)) return rc;
  post.flush();
  T_TRY(hipGetLastError());
  T_TRY(hipStreamSynchronize(t->st));
  ctd_tail_params prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.max_candidates = max_candidates;
  prm.unclip_ratio = unclip_ratio;
  return db_collect(t, db, &prm);
}

int ctd_tail_db_boxes(ctd_tail* t, int32_t B, int32_t Hn, int32_t Wn, const float* prob_dev, int64_t prob_stride,
                      const uint8_t* bitmap_dev, int32_t max_candidates, double unclip_ratio) {
  return drained(t, tail_db_boxes_impl(t, B, Hn, Wn, prob_dev, prob_stride, bitmap_dev, max_candidates, unclip_ratio));
}

static int tail_refine_impl(ctd_tail* t, int32_t n_pages, const ctd_tail_page* pages, const uint8_t* const* masks_host,
                            const int32_t* blk_xyxy, const int32_t* blk_counts, int32_t refine_mode,
                            int32_t keep_undetected_mask, uint8_t* const* mask_out, uint8_t* const* refined_out) {
  if (!t || n_pages < 1 || !pages || !masks_host || !blk_counts || !refined_out)
    return ctd_fail_msg(CTD_ERR_INVALID, "ctd_tail_refine: bad arguments");
  T_TRY(hipSetDevice(t->device));
  t->n_lds = t->n_canvas = t->n_ovf = 0, t->ms_lds_wait = 0;
  hipStream_t st = t->st;
  if (int rc = layout_pages(t, n_pages, pages)) return rc;
  GET(t->d_pmask, t->ptotal, uint8_t, pmask);
  GET(t->d_refined, t->ptotal, uint8_t, refined);
  GET(t->h_pmask, t->ptotal, uint8_t, hpmask);
  T_TRY(hipMemsetAsync(refined, 0, t->ptotal, st));
  std::memset(hpmask, 0, t->ptotal);
  std::vector<WinReq> reqs;
  std::vector<std::vector<int32_t>> bxy(n_pages);
  const int32_t* q = blk_xyxy;
  for (int b = 0; b < n_pages; ++b) {
    if (!pages[b].img_dev || !masks_host[b]) return ctd_fail_msg(CTD_ERR_INVALID, "ctd_tail_refine: null page");
    std::memcpy(hpmask + t->poff[b], masks_host[b], (size_t)pages[b].im_h * pages[b].im_w);
    for (int k = 0; k < blk_counts[b]; ++k, q += 4) {
      WinReq wq;
      wq.page = b;
      if (block_window(q, pages[b].im_w, pages[b].im_h, wq)) reqs.push_back(wq);
      bxy[b].insert(bxy[b].end(), q, q + 4);
    }
  }
  T_TRY(hipMemcpyAsync(pmask, hpmask, t->ptotal, hipMemcpyHostToDevice, st));
  if (int rc = refine_windows(t, reqs, refine_mode)) return rc;
  if (keep_undetected_mask)
    if (int rc = undetected_pass(t, bxy, refine_mode)) return rc;
  return download_pages(t, keep_undetected_mask != 0, mask_out, refined_out);
}

int ctd_tail_refine(ctd_tail* t, int32_t n_pages, const ctd_tail_page* pages, const uint8_t* const* masks_host,
                    const int32_t* blk_xyxy, const int32_t* blk_counts, int32_t refine_mode, int32_t keep_undetected_mask,
                    uint8_t* const* mask_out, uint8_t* const* refined_out) {
  return drained(t, tail_refine_impl(t, n_pages, pages, masks_host, blk_xyxy, blk_counts, refine_mode, keep_undetected_mask,
                                     mask_out, refined_out));
}

int ctd_tail_page_counts(const ctd_tail* t, int32_t page, int32_t* n_blocks, int32_t* n_lines, int32_t* n_dist,
                         int32_t* n_db_boxes, int32_t* n_yolo) {
  if (!t || page < 0 || page >= (int)t->out.size()) return ctd_fail_msg(CTD_ERR_INVALID, "bad page index");
  const PageOut& po = t->out[page];
  if (n_blocks) *n_blocks = (int32_t)po.blks.size();
  if (n_lines) *n_lines = (int32_t)(po.lines.size() / 8);
  if (n_dist) *n_dist = (int32_t)(po.dist.size() / 3);
  if (n_db_boxes) *n_db_boxes = (int32_t)po.db_scores.size();
  if (n_yolo) *n_yolo = (int32_t)po.yolo_cls.size();
  return CTD_OK;
}

int ctd_tail_batch_counts(const ctd_tail* t, int32_t* counts) {
  if (!t || !counts) return ctd_fail_msg(CTD_ERR_INVALID, "ctd_tail_batch_counts: null argument");
  for (size_t b = 0; b < t->out.size(); ++b) {
    const PageOut& po = t->out[b];
    int32_t* c = counts + 5 * b;
    c[0] = (int32_t)po.blks.size(), c[1] = (int32_t)(po.lines.size() / 8), c[2] = (int32_t)(po.dist.size() / 3);
    c[3] = (int32_t)po.db_scores.size(), c[4] = (int32_t)po.yolo_cls.size();
  }
  return CTD_OK;
}

int ctd_tail_batch_fetch(const ctd_tail* t, ctd_blk* blocks, int32_t* lines, double* dist) {
  if (!t) return ctd_fail_msg(CTD_ERR_INVALID, "ctd_tail_batch_fetch: null tail");
  for (const PageOut& po : t->out) {
    if (blocks && !po.blks.empty()) std::memcpy(blocks, po.blks.data(), po.blks.size() * sizeof(ctd_blk));
    if (blocks) blocks += po.blks.size();
    if (lines && !po.lines.empty()) std::memcpy(lines, po.lines.data(), po.lines.size() * 4);
    if (lines) lines += po.lines.size();
    if (dist && !po.dist.empty()) std::memcpy(dist, po.dist.data(), po.dist.size() * 8);
    if (dist) dist += po.dist.size();
  }
  return CTD_OK;
}

int ctd_tail_pack_records(const ctd_tail* t, int32_t cap_blk, int32_t cap_line, double* out) {
  if (!t || !out || cap_blk < 0 || cap_line < 0) return ctd_fail_msg(CTD_ERR_INVALID, "ctd_tail_pack_records: bad arguments");
  const size_t R = 4 + (size_t)cap_blk * 12 + (size_t)cap_line * 8;
  std::memset(out, 0, t->out.size() * R * sizeof(double));
  for (size_t b = 0; b < t->out.size(); ++b) {
    const PageOut& po = t->out[b];
    double* r = out + b * R;
    double* rl = r + 4 + (size_t)cap_blk * 12;
    long long nl = 0;
    for (size_t i = 0; i < po.blks.size(); ++i) {
      const ctd_blk& k = po.blks[i];
      if ((int)i < cap_blk) {
        double* q = r + 4 + i * 12;
        q[0] = k.xyxy[0], q[1] = k.xyxy[1], q[2] = k.xyxy[2], q[3] = k.xyxy[3];
        q[4] = k.language, q[5] = k.vertical ? 1 : 0, q[6] = k.angle;
        q[7] = k.font_is_float ? k.font_size : (double)(long long)k.font_size;
        q[8] = k.n_lines, q[9] = k.norm, q[10] = k.vec[0], q[11] = k.vec[1];
      }
      for (int j = 0; j < k.n_lines; ++j, ++nl) {
        if (nl >= cap_line) continue;
        const int32_t* src = po.lines.data() + ((size_t)k.line_off + j) * 8;
        for (int e = 0; e < 8; ++e) rl[nl * 8 + e] = src[e];
      }
    }
    r[0] = (double)po.blks.size(), r[1] = (double)nl, r[2] = cap_blk, r[3] = cap_line;
  }
  return CTD_OK;
}

int ctd_tail_set_threads(ctd_tail* t, int32_t n) {
  if (!t || n < 1) return ctd_fail_msg(CTD_ERR_INVALID, "ctd_tail_set_threads: n >= 1");
  t->host_threads = n;
  return CTD_OK;
}

int ctd_tail_page_fetch(const ctd_tail* t, int32_t page, ctd_blk* blocks, int32_t* lines, double* dist, int16_t* db_boxes,
                        float* db_scores, int32_t* yolo_xyxy, int32_t* yolo_cls, float* yolo_conf) {
  if (!t || page < 0 || page >= (int)t->out.size()) return ctd_fail_msg(CTD_ERR_INVALID, "bad page index");
  const PageOut& po = t->out[page];
  if (blocks && !po.blks.empty()) std::memcpy(blocks, po.blks.data(), po.blks.size() * sizeof(ctd_blk));
  if (lines && !po.lines.empty()) std::memcpy(lines, po.lines.data(), po.lines.size() * 4);
  if (dist && !po.dist.empty()) std::memcpy(dist, po.dist.data(), po.dist.size() * 8);
  if (db_boxes && !po.db_boxes.empty()) std::memcpy(db_boxes, po.db_boxes.data(), po.db_boxes.size() * 2);
  if (db_scores && !po.db_scores.empty()) std::memcpy(db_scores, po.db_scores.data(), po.db_scores.size() * 4);
  if (yolo_xyxy && !po.yolo.empty()) std::memcpy(yolo_xyxy, po.yolo.data(), po.yolo.size() * 4);
  if (yolo_cls && !po.yolo_cls.empty()) std::memcpy(yolo_cls, po.yolo_cls.data(), po.yolo_cls.size() * 4);
  if (yolo_conf && !po.yolo_conf.empty()) std::memcpy(yolo_conf, po.yolo_conf.data(), po.yolo_conf.size() * 4);
  return CTD_OK;
}

}  // extern "C"
