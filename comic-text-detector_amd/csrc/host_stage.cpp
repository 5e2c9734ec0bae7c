// Host-side staging of a batch of pages into ONE pinned buffer (the caller then issues one async H2D copy).
// The reference's callers hand `TextDetector.__call__` numpy images (reference inference.py:141); copying 32
// pages of 3 MB from Python holds the interpreter lock (numpy slice assignment) or fights over an intra-op
// thread pool (torch copy_) -- both measured on the GPU box -- so the gather is a plain C call: ctypes releases
// the lock for its duration, and a few threads split the pages.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/ctd_hip.h"

extern "C" int ctd_host_gather(void* dst, const void* const* srcs, const size_t* sizes, int32_t n, int32_t threads) {
  if (n < 0 || (n > 0 && (!dst || !srcs || !sizes))) return CTD_ERR_INVALID;
  std::vector<size_t> off((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) {
    if (sizes[i] && !srcs[i]) return CTD_ERR_INVALID;
    off[i + 1] = off[i] + sizes[i];
  }
  const int nt = std::max(1, std::min(threads, n));
  auto work = [&](int t) {
    for (int i = t; i < n; i += nt)
      if (sizes[i]) std::memcpy((char*)dst + off[i], srcs[i], sizes[i]);
  };
  if (nt == 1) {
    work(0);
    return CTD_OK;
  }
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return CTD_OK;
}
