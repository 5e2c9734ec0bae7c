// What numpy itself computes, where the reference's result depends on it (round 6, DESIGN 5 "ties"): numpy's float64 `arccos` on an
// AVX-512 host is Intel SVML's `__svml_acos8_ha` (last bit differs from glibc's `acos` on ~9 % of arguments), and its default
// `argsort` of 64-bit keys is x86-simd-sort on AVX-512 / AVX2 hosts (the order of EQUAL keys beyond 16 elements is its own).
// Both decide the reference's results on ties -- lines of one text row, histogram bins of equal count -- so the native host code
// calls the very functions numpy calls: they are exported by numpy's `_multiarray_umath` module, which is looked up among the
// process's loaded objects (once).  Where numpy itself would not take those paths -- no numpy in the process, no AVX-512 / AVX2, a
// build without them -- the callers fall back to libm / a stable sort.
#pragma once
#include <dlfcn.h>
#include <immintrin.h>
#include <link.h>

#include <cmath>
#include <cstring>

namespace npd {

inline void* find_symbol(const char* sym) {
  struct Ctx { const char* sym; void* fn; } ctx{sym, nullptr};
  dl_iterate_phdr(
      [](struct dl_phdr_info* info, size_t, void* data) -> int {
        Ctx* c = (Ctx*)data;
        if (!info->dlpi_name || !std::strstr(info->dlpi_name, "_multiarray_umath")) return 0;
        if (void* h = dlopen(info->dlpi_name, RTLD_LAZY | RTLD_NOLOAD)) {
          c->fn = dlsym(h, c->sym);
          dlclose(h);                                      // drops the count RTLD_NOLOAD added; the module stays loaded
        }
        return c->fn != nullptr;
      },
      &ctx);
  return ctx.fn;
}
// numpy's AVX512_SKX dispatch target = F + CD + BW + DQ + VL
inline bool cpu_avx512_skx() {
  return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512cd") && __builtin_cpu_supports("avx512bw") &&
         __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
}
__attribute__((target("avx512f"))) inline double svml_call1(void* fn, double x) {
  typedef __m512d (*vfn)(__m512d);
  return _mm512_cvtsd_f64(((vfn)fn)(_mm512_set1_pd(x)));
}
// np.arccos of one float64
inline double arccos(double c) {
  static void* const fn = cpu_avx512_skx() ? find_symbol("__svml_acos8_ha") : nullptr;   // once per process, thread-safe
  return fn ? svml_call1(fn, c) : std::acos(c);
}
// np.argsort(keys) with numpy's default kind for float64 / int64 keys: `idx` must hold 0 .. n-1; false = not available (the
// caller sorts stably)
typedef void (*argsort_f64_fn)(double*, long*, long);
typedef void (*argsort_i64_fn)(long*, long*, long);
inline bool argsort_f64(double* keys, long* idx, long n) {
  static const argsort_f64_fn fn = (argsort_f64_fn)(cpu_avx512_skx() ? find_symbol("_ZN2np10qsort_simd19ArgQSort_AVX512_SKXIdEEvPT_Pll")
                                   : __builtin_cpu_supports("avx2") ? find_symbol("_ZN2np10qsort_simd13ArgQSort_AVX2IdEEvPT_Pll") : nullptr);
  if (!fn) return false;
  fn(keys, idx, n);
  return true;
}
inline bool argsort_i64(long* keys, long* idx, long n) {
  static const argsort_i64_fn fn = (argsort_i64_fn)(cpu_avx512_skx() ? find_symbol("_ZN2np10qsort_simd19ArgQSort_AVX512_SKXIlEEvPT_Pll")
                                   : __builtin_cpu_supports("avx2") ? find_symbol("_ZN2np10qsort_simd13ArgQSort_AVX2IlEEvPT_Pll") : nullptr);
  if (!fn) return false;
  fn(keys, idx, n);
  return true;
}

}  // namespace npd
