// detops_emu_devlib.h — host stand-ins for maskrcnn-benchmark_amd/csrc/detops_devlib.h (radix sort / exclusive scan).
// Test infrastructure: only tests/emu builds with DETOPS_CPU_EMU.
#pragma once
#include <algorithm>

static inline size_t detops_sort_u64_bytes(int n) { (void)n; return 256; }

static inline int detops_sort_u64(void* scratch, size_t scratch_bytes, const unsigned long long* in,
                                  unsigned long long* out, int n, hipStream_t st) {
  (void)scratch; (void)scratch_bytes; (void)st;
  std::copy(in, in + n, out);
  std::sort(out, out + n);
  return 0;
}

static inline bool detops_exclusive_sum_i32_bytes(int n, size_t* bytes) { (void)n; *bytes = 0; return true; }

static inline int detops_exclusive_sum_i32(void* scratch, size_t scratch_bytes, const int32_t* in, int32_t* out, int n,
                                           hipStream_t st) {
  (void)scratch; (void)scratch_bytes; (void)st;
  int32_t run = 0;
  for (int i = 0; i < n; ++i) { const int32_t c = in[i]; out[i] = run; run += c; }
  return 0;
}
