// detops_devlib.h — the two device-library primitives the kernels' host code uses (hipCUB radix sort of 64-bit keys,
// hipCUB exclusive prefix sum), each with its stand-in for the host emulation (tests/emu).  Together with
// detops_common.h this is the ONLY place that knows about DETOPS_CPU_EMU; hipCUB is a heavy include, so only the two
// translation units that sort / scan (nms.hip, deform_conv.hip) pull this header in.
#pragma once
#include "detops_common.h"

#ifndef DETOPS_CPU_EMU
#include <hipcub/hipcub.hpp>
#else
#include <algorithm>
#endif

// bytes of scratch detops_sort_u64 needs for n keys (0 when the query itself fails, e.g. no device visible)
static inline size_t detops_sort_u64_bytes(int n) {
#ifndef DETOPS_CPU_EMU
  size_t bytes = 0;
  const hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, static_cast<unsigned long long*>(nullptr),
                                                         static_cast<unsigned long long*>(nullptr), n);
  return e == hipSuccess ? bytes : 0;
#else
  (void)n;
  return 256;
#endif
}

// ascending sort of n 64-bit keys, in -> out, on `st`
static inline int detops_sort_u64(void* scratch, size_t scratch_bytes, const unsigned long long* in,
                                  unsigned long long* out, int n, hipStream_t st) {
#ifndef DETOPS_CPU_EMU
  return static_cast<int>(hipcub::DeviceRadixSort::SortKeys(scratch, scratch_bytes, in, out, n, 0, 64, st));
#else
  (void)scratch; (void)scratch_bytes; (void)st;
  std::copy(in, in + n, out);
  std::sort(out, out + n);
  return 0;
#endif
}

// bytes of scratch detops_exclusive_sum_i32 needs for n items; false when the query fails
static inline bool detops_exclusive_sum_i32_bytes(int n, size_t* bytes) {
#ifndef DETOPS_CPU_EMU
  return hipcub::DeviceScan::ExclusiveSum(nullptr, *bytes, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                                          n) == hipSuccess;
#else
  (void)n;
  *bytes = 0;
  return true;
#endif
}

// out[i] = in[0] + ... + in[i - 1] for i in [0, n), on `st`
static inline int detops_exclusive_sum_i32(void* scratch, size_t scratch_bytes, const int32_t* in, int32_t* out, int n,
                                           hipStream_t st) {
#ifndef DETOPS_CPU_EMU
  return static_cast<int>(hipcub::DeviceScan::ExclusiveSum(scratch, scratch_bytes, in, out, n, st));
#else
  (void)scratch; (void)scratch_bytes; (void)st;
  int32_t run = 0;
  for (int i = 0; i < n; ++i) { const int32_t c = in[i]; out[i] = run; run += c; }
  return 0;
#endif
}
