// detops_devlib.h — the two device-library primitives the kernels' host code uses (hipCUB radix sort of 64-bit keys,
// hipCUB exclusive prefix sum).  hipCUB is a heavy include, so only the two translation units that sort / scan (nms.hip,
// deform_conv.hip) pull this header in.  Host emulation (tests/emu): the same four functions come from
// tests/emu/detops_emu_devlib.h.
#pragma once
#include "detops_common.h"

#ifdef DETOPS_CPU_EMU
#include "detops_emu_devlib.h"
#else
#include <hipcub/hipcub.hpp>

// bytes of scratch detops_sort_u64 needs for n keys (0 when the query itself fails, e.g. no device visible)
static inline size_t detops_sort_u64_bytes(int n) {
  size_t bytes = 0;
  const hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, static_cast<unsigned long long*>(nullptr),
                                                         static_cast<unsigned long long*>(nullptr), n);
  return e == hipSuccess ? bytes : 0;
}

// ascending sort of n 64-bit keys, in -> out, on `st`
static inline int detops_sort_u64(void* scratch, size_t scratch_bytes, const unsigned long long* in,
                                  unsigned long long* out, int n, hipStream_t st) {
  return static_cast<int>(hipcub::DeviceRadixSort::SortKeys(scratch, scratch_bytes, in, out, n, 0, 64, st));
}

// bytes of scratch detops_exclusive_sum_i32 needs for n items; false when the query fails
static inline bool detops_exclusive_sum_i32_bytes(int n, size_t* bytes) {
  return hipcub::DeviceScan::ExclusiveSum(nullptr, *bytes, static_cast<int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                                          n) == hipSuccess;
}

// out[i] = in[0] + ... + in[i - 1] for i in [0, n), on `st`
static inline int detops_exclusive_sum_i32(void* scratch, size_t scratch_bytes, const int32_t* in, int32_t* out, int n,
                                           hipStream_t st) {
  return static_cast<int>(hipcub::DeviceScan::ExclusiveSum(scratch, scratch_bytes, in, out, n, st));
}
#endif   // DETOPS_CPU_EMU
