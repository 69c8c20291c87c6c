// detops_common.h — shared host/device helpers for libdetops_gfx950 (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "detops.h"

#define DETOPS_API extern "C" __attribute__((visibility("default")))

static inline hipStream_t as_stream(detops_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Launch epilogue: surfaces launch-configuration errors as the function's return code.
static inline int launch_status() { return static_cast<int>(hipGetLastError()); }

#define DETOPS_HIP_TRY(expr)                       \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

constexpr int kWave = 64;        // CDNA4 wavefront
constexpr int kNumCU = 256;      // MI355X
constexpr int kNumXCD = 8;

// Workgroup index swizzle: consecutive logical ids land on the same XCD (the dispatcher places
// block b on XCD b % 8 — observed, used for L2 locality only, never for correctness).
__device__ __forceinline__ int xcd_swizzle(int bid, int nblocks) {
  const int per = nblocks / kNumXCD;
  if (per * kNumXCD != nblocks) return bid;  // only swizzle evenly divisible grids
  return (bid % kNumXCD) * per + bid / kNumXCD;
}
