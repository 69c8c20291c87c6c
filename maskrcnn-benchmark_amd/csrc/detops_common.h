// detops_common.h — shared host/device helpers for libdetops_gfx950 (gfx950 / CDNA4 only).
#pragma once
#ifdef DETOPS_CPU_EMU  // tests/emu: the kernels compiled as host C++ (logic checks without a GPU)
#include "hip_cpu_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
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

// Dynamically sized LDS of a kernel, as `T name[]` (16-byte aligned).
#ifdef DETOPS_CPU_EMU
#define DETOPS_DYNAMIC_LDS(T, name) T* name = reinterpret_cast<T*>(emu::dynamic_lds())
#else
#define DETOPS_DYNAMIC_LDS(T, name) extern __shared__ __align__(16) T name[]
#endif

// Work counters of the host emulation (tests/emu): compiled out of the device build.
#ifdef DETOPS_CPU_EMU
#define DETOPS_STAT(name, n) emu::stat(name, n)
#else
#define DETOPS_STAT(name, n) ((void)0)
#endif

// Resident workgroups per CU of a kernel at a block size / dynamic LDS size (sizes persistent grids; the
// value is advisory and only affects speed).  Host emulation: a fixed 4.
#ifdef DETOPS_CPU_EMU
#define DETOPS_OCCUPANCY(out, kernel, block, lds) ((out) = 4)
#else
#define DETOPS_OCCUPANCY(out, kernel, block, lds)                                                         \
  do {                                                                                                    \
    int _n = 0;                                                                                           \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&_n, reinterpret_cast<const void*>(kernel), (block), (lds)) == hipSuccess && _n > 0) \
      (out) = _n;                                                                                         \
  } while (0)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() lowers to `s_waitcnt vmcnt(0) lgkmcnt(0);
// s_barrier`: it drains every outstanding GLOBAL load and store of the wave first, which serialises a
// software pipeline (prefetch loads in flight across the barrier, fire-and-forget result stores).  Use only
// where no thread reads global memory another thread of the workgroup wrote before the barrier.
#ifdef DETOPS_CPU_EMU
#define DETOPS_LDS_BARRIER() __syncthreads()
#else
#define DETOPS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// Scheduling pin: the four values must be in registers here, so every load that produces them has been ISSUED
// before this point (hipcc's scheduler otherwise serialises LDS reads one `s_waitcnt lgkmcnt(0)` at a time to
// save registers).  No instruction is emitted.
#ifdef DETOPS_CPU_EMU
#define DETOPS_PIN4(a, b, c, d) ((void)0)
#else
#define DETOPS_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#endif

// Ordering point for LDS data handed from some lanes of a wave to other lanes of the SAME wave (no other wave
// touches the region): the hardware executes a wave's LDS operations in order, so only the compiler must be kept
// from moving the reads above the writes.  The emulation runs lanes as fibers and needs a real rendezvous.
#ifdef DETOPS_CPU_EMU
#define DETOPS_WAVE_SYNC() ((void)__ballot(1))
#else
#define DETOPS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

constexpr int kWave = 64;        // CDNA4 wavefront
constexpr int kNumCU = 256;      // MI355X
constexpr int kNumXCD = 8;

// XCD-contiguous remap of a linear workgroup id (any grid size): the dispatcher deals consecutive
// workgroups round-robin over the 8 XCDs (observed; speed only, never correctness), so id b runs
// on XCD b % 8.  The remap gives every XCD one CONTIGUOUS range of logical ids — neighbouring work
// items then share that XCD's L2.  Bijective on [0, n).
__device__ __forceinline__ int64_t xcd_contiguous(int64_t bid, int64_t n) {
  const int64_t q = n / kNumXCD, r = n % kNumXCD;
  const int64_t x = bid % kNumXCD, i = bid / kNumXCD;
  return x * q + (x < r ? x : r) + i;
}

// Workgroup index swizzle: consecutive logical ids land on the same XCD (the dispatcher places
// block b on XCD b % 8 — observed, used for L2 locality only, never for correctness).
__device__ __forceinline__ int xcd_swizzle(int bid, int nblocks) {
  const int per = nblocks / kNumXCD;
  if (per * kNumXCD != nblocks) return bid;  // only swizzle evenly divisible grids
  return (bid % kNumXCD) * per + bid / kNumXCD;
}
