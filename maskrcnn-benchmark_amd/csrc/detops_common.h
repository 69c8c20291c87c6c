// detops_common.h — shared host/device helpers for libdetops_gfx950 (gfx950 / CDNA4 only).
#pragma once
#include <stdint.h>

typedef unsigned long long detops_u64;
constexpr int kSpinBudget = 1 << 22;   // polls before a wait on another workgroup's flag gives up

// ---- the device primitives.  This is the ONE place in the library sources that knows about the host emulation
// (tests/emu compiles the kernels as host C++ for logic checks without a GPU): there, every name defined between
// here and the #endif comes from tests/emu/detops_emu_shims.h instead.
#ifdef DETOPS_CPU_EMU
#include "detops_emu_shims.h"
#else
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

// Dynamically sized LDS of a kernel, as `T name[]` (16-byte aligned).
#define DETOPS_DYNAMIC_LDS(T, name) extern __shared__ __align__(16) T name[]

// Work counters of the host emulation (tests/emu): compiled out of the device build.
#define DETOPS_STAT(name, n) ((void)0)

// Resident workgroups per CU of a kernel at a block size / dynamic LDS size (sizes persistent grids; the
// value is advisory and only affects speed).  Host emulation: a fixed 4.
#define DETOPS_OCCUPANCY(out, kernel, block, lds)                                                         \
  do {                                                                                                    \
    int _n = 0;                                                                                           \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&_n, reinterpret_cast<const void*>(kernel), (block), (lds)) == hipSuccess && _n > 0) \
      (out) = _n;                                                                                         \
  } while (0)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() lowers to `s_waitcnt vmcnt(0) lgkmcnt(0);
// s_barrier`: it drains every outstanding GLOBAL load and store of the wave first, which serialises a
// software pipeline (prefetch loads in flight across the barrier, fire-and-forget result stores).  Use only
// where no thread reads global memory another thread of the workgroup wrote before the barrier.
#define DETOPS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Scheduling pin: the four values must be in registers here, so every load that produces them has been ISSUED
// before this point (hipcc's scheduler otherwise serialises LDS reads one `s_waitcnt lgkmcnt(0)` at a time to
// save registers).  No instruction is emitted.
#define DETOPS_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define DETOPS_PIN6(a, b, c, d, e, f) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))
// the same for three values, one of which only SOME paths below use: the compiler can neither sink that load under the
// branch nor wait for the three one by one — the requests go out together, one round trip
#define DETOPS_KEEP_TOGETHER3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))

// Ordering point for LDS data handed from some lanes of a wave to other lanes of the SAME wave (no other wave
// touches the region): the hardware executes a wave's LDS operations in order, so only the compiler must be kept
// from moving the reads above the writes.  The emulation runs lanes as fibers and needs a real rendezvous.
#define DETOPS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---- LDS-DMA (gfx950 global_load_lds_dwordx4): 16 bytes per lane, global (per-lane, dword-aligned address) ->
// LDS (wave-uniform base + lane * 16).
//   glds16        compiler-visible builtin: hipcc counts it and waits vmcnt(0) before any LDS read that may alias
//                 (right for a one-batch-ahead double buffer: roi_align_fwd.hip);
//   glds16_async  hidden in inline asm (cdna_hip_programming.md section 5.7): hipcc emits NO wait for it — the kernel
//                 counts completion itself with DETOPS_VMCNT_WAIT(n) (at most n of this wave's vector-memory
//                 operations still outstanding; they retire in issue order) and orders other waves' reads with a
//                 barrier.  This is what lets several stagings stay in flight across barriers (a ring of LDS slots).
//                 `active` lanes transfer; the instruction is issued by the wave either way (call it in wave-uniform
//                 control flow so that every wave's count is known).
// Host emulation: glds16 copies at once; glds16_async QUEUES the copy and DETOPS_VMCNT_WAIT applies all but the
// newest n instructions' copies — a read that is not covered by a wait sees stale LDS, as on the device.
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds16_async(bool active, const float* g, float* lds_wave_base) {
  typedef __attribute__((address_space(3))) float* lds_fptr_t;
  const uint32_t dst = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_fptr_t)lds_wave_base)));
  unsigned keep;
  if (active)   // M0 is compiler-reserved: saved, written and restored inside the one statement that reads it
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
// scalar base (wave-uniform 64-bit pointer) + per-lane unsigned 32-bit byte offset: the address arithmetic of a
// homogeneous instruction stays on the scalar unit
__device__ __forceinline__ void glds16_async_so(bool active, const float* sbase, unsigned voff_bytes, float* lds_wave_base) {
  typedef __attribute__((address_space(3))) float* lds_fptr_t;
  const uint32_t dst = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_fptr_t)lds_wave_base)));
  unsigned keep;
  if (active)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(dst) : "memory");
}
#define DETOPS_VMCNT_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

// ---- write-through (sc1) 16-byte global store and its matching load: partial results handed to ANOTHER workgroup
// inside one launch (cdna_hip_programming.md Guideline 16, "sc1 slab stores"): the store bypasses the XCD's
// non-coherent L2, DETOPS_VMCNT_WAIT(0) + a barrier + a relaxed agent-scope atomic then publishes it.
__device__ __forceinline__ void store_f4_wt(float* p, float4 v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
}

// Acquire at agent scope (buffer_inv sc1: drops this CU's L1 lines) — executed by ONE lane of the workgroup that
// is about to read another workgroup's published partial results, followed by a barrier.
#define DETOPS_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")

// ---- flags between workgroups of ONE launch (producer has a LOWER blockIdx than every consumer: workgroups are
// dispatched in index order, so a waiting workgroup only ever waits for resident or finished ones).  Producer:
// data stores -> detops_release_agent() by every storing wave -> (barrier) -> flag_store / flag_add by one lane.
// Consumer: spin on flag_load (acquire, agent scope: other XCDs' L2 lines are invalidated) -> plain loads.
// Spins are bounded (a wedged producer must not hang the device); the emulation runs workgroups one after another,
// so there a flag that is not already set is a design error and aborts.
__device__ __forceinline__ void detops_release_agent() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ void flag_store(detops_u64* p, detops_u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// for data already in memory (write-through stores + DETOPS_VMCNT_WAIT(0) + barrier): no release fence, i.e. no
// write-back of the XCD's whole L2 (buffer_wbl2) in front of the flag
__device__ __forceinline__ void flag_store_relaxed(detops_u64* p, detops_u64 v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ detops_u64 flag_load(const detops_u64* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int flag_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_add(int* p, int v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// polling form: relaxed (an sc1 load, no cache invalidate per poll); DETOPS_ACQUIRE_AGENT() once after the wait
__device__ __forceinline__ detops_u64 flag_peek(const detops_u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int flag_peek(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// write-through 8-byte store (see store_f4_wt): followed by DETOPS_VMCNT_WAIT(0) and flag_add it publishes a result
// WITHOUT the L2 write-back a release fence costs — the form for thousands of small producers
__device__ __forceinline__ void store_u64_wt(detops_u64* p, detops_u64 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_u32_wt(void* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
// load that observes another workgroup's write-through stores WITHOUT an acquire fence (buffer_inv sc1 drops the
// whole XCD's non-coherent cache lines: ruinous when thousands of consumer waves each execute one)
__device__ __forceinline__ float load_f32_coherent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool spin_again(int& budget) {   // false: give up (seconds of polling)
  __builtin_amdgcn_s_sleep(64);
  return --budget > 0;
}

// ---- relaxed agent-scope fetch-add (a slot request whose result nothing else is ordered against), the packed 16-bit
// hardware atomics (two adjacent elements, 4-byte aligned: global_atomic_pk_add_f16 / _bf16 instead of two
// compare-and-swap loops on the containing dword) and the constant 100 MHz wall clock (diagnostic timelines)
__device__ __forceinline__ int detops_fetch_add_relaxed(int32_t* p, int v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void detops_atomic_add2(__half* p, float v0, float v1) {
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  const h2v v = {static_cast<_Float16>(v0), static_cast<_Float16>(v1)};
  __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2v*)(p), v);
}
__device__ __forceinline__ void detops_atomic_add2(__hip_bfloat16* p, float v0, float v1) {
  typedef __bf16 b2v __attribute__((ext_vector_type(2)));
  const b2v v = {static_cast<__bf16>(v0), static_cast<__bf16>(v1)};
  __builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) b2v*)(p), v);
}
__device__ __forceinline__ long long detops_wall_clock() { return static_cast<long long>(wall_clock64()); }

// ---- hardware transcendental / matrix instructions and their host-emulation stand-ins
__device__ __forceinline__ float detops_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float detops_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }   // v_log_f32 is log2
#define DETOPS_MFMA_32x32x16_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define DETOPS_MFMA_32x32x16_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// Workgroups of `kernel` (block size, dynamic LDS) the current device holds at once; -1 when unknown and in the host
// emulation (which runs workgroups one after another).
template <typename K>
static inline int detops_resident_workgroups(K kernel, int block, size_t lds) {
  int dev = 0, cus = 0, per_cu = 1;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    return -1;
  DETOPS_OCCUPANCY(per_cu, kernel, block, lds);
  return cus * per_cu;
}
#endif   // DETOPS_CPU_EMU

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



// ---- tuning / test switches (read once at library load from DETOPS_TUNING="key=value,...", or set through
// detops_tuning_set(); never read from the environment on the launch path)
struct DetopsTuning {
  int roi_bwd_impl;        // 0 auto | 1 ring (needs a workspace) | 2 scan | 3 atomic scatter | 4 acc (small single map, needs a workspace)
  int roi_bwd_seg;         // ring: hits per segment before a tile's hit list is split over workgroups (0 = default)
  int roi_bwd_split;       // ring: 0 floor(c / seg) segments | 1 ceil(c / seg) segments of at most seg hits
  int roi_bwd_maxseg;      // ring: segments per tile at most (0 = default 8)
  int roi_bwd_extras;      // ring: capacity of the extra-segment table (0 = default 128)
  int roi_bwd_ct;          // ring: channels per unit of the 7x7 kernel, 16 | 32 (0 = default = 32)
  int roi_bwd_ring;        // ring: LDS slots (hits in flight) of the 7x7 kernel, 2 | 3 | 4 (0 = default)
  int roi_bwd_groups;      // scan: ROI-list split over blockIdx.y (0 = auto)
  int roi_bwd_scan_ct;     // scan: channels per workgroup, 4 | 16 (0 = auto)
  int roi_bwd_debug;       // ablation bits (diagnosis only)
  int nms_fused;           // 0 / 1 single launch for n <= 4096 | 2 three launches (sort, mask, scan) | 3 single launch, scans dispatched last
  int nms_fault;           // tests: 1 = the fused launch's sort workgroups publish a wrong token (every consumer wait times out)
  int nms_spin_budget;     // polls before a wait of the fused launch gives up (0 = default: seconds)
  int nms_debug;           // diagnosis only: 1 = the fused launch's tile waves skip the IoU loop (timing ablation: results are WRONG), 4 = record the wall-clock timeline of segment 0 (detops_debug_nms_timeline)
  int nms_no_presorted;    // A/B: 1 = the fused launch's sorts always run the network (0 = default: input that is already in score order skips it)
  int nms_no_repair;       // tests / A-B: 1 = do not launch nms_repair_kernel behind the fused launch (failed segments stay at num_keep = -1)
  int roi_fwd_impl;        // 0 auto | 1 generic gather kernel
  int roi_fwd_order;       // 0 auto | 1 never rank | 2 rank even for tiny maps
  int roi_fwd_order_mink;  // smallest K that gets the ranking pre-pass (0 = default)
  int roi_fwd_records;     // 0 auto (per-ROI sample records from the pre-pass when a workspace is given, static staging offsets) | 1 off | 2 records only
  int roi_fwd_ct;          // channels per workgroup of the fast forward (0 = default)
  int dcn_col2im;          // 0 auto | 1 gather | 2 scatter | 3 ell
  int dcn_fused;           // 0 auto | 1 force | 2 off
  int dcn_gather_xcd;      // 0 auto (XCD-contiguous block order) | 1 plain block order
  int dcn_nhwc;            // 0 auto (channels-last pipeline where supported) | 2 off (reference-layout kernels)
  int dcn_ell_build;       // inverted index of the deformable-conv input gradient: 0 tile-owner build (LDS lists) | 1 scatter build (global atomics + sort launch)
};
DetopsTuning& detops_tuning();

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
